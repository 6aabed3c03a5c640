// micro-benchmark: dependent-chain latency of packed f32 VALU ops on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int DIST>
__global__ void k(float* out, int iters)
{
    v2f a0={1.f,2.f},a1={3.f,4.f},a2={5.f,6.f},a3={7.f,8.f};
    v2f b={1e-9f*threadIdx.x,1e-9f};
    for (int i=0;i<iters;i++){
        if constexpr (DIST==1) { REP16(asm volatile("v_pk_add_f32 %0,%0,%1\n v_pk_add_f32 %0,%0,%1\n v_pk_add_f32 %0,%0,%1\n v_pk_add_f32 %0,%0,%1\n v_pk_add_f32 %0,%0,%1\n v_pk_add_f32 %0,%0,%1\n v_pk_add_f32 %0,%0,%1\n v_pk_add_f32 %0,%0,%1" : "+v"(a0) : "v"(b));) }
        else if constexpr (DIST==2) { REP16(asm volatile("v_pk_add_f32 %0,%0,%2\n v_pk_add_f32 %1,%1,%2\n v_pk_add_f32 %0,%0,%2\n v_pk_add_f32 %1,%1,%2\n v_pk_add_f32 %0,%0,%2\n v_pk_add_f32 %1,%1,%2\n v_pk_add_f32 %0,%0,%2\n v_pk_add_f32 %1,%1,%2" : "+v"(a0),"+v"(a1) : "v"(b));) }
        else { REP16(asm volatile("v_pk_add_f32 %0,%0,%4\n v_pk_add_f32 %1,%1,%4\n v_pk_add_f32 %2,%2,%4\n v_pk_add_f32 %3,%3,%4\n v_pk_add_f32 %0,%0,%4\n v_pk_add_f32 %1,%1,%4\n v_pk_add_f32 %2,%2,%4\n v_pk_add_f32 %3,%3,%4" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3) : "v"(b));) }
    }
    out[blockIdx.x*blockDim.x+threadIdx.x]=a0.x+a1.x+a2.x+a3.x+a0.y;
}
template<int DIST> void run(int wg_per_cu){
    float* out; hipMalloc(&out, 256*8*1024*4);
    int iters=2000, threads=256; int grid=256*wg_per_cu;
    hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<DIST><<<grid,threads>>>(out,10); hipDeviceSynchronize();
    hipEventRecord(e0); k<DIST><<<grid,threads>>>(out,iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms,e0,e1);
    double ns_per_wave_instr = ms*1e6/((double)iters*128);
    printf("dep distance %d, waves/simd=%d: %.2f ns per instr per wave (%.1f cyc@2.0GHz), SIMD-level %.2f ns/instr\n",DIST,wg_per_cu,ns_per_wave_instr,ns_per_wave_instr*2.0,ns_per_wave_instr/wg_per_cu);
    hipFree(out);
}
int main(){ for (int occ : {1,2,3}) { run<1>(occ); run<2>(occ); run<4>(occ);} return 0; }
