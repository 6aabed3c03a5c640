// micro-benchmark: issue rate of packed vs scalar f32 VALU ops on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void k(float* out, int iters)
{
    v2f a0={1.f,2.f},a1={3.f,4.f},a2={5.f,6.f},a3={7.f,8.f},a4={1.5f,2.5f},a5={3.5f,4.5f},a6={5.5f,6.5f},a7={7.5f,8.5f};
    v2f b={1e-9f*threadIdx.x,1e-9f};
    for (int i=0;i<iters;i++){
        if constexpr (MODE==0) { // 8 independent pk_add chains, 16 groups = 128 instr
            REP16(asm volatile("v_pk_add_f32 %0,%0,%8\n v_pk_add_f32 %1,%1,%8\n v_pk_add_f32 %2,%2,%8\n v_pk_add_f32 %3,%3,%8\n v_pk_add_f32 %4,%4,%8\n v_pk_add_f32 %5,%5,%8\n v_pk_add_f32 %6,%6,%8\n v_pk_add_f32 %7,%7,%8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b));)
        } else if constexpr (MODE==1) { // 8 scalar adds
            REP16(asm volatile("v_add_f32 %0,%0,%8\n v_add_f32 %1,%1,%8\n v_add_f32 %2,%2,%8\n v_add_f32 %3,%3,%8\n v_add_f32 %4,%4,%8\n v_add_f32 %5,%5,%8\n v_add_f32 %6,%6,%8\n v_add_f32 %7,%7,%8" : "+v"(a0.x),"+v"(a1.x),"+v"(a2.x),"+v"(a3.x),"+v"(a4.x),"+v"(a5.x),"+v"(a6.x),"+v"(a7.x) : "v"(b.x));)
        } else if constexpr (MODE==2) { // pk_fma
            REP16(asm volatile("v_pk_fma_f32 %0,%0,%8,%8\n v_pk_fma_f32 %1,%1,%8,%8\n v_pk_fma_f32 %2,%2,%8,%8\n v_pk_fma_f32 %3,%3,%8,%8\n v_pk_fma_f32 %4,%4,%8,%8\n v_pk_fma_f32 %5,%5,%8,%8\n v_pk_fma_f32 %6,%6,%8,%8\n v_pk_fma_f32 %7,%7,%8,%8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b));)
        } else if constexpr (MODE==3) { // scalar fma
            REP16(asm volatile("v_fma_f32 %0,%0,%8,%8\n v_fma_f32 %1,%1,%8,%8\n v_fma_f32 %2,%2,%8,%8\n v_fma_f32 %3,%3,%8,%8\n v_fma_f32 %4,%4,%8,%8\n v_fma_f32 %5,%5,%8,%8\n v_fma_f32 %6,%6,%8,%8\n v_fma_f32 %7,%7,%8,%8" : "+v"(a0.x),"+v"(a1.x),"+v"(a2.x),"+v"(a3.x),"+v"(a4.x),"+v"(a5.x),"+v"(a6.x),"+v"(a7.x) : "v"(b.x));)
        } else { // pk_add with op_sel modifiers
            REP16(asm volatile("v_pk_add_f32 %0,%0,%8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]\n v_pk_add_f32 %1,%1,%8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]\n v_pk_add_f32 %2,%2,%8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]\n v_pk_add_f32 %3,%3,%8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]\n v_pk_add_f32 %4,%4,%8 op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_add_f32 %5,%5,%8 op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_add_f32 %6,%6,%8 op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_add_f32 %7,%7,%8 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b));)
        }
    }
    out[blockIdx.x*blockDim.x+threadIdx.x]=a0.x+a1.x+a2.x+a3.x+a4.x+a5.x+a6.x+a7.x+a0.y+a1.y+a7.y;
}
template<int MODE> void run(const char* name, int wg_per_cu, int threads){
    float* out; hipMalloc(&out, 256*8*1024*4);
    int iters=2000; int grid=256*wg_per_cu;
    hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid,threads>>>(out,10); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<grid,threads>>>(out,iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms,e0,e1);
    double instr_per_wave = (double)iters*128;
    double waves_per_simd = (double)wg_per_cu*threads/64/4;
    // cycles per wave-instruction per SIMD assuming 2.4 GHz
    double ns_per = ms*1e6/ (instr_per_wave*waves_per_simd);
    printf("%-14s wg/cu=%d thr=%d waves/simd=%.1f : %.3f ms, %.2f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)\n",name,wg_per_cu,threads,waves_per_simd,ms,ns_per,ns_per*2.4);
    hipFree(out);
}
int main(){
    for (int occ : {1,2,4}) {
        run<0>("v_pk_add_f32",occ,256); run<4>("v_pk_add opsel",occ,256); run<1>("v_add_f32",occ,256); run<2>("v_pk_fma_f32",occ,256); run<3>("v_fma_f32",occ,256);
    }
    return 0;
}
