// micro-benchmark: streaming a[i] = a[i] + b[i] (24 B/sample) variants on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void add_k(float4* __restrict__ a, const float4* __restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        float4 x[UNROLL], y[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { if (NT) { f4 t = __builtin_nontemporal_load((const f4*)&a[i + u * stride]); x[u] = make_float4(t.x,t.y,t.z,t.w); f4 s2 = __builtin_nontemporal_load((const f4*)&b[i + u * stride]); y[u] = make_float4(s2.x,s2.y,s2.z,s2.w);} else { x[u] = a[i + u * stride]; y[u] = b[i + u * stride]; } }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { float4 r = make_float4(x[u].x + y[u].x, x[u].y + y[u].y, x[u].z + y[u].z, x[u].w + y[u].w); if (NT) { f4 t = {r.x,r.y,r.z,r.w}; __builtin_nontemporal_store(t, (f4*)&a[i + u * stride]); } else a[i + u * stride] = r; }
    }
    for (; i < n; i += stride) { float4 x = a[i], y = b[i]; a[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w); }
}
// block-contiguous variant: each block owns a contiguous tile (UNROLL*256 float4) per iteration
template <int UNROLL>
__global__ __launch_bounds__(256) void add_tile(float4* __restrict__ a, const float4* __restrict__ b, size_t n)
{
    const size_t tile = (size_t)UNROLL * 256;
    for (size_t base = (size_t)blockIdx.x * tile; base < n; base += (size_t)gridDim.x * tile) {
        float4 x[UNROLL], y[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { size_t i = base + u * 256 + threadIdx.x; if (i < n) { x[u] = a[i]; y[u] = b[i]; } }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { size_t i = base + u * 256 + threadIdx.x; if (i < n) a[i] = make_float4(x[u].x + y[u].x, x[u].y + y[u].y, x[u].z + y[u].z, x[u].w + y[u].w); }
    }
}
template <class F> float timeit(F f){ hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1); f(0); f(1); hipDeviceSynchronize(); float best=1e9; for(int r=0;r<5;r++){ hipEventRecord(e0); for(int i=0;i<10;i++) f(i); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); if(ms/10<best) best=ms/10;} return best; }
int main(){
    const size_t n = (size_t)1 << 24;   // float4 = 2 samples; 2^24 float4 = 256 MiB
    const int NB = 4; float4 *A[NB], *B[NB];
    for (int i=0;i<NB;i++){ hipMalloc(&A[i], n*16); hipMalloc(&B[i], n*16); hipMemset(A[i],0,n*16); hipMemset(B[i],0,n*16);} 
    auto rep=[&](const char* name, float ms){ printf("%-34s %8.1f us %8.1f GB/s\n", name, ms*1e3, 3.0*n*16/ms/1e6); };
    for (int cap : {4, 8, 16, 32, 64}) {
        int grid = 256*cap; char nm[64];
        snprintf(nm,64,"stride U4 cap=%d",cap); rep(nm, timeit([&](int i){ add_k<4,false><<<grid,256>>>(A[i%NB],B[i%NB],n); }));
        snprintf(nm,64,"stride U8 cap=%d",cap); rep(nm, timeit([&](int i){ add_k<8,false><<<grid,256>>>(A[i%NB],B[i%NB],n); }));
        snprintf(nm,64,"stride U4 NT cap=%d",cap); rep(nm, timeit([&](int i){ add_k<4,true><<<grid,256>>>(A[i%NB],B[i%NB],n); }));
        snprintf(nm,64,"tile U4 cap=%d",cap); rep(nm, timeit([&](int i){ add_tile<4><<<grid,256>>>(A[i%NB],B[i%NB],n); }));
        snprintf(nm,64,"tile U8 cap=%d",cap); rep(nm, timeit([&](int i){ add_tile<8><<<grid,256>>>(A[i%NB],B[i%NB],n); }));
    }
    { int grid=(int)(n/256/4); rep("tile U4 one tile per WG", timeit([&](int i){ add_tile<4><<<grid,256>>>(A[i%NB],B[i%NB],n); })); }
    { int grid=(int)(n/256/1); rep("tile U1 one tile per WG", timeit([&](int i){ add_tile<1><<<grid,256>>>(A[i%NB],B[i%NB],n); })); }
    return 0;
}
