// micro-benchmark: streaming copy out[i] = in[i] with 8-byte vs 16-byte lanes, in the access
// shape of the FFT kernels (a workgroup of 128 lanes walks a 16 KiB block: 16 x 8 B per lane at
// stride 1 KiB, or 8 x 16 B per lane at stride 2 KiB)
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename V, int PER>
__global__ __launch_bounds__(128) void blockcopy(const V* __restrict__ in, V* __restrict__ out, size_t nblocks)
{
    for (size_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const V* s = in + b * (128 * PER) + threadIdx.x;
        V* d = out + b * (128 * PER) + threadIdx.x;
        V v[PER];
#pragma unroll
        for (int m = 0; m < PER; m++) v[m] = s[m * 128];
#pragma unroll
        for (int m = 0; m < PER; m++) d[m * 128] = v[m];
    }
}
template <class F> float timeit(F f){ hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1); f(0); f(1); hipDeviceSynchronize(); float best=1e9; for(int r=0;r<5;r++){ hipEventRecord(e0); for(int i=0;i<10;i++) f(i); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); if(ms/10<best) best=ms/10;} return best; }
int main(){
    const size_t bytes = (size_t)256 << 20; const int NB = 4; void *A[NB], *B[NB];
    for (int i=0;i<NB;i++){ hipMalloc(&A[i], bytes); hipMalloc(&B[i], bytes); hipMemset(A[i],1,bytes); }
    const size_t nblocks = bytes / 16384;
    for (int grid : {1024, 2048, 16384}) {
        float t8 = timeit([&](int i){ blockcopy<float2,16><<<grid,128>>>((const float2*)A[i%NB],(float2*)B[i%NB],nblocks); });
        float t16 = timeit([&](int i){ blockcopy<float4,8><<<grid,128>>>((const float4*)A[i%NB],(float4*)B[i%NB],nblocks); });
        printf("grid=%5d  8 B lanes: %7.1f us %7.1f GB/s   16 B lanes: %7.1f us %7.1f GB/s\n", grid, t8*1e3, 2.0*bytes/t8/1e6, t16*1e3, 2.0*bytes/t16/1e6);
    }
    return 0;
}
