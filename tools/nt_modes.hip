// Cache-policy bits of buffer loads/stores (aux: bit0 sc0, bit1 nt, bit4 sc1) on a streaming copy:
// which combination moves 2 x 256 MiB fastest on MI355X?   hipcc --offload-arch=gfx950 -O3 nt_modes.hip -o nt_modes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int AL, int AS>
__global__ __launch_bounds__(256) void copy_k(const float4 *in, float4 *out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // one 4 KiB tile per workgroup through a descriptor
    auto ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(in + (size_t)blockIdx.x * 256), 0, 4096, 0x00020000);
    auto ro = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)blockIdx.x * 256, 0, 4096, 0x00020000);
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ri, threadIdx.x * 16, 0, AL);
    __builtin_amdgcn_raw_buffer_store_b128(v, ro, threadIdx.x * 16, 0, AS);
}
template <int AL, int AS>
float run(const float4 *in[], float4 *out[], size_t n)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<float> ts;
    for (int r = 0; r < 5; r++) {
        for (int i = 0; i < 5; i++) copy_k<AL, AS><<<(unsigned)(n / 256), 256>>>(in[i % 4], out[i % 4], n);
        (void)hipEventRecord(e0);
        for (int i = 0; i < 40; i++) copy_k<AL, AS><<<(unsigned)(n / 256), 256>>>(in[i % 4], out[i % 4], n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms / 40);
    }
    std::sort(ts.begin(), ts.end());
    return ts[2];
}
#define RUN(AL, AS) { float ms = run<AL, AS>(in, out, n); printf("load aux %2d  store aux %2d : %7.1f us  %7.1f GB/s\n", AL, AS, ms * 1e3, 2.0 * n * 16 / ms / 1e6); }
int main()
{
    const size_t n = (size_t)1 << 24;          // 16 Mi float4 = 256 MiB
    const float4 *in[4]; float4 *out[4];
    for (int i = 0; i < 4; i++) { float4 *a, *b; (void)hipMalloc(&a, n * 16); (void)hipMalloc(&b, n * 16); (void)hipMemset(a, 1, n * 16); in[i] = a; out[i] = b; }
    RUN(0, 0) RUN(2, 2) RUN(0, 2) RUN(2, 0) RUN(3, 3) RUN(18, 18) RUN(19, 19) RUN(2, 18) RUN(2, 19) RUN(18, 2) RUN(16, 16) RUN(1, 1) RUN(17, 17) RUN(2, 3) RUN(3, 2)
    RUN(0, 0)
    return 0;
}
