// shape_lab.hip -- how fast does the memory access SHAPE of the fused FIR kernel stream, without any arithmetic?
// A persistent grid of 128-lane workgroups walks 16 KiB blocks (16 x 8 B per lane at a stride of 1 KiB, the
// kernel's window layout), loads block k+D while block k is stored.  Variants: cache-policy bits, prefetch
// depth D, workgroups per CU, an artificial delay between load and store (the transform's latency).
//   hipcc -O3 --offload-arch=gfx950 tools/shape_lab.hip -o tools/bin/shape_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// D = prefetch depth in blocks (1 = the FIR kernel's), AL/AS = aux bits of loads / stores, SPIN = s_sleep units
// between the arrival of a block and its stores
template <int D, int AL, int AS, int SPIN>
__global__ __launch_bounds__(128) void walk(const float2 *in, float2 *out, long long nblocks)
{
    u32x2 buf[D + 1][16];
    const int tid = threadIdx.x;
    auto load = [&](int slot, long long b) {
        const int bytes = b < nblocks ? 16384 : 0;
        auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(in + (b < nblocks ? b : 0) * 2048), 0, bytes, 0x00020000);
#pragma unroll
        for (int m = 0; m < 16; m++) buf[slot][m] = __builtin_amdgcn_raw_buffer_load_b64(rs, (tid + m * 128) * 8, 0, AL);
    };
#pragma unroll
    for (int d = 0; d < D; d++) load(d, (long long)blockIdx.x + (long long)d * gridDim.x);
    long long b = blockIdx.x;
    // the slots rotate at compile time: unroll D+1 iterations per trip
    for (;; ) {
#pragma unroll
        for (int u = 0; u <= D; u++) {
            if (b >= nblocks) return;
            load((u + D) % (D + 1), b + (long long)D * gridDim.x);
            if (SPIN) __builtin_amdgcn_s_sleep(SPIN);
            auto rs = __builtin_amdgcn_make_buffer_rsrc(out + b * 2048, 0, 16384, 0x00020000);
#pragma unroll
            for (int m = 0; m < 16; m++) __builtin_amdgcn_raw_buffer_store_b64(buf[u][m], rs, (tid + m * 128) * 8, 0, AS);
            b += gridDim.x;
        }
    }
}

// SPLIT variant: the next block's 16 loads and this block's 16 stores issued in Q groups spread over the block's
// (artificial) latency instead of two bursts
template <int Q, int AL, int AS, int SPIN>
__global__ __launch_bounds__(128) void walk_split(const float2 *in, float2 *out, long long nblocks)
{
    u32x2 cur[16], nxt[16];
    const int tid = threadIdx.x;
    auto rsrc_in = [&](long long b) {
        const int bytes = b < nblocks ? 16384 : 0;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(in + (b < nblocks ? b : 0) * 2048), 0, bytes, 0x00020000);
    };
    {
        auto rs = rsrc_in(blockIdx.x);
#pragma unroll
        for (int m = 0; m < 16; m++) nxt[m] = __builtin_amdgcn_raw_buffer_load_b64(rs, (tid + m * 128) * 8, 0, AL);
    }
    for (long long b = blockIdx.x; b < nblocks; b += gridDim.x) {
#pragma unroll
        for (int m = 0; m < 16; m++) cur[m] = nxt[m];
        auto ri = rsrc_in(b + gridDim.x);
        auto ro = __builtin_amdgcn_make_buffer_rsrc(out + b * 2048, 0, 16384, 0x00020000);
#pragma unroll
        for (int q = 0; q < Q; q++) {
#pragma unroll
            for (int m = q * (16 / Q); m < (q + 1) * (16 / Q); m++) nxt[m] = __builtin_amdgcn_raw_buffer_load_b64(ri, (tid + m * 128) * 8, 0, AL);
            if (SPIN) __builtin_amdgcn_s_sleep(SPIN / Q);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = q * (16 / Q); m < (q + 1) * (16 / Q); m++) __builtin_amdgcn_raw_buffer_store_b64(cur[m], ro, (tid + m * 128) * 8, 0, AS);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// one block per workgroup, no loop (what the hardware dispatcher does with the same bytes)
template <int AL, int AS>
__global__ __launch_bounds__(128) void oneshot(const float2 *in, float2 *out, long long nblocks)
{
    const int tid = threadIdx.x;
    const long long b = blockIdx.x;
    auto ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(in + b * 2048), 0, 16384, 0x00020000);
    auto ro = __builtin_amdgcn_make_buffer_rsrc(out + b * 2048, 0, 16384, 0x00020000);
    u32x2 v[16];
#pragma unroll
    for (int m = 0; m < 16; m++) v[m] = __builtin_amdgcn_raw_buffer_load_b64(ri, (tid + m * 128) * 8, 0, AL);
#pragma unroll
    for (int m = 0; m < 16; m++) __builtin_amdgcn_raw_buffer_store_b64(v[m], ro, (tid + m * 128) * 8, 0, AS);
}

// the same 16 KiB blocks with 16-byte accesses: 8 per lane at a stride of 2 KiB (persistent, next block prefetched)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int AL, int AS, int SPIN>
__global__ __launch_bounds__(128) void walk16(const float4 *in, float4 *out, long long nblocks)
{
    u32x4 cur[8], nxt[8];
    const int tid = threadIdx.x;
    auto rsrc_in = [&](long long b) {
        const int bytes = b < nblocks ? 16384 : 0;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(in + (b < nblocks ? b : 0) * 1024), 0, bytes, 0x00020000);
    };
    {
        auto rs = rsrc_in(blockIdx.x);
#pragma unroll
        for (int m = 0; m < 8; m++) nxt[m] = __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + m * 128) * 16, 0, AL);
    }
    for (long long b = blockIdx.x; b < nblocks; b += gridDim.x) {
#pragma unroll
        for (int m = 0; m < 8; m++) cur[m] = nxt[m];
        auto ri = rsrc_in(b + gridDim.x);
        auto ro = __builtin_amdgcn_make_buffer_rsrc(out + b * 1024, 0, 16384, 0x00020000);
#pragma unroll
        for (int m = 0; m < 8; m++) nxt[m] = __builtin_amdgcn_raw_buffer_load_b128(ri, (tid + m * 128) * 16, 0, AL);
        if (SPIN) __builtin_amdgcn_s_sleep(SPIN);
#pragma unroll
        for (int m = 0; m < 8; m++) __builtin_amdgcn_raw_buffer_store_b128(cur[m], ro, (tid + m * 128) * 16, 0, AS);
    }
}
template <int AL, int AS>
__global__ __launch_bounds__(128) void oneshot16(const float4 *in, float4 *out, long long nblocks)
{
    const int tid = threadIdx.x;
    const long long b = blockIdx.x;
    auto ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(in + b * 1024), 0, 16384, 0x00020000);
    auto ro = __builtin_amdgcn_make_buffer_rsrc(out + b * 1024, 0, 16384, 0x00020000);
    u32x4 v[8];
#pragma unroll
    for (int m = 0; m < 8; m++) v[m] = __builtin_amdgcn_raw_buffer_load_b128(ri, (tid + m * 128) * 16, 0, AL);
#pragma unroll
    for (int m = 0; m < 8; m++) __builtin_amdgcn_raw_buffer_store_b128(v[m], ro, (tid + m * 128) * 16, 0, AS);
}

template <class F> float timeit(F f)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; i++) f(i);
    (void)hipDeviceSynchronize();
    std::vector<float> ts;
    for (int r = 0; r < 5; r++) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 40; i++) f(i);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms / 40);
    }
    std::sort(ts.begin(), ts.end());
    return ts[2];
}

int main()
{
    const size_t bytes = (size_t)128 << 20;      // 16 Mi samples, as C3
    const int NB = 6;
    float2 *A[NB], *B[NB];
    for (int i = 0; i < NB; i++) { CK(hipMalloc(&A[i], bytes)); CK(hipMalloc(&B[i], bytes)); CK(hipMemset(A[i], 1, bytes)); }
    const long long nblocks = bytes / 16384;
    hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
#define RUN(D, AL, AS, SPIN, GRID, TWOQ)                                                                              \
    {                                                                                                                 \
        float ms = timeit([&](int i) { walk<D, AL, AS, SPIN><<<GRID, 128, 0, (TWOQ && (i & 1)) ? s2 : 0>>>(A[i % NB], B[i % NB], nblocks); }); \
        (void)hipStreamSynchronize(s2);                                                                               \
        printf("depth %d  load aux %2d  store aux %2d  sleep %3d  grid %5d  %s : %6.1f us  %6.1f GB/s\n", D, AL, AS, SPIN, GRID, TWOQ ? "2q" : "1q", ms * 1e3, 2.0 * bytes / ms / 1e6); \
    }
    RUN(1, 0, 0, 0, 1024, 0) RUN(1, 2, 18, 0, 1024, 0) RUN(1, 2, 18, 0, 1024, 1)
    RUN(2, 2, 18, 0, 1024, 0) RUN(2, 2, 18, 0, 1024, 1) RUN(3, 2, 18, 0, 1024, 0)
    RUN(1, 2, 18, 0, 2048, 0) RUN(1, 2, 18, 0, 2048, 1) RUN(2, 2, 18, 0, 2048, 0)
    RUN(1, 2, 18, 64, 1024, 0) RUN(1, 2, 18, 64, 1024, 1) RUN(2, 2, 18, 64, 1024, 0) RUN(2, 2, 18, 64, 1024, 1)
    RUN(1, 2, 18, 120, 1024, 0) RUN(1, 2, 18, 120, 1024, 1) RUN(2, 2, 18, 120, 1024, 0) RUN(2, 2, 18, 120, 1024, 1) RUN(3, 2, 18, 120, 1024, 1)
    RUN(1, 2, 2, 0, 1024, 0) RUN(1, 0, 18, 0, 1024, 0) RUN(1, 2, 18, 0, 1024, 0)
#define RUNS(Q, SPIN, TWOQ)                                                                                           \
    {                                                                                                                 \
        float ms = timeit([&](int i) { walk_split<Q, 2, 18, SPIN><<<1024, 128, 0, (TWOQ && (i & 1)) ? s2 : 0>>>(A[i % NB], B[i % NB], nblocks); }); \
        (void)hipStreamSynchronize(s2);                                                                               \
        printf("split %d  sleep %3d  grid 1024  %s : %6.1f us  %6.1f GB/s\n", Q, SPIN, TWOQ ? "2q" : "1q", ms * 1e3, 2.0 * bytes / ms / 1e6); \
    }
    RUNS(1, 120, 0) RUNS(2, 120, 0) RUNS(4, 120, 0) RUNS(8, 120, 0) RUNS(16, 120, 0)
    RUNS(1, 120, 1) RUNS(4, 120, 1) RUNS(16, 120, 1)
    RUNS(1, 0, 0) RUNS(4, 0, 0) RUNS(16, 0, 0)
    {
        float ms = timeit([&](int i) { oneshot<2, 18><<<(unsigned)nblocks, 128>>>(A[i % NB], B[i % NB], nblocks); });
        printf("one block per workgroup, %lld workgroups : %6.1f us  %6.1f GB/s\n", nblocks, ms * 1e3, 2.0 * bytes / ms / 1e6);
    }
#define RUN16(SPIN, TWOQ)                                                                                             \
    {                                                                                                                 \
        float ms = timeit([&](int i) { walk16<2, 18, SPIN><<<1024, 128, 0, (TWOQ && (i & 1)) ? s2 : 0>>>((const float4 *)A[i % NB], (float4 *)B[i % NB], nblocks); }); \
        (void)hipStreamSynchronize(s2);                                                                               \
        printf("16-byte lanes (8 per lane)  sleep %3d  grid 1024  %s : %6.1f us  %6.1f GB/s\n", SPIN, TWOQ ? "2q" : "1q", ms * 1e3, 2.0 * bytes / ms / 1e6); \
    }
    RUN16(0, 0) RUN16(0, 1) RUN16(120, 0) RUN16(120, 1)
    RUN(1, 2, 18, 0, 1024, 0) RUN(1, 2, 18, 120, 1024, 1)
    {
        float ms = timeit([&](int i) { oneshot16<2, 18><<<(unsigned)nblocks, 128>>>((const float4 *)A[i % NB], (float4 *)B[i % NB], nblocks); });
        printf("one block per workgroup, 16-byte lanes : %6.1f us  %6.1f GB/s\n", ms * 1e3, 2.0 * bytes / ms / 1e6);
        ms = timeit([&](int i) { oneshot<2, 18><<<(unsigned)nblocks, 128>>>(A[i % NB], B[i % NB], nblocks); });
        printf("one block per workgroup,  8-byte lanes : %6.1f us  %6.1f GB/s\n", ms * 1e3, 2.0 * bytes / ms / 1e6);
    }
    return 0;
}
