// shape_ab.hip -- interleaved A/B of memory access shapes for the fused FIR kernel, WITH the kernel's other
// constraints in place: 4 workgroups of 128 lanes per CU (LDS-limited like the real kernel), a persistent loop with
// the next 16 KiB block prefetched into registers, and ~600 packed VALU instructions per wave between the arrival of
// a block and its stores (the transform's issue time; tools/shape_lab.hip used s_sleep, which leaves the issue slots
// free).  Question: does the width of the accesses (8 B vs 16 B per lane) or spreading the loads over the block's
// compute change what the shape streams?
//   hipcc -O3 --offload-arch=gfx950 tools/shape_ab.hip -o tools/bin/shape_ab && tools/bin/shape_ab [rounds=7]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// N packed fused multiply-adds on 8 independent accumulators (no memory, no LDS)
template <int N> __device__ __forceinline__ void burn(f2 (&acc)[8], f2 k)
{
#pragma unroll 1
    for (int i = 0; i < N / 8; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(acc[j]) : "v"(k));
    }
}

constexpr int kLds = 34816;      // bytes per workgroup: 4 workgroups per CU, as the real kernel

// W = 8: 16 x 8-byte accesses per lane at 1 KiB stride; W = 16: 8 x 16-byte accesses at 2 KiB stride
// SPREAD: the prefetch is issued in four instalments between four quarters of the compute
template <int W, bool SPREAD, int BURN>
__global__ __launch_bounds__(128) void walk(const char *in, char *out, long long nblocks, float kf)
{
    __shared__ char pad[kLds];
    constexpr int NA = 16384 / (128 * W);                 // accesses per lane
    typedef typename std::conditional<W == 8, u32x2, u32x4>::type V;
    V cur[NA], nxt[NA];
    const int tid = threadIdx.x;
    if (kf == 12345.f) pad[tid] = 1;                      // keeps the allocation
    f2 acc[8]; for (int j = 0; j < 8; j++) acc[j] = f2{(float)tid, (float)j};
    const f2 k = {kf, kf};
    auto rsrc_in = [&](long long b) {
        const int bytes = b < nblocks ? 16384 : 0;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(in + (b < nblocks ? b : 0) * 16384), 0, bytes, 0x00020000);
    };
    auto ld = [&](decltype(rsrc_in(0)) rs, int m0, int m1) {
#pragma unroll
        for (int m = 0; m < NA; m++) if (m >= m0 && m < m1) {
            if constexpr (W == 8) nxt[m] = __builtin_amdgcn_raw_buffer_load_b64(rs, (tid + m * 128) * 8, 0, 2);
            else nxt[m] = __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + m * 128) * 16, 0, 2);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    { auto rs = rsrc_in(blockIdx.x); ld(rs, 0, NA); }
    for (long long b = blockIdx.x; b < nblocks; b += gridDim.x) {
#pragma unroll
        for (int m = 0; m < NA; m++) cur[m] = nxt[m];
        auto ri = rsrc_in(b + gridDim.x);
        auto ro = __builtin_amdgcn_make_buffer_rsrc(out + b * 16384, 0, 16384, 0x00020000);
        if constexpr (SPREAD) {
#pragma unroll
            for (int q = 0; q < 4; q++) { ld(ri, q * NA / 4, (q + 1) * NA / 4); burn<BURN / 4>(acc, k); __builtin_amdgcn_sched_barrier(0); }
        } else {
            ld(ri, 0, NA);
            burn<BURN>(acc, k);
            __builtin_amdgcn_sched_barrier(0);
        }
        // fold the accumulators into the data so that the compute is live (kf = 0 in the runs: the data is unchanged)
        f2 t = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
        const unsigned fold = (t.x == 12345.f && t.y == 54321.f) ? 1u : 0u;
#pragma unroll
        for (int m = 0; m < NA; m++) {
            V v = cur[m]; v.x ^= fold;
            if constexpr (W == 8) __builtin_amdgcn_raw_buffer_store_b64(v, ro, (tid + m * 128) * 8, 0, 18);
            else __builtin_amdgcn_raw_buffer_store_b128(v, ro, (tid + m * 128) * 16, 0, 18);
        }
    }
}

// reference point: the element-wise copy shape (one 16-byte access per lane, the whole grid)
__global__ __launch_bounds__(256) void clone16(const u32x4 *in, u32x4 *out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

struct Var { const char *name; void (*fn)(const char *, char *, long long, float); int grid2q; };

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 7;
    const size_t bytes = (size_t)128 << 20;      // 16 Mi samples, as C3
    const int NB = 6;
    char *A[NB], *B[NB];
    for (int i = 0; i < NB; i++) { CK(hipMalloc(&A[i], bytes)); CK(hipMalloc(&B[i], bytes)); CK(hipMemset(A[i], 1, bytes)); }
    const long long nblocks = bytes / 16384;
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    std::vector<Var> vars = {
        {"8 B  burst  burn 600", walk<8, false, 600>, 768},   {"16 B burst  burn 600", walk<16, false, 600>, 768},
        {"8 B  spread burn 600", walk<8, true, 600>, 768},    {"16 B spread burn 600", walk<16, true, 600>, 768},
        {"8 B  burst  burn 1200", walk<8, false, 1200>, 768}, {"16 B burst  burn 1200", walk<16, false, 1200>, 768},
        {"8 B  burst  burn 0", walk<8, false, 0>, 768},       {"16 B burst  burn 0", walk<16, false, 0>, 768},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int L = 40;
    for (int mode = 0; mode < 2; mode++) {          // 0: one queue, full grid; 1: two queues, 3/4 grids
        std::vector<std::vector<float>> t(vars.size());
        for (int r = 0; r < rounds + 1; r++) {
            for (size_t v = 0; v < vars.size(); v++) {
                auto go = [&](int i) {
                    hipStream_t s = (mode && (i & 1)) ? s2 : s1;
                    hipLaunchKernelGGL(vars[v].fn, dim3(mode ? vars[v].grid2q : 1024), dim3(128), 0, s, A[i % NB], B[i % NB], nblocks, 0.f);
                };
                for (int i = 0; i < 10; i++) go(i);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, s1));
                for (int i = 0; i < L; i++) go(i);
                CK(hipStreamSynchronize(s2));
                CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r) t[v].push_back(ms / L * 1e3f);      // round 0 = warm-up
            }
        }
        printf("== %s, %d interleaved rounds x %d launches (us per 256 MiB of copy; median, min) ==\n", mode ? "two queues, 3/4 grid" : "one queue, full grid", rounds, L);
        for (size_t v = 0; v < vars.size(); v++) {
            std::sort(t[v].begin(), t[v].end());
            const float med = t[v][t[v].size() / 2];
            printf("  %-24s %7.2f %7.2f   %7.1f GB/s\n", vars[v].name, med, t[v][0], 2.0 * bytes / med / 1e3);
        }
    }
    {
        std::vector<float> t;
        for (int r = 0; r < rounds; r++) {
            for (int i = 0; i < 10; i++) clone16<<<(unsigned)(bytes / 4096), 256, 0, s1>>>((const u32x4 *)A[i % NB], (u32x4 *)B[i % NB]);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s1));
            for (int i = 0; i < L; i++) clone16<<<(unsigned)(bytes / 4096), 256, 0, s1>>>((const u32x4 *)A[i % NB], (u32x4 *)B[i % NB]);
            CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms / L * 1e3f);
        }
        std::sort(t.begin(), t.end());
        printf("  %-24s %7.2f %7.2f   %7.1f GB/s\n", "clone (16 B per lane)", t[t.size() / 2], t[0], 2.0 * bytes / t[t.size() / 2] / 1e3);
    }
    return 0;
}
