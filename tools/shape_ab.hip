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


// DMA: the next block goes straight into a 16 KiB LDS landing image (buffer_load_dwordx4 ... lds, 1 KiB per
// wave-instruction, 8 per wave and block instead of 16 register loads) and is read from there with ds_read_b64 at
// the start of its own iteration: no prefetch registers, no register copy per block.  The image is lane-linear per
// wave (dest = base + lane * 16), so the permutation sits on the SOURCE side: instruction j of wave w fetches the
// 512-byte runs of window rows m = 2j and 2j+1 that this very wave reads in pass 0 (element tid + m*128) -- no
// other wave ever reads what a wave landed, so its own vmcnt orders the ds_reads and no barrier is needed.
typedef __attribute__((address_space(3))) void *lds_vptr;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char *dst, int voff, int soff)
{
#if __HIP_DEVICE_COMPILE__
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)dst, 16, voff, soff, 0, 2);
#endif
}
template <bool SPREAD, int BURN>
__global__ __launch_bounds__(128) void walk_dma(const char *in, char *out, long long nblocks, float kf)
{
    __shared__ __attribute__((aligned(1024))) char land[kLds];      // 16 KiB used; the rest keeps 4 workgroups per CU
    u32x2 cur[16];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    if (kf == 12345.f) land[tid + 20000] = 1;
    f2 acc[8]; for (int j = 0; j < 8; j++) acc[j] = f2{(float)tid, (float)j};
    const f2 k = {kf, kf};
    auto rsrc_in = [&](long long b) {
        const int bytes = b < nblocks ? 16384 : 0;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(in + (b < nblocks ? b : 0) * 16384), 0, bytes, 0x00020000);
    };
    const int voff = (l >> 5) * 1024 + w * 512 + (l & 31) * 16;      // + j * 2048
    auto dma = [&](decltype(rsrc_in(0)) rs, int j0, int j1) {
#pragma unroll
        for (int j = 0; j < 8; j++) if (j >= j0 && j < j1)
            dma16(rs, land + w * 8192 + j * 1024, voff, j * 2048);
        __builtin_amdgcn_sched_barrier(0);
    };
    { auto rs = rsrc_in(blockIdx.x); dma(rs, 0, 8); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (long long b = blockIdx.x; b < nblocks; b += gridDim.x) {
#pragma unroll
        for (int m = 0; m < 16; m++) cur[m] = *reinterpret_cast<const u32x2 *>(land + w * 8192 + (m >> 1) * 1024 + (m & 1) * 512 + l * 8);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        auto ri = rsrc_in(b + gridDim.x);
        auto ro = __builtin_amdgcn_make_buffer_rsrc(out + b * 16384, 0, 16384, 0x00020000);
        if constexpr (SPREAD) {
#pragma unroll
            for (int q = 0; q < 4; q++) { dma(ri, q * 2, q * 2 + 2); burn<BURN / 4>(acc, k); __builtin_amdgcn_sched_barrier(0); }
        } else {
            dma(ri, 0, 8);
            burn<BURN>(acc, k);
            __builtin_amdgcn_sched_barrier(0);
        }
        f2 t = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
        const unsigned fold = (t.x == 12345.f && t.y == 54321.f) ? 1u : 0u;
#pragma unroll
        for (int m = 0; m < 16; m++) {
            u32x2 v = cur[m]; v.x ^= fold;
            __builtin_amdgcn_raw_buffer_store_b64(v, ro, (tid + m * 128) * 8, 0, 18);
        }
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // the 8 DMA pieces landed, this block's 16 stores still in flight
    }
}

// reference point: the element-wise copy shape (one 16-byte access per lane, the whole grid)
__global__ __launch_bounds__(256) void clone16(const u32x4 *in, u32x4 *out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

__global__ void fill_idx(unsigned *p, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = (unsigned)(i * 2654435761u); }

// what does an out-of-range lane of an LDS-DMA write?  The landing image is preset to 0xAAAAAAAA, the descriptor covers
// `bytes` of a 1 KiB row, one wave issues one piece; out[] receives the image afterwards.
__global__ __launch_bounds__(64) void oob_probe(const char *in, unsigned *out, int bytes)
{
    __shared__ __attribute__((aligned(1024))) char land[1024];
    const int l = threadIdx.x;
    for (int i = 0; i < 4; i++) reinterpret_cast<unsigned *>(land)[l * 4 + i] = 0xAAAAAAAAu;
    __syncthreads();
    auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(in), 0, bytes, 0x00020000);
    dma16(rs, land, l * 16, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; i++) out[l * 4 + i] = reinterpret_cast<unsigned *>(land)[l * 4 + i];
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
        {"LDS-DMA burst burn 600", walk_dma<false, 600>, 768}, {"LDS-DMA spread burn 600", walk_dma<true, 600>, 768},
        {"LDS-DMA burst burn 1200", walk_dma<false, 1200>, 768}, {"LDS-DMA burst burn 0", walk_dma<false, 0>, 768},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    {   // every shape must be a copy: index pattern through each variant once, compared on the host
        fill_idx<<<(unsigned)(bytes / 4 / 256), 256>>>((unsigned *)A[0], bytes / 4);
        std::vector<unsigned> ha(bytes / 4), hb(bytes / 4);
        CK(hipMemcpy(ha.data(), A[0], bytes, hipMemcpyDeviceToHost));
        for (size_t v = 0; v < vars.size(); v++) {
            CK(hipMemset(B[0], 0, bytes));
            hipLaunchKernelGGL(vars[v].fn, dim3(1000), dim3(128), 0, 0, A[0], B[0], nblocks, 0.f);
            CK(hipMemcpy(hb.data(), B[0], bytes, hipMemcpyDeviceToHost));
            size_t bad = 0; for (size_t i = 0; i < ha.size(); i++) bad += ha[i] != hb[i];
            printf("verify %-26s %s (%zu words differ)\n", vars[v].name, bad ? "WRONG" : "copy ok", bad);
        }
        unsigned *po; CK(hipMalloc(&po, 1024));
        for (int bytes_in : {1024, 1000, 512, 8, 0}) {
            oob_probe<<<1, 64>>>(A[0], po, bytes_in);
            unsigned ho[256]; CK(hipMemcpy(ho, po, 1024, hipMemcpyDeviceToHost));
            int same = 0, zero = 0, kept = 0;
            for (int i = 0; i < 256; i++) { same += ho[i] == ha[i]; zero += ho[i] == 0; kept += ho[i] == 0xAAAAAAAAu; }
            printf("LDS-DMA, descriptor of %4d bytes under a 1 KiB piece: %3d dwords = source, %3d zero, %3d untouched\n", bytes_in, same, zero, kept);
        }
        CK(hipMemset(A[0], 1, bytes));
    }
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
