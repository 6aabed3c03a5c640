// rng_lab.hip -- where does the complex-normal generator (csrc/aeth_rng.h) spend its time, which launch shape suits it,
// and is its device square root the host's sqrtf?  Fill of 2^25 samples (268 MB of stores: 33.6 us at 8 TB/s):
//   stages   philox alone | + r^2 = -2 ln u | + sqrt | philox + (cos, sin) | the whole generator
//   shapes   K pairs per lane (a grid's width apart, or one contiguous run per block), non-temporal or plain stores,
//            stores as they come or all at the end, block size
//   checks   device == host on 2^21 samples; r = sqrt(-2 ln u) on the device against the host's sqrtf for ALL 2^23 u;
//            moments of the samples
// History of the floating-point stage on this lab (profiles/r04_rng_lab.txt): version 2 (Cephes log, compare-and-select
// quadrants, the compiler's correctly rounded sqrt) 73 us, version 3 (csrc/aeth_rng.h today) 55 us one pair per lane,
// 50 us two.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iaether_primitives_amd/csrc tools/rng_lab.hip -o tools/bin/rng_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#define AETH_RNG_FN __host__ __device__ static inline
#include "aeth_rng.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

enum { FULL, PHILOX, R2, SQRT, COSSIN };
template <int VAR, int K, int MAP, bool NT, bool DEFER, int BLOCK>
__global__ __launch_bounds__(BLOCK) void fill(float4 *__restrict__ x, size_t npairs, float scale, uint64_t seed)
{
    const size_t stride = MAP ? BLOCK : (size_t)gridDim.x * BLOCK;
    const size_t p0 = MAP ? (size_t)blockIdx.x * BLOCK * K + threadIdx.x : (size_t)blockIdx.x * BLOCK + threadIdx.x;
    f4 o[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const size_t p = p0 + k * stride;
        uint32_t w[4];
        aeth_rng_draw(p, seed, w);
        float a0, a1, a2, a3;
        if constexpr (VAR == FULL) { aeth_rng_normal_pair(w[0], w[1], &a0, &a1); aeth_rng_normal_pair(w[2], w[3], &a2, &a3); }
        else if constexpr (VAR == PHILOX) { a0 = __uint_as_float(w[0]); a1 = __uint_as_float(w[1]); a2 = __uint_as_float(w[2]); a3 = __uint_as_float(w[3]); }
        else if constexpr (VAR == R2) { a0 = aeth_rng_r2((w[0] >> 8) | 1u); a1 = __uint_as_float(w[1]); a2 = aeth_rng_r2((w[2] >> 8) | 1u); a3 = __uint_as_float(w[3]); }
        else if constexpr (VAR == SQRT) { a0 = aeth_rng_sqrt(aeth_rng_r2((w[0] >> 8) | 1u)); a1 = __uint_as_float(w[1]); a2 = aeth_rng_sqrt(aeth_rng_r2((w[2] >> 8) | 1u)); a3 = __uint_as_float(w[3]); }
        else { aeth_rng_cossin(w[1], &a0, &a1); aeth_rng_cossin(w[3], &a2, &a3); }
        o[k] = f4{a0 * scale, a1 * scale, a2 * scale, a3 * scale};
        if constexpr (!DEFER) { if (p < npairs) { if constexpr (NT) __builtin_nontemporal_store(o[k], reinterpret_cast<f4 *>(x + p)); else *reinterpret_cast<f4 *>(x + p) = o[k]; } }
    }
    if constexpr (DEFER) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const size_t p = p0 + k * stride;
            if (p < npairs) { if constexpr (NT) __builtin_nontemporal_store(o[k], reinterpret_cast<f4 *>(x + p)); else *reinterpret_cast<f4 *>(x + p) = o[k]; }
        }
    }
}
template <int VAR, int K = 1, int MAP = 0, bool NT = true, bool DEFER = false, int BLOCK = 256> static double t_us(float4 *d, size_t npairs)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned grid = (unsigned)((npairs + BLOCK * K - 1) / (BLOCK * K));
    for (int i = 0; i < 5; i++) fill<VAR, K, MAP, NT, DEFER, BLOCK><<<grid, BLOCK>>>(d, npairs, 0.7f, 815);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; i++) fill<VAR, K, MAP, NT, DEFER, BLOCK><<<grid, BLOCK>>>(d, npairs, 0.7f, 815);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 20 * 1e3;
}
__global__ void all_radii(float *out)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    out[k] = aeth_rng_sqrt(aeth_rng_r2(2 * k + 1));
}

int main()
{
    const size_t n = (size_t)1 << 25, np = n / 2;
    float4 *d; CK(hipMalloc((void **)&d, n * 8));
    printf("fill of 2^25 samples, us per launch (268 MB of stores = 33.6 us at 8 TB/s); one pair per lane unless said\n");
    printf("  philox only %5.1f   + r^2 %5.1f   + sqrt %5.1f   philox + cossin %5.1f   whole generator %5.1f\n",
           t_us<PHILOX>(d, np), t_us<R2>(d, np), t_us<SQRT>(d, np), t_us<COSSIN>(d, np), t_us<FULL>(d, np));
    printf("launch shapes of the whole generator (K pairs per lane):\n");
    printf("  K=1  nt %5.1f  plain %5.1f   block 512: %5.1f  block 128: %5.1f  block 64: %5.1f\n", t_us<FULL, 1, 0, true>(d, np), t_us<FULL, 1, 0, false>(d, np),
           t_us<FULL, 1, 0, true, false, 512>(d, np), t_us<FULL, 1, 0, true, false, 128>(d, np), t_us<FULL, 1, 0, true, false, 64>(d, np));
    printf("  K=2  a grid apart: nt %5.1f  plain %5.1f  nt, stores last %5.1f | one run per block: nt %5.1f  plain %5.1f  nt, stores last %5.1f\n",
           t_us<FULL, 2, 0, true>(d, np), t_us<FULL, 2, 0, false>(d, np), t_us<FULL, 2, 0, true, true>(d, np),
           t_us<FULL, 2, 1, true>(d, np), t_us<FULL, 2, 1, false>(d, np), t_us<FULL, 2, 1, true, true>(d, np));
    printf("  K=4  a grid apart: nt %5.1f  plain %5.1f  nt, stores last %5.1f | one run per block: nt %5.1f  plain %5.1f  nt, stores last %5.1f\n",
           t_us<FULL, 4, 0, true>(d, np), t_us<FULL, 4, 0, false>(d, np), t_us<FULL, 4, 0, true, true>(d, np),
           t_us<FULL, 4, 1, true>(d, np), t_us<FULL, 4, 1, false>(d, np), t_us<FULL, 4, 1, true, true>(d, np));
    // device against host
    std::vector<float> b(1 << 22);
    fill<FULL, 1, 0, true, false, 256><<<(1 << 20) / 256, 256>>>(d, 1 << 20, 0.7f, 815); CK(hipMemcpy(b.data(), d, b.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; double s1 = 0, s2 = 0, s4 = 0, mx = 0;
    for (size_t p = 0; p < (1 << 20); p++) {
        uint32_t w[4]; float o[4];
        aeth_rng_draw(p, 815, w);
        aeth_rng_normal_pair(w[0], w[1], &o[0], &o[1]); aeth_rng_normal_pair(w[2], w[3], &o[2], &o[3]);
        for (int k = 0; k < 4; k++) { float v = o[k] * 0.7f; bad += memcmp(&v, &b[4 * p + k], 4) != 0; double t = o[k]; s1 += t; s2 += t * t; s4 += t * t * t * t; if (fabs(t) > mx) mx = fabs(t); }
    }
    const double N = 4.0 * (1 << 20);
    printf("device == host on 2^21 samples: %zu differ;  mean %.5f var %.5f kurtosis %.4f max |z| %.3f\n", bad, s1 / N, s2 / N, s4 / N / (s2 / N) / (s2 / N), mx);
    std::vector<float> hs(1 << 23);
    float *ds; CK(hipMalloc((void **)&ds, hs.size() * 4));
    all_radii<<<(1 << 23) / 256, 256>>>(ds);
    CK(hipMemcpy(hs.data(), ds, hs.size() * 4, hipMemcpyDeviceToHost));
    size_t bads = 0;
    for (uint32_t k = 0; k < (1u << 23); k++) { float r = aeth_rng_sqrt(aeth_rng_r2(2 * k + 1)); bads += memcmp(&r, &hs[k], 4) != 0; }
    printf("r = sqrt(-2 ln u) for all 2^23 u: device (rsq + one step) vs host (sqrtf): %zu differ\n", bads);
    return bad || bads;
}
