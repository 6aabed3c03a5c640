// aeth_noise.hip -- additive white Gaussian noise on a device-resident signal
// (reference: src/noise.rs:29-59, Awgn::new / next / apply).  SURVEY 8f "next" row #1.
//
// The arithmetic around the random stream follows the reference exactly:
//   scale = power.sqrt()                          (noise.rs:35)
//   next  = (N(0,1) as f32 * scale, N(0,1) as f32 * scale)      (noise.rs:39-43)
//   apply : s += next.scale(scale)                (noise.rs:53-59: scaled TWICE, so the noise
//                                                  amplitude is proportional to `power`)
// The stream itself is the build's own counter-based generator (aeth_rng.h): the reference's
// StdRng/ziggurat sequence cannot be reproduced.  Compiled with -ffp-contract=off.
// One lane per PAIR of samples (one Philox call, one 16-byte access), kPairs pairs per lane a grid's width apart: the
// generator is issue-bound (about 130 vector instructions per pair), and a wave that ends on its only store holds its
// slot for the store's latency with nothing to issue -- with a second pair behind it the slot keeps computing
// (tools/rng_lab.hip: 55 -> 50 us per 2^25 samples; more than two measured no better).
#include "aeth_internal.h"

#define AETH_RNG_FN __host__ __device__ static inline
#include "aeth_rng.h"

namespace {

constexpr int kBlock = 256;
constexpr int kPairs = 2;
inline unsigned pair_grid(size_t pairs) { return (unsigned)((pairs + (size_t)kBlock * kPairs - 1) / ((size_t)kBlock * kPairs)); }

template <bool NT>
__global__ __launch_bounds__(kBlock) void awgn_apply_kernel(float2 *__restrict__ x, size_t n, float scale,
                                                            uint64_t seed, uint64_t offset, int wide)
{
    // the samples of ALL of the lane's pairs are requested first and used last: the noise does not depend on them, so
    // their latency rides under the generator's arithmetic instead of being waited for behind it
    float4 xv[kPairs];
#pragma unroll
    for (int k = 0; k < kPairs; k++) {
        const size_t i0 = 2 * ((size_t)blockIdx.x * kBlock + threadIdx.x + (size_t)k * gridDim.x * kBlock);
        xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i0 >= n) break;
        if (wide && i0 + 1 < n) xv[k] = aeth::nt_load<NT>(reinterpret_cast<float4 *>(x + i0));
        else {
            const float2 a = x[i0];
            xv[k].x = a.x; xv[k].y = a.y;
            if (i0 + 1 < n) { const float2 b = x[i0 + 1]; xv[k].z = b.x; xv[k].w = b.y; }
        }
    }
#pragma unroll
    for (int k = 0; k < kPairs; k++) {
        const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x + (size_t)k * gridDim.x * kBlock;       // pair index
        const size_t i0 = 2 * p;
        if (i0 >= n) return;
        // stream position of sample i is offset + i; an odd offset shifts the pairing, so draw per sample then
        float n0r, n0i, n1r = 0.f, n1i = 0.f;
        if ((offset & 1) == 0) {
            uint32_t w[4];
            const uint64_t call = (offset + i0) >> 1;
            aeth_rng_draw(call, seed, w);
            aeth_rng_normal_pair(w[0], w[1], &n0r, &n0i);
            aeth_rng_normal_pair(w[2], w[3], &n1r, &n1i);
        } else {
            aeth_rng_cnormal(seed, offset + i0, &n0r, &n0i);
            if (i0 + 1 < n) aeth_rng_cnormal(seed, offset + i0 + 1, &n1r, &n1i);
        }
        float4 v = xv[k];
        v.x = v.x + (n0r * scale) * scale;          // noise.rs:41 then :58
        v.y = v.y + (n0i * scale) * scale;
        v.z = v.z + (n1r * scale) * scale;
        v.w = v.w + (n1i * scale) * scale;
        if (wide && i0 + 1 < n) aeth::nt_store<NT>(reinterpret_cast<float4 *>(x + i0), v);
        else {
            x[i0] = make_float2(v.x, v.y);
            if (i0 + 1 < n) x[i0 + 1] = make_float2(v.z, v.w);
        }
    }
}

// Awgn::fill / Awgn::iter (noise.rs:61-84): target[i] = next() = (N(0,1) as f32 * scale, ...), scaled ONCE
template <bool NT>
__global__ __launch_bounds__(kBlock) void awgn_fill_kernel(float2 *__restrict__ x, size_t n, float scale,
                                                           uint64_t seed, uint64_t offset, int wide)
{
#pragma unroll
    for (int k = 0; k < kPairs; k++) {
        const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x + (size_t)k * gridDim.x * kBlock;
        const size_t i0 = 2 * p;
        if (i0 >= n) return;
        float n0r, n0i, n1r = 0.f, n1i = 0.f;
        if ((offset & 1) == 0) {
            uint32_t w[4];
            const uint64_t call = (offset + i0) >> 1;
            aeth_rng_draw(call, seed, w);
            aeth_rng_normal_pair(w[0], w[1], &n0r, &n0i);
            aeth_rng_normal_pair(w[2], w[3], &n1r, &n1i);
        } else {
            aeth_rng_cnormal(seed, offset + i0, &n0r, &n0i);
            if (i0 + 1 < n) aeth_rng_cnormal(seed, offset + i0 + 1, &n1r, &n1i);
        }
        if (wide && i0 + 1 < n) {
            aeth::nt_store<NT>(reinterpret_cast<float4 *>(x + i0), make_float4(n0r * scale, n0i * scale, n1r * scale, n1i * scale));
        } else {
            x[i0] = make_float2(n0r * scale, n0i * scale);
            if (i0 + 1 < n) x[i0 + 1] = make_float2(n1r * scale, n1i * scale);
        }
    }
}

// the generator's integer stage on its own: out[i] = Philox4x32-R(counter = in[i][0..3], key = in[i][4..5])
template <int R>
__global__ __launch_bounds__(kBlock) void philox_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t w[4];
    aeth_philox4x32(R, in[6 * i], in[6 * i + 1], in[6 * i + 2], in[6 * i + 3], in[6 * i + 4], in[6 * i + 5], w);
    out[4 * i] = w[0]; out[4 * i + 1] = w[1]; out[4 * i + 2] = w[2]; out[4 * i + 3] = w[3];
}

// the generator's floating-point stage on its own: out[i] = the complex normal of the word pair (ab[i][0], ab[i][1])
__global__ __launch_bounds__(kBlock) void normal_pairs_kernel(const uint2 *__restrict__ ab, float2 *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint2 w = ab[i];
    float2 z;
    aeth_rng_normal_pair(w.x, w.y, &z.x, &z.y);
    out[i] = z;
}

}  // namespace

extern "C" {

int aeth_awgn_fill(aeth_ctx *ctx, aeth_cf32 *target, size_t n, float power, uint64_t seed, uint64_t offset)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    if (n == 0) return AETH_OK;
    AETH_REQUIRE(target, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(aeth::aligned8(target), AETH_E_ALIGN, "target not 8-byte aligned");
    AETH_REQUIRE(power >= 0.0f, AETH_E_ARG, "noise power must be >= 0");
    const float scale = sqrtf(power);                       // noise.rs:35
    aeth::DeviceGuard dev_guard(ctx->device);
    const size_t pairs = (n + 1) / 2;
    auto kern = aeth::streams_past_cache(n * sizeof(float2)) ? awgn_fill_kernel<true> : awgn_fill_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(pair_grid(pairs)), dim3(kBlock), 0, aeth::ctx_stream(ctx),
                       reinterpret_cast<float2 *>(target), n, scale, seed, offset, aeth::aligned16(target) ? 1 : 0);
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

int aeth_rng_philox4x32(aeth_ctx *ctx, const uint32_t *ctr_key_dev, size_t n, int rounds, uint32_t *out_dev)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    AETH_REQUIRE(rounds == 7 || rounds == 10, AETH_E_UNSUPPORTED, "Philox4x32-%d: 7 (the generator's) or 10 rounds", rounds);
    if (n == 0) return AETH_OK;
    AETH_REQUIRE(ctr_key_dev && out_dev, AETH_E_ARG, "null pointer");
    aeth::DeviceGuard dev_guard(ctx->device);
    auto kern = rounds == 7 ? philox_kernel<7> : philox_kernel<10>;
    hipLaunchKernelGGL(kern, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, aeth::ctx_stream(ctx), ctr_key_dev, out_dev, n);
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

int aeth_rng_normal_pairs(aeth_ctx *ctx, const uint32_t *ab_dev, size_t n, aeth_cf32 *out_dev)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    if (n == 0) return AETH_OK;
    AETH_REQUIRE(ab_dev && out_dev, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(aeth::aligned8(ab_dev) && aeth::aligned8(out_dev), AETH_E_ALIGN, "pointers not 8-byte aligned");
    aeth::DeviceGuard dev_guard(ctx->device);
    hipLaunchKernelGGL(normal_pairs_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, aeth::ctx_stream(ctx),
                       reinterpret_cast<const uint2 *>(ab_dev), reinterpret_cast<float2 *>(out_dev), n);
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

int aeth_rng_philox4x32_10(aeth_ctx *ctx, const uint32_t *ctr_key_dev, size_t n, uint32_t *out_dev)
{
    return aeth_rng_philox4x32(ctx, ctr_key_dev, n, 10, out_dev);
}

int aeth_awgn_apply(aeth_ctx *ctx, aeth_cf32 *signal, size_t n, float power, uint64_t seed, uint64_t offset)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    if (n == 0) return AETH_OK;
    AETH_REQUIRE(signal, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(aeth::aligned8(signal), AETH_E_ALIGN, "signal not 8-byte aligned");
    AETH_REQUIRE(power >= 0.0f, AETH_E_ARG, "noise power must be >= 0");
    const float scale = sqrtf(power);                       // noise.rs:35
    aeth::DeviceGuard dev_guard(ctx->device);
    const size_t pairs = (n + 1) / 2;
    auto kern = aeth::streams_past_cache(2 * n * sizeof(float2)) ? awgn_apply_kernel<true> : awgn_apply_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(pair_grid(pairs)), dim3(kBlock), 0, aeth::ctx_stream(ctx),
                       reinterpret_cast<float2 *>(signal), n, scale, seed, offset, aeth::aligned16(signal) ? 1 : 0);
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

}  // extern "C"
