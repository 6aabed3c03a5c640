// aeth_sampling.hip -- linear interpolation and integer decimation
// (reference: src/sampling.rs:7-62).  Compiled with -ffp-contract=off so the
// interpolated values are bit-identical to the Rust path.
//
// interpolate is store-bound (8 B read per 8*(n_between+1) B written): one lane per
// OUTPUT sample so that every store instruction covers 512 contiguous bytes; the
// two source samples of a lane come from L1/L2 (neighbouring lanes share them).
// downsample is a strided gather: one lane per output, the read side touches one
// DRAM sector per sample once dec*elem_size >= 32 B.
#include "aeth_internal.h"

namespace {

using aeth::FastDiv;
using aeth::make_fastdiv;
using aeth::fdiv;

constexpr int kBlock = 256;

// Linear interpolation, frames of S inputs -> frames of Lo = S + (S-1)*nb outputs.
//
// A workgroup owns 512 consecutive outputs (two per lane, one 16-byte store when the
// destination allows).  Phase 1: the few input windows those outputs fall into (about
// 512/(nb+1)) get their per-window constants -- x1 and the two rates, i.e. the two f32
// divisions of sampling.rs:12-13 -- computed ONCE into LDS.  Phase 2: every output is
// one LDS read, a multiply and an add per component, in the reference's operation
// order (sampling.rs:18-19), so results stay bit-identical while the divisions per
// output drop from 2 to ~2/(nb+1).  Slots run on across frame boundaries; the last
// sample of a frame is a pseudo-window with zero rates (sampling.rs:23).
constexpr int kChunks = 4;                      // 16-byte stores per lane
constexpr int kOutPerWG = 2 * kBlock * kChunks; // outputs per workgroup: amortises the load -> LDS -> store latency chain

// 32-bit index version (total < 2^31): the common case.  NT: outputs stored with the non-temporal hint
template <bool NT>
__global__ __launch_bounds__(kBlock) void interpolate_kernel32(const float2 *__restrict__ src, float2 *__restrict__ dst,
                                                               uint32_t S, FastDiv Lo, uint32_t total, FastDiv nb1,
                                                               float div, int compat_im, int wide_store)
{
    extern __shared__ __attribute__((aligned(16))) float4 slot[];   // (x1.re, imaginary base, rate.0, rate.1), sized by the host
    const uint32_t o0 = blockIdx.x * kOutPerWG;             // first output of this workgroup
    const uint32_t f0 = fdiv(o0, Lo);
    const uint32_t w0 = fdiv(o0 - f0 * Lo.d, nb1);          // its window (sampling.rs:8)
    uint32_t olast = o0 + kOutPerWG - 1;
    if (olast >= total) olast = total - 1;
    const uint32_t fl = fdiv(olast, Lo);
    const uint32_t wl = fdiv(olast - fl * Lo.d, nb1);
    const uint32_t nslots = (fl - f0) * S + wl - w0 + 1;
    for (uint32_t q = threadIdx.x; q < nslots; q += kBlock) {
        uint32_t w = w0 + q, f = f0;
        while (w >= S) { w -= S; f++; }                     // slots continue into the next frame
        const float2 *s = src + (size_t)f * S;
        const float2 x1 = s[w];
        float4 e;
        if (w == S - 1) e = make_float4(x1.x, x1.y, 0.0f, 0.0f);      // dst.push(*src.last())  (sampling.rs:23)
        else {
            const float2 x2 = s[w + 1];
            e.x = x1.x;
            e.y = compat_im ? x1.x : x1.y;                  // sampling.rs:19 uses x1.re (sic)
            e.z = (x2.x - x1.x) / div;                      // sampling.rs:12
            e.w = (x2.y - x1.y) / div;                      // sampling.rs:13
        }
        slot[q] = e;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kChunks; c++) {
        const uint32_t o = o0 + c * (2 * kBlock) + 2 * threadIdx.x;
        if (o >= total) return;
        uint32_t f = fdiv(o, Lo), j = o - f * Lo.d;
        uint32_t w = fdiv(j, nb1);
        uint32_t i = j - w * nb1.d;                         // 0..=n_between (sampling.rs:16)
        uint32_t q = (f - f0) * S + w - w0;
        float4 e = slot[q];
        float fi = (float)i;
        const float2 v0 = make_float2(e.x + fi * e.z, e.y + fi * e.w);    // sampling.rs:18-19
        if (o + 1 >= total) { dst[o] = v0; return; }
        if (j + 1 == Lo.d) { q = (f + 1 - f0) * S - w0; i = 0; }          // first output of the next frame
        else if (++i == nb1.d) { i = 0; q++; }
        e = slot[q];
        fi = (float)i;
        const float2 v1 = make_float2(e.x + fi * e.z, e.y + fi * e.w);
        if (wide_store) aeth::nt_store<NT>(reinterpret_cast<float4 *>(dst + o), make_float4(v0.x, v0.y, v1.x, v1.y));
        else { dst[o] = v0; dst[o + 1] = v1; }
    }
}

// 64-bit fallback (more than 2^31 outputs): one lane per output, plain division
__global__ __launch_bounds__(kBlock) void interpolate_kernel64(const float2 *__restrict__ src, float2 *__restrict__ dst,
                                                               size_t S, size_t Lo, size_t total, unsigned nb1,
                                                               float div, int compat_im)
{
    const size_t o = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (o >= total) return;
    const size_t f = o / Lo, j = o - f * Lo;
    const float2 *s = src + f * S;
    const size_t w = j / nb1;
    const unsigned i = (unsigned)(j - w * nb1);
    float2 out;
    if (w >= S - 1) out = s[S - 1];
    else {
        const float2 x1 = s[w], x2 = s[w + 1];
        const float r0 = (x2.x - x1.x) / div, r1 = (x2.y - x1.y) / div;
        const float fi = (float)i;
        out.x = x1.x + fi * r0;
        out.y = (compat_im ? x1.x : x1.y) + fi * r1;
    }
    dst[o] = out;
}

template <typename T, bool NT>
__global__ __launch_bounds__(kBlock) void downsample_kernel(const T *__restrict__ src, T *__restrict__ dst,
                                                            size_t n_dst, size_t dec)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n_dst) {                               // *c = src[i * dec]       (sampling.rs:39-41)
        if constexpr (sizeof(T) >= 4) aeth::nt_store<NT>(dst + i, src[i * dec]);
        else dst[i] = src[i * dec];
    }
}

// one item per lane, the grid covers everything (see aeth_vecops.hip on why no capped loop)
inline unsigned grid_for(const aeth_ctx *, size_t items)
{
    size_t blocks = (items + kBlock - 1) / kBlock;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

int interpolate_impl(aeth_ctx *ctx, const aeth_cf32 *src, size_t S, size_t batch, aeth_cf32 *dst,
                     size_t cap, size_t nb, int compat, size_t *n_written)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    if (n_written) *n_written = 0;
    AETH_REQUIRE(S > 0, AETH_E_LEN, "interpolate on an empty src (the reference panics: sampling.rs:23)");
    AETH_REQUIRE(nb < 0x7fffffffu, AETH_E_ARG, "n_between too large");
    if (batch == 0) return AETH_OK;
    AETH_REQUIRE(src && dst, AETH_E_ARG, "null pointer");
    aeth::DeviceGuard dev_guard(ctx->device);
    AETH_REQUIRE(aeth::aligned8(src) && aeth::aligned8(dst), AETH_E_ALIGN, "pointer not 8-byte aligned");
    const size_t Lo = S + (S - 1) * nb;
    AETH_REQUIRE(cap >= Lo * batch, AETH_E_LEN, "dst capacity %zu < %zu", cap, Lo * batch);
    const size_t total = Lo * batch;
    const int wide = aeth::aligned16(dst) ? 1 : 0;
    if (total < 0x7fffffffull && S < 0x7fffffffull) {
        const dim3 g((unsigned)((total + kOutPerWG - 1) / kOutPerWG)), b(kBlock);
        // slots a workgroup can need: one per window its outputs touch, plus the pseudo-window
        // and a partial window per frame boundary it crosses
        size_t slots = kOutPerWG / (nb + 1) + 2 * (kOutPerWG / Lo + 2) + 4;
        if (slots > (size_t)kOutPerWG + 4) slots = kOutPerWG + 4;
        auto k32 = aeth::streams_past_cache(total * sizeof(float2)) ? interpolate_kernel32<true> : interpolate_kernel32<false>;
        hipLaunchKernelGGL(k32, g, b, slots * sizeof(float4), aeth::ctx_stream(ctx), reinterpret_cast<const float2 *>(src),
                           reinterpret_cast<float2 *>(dst), (uint32_t)S, make_fastdiv((uint32_t)Lo), (uint32_t)total,
                           make_fastdiv((uint32_t)(nb + 1)), (float)(nb + 1), compat, wide);
    } else {
        hipLaunchKernelGGL(interpolate_kernel64, dim3(grid_for(ctx, total)), dim3(kBlock), 0, aeth::ctx_stream(ctx),
                           reinterpret_cast<const float2 *>(src), reinterpret_cast<float2 *>(dst), S, Lo, total,
                           (unsigned)(nb + 1), (float)(nb + 1), compat);
    }
    AETH_HIP(hipGetLastError());
    if (n_written) *n_written = Lo * batch;
    return AETH_OK;
}

}  // namespace

extern "C" {

int aeth_interpolate(aeth_ctx *ctx, const aeth_cf32 *src, size_t n_src, aeth_cf32 *dst, size_t cap,
                     size_t nb, int compat, size_t *n_written)
{
    return interpolate_impl(ctx, src, n_src, 1, dst, cap, nb, compat, n_written);
}

int aeth_interpolate_frames(aeth_ctx *ctx, const aeth_cf32 *src, size_t frame_len, size_t batch,
                            aeth_cf32 *dst, size_t cap, size_t nb, int compat, size_t *n_written)
{
    return interpolate_impl(ctx, src, frame_len, batch, dst, cap, nb, compat, n_written);
}

int aeth_host_interpolate(aeth_ctx *ctx, const aeth_cf32 *src, size_t n_src, aeth_cf32 *dst, size_t cap,
                          size_t nb, int compat, size_t *n_written)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    if (n_written) *n_written = 0;
    AETH_REQUIRE(n_src > 0, AETH_E_LEN, "interpolate on an empty src (the reference panics: sampling.rs:23)");
    AETH_REQUIRE(src && dst, AETH_E_ARG, "null pointer");
    const size_t Lo = n_src + (n_src - 1) * nb;
    AETH_REQUIRE(cap >= Lo, AETH_E_LEN, "dst capacity %zu < %zu", cap, Lo);
    aeth::DeviceGuard dev_guard(ctx->device);
    aeth::HostIO io;
    int rc = io.open(ctx, n_src * sizeof(aeth_cf32), Lo * sizeof(aeth_cf32)); if (rc) return rc;
    rc = io.put(0, src, n_src * sizeof(aeth_cf32)); if (rc) return rc;
    rc = interpolate_impl(ctx, (const aeth_cf32 *)io.buf[0], n_src, 1, (aeth_cf32 *)io.buf[1], Lo, nb, compat, n_written);
    if (rc) return rc;
    return io.get(dst, 1, Lo * sizeof(aeth_cf32));
}

/* Which build of the reference an entry point matches (src/sampling.rs:32-36 is a debug_assert_eq!):
 *   BUILD_DEBUG    `cargo build` / `cargo test`: the divisibility assert is compiled in and panics
 *   BUILD_RELEASE  `cargo build --release` / `cargo bench` (benches/benches.rs:113,130 runs 8096 -> 512): the assert is
 *                  compiled out, dec = src.len() / dst.len() floors and every dst[i] = src[i * dec] that stays inside
 *                  src is taken; what still panics there is the division by zero of an empty dst, an index past the
 *                  end of src (only an empty src can do that, since (n_dst - 1) * (n_src / n_dst) < n_src) and, for
 *                  downsample_sb alone, step_by(0) when src is shorter than dst (:58-61) */
enum { BUILD_DEBUG = 0, BUILD_RELEASE = 1 };

static int downsample_check(size_t n_src, size_t n_dst, int build, int step_by)
{
    AETH_REQUIRE(n_dst > 0, AETH_E_LEN, "downsample into an empty dst (division by zero in the reference)");
    if (build == BUILD_DEBUG) {
        AETH_REQUIRE(n_src % n_dst == 0, AETH_E_LEN, AETH_MSG_DECIM);
        /* 0 % n == 0 passes the reference's assert too; dec = 0 then reads src[0] of an empty slice and panics (:39-41) */
        AETH_REQUIRE(n_src >= n_dst, AETH_E_LEN, "downsample from an empty src (the reference panics: index out of bounds)");
    } else {
        AETH_REQUIRE(n_src > 0, AETH_E_LEN, "downsample from an empty src (the reference panics: index out of bounds)");
        AETH_REQUIRE(!step_by || n_src >= n_dst, AETH_E_LEN,
                     "downsample_sb with src shorter than dst (the reference panics: step_by(0), sampling.rs:58-61)");
    }
    return AETH_OK;
}

static int downsample_dev(aeth_ctx *ctx, const void *src, size_t n_src, void *dst, size_t n_dst, size_t elem, int build, int step_by)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    int rc = downsample_check(n_src, n_dst, build, step_by); if (rc) return rc;
    AETH_REQUIRE(src && dst, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(elem == 1 || elem == 2 || elem == 4 || elem == 8 || elem == 16, AETH_E_ARG,
                 "elem_size %zu not in {1,2,4,8,16}", elem);
    AETH_REQUIRE(((uintptr_t)src % elem) == 0 && ((uintptr_t)dst % elem) == 0, AETH_E_ALIGN,
                 "pointer not aligned to elem_size");
    const size_t dec = n_src / n_dst;               /* :38 -- floors; 0 when src is shorter than dst (every dst[i] = src[0]) */
    aeth::DeviceGuard dev_guard(ctx->device);
    const dim3 g(grid_for(ctx, n_dst)), b(kBlock);
    const bool nt = aeth::streams_past_cache(n_dst * elem * 2);
#define AETH_DS(TT)                                                                                                       \
    do {                                                                                                                  \
        if (nt) hipLaunchKernelGGL((downsample_kernel<TT, true>), g, b, 0, aeth::ctx_stream(ctx), (const TT *)src, (TT *)dst, n_dst, dec);  \
        else hipLaunchKernelGGL((downsample_kernel<TT, false>), g, b, 0, aeth::ctx_stream(ctx), (const TT *)src, (TT *)dst, n_dst, dec);    \
    } while (0)
    switch (elem) {
    case 1:  AETH_DS(uint8_t); break;
    case 2:  AETH_DS(uint16_t); break;
    case 4:  AETH_DS(uint32_t); break;
    case 8:  AETH_DS(uint2); break;
    default: AETH_DS(uint4); break;
    }
#undef AETH_DS
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

static int downsample_host(aeth_ctx *ctx, const void *src, size_t n_src, void *dst, size_t n_dst, size_t elem, int build, int step_by)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    int rc = downsample_check(n_src, n_dst, build, step_by); if (rc) return rc;
    AETH_REQUIRE(src && dst, AETH_E_ARG, "null pointer");
    aeth::DeviceGuard dev_guard(ctx->device);
    aeth::HostIO io;
    rc = io.open(ctx, n_src * elem, n_dst * elem); if (rc) return rc;
    rc = io.put(0, src, n_src * elem); if (rc) return rc;
    rc = downsample_dev(ctx, io.buf[0], n_src, io.buf[1], n_dst, elem, build, step_by);
    if (rc) return rc;
    return io.get(dst, 1, n_dst * elem);
}

int aeth_downsample(aeth_ctx *ctx, const void *src, size_t n_src, void *dst, size_t n_dst, size_t elem)
{
    return downsample_dev(ctx, src, n_src, dst, n_dst, elem, BUILD_DEBUG, 0);
}

int aeth_host_downsample(aeth_ctx *ctx, const void *src, size_t n_src, void *dst, size_t n_dst, size_t elem)
{
    return downsample_host(ctx, src, n_src, dst, n_dst, elem, BUILD_DEBUG, 0);
}

int aeth_downsample_release(aeth_ctx *ctx, const void *src, size_t n_src, void *dst, size_t n_dst, size_t elem, int step_by)
{
    return downsample_dev(ctx, src, n_src, dst, n_dst, elem, BUILD_RELEASE, step_by);
}

int aeth_host_downsample_release(aeth_ctx *ctx, const void *src, size_t n_src, void *dst, size_t n_dst, size_t elem, int step_by)
{
    return downsample_host(ctx, src, n_src, dst, n_dst, elem, BUILD_RELEASE, step_by);
}

}  // extern "C"
