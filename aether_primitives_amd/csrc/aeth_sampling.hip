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

constexpr int kBlock = 256;

// one interpolated output: window w, step i of frame f (sampling.rs:8-23)
__device__ __forceinline__ float2 interp_at(const float2 *__restrict__ s, size_t S, size_t w, unsigned i, float div,
                                            int compat_im)
{
    if (w >= S - 1) return s[S - 1];              // dst.push(*src.last())   (sampling.rs:23)
    float2 x1 = s[w], x2 = s[w + 1];
    float r0 = (x2.x - x1.x) / div;               // (sampling.rs:12)
    float r1 = (x2.y - x1.y) / div;               // (sampling.rs:13)
    float fi = (float)i;                          // (sampling.rs:16)
    float2 out;
    out.x = x1.x + fi * r0;                       // (sampling.rs:18)
    out.y = (compat_im ? x1.x : x1.y) + fi * r1;  // (sampling.rs:19, sic)
    return out;
}

// frames of S inputs -> frames of Lo = S + (S-1)*nb outputs.  Two consecutive outputs per
// lane (one 16-byte store when the destination allows): the index of the second follows
// from the first without another division.  IDX = uint32_t when everything fits 32 bits.
template <typename IDX>
__global__ __launch_bounds__(kBlock) void interpolate_kernel(const float2 *__restrict__ src, float2 *__restrict__ dst,
                                                             IDX S, IDX Lo, IDX total, unsigned nb1, float div,
                                                             int compat_im, int single_frame, int wide_store)
{
    const IDX o = ((IDX)blockIdx.x * kBlock + threadIdx.x) * 2;
    if (o >= total) return;
    IDX f = 0, j = o;
    if (!single_frame) { f = o / Lo; j = o - f * Lo; }      // position inside the output frame
    IDX w = j / nb1;                                        // window index  (sampling.rs:8)
    unsigned i = (unsigned)(j - w * nb1);                   // 0..=n_between
    const float2 v0 = interp_at(src + (size_t)f * S, S, w, i, div, compat_im);
    if (o + 1 >= total) { dst[o] = v0; return; }
    // next output: same window one step on, or the next window, or the next frame
    if (++j == Lo) { j = 0; f++; w = 0; i = 0; }
    else if (++i == nb1) { i = 0; w++; }
    const float2 v1 = interp_at(src + (size_t)f * S, S, w, i, div, compat_im);
    if (wide_store) *reinterpret_cast<float4 *>(dst + o) = make_float4(v0.x, v0.y, v1.x, v1.y);
    else { dst[o] = v0; dst[o + 1] = v1; }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void downsample_kernel(const T *__restrict__ src, T *__restrict__ dst,
                                                            size_t n_dst, size_t dec)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n_dst) dst[i] = src[i * dec];         // *c = src[i * dec]       (sampling.rs:39-41)
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
    AETH_REQUIRE(aeth::aligned8(src) && aeth::aligned8(dst), AETH_E_ALIGN, "pointer not 8-byte aligned");
    const size_t Lo = S + (S - 1) * nb;
    AETH_REQUIRE(cap >= Lo * batch, AETH_E_LEN, "dst capacity %zu < %zu", cap, Lo * batch);
    const size_t total = Lo * batch;
    const int wide = aeth::aligned16(dst) ? 1 : 0;
    const dim3 g(grid_for(ctx, (total + 1) / 2)), b(kBlock);
    if (total < 0xffffffffull && S < 0xffffffffull)
        hipLaunchKernelGGL(interpolate_kernel<uint32_t>, g, b, 0, ctx->stream, reinterpret_cast<const float2 *>(src),
                           reinterpret_cast<float2 *>(dst), (uint32_t)S, (uint32_t)Lo, (uint32_t)total,
                           (unsigned)(nb + 1), (float)(nb + 1), compat, batch == 1 ? 1 : 0, wide);
    else
        hipLaunchKernelGGL(interpolate_kernel<uint64_t>, g, b, 0, ctx->stream, reinterpret_cast<const float2 *>(src),
                           reinterpret_cast<float2 *>(dst), (uint64_t)S, (uint64_t)Lo, (uint64_t)total,
                           (unsigned)(nb + 1), (float)(nb + 1), compat, batch == 1 ? 1 : 0, wide);
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
    int rc = aeth::ctx_stage(ctx, 0, n_src * sizeof(aeth_cf32)); if (rc) return rc;
    rc = aeth::ctx_stage(ctx, 1, Lo * sizeof(aeth_cf32)); if (rc) return rc;
    AETH_HIP(hipMemcpyAsync(ctx->stage[0], src, n_src * sizeof(aeth_cf32), hipMemcpyHostToDevice, ctx->stream));
    rc = interpolate_impl(ctx, (const aeth_cf32 *)ctx->stage[0], n_src, 1, (aeth_cf32 *)ctx->stage[1], Lo, nb, compat, n_written);
    if (rc) return rc;
    AETH_HIP(hipMemcpyAsync(dst, ctx->stage[1], Lo * sizeof(aeth_cf32), hipMemcpyDeviceToHost, ctx->stream));
    AETH_HIP(hipStreamSynchronize(ctx->stream));
    return AETH_OK;
}

int aeth_downsample(aeth_ctx *ctx, const void *src, size_t n_src, void *dst, size_t n_dst, size_t elem)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    AETH_REQUIRE(n_dst > 0, AETH_E_LEN, "downsample into an empty dst (division by zero in the reference)");
    AETH_REQUIRE(n_src % n_dst == 0, AETH_E_LEN, AETH_MSG_DECIM);
    AETH_REQUIRE(src && dst, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(elem == 1 || elem == 2 || elem == 4 || elem == 8 || elem == 16, AETH_E_ARG,
                 "elem_size %zu not in {1,2,4,8,16}", elem);
    AETH_REQUIRE(((uintptr_t)src % elem) == 0 && ((uintptr_t)dst % elem) == 0, AETH_E_ALIGN,
                 "pointer not aligned to elem_size");
    const size_t dec = n_src / n_dst;
    const dim3 g(grid_for(ctx, n_dst)), b(kBlock);
    switch (elem) {
    case 1:  hipLaunchKernelGGL(downsample_kernel<uint8_t>,  g, b, 0, ctx->stream, (const uint8_t *)src,  (uint8_t *)dst,  n_dst, dec); break;
    case 2:  hipLaunchKernelGGL(downsample_kernel<uint16_t>, g, b, 0, ctx->stream, (const uint16_t *)src, (uint16_t *)dst, n_dst, dec); break;
    case 4:  hipLaunchKernelGGL(downsample_kernel<uint32_t>, g, b, 0, ctx->stream, (const uint32_t *)src, (uint32_t *)dst, n_dst, dec); break;
    case 8:  hipLaunchKernelGGL(downsample_kernel<uint2>,    g, b, 0, ctx->stream, (const uint2 *)src,    (uint2 *)dst,    n_dst, dec); break;
    default: hipLaunchKernelGGL(downsample_kernel<uint4>,    g, b, 0, ctx->stream, (const uint4 *)src,    (uint4 *)dst,    n_dst, dec); break;
    }
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

int aeth_host_downsample(aeth_ctx *ctx, const void *src, size_t n_src, void *dst, size_t n_dst, size_t elem)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    AETH_REQUIRE(n_dst > 0, AETH_E_LEN, "downsample into an empty dst (division by zero in the reference)");
    AETH_REQUIRE(n_src % n_dst == 0, AETH_E_LEN, AETH_MSG_DECIM);
    AETH_REQUIRE(src && dst, AETH_E_ARG, "null pointer");
    int rc = aeth::ctx_stage(ctx, 0, n_src * elem); if (rc) return rc;
    rc = aeth::ctx_stage(ctx, 1, n_dst * elem); if (rc) return rc;
    AETH_HIP(hipMemcpyAsync(ctx->stage[0], src, n_src * elem, hipMemcpyHostToDevice, ctx->stream));
    rc = aeth_downsample(ctx, ctx->stage[0], n_src, ctx->stage[1], n_dst, elem);
    if (rc) return rc;
    AETH_HIP(hipMemcpyAsync(dst, ctx->stage[1], n_dst * elem, hipMemcpyDeviceToHost, ctx->stream));
    AETH_HIP(hipStreamSynchronize(ctx->stream));
    return AETH_OK;
}

}  // extern "C"
