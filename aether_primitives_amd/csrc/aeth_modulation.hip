// aeth_modulation.hip -- bit -> symbol map and hard demodulation
// (reference: src/modulation.rs:5-149).  SURVEY 8f "next" row #1: completes BASELINE
// config 4 (QPSK mod -> AWGN -> FFT-2048 correlate -> hard demod) on the device.
// Compiled with -ffp-contract=off: the distance compare must round as the Rust code does.
// HBM-bound: modulate reads 1-2 B and writes 8 B per symbol; demod reads 8 B, writes 1-2 B.
#include "aeth_internal.h"

namespace {

constexpr int kBlock = 256;

struct Table4 { float2 s[4]; };

// GENERIC_BPSK_TABLE / GENERIC_QPSK_TABLE, src/modulation.rs:77,87-92
const float2 kBpsk[2] = {{1.f, 1.f}, {-1.f, -1.f}};
const float2 kQpsk[4] = {{1.f, 1.f}, {-1.f, 1.f}, {1.f, -1.f}, {-1.f, -1.f}};

template <int BPS>
__global__ __launch_bounds__(kBlock) void modulate_kernel(const uint8_t *__restrict__ bits, float2 *__restrict__ out,
                                                          size_t nsym, Table4 t)
{
    const size_t s = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= nsym) return;
    unsigned idx;
    if constexpr (BPS == 1) idx = bits[s] & 1u;                                   // modulation.rs:9-12
    else {
        const uchar2 b = reinterpret_cast<const uchar2 *>(bits)[s];
        idx = ((b.y & 1u) << 1) + (b.x & 1u);                                     // modulation.rs:21-24
    }
    out[s] = t.s[idx];                                                            // modulation.rs:115-121
}

template <int BPS>
__global__ __launch_bounds__(kBlock) void demod_kernel(const float2 *__restrict__ sym, uint8_t *__restrict__ bits,
                                                       size_t nsym, Table4 t, int compat)
{
    const size_t s = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= nsym) return;
    const float2 v = sym[s];
    unsigned best = 0;
    float bd = 0.f;
#pragma unroll
    for (unsigned i = 0; i < (BPS == 1 ? 2u : 4u); i++) {
        const float dr = v.x - t.s[i].x, di = v.y - t.s[i].y;
        const float d = dr * dr + di * di;                                        // modulation.rs:36-41
        if (i == 0 || d < bd) { best = i; bd = d; }                               // first minimum wins (:46-49)
    }
    if constexpr (BPS == 1) bits[s] = (uint8_t)(best & 1u);                       // modulation.rs:143
    else {
        uchar2 o;
        o.x = (uint8_t)(best & 1u);                                               // modulation.rs:53
        o.y = (uint8_t)(compat ? (best & 2u) : ((best >> 1) & 1u));               // modulation.rs:54 (`idx & 1u8 << 1`)
        reinterpret_cast<uchar2 *>(bits)[s] = o;
    }
}

int fill_table(Table4 &t, int bps, const aeth_cf32 *host)
{
    AETH_REQUIRE(bps == 1 || bps == 2, AETH_E_UNSUPPORTED, "bits_per_symbol %d: only BPSK (1) and QPSK (2) tables exist in the reference", bps);
    const int n = bps == 1 ? 2 : 4;
    for (int i = 0; i < 4; i++) t.s[i] = make_float2(0.f, 0.f);
    for (int i = 0; i < n; i++)
        t.s[i] = host ? make_float2(host[i].re, host[i].im) : (bps == 1 ? kBpsk[i] : kQpsk[i]);
    return AETH_OK;
}

inline unsigned grid_for(size_t items) { size_t b = (items + kBlock - 1) / kBlock; return (unsigned)(b < 1 ? 1 : b); }

}  // namespace

extern "C" {

int aeth_modulate(aeth_ctx *ctx, const uint8_t *bits, size_t nbits, int bps, const aeth_cf32 *table, aeth_cf32 *out,
                  size_t n_out)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    Table4 t;
    int rc = fill_table(t, bps, table); if (rc) return rc;
    AETH_REQUIRE(nbits % (size_t)bps == 0, AETH_E_LEN, "bit count %zu is not a multiple of BITS_PER_SYMBOL %d", nbits, bps);
    AETH_REQUIRE(n_out == nbits / (size_t)bps, AETH_E_LEN, "output holds %zu symbols, input gives %zu", n_out, nbits / (size_t)bps);
    if (n_out == 0) return AETH_OK;
    aeth::DeviceGuard dev_guard(ctx->device);
    AETH_REQUIRE(bits && out, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(aeth::aligned8(out) && ((uintptr_t)bits % (size_t)bps) == 0, AETH_E_ALIGN, "pointer alignment");
    if (bps == 1) hipLaunchKernelGGL(modulate_kernel<1>, dim3(grid_for(n_out)), dim3(kBlock), 0, ctx->stream, bits, (float2 *)out, n_out, t);
    else          hipLaunchKernelGGL(modulate_kernel<2>, dim3(grid_for(n_out)), dim3(kBlock), 0, ctx->stream, bits, (float2 *)out, n_out, t);
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

int aeth_demod_naive(aeth_ctx *ctx, const aeth_cf32 *sym, size_t nsym, int bps, const aeth_cf32 *table, uint8_t *bits,
                     size_t nbits_out, int compat)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    Table4 t;
    int rc = fill_table(t, bps, table); if (rc) return rc;
    AETH_REQUIRE(nbits_out == nsym * (size_t)bps, AETH_E_LEN, "output holds %zu bits, input gives %zu", nbits_out, nsym * (size_t)bps);
    if (nsym == 0) return AETH_OK;
    aeth::DeviceGuard dev_guard(ctx->device);
    AETH_REQUIRE(sym && bits, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(aeth::aligned8(sym) && ((uintptr_t)bits % (size_t)bps) == 0, AETH_E_ALIGN, "pointer alignment");
    if (bps == 1) hipLaunchKernelGGL(demod_kernel<1>, dim3(grid_for(nsym)), dim3(kBlock), 0, ctx->stream, (const float2 *)sym, bits, nsym, t, compat);
    else          hipLaunchKernelGGL(demod_kernel<2>, dim3(grid_for(nsym)), dim3(kBlock), 0, ctx->stream, (const float2 *)sym, bits, nsym, t, compat);
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

}  // extern "C"
