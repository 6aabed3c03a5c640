// aeth_fft_ragged.hip -- the "stockham_mixed_ragged" path of trait Fft (reference src/fft.rs:48-77): every length
// 2^a 3^b 5^c from 3 to 20480 that is no power of two, 8192, 16384, lengths with a factor 7 up to 4096 and with a
// factor 11, 13, 17, 19 or 23 up to 2048 -- one measured decomposition each (aeth_fft_ragged.h has the kernel, aeth_fft_ragged_table.inc
// the table).  The table is compiled in four slices -- this file once per slice with -DAETH_RAGGED_PART=k -- so
// that the build spreads over the cores.
#include "aeth_fft_ragged.h"
#include "aeth_fft_plan.h"

#ifndef AETH_RAGGED_PART
#error "compile with -DAETH_RAGGED_PART=0..3"
#endif

using namespace aeth::fftk;

namespace {

template <class C, int LANES_PER_CU>
int launch_ragged(const aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale)
{
    const aeth_ctx *ctx = plan->ctx;
    const bool nt = aeth::streams_past_cache(2 * batch * (size_t)C::N * sizeof(float2));
    const size_t ngroups = (batch + C::F - 1) / C::F;
    const size_t cap = (size_t)ctx->num_cus * (LANES_PER_CU / C::WG);      // persistent grid
    int grid = (int)(ngroups < cap ? ngroups : cap);
    if (grid < 1) grid = 1;
#define AETH_FFT_RAGGED(SS, NN) hipLaunchKernelGGL((fft_ragged_kernel<C, SS, NN>), dim3(grid), dim3(C::WG), 0, aeth::ctx_stream(ctx), (const cf *)in, (cf *)out, (const cf *)plan->tw_lane_dev, batch, scale)
    if (sign > 0) { if (nt) AETH_FFT_RAGGED(+1, true); else AETH_FFT_RAGGED(+1, false); }
    else          { if (nt) AETH_FFT_RAGGED(-1, true); else AETH_FFT_RAGGED(-1, false); }
#undef AETH_FFT_RAGGED
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

template <class C>
int build_ragged_table(aeth_fft *plan)
{
    AETH_HIP(hipMalloc((void **)&plan->tw_lane_dev, (size_t)C::TW * C::T * sizeof(float2)));
    hipLaunchKernelGGL((build_ragged_twiddles<C>), dim3(1), dim3((C::T + 63) / 64 * 64), 0, aeth::ctx_stream(plan->ctx),
                       (const cf *)plan->tw_dev, (cf *)plan->tw_lane_dev);
    AETH_HIP(hipGetLastError());
    AETH_HIP(hipStreamSynchronize(aeth::ctx_stream(plan->ctx)));
    return AETH_OK;
}

}  // namespace

// table rows of this slice become switch cases, the other slices' rows vanish
#define AETH_RAGGED_ROW(N, T, WG, LANES, PAD, ...)                                                           \
    case N: {                                                                                           \
        using C = RCfg<N, T, WG, PAD, __VA_ARGS__>;                                                          \
        return in ? launch_ragged<C, LANES>(plan, in, out, batch, sign, scale) : build_ragged_table<C>(plan); \
    }
#define AETH_RAGGED_SKIP(...)
#if AETH_RAGGED_PART == 0
#define AETH_RAGGED_P0 AETH_RAGGED_ROW
#define AETH_RAGGED_SLICE fft_ragged_slice0
#else
#define AETH_RAGGED_P0 AETH_RAGGED_SKIP
#endif
#if AETH_RAGGED_PART == 1
#define AETH_RAGGED_P1 AETH_RAGGED_ROW
#define AETH_RAGGED_SLICE fft_ragged_slice1
#else
#define AETH_RAGGED_P1 AETH_RAGGED_SKIP
#endif
#if AETH_RAGGED_PART == 2
#define AETH_RAGGED_P2 AETH_RAGGED_ROW
#define AETH_RAGGED_SLICE fft_ragged_slice2
#else
#define AETH_RAGGED_P2 AETH_RAGGED_SKIP
#endif
#if AETH_RAGGED_PART == 3
#define AETH_RAGGED_P3 AETH_RAGGED_ROW
#define AETH_RAGGED_SLICE fft_ragged_slice3
#else
#define AETH_RAGGED_P3 AETH_RAGGED_SKIP
#endif

namespace aeth {

// in == nullptr: build the plan's per-lane twiddle table; otherwise transform.  kRaggedNotHere: not in this slice.
int AETH_RAGGED_SLICE(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale)
{
    switch (plan->len) {
#include "aeth_fft_ragged_table.inc"
    default: return kRaggedNotHere;
    }
}

#if AETH_RAGGED_PART == 0
static int ragged_any(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale)
{
    int rc = fft_ragged_slice0(plan, in, out, batch, sign, scale);
    if (rc == kRaggedNotHere) rc = fft_ragged_slice1(plan, in, out, batch, sign, scale);
    if (rc == kRaggedNotHere) rc = fft_ragged_slice2(plan, in, out, batch, sign, scale);
    if (rc == kRaggedNotHere) rc = fft_ragged_slice3(plan, in, out, batch, sign, scale);
    if (rc == kRaggedNotHere) rc = aeth::set_error(AETH_E_UNSUPPORTED, "stockham_mixed_ragged: length %zu", plan->len);
    return rc;
}

int fft_plan_ragged(aeth_fft *plan) { return ragged_any(plan, nullptr, nullptr, 0, +1, 1.0f); }

int fft_run_ragged(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale)
{
    AETH_REQUIRE(in && out, AETH_E_ARG, "null argument");
    return ragged_any(plan, in, out, batch, sign, scale);
}

bool fft_ragged_supported(size_t len)
{
#undef AETH_RAGGED_P0
#undef AETH_RAGGED_P1
#undef AETH_RAGGED_P2
#undef AETH_RAGGED_P3
#define AETH_RAGGED_P0(N, ...) case N:
#define AETH_RAGGED_P1(N, ...) case N:
#define AETH_RAGGED_P2(N, ...) case N:
#define AETH_RAGGED_P3(N, ...) case N:
    switch (len) {
#include "aeth_fft_ragged_table.inc"
        return true;
    default: return false;
    }
}
#endif

}  // namespace aeth
