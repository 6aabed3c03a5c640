// aeth_fir.hip -- fused  FFT -> (* H) -> IFFT  kernel.
//
// The reference has no FIR (src/fir.rs:3-22 holds taps and an empty scratch, no
// filter method; README.md:95-96 lists it as TODO).  What it does have is the
// frequency-domain multiply chain
//     input.vec_rfft(&mut fft, s).vec_mul(&sig).vec_rifft(&mut fft, s)
// (benches/benches.rs:410-416).  Run over overlapping blocks that chain IS
// overlap-save convolution, so the FIR here is defined as exactly that:
//     block  = x[b*hop - ov .. b*hop - ov + N)            ov = N - hop >= ntaps-1
//     Y      = bwd( fwd(block) * fwd(taps || 0), Scale::N )
//     y[b*hop .. (b+1)*hop) = Y[ov .. N)
// with fwd/bwd the reference's transforms (+j / -j exponent, src/fft.rs:148,150).
//
// One kernel does the whole chain per block: HBM is touched once for the input
// window (8*N/hop B per output) and once for the output (8 B); the forward
// transform's last pass leaves the spectrum in exactly the register slots the
// inverse transform's first pass reads, so the multiply by H happens in registers
// with no exchange.  H and the twiddles live in registers across a persistent
// loop over blocks.  For windows of 512 samples and more hop is rounded down to a multiple of
// 64 samples so that every block's window and output start on a 512-byte boundary.
#include "aeth_internal.h"
#include "aeth_fft_core.h"
#include "aeth_fft_plan.h"
#include "aeth_fir_kernel.h"
#include "aeth_host.h"

#include <cstdlib>
#include <new>
#include <vector>

using namespace aeth::fftk;


using namespace aeth::firk;

namespace {

template <class C, bool SCALED>
int launch_fmi(aeth_ctx *ctx, const FmiArgs &a, hipStream_t stream)
{
    long long ngroups = (a.nblocks + C::F - 1) / C::F;
    // persistent grid = what is resident at once: the kernel's ~230 VGPRs allow 2 waves per SIMD, 8 per CU
    // (the lab's 8-points-per-lane build of N = 2048 needs half the registers: 16 waves per CU)
    constexpr bool kLab8 = (C::N == 2048 && C::P == 8);
    [[maybe_unused]] const bool lab8_four = kLab8 && aeth::lab_int("AETH_DEMOD_P8", 0) == 4;    // 128 registers forced: 4 workgroups per CU
    const int kWavesPerCu = kLab8 ? (lab8_four ? 16 : 12) : 8;                   // 146 registers: 3 waves per SIMD
    long long cap = (long long)ctx->num_cus * kWavesPerCu / (C::WG / 64);
    // A launch that runs BESIDE its predecessor on the overlap lane takes three of the four 128-lane workgroups a CU
    // holds: a full grid keeps every wave slot until its last round, so its successor could only overlap that tail;
    // with a slot per CU free the two run side by side from the start.  tools/fir_lab, two queues, grids 704 ... 800
    // of 1024: 46.6 us per launch against 47.5; through the library, alternating in one process
    // (tools/overlap_grid_ab.py, 25 rounds): regions of 20 launches -1.8 %, of 200 launches -1.4 %.  A launch on its own
    // (one queue, or the first of a chain) keeps the full grid: 53.5 us against 54.4.
    // Tuning (sixteenths of the resident grid): AETH_FIR_GRID_CHAINED for a launch beside its predecessor,
    // AETH_FIR_GRID_FIRST for the first launch of a chain / a lone launch on a context with the lane on.
    if (ctx->overlap && !ctx->stream_shared && C::WG == 128) {
        const int num = ctx->last_chained ? aeth::tuning_int("AETH_FIR_GRID_CHAINED", 12) : aeth::tuning_int("AETH_FIR_GRID_FIRST", 16);
        if (num > 0 && num != 16) cap = cap * num / 16;      // above 16: more workgroups than fit at once (they queue for slots)
    }
    int grid = (int)(ngroups < cap ? ngroups : cap);
    if (grid < 1) grid = 1;
    FmiArgs b = a;
    if (b.frame_n == 0) b.frame_n = C::N;
    const bool nt = aeth::streams_past_cache(2 * (size_t)a.n * sizeof(float2));
    if constexpr (SCALED) {
        if (b.chirp) {                                      // chirp-z frames always carry a scale factor
            if (nt) hipLaunchKernelGGL((fmi_kernel<C, true, 1, true, true>), dim3(grid), dim3(C::WG), 0, stream, b);
            else hipLaunchKernelGGL((fmi_kernel<C, true, 1, false, true>), dim3(grid), dim3(C::WG), 0, stream, b);
            AETH_HIP(hipGetLastError());
            return AETH_OK;
        }
    }
    // one-frame workgroups run their LDS exchanges at raised wave priority (V_PRIO: -0.3 ... -0.5 us per 16 Mi-sample
    // launch in tools/fir_lab, A/B in one process); the other variants of aeth_fir_kernel.h measured null or negative
    // ... and N = 2048 (the configuration it was measured on) keeps its exchange image XOR-swizzled instead of padded:
    // no two-way conflict on the contiguous reads, transform-only time 39.8 -> 34.8 us, launch 54.8 -> 53.4 us
    constexpr int VAR = (C::F == 1) ? (V_PRIO | ((C::N == 2048 && C::P == 16) ? V_XOR : 0)) : 0;
    if constexpr (C::F == 1) {
        if (b.bits) {                                       // hard demodulation instead of the sample store
            // the decision's mode is a template parameter (aeth_fir_kernel.h: demod_block): BPSK, QPSK with a
            // separable table, QPSK with any other table -- nothing about it is tested per sample
#define AETH_DM(DMV)                                                                                                                \
            do {                                                                                                                    \
                if constexpr (kLab8) {                                                                                              \
                    if (lab8_four) { hipLaunchKernelGGL((fmi_kernel<C, SCALED, 4, true, false, VAR | V_DEMOD | (DMV)>), dim3(grid), dim3(C::WG), 0, stream, b); break; } \
                }                                                                                                                   \
                if (nt) hipLaunchKernelGGL((fmi_kernel<C, SCALED, 1, true, false, VAR | V_DEMOD | (DMV)>), dim3(grid), dim3(C::WG), 0, stream, b); \
                else hipLaunchKernelGGL((fmi_kernel<C, SCALED, 1, false, false, VAR | V_DEMOD | (DMV)>), dim3(grid), dim3(C::WG), 0, stream, b);   \
            } while (0)
            if (b.bps == 1) AETH_DM(V_DM_BPSK);
            else if (b.demod_sep) AETH_DM(0);
            else AETH_DM(V_DM_QGEN);
#undef AETH_DM
            AETH_HIP(hipGetLastError());
            return AETH_OK;
        }
    }
    if constexpr (C::F == 1 && !SCALED) {
        if (b.dec.d > 1) {                                  // decimating store (aeth_fir_exec_decim)
            // without the swizzle: with it the N = 2048 build needs 260 VGPRs and drops to one wave per SIMD (62 us
            // per 16 Mi-sample launch against 50)
            constexpr int DV = (VAR & ~V_XOR) | V_DECIM;
            if (nt) hipLaunchKernelGGL((fmi_kernel<C, false, 2, true, false, DV>), dim3(grid), dim3(C::WG), 0, stream, b);
            else hipLaunchKernelGGL((fmi_kernel<C, false, 2, false, false, DV>), dim3(grid), dim3(C::WG), 0, stream, b);
            AETH_HIP(hipGetLastError());
            return AETH_OK;
        }
    }
    // A launch that runs on its own (one queue, or the head of a chain) issues the next window's loads in four
    // instalments between the passes of the forward transform instead of one burst (V_SPREAD): 54.25 -> 53.34 us per
    // 16 Mi-sample launch on one queue, bit-identical output; beside another launch the burst form wins (47.5 against
    // 47.8 us), so chained launches keep it (profiles/r03_fir_lab_variants.txt).  AETH_FIR_SPREAD=0/1 forces either.
    if constexpr (C::F == 1 && !SCALED && C::NPASS == 3) {
        const int sp = aeth::tuning_int("AETH_FIR_SPREAD", -1);
        const bool chained = ctx->overlap && !ctx->stream_shared && ctx->last_chained;
        if (nt && (sp < 0 ? !chained : sp != 0)) {
            hipLaunchKernelGGL((fmi_kernel<C, false, 1, true, false, VAR | V_SPREAD>), dim3(grid), dim3(C::WG), 0, stream, b);
            AETH_HIP(hipGetLastError());
            return AETH_OK;
        }
    }
    if (nt) hipLaunchKernelGGL((fmi_kernel<C, SCALED, 1, true, false, VAR>), dim3(grid), dim3(C::WG), 0, stream, b);
    else hipLaunchKernelGGL((fmi_kernel<C, SCALED, 1, false, false, VAR>), dim3(grid), dim3(C::WG), 0, stream, b);
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

int dispatch_fmi(aeth_ctx *ctx, size_t fft_len, const FmiArgs &a, hipStream_t stream = nullptr)
{
    if (!stream) stream = aeth::ctx_stream(ctx);
    aeth::DeviceGuard dev_guard(ctx->device);
    const bool scaled = a.chirp != nullptr || !(a.s_fwd == 1.0f && a.s_bwd == 1.0f);
#if AETH_LAB
    // lab (make LAB=1): the demodulating build of N = 2048 with 256 lanes x 8 points (radices 8.8.8.4: three exchanges
    // per transform, half the registers, four waves per SIMD) -- the kernel is bound by its transform time at two
    // waves per SIMD; does twice the occupancy buy more than the extra exchange costs?
    if (fft_len == 2048 && a.bits && !a.chirp && aeth::lab_int("AETH_DEMOD_P8", 0)) {
        FmiArgs b8 = a;
        b8.twL = nullptr;                                    // the plan's lane table is laid out for 16 points per lane
        using C8 = Cfg<2048, 8, 8, 8, 8, 4>;
        return scaled ? launch_fmi<C8, true>(ctx, b8, stream) : launch_fmi<C8, false>(ctx, b8, stream);
    }
#endif
#define AETH_BODY(NN)                                                            \
    return scaled ? launch_fmi<typename CfgFor<NN>::type, true>(ctx, a, stream)  \
                  : launch_fmi<typename CfgFor<NN>::type, false>(ctx, a, stream)
    AETH_POW2_SWITCH(fft_len, AETH_BODY, return aeth::set_error(AETH_E_UNSUPPORTED, "fused FFT*H*IFFT: length %zu", fft_len))
#undef AETH_BODY
}

bool is_pow2(size_t n) { return n && (n & (n - 1)) == 0; }

// [a, a + na) and [b, b + nb) share a byte?  (elements of 8 bytes)
bool touch(const aeth_cf32 *a, size_t na, const aeth_cf32 *b, size_t nb)
{
    if (!a || !b || !na || !nb) return false;
    const uintptr_t a0 = (uintptr_t)a, a1 = a0 + na * sizeof(aeth_cf32), b0 = (uintptr_t)b, b1 = b0 + nb * sizeof(aeth_cf32);
    return a0 < b1 && b0 < a1;
}

}  // namespace

namespace aeth {

// Chirp-z transform of `batch` frames of n samples in ONE launch: x*chirp -> fwd_M -> *filt -> bwd_M -> *chirp,
// zero-padded to M = sub->len in registers (aeth_fft_big.hip: fft_run_bluestein, M <= 4096).
int fmi_bluestein(aeth_fft *sub, const float2 *in, float2 *out, size_t n, size_t batch, const float2 *chirp,
                  const float2 *filt, int conj, float scale)
{
    FmiArgs a;
    a.in = (const cf *)in; a.out = (cf *)out; a.hist = nullptr; a.Hf = (const cf *)filt;
    a.twN = (const cf *)sub->tw_dev; a.twL = (const cf *)sub->tw_lane_dev;
    a.n = (long long)(n * batch); a.nblocks = (long long)batch;
    a.hop = (int)n; a.ov = 0; a.nhist = 0;
    a.s_fwd = 1.0f; a.s_bwd = scale;
    a.chirp = (const cf *)chirp; a.frame_n = (int)n; a.conj = conj;
    return dispatch_fmi(sub->ctx, sub->len, a);
}

}  // namespace aeth

extern "C" {

/* benches/benches.rs:410-416, per frame, in place */
int aeth_fft_mul_ifft(aeth_fft *plan, aeth_cf32 *frames, size_t n_total, size_t batch, const aeth_cf32 *sig,
                      size_t n_sig, int kind_fwd, float x_fwd, int kind_bwd, float x_bwd)
{
    AETH_REQUIRE(plan, AETH_E_ARG, "plan is null");
    AETH_REQUIRE(n_total == batch * plan->len, AETH_E_LEN, AETH_MSG_FFT_LEN);      /* fft.rs:185-189 */
    AETH_REQUIRE(n_sig == plan->len, AETH_E_LEN, AETH_MSG_VEC_LEN);                 /* vecops.rs:100-104 */
    AETH_REQUIRE(kind_fwd >= 0 && kind_fwd <= 3 && kind_bwd >= 0 && kind_bwd <= 3, AETH_E_ARG, "bad scale kind");
    if (batch == 0) return AETH_OK;
    AETH_REQUIRE(frames && sig, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(aeth::aligned8(frames) && aeth::aligned8(sig), AETH_E_ALIGN, "pointer not 8-byte aligned");
    if (plan->algo != aeth::FFT_ALGO_POW2 || plan->len > 4096) {
        // generic lengths (and the 8192-point frame, too wide for the fused kernel's registers): the three trait calls, unfused
        int rc = aeth_fft_exec(plan, frames, n_total, frames, batch, AETH_SIGN_REF_FWD, kind_fwd, x_fwd);
        if (rc) return rc;
        rc = aeth_vec_mul_frames(plan->ctx, frames, plan->len, batch, sig, n_sig);    // one launch, sig shared by every frame
        if (rc) return rc;
        return aeth_fft_exec(plan, frames, n_total, frames, batch, AETH_SIGN_REF_BWD, kind_bwd, x_bwd);
    }
    FmiArgs a;
    a.in = (const cf *)frames; a.out = (cf *)frames; a.hist = nullptr; a.Hf = (const cf *)sig;
    a.twN = (const cf *)plan->tw_dev; a.twL = (const cf *)plan->tw_lane_dev; a.n = (long long)n_total; a.nblocks = (long long)batch;
    a.hop = (int)plan->len; a.ov = 0; a.nhist = 0;
    a.s_fwd = aeth_scale_factor(kind_fwd, plan->len, x_fwd);
    a.s_bwd = aeth_scale_factor(kind_bwd, plan->len, x_bwd);
    return dispatch_fmi(plan->ctx, plan->len, a);
}

/* frames.vec_rfft(fft, s).vec_mul(&sig).vec_rifft(fft, s) per frame (benches/benches.rs:410-416), then
 * Modulation::demod_naive on the result (examples/modem.rs:28-31) -- BASELINE config 4's receive side.  For the
 * one-frame-per-workgroup lengths the correlator output is demodulated in registers and only the bit bytes are
 * written (8 B read + bits per sample instead of 8 R + 8 W + 8 R + bits); `frames` is not modified. */
int aeth_fft_mul_ifft_demod(aeth_fft *plan, const aeth_cf32 *frames, size_t n_total, size_t batch, const aeth_cf32 *sig,
                            size_t n_sig, int kind_fwd, float x_fwd, int kind_bwd, float x_bwd, int bps,
                            const aeth_cf32 *table, uint8_t *bits_out, size_t nbits_out, int compat)
{
    AETH_REQUIRE(plan, AETH_E_ARG, "plan is null");
    AETH_REQUIRE(n_total == batch * plan->len, AETH_E_LEN, AETH_MSG_FFT_LEN);
    AETH_REQUIRE(n_sig == plan->len, AETH_E_LEN, AETH_MSG_VEC_LEN);
    AETH_REQUIRE(kind_fwd >= 0 && kind_fwd <= 3 && kind_bwd >= 0 && kind_bwd <= 3, AETH_E_ARG, "bad scale kind");
    AETH_REQUIRE(bps == 1 || bps == 2, AETH_E_UNSUPPORTED, "bits_per_symbol %d: BPSK (1) or QPSK (2)", bps);
    AETH_REQUIRE(nbits_out == n_total * (size_t)bps, AETH_E_LEN, "output holds %zu bits, input gives %zu", nbits_out, n_total * (size_t)bps);
    if (batch == 0) return AETH_OK;
    AETH_REQUIRE(frames && sig && bits_out, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(aeth::aligned8(frames) && aeth::aligned8(sig) && ((uintptr_t)bits_out % (size_t)bps) == 0, AETH_E_ALIGN, "pointer alignment");
    static const aeth_cf32 kB[2] = {{1.f, 1.f}, {-1.f, -1.f}};                              /* modulation.rs:77 */
    static const aeth_cf32 kQ[4] = {{1.f, 1.f}, {-1.f, 1.f}, {1.f, -1.f}, {-1.f, -1.f}};    /* modulation.rs:87-92 */
    const aeth_cf32 *tb = table ? table : (bps == 1 ? kB : kQ);
    if (plan->algo != aeth::FFT_ALGO_POW2 || plan->len < 1024 || plan->len > 4096) {
        // other lengths: the chain on a copy in the plan's temp, then the stand-alone demodulator
        int rc = aeth::fft_ensure_tmp(plan, n_total); if (rc) return rc;
        rc = aeth_copy_dev(plan->ctx, plan->tmp_dev, frames, n_total * sizeof(float2)); if (rc) return rc;
        rc = aeth_fft_mul_ifft(plan, (aeth_cf32 *)plan->tmp_dev, n_total, batch, sig, n_sig, kind_fwd, x_fwd, kind_bwd, x_bwd);
        if (rc) return rc;
        return aeth_demod_naive(plan->ctx, (const aeth_cf32 *)plan->tmp_dev, n_total, bps, table, bits_out, nbits_out, compat);
    }
    FmiArgs a;
    a.in = (const cf *)frames; a.out = nullptr; a.hist = nullptr; a.Hf = (const cf *)sig;
    a.twN = (const cf *)plan->tw_dev; a.twL = (const cf *)plan->tw_lane_dev; a.n = (long long)n_total; a.nblocks = (long long)batch;
    a.hop = (int)plan->len; a.ov = 0; a.nhist = 0;
    a.s_fwd = aeth_scale_factor(kind_fwd, plan->len, x_fwd);
    a.s_bwd = aeth_scale_factor(kind_bwd, plan->len, x_bwd);
    a.bits = bits_out; a.bps = bps; a.demod_compat = compat;
    for (int i = 0; i < (bps == 1 ? 2 : 4); i++) { cf t = {tb[i].re, tb[i].im}; a.tab[i] = t; }
    // demod_naive scans 2 * bps candidates (modulation.rs:135): all four for QPSK
    a.demod_sep = bps == 2 && tb[0].re == tb[2].re && tb[1].re == tb[3].re && tb[0].im == tb[1].im && tb[2].im == tb[3].im;
    // out of place (frames are only read), so consecutive calls on disjoint buffers can run on the context's two
    // queues like consecutive aeth_fir_exec calls do (aeth_ctx_set_overlap): the drain of one beside the fill of the next
    // (a reference signal that the previous chained launch may still be writing keeps this call on the in-order stream)
    aeth_ctx *cx = plan->ctx;
    const bool sig_busy = cx->chain_last >= 0 && (uintptr_t)sig < cx->last_out[1] && cx->last_out[0] < (uintptr_t)(sig + n_sig);
    hipStream_t lane = sig_busy ? aeth::ctx_stream(cx)
                                : aeth::ctx_fir_lane(cx, (uintptr_t)frames, (uintptr_t)(frames + n_total), (uintptr_t)bits_out,
                                                     (uintptr_t)(bits_out + nbits_out));
    return dispatch_fmi(plan->ctx, plan->len, a, lane);
}

int aeth_fir_create(aeth_ctx *ctx, const aeth_cf32 *taps, size_t ntaps, size_t fft_len, aeth_fir **out)
{
    AETH_REQUIRE(ctx && out, AETH_E_ARG, "null argument");
    *out = nullptr;
    AETH_REQUIRE(taps && ntaps >= 1, AETH_E_ARG, "need at least one tap");
    AETH_REQUIRE(is_pow2(fft_len) && fft_len >= 2 && fft_len <= 4096, AETH_E_UNSUPPORTED,
                 "fft_len %zu: need a power of two in [2, 4096]", fft_len);
    AETH_REQUIRE(2 * ntaps <= fft_len, AETH_E_ARG, "fft_len %zu < 2*ntaps (%zu)", fft_len, 2 * ntaps);
    aeth::DeviceGuard g(ctx->device);
    aeth_fir *f = new (std::nothrow) aeth_fir();
    AETH_REQUIRE(f, AETH_E_NOMEM, "out of host memory");
    f->ctx = ctx; f->ntaps = ntaps; f->fft_len = fft_len;
    // outputs per block: the one-frame-per-workgroup lengths round it down to 64 samples so that every
    // window and output block starts on a 512-byte boundary (descriptor loads, full lines); the short
    // windows, several to a workgroup, are bound by the block count and keep every output they can
    // (measured: fft_len 64, 8 taps: hop 57 -> 170 GS/s, hop 48 -> 155 GS/s)
    size_t L = fft_len - ntaps + 1;
    f->hop = (fft_len >= 512 && L >= 64) ? (L / 64) * 64 : L;
    int rc = aeth_fft_create(ctx, fft_len, 1, &f->fft);
    if (rc == AETH_OK) {
        hipError_t e = hipMalloc((void **)&f->Hf, fft_len * sizeof(float2));
        if (e != hipSuccess) rc = aeth::hip_fail(e, "hipMalloc");
    }
    if (rc == AETH_OK) {
        std::vector<aeth_cf32> padded(fft_len, aeth_cf32{0.f, 0.f});
        for (size_t k = 0; k < ntaps; k++) padded[k] = taps[k];
        rc = aeth_upload(ctx, f->Hf, padded.data(), fft_len * sizeof(float2));
    }
    // H = fwd(taps || 0), then Scale::N folded in (x 1/N is exact for a power of two)
    if (rc == AETH_OK)
        rc = aeth_fft_exec(f->fft, (aeth_cf32 *)f->Hf, fft_len, (aeth_cf32 *)f->Hf, 1, AETH_SIGN_REF_FWD,
                           AETH_SCALE_N, 0.f);
    if (rc == AETH_OK) rc = aeth_ctx_sync(ctx);
    if (rc != AETH_OK) { aeth_fir_destroy(f); return rc; }
    *out = f;
    return AETH_OK;
}

int aeth_fir_destroy(aeth_fir *f)
{
    if (!f) return AETH_OK;
    aeth::DeviceGuard g(f->ctx->device);
    (void)hipStreamSynchronize(aeth::ctx_stream(f->ctx));
    if (f->Hf) (void)hipFree(f->Hf);
    if (f->fft) aeth_fft_destroy(f->fft);
    delete f;
    return AETH_OK;
}

size_t aeth_fir_ntaps(const aeth_fir *f) { return f ? f->ntaps : 0; }
size_t aeth_fir_fft_len(const aeth_fir *f) { return f ? f->fft_len : 0; }
size_t aeth_fir_hop(const aeth_fir *f) { return f ? f->hop : 0; }

int aeth_fir_exec(aeth_fir *f, const aeth_cf32 *hist, const aeth_cf32 *in, size_t n, aeth_cf32 *out)
{
    AETH_REQUIRE(f, AETH_E_ARG, "fir is null");
    if (n == 0) return AETH_OK;
    AETH_REQUIRE(in && out, AETH_E_ARG, "null pointer");
    // blocks run concurrently and their windows reach into the neighbours' outputs: ANY overlap of the output range
    // with the input or the history reads samples that were already overwritten (out = in + 100 as much as out = in)
    AETH_REQUIRE(!touch(out, n, in, n) && !touch(out, n, hist, f->ntaps - 1), AETH_E_ARG,
                 "FIR cannot run in place: the output range overlaps the input (or its history)");
    AETH_REQUIRE(aeth::aligned8(in) && aeth::aligned8(out) && aeth::aligned8(hist), AETH_E_ALIGN,
                 "pointer not 8-byte aligned");
    FmiArgs a;
    a.in = (const cf *)in; a.out = (cf *)out; a.hist = (const cf *)hist; a.Hf = (const cf *)f->Hf; a.twN = (const cf *)f->fft->tw_dev; a.twL = (const cf *)f->fft->tw_lane_dev;
    a.n = (long long)n; a.hop = (int)f->hop; a.ov = (int)(f->fft_len - f->hop); a.nhist = (int)(f->ntaps - 1);
    a.nblocks = (long long)((n + f->hop - 1) / f->hop);
    a.s_fwd = 1.0f; a.s_bwd = 1.0f;
    // independent consecutive launches alternate between the context's two queues (aeth_ctx_set_overlap); a history
    // buffer is usually the tail of something just written, so such calls stay on the in-order stream
    hipStream_t lane = hist ? aeth::ctx_stream(f->ctx)
                            : aeth::ctx_fir_lane(f->ctx, (uintptr_t)in, (uintptr_t)(in + n), (uintptr_t)out, (uintptr_t)(out + n));
    return dispatch_fmi(f->ctx, f->fft_len, a, lane);
}

/* fir, then sampling::downsample(&y, &mut dst) (src/sampling.rs:28-42) in one pass: out[i] = y[i * dec],
 * dec = n / n_out; the filter's kernel simply does not store the samples downsample would skip */
int aeth_fir_exec_decim(aeth_fir *f, const aeth_cf32 *hist, const aeth_cf32 *in, size_t n, aeth_cf32 *out, size_t n_out)
{
    AETH_REQUIRE(f, AETH_E_ARG, "fir is null");
    if (n == 0 && n_out == 0) return AETH_OK;
    AETH_REQUIRE(n_out > 0 && n % n_out == 0, AETH_E_ARG, AETH_MSG_DECIM);          /* sampling.rs:32-36 */
    AETH_REQUIRE(n >= n_out, AETH_E_LEN, "downsample from an empty src (the reference panics: index out of bounds)");
    const size_t dec = n / n_out;
    if (dec == 1) return aeth_fir_exec(f, hist, in, n, out);
    AETH_REQUIRE(in && out, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(!touch(out, n_out, in, n) && !touch(out, n_out, hist, f->ntaps - 1), AETH_E_ARG,
                 "FIR cannot run in place: the output range overlaps the input (or its history)");
    AETH_REQUIRE(aeth::aligned8(in) && aeth::aligned8(out) && aeth::aligned8(hist), AETH_E_ALIGN,
                 "pointer not 8-byte aligned");
    AETH_REQUIRE(f->fft_len >= 1024 && f->fft_len <= 4096, AETH_E_UNSUPPORTED,
                 "decimating store: fft_len %zu (one-block-per-workgroup lengths 1024 .. 4096 only)", f->fft_len);
    AETH_REQUIRE(n < ((size_t)1 << 31), AETH_E_UNSUPPORTED, "decimating store: %zu samples (32-bit index arithmetic)", n);
    FmiArgs a;
    a.in = (const cf *)in; a.out = (cf *)out; a.hist = (const cf *)hist; a.Hf = (const cf *)f->Hf;
    a.twN = (const cf *)f->fft->tw_dev; a.twL = (const cf *)f->fft->tw_lane_dev;
    a.n = (long long)n; a.hop = (int)f->hop; a.ov = (int)(f->fft_len - f->hop); a.nhist = (int)(f->ntaps - 1);
    a.nblocks = (long long)((n + f->hop - 1) / f->hop);
    a.s_fwd = 1.0f; a.s_bwd = 1.0f;
    a.dec = aeth::make_fastdiv((uint32_t)dec); a.n_out = (long long)n_out;
    return dispatch_fmi(f->ctx, f->fft_len, a, aeth::ctx_stream(f->ctx));
}

}  // extern "C"

namespace aeth {

int fir_exec_on(aeth_fir *f, hipStream_t stream, const aeth_cf32 *hist, const aeth_cf32 *in, size_t n, aeth_cf32 *out)
{
    FmiArgs a;
    a.in = (const cf *)in; a.out = (cf *)out; a.hist = (const cf *)hist; a.Hf = (const cf *)f->Hf;
    a.twN = (const cf *)f->fft->tw_dev; a.twL = (const cf *)f->fft->tw_lane_dev;
    a.n = (long long)n; a.hop = (int)f->hop; a.ov = (int)(f->fft_len - f->hop); a.nhist = (int)(f->ntaps - 1);
    a.nblocks = (long long)((n + f->hop - 1) / f->hop);
    a.s_fwd = 1.0f; a.s_bwd = 1.0f;
    return dispatch_fmi(f->ctx, f->fft_len, a, stream);
}

}  // namespace aeth

extern "C" {

int aeth_fir_exec_host(aeth_fir *f, const aeth_cf32 *hist, const aeth_cf32 *in, size_t n, aeth_cf32 *out)
{
    AETH_REQUIRE(f, AETH_E_ARG, "fir is null");
    if (n == 0) return AETH_OK;
    AETH_REQUIRE(in && out, AETH_E_ARG, "null pointer");
    aeth_ctx *ctx = f->ctx;
    aeth::DeviceGuard dev_guard(ctx->device);
    const size_t nh = f->ntaps - 1;
    const size_t bytes = n * sizeof(float2);
    int rc = aeth::ctx_stage(ctx, 0, (n + nh) * sizeof(float2)); if (rc) return rc;
    rc = aeth::ctx_stage(ctx, 1, bytes); if (rc) return rc;
    float2 *dh = (float2 *)ctx->stage[0];
    float2 *din = dh + nh;     // nh*8 bytes in: keeps 8-byte alignment
    if (hist && nh) AETH_HIP(hipMemcpyAsync(dh, hist, nh * sizeof(float2), hipMemcpyHostToDevice, aeth::ctx_stream(ctx)));
    AETH_HIP(hipMemcpyAsync(din, in, bytes, hipMemcpyHostToDevice, aeth::ctx_stream(ctx)));
    rc = aeth_fir_exec(f, hist ? (const aeth_cf32 *)dh : nullptr, (const aeth_cf32 *)din, n, (aeth_cf32 *)ctx->stage[1]);
    if (rc) return rc;
    AETH_HIP(hipMemcpyAsync(out, ctx->stage[1], bytes, hipMemcpyDeviceToHost, aeth::ctx_stream(ctx)));
    AETH_HIP(hipStreamSynchronize(aeth::ctx_stream(ctx)));
    return AETH_OK;
}

}  // extern "C"
