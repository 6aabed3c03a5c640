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
// loop over blocks.  hop is rounded down to a multiple of 64 samples so that every
// block's window and output start on a 512-byte boundary.
#include "aeth_internal.h"
#include "aeth_fft_core.h"
#include "aeth_fft_plan.h"

#include <new>
#include <vector>

using namespace aeth::fftk;

struct aeth_fir {
    aeth_ctx *ctx = nullptr;
    size_t ntaps = 0, fft_len = 0, hop = 0;
    aeth_fft *fft = nullptr;      // owns the twiddle table; used once to transform the taps
    float2 *Hf = nullptr;         // fwd(taps || 0) / N  (1/N folded in: exact, N is a power of two)
};

namespace {

struct FmiArgs {
    const cf *in;
    cf *out;
    const cf *hist;       // ntaps-1 samples preceding in[0], or null
    const cf *Hf;         // N spectrum multipliers, natural order
    const cf *twN;
    long long n;          // samples in `in` / outputs wanted
    long long nblocks;
    int hop, ov, nhist;
    float s_fwd, s_bwd;
};

template <class C>
__global__ __launch_bounds__(C::WG) void fmi_kernel(FmiArgs a)
{
    __shared__ cf lds_all[C::LDS_ELEMS];
    const int tid = threadIdx.x % C::T;
    const int fl = threadIdx.x / C::T;
    cf *lds = lds_all + fl * C::LDS_FRAME;

    cf tw[C::TW];
    load_twiddles<C>(tw, a.twN, tid);
    cf H[C::P];
#pragma unroll
    for (int m = 0; m < C::P; m++) H[m] = a.Hf[tid + m * C::T];

    const long long ngroups = (a.nblocks + C::F - 1) / C::F;
    for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const long long blk = g * C::F + fl;
        const bool active = blk < a.nblocks;
        const long long win0 = blk * a.hop - a.ov;          // first input sample of the window
        cf w[C::P];
        if (active && win0 >= 0 && win0 + C::N <= a.n) {
            const cf *src = a.in + win0 + tid;
#pragma unroll
            for (int m = 0; m < C::P; m++) w[m] = cswap(src[m * C::T]);     // fwd = +j exponent
        } else {
#pragma unroll
            for (int m = 0; m < C::P; m++) {
                const long long gi = win0 + tid + m * C::T;
                cf v = mk(0.f, 0.f);
                if (active) {
                    if (gi >= 0) { if (gi < a.n) v = a.in[gi]; }
                    else if (a.hist && gi >= -(long long)a.nhist) v = a.hist[a.nhist + gi];
                }
                w[m] = cswap(v);
            }
        }
        fft_in_regs<C>(w, tw, lds, tid);
#pragma unroll
        for (int m = 0; m < C::P; m++) {
            cf X = cscale(cswap(w[m]), a.s_fwd);            // Scale of vec_rfft
            w[m] = cmul(X, H[m]);                           // vec_mul (vecops.rs:99-112)
        }
        fft_in_regs<C>(w, tw, lds, tid);                    // bwd = -j exponent
        if (active) {
            const long long out0 = blk * a.hop - a.ov + tid;    // output index of slot m=0 (may be < blk*hop)
#pragma unroll
            for (int m = 0; m < C::P; m++) {
                const int e = tid + m * C::T;
                const long long o = out0 + m * C::T;
                if (e >= a.ov && o < a.n) a.out[o] = cscale(w[m], a.s_bwd);
            }
        }
    }
}

template <class C>
int launch_fmi(aeth_ctx *ctx, const FmiArgs &a)
{
    long long ngroups = (a.nblocks + C::F - 1) / C::F;
    long long cap = (long long)ctx->num_cus * 8;
    int grid = (int)(ngroups < cap ? ngroups : cap);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((fmi_kernel<C>), dim3(grid), dim3(C::WG), 0, ctx->stream, a);
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

int dispatch_fmi(aeth_ctx *ctx, size_t fft_len, const FmiArgs &a)
{
#define AETH_BODY(NN) return launch_fmi<typename CfgFor<NN>::type>(ctx, a)
    AETH_POW2_SWITCH(fft_len, AETH_BODY, return aeth::set_error(AETH_E_UNSUPPORTED, "fused FFT*H*IFFT: length %zu", fft_len))
#undef AETH_BODY
}

bool is_pow2(size_t n) { return n && (n & (n - 1)) == 0; }

}  // namespace

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
    if (plan->algo != aeth::FFT_ALGO_POW2) {
        // generic lengths: the three trait calls, unfused
        int rc = aeth_fft_exec(plan, frames, n_total, frames, batch, AETH_SIGN_REF_FWD, kind_fwd, x_fwd);
        if (rc) return rc;
        for (size_t f = 0; f < batch && rc == AETH_OK; f++)
            rc = aeth_vec_mul(plan->ctx, frames + f * plan->len, plan->len, sig, n_sig);
        if (rc) return rc;
        return aeth_fft_exec(plan, frames, n_total, frames, batch, AETH_SIGN_REF_BWD, kind_bwd, x_bwd);
    }
    FmiArgs a;
    a.in = (const cf *)frames; a.out = (cf *)frames; a.hist = nullptr; a.Hf = (const cf *)sig;
    a.twN = plan->tw_dev; a.n = (long long)n_total; a.nblocks = (long long)batch;
    a.hop = (int)plan->len; a.ov = 0; a.nhist = 0;
    a.s_fwd = aeth_scale_factor(kind_fwd, plan->len, x_fwd);
    a.s_bwd = aeth_scale_factor(kind_bwd, plan->len, x_bwd);
    return dispatch_fmi(plan->ctx, plan->len, a);
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
    size_t L = fft_len - ntaps + 1;
    f->hop = (L >= 64) ? (L / 64) * 64 : L;
    int rc = aeth_fft_create(ctx, fft_len, 1, &f->fft);
    if (rc == AETH_OK) {
        hipError_t e = hipMalloc((void **)&f->Hf, fft_len * sizeof(cf));
        if (e != hipSuccess) rc = aeth::hip_fail(e, "hipMalloc");
    }
    if (rc == AETH_OK) {
        std::vector<aeth_cf32> padded(fft_len, aeth_cf32{0.f, 0.f});
        for (size_t k = 0; k < ntaps; k++) padded[k] = taps[k];
        rc = aeth_upload(ctx, f->Hf, padded.data(), fft_len * sizeof(cf));
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
    (void)hipStreamSynchronize(f->ctx->stream);
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
    AETH_REQUIRE(in != out, AETH_E_ARG, "FIR cannot run in place (blocks overlap)");
    AETH_REQUIRE(aeth::aligned8(in) && aeth::aligned8(out) && aeth::aligned8(hist), AETH_E_ALIGN,
                 "pointer not 8-byte aligned");
    FmiArgs a;
    a.in = (const cf *)in; a.out = (cf *)out; a.hist = (const cf *)hist; a.Hf = f->Hf; a.twN = f->fft->tw_dev;
    a.n = (long long)n; a.hop = (int)f->hop; a.ov = (int)(f->fft_len - f->hop); a.nhist = (int)(f->ntaps - 1);
    a.nblocks = (long long)((n + f->hop - 1) / f->hop);
    a.s_fwd = 1.0f; a.s_bwd = 1.0f;
    return dispatch_fmi(f->ctx, f->fft_len, a);
}

int aeth_fir_exec_host(aeth_fir *f, const aeth_cf32 *hist, const aeth_cf32 *in, size_t n, aeth_cf32 *out)
{
    AETH_REQUIRE(f, AETH_E_ARG, "fir is null");
    if (n == 0) return AETH_OK;
    AETH_REQUIRE(in && out, AETH_E_ARG, "null pointer");
    aeth_ctx *ctx = f->ctx;
    const size_t nh = f->ntaps - 1;
    const size_t bytes = n * sizeof(cf);
    int rc = aeth::ctx_stage(ctx, 0, (n + nh) * sizeof(cf)); if (rc) return rc;
    rc = aeth::ctx_stage(ctx, 1, bytes); if (rc) return rc;
    cf *dh = (cf *)ctx->stage[0];
    cf *din = dh + nh;     // nh*8 bytes in: keeps 8-byte alignment
    if (hist && nh) AETH_HIP(hipMemcpyAsync(dh, hist, nh * sizeof(cf), hipMemcpyHostToDevice, ctx->stream));
    AETH_HIP(hipMemcpyAsync(din, in, bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = aeth_fir_exec(f, hist ? (const aeth_cf32 *)dh : nullptr, (const aeth_cf32 *)din, n, (aeth_cf32 *)ctx->stage[1]);
    if (rc) return rc;
    AETH_HIP(hipMemcpyAsync(out, ctx->stage[1], bytes, hipMemcpyDeviceToHost, ctx->stream));
    AETH_HIP(hipStreamSynchronize(ctx->stream));
    return AETH_OK;
}

}  // extern "C"
