// aeth_fft.hip -- batched complex FFT behind the reference's `Fft` trait
// (src/fft.rs:48-77) / `Cfft` (src/fft.rs:134-235).
//
// Kernel paths (chosen per length at plan time):
//   stockham_pow2   N = 2..8192, power of two: register radix-16/8/4 Stockham, one
//                   frame per T = N/P lanes, data exchanged through LDS between
//                   passes, twiddles held in registers across a persistent loop over
//                   frames.  16 B/sample of HBM traffic (8 R + 8 W), scale fused.
//   stockham_mixed_ragged  every other N = 2^a 3^b 5^c up to 20480 (one workgroup, up to
//                   160 KiB of LDS per frame), 16384, factor 7 up to 4096, factors 11 / 13
//                   up to 2048: the same idea with
//                   passes that do not share a lane shape (aeth_fft_ragged.h/.hip),
//                   one measured decomposition per length.
//   stockham_mixed  other N <= 8192 with prime factors <= 61: one workgroup per frame,
//                   LDS ping-pong, radix 2/3/4/5/7/8 butterflies (3 and 5 in
//                   sum/difference form so that constant inputs give exactly-zero
//                   bins, as the reference's own FFT tests expect: fft.rs:93-104,
//                   vecops.rs:443-463 at N=100), generic O(r^2) for other primes.
//   fourstep_mixed  N > 8192 = n1 * n2 with both factors <= 8192: three transposes around
//                   the batched transforms of the factors (aeth_fft_big.hip).
//   fourstep_pow2   N = 2^14..2^24: N1 x N2 decomposition, two launches through the plan's scratch (BASELINE
//                   config 5, N = 65536 = 256 x 256); 2^23 and 2^24 in the deep form (columns, nested four-step
//                   rows, transpose: four launches, twice the scratch) -- aeth_fft_big.hip.
//   bluestein       everything else: chirp-z through a power-of-two convolution.
#include "aeth_internal.h"
#include "aeth_fft_core.h"
#include "aeth_fft_plan.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

using namespace aeth::fftk;

namespace {

// =============================== stockham_pow2 ================================
#ifndef AETH_STAGE_T
#define AETH_STAGE_T 24
#endif

// Frames of a few lanes (T < AETH_STAGE_T): a lane's points are 8*T bytes apart in memory, so direct accesses touch a
// separate 64-byte segment per lane and instruction.  The workgroup's F frames are contiguous, so they go through
// LDS instead: coalesced 8-byte accesses on the memory side, one pad slot per frame on the LDS side (frames would
// otherwise sit a multiple of the bank count apart).  Needs the CU full of waves to hide the extra hop.
template <class C> constexpr bool pow2_staged_io() { return C::T < AETH_STAGE_T && C::F > 1 && C::N >= 4; }
template <class C> struct SingleImage : C {
    static constexpr bool DB = false;
    static constexpr int LDS_TOTAL = C::LDS_ELEMS;
};

// NT: frames are streamed with the non-temporal hint (batches beyond the cache; aeth_internal.h)
template <class C0, int S, bool NT>
__global__ __launch_bounds__(C0::WG) void fft_pow2_kernel(const cf *in, cf *out,
                                                          const cf *__restrict__ twL, size_t batch, float scale, int mirror)
{
    // mirror: the frame is stored with its halves swapped (vec_mirror, vecops.rs:157-161, behind vec_rfft as in
    // util/plot.rs:59-61): output slot m goes where slot m ^ P/2 would, i.e. element e to e ^ N/2 -- two base
    // offsets instead of one, no extra pass over memory
    const int msh = mirror ? (C0::P / 2) * C0::T : 0;
    constexpr bool STAGED = pow2_staged_io<C0>();
    // staged configurations: ONE exchange image, and the I/O staging area shares its LDS (the image is idle
    // while frames are copied in and out) -- LDS per workgroup is what bounds the waves per CU here
    using C = typename std::conditional<STAGED, SingleImage<C0>, C0>::type;
    constexpr int IO_ELEMS = STAGED ? C::F * (C::N + 1) : 0;
    constexpr int LDS_N = C::LDS_TOTAL > IO_ELEMS ? C::LDS_TOTAL : IO_ELEMS;
    __shared__ cf lds_all[LDS_N];
    cf *lds_io = lds_all;
    constexpr bool WHOLE = C::F == 1 && C::IDLE == 0;       // the workgroup is exactly one frame
    const int tid = WHOLE ? (int)threadIdx.x : (int)(threadIdx.x % C::T);
    const int fl = WHOLE ? 0 : (int)(threadIdx.x / C::T);   // idle lanes: fl == F, the dummy LDS frame
    cf *lds = lds_all + fl * C::LDS_FRAME;

    cf tw[C::TW];
    load_twiddles_lane<C>(tw, twL, tid);

    const size_t ngroups = (batch + C::F - 1) / C::F;
    bool par = false;
    for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const size_t frame = g * C::F + fl;
        const bool active = frame < batch && (C::IDLE == 0 || fl < C::F);     // idle lanes ride along on the dummy LDS frame
        const cf *src = in + frame * C::N + tid;
        cf *dst = out + frame * C::N + tid;
        cf w[C::P];
        const size_t base = g * C::F * (size_t)C::N;
        const size_t left = batch * (size_t)C::N - base;                     // elements from this group to the end
        const int have = left < (size_t)(C::F * C::N) ? (int)left : C::F * C::N;
        if constexpr (STAGED) {
            // all loads of the group first, then the LDS writes: one memory round trip per group, not one per line
            constexpr int ITER = (C::F * C::N + C::WG - 1) / C::WG;
            cf stage[ITER];
#pragma unroll
            for (int q = 0; q < ITER; q++) {
                const int e = threadIdx.x + q * C::WG;
                stage[q] = e < have ? aeth::nt_load<NT>(in + base + e) : mk(0.f, 0.f);
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < ITER; q++) {
                const int e = threadIdx.x + q * C::WG;
                if (e < C::F * C::N) lds_io[e + e / C::N] = stage[q];
            }
            __syncthreads();
            const int fr = fl < C::F ? fl : 0;
#pragma unroll
            for (int m = 0; m < C::P; m++) w[m] = lds_io[fr * (C::N + 1) + tid + m * C::T];
        } else {
#pragma unroll
            for (int m = 0; m < C::P; m++) w[m] = active ? aeth::nt_load<NT>(src + m * C::T) : mk(0.f, 0.f);
        }
        // an odd number of exchanges per transform flips the image parity every frame
        if (fft_next_par<C>(0) == 0 || !par) fft_in_regs<C, S, 0>(w, tw, lds, tid);
        else fft_in_regs<C, S, 1>(w, tw, lds, tid);
        par = (fft_next_par<C>(0) != 0) && !par;
        const cf ss = mk(scale, scale);
        if constexpr (STAGED) {
            __syncthreads();
            if (fl < C::F) {
#pragma unroll
                for (int m = 0; m < C::P; m++)
                    lds_io[fl * (C::N + 1) + tid + m * C::T + (m < C::P / 2 ? msh : -msh)] = cscale_k(w[m], ss);
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < (C::F * C::N + C::WG - 1) / C::WG; q++) {
                const int e = threadIdx.x + q * C::WG;
                if (e < have) aeth::nt_store<NT>(out + base + e, lds_io[e + e / C::N]);
            }
        } else if (active) {
#pragma unroll
            for (int m = 0; m < C::P; m++) aeth::nt_store<NT>(dst + m * C::T + (m < C::P / 2 ? msh : -msh), cscale_k(w[m], ss));
        }
    }
}

// One-frame-per-workgroup sizes (N >= 1024): the same pipeline as the fused FIR kernel --
// branch-free loop, frames through buffer descriptors (a zero-length descriptor turns the
// prefetch of a frame past the batch into a no-op), next frame prefetched into registers
// while this one is transformed, table loads drained once before the loop.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <class C, int S, bool NT>
__global__ __launch_bounds__(C::WG) void fft_pow2_stream_kernel(const cf *in, cf *out, const cf *__restrict__ twL,
                                                                 size_t batch, float scale, int mirror)
{
    const int msh = mirror ? (C::P / 2) * C::T * 8 : 0;      // see fft_pow2_kernel
    static_assert(C::F == 1, "one frame per workgroup");
    __shared__ cf lds[C::LDS_TOTAL];
    const int tid = threadIdx.x;
    auto fetch = [&](cf (&x)[C::P], size_t frame) {
        const bool active = frame < batch;
        auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf *>(in + (active ? frame : 0) * C::N), 0,
                                                    active ? C::N * 8 : 0, 0x00020000);
#pragma unroll
        for (int m = 0; m < C::P; m++)
            x[m] = __builtin_bit_cast(cf, __builtin_amdgcn_raw_buffer_load_b64(rs, (tid + m * C::T) * 8, 0, NT ? 2 : 0));
    };
    cf nx[C::P];
    fetch(nx, blockIdx.x);
    cf tw[C::TW];
    load_twiddles_lane<C>(tw, twL, tid);
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0) once, not inside the loop (see aeth_fir.hip)
    const cf ss = mk(scale, scale);
    bool par = false;
#pragma unroll 1
    for (size_t g = blockIdx.x; g < batch; g += gridDim.x) {
        cf w[C::P];
#pragma unroll
        for (int m = 0; m < C::P; m++) w[m] = nx[m];
        fetch(nx, g + gridDim.x);
        if (fft_next_par<C>(0) == 0 || !par) fft_in_regs<C, S, 0>(w, tw, lds, tid);
        else fft_in_regs<C, S, 1>(w, tw, lds, tid);
        par = (fft_next_par<C>(0) != 0) && !par;
        auto ws = __builtin_amdgcn_make_buffer_rsrc(out + g * C::N, 0, C::N * 8, 0x00020000);
#pragma unroll
        for (int m = 0; m < C::P; m++)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, cscale_k(w[m], ss)), ws, (tid + m * C::T) * 8 + (m < C::P / 2 ? msh : -msh), 0, NT ? 18 : 0);   // aux: bit 1 = non-temporal, bit 4 = sc1 (tools/nt_modes.hip)
    }
}

template <class C>
int launch_pow2(const aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale, int mirror)
{
    const aeth_ctx *ctx = plan->ctx;
    const bool nt = aeth::streams_past_cache(2 * batch * (size_t)C::N * sizeof(float2));
    size_t ngroups = (batch + C::F - 1) / C::F;
    size_t cap = (size_t)ctx->num_cus * 8;
    int grid = (int)(ngroups < cap ? ngroups : cap);
    if (grid < 1) grid = 1;
    if constexpr (C::F == 1) {
        // resident workgroups only: 4 per CU (the register prefetch puts these kernels at 2 waves per SIMD)
        size_t cap2 = (size_t)ctx->num_cus * (512 / C::WG);
        int grid2 = (int)(batch < cap2 ? batch : cap2);
        if (grid2 < 1) grid2 = 1;
        if (!aeth::lab_int("AETH_FFT_NOSTREAM", 0)) {
#define AETH_FFT_STREAM(SS, NN) hipLaunchKernelGGL((fft_pow2_stream_kernel<C, SS, NN>), dim3(grid2), dim3(C::WG), 0, aeth::ctx_stream(ctx), (const cf *)in, (cf *)out, (const cf *)plan->tw_lane_dev, batch, scale, mirror)
            if (sign > 0) { if (nt) AETH_FFT_STREAM(+1, true); else AETH_FFT_STREAM(+1, false); }
            else          { if (nt) AETH_FFT_STREAM(-1, true); else AETH_FFT_STREAM(-1, false); }
#undef AETH_FFT_STREAM
            AETH_HIP(hipGetLastError());
            return AETH_OK;
        }
    }
#define AETH_FFT_PLAIN(SS, NN) hipLaunchKernelGGL((fft_pow2_kernel<C, SS, NN>), dim3(grid), dim3(C::WG), 0, aeth::ctx_stream(ctx), (const cf *)in, (cf *)out, (const cf *)plan->tw_lane_dev, batch, scale, mirror)
    if (sign > 0) { if (nt) AETH_FFT_PLAIN(+1, true); else AETH_FFT_PLAIN(+1, false); }
    else          { if (nt) AETH_FFT_PLAIN(-1, true); else AETH_FFT_PLAIN(-1, false); }
#undef AETH_FFT_PLAIN
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

int dispatch_pow2(const aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale, int mirror = 0)
{
#define AETH_BODY(NN) return launch_pow2<typename CfgFor<NN>::type>(plan, in, out, batch, sign, scale, mirror)
    AETH_POW2_SWITCH_XL(plan->len, AETH_BODY, return aeth::set_error(AETH_E_UNSUPPORTED, "stockham_pow2: length %zu", plan->len))
#undef AETH_BODY
}

template <class C>
int build_lane_table(aeth_fft *plan)
{
    const size_t elems = (size_t)LaneTable<C>::ELEMS;
    AETH_HIP(hipMalloc((void **)&plan->tw_lane_dev, elems * sizeof(float2)));
    hipLaunchKernelGGL((build_lane_twiddles<C>), dim3(1), dim3(C::T < 64 ? 64 : C::T), 0, aeth::ctx_stream(plan->ctx),
                       (const cf *)plan->tw_dev, (cf *)plan->tw_lane_dev);
    AETH_HIP(hipGetLastError());
    AETH_HIP(hipStreamSynchronize(aeth::ctx_stream(plan->ctx)));
    return AETH_OK;
}

int plan_pow2(aeth_fft *plan)
{
#define AETH_BODY(NN) return build_lane_table<typename CfgFor<NN>::type>(plan)
    AETH_POW2_SWITCH_XL(plan->len, AETH_BODY, return aeth::set_error(AETH_E_UNSUPPORTED, "stockham_pow2: length %zu", plan->len))
#undef AETH_BODY
}

// =============================== stockham_mixed ===============================
constexpr int kMixedWG = 256;
constexpr int kMaxFactors = 16;

// One Stockham pass: m = n/R butterflies of radix R, p = product of the radices before it.
struct MixedPass {
    int R, p, m;
    int tw_off;                 // this pass's twiddles in the plan's pass table: [(r-1)*p + k] = W_{pR}^{rk}
    aeth::FastDiv pdiv;         // i / p by multiply-high
};

struct MixedDesc {
    int n;
    int nfac;
    int stage;                  // 1: copy the frame to LDS first (single pass, or a generic prime radix in front)
    MixedPass pass[kMaxFactors];
};

__device__ __forceinline__ cf tw_at(const cf *__restrict__ twN, int idx) { return twN[idx]; }

// radix-3 / radix-5 in sum/difference form (exact zeros for constant input)
__device__ __forceinline__ void bfly3(cf &a, cf &b, cf &c)
{
    const float s60 = 0.86602540378443864676f;
    cf t1 = cadd(b, c), t2 = csub(b, c);
    cf y0 = cadd(a, t1);
    cf h = mk(y0.x + (-1.5f) * t1.x, y0.y + (-1.5f) * t1.y);
    cf jt = cscale(mul_mj(t2), s60);
    a = y0;
    b = cadd(h, jt);
    c = csub(h, jt);
}

__device__ __forceinline__ void bfly5(cf &x0, cf &x1, cf &x2, cf &x3, cf &x4)
{
    const float ca = -1.25f, cb = 0.55901699437494742410f;
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    cf t1 = cadd(x1, x4), t2 = cadd(x2, x3), t3 = csub(x1, x4), t4 = csub(x2, x3);
    cf t5 = cadd(t1, t2);
    cf y0 = cadd(x0, t5);
    cf m1 = mk(y0.x + ca * t5.x, y0.y + ca * t5.y);
    cf m2 = cscale(csub(t1, t2), cb);
    cf a1 = cadd(m1, m2), a2 = csub(m1, m2);
    cf b1 = mk(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
    cf b2 = mk(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
    cf jb1 = mul_mj(b1), jb2 = mul_mj(b2);
    x0 = y0;
    x1 = cadd(a1, jb1);
    x4 = csub(a1, jb1);
    x2 = cadd(a2, jb2);
    x3 = csub(a2, jb2);
}

__device__ __forceinline__ void bfly7(cf (&u)[7], const cf *__restrict__ twN, int n)
{
    // symmetric-pair form: y_k = x0 + sum_m c(km) (x_m + x_{7-m}) - j s(km) (x_m - x_{7-m})
    cf s[3], d[3];
#pragma unroll
    for (int m = 0; m < 3; m++) { s[m] = cadd(u[m + 1], u[6 - m]); d[m] = csub(u[m + 1], u[6 - m]); }
    cf x0 = u[0];
    u[0] = cadd(cadd(x0, s[0]), cadd(s[1], s[2]));
    const int step = n / 7;
#pragma unroll
    for (int k = 1; k <= 3; k++) {
        cf re = x0, im = mk(0.f, 0.f);
#pragma unroll
        for (int m = 1; m <= 3; m++) {
            cf wkm = twN[((k * m) % 7) * step];       // (cos, -sin)
            re.x += wkm.x * s[m - 1].x; re.y += wkm.x * s[m - 1].y;
            im.x += wkm.y * d[m - 1].x; im.y += wkm.y * d[m - 1].y;   // wkm.y = -sin
        }
        // x_m w^m + x_{7-m} w^{-m} = c*s_m + j*(w.y)*d_m  with w = c + j*w.y
        cf jim = mk(-im.y, im.x);
        u[k] = cadd(re, jim);
        u[7 - k] = csub(re, jim);
    }
}

// odd prime R in registers, the same symmetric-pair form: (R-1)^2/2 real multiply-adds on packed pairs instead
// of the (R-1)^2 complex ones of the O(R^2) pass, roots W_R^1..W_R^{(R-1)/2} read once (wave-uniform addresses)
template <int R>
__device__ __forceinline__ void bfly_sym(cf (&u)[R], const cf *__restrict__ twN, int n)
{
    constexpr int H = (R - 1) / 2;
    cf s[H], d[H], w[H];
    const int step = n / R;
#pragma unroll
    for (int m = 0; m < H; m++) { s[m] = cadd(u[m + 1], u[R - 1 - m]); d[m] = csub(u[m + 1], u[R - 1 - m]); w[m] = twN[(m + 1) * step]; }
    const cf x0 = u[0];
    cf sum = x0;
#pragma unroll
    for (int m = 0; m < H; m++) sum = cadd(sum, s[m]);
    u[0] = sum;
#pragma unroll
    for (int k = 1; k <= H; k++) {
        cf re = x0, im = mk(0.f, 0.f);
#pragma unroll
        for (int m = 1; m <= H; m++) {
            const int idx = (k * m) % R;                    // W_R^idx = (cos, -sin); W_R^{R-idx} is its conjugate
            const cf wa = idx > H ? w[R - idx - 1] : w[idx - 1];
            const float sn = idx > H ? -wa.y : wa.y;
            re.x += wa.x * s[m - 1].x; re.y += wa.x * s[m - 1].y;
            im.x += sn * d[m - 1].x; im.y += sn * d[m - 1].y;
        }
        const cf jim = mk(-im.y, im.x);
        u[k] = cadd(re, jim);
        u[R - k] = csub(re, jim);
    }
}

template <int R>
__device__ __forceinline__ void mixed_bfly(cf (&u)[R], const cf *__restrict__ twN, int n)
{
    if constexpr (R == 3) bfly3(u[0], u[1], u[2]);
    else if constexpr (R == 5) bfly5(u[0], u[1], u[2], u[3], u[4]);
    else if constexpr (R == 7) bfly7(u, twN, n);
    else if constexpr (R == 11 || R == 13 || R == 17 || R == 19 || R == 23) bfly_sym<R>(u, twN, n);
    else Bfly<R, -1>::run(u);
}

// One pass with a compile-time radix.  The first pass reads the frame straight from global
// memory (unless staged), the last one writes it back; both are coalesced (lane i touches
// element i + r*m resp. i + r*p with p = m in the last pass).  Twiddles come from the plan's
// per-pass table, contiguous in k, so adjacent lanes read adjacent entries.
template <int R, bool SWAP, int PTS>
__device__ __forceinline__ void mixed_pass(const MixedPass &ps, int n, int lane, int tpf, bool from_global,
                                           bool to_global, bool active, const cf *gin, cf *gout, const cf *X, cf *Y,
                                           const cf *twP, const cf *__restrict__ twN, float scale)
{
    const int m = ps.m, p = ps.p;
    const cf *tw = twP + ps.tw_off;
    auto load = [&](cf (&u)[R], int i) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            cf v;
            if (from_global) { v = active ? gin[i + r * m] : mk(0.f, 0.f); if (SWAP) v = cswap(v); }
            else v = X[i + r * m];
            u[r] = v;
        }
    };
    auto finish = [&](cf (&u)[R], int i) {
        const int k = i - (int)aeth::fdiv((uint32_t)i, ps.pdiv) * p;
        const int j = (i - k) * R + k;
        if (p > 1) {
#pragma unroll
            for (int r = 1; r < R; r++) u[r] = cmul_plain(u[r], tw[(r - 1) * p + k]);
        }
        mixed_bfly<R>(u, twN, n);
        if (to_global) {
            if (active) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    cf v = cscale(u[r], scale);
                    gout[j + r * p] = SWAP ? cswap(v) : v;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; r++) Y[j + r * p] = u[r];
        }
    };
    if constexpr (PTS == 0) {
        for (int i = lane; i < m; i += tpf) {
            cf u[R];
            load(u, i);
            finish(u, i);
        }
    } else {
        // ONE LDS image per frame (Y == X): a lane first pulls every input of all its butterflies into
        // registers (n / tpf <= PTS points), the workgroup meets, then the outputs overwrite the image.
        // Twice the frames in flight for the same LDS, which is what these latency-bound passes lack.
        constexpr int MAXB = (PTS + R - 1) / R;
        cf u[MAXB][R];
#pragma unroll
        for (int b = 0; b < MAXB; b++) {
            const int i = lane + b * tpf;
            if (i < m) load(u[b], i);
        }
        if (!from_global && !to_global) __syncthreads();
#pragma unroll
        for (int b = 0; b < MAXB; b++) {
            const int i = lane + b * tpf;
            if (i < m) finish(u[b], i);
        }
    }
}

// generic prime radix (11 .. 61): O(R^2), inputs re-read from LDS (always staged or behind another pass)
template <bool SWAP>
__device__ __forceinline__ void mixed_pass_prime(const MixedPass &ps, int n, int lane, int tpf, bool to_global,
                                                 bool active, cf *gout, const cf *X, cf *Y,
                                                 const cf *twP, const cf *__restrict__ twN, float scale)
{
    const int R = ps.R, m = ps.m, p = ps.p;
    const cf *tw = twP + ps.tw_off;
    const int rstep = n / R;
    for (int i = lane; i < m; i += tpf) {
        const int k = i - (int)aeth::fdiv((uint32_t)i, ps.pdiv) * p;
        const int j = (i - k) * R + k;
        auto ld = [&](int r) -> cf {
            cf v = X[i + r * m];
            return (p > 1 && r > 0) ? cmul_plain(v, tw[(r - 1) * p + k]) : v;
        };
        for (int q = 0; q < R; q++) {
            cf acc = ld(0);
            for (int r = 1; r < R; r++) acc = cadd_plain(acc, cmul_plain(ld(r), twN[((r * q) % R) * rstep]));
            if (to_global) {
                acc = cscale(acc, scale);
                if (active) gout[j + q * p] = SWAP ? cswap(acc) : acc;
            } else Y[j + q * p] = acc;
        }
    }
}

// PTS = 0: two LDS images per frame (ping-pong); PTS > 0: one image, a lane holds up to PTS points across the barrier
// BIGR: the register butterflies for 11 .. 23 are compiled in (144-170 VGPRs against 83-91 without them, so plans
// that have no such factor keep the lean build and its five waves per SIMD)
template <bool SWAP, int PTS, bool BIGR>
__global__ __launch_bounds__(kMixedWG) void fft_mixed_kernel(const cf *in, cf *out, const cf *__restrict__ twN,
                                                              const cf *twP, MixedDesc d, size_t batch,
                                                              float scale, int tpf, int tw_lds)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // tpf lanes per frame, kMixedWG / tpf frames per workgroup (small lengths would leave
    // most of a 256-lane workgroup idle: N = 100 has 25 radix-4 butterflies per pass)
    const int n = d.n;
    const int fpw = kMixedWG / tpf;
    const int fl = threadIdx.x / tpf;
    const int lane = threadIdx.x % tpf;
    constexpr bool INPLACE = PTS > 0;
    constexpr int IMAGES = INPLACE ? 1 : 2;
    cf *bufA = reinterpret_cast<cf *>(smem_raw) + (size_t)fl * IMAGES * n;
    cf *bufB = INPLACE ? bufA : bufA + n;
    if (tw_lds) {
        // the pass twiddles sit behind the frame images: an LDS read per factor instead of a global one
        cf *t = reinterpret_cast<cf *>(smem_raw) + (size_t)fpw * IMAGES * n;
        for (int e = threadIdx.x; e < tw_lds; e += kMixedWG) t[e] = twP[e];
        twP = t;
    }
    const size_t ngroups = (batch + fpw - 1) / fpw;
    for (size_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const size_t frame = grp * fpw + fl;
        const bool active = frame < batch;
        const cf *gin = in + frame * (size_t)n;
        cf *gout = out + frame * (size_t)n;
        __syncthreads();                                   // the previous group's last pass is done with LDS
        if (d.nfac == 0) {      // len == 1: the DFT is the identity
            if (active && lane == 0) gout[0] = cscale(gin[0], scale);
            continue;
        }
        cf *X = bufA, *Y = bufB;
        if (d.stage) {
            // single pass (every lane reads what other lanes overwrite) or a prime radix in front
            // (re-reads its inputs R times): the frame goes to LDS first
            for (int e = lane; e < n; e += tpf) {
                cf v = active ? gin[e] : mk(0.f, 0.f);
                bufB[e] = SWAP ? cswap(v) : v;
            }
            __syncthreads();
            X = bufB; Y = bufA;
        }
        for (int s = 0; s < d.nfac; s++) {
            const MixedPass &ps = d.pass[s];
            const bool fg = (s == 0) && !d.stage, tg = (s == d.nfac - 1);
#define AETH_PASS(RR) mixed_pass<RR, SWAP, PTS>(ps, n, lane, tpf, fg, tg, active, gin, gout, X, Y, twP, twN, scale)
            switch (ps.R) {
            case 2: AETH_PASS(2); break;
            case 3: AETH_PASS(3); break;
            case 4: AETH_PASS(4); break;
            case 5: AETH_PASS(5); break;
            case 7: AETH_PASS(7); break;
            case 8: AETH_PASS(8); break;
            default:
                if constexpr (BIGR) {
                    switch (ps.R) {
                    case 11: AETH_PASS(11); break;
                    case 13: AETH_PASS(13); break;
                    case 17: AETH_PASS(17); break;
                    case 19: AETH_PASS(19); break;
                    case 23: AETH_PASS(23); break;
                    default: mixed_pass_prime<SWAP>(ps, n, lane, tpf, tg, active, gout, X, Y, twP, twN, scale); break;
                    }
                } else mixed_pass_prime<SWAP>(ps, n, lane, tpf, tg, active, gout, X, Y, twP, twN, scale);
                break;
            }
#undef AETH_PASS
            __syncthreads();
            cf *t = X; X = Y; Y = t;
        }
    }
}

int launch_mixed(const aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale)
{
    const aeth_ctx *ctx = plan->ctx;
    MixedDesc d;
    d.n = (int)plan->len;
    d.nfac = (int)plan->factors.size();
    int pp = 1, off = 0;
    for (int i = 0; i < d.nfac; i++) {
        MixedPass &ps = d.pass[i];
        ps.R = plan->factors[i]; ps.p = pp; ps.m = d.n / ps.R;
        ps.pdiv = aeth::make_fastdiv((uint32_t)pp);
        ps.tw_off = off;
        if (pp > 1) off += (ps.R - 1) * pp;
        pp *= ps.R;
    }
    auto small_radix = [](int r) { return r == 2 || r == 3 || r == 4 || r == 5 || r == 7 || r == 8; };
    auto big_radix = [](int r) { return r == 11 || r == 13 || r == 17 || r == 19 || r == 23; };
    bool bigr = false;
    for (int i = 0; i < d.nfac; i++) bigr = bigr || big_radix(d.pass[i].R);
    bigr = bigr && aeth::lab_int("AETH_MIXED_BIGR", 1) != 0;
    d.stage = (d.nfac == 1 || (d.nfac > 0 && !small_radix(d.pass[0].R) && !(bigr && big_radix(d.pass[0].R)))) ? 1 : 0;
    // Lanes per frame.  These kernels are latency-bound (a chain of short passes with a barrier each), so
    // what counts is how many frames a CU has in flight, and the number of workgroups per CU is capped by
    // registers (~6).  So: as many frames per workgroup as ~36 KiB of LDS images allow (4-5 workgroups
    // per CU), each lane looping over its frame's butterflies, but never more lanes than the widest pass
    // has butterflies (N / smallest radix) and at least 8.
    int minr = d.nfac ? d.pass[0].R : 1;
    for (int i = 1; i < d.nfac; i++) if (d.pass[i].R < minr) minr = d.pass[i].R;
    const int need = d.nfac ? (d.n + minr - 1) / minr : d.n;
    // one LDS image per frame (passes exchange in place) whenever every radix has a register form
    bool all_small = d.nfac > 0;
    for (int i = 0; i < d.nfac; i++) all_small = all_small && small_radix(d.pass[i].R);
    const bool inplace = all_small && aeth::lab_int("AETH_MIXED_INPLACE", 1) != 0;
    const size_t frame_bytes = (size_t)((inplace && plan->len > 4096) ? 1 : 2) * plan->len * sizeof(cf);
    int tpf = aeth::lab_int("AETH_MIXED_TPF", 0);
    if (tpf < 1 || tpf > kMixedWG || (tpf & (tpf - 1))) {
        int widest = 1;
        while (widest < need && widest < kMixedWG) widest <<= 1;
        tpf = widest < 8 ? widest : 8;
        // the LDS bound is the hard one (a frame of 4097..8192 samples takes one workgroup and up to 128 KiB)
        while (tpf < kMixedWG && (size_t)(kMixedWG / tpf) * frame_bytes > 36 * 1024) tpf <<= 1;
    }
    while (inplace && tpf < kMixedWG && (size_t)32 * tpf < plan->len) tpf <<= 1;   // a lane holds n/tpf <= 32 points
    // the one-image form (a lane keeps up to 32 points across a barrier: 142-190 VGPRs, two workgroups per
    // CU, one more barrier per pass) pays off only where two images of the frame would leave room for a
    // single workgroup per CU (measured: N = 6000 0.78 -> 1.39 TB/s, N <= 2048 up to 45 % slower)
    const int pts = (inplace && plan->len > 4096) ? 32 : 0;
    const int fpw = kMixedWG / tpf;
    const size_t ngroups = (batch + fpw - 1) / fpw;
    size_t cap = (size_t)ctx->num_cus * 8;
    int grid = (int)(ngroups < cap ? ngroups : cap);
    if (grid < 1) grid = 1;
    size_t shmem = (size_t)fpw * (size_t)(pts ? 1 : 2) * plan->len * sizeof(cf);
    int tw_lds = 0;
    if (off > 0 && shmem + (size_t)off * sizeof(cf) <= 40 * 1024) {     // keeps >= 4 workgroups per CU
        tw_lds = off;
        shmem += (size_t)off * sizeof(cf);
    }
#define AETH_MIXED_LAUNCH(SW, IP)                                                                                        \
    if (bigr && IP == 0)                                                                                                 \
        hipLaunchKernelGGL((fft_mixed_kernel<SW, 0, true>), dim3(grid), dim3(kMixedWG), shmem, aeth::ctx_stream(ctx), (const cf *)in, \
                           (cf *)out, (const cf *)plan->tw_dev, (const cf *)plan->tw_pass_dev, d, batch, scale, tpf, tw_lds); \
    else                                                                                                                 \
    hipLaunchKernelGGL((fft_mixed_kernel<SW, IP, false>), dim3(grid), dim3(kMixedWG), shmem, aeth::ctx_stream(ctx), (const cf *)in,       \
                       (cf *)out, (const cf *)plan->tw_dev, (const cf *)plan->tw_pass_dev, d, batch, scale, tpf, tw_lds)
#define AETH_MIXED_SIGN(IP) do { if (sign > 0) AETH_MIXED_LAUNCH(true, IP); else AETH_MIXED_LAUNCH(false, IP); } while (0)
    switch (pts) {
    case 32: AETH_MIXED_SIGN(32); break;
    default: AETH_MIXED_SIGN(0); break;
    }
#undef AETH_MIXED_SIGN
#undef AETH_MIXED_LAUNCH
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

// ================================ planning ====================================
bool is_pow2(size_t n) { return n && (n & (n - 1)) == 0; }

size_t largest_prime_factor(size_t n)
{
    size_t best = 1;
    for (size_t f = 2; f * f <= n; f++)
        while (n % f == 0) { best = f; n /= f; }
    return n > 1 ? n : best;
}

// radix schedule for stockham_mixed: 8s and 4s first, then 2, 3, 5, 7, other primes
bool factorize_mixed(size_t n, std::vector<int> &fac)
{
    fac.clear();
    while (n % 8 == 0) { fac.push_back(8); n /= 8; }
    while (n % 4 == 0) { fac.push_back(4); n /= 4; }
    while (n % 2 == 0) { fac.push_back(2); n /= 2; }
    for (size_t f = 3; f <= 61 && n > 1; f += 2)
        while (n % f == 0) { fac.push_back((int)f); n /= f; }
    return n == 1 && fac.size() <= (size_t)kMaxFactors;
}

// len = n1 * n2, both factors at most 8192 and served by a single-workgroup kernel; the pair nearest the square
// root whose factors are both register-resident (power of two or in the ragged table), else the nearest pair the
// LDS mixed-radix kernel can do.
bool split_fourstep_mixed(size_t len, size_t *n1, size_t *n2)
{
    auto fast = [](size_t n) { return (is_pow2(n) && n <= 8192) || aeth::fft_ragged_supported(n); };
    // a small first factor over one register-resident transform (up to 20480 points): no transposes.  The batched
    // transform is the bulk of the three launches and the rows from 8192 points down run at 4.3-5.9 TB/s where the
    // largest ones reach 2.9-4.1, so the smallest factor that brings the rest under that length goes first
    // (40960 as 5 x 8192 instead of 2 x 20480)
    const size_t pref = (size_t)aeth::lab_int("AETH_4SM_PREF_M", 8192);
    for (int pass = 0; pass < 2; pass++)
        for (size_t r = 2; r <= 16; r++) {
            if (!aeth::fourstep_small_factor(r) || len % r) continue;
            if (pass == 0 && len / r > pref) continue;
            if (fast(len / r)) { *n1 = r; *n2 = len / r; return true; }
        }
    size_t root = 1;
    while ((root + 1) * (root + 1) <= len) root++;
    size_t fb1 = 0, fb2 = 0;
    std::vector<int> fac;
    for (size_t a = root; a >= 2; a--) {
        if (len % a) continue;
        const size_t b = len / a;
        if (b > 8192) break;
        if (fast(a) && fast(b)) { *n1 = a; *n2 = b; return true; }
        if (!fb1 && (fast(a) || factorize_mixed(a, fac)) && (fast(b) || factorize_mixed(b, fac))) { fb1 = a; fb2 = b; }
    }
    if (fb1) { *n1 = fb1; *n2 = fb2; return true; }
    return false;
}

int make_twiddles(aeth_ctx *ctx, size_t n, float2 **out_dev)
{
    std::vector<float2> h(n ? n : 1);
    for (size_t k = 0; k < n; k++) {
        double a = -2.0 * M_PI * (double)k / (double)n;
        h[k] = make_float2((float)cos(a), (float)sin(a));
    }
    AETH_HIP(hipMalloc((void **)out_dev, h.size() * sizeof(float2)));
    AETH_HIP(hipMemcpyAsync(*out_dev, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice, aeth::ctx_stream(ctx)));
    AETH_HIP(hipStreamSynchronize(aeth::ctx_stream(ctx)));
    return AETH_OK;
}

// per-pass twiddle tables of stockham_mixed: pass s (radix R, p = product of the radices before it)
// multiplies input r of butterfly k by W_{pR}^{rk}; stored [(r-1)*p + k] so adjacent lanes read adjacent
// entries.  Angles are formed in double from the exact integer r*k, like the master table.
int plan_mixed(aeth_fft *plan)
{
    std::vector<float2> h;
    size_t pp = 1;
    for (int R : plan->factors) {
        if (pp > 1) {
            const double base = -2.0 * M_PI / (double)(pp * (size_t)R);
            for (int r = 1; r < R; r++)
                for (size_t k = 0; k < pp; k++) {
                    const double a = base * (double)((size_t)r * k);
                    h.push_back(make_float2((float)cos(a), (float)sin(a)));
                }
        }
        pp *= (size_t)R;
    }
    if (h.empty()) h.push_back(make_float2(1.f, 0.f));
    AETH_HIP(hipMalloc((void **)&plan->tw_pass_dev, h.size() * sizeof(float2)));
    AETH_HIP(hipMemcpyAsync(plan->tw_pass_dev, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice, aeth::ctx_stream(plan->ctx)));
    AETH_HIP(hipStreamSynchronize(aeth::ctx_stream(plan->ctx)));
    return AETH_OK;
}

}  // namespace

namespace aeth {

int fft_ensure_tmp(aeth_fft *plan, size_t elems)
{
    if (plan->tmp_elems >= elems) return AETH_OK;
    // the plan's device, whatever the caller's current one is: every caller (tfwd/tbwd, correlate + demod, the host
    // flavours) allocates through here, and a temp on the wrong device is a fault in the kernels that follow
    aeth::DeviceGuard dev_guard(plan->ctx->device);
    if (plan->tmp_dev) {
        AETH_HIP(hipStreamSynchronize(aeth::ctx_stream(plan->ctx)));
        AETH_HIP(hipFree(plan->tmp_dev));
        plan->tmp_dev = nullptr;
        plan->tmp_elems = 0;
    }
    AETH_HIP(hipMalloc((void **)&plan->tmp_dev, elems * sizeof(float2)));
    plan->tmp_elems = elems;
    return AETH_OK;
}

// device-pointer transform of `batch` frames; in may equal out
int fft_run(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale)
{
    if (batch == 0 || plan->len == 0) return AETH_OK;
    aeth::DeviceGuard dev_guard(plan->ctx->device);
    switch (plan->algo) {
    case FFT_ALGO_POW2:  return dispatch_pow2(plan, in, out, batch, sign, scale);
    case FFT_ALGO_MIXED: return launch_mixed(plan, in, out, batch, sign, scale);
    case FFT_ALGO_RAGGED: return fft_run_ragged(plan, in, out, batch, sign, scale);
    case FFT_ALGO_FOURSTEP: return fft_run_fourstep(plan, in, out, batch, sign, scale);
    case FFT_ALGO_FOURSTEP_MIXED: return fft_run_fourstep_mixed(plan, in, out, batch, sign, scale);
    case FFT_ALGO_BLUESTEIN: return fft_run_bluestein(plan, in, out, batch, sign, scale);
    default: return set_error(AETH_E_UNSUPPORTED, "no kernel path for length %zu", plan->len);
    }
}

}  // namespace aeth

extern "C" {

int aeth_fft_create(aeth_ctx *ctx, size_t len, size_t max_batch, aeth_fft **out)
{
    AETH_REQUIRE(ctx && out, AETH_E_ARG, "null argument");
    *out = nullptr;
    AETH_REQUIRE(len >= 1, AETH_E_ARG, "FFT length must be >= 1");
    AETH_REQUIRE(len <= ((size_t)1 << 24), AETH_E_UNSUPPORTED, "FFT length %zu > 2^24", len);
    aeth::DeviceGuard g(ctx->device);
    aeth_fft *p = new (std::nothrow) aeth_fft();
    AETH_REQUIRE(p, AETH_E_NOMEM, "out of host memory");
    p->ctx = ctx;
    p->len = len;
    if (max_batch < 1) max_batch = 1;
    int rc = AETH_OK;
    if (len == 1) {
        p->algo = aeth::FFT_ALGO_MIXED;      // zero passes: copy + scale
        p->factors.clear();
        p->algo_name = "identity";
    } else if (is_pow2(len) && (len <= 4096 || (len == 8192 && (!aeth::fft_ragged_supported(len) || aeth::lab_int("AETH_FFT_NORAGGED", 0))))) {
        // 8192 itself has a row in the ragged table (256 lanes x 32 points, radices 32 32 8, unpadded image: 102 us
        // per 32 Mi samples against 124 us for the 512-lane x 16-point kernel here, which spills at two waves per SIMD)
        p->algo = aeth::FFT_ALGO_POW2;
        p->algo_name = "stockham_pow2";
    } else if (aeth::fft_ragged_supported(len) && !aeth::lab_int("AETH_FFT_NORAGGED", 0)) {
        p->algo = aeth::FFT_ALGO_RAGGED;
        p->algo_name = "stockham_mixed_ragged";
    } else if (2 * len - 1 <= 4096 && largest_prime_factor(len) >= (size_t)aeth::lab_int("AETH_FFT_PRIME_BLU_MIN", 17) && aeth::lab_int("AETH_FFT_PRIME_BLU", 1)) {
        // a prime factor from 17 up and a chirp-z convolution that fits the one-launch kernel (M <= 4096): 1.4-2.6 TB/s
        // against 0.05-0.7 TB/s through the generic O(r^2) pass of the LDS kernel and 0.4-1.0 TB/s through
        // fourstep_mixed (tools/prime_route.py: 17 ... 2047, x1.1 ... x50)
        p->algo = aeth::FFT_ALGO_BLUESTEIN;
        p->algo_name = "bluestein";
        rc = aeth::fft_plan_bluestein(p);
    } else if (len <= 8192 && factorize_mixed(len, p->factors)) {      // two LDS images of the frame: 128 KiB at most
        p->algo = aeth::FFT_ALGO_MIXED;
        p->algo_name = "stockham_mixed";
    } else if (is_pow2(len)) {
        p->algo = aeth::FFT_ALGO_FOURSTEP;
        p->algo_name = "fourstep_pow2";
        rc = aeth::fft_plan_fourstep(p);
    } else if (!aeth::lab_int("AETH_FFT_NO4SMIXED", 0) && split_fourstep_mixed(len, &p->n1, &p->n2)) {
        p->algo = aeth::FFT_ALGO_FOURSTEP_MIXED;
        p->algo_name = "fourstep_mixed";
        rc = aeth::fft_plan_fourstep_mixed(p);
    } else {
        p->algo = aeth::FFT_ALGO_BLUESTEIN;
        p->algo_name = "bluestein";
        rc = aeth::fft_plan_bluestein(p);
    }
    if (rc == AETH_OK) rc = make_twiddles(ctx, len, &p->tw_dev);
    if (rc == AETH_OK && p->algo == aeth::FFT_ALGO_POW2) rc = plan_pow2(p);
    if (rc == AETH_OK && p->algo == aeth::FFT_ALGO_MIXED) rc = plan_mixed(p);
    if (rc == AETH_OK && p->algo == aeth::FFT_ALGO_RAGGED) rc = aeth::fft_plan_ragged(p);
    // Cfft.tmp (fft.rs:141,155) is allocated on first use: only tfwd/tbwd and the host-slice flavours lend or stage
    // through it, and the sub-plans of the four-step / chirp-z paths and the FIR's plan never do (a 2^24-point child
    // would otherwise hold 256 MiB of device and 256 MiB of pinned host memory for nothing)
    (void)max_batch;
    if (rc != AETH_OK) { aeth_fft_destroy(p); return rc; }
    *out = p;
    return AETH_OK;
}

int aeth_fft_destroy(aeth_fft *p)
{
    if (!p) return AETH_OK;
    aeth::DeviceGuard g(p->ctx->device);
    (void)hipStreamSynchronize(aeth::ctx_stream(p->ctx));
    aeth::fft_plan_release_children(p);
    if (p->tw_dev) (void)hipFree(p->tw_dev);
    if (p->tw_lane_dev) (void)hipFree(p->tw_lane_dev);
    if (p->tw_pass_dev) (void)hipFree(p->tw_pass_dev);
    if (p->tmp_dev) (void)hipFree(p->tmp_dev);
    if (p->tmp_host) (void)hipHostFree(p->tmp_host);
    delete p;
    return AETH_OK;
}

}  // extern "C"

/* VecOps::vec_fft / vec_ifft (src/vecops.rs:184-196): `Cfft::with_len(self.len())` then ifwd / ibwd -- the reference plans
 * on EVERY call ("planning + allocation every call").  Here the context keeps the plans these one-shot calls
 * have built (the eight most recently used lengths; a plan is its twiddle tables and scratch), so the second vec_fft of a
 * length costs what vec_rfft with a reused plan costs.  aeth_ctx_trim / aeth_ctx_destroy free them. */
namespace {
constexpr size_t kFftCacheMax = 8, kFftCachePoints = (size_t)1 << 24;
struct FftCache { std::vector<aeth_fft *> plans; };

int fft_cache_get(aeth_ctx *ctx, size_t len, aeth_fft **out)
{
    if (!ctx->fft_cache) {
        ctx->fft_cache = new (std::nothrow) FftCache();
        AETH_REQUIRE(ctx->fft_cache, AETH_E_NOMEM, "out of host memory");
    }
    auto &v = static_cast<FftCache *>(ctx->fft_cache)->plans;
    for (size_t i = 0; i < v.size(); i++)
        if (v[i]->len == len) { aeth_fft *p = v[i]; v.erase(v.begin() + (long)i); v.insert(v.begin(), p); *out = p; return AETH_OK; }
    aeth_fft *p = nullptr;
    int rc = aeth_fft_create(ctx, len, 1, &p); if (rc) return rc;
    v.insert(v.begin(), p);
    // at most eight plans and 2^24 points of them in total (a plan's tables and scratch are a few times its length): the least
    // recently used ones go first, the plan just built always stays
    size_t total = 0;
    for (aeth_fft *q : v) total += q->len;
    while (v.size() > 1 && (v.size() > kFftCacheMax || total > kFftCachePoints)) {
        total -= v.back()->len;
        (void)aeth_fft_destroy(v.back()); v.pop_back();
    }
    *out = p;
    return AETH_OK;
}
}  // namespace

namespace aeth {
void fft_cache_release(aeth_ctx *ctx)
{
    if (!ctx->fft_cache) return;
    FftCache *c = static_cast<FftCache *>(ctx->fft_cache);
    for (aeth_fft *p : c->plans) (void)aeth_fft_destroy(p);
    delete c;
    ctx->fft_cache = nullptr;
}
}  // namespace aeth

extern "C" {

int aeth_vec_fft(aeth_ctx *ctx, aeth_cf32 *x_dev, size_t n, int sign, int kind, float x)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    AETH_REQUIRE(n >= 1, AETH_E_ARG, "FFT length must be >= 1");
    aeth_fft *p = nullptr;
    int rc = fft_cache_get(ctx, n, &p); if (rc) return rc;
    return aeth_fft_exec(p, x_dev, n, x_dev, 1, sign, kind, x);
}

int aeth_host_vec_fft(aeth_ctx *ctx, aeth_cf32 *x_host, size_t n, int sign, int kind, float x)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    AETH_REQUIRE(n >= 1, AETH_E_ARG, "FFT length must be >= 1");
    aeth_fft *p = nullptr;
    int rc = fft_cache_get(ctx, n, &p); if (rc) return rc;
    return aeth_fft_exec_host(p, x_host, n, x_host, n, sign, kind, x);
}

size_t aeth_fft_len(const aeth_fft *p) { return p ? p->len : 0; }
const char *aeth_fft_algorithm(const aeth_fft *p) { return p ? p->algo_name : ""; }

// device temp of >= elems and (host = true) the pinned 2*len mirror tfwd/tbwd lend out
static int ensure_temps(aeth_fft *p, size_t elems, bool host)
{
    aeth::DeviceGuard g(p->ctx->device);
    int rc = aeth::fft_ensure_tmp(p, elems); if (rc) return rc;
    if (host && !p->tmp_host) {
        hipError_t e = hipHostMalloc((void **)&p->tmp_host, 2 * p->len * sizeof(float2), hipHostMallocDefault);
        if (e != hipSuccess) return aeth::hip_fail(e, "hipHostMalloc");
    }
    return AETH_OK;
}

static int check_exec(const aeth_fft *p, int sign, int kind)
{
    AETH_REQUIRE(p, AETH_E_ARG, "plan is null");
    AETH_REQUIRE(sign == AETH_SIGN_REF_FWD || sign == AETH_SIGN_REF_BWD, AETH_E_ARG, "sign must be +1 or -1");
    AETH_REQUIRE(kind >= AETH_SCALE_NONE && kind <= AETH_SCALE_X, AETH_E_ARG, "bad scale kind %d", kind);
    return AETH_OK;
}

int aeth_fft_exec(aeth_fft *p, const aeth_cf32 *in, size_t n_in, aeth_cf32 *out, size_t batch, int sign,
                  int kind, float x)
{
    int rc = check_exec(p, sign, kind); if (rc) return rc;
    AETH_REQUIRE(n_in == batch * p->len, AETH_E_LEN, AETH_MSG_FFT_LEN);     /* fft.rs:163-167 */
    if (batch == 0) return AETH_OK;
    AETH_REQUIRE(in && out, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(aeth::aligned8(in) && aeth::aligned8(out), AETH_E_ALIGN, "pointer not 8-byte aligned");
    const float s = aeth_scale_factor(kind, p->len, x);                     /* fft.rs:22-37, n = frame length */
    return aeth::fft_run(p, (const float2 *)in, (float2 *)out, batch, sign, s);
}

/* c.vec_rfft(&mut fft, s).vec_mirror() per frame (src/util/plot.rs:59-61) in one call */
int aeth_fft_exec_mirrored(aeth_fft *p, const aeth_cf32 *in, size_t n_in, aeth_cf32 *out, size_t batch, int sign,
                           int kind, float x)
{
    int rc = check_exec(p, sign, kind); if (rc) return rc;
    AETH_REQUIRE(n_in == batch * p->len, AETH_E_LEN, AETH_MSG_FFT_LEN);     /* fft.rs:163-167 */
    if (batch == 0) return AETH_OK;
    AETH_REQUIRE(in && out, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(aeth::aligned8(in) && aeth::aligned8(out), AETH_E_ALIGN, "pointer not 8-byte aligned");
    const float s = aeth_scale_factor(kind, p->len, x);
    if (p->algo == aeth::FFT_ALGO_POW2 && p->len >= 2) {
        // register-resident transforms: the swap of the halves is folded into the store addresses
        aeth::DeviceGuard dev_guard(p->ctx->device);
        return dispatch_pow2(p, (const float2 *)in, (float2 *)out, batch, sign, s, 1);
    }
    rc = aeth::fft_run(p, (const float2 *)in, (float2 *)out, batch, sign, s); if (rc) return rc;
    return aeth_vec_mirror_frames(p->ctx, out, p->len, batch);              /* vecops.rs:157-161 per frame */
}

/* per frame: c.vec_rfft / vec_rifft (sign, scale) then sampling::interpolate(&c, &mut dst, n_between) (sampling.rs:7-24);
 * `in` is left as it was */
int aeth_fft_exec_interpolate(aeth_fft *p, const aeth_cf32 *in, size_t n_in, size_t batch, int sign, int kind, float x,
                              aeth_cf32 *dst, size_t dst_cap, size_t n_between, int compat_im, size_t *n_written)
{
    if (n_written) *n_written = 0;
    int rc = check_exec(p, sign, kind); if (rc) return rc;
    AETH_REQUIRE(n_in == batch * p->len, AETH_E_LEN, AETH_MSG_FFT_LEN);     /* fft.rs:163-167 */
    AETH_REQUIRE(p->len > 0, AETH_E_LEN, "interpolate on an empty src (the reference panics: sampling.rs:23)");
    if (batch == 0) return AETH_OK;
    AETH_REQUIRE(in && dst, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(aeth::aligned8(in) && aeth::aligned8(dst), AETH_E_ALIGN, "pointer not 8-byte aligned");
    AETH_REQUIRE(n_between < 0x7fffffffu, AETH_E_ARG, "n_between too large");
    const size_t Lo = p->len + (p->len - 1) * n_between;
    AETH_REQUIRE(dst_cap >= Lo * batch, AETH_E_LEN, "dst capacity %zu < %zu", dst_cap, Lo * batch);
    const float s = aeth_scale_factor(kind, p->len, x);
    // the two steps through the plan's temp (Cfft.tmp, fft.rs:141); a fused last pass was measured slower (aeth_fft_big.hip)
    rc = ensure_temps(p, n_in, false); if (rc) return rc;
    rc = aeth::fft_run(p, (const float2 *)in, p->tmp_dev, batch, sign, s); if (rc) return rc;
    return aeth_interpolate_frames(p->ctx, (const aeth_cf32 *)p->tmp_dev, p->len, batch, dst, dst_cap, n_between, compat_im, n_written);
}

int aeth_fft_exec_tmp(aeth_fft *p, const aeth_cf32 *in, size_t n_in, size_t batch, int sign, int kind, float x,
                      const aeth_cf32 **view)
{
    int rc = check_exec(p, sign, kind); if (rc) return rc;
    AETH_REQUIRE(view, AETH_E_ARG, "view is null");
    *view = nullptr;
    AETH_REQUIRE(n_in == batch * p->len, AETH_E_LEN, AETH_MSG_FFT_LEN);     /* fft.rs:207-211 */
    if (batch == 0) return AETH_OK;
    AETH_REQUIRE(in, AETH_E_ARG, "null pointer");
    rc = aeth::fft_ensure_tmp(p, 2 * n_in); if (rc) return rc;
    float2 *dst = p->tmp_dev + n_in;                                            /* tmp[len..], fft.rs:213 */
    const float s = aeth_scale_factor(kind, p->len, x);
    rc = aeth::fft_run(p, (const float2 *)in, dst, batch, sign, s); if (rc) return rc;
    *view = (const aeth_cf32 *)dst;
    return AETH_OK;
}

// one frame per call on host memory: frames up to 8192 points (one launch, coalesced rows) read and write the pinned bounce
// buffers directly; longer ones (several launches, strided first pass) go through device memory as before
static constexpr size_t kFftZeroCopyMax = (size_t)64 << 10;

int aeth_fft_exec_host(aeth_fft *p, const aeth_cf32 *in, size_t n_in, aeth_cf32 *out, size_t n_out, int sign,
                       int kind, float x)
{
    int rc = check_exec(p, sign, kind); if (rc) return rc;
    AETH_REQUIRE(n_in == p->len, AETH_E_LEN, AETH_MSG_FFT_LEN);
    AETH_REQUIRE(n_out == p->len, AETH_E_LEN, "Output and FFT must be the same length");
    AETH_REQUIRE(in && out, AETH_E_ARG, "null pointer");
    aeth::DeviceGuard dev_guard(p->ctx->device);
    const size_t bytes = p->len * sizeof(float2);
    const float s = aeth_scale_factor(kind, p->len, x);
    if (bytes <= kFftZeroCopyMax) {
        // one frame per call is latency-bound: the kernel reads the frame from and writes the spectrum to pinned host
        // memory (aeth::HostIO), one launch and one wait instead of H2D + launch + D2H + wait
        aeth::HostIO io;
        rc = io.open(p->ctx, bytes, bytes); if (rc) return rc;
        rc = io.put(0, in, bytes); if (rc) return rc;                               /* tmp[..len] <- input, fft.rs:168 */
        rc = aeth::fft_run(p, (const float2 *)io.buf[0], (float2 *)io.buf[1], 1, sign, s); if (rc) return rc;
        return io.get(out, 1, bytes);
    }
    rc = ensure_temps(p, 2 * p->len, false); if (rc) return rc;
    hipStream_t st = aeth::ctx_stream(p->ctx);
    AETH_HIP(hipMemcpyAsync(p->tmp_dev, in, bytes, hipMemcpyHostToDevice, st));   /* tmp[..len] <- input, fft.rs:168 */
    rc = aeth::fft_run(p, p->tmp_dev, p->tmp_dev + p->len, 1, sign, s); if (rc) return rc;
    AETH_HIP(hipMemcpyAsync(out, p->tmp_dev + p->len, bytes, hipMemcpyDeviceToHost, st));
    AETH_HIP(hipStreamSynchronize(st));
    return AETH_OK;
}

int aeth_fft_exec_tmp_host(aeth_fft *p, const aeth_cf32 *in, size_t n_in, int sign, int kind, float x,
                           const aeth_cf32 **view)
{
    int rc = check_exec(p, sign, kind); if (rc) return rc;
    AETH_REQUIRE(view, AETH_E_ARG, "view is null");
    *view = nullptr;
    AETH_REQUIRE(n_in == p->len, AETH_E_LEN, AETH_MSG_FFT_LEN);
    AETH_REQUIRE(in, AETH_E_ARG, "null pointer");
    rc = ensure_temps(p, 2 * p->len, true); if (rc) return rc;
    aeth::DeviceGuard dev_guard(p->ctx->device);
    hipStream_t st = aeth::ctx_stream(p->ctx);
    const size_t bytes = p->len * sizeof(float2);
    const float s = aeth_scale_factor(kind, p->len, x);
    if (bytes <= kFftZeroCopyMax) {
        // tmp_host is pinned: the kernel writes the lent half of it directly
        aeth::HostIO io;
        rc = io.open(p->ctx, bytes, 0); if (rc) return rc;
        rc = io.put(0, in, bytes); if (rc) return rc;
        rc = aeth::fft_run(p, (const float2 *)io.buf[0], p->tmp_host + p->len, 1, sign, s); if (rc) return rc;
        rc = io.wait(); if (rc) return rc;
        *view = (const aeth_cf32 *)(p->tmp_host + p->len);
        return AETH_OK;
    }
    AETH_HIP(hipMemcpyAsync(p->tmp_dev, in, bytes, hipMemcpyHostToDevice, st));
    rc = aeth::fft_run(p, p->tmp_dev, p->tmp_dev + p->len, 1, sign, s); if (rc) return rc;
    AETH_HIP(hipMemcpyAsync(p->tmp_host + p->len, p->tmp_dev + p->len, bytes, hipMemcpyDeviceToHost, st));
    AETH_HIP(hipStreamSynchronize(st));
    *view = (const aeth_cf32 *)(p->tmp_host + p->len);
    return AETH_OK;
}

}  // extern "C"
