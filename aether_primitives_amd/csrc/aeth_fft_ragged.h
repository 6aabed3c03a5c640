// aeth_fft_ragged.h -- register-resident Stockham transform whose passes need not share a lane shape
// ("stockham_mixed_ragged", behind trait Fft, reference src/fft.rs:48-77).
//
// aeth_fft_core.h keeps P = N/T points in every lane through all passes, so each radix has to divide P; lengths
// with the factors 2, 3 and 5 all present (120, 600, 1200, 3600 ... the DFT-precoding sizes) then need 30-point
// lanes, and with the twiddles of every pass resident that is 270-410 registers: one wave per SIMD.  Here a pass
// of radix R is simply M = N/R butterflies dealt out to the T lanes of the frame, ceil(M/T) each, the last round
// predicated when T does not divide M.  A lane holds one butterfly (R points) at a time plus the twiddles of its
// own butterflies, the radices only have to multiply to N, and some lanes idle in some passes -- which costs
// nothing on a transform that waits for HBM anyway.
//
//   pass s, butterfly i = tid + b*T (< M):  u[r] = element i + r*M of the current array (global memory in the
//   first pass, the LDS image after that), u[r] *= W_{pR}^{r*(i mod p)}, radix-R DFT, result r goes to element
//   (i - i mod p)*R + (i mod p) + r*p (LDS image), or i + r*M of the output frame in the last pass.
//
// Frames of few lanes are copied between HBM and LDS in whole-workgroup coalesced rows first (STAGED), as in
// fft_pow2_kernel.
#pragma once

#include "aeth_fft_core.h"
#include "aeth_internal.h"

namespace aeth {
namespace fftk {

#ifndef AETH_RAGGED_DB_LIMIT
#define AETH_RAGGED_DB_LIMIT (40 * 1024)   /* two exchange images up to this many bytes per workgroup */
#endif

// PAD_: one pad slot per 2^PAD_ elements of the exchange image (4 as in aeth_fft_core.h; 0: none -- the largest
// frames trade bank conflicts for a third workgroup per CU)
template <int N_, int T_, int WG_, int PAD_, int R0_, int R1_ = 1, int R2_ = 1, int R3_ = 1, int STAGE_ = -1>
struct RCfg {
    static constexpr int N = N_, T = T_, WG = WG_;
    static constexpr int F = WG / T;                 // frames per workgroup
    static constexpr int IDLE = WG - F * T;          // lanes beyond the last whole frame
    static constexpr int NPASS = (R1_ == 1) ? 1 : (R2_ == 1) ? 2 : (R3_ == 1) ? 3 : 4;
    static constexpr int radix(int s) { return s == 0 ? R0_ : s == 1 ? R1_ : s == 2 ? R2_ : R3_; }
    static constexpr int pbefore(int s)
    {
        int p = 1;
        for (int i = 0; i < s; i++) p *= radix(i);
        return p;
    }
    static constexpr int bflies(int s) { return N / radix(s); }                   // M
    static constexpr int perlane(int s) { return (bflies(s) + T - 1) / T; }        // B
    static constexpr int twcount(int s) { return s == 0 ? 0 : perlane(s) * (radix(s) - 1); }
    static constexpr int twoff(int s)
    {
        int o = 0;
        for (int i = 0; i < s; i++) o += twcount(i);
        return o;
    }
    static constexpr int TW = twoff(NPASS) > 0 ? twoff(NPASS) : 1;
    static constexpr int FRAME = PAD_ > 0 ? N + (N >> PAD_) : N;          // padded LDS image of one frame
    static __device__ __forceinline__ int at(int e) { return PAD_ > 0 ? e + (e >> PAD_) : e; }
    static constexpr int IMAGE = (NPASS > 1) ? FRAME * F : 0;
    static constexpr bool DB = (NPASS > 1) && (2 * IMAGE * 8 <= AETH_RAGGED_DB_LIMIT);
    // frames whose first or last pass touches memory in short rows go through an LDS copy of the group
    static constexpr bool STAGED = STAGE_ >= 0 ? (STAGE_ != 0)
                                               : ((bflies(0) < 48 || bflies(NPASS - 1) < 48) && F > 1);
    static constexpr int IO = STAGED ? F * (N + 1) : 0;
    static constexpr int LDS_TOTAL = (DB ? 2 * IMAGE : IMAGE) + IO > 0 ? (DB ? 2 * IMAGE : IMAGE) + IO : 1;
    static_assert(R0_ * R1_ * R2_ * R3_ == N_, "radices must multiply to N");
    static_assert(F >= 1, "a frame needs T lanes");
    static_assert(WG % 64 == 0, "whole waves");
};

// per-lane twiddle table of a plan: row (pass, butterfly round b, power r) holds T entries
template <class C>
__global__ void build_ragged_twiddles(const cf *__restrict__ twN, cf *__restrict__ twL)
{
    const int tid = threadIdx.x;
    if (tid >= C::T) return;
    for (int s = 1; s < C::NPASS; s++) {
        const int R = C::radix(s), p = C::pbefore(s), step = C::N / (p * R);
        for (int b = 0; b < C::perlane(s); b++) {
            const int k = (tid + b * C::T) % p;
            for (int r = 1; r < R; r++) twL[(C::twoff(s) + b * (R - 1) + (r - 1)) * C::T + tid] = twN[r * k * step];
        }
    }
}

template <class C, int PASS, int S, bool NT>
__device__ __forceinline__ void ragged_pass(const cf (&tw)[C::TW], const cf *rd, cf *wr, bool live, int tid, cf ss)
{
    constexpr int R = C::radix(PASS), M = C::bflies(PASS), B = C::perlane(PASS), p = C::pbefore(PASS);
    constexpr bool first = PASS == 0, last = PASS == C::NPASS - 1;
    constexpr bool from_mem = first && !C::STAGED, to_mem = last && !C::STAGED;
    cf x[B][R];
    // every load of the pass before the first butterfly: one memory round trip per pass
#pragma unroll
    for (int b = 0; b < B; b++) {
        const int i = tid + b * C::T;
        const bool act = live && ((b + 1) * C::T <= M || i < M);
#pragma unroll
        for (int r = 0; r < R; r++) {
            if constexpr (from_mem) x[b][r] = act ? aeth::nt_load<NT>(rd + i + r * M) : mk(0.f, 0.f);
            else if constexpr (first) x[b][r] = act ? rd[i + r * M] : mk(0.f, 0.f);          // staged copy: unpadded
            else x[b][r] = act ? rd[C::at(i + r * M)] : mk(0.f, 0.f);
        }
    }
    // one exchange image: everybody has read it before anybody overwrites it
    if constexpr (!C::DB && !last) __syncthreads();
#pragma unroll
    for (int b = 0; b < B; b++) {
        const int i = tid + b * C::T;
        const bool act = live && ((b + 1) * C::T <= M || i < M);
        cf(&u)[R] = x[b];
        if constexpr (p > 1) {
#pragma unroll
            for (int r = 1; r < R; r++) u[r] = ctw<S>(u[r], tw[C::twoff(PASS) + b * (R - 1) + (r - 1)]);
        }
        Bfly<R, S>::run(u);
        if (act) {
            if constexpr (to_mem) {
#pragma unroll
                for (int r = 0; r < R; r++) aeth::nt_store<NT>(wr + i + r * M, cscale_k(u[r], ss));
            } else if constexpr (last) {
#pragma unroll
                for (int r = 0; r < R; r++) wr[i + r * M] = cscale_k(u[r], ss);
            } else {
                const int k = i % p;
                const int j = (i - k) * R + k;
#pragma unroll
                for (int r = 0; r < R; r++) wr[C::at(j + r * p)] = u[r];
            }
        }
    }
}

// NT: frames are streamed with the non-temporal hint (batches beyond the cache; aeth_internal.h)
// Workgroups of 256 lanes and more are held to two waves per SIMD (256 registers): past that one workgroup has
// the CU to itself and every barrier stalls it (N = 6000: 151 -> 123 us).  Smaller workgroups keep their registers:
// several of them share a CU at one wave per SIMD, and spilling costs more than that (N = 480: 100 -> 148 us).
template <class C> constexpr int ragged_min_waves() { return C::WG >= 256 ? 2 : 1; }

template <class C, int S, bool NT>
__global__ __launch_bounds__(C::WG, ragged_min_waves<C>()) void fft_ragged_kernel(const cf *in, cf *out, const cf *__restrict__ twL,
                                                            size_t batch, float scale)
{
    __shared__ cf lds_all[C::LDS_TOTAL];
    cf *const lds_io = lds_all + (C::DB ? 2 * C::IMAGE : C::IMAGE);
    const int fl = (int)(threadIdx.x / C::T);
    const int tid = (int)threadIdx.x - fl * C::T;
    const bool lane_ok = C::IDLE == 0 || fl < C::F;
    const int fr = lane_ok ? fl : 0;

    cf tw[C::TW];
#pragma unroll
    for (int q = 0; q < C::TW; q++) tw[q] = (C::NPASS > 1) ? twL[q * C::T + tid] : mk(1.f, 0.f);

    const cf ss = mk(scale, scale);
    const size_t ngroups = (batch + C::F - 1) / C::F;
    unsigned par = 0;
    for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const size_t frame = g * C::F + fr;
        const bool live = lane_ok && frame < batch;
        const size_t base = g * C::F * (size_t)C::N;
        const size_t left = batch * (size_t)C::N - base;
        const int have = left < (size_t)(C::F * C::N) ? (int)left : C::F * C::N;
        const cf *src = in + frame * C::N;
        cf *dst = out + frame * C::N;
        if constexpr (C::STAGED) {
            constexpr int ITER = (C::F * C::N + C::WG - 1) / C::WG;
            cf stage[ITER];
#pragma unroll
            for (int q = 0; q < ITER; q++) {
                const int e = threadIdx.x + q * C::WG;
                stage[q] = e < have ? aeth::nt_load<NT>(in + base + e) : mk(0.f, 0.f);
            }
            __syncthreads();                                    // the previous group has been copied out
#pragma unroll
            for (int q = 0; q < ITER; q++) {
                const int e = threadIdx.x + q * C::WG;
                if (e < C::F * C::N) lds_io[e + e / C::N] = stage[q];
            }
            __syncthreads();
            src = lds_io + fr * (C::N + 1);
            dst = lds_io + fr * (C::N + 1);
        }
        // exchange e of this frame goes through image (par + e) & 1 when there are two; one barrier per exchange
        // then suffices (an image is rewritten only after the barrier of the exchange in between)
        auto image = [&](int e) -> cf * {
            return lds_all + fr * C::FRAME + ((C::DB && ((par + e) & 1)) ? C::IMAGE : 0);
        };
        if constexpr (C::NPASS == 1) {
            ragged_pass<C, 0, S, NT>(tw, src, dst, live, tid, ss);
        } else {
            ragged_pass<C, 0, S, NT>(tw, src, image(0), live, tid, ss);
            __syncthreads();
            if constexpr (C::NPASS > 2) {
                ragged_pass<C, 1, S, NT>(tw, image(0), image(1), live, tid, ss);
                __syncthreads();
            }
            if constexpr (C::NPASS > 3) {
                ragged_pass<C, 2, S, NT>(tw, image(1), image(2), live, tid, ss);
                __syncthreads();
            }
            ragged_pass<C, C::NPASS - 1, S, NT>(tw, image(C::NPASS - 2), dst, live, tid, ss);
            par = (par + C::NPASS - 1) & 1;
        }
        if constexpr (C::STAGED) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < (C::F * C::N + C::WG - 1) / C::WG; q++) {
                const int e = threadIdx.x + q * C::WG;
                if (e < have) aeth::nt_store<NT>(out + base + e, lds_io[e + e / C::N]);
            }
        }
    }
}

}  // namespace fftk
}  // namespace aeth
