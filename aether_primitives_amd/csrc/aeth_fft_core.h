// aeth_fft_core.h -- device-side building blocks of the power-of-two FFT:
// register butterflies (radix 2/4/8/16) and the Stockham autosort passes that
// exchange data through LDS.  Shared by aeth_fft.hip (batched transforms behind
// trait Fft, reference src/fft.rs:48-77) and aeth_fir.hip (fused FFT * H * IFFT).
//
// Shape of one transform of N points by T = N/P lanes, P points per lane:
//   register slot w[m]  <->  element  tid + m*T  of the current N-array
//   pass s (radix R, B = P/R butterflies per lane, p = product of earlier radices):
//     butterfly b works on i = tid + b*T:  u[r] = w[b + r*B]      (element i + r*N/R)
//     u[r] *= W_{pR}^{r*(i mod p)}; radix-R DFT in registers;
//     result r goes to element (i - i mod p)*R + (i mod p) + r*p  of the next array
//   so every pass READS at stride N/R (lane-contiguous: coalesced from HBM,
//   conflict-free from LDS) and only the LDS WRITE is scattered; the last pass
//   leaves its results in the w[] slots of the natural-order output, ready for a
//   coalesced store -- or for the first pass of the next transform (FIR) with no
//   exchange in between.
// Only the -j exponent is coded.  The +j transform (the reference's `fwd`,
// src/fft.rs:148) swaps re/im on load and on store: swap(DFT-(swap x)) = DFT+(x),
// which is a register renaming at compile time.
#pragma once

#include <hip/hip_runtime.h>

namespace aeth {
namespace fftk {

using cf = float2;

__device__ __forceinline__ cf mk(float a, float b) { return make_float2(a, b); }
__device__ __forceinline__ cf cadd(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cf cmul(cf a, cf w) { return mk(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }
__device__ __forceinline__ cf cscale(cf a, float s) { return mk(a.x * s, a.y * s); }
__device__ __forceinline__ cf mul_mj(cf a) { return mk(a.y, -a.x); }   // a * (-j)
__device__ __forceinline__ cf cswap(cf a) { return mk(a.y, a.x); }

constexpr float kSqrtHalf = 0.70710678118654752440f;
constexpr float kCosPi8 = 0.92387953251128675613f;
constexpr float kSinPi8 = 0.38268343236508977173f;

// a * W8^1 = a * (1 - j)/sqrt2 ; a * W8^3 = a * (-1 - j)/sqrt2
__device__ __forceinline__ cf mul_w8_1(cf a) { return mk((a.x + a.y) * kSqrtHalf, (a.y - a.x) * kSqrtHalf); }
__device__ __forceinline__ cf mul_w8_3(cf a) { return mk((a.y - a.x) * kSqrtHalf, -(a.x + a.y) * kSqrtHalf); }

// a * W16^M, W16 = exp(-2 pi i / 16)
template <int M>
__device__ __forceinline__ cf mul_w16(cf a)
{
    if constexpr (M == 0) return a;
    else if constexpr (M == 1) return cmul(a, mk(kCosPi8, -kSinPi8));
    else if constexpr (M == 2) return mul_w8_1(a);
    else if constexpr (M == 3) return cmul(a, mk(kSinPi8, -kCosPi8));
    else if constexpr (M == 4) return mul_mj(a);
    else if constexpr (M == 6) return mul_w8_3(a);
    else if constexpr (M == 9) return cmul(a, mk(-kCosPi8, kSinPi8));
    else { static_assert(M < 0, "unsupported W16 power"); return a; }
}

__device__ __forceinline__ void dft4(cf a, cf b, cf c, cf d, cf &y0, cf &y1, cf &y2, cf &y3)
{
    cf s0 = cadd(a, c), s1 = csub(a, c), s2 = cadd(b, d), s3 = mul_mj(csub(b, d));
    y0 = cadd(s0, s2);
    y2 = csub(s0, s2);
    y1 = cadd(s1, s3);
    y3 = csub(s1, s3);
}

// in-register radix-R DFT, natural-order in, natural-order out
template <int R> struct Bfly;

template <> struct Bfly<1> { static __device__ __forceinline__ void run(cf (&)[1]) {} };

template <> struct Bfly<2> {
    static __device__ __forceinline__ void run(cf (&u)[2])
    {
        cf a = u[0], b = u[1];
        u[0] = cadd(a, b);
        u[1] = csub(a, b);
    }
};

template <> struct Bfly<4> {
    static __device__ __forceinline__ void run(cf (&u)[4]) { dft4(u[0], u[1], u[2], u[3], u[0], u[1], u[2], u[3]); }
};

template <> struct Bfly<8> {
    static __device__ __forceinline__ void run(cf (&u)[8])
    {
        cf e0, e1, e2, e3, o0, o1, o2, o3;
        dft4(u[0], u[2], u[4], u[6], e0, e1, e2, e3);
        dft4(u[1], u[3], u[5], u[7], o0, o1, o2, o3);
        o1 = mul_w8_1(o1);
        o2 = mul_mj(o2);
        o3 = mul_w8_3(o3);
        u[0] = cadd(e0, o0); u[4] = csub(e0, o0);
        u[1] = cadd(e1, o1); u[5] = csub(e1, o1);
        u[2] = cadd(e2, o2); u[6] = csub(e2, o2);
        u[3] = cadd(e3, o3); u[7] = csub(e3, o3);
    }
};

template <> struct Bfly<16> {
    static __device__ __forceinline__ void run(cf (&u)[16])
    {
        cf t0[4], t1[4], t2[4], t3[4];
        dft4(u[0], u[4], u[8],  u[12], t0[0], t0[1], t0[2], t0[3]);
        dft4(u[1], u[5], u[9],  u[13], t1[0], t1[1], t1[2], t1[3]);
        dft4(u[2], u[6], u[10], u[14], t2[0], t2[1], t2[2], t2[3]);
        dft4(u[3], u[7], u[11], u[15], t3[0], t3[1], t3[2], t3[3]);
        t1[1] = mul_w16<1>(t1[1]); t1[2] = mul_w16<2>(t1[2]); t1[3] = mul_w16<3>(t1[3]);
        t2[1] = mul_w16<2>(t2[1]); t2[2] = mul_w16<4>(t2[2]); t2[3] = mul_w16<6>(t2[3]);
        t3[1] = mul_w16<3>(t3[1]); t3[2] = mul_w16<6>(t3[2]); t3[3] = mul_w16<9>(t3[3]);
        dft4(t0[0], t1[0], t2[0], t3[0], u[0], u[4], u[8],  u[12]);
        dft4(t0[1], t1[1], t2[1], t3[1], u[1], u[5], u[9],  u[13]);
        dft4(t0[2], t1[2], t2[2], t3[2], u[2], u[6], u[10], u[14]);
        dft4(t0[3], t1[3], t2[3], t3[3], u[3], u[7], u[11], u[15]);
    }
};

// ---- compile-time description of one transform size ---------------------------
template <int N_, int P_, int R0_, int R1_ = 1, int R2_ = 1, int R3_ = 1>
struct Cfg {
    static constexpr int N = N_;
    static constexpr int P = P_;                 // points per lane
    static constexpr int T = N_ / P_;            // lanes per frame
    static constexpr int NPASS = (R1_ == 1) ? 1 : (R2_ == 1) ? 2 : (R3_ == 1) ? 3 : 4;
    static constexpr int WG = (T >= 64) ? T : 64;   // workgroup size
    static constexpr int F = WG / T;                // frames per workgroup
    static constexpr int radix(int s) { return s == 0 ? R0_ : s == 1 ? R1_ : s == 2 ? R2_ : R3_; }
    static constexpr int pbefore(int s)
    {
        int p = 1;
        for (int i = 0; i < s; i++) p *= radix(i);
        return p;
    }
    // twiddle registers: pass s >= 1 keeps B*(R-1) of them
    static constexpr int twcount(int s) { return s == 0 ? 0 : (P_ / radix(s)) * (radix(s) - 1); }
    static constexpr int twoff(int s)
    {
        int o = 0;
        for (int i = 0; i < s; i++) o += twcount(i);
        return o;
    }
    static constexpr int TW = twoff(NPASS) > 0 ? twoff(NPASS) : 1;
    // LDS image of one frame: one pad slot per 16 elements (kills the 16-way write
    // conflict of the first exchange, whose lanes write at stride 16 elements)
    static constexpr int LDS_FRAME = (NPASS > 1) ? (N_ + N_ / 16) : 0;
    static constexpr int LDS_ELEMS = (LDS_FRAME * F > 0) ? LDS_FRAME * F : 1;
    static_assert(R0_ * R1_ * R2_ * R3_ == N_, "radices must multiply to N");
    static_assert(N_ % P_ == 0 && P_ % R0_ == 0 && P_ % R1_ == 0 && P_ % R2_ == 0 && P_ % R3_ == 0, "bad P");
    static_assert(WG % T == 0, "frames must tile the workgroup");
};

__device__ __forceinline__ int lidx(int e) { return e + (e >> 4); }

// twN: master table exp(-2 pi i k / N), k in [0, N)
template <class C, int S>
__device__ __forceinline__ void load_tw_pass(cf (&tw)[C::TW], const cf *__restrict__ twN, int tid)
{
    constexpr int R = C::radix(S), B = C::P / R, p = C::pbefore(S);
    constexpr int step = C::N / (p * R);
#pragma unroll
    for (int b = 0; b < B; b++) {
        const int k = (tid + b * C::T) & (p - 1);
#pragma unroll
        for (int r = 1; r < R; r++) tw[C::twoff(S) + b * (R - 1) + (r - 1)] = twN[r * k * step];
    }
}

template <class C>
__device__ __forceinline__ void load_twiddles(cf (&tw)[C::TW], const cf *__restrict__ twN, int tid)
{
    if constexpr (C::NPASS > 1) load_tw_pass<C, 1>(tw, twN, tid);
    if constexpr (C::NPASS > 2) load_tw_pass<C, 2>(tw, twN, tid);
    if constexpr (C::NPASS > 3) load_tw_pass<C, 3>(tw, twN, tid);
}

template <class C, int S>
__device__ __forceinline__ void run_pass(cf (&w)[C::P], const cf (&tw)[C::TW], cf *__restrict__ lds, int tid)
{
    constexpr int R = C::radix(S), B = C::P / R, p = C::pbefore(S);
    constexpr bool last = (S == C::NPASS - 1);
    if constexpr (!last) __syncthreads();   // earlier readers of this LDS image are done
#pragma unroll
    for (int b = 0; b < B; b++) {
        cf u[R];
#pragma unroll
        for (int r = 0; r < R; r++) u[r] = w[b + r * B];
        if constexpr (p > 1) {
#pragma unroll
            for (int r = 1; r < R; r++) u[r] = cmul(u[r], tw[C::twoff(S) + b * (R - 1) + (r - 1)]);
        }
        Bfly<R>::run(u);
        if constexpr (last) {
#pragma unroll
            for (int r = 0; r < R; r++) w[b + r * B] = u[r];
        } else {
            const int i = tid + b * C::T;
            const int k = i & (p - 1);
            const int j = (i - k) * R + k;
#pragma unroll
            for (int r = 0; r < R; r++) lds[lidx(j + r * p)] = u[r];
        }
    }
    if constexpr (!last) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < C::P; m++) w[m] = lds[lidx(tid + m * C::T)];
    }
}

// full transform of the frame held in w[] (slot m = element tid + m*T), -j exponent
template <class C>
__device__ __forceinline__ void fft_in_regs(cf (&w)[C::P], const cf (&tw)[C::TW], cf *__restrict__ lds, int tid)
{
    run_pass<C, 0>(w, tw, lds, tid);
    if constexpr (C::NPASS > 1) run_pass<C, 1>(w, tw, lds, tid);
    if constexpr (C::NPASS > 2) run_pass<C, 2>(w, tw, lds, tid);
    if constexpr (C::NPASS > 3) run_pass<C, 3>(w, tw, lds, tid);
}

// ---- the size table: one tuned decomposition per power of two ------------------
template <int N> struct CfgFor;
template <> struct CfgFor<2>    { using type = Cfg<2, 2, 2>; };
template <> struct CfgFor<4>    { using type = Cfg<4, 4, 4>; };
template <> struct CfgFor<8>    { using type = Cfg<8, 8, 8>; };
template <> struct CfgFor<16>   { using type = Cfg<16, 16, 16>; };
template <> struct CfgFor<32>   { using type = Cfg<32, 8, 8, 4>; };
template <> struct CfgFor<64>   { using type = Cfg<64, 8, 8, 8>; };
template <> struct CfgFor<128>  { using type = Cfg<128, 16, 16, 8>; };
template <> struct CfgFor<256>  { using type = Cfg<256, 16, 16, 16>; };
template <> struct CfgFor<512>  { using type = Cfg<512, 8, 8, 8, 8>; };
template <> struct CfgFor<1024> { using type = Cfg<1024, 16, 16, 16, 4>; };
template <> struct CfgFor<2048> { using type = Cfg<2048, 16, 16, 16, 8>; };
template <> struct CfgFor<4096> { using type = Cfg<4096, 16, 16, 16, 16>; };

// expands BODY(N) for the runtime length `len` (power of two, 2..4096)
#define AETH_POW2_SWITCH(len, BODY, DEFAULT)                                            \
    switch (len) {                                                                     \
    case 2: BODY(2); case 4: BODY(4); case 8: BODY(8); case 16: BODY(16);              \
    case 32: BODY(32); case 64: BODY(64); case 128: BODY(128); case 256: BODY(256);    \
    case 512: BODY(512); case 1024: BODY(1024); case 2048: BODY(2048);                 \
    case 4096: BODY(4096);                                                             \
    default: DEFAULT;                                                                  \
    }

}  // namespace fftk
}  // namespace aeth
