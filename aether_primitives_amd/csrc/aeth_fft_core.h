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
//
// Arithmetic: a cf32 sample is one 64-bit VGPR pair and every complex operation is
// one or two packed-f32 VALU instructions (v_pk_add/mul/fma_f32).  The op_sel /
// neg_lo / neg_hi operand modifiers do the re<->im swaps and sign flips that
// multiplication by +-j, conjugation and the complex product need, so there are no
// v_mov shuffles; hipcc does not select those modifiers from C++, hence the
// one-instruction inline-asm primitives below (plain VALU: no hazards or waitcnts of
// their own, freely scheduled by the compiler).
//
// Direction is a template parameter S (sign of the exponent).  Tables always hold
// exp(-2 pi i k / N); S = +1 multiplies by the conjugate and flips the +-j rotations
// at compile time.  The reference's `fwd` is S = +1 (src/fft.rs:148 plans it with
// rustfft's inverse = true), `bwd` is S = -1.
#pragma once

#include <hip/hip_runtime.h>

#ifndef AETH_LDS_DB_LIMIT
#define AETH_LDS_DB_LIMIT (36 * 1024)   /* two exchange images per workgroup up to this many bytes (N <= 2048: one barrier per exchange; more would cost occupancy) */
#endif

namespace aeth {
namespace fftk {

typedef float cf __attribute__((ext_vector_type(2)));   // (re, im): one 64-bit register pair

__device__ __forceinline__ cf mk(float a, float b) { cf r = {a, b}; return r; }

// ---- one-instruction packed primitives ------------------------------------------
__device__ __forceinline__ cf cadd(cf a, cf b)
{
    cf d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d;
}
__device__ __forceinline__ cf csub(cf a, cf b)
{
    cf d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d;
}
// a + (-j) b = (a.re + b.im, a.im - b.re)
__device__ __forceinline__ cf cadd_mjb(cf a, cf b)
{
    cf d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a + (+j) b = (a.re - b.im, a.im + b.re)
__device__ __forceinline__ cf cadd_pjb(cf a, cf b)
{
    cf d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a + rot*b and a - rot*b with rot = exp(S*j*pi/2) = S*j
template <int S> __device__ __forceinline__ cf cadd_rot(cf a, cf b) { return S < 0 ? cadd_mjb(a, b) : cadd_pjb(a, b); }
template <int S> __device__ __forceinline__ cf csub_rot(cf a, cf b) { return S < 0 ? cadd_pjb(a, b) : cadd_mjb(a, b); }

// Dependent instruction pairs are kept inside ONE asm statement: hipcc pads an s_nop
// between two asm statements when the second reads what the first wrote (it cannot see
// that v_pk_*_f32 has no dst-select forwarding hazard); inside a statement the hardware's
// own VALU interlock is all that is needed.
#define AETH_CMUL_MUL  " op_sel:[0,0] op_sel_hi:[0,1]"
#define AETH_CMUL_FMA  " op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[0,0,0]"
#define AETH_CMULC_MUL " op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]"
#define AETH_CMULC_FMA " op_sel:[1,1,0] op_sel_hi:[1,0,1]"

// a * w (w in VGPRs)
__device__ __forceinline__ cf cmul(cf a, cf w)
{
    cf d;
    asm("v_pk_mul_f32 %0, %1, %2" AETH_CMUL_MUL "\n\tv_pk_fma_f32 %0, %1, %2, %0" AETH_CMUL_FMA
        : "=&v"(d) : "v"(a), "v"(w));
    return d;
}
// a * conj(w)
__device__ __forceinline__ cf cmul_conj(cf a, cf w)
{
    cf d;
    asm("v_pk_mul_f32 %0, %1, %2" AETH_CMULC_MUL "\n\tv_pk_fma_f32 %0, %1, %2, %0" AETH_CMULC_FMA
        : "=&v"(d) : "v"(a), "v"(w));
    return d;
}
// a * (S < 0 ? w : conj(w)) : twiddle for exponent sign S from a table of exp(-j...)
template <int S> __device__ __forceinline__ cf ctw(cf a, cf w) { return S < 0 ? cmul(a, w) : cmul_conj(a, w); }

// the same with a wave-uniform constant held in an SGPR pair
__device__ __forceinline__ cf cmul_k(cf a, cf w)
{
    cf d;
    asm("v_pk_mul_f32 %0, %1, %2" AETH_CMUL_MUL "\n\tv_pk_fma_f32 %0, %1, %2, %0" AETH_CMUL_FMA
        : "=&v"(d) : "v"(a), "s"(w));
    return d;
}
__device__ __forceinline__ cf cmul_conj_k(cf a, cf w)
{
    cf d;
    asm("v_pk_mul_f32 %0, %1, %2" AETH_CMULC_MUL "\n\tv_pk_fma_f32 %0, %1, %2, %0" AETH_CMULC_FMA
        : "=&v"(d) : "v"(a), "s"(w));
    return d;
}
template <int S> __device__ __forceinline__ cf ctw_k(cf a, cf w) { return S < 0 ? cmul_k(a, w) : cmul_conj_k(a, w); }

// a * s, s real and wave-uniform; ss = (s, s)
__device__ __forceinline__ cf cscale_k(cf a, cf ss)
{
    cf d; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "s"(ss)); return d;
}

// rotations by odd multiples of pi/4, as "x +- (+-j) x" sums:
//   R1(a) = sqrt2 * a * exp(S*j*pi/4)    S=-1: a(1-j) = (re+im, im-re);   S=+1: a(1+j) = (re-im, im+re)
//   R3(a) = sqrt2 * a * exp(S*j*3pi/4)   S=-1: a(-1-j) = (im-re, -re-im); S=+1: a(-1+j) = (-re-im, re-im)
#define AETH_R1_M " op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]"
#define AETH_R1_P " op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]"
#define AETH_R3_M " op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,1]"
#define AETH_R3_P " op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,1] neg_hi:[1,0]"

// a * W8^K for exponent sign S (K = 1 or 3): rotate, then scale by 1/sqrt2 (hh = (sqrt.5, sqrt.5))
template <int S, int K> __device__ __forceinline__ cf cmul_w8(cf a, cf hh)
{
    cf d;
    if constexpr (K == 1 && S < 0) asm("v_pk_add_f32 %0, %1, %1" AETH_R1_M "\n\tv_pk_mul_f32 %0, %0, %2" : "=&v"(d) : "v"(a), "s"(hh));
    if constexpr (K == 1 && S > 0) asm("v_pk_add_f32 %0, %1, %1" AETH_R1_P "\n\tv_pk_mul_f32 %0, %0, %2" : "=&v"(d) : "v"(a), "s"(hh));
    if constexpr (K == 3 && S < 0) asm("v_pk_add_f32 %0, %1, %1" AETH_R3_M "\n\tv_pk_mul_f32 %0, %0, %2" : "=&v"(d) : "v"(a), "s"(hh));
    if constexpr (K == 3 && S > 0) asm("v_pk_add_f32 %0, %1, %1" AETH_R3_P "\n\tv_pk_mul_f32 %0, %0, %2" : "=&v"(d) : "v"(a), "s"(hh));
    return d;
}
// (e + W8^K o, e - W8^K o): rotate o, then two fused multiply-adds by +-1/sqrt2
template <int S, int K> __device__ __forceinline__ void bfly_w8(cf e, cf o, cf hh, cf &plus, cf &minus)
{
    cf r;
#define AETH_W8_TAIL "\n\tv_pk_fma_f32 %0, %2, %5, %4\n\tv_pk_fma_f32 %1, %2, %5, %4 neg_lo:[1,0,0] neg_hi:[1,0,0]"
    if constexpr (K == 1 && S < 0) asm("v_pk_add_f32 %2, %3, %3" AETH_R1_M AETH_W8_TAIL : "=&v"(plus), "=&v"(minus), "=&v"(r) : "v"(o), "v"(e), "s"(hh));
    if constexpr (K == 1 && S > 0) asm("v_pk_add_f32 %2, %3, %3" AETH_R1_P AETH_W8_TAIL : "=&v"(plus), "=&v"(minus), "=&v"(r) : "v"(o), "v"(e), "s"(hh));
    if constexpr (K == 3 && S < 0) asm("v_pk_add_f32 %2, %3, %3" AETH_R3_M AETH_W8_TAIL : "=&v"(plus), "=&v"(minus), "=&v"(r) : "v"(o), "v"(e), "s"(hh));
    if constexpr (K == 3 && S > 0) asm("v_pk_add_f32 %2, %3, %3" AETH_R3_P AETH_W8_TAIL : "=&v"(plus), "=&v"(minus), "=&v"(r) : "v"(o), "v"(e), "s"(hh));
#undef AETH_W8_TAIL
}

// ---- plain C++ helpers (generic kernels, not on the hot path) ---------------------
__device__ __forceinline__ cf cswap(cf a) { return mk(a.y, a.x); }
__device__ __forceinline__ cf cscale(cf a, float s) { return mk(a.x * s, a.y * s); }
__device__ __forceinline__ cf mul_mj(cf a) { return mk(a.y, -a.x); }
__device__ __forceinline__ cf cmul_plain(cf a, cf w) { return mk(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }
__device__ __forceinline__ cf cadd_plain(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub_plain(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }

constexpr float kSqrtHalf = 0.70710678118654752440f;
constexpr float kCosPi8 = 0.92387953251128675613f;
constexpr float kSinPi8 = 0.38268343236508977173f;

// 4-point DFT in place on (a, b, c, d) -> (y0, y1, y2, y3), exponent sign S.
//   y1 = (a-c) + rot (b-d),  y3 = (a-c) - rot (b-d),  rot = S*j
// One asm statement of 8 packed instructions: hipcc pads a wait state after every asm
// statement whose result the next instruction may read, so grouping the butterfly keeps
// those s_nop's out of the inner sequence.  CROT: input c stands for rot*c (the free
// W16^4 twiddle inside radix 16).
#define AETH_MJ " op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]"   /* x + (-j) y */
#define AETH_PJ " op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]"   /* x + (+j) y */
#define AETH_NEG " neg_lo:[0,1] neg_hi:[0,1]"                               /* x - y      */
template <int S, bool CROT>
__device__ __forceinline__ void dft4_inplace(cf &a, cf &b, cf &c, cf &d)
{
    cf t0, t1, t2, t3;
    if constexpr (S < 0 && !CROT)
        asm("v_pk_add_f32 %4, %0, %2\n\tv_pk_add_f32 %5, %0, %2" AETH_NEG "\n\t"
            "v_pk_add_f32 %6, %1, %3\n\tv_pk_add_f32 %7, %1, %3" AETH_NEG "\n\t"
            "v_pk_add_f32 %0, %4, %6\n\tv_pk_add_f32 %2, %4, %6" AETH_NEG "\n\t"
            "v_pk_add_f32 %1, %5, %7" AETH_MJ "\n\tv_pk_add_f32 %3, %5, %7" AETH_PJ
            : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));
    else if constexpr (S > 0 && !CROT)
        asm("v_pk_add_f32 %4, %0, %2\n\tv_pk_add_f32 %5, %0, %2" AETH_NEG "\n\t"
            "v_pk_add_f32 %6, %1, %3\n\tv_pk_add_f32 %7, %1, %3" AETH_NEG "\n\t"
            "v_pk_add_f32 %0, %4, %6\n\tv_pk_add_f32 %2, %4, %6" AETH_NEG "\n\t"
            "v_pk_add_f32 %1, %5, %7" AETH_PJ "\n\tv_pk_add_f32 %3, %5, %7" AETH_MJ
            : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));
    else if constexpr (S < 0 && CROT)
        asm("v_pk_add_f32 %4, %0, %2" AETH_MJ "\n\tv_pk_add_f32 %5, %0, %2" AETH_PJ "\n\t"
            "v_pk_add_f32 %6, %1, %3\n\tv_pk_add_f32 %7, %1, %3" AETH_NEG "\n\t"
            "v_pk_add_f32 %0, %4, %6\n\tv_pk_add_f32 %2, %4, %6" AETH_NEG "\n\t"
            "v_pk_add_f32 %1, %5, %7" AETH_MJ "\n\tv_pk_add_f32 %3, %5, %7" AETH_PJ
            : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));
    else
        asm("v_pk_add_f32 %4, %0, %2" AETH_PJ "\n\tv_pk_add_f32 %5, %0, %2" AETH_MJ "\n\t"
            "v_pk_add_f32 %6, %1, %3\n\tv_pk_add_f32 %7, %1, %3" AETH_NEG "\n\t"
            "v_pk_add_f32 %0, %4, %6\n\tv_pk_add_f32 %2, %4, %6" AETH_NEG "\n\t"
            "v_pk_add_f32 %1, %5, %7" AETH_PJ "\n\tv_pk_add_f32 %3, %5, %7" AETH_MJ
            : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));
}

template <int S>
__device__ __forceinline__ void dft4(cf a, cf b, cf c, cf d, cf &y0, cf &y1, cf &y2, cf &y3)
{
    dft4_inplace<S, false>(a, b, c, d);
    y0 = a; y1 = b; y2 = c; y3 = d;
}
template <int S>
__device__ __forceinline__ void dft4_crot(cf a, cf b, cf c, cf d, cf &y0, cf &y1, cf &y2, cf &y3)
{
    dft4_inplace<S, true>(a, b, c, d);
    y0 = a; y1 = b; y2 = c; y3 = d;
}

// in-register radix-R DFT, natural-order in, natural-order out, exponent sign S
template <int R, int S> struct Bfly;

template <int S> struct Bfly<1, S> { static __device__ __forceinline__ void run(cf (&)[1]) {} };

template <int S> struct Bfly<2, S> {
    static __device__ __forceinline__ void run(cf (&u)[2])
    {
        cf a = u[0], b = u[1];
        u[0] = cadd(a, b);
        u[1] = csub(a, b);
    }
};

template <int S> struct Bfly<4, S> {
    static __device__ __forceinline__ void run(cf (&u)[4]) { dft4<S>(u[0], u[1], u[2], u[3], u[0], u[1], u[2], u[3]); }
};

template <int S> struct Bfly<8, S> {
    static __device__ __forceinline__ void run(cf (&u)[8])
    {
        const cf hh = mk(kSqrtHalf, kSqrtHalf);
        cf e0 = u[0], e1 = u[2], e2 = u[4], e3 = u[6], o0 = u[1], o1 = u[3], o2 = u[5], o3 = u[7];
        dft4_inplace<S, false>(e0, e1, e2, e3);
        dft4_inplace<S, false>(o0, o1, o2, o3);
        u[0] = cadd(e0, o0);        u[4] = csub(e0, o0);
        bfly_w8<S, 1>(e1, o1, hh, u[1], u[5]);
        u[2] = cadd_rot<S>(e2, o2); u[6] = csub_rot<S>(e2, o2);
        bfly_w8<S, 3>(e3, o3, hh, u[3], u[7]);
    }
};

template <int S> struct Bfly<16, S> {
    static __device__ __forceinline__ void run(cf (&u)[16])
    {
        const cf hh = mk(kSqrtHalf, kSqrtHalf);
        const cf w1 = mk(kCosPi8, -kSinPi8), w3 = mk(kSinPi8, -kCosPi8), w9 = mk(-kCosPi8, kSinPi8);
        // rows: 4-point DFTs over p of x[q + 4p]; t_q[kp] lands in u[q + 4 kp]
        dft4_inplace<S, false>(u[0], u[4], u[8],  u[12]);
        dft4_inplace<S, false>(u[1], u[5], u[9],  u[13]);
        dft4_inplace<S, false>(u[2], u[6], u[10], u[14]);
        dft4_inplace<S, false>(u[3], u[7], u[11], u[15]);
        // column kp: z[q] = t_q[kp] * W16^(q*kp), then a 4-point DFT over q -> X[kp + 4 kq]
        cf c0[4] = {u[0], u[1], u[2], u[3]};
        cf c1[4] = {u[4], ctw_k<S>(u[5], w1), cmul_w8<S, 1>(u[6], hh), ctw_k<S>(u[7], w3)};          // W16^1,2,3
        cf c2[4] = {u[8], cmul_w8<S, 1>(u[9], hh), u[10], cmul_w8<S, 3>(u[11], hh)};               // W16^2,(4),6
        cf c3[4] = {u[12], ctw_k<S>(u[13], w3), cmul_w8<S, 3>(u[14], hh), ctw_k<S>(u[15], w9)};     // W16^3,6,9
        dft4_inplace<S, false>(c0[0], c0[1], c0[2], c0[3]);
        dft4_inplace<S, false>(c1[0], c1[1], c1[2], c1[3]);
        dft4_inplace<S, true>(c2[0], c2[1], c2[2], c2[3]);      // W16^4 = rot, folded into the adds
        dft4_inplace<S, false>(c3[0], c3[1], c3[2], c3[3]);
#pragma unroll
        for (int kq = 0; kq < 4; kq++) {
            u[0 + 4 * kq] = c0[kq];
            u[1 + 4 * kq] = c1[kq];
            u[2 + 4 * kq] = c2[kq];
            u[3 + 4 * kq] = c3[kq];
        }
    }
};

// ---- radices 3 and 5, and composite radices built from two register levels ---------------
// (register-resident transforms of lengths with factors 3 and 5: 100 = 10*10, 1000 = 10*10*10, 1536 = 24*8*8 ...)
constexpr double kPiD = 3.14159265358979323846264338327950288;
constexpr double c_sin_taylor(double x)
{
    double term = x, sum = x;
    for (int k = 1; k < 14; k++) { term *= -x * x / ((2 * k) * (2 * k + 1)); sum += term; }
    return sum;
}
constexpr double c_sin(double x)
{
    while (x > kPiD) x -= 2 * kPiD;
    while (x < -kPiD) x += 2 * kPiD;
    if (x > kPiD / 2) x = kPiD - x;
    if (x < -kPiD / 2) x = -kPiD - x;
    return c_sin_taylor(x);
}
constexpr double c_cos(double x) { return c_sin(x + kPiD / 2); }

template <int S> struct Bfly<3, S> {
    static __device__ __forceinline__ void run(cf (&u)[3])
    {
        constexpr float s60 = (float)c_sin(kPiD / 3);
        const cf t1 = cadd(u[1], u[2]), t2 = csub(u[1], u[2]);
        const cf y0 = cadd(u[0], t1);
        const cf h = cadd(y0, cscale_k(t1, mk(-1.5f, -1.5f)));
        const cf jt = cscale_k(t2, mk(s60, s60));
        u[0] = y0;
        u[1] = cadd_rot<S>(h, jt);
        u[2] = csub_rot<S>(h, jt);
    }
};

template <int S> struct Bfly<5, S> {
    static __device__ __forceinline__ void run(cf (&u)[5])
    {
        constexpr float cb = (float)((c_cos(2 * kPiD / 5) - c_cos(4 * kPiD / 5)) / 2);   // 0.559...
        constexpr float s1 = (float)c_sin(2 * kPiD / 5), s2 = (float)c_sin(4 * kPiD / 5);
        const cf t1 = cadd(u[1], u[4]), t2 = cadd(u[2], u[3]), t3 = csub(u[1], u[4]), t4 = csub(u[2], u[3]);
        const cf t5 = cadd(t1, t2);
        const cf y0 = cadd(u[0], t5);
        const cf m1 = cadd(y0, cscale_k(t5, mk(-1.25f, -1.25f)));
        const cf m2 = cscale_k(csub(t1, t2), mk(cb, cb));
        const cf a1 = cadd(m1, m2), a2 = csub(m1, m2);
        const cf b1 = cadd(cscale_k(t3, mk(s1, s1)), cscale_k(t4, mk(s2, s2)));
        const cf b2 = csub(cscale_k(t3, mk(s2, s2)), cscale_k(t4, mk(s1, s1)));
        u[0] = y0;
        u[1] = cadd_rot<S>(a1, b1);
        u[4] = csub_rot<S>(a1, b1);
        u[2] = cadd_rot<S>(a2, b2);
        u[3] = csub_rot<S>(a2, b2);
    }
};

template <int S> struct Bfly<7, S> {
    static __device__ __forceinline__ void run(cf (&u)[7])
    {
        constexpr float c1 = (float)c_cos(2 * kPiD / 7), c2 = (float)c_cos(4 * kPiD / 7), c3 = (float)c_cos(6 * kPiD / 7);
        constexpr float s1 = (float)c_sin(2 * kPiD / 7), s2 = (float)c_sin(4 * kPiD / 7), s3 = (float)c_sin(6 * kPiD / 7);
        const cf t1 = cadd(u[1], u[6]), t2 = cadd(u[2], u[5]), t3 = cadd(u[3], u[4]);
        const cf d1 = csub(u[1], u[6]), d2 = csub(u[2], u[5]), d3 = csub(u[3], u[4]);
        const cf y0 = cadd(cadd(u[0], t1), cadd(t2, t3));
        const cf a1 = u[0] + t1 * c1 + t2 * c2 + t3 * c3;
        const cf a2 = u[0] + t1 * c2 + t2 * c3 + t3 * c1;
        const cf a3 = u[0] + t1 * c3 + t2 * c1 + t3 * c2;
        const cf b1 = d1 * s1 + d2 * s2 + d3 * s3;
        const cf b2 = d1 * s2 - d2 * s3 - d3 * s1;
        const cf b3 = d1 * s3 - d2 * s1 + d3 * s2;
        u[0] = y0;
        u[1] = cadd_rot<S>(a1, b1);
        u[6] = csub_rot<S>(a1, b1);
        u[2] = cadd_rot<S>(a2, b2);
        u[5] = csub_rot<S>(a2, b2);
        u[3] = cadd_rot<S>(a3, b3);
        u[4] = csub_rot<S>(a3, b3);
    }
};

// odd prime P by the definition, folded over the pairs (k, P-k): (P-1)^2/2 real-by-complex multiply-adds with the
// cosines and sines from a compile-time table (11 and 13: beyond that the LDS kernel's generic pass)
template <int P> struct OddPrimeTable {
    float c[P], s[P];
    constexpr OddPrimeTable() : c{}, s{}
    {
        for (int i = 0; i < P; i++) {
            c[i] = (float)c_cos(2 * kPiD * (double)i / (double)P);
            s[i] = (float)c_sin(2 * kPiD * (double)i / (double)P);
        }
    }
};

template <int P, int S> struct BflyOddPrime {
    static __device__ __forceinline__ void run(cf (&u)[P])
    {
        constexpr int H = (P - 1) / 2;
        constexpr OddPrimeTable<P> tab{};
        cf t[H], d[H];
#pragma unroll
        for (int k = 0; k < H; k++) {
            t[k] = cadd(u[k + 1], u[P - 1 - k]);
            d[k] = csub(u[k + 1], u[P - 1 - k]);
        }
        const cf x0 = u[0];
        cf y0 = x0;
#pragma unroll
        for (int k = 0; k < H; k++) y0 = cadd(y0, t[k]);
        u[0] = y0;
#pragma unroll
        for (int m = 1; m <= H; m++) {
            cf a = x0, b = mk(0.f, 0.f);
#pragma unroll
            for (int k = 1; k <= H; k++) {
                a = a + t[k - 1] * tab.c[(m * k) % P];
                b = b + d[k - 1] * tab.s[(m * k) % P];
            }
            u[m] = cadd_rot<S>(a, b);
            u[P - m] = csub_rot<S>(a, b);
        }
    }
};
template <int S> struct Bfly<11, S> : BflyOddPrime<11, S> {};
template <int S> struct Bfly<13, S> : BflyOddPrime<13, S> {};
template <int S> struct Bfly<17, S> : BflyOddPrime<17, S> {};
template <int S> struct Bfly<19, S> : BflyOddPrime<19, S> {};
template <int S> struct Bfly<23, S> : BflyOddPrime<23, S> {};

// R = R1*R2:  X[k1 + R1*k2] = sum_n2 W_R2^(n2 k2) [ W_R^(n2 k1) sum_n1 x[n1*R2 + n2] W_R1^(n1 k1) ]
// the inner twiddles W_R^(n2 k1) are compile-time constants held in SGPR pairs
template <int R1, int R2, int S> struct BflyC {
    static __device__ __forceinline__ void run(cf (&u)[R1 * R2])
    {
        constexpr int R = R1 * R2;
#pragma unroll
        for (int n2 = 0; n2 < R2; n2++) {
            cf t[R1];
#pragma unroll
            for (int n1 = 0; n1 < R1; n1++) t[n1] = u[n1 * R2 + n2];
            Bfly<R1, S>::run(t);
#pragma unroll
            for (int k1 = 0; k1 < R1; k1++) {
                if (n2 > 0 && k1 > 0) {
                    const double ang = 2 * kPiD * (double)((n2 * k1) % R) / (double)R;
                    u[k1 * R2 + n2] = ctw_k<S>(t[k1], mk((float)c_cos(ang), (float)(-c_sin(ang))));
                } else u[k1 * R2 + n2] = t[k1];
            }
        }
        cf x[R];
#pragma unroll
        for (int k1 = 0; k1 < R1; k1++) {
            cf t[R2];
#pragma unroll
            for (int n2 = 0; n2 < R2; n2++) t[n2] = u[k1 * R2 + n2];
            Bfly<R2, S>::run(t);
#pragma unroll
            for (int k2 = 0; k2 < R2; k2++) x[k1 + R1 * k2] = t[k2];
        }
#pragma unroll
        for (int r = 0; r < R; r++) u[r] = x[r];
    }
};
template <int S> struct Bfly<6, S>  : BflyC<3, 2, S> {};
template <int S> struct Bfly<10, S> : BflyC<5, 2, S> {};
template <int S> struct Bfly<12, S> : BflyC<4, 3, S> {};
template <int S> struct Bfly<15, S> : BflyC<5, 3, S> {};
template <int S> struct Bfly<20, S> : BflyC<5, 4, S> {};
template <int S> struct Bfly<24, S> : BflyC<8, 3, S> {};
template <int S> struct Bfly<25, S> : BflyC<5, 5, S> {};
template <int S> struct Bfly<9, S>  : BflyC<3, 3, S> {};
template <int S> struct Bfly<18, S> : BflyC<6, 3, S> {};
template <int S> struct Bfly<27, S> : BflyC<9, 3, S> {};
template <int S> struct Bfly<30, S> : BflyC<6, 5, S> {};
template <int S> struct Bfly<32, S> : BflyC<8, 4, S> {};
template <int S> struct Bfly<14, S> : BflyC<7, 2, S> {};
template <int S> struct Bfly<21, S> : BflyC<7, 3, S> {};
template <int S> struct Bfly<28, S> : BflyC<7, 4, S> {};
template <int S> struct Bfly<22, S> : BflyC<11, 2, S> {};
template <int S> struct Bfly<26, S> : BflyC<13, 2, S> {};
template <int S> struct Bfly<34, S> : BflyC<17, 2, S> {};

// ---- compile-time description of one transform size ---------------------------
// WG_ = 0: the power-of-two rule (T lanes, at least 64).  WG_ > 0: that many lanes; F = WG / T frames, the
// WG - F*T lanes left over idle on a dummy LDS frame (lengths whose T does not divide a wave)
template <int N_, int P_, int R0_, int R1_ = 1, int R2_ = 1, int R3_ = 1, int WG_ = 0>
struct Cfg {
    static constexpr int N = N_;
    static constexpr int P = P_;                 // points per lane
    static constexpr int T = N_ / P_;            // lanes per frame
    static constexpr int NPASS = (R1_ == 1) ? 1 : (R2_ == 1) ? 2 : (R3_ == 1) ? 3 : 4;
    static constexpr int WG = WG_ > 0 ? WG_ : (T >= 64) ? T : 64;   // workgroup size
    static constexpr int F = WG / T;                // frames per workgroup
    static constexpr int IDLE = WG - F * T;         // lanes beyond the last whole frame
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
    static constexpr int LDS_ELEMS = (LDS_FRAME * F > 0) ? LDS_FRAME * (F + (IDLE > 0 ? 1 : 0)) : 1;   // one image of the workgroup's frames (+ a dummy one for idle lanes)
    static constexpr bool DB = (NPASS > 1) && (2 * LDS_ELEMS * 8 <= AETH_LDS_DB_LIMIT);   // room for two images
    static constexpr int LDS_TOTAL = DB ? 2 * LDS_ELEMS : LDS_ELEMS;
    static_assert(R0_ * R1_ * R2_ * R3_ == N_, "radices must multiply to N");
    static_assert(N_ % P_ == 0 && P_ % R0_ == 0 && P_ % R1_ == 0 && P_ % R2_ == 0 && P_ % R3_ == 0, "bad P");
    static_assert(WG_ > 0 || WG % T == 0, "frames must tile the workgroup");
    static_assert(F >= 1, "a frame needs T lanes");
};

__device__ __forceinline__ int lidx(int e) { return e + (e >> 4); }
// XOR swizzle of the exchange image (tuning, XP bit 3): element e sits in slot e ^ ((e >> 4) & 15) -- a permutation
// inside every aligned group of 16 elements.  Like the pad it spreads the stride-16 writes of the first exchange over
// 16 slots, and unlike the pad it keeps 32 consecutive elements inside one 256-byte bank row, so the contiguous
// reads lose their two-way conflict.
template <int XP> __device__ __forceinline__ int pidx(int e) { return (XP & 8) ? (e ^ ((e >> 4) & 15)) : lidx(e); }

// twN: master table exp(-2 pi i k / N), k in [0, N)
template <class C, int PASS>
__device__ __forceinline__ void load_tw_pass(cf (&tw)[C::TW], const cf *__restrict__ twN, int tid)
{
    constexpr int R = C::radix(PASS), B = C::P / R, p = C::pbefore(PASS);
    constexpr int step = C::N / (p * R);
#pragma unroll
    for (int b = 0; b < B; b++) {
        const int k = (tid + b * C::T) % p;
#pragma unroll
        for (int r = 1; r < R; r++) tw[C::twoff(PASS) + b * (R - 1) + (r - 1)] = twN[r * k * step];
    }
}

template <class C>
__device__ __forceinline__ void load_twiddles(cf (&tw)[C::TW], const cf *__restrict__ twN, int tid)
{
    if constexpr (C::NPASS > 1) load_tw_pass<C, 1>(tw, twN, tid);
    if constexpr (C::NPASS > 2) load_tw_pass<C, 2>(tw, twN, tid);
    if constexpr (C::NPASS > 3) load_tw_pass<C, 3>(tw, twN, tid);
}

// The same registers from the plan's per-lane table (built once per plan by
// build_lane_twiddles): one coalesced load per register instead of a 64-way gather from
// the master table, which matters because every launch pays this prologue.
// Layout: the registers of pass s, butterfly b, power r form row (s, b, r).  The
// twiddle depends on the lane only through k = (tid + b*T) mod p_s, so a row holds
// min(p_s, T) entries: T when p_s >= T (one per lane), p_s when the pass right after
// the first exchange repeats every p_s lanes (2048 = 16*16*8: 16 entries instead of 128;
// the table every workgroup pulls through L2 shrinks from 58 KiB to 30 KiB).
template <class C> struct LaneTable {
    static constexpr int rowlen(int pass) { return C::pbefore(pass) < C::T ? C::pbefore(pass) : C::T; }
    static constexpr int rows(int pass) { return C::twcount(pass); }
    static constexpr int offset(int pass)
    {
        int o = 0;
        for (int i = 1; i < pass; i++) o += rows(i) * rowlen(i);
        return o;
    }
    static constexpr int ELEMS = offset(C::NPASS) > 0 ? offset(C::NPASS) : 1;
};

template <class C, int PASS>
__device__ __forceinline__ void lane_tw_pass(cf (&tw)[C::TW], const cf *__restrict__ twL, int tid, bool store)
{
    constexpr int R = C::radix(PASS), B = C::P / R, p = C::pbefore(PASS);
    constexpr int L = LaneTable<C>::rowlen(PASS);
    cf *wr = const_cast<cf *>(twL);
#pragma unroll
    for (int b = 0; b < B; b++) {
        const int lane = (p < C::T) ? ((tid + b * C::T) % p) : tid;
#pragma unroll
        for (int r = 1; r < R; r++) {
            const int slot = C::twoff(PASS) + b * (R - 1) + (r - 1);
            const int idx = LaneTable<C>::offset(PASS) + (b * (R - 1) + (r - 1)) * L + lane;
            if (store) wr[idx] = tw[slot];
            else tw[slot] = twL[idx];
        }
    }
}

template <class C>
__device__ __forceinline__ void load_twiddles_lane(cf (&tw)[C::TW], const cf *__restrict__ twL, int tid)
{
    if constexpr (C::NPASS > 1) lane_tw_pass<C, 1>(tw, twL, tid, false);
    if constexpr (C::NPASS > 2) lane_tw_pass<C, 2>(tw, twL, tid, false);
    if constexpr (C::NPASS > 3) lane_tw_pass<C, 3>(tw, twL, tid, false);
}

template <class C>
__global__ void build_lane_twiddles(const cf *__restrict__ twN, cf *__restrict__ twL)
{
    const int tid = threadIdx.x;
    if (tid >= C::T) return;
    cf tw[C::TW];
    load_twiddles<C>(tw, twN, tid);          // (lanes that share a row entry write the same value)
    if constexpr (C::NPASS > 1) lane_tw_pass<C, 1>(tw, twL, tid, true);
    if constexpr (C::NPASS > 2) lane_tw_pass<C, 2>(tw, twL, tid, true);
    if constexpr (C::NPASS > 3) lane_tw_pass<C, 3>(tw, twL, tid, true);
}

// Exchange buffers: where two LDS images of the frame fit (C::DB) consecutive exchanges
// ping-pong between them, which needs ONE barrier per exchange (write k -> barrier ->
// read k; image k is next written only after the barrier of exchange k+1, which every
// lane passes after it has finished reading image k).  With a single image a second
// barrier in front of the writes keeps late readers safe.
// XP (tuning): bit 0: the exchange runs at s_setprio 1, the butterflies at 0; bit 1: no barriers, bit 2: no LDS
// traffic at all (both diagnosis only, wrong results)
// XP bit 4 (16): raw `s_barrier` behind an explicit lgkmcnt(0) instead of __syncthreads(): the fence of __syncthreads()
// drains vmcnt(0) whenever an LDS-DMA (a pending LDS write on the VM counter) is in flight, which would make every
// exchange wait for the window that is still landing (aeth_fir_kernel.h, V_DMA)
template <int XP> __device__ __forceinline__ void wg_barrier()
{
    if constexpr (XP & 16) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    } else __syncthreads();
}

template <class C, int PASS, int S, int PAR, int XP = 0>
__device__ __forceinline__ void run_pass(cf (&w)[C::P], const cf (&tw)[C::TW], cf *__restrict__ lds, int tid)
{
    constexpr int R = C::radix(PASS), B = C::P / R, p = C::pbefore(PASS);
    constexpr bool last = (PASS == C::NPASS - 1);
    constexpr bool NOLDS = (XP & 4) != 0;
    cf *img = lds;
    if constexpr (!last && !NOLDS) {
        if constexpr (C::DB) img = lds + (((PAR + PASS) & 1) ? C::LDS_ELEMS : 0);
        else if constexpr (XP & 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        else wg_barrier<XP>();              // earlier readers of the single image are done
    }
#pragma unroll
    for (int b = 0; b < B; b++) {
        cf u[R];
#pragma unroll
        for (int r = 0; r < R; r++) u[r] = w[b + r * B];
        if constexpr (p > 1) {
#pragma unroll
            for (int r = 1; r < R; r++) u[r] = ctw<S>(u[r], tw[C::twoff(PASS) + b * (R - 1) + (r - 1)]);
        }
        Bfly<R, S>::run(u);
        if constexpr (last || NOLDS) {
#pragma unroll
            for (int r = 0; r < R; r++) w[b + r * B] = u[r];
        } else {
            const int i = tid + b * C::T;
            const int k = i % p;
            const int j = (i - k) * R + k;
#pragma unroll
            for (int r = 0; r < R; r++) img[pidx<XP>(j + r * p)] = u[r];
        }
    }
    if constexpr (!last && !NOLDS) {
        if constexpr (XP & 1) __builtin_amdgcn_s_setprio(1);
        if constexpr (XP & 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        else wg_barrier<XP>();
#pragma unroll
        for (int m = 0; m < C::P; m++) w[m] = img[pidx<XP>(tid + m * C::T)];
        if constexpr (XP & 1) __builtin_amdgcn_s_setprio(0);
    }
}

// full transform of the frame held in w[] (slot m = element tid + m*T), exponent sign S.
// PAR = parity of the number of exchanges done so far on this LDS allocation (C::DB only);
// the caller keeps it consistent (see fft_next_par).
template <class C, int S, int PAR = 0, int XP = 0>
__device__ __forceinline__ void fft_in_regs(cf (&w)[C::P], const cf (&tw)[C::TW], cf *__restrict__ lds, int tid)
{
    run_pass<C, 0, S, PAR, XP>(w, tw, lds, tid);
    if constexpr (C::NPASS > 1) run_pass<C, 1, S, PAR, XP>(w, tw, lds, tid);
    if constexpr (C::NPASS > 2) run_pass<C, 2, S, PAR, XP>(w, tw, lds, tid);
    if constexpr (C::NPASS > 3) run_pass<C, 3, S, PAR, XP>(w, tw, lds, tid);
}
template <class C> constexpr int fft_next_par(int par) { return (par + C::NPASS - 1) & 1; }

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
// 8192: the largest frame whose padded LDS image (68 KiB) and 512 lanes x 16 points still leave the
// register prefetch stage room at 2 waves per SIMD; plain transforms only (AETH_POW2_SWITCH_XL)
template <> struct CfgFor<8192> { using type = Cfg<8192, 16, 16, 16, 8, 4>; };

// expands BODY(N) for the runtime length `len` (power of two, 2..4096)
#define AETH_POW2_SWITCH(len, BODY, DEFAULT)                                            \
    switch (len) {                                                                     \
    case 2: BODY(2); case 4: BODY(4); case 8: BODY(8); case 16: BODY(16);              \
    case 32: BODY(32); case 64: BODY(64); case 128: BODY(128); case 256: BODY(256);    \
    case 512: BODY(512); case 1024: BODY(1024); case 2048: BODY(2048);                 \
    case 4096: BODY(4096);                                                             \
    default: DEFAULT;                                                                  \
    }

// the same plus the lengths only the plain transform kernels are built for
#define AETH_POW2_SWITCH_XL(len, BODY, DEFAULT)                                         \
    switch (len) {                                                                     \
    case 2: BODY(2); case 4: BODY(4); case 8: BODY(8); case 16: BODY(16);              \
    case 32: BODY(32); case 64: BODY(64); case 128: BODY(128); case 256: BODY(256);    \
    case 512: BODY(512); case 1024: BODY(1024); case 2048: BODY(2048);                 \
    case 4096: BODY(4096); case 8192: BODY(8192);                                      \
    default: DEFAULT;                                                                  \
    }

}  // namespace fftk
}  // namespace aeth
