// aeth_rng.h -- counter-based complex normal generator used by the AWGN kernel.
// Plain C, integer + f32 (+,-,*,sqrt, explicit fmaf) only, fixed operation order: compiled with
// -ffp-contract=off it is bit-reproducible between host compilers and the GPU.
#pragma once
#include <math.h>
#include <stdint.h>
#ifndef AETH_RNG_FN
#define AETH_RNG_FN static inline
#endif

/* ---- counter-based complex normal generator (the BUILD's own; the reference's
 * StdRng + rand_distr::Normal stream, src/noise.rs:2-4,29-44, cannot be reproduced).
 * Philox4x32-7 keyed by the seed, counter = pair index; one call -> four 32-bit words ->
 * two complex samples by Box-Muller.  Every floating-point step is +, -, *, a correctly rounded sqrt or an
 * EXPLICIT fmaf (correctly rounded by definition, one instruction on the GPU) on f32 in a fixed
 * order -- no division, no transcendental libm call -- so CPU oracle and GPU kernel agree bit for
 * bit when both are compiled without implicit contraction.
 * History: round 3 made the polynomial steps fmaf and the quadrant selection branch-free; round 4
 * went from ten Philox rounds to SEVEN (Salmon et al., SC'11, table 2: Philox4x32-7 is the smallest
 * round count of this width that passes BigCrush -- ten is the authors' safety margin, not the
 * reference's choice: the reference draws from rand's StdRng), from ln m = 2 atanh((m-1)/(m+1))
 * with its f32 division to a division-free polynomial in m - 1, and then to the version-3 floating-point
 * stage below (minimax polynomials, no compare or select, a two-step square root).  The sample values
 * changed with each step; the integer stage is pinned for BOTH round counts by the Random123 known answers. ---- */
#define AETH_RNG_ROUNDS 7
AETH_RNG_FN void aeth_philox4x32(int rounds, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                 uint32_t out[4])
{
#if defined(__clang__)
#pragma unroll
#endif
    for (int r = 0; r < rounds; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* the generator's draw: counter = (call, 0), key = seed */
AETH_RNG_FN void aeth_rng_draw(uint64_t call, uint64_t seed, uint32_t out[4])
{
    aeth_philox4x32(AETH_RNG_ROUNDS, (uint32_t)call, (uint32_t)(call >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), out);
}

/* ---- the floating-point stage, version 3 (round 4): 42 operations per sample where version 2 had 64, none of them a
 * compare or a select.  (a, b) = two 32-bit words of one Philox call:
 *     u   = k / 2^24,  k = (a >> 8) | 1  -- an odd 24-bit integer, so u is never 0 or 1 and r^2 > 0
 *     r^2 = -2 ln u,   u = m 2^e with m in [sqrt(.5), sqrt(2)) (integer arithmetic on the bits of (float)k),
 *           f = m - 1,  -2 ln m = -2 f + f^2 + f^3 P(f),  P = degree-6 minimax fit of -2 (ln(1 + f) - f + f^2/2) / f^3
 *           (3e-8 relative), e (-2 ln 2) added last
 *     r   = sqrt(r^2), correctly rounded (see aeth_rng_sqrt)
 *     t   = pi (int32)b / 2^31 in [-pi, pi],  psi = pi/2 - |t| in [-pi/2, pi/2]:
 *           cos t = sin psi = psi + psi^3 S(psi^2),   sin t = sign(t) cos psi = sign(t) (1 + psi^2 C(psi^2)),
 *           S, C = degree-3 minimax fits (5e-9 / 5e-8) -- the fold onto psi needs no quadrant swap
 *     z   = (r cos t, r sin t)
 * Measured (tools/rng_lab.hip, profiles/r04_rng_lab.txt): fill of 2^25 samples 73 -> 50 us.  Accuracy against f64:
 * r 1.1e-7 relative, cos / sin 2-3.5e-7 absolute. ---- */
AETH_RNG_FN float aeth_rng_r2(uint32_t k)                  /* -2 ln(k / 2^24), 1 <= k < 2^24 */
{
    union { float f; uint32_t i; } v; v.f = (float)k;       /* exact: k has 24 bits */
    /* mantissa + (1.0f - 0x3f3504f4) carries into the exponent field exactly when m > 1.41421356f */
    const uint32_t ix = v.i + (0x3f800000u - 0x3f3504f4u);
    const float fe = (float)((int)(ix >> 23) - (127 + 24));
    v.i = (ix & 0x007fffffu) + 0x3f3504f4u;                 /* m in [sqrt(.5), sqrt(2)) */
    const float f = v.f - 1.0f, z = f * f;
    float p = -1.783536927e-01f;
    p = fmaf(p, f, 2.870037524e-01f);
    p = fmaf(p, f, -2.975402362e-01f);
    p = fmaf(p, f, 3.313127281e-01f);
    p = fmaf(p, f, -3.993078812e-01f);
    p = fmaf(p, f, 5.000349384e-01f);
    p = fmaf(p, f, -6.666771630e-01f);
    float y = fmaf(p, f * z, z);
    y = fmaf(-2.0f, f, y);
    return fmaf(fe, -1.38629436f, y);
}

/* correctly rounded square root of x in [1e-7, 34] (every value aeth_rng_r2 returns).  The host calls sqrtf.  The
 * device takes the hardware's reciprocal square root y (1 ulp), s = x y, and one correcting step s + (x - s^2) y/2
 * in two fused multiply-adds; that this is sqrtf(x) bit for bit is CHECKED, not argued: tools/rng_lab.hip runs all
 * 2^23 values of k through both (0 differ), and the parity tests compare the two on every sample they draw. */
AETH_RNG_FN float aeth_rng_sqrt(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_rsqf(x), s = x * y;
    return fmaf(fmaf(-s, s, x), 0.5f * y, s);
#else
    return sqrtf(x);
#endif
}

/* (cos, sin)(pi (int32)b / 2^31) */
AETH_RNG_FN void aeth_rng_cossin(uint32_t b, float *c, float *s)
{
    union { float f; uint32_t i; } t, cs;
    t.f = (float)(int32_t)b * 1.46291808e-09f;              /* pi / 2^31 */
    const float psi = 1.57079633f - fabsf(t.f), z = psi * psi;
    float q = 2.608931464e-06f;
    q = fmaf(q, z, -1.981111218e-04f);
    q = fmaf(q, z, 8.333088159e-03f);
    q = fmaf(q, z, -1.666666046e-01f);
    *c = fmaf(q, psi * z, psi);
    float g = 2.319437319e-05f;
    g = fmaf(g, z, -1.385592655e-03f);
    g = fmaf(g, z, 4.166398936e-02f);
    g = fmaf(g, z, -4.999993229e-01f);
    cs.f = fmaf(g, z, 1.0f);
    cs.i ^= t.i & 0x80000000u;
    *s = cs.f;
}

/* two 32-bit words -> one complex standard normal (unit variance per component) */
AETH_RNG_FN void aeth_rng_normal_pair(uint32_t a, uint32_t b, float *z0, float *z1)
{
    const float r = aeth_rng_sqrt(aeth_rng_r2((a >> 8) | 1u));
    float c, s;
    aeth_rng_cossin(b, &c, &s);
    *z0 = r * c;
    *z1 = r * s;
}

/* complex sample number `idx` of stream (seed): samples 2k and 2k+1 share Philox call k */
AETH_RNG_FN void aeth_rng_cnormal(uint64_t seed, uint64_t idx, float *re, float *im)
{
    uint32_t w[4];
    const uint64_t call = idx >> 1;
    aeth_rng_draw(call, seed, w);
    if (idx & 1) aeth_rng_normal_pair(w[2], w[3], re, im);
    else aeth_rng_normal_pair(w[0], w[1], re, im);
}
