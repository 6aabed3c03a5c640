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
 * two complex samples by Box-Muller.  Every floating-point step is +, -, *, sqrt or an
 * EXPLICIT fmaf (correctly rounded by definition, one instruction on the GPU) on f32 in a fixed
 * order -- no division, no transcendental libm call -- so CPU oracle and GPU kernel agree bit for
 * bit when both are compiled without implicit contraction.
 * History: round 3 made the polynomial steps fmaf and the quadrant selection branch-free; round 4
 * went from ten Philox rounds to SEVEN (Salmon et al., SC'11, table 2: Philox4x32-7 is the smallest
 * round count of this width that passes BigCrush -- ten is the authors' safety margin, not the
 * reference's choice: the reference draws from rand's StdRng) and from ln m = 2 atanh((m-1)/(m+1))
 * with its f32 division to a division-free polynomial in m - 1.  The sample values changed with
 * each step; the integer stage is pinned for BOTH round counts by the Random123 known answers. ---- */
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

/* natural log of u in (0, 1], ~1 ulp: u = m * 2^e, m in [sqrt(.5), sqrt(2)), f = m - 1,
 *     ln m = f - f^2/2 + f^3 P(f),   P of degree 8 (the single-precision coefficients of Cephes' logf, S. Moshier),
 *     ln u = ln m + e ln 2 with ln 2 split into 0.693359375 - 2.12194440e-4 so that e * hi is exact.
 * No division: the atanh form this replaces spent a third of its instructions on (m - 1) / (m + 1). */
AETH_RNG_FN float aeth_rng_log(float u)
{
    union { float f; uint32_t i; } v; v.f = u;
    int e = (int)((v.i >> 23) & 0xff) - 127;
    v.i = (v.i & 0x007fffffu) | 0x3f800000u;               /* m in [1, 2) */
    float m = v.f;
    const int hi = m > 1.41421356f;                         /* fold [sqrt 2, 2) onto [sqrt .5, 1) */
    m = hi ? m * 0.5f : m;
    e += hi;
    const float f = m - 1.0f, z = f * f, fe = (float)e;
    float p = 7.0376836292e-2f;
    p = fmaf(p, f, -1.1514610310e-1f);
    p = fmaf(p, f, 1.1676998740e-1f);
    p = fmaf(p, f, -1.2420140846e-1f);
    p = fmaf(p, f, 1.4249322787e-1f);
    p = fmaf(p, f, -1.6668057665e-1f);
    p = fmaf(p, f, 2.0000714765e-1f);
    p = fmaf(p, f, -2.4999993993e-1f);
    p = fmaf(p, f, 3.3333331174e-1f);
    float y = (p * f) * z;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    return fmaf(fe, 0.693359375f, f + y);
}

/* (cos, sin)(2 pi w / 2^24), w a 24-bit integer: quadrant by the top two bits, then
 * Taylor polynomials of sin/cos(pi/2 * t), t in [0, 1) */
AETH_RNG_FN void aeth_rng_cossin(uint32_t w24, float *c, float *s)
{
    const uint32_t q = (w24 >> 22) & 3u;
    const float t = (float)(w24 & 0x3fffffu) * (1.0f / 4194304.0f);
    const float x = t * 1.57079633f, x2 = x * x;
    float sp = -2.50521084e-08f;                            /* -1/11! */
    sp = fmaf(sp, x2, 2.75573192e-06f);
    sp = fmaf(sp, x2, -1.98412698e-04f);
    sp = fmaf(sp, x2, 8.33333333e-03f);
    sp = fmaf(sp, x2, -1.66666667e-01f);
    sp = fmaf(sp, x2, 1.0f);
    sp = sp * x;
    float cp = -2.75573192e-07f;                            /* -1/10! */
    cp = fmaf(cp, x2, 2.48015873e-05f);
    cp = fmaf(cp, x2, -1.38888889e-03f);
    cp = fmaf(cp, x2, 4.16666667e-02f);
    cp = fmaf(cp, x2, -0.5f);
    cp = fmaf(cp, x2, 1.0f);
    /* quadrant: (cp, sp), (-sp, cp), (-cp, -sp), (sp, -cp) -- a select and a sign-bit flip each, no branches */
    union { float f; uint32_t i; } a, b;
    a.f = (q & 1u) ? sp : cp;
    b.f = (q & 1u) ? cp : sp;
    a.i ^= (((q + 1u) >> 1) & 1u) << 31;                    /* cos is negative in quadrants 1 and 2 */
    b.i ^= (q >> 1) << 31;                                  /* sin in quadrants 2 and 3 */
    *c = a.f; *s = b.f;
}

/* two 32-bit words -> one complex standard normal (unit variance per component) */
AETH_RNG_FN void aeth_rng_normal_pair(uint32_t a, uint32_t b, float *z0, float *z1)
{
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);     /* (0, 1] */
    const float r = sqrtf(-2.0f * aeth_rng_log(u1));
    float c, s;
    aeth_rng_cossin(b >> 8, &c, &s);
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
