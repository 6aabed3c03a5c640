/* The generator's definition, checked without a GPU: the product header (csrc/aeth_rng.h, its host path: sqrtf where
 * the device takes v_rsq_f32 + one step) against the oracle's independent restatement (oracle/awgn_restatement.inc via
 * orc_rng_cnormal), and both against f64 on the accuracy the header claims.  Driven by tests/test_rng_definition.py. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#define AETH_RNG_FN static inline
#include "aeth_rng.h"

void orc_rng_cnormal(uint64_t seed, uint64_t idx, float *re, float *im);

int main(void)
{
    size_t differ = 0;
    const uint64_t seeds[3] = {815, 0, 0xfedcba9876543210ull};
    for (int s = 0; s < 3; s++)
        for (uint64_t i = 0; i < (1u << 21); i++) {
            const uint64_t idx = i + (s == 2 ? (1ull << 40) : 0);
            float a, b, c, d;
            aeth_rng_cnormal(seeds[s], idx, &a, &b);
            orc_rng_cnormal(seeds[s], idx, &c, &d);
            differ += memcmp(&a, &c, 4) != 0 || memcmp(&b, &d, 4) != 0;
        }
    /* accuracy of the two stages against f64: every 64th u, every 4099th angle */
    double e_r = 0, e_cs = 0;
    for (uint32_t k = 1; k < (1u << 24); k += 128) {
        const double t = sqrt(-2.0 * log(k / 16777216.0));
        const double e = fabs(aeth_rng_sqrt(aeth_rng_r2(k)) - t) / t;
        if (e > e_r) e_r = e;
    }
    for (uint64_t b = 0; b < (1ull << 32); b += 4099) {
        float c, s;
        aeth_rng_cossin((uint32_t)b, &c, &s);
        const double th = 3.14159265358979323846 * (double)(int32_t)(uint32_t)b / 2147483648.0;
        const double e = fmax(fabs(c - cos(th)), fabs(s - sin(th)));
        if (e > e_cs) e_cs = e;
    }
    printf("differ %zu radius_rel_err %.3e cossin_abs_err %.3e\n", differ, e_r, e_cs);
    return 0;
}
