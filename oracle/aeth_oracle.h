/*
 * aeth_oracle.h -- CPU restatement of the aether_primitives hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (include/, the
 * aether_primitives_amd package, libaether_hip.so) may include, link or call
 * this.  Allowed users: tests/, __graft_entry__.smoke(), and the
 * `cpu_baseline` leg of bench.py.
 *
 * Every function cites the reference source (path:line under the upstream
 * razorheadfx/aether_primitives tree) whose behaviour it restates.
 *
 * Parity status:
 *   - element-wise VecOps, Scale, sampling, assert_evm!: pinned by the
 *     reference's own known-answer tests (tests/golden/reference_kat.json).
 *   - FFT arithmetic lives in the third-party crate `rustfft ^3.0`
 *     (Cargo.toml:27; no Cargo.lock => 3.0.x unpinned) which is NOT present in
 *     the reference tree.  The oracle restates the published algorithm class
 *     (mixed-radix decimation-in-time Cooley-Tukey, radix-4 preferred for
 *     powers of two, twiddles computed in f64 and rounded to f32) and is
 *     pinned only by the reference's call sites and golden vectors
 *     (fft.rs:93-117, vecops.rs:443-463).  FFT values on non-DC spectra, the
 *     exponent sign and per-element rounding are "parity unpinned"; the f64
 *     transforms here are the ground truth for those.
 *   - FIR does not exist in the reference (fir.rs:3-22 is a stub); it is
 *     defined from the reference's correlator chain (benches/benches.rs:410-416)
 *     and pinned by direct convolution in f64.
 */
#ifndef AETH_ORACLE_H
#define AETH_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/lib.rs:8-12 -- cf32 = Complex<f32>, repr(C) {re, im} */
typedef struct { float re, im; } orc_cf32;
/* src/lib.rs:14-17 -- cf64 */
typedef struct { double re, im; } orc_cf64;

/* src/fft.rs:6-18 -- enum Scale */
enum { ORC_SCALE_NONE = 0, ORC_SCALE_SN = 1, ORC_SCALE_N = 2, ORC_SCALE_X = 3 };

/* Sign of the DFT exponent.  Reference binding (src/fft.rs:148,150 with
 * rustfft 3.x FFTplanner::new(inverse)): Fft::fwd -> +1, Fft::bwd -> -1. */
enum { ORC_SIGN_REF_FWD = +1, ORC_SIGN_REF_BWD = -1 };

/* ---- element-wise VecOps: src/vecops.rs:94-177 ------------------------- */
void orc_vec_scale(orc_cf32 *x, size_t n, float s);                       /* :94-97   */
int  orc_vec_mul(orc_cf32 *a, size_t na, const orc_cf32 *b, size_t nb);   /* :99-112  */
int  orc_vec_div(orc_cf32 *a, size_t na, const orc_cf32 *b, size_t nb);   /* :114-125 */
void orc_vec_conj(orc_cf32 *x, size_t n);                                 /* :127-130 */
int  orc_vec_add(orc_cf32 *a, size_t na, const orc_cf32 *b, size_t nb);   /* :132-142 */
int  orc_vec_sub(orc_cf32 *a, size_t na, const orc_cf32 *b, size_t nb);   /* :144-155 */
void orc_vec_mirror(orc_cf32 *x, size_t n);                               /* :157-161 */
int  orc_vec_clone(orc_cf32 *a, size_t na, const orc_cf32 *b, size_t nb); /* :163-172 */
void orc_vec_zero(orc_cf32 *x, size_t n);                                 /* :174-177 */

/* ---- Scale: src/fft.rs:22-37 ------------------------------------------ */
float orc_scale_factor(int kind, size_t n, float x);
void  orc_scale_apply(int kind, float x, orc_cf32 *data, size_t n);

/* ---- FFT (restating rustfft's role behind src/fft.rs:147-235) ---------- */
typedef struct orc_fft_plan orc_fft_plan;
orc_fft_plan *orc_fft_plan_create(size_t n);                /* Cfft::with_len, fft.rs:147-158 */
void          orc_fft_plan_destroy(orc_fft_plan *p);
size_t        orc_fft_plan_len(const orc_fft_plan *p);      /* fft.rs:232-234 */
/* unnormalised DFT out[k] = sum_n in[n] exp(sign*2*pi*i*n*k/N); in != out */
void orc_fft_process(orc_fft_plan *p, const orc_cf32 *in, orc_cf32 *out, int sign);
/* Cfft exec variants: copy -> process -> scale (fft.rs:162-230).
 * Return 0, or -1 on "Input and FFT must be the same length". */
int orc_cfft_outofplace(orc_fft_plan *p, const orc_cf32 *in, size_t n_in,
                        orc_cf32 *out, int sign, int scale_kind, float x);   /* fwd/bwd  :162-182 */
int orc_cfft_inplace(orc_fft_plan *p, orc_cf32 *io, size_t n,
                     int sign, int scale_kind, float x);                     /* ifwd/ibwd :184-204 */
const orc_cf32 *orc_cfft_tmp(orc_fft_plan *p, const orc_cf32 *in, size_t n_in,
                             int sign, int scale_kind, float x);             /* tfwd/tbwd :206-230 */

/* f64 ground truth: recursive mixed radix in double (any n), and O(n^2) DFT */
void orc_fft_f64(const orc_cf64 *in, orc_cf64 *out, size_t n, int sign);
void orc_dft_naive_f64(const orc_cf64 *in, orc_cf64 *out, size_t n, int sign);

/* ---- FIR (defined by the build; see header comment) -------------------- */
/* y[n] = sum_{k<ntaps} h[k] x[n-k], zero initial state unless hist != NULL,
 * in which case hist[0..ntaps-2] are x[-(ntaps-1)] .. x[-1]. f64 accumulate. */
void orc_fir_direct_f64(const orc_cf32 *h, size_t ntaps, const orc_cf32 *hist,
                        const orc_cf32 *x, size_t n, orc_cf64 *y);
/* Overlap-save with the reference chain rfft -> vec_mul -> rifft (Scale::N)
 * (benches/benches.rs:410-416), hop L outputs per fft_len block. f32. */
int orc_fir_ols_f32(const orc_cf32 *h, size_t ntaps, size_t fft_len, size_t hop,
                    const orc_cf32 *hist, const orc_cf32 *x, size_t n, orc_cf32 *y);
/* same, blocks spread over `threads` host threads (cpu_baseline only) */
int orc_fir_ols_f32_mt(const orc_cf32 *h, size_t ntaps, size_t fft_len, size_t hop,
                       const orc_cf32 *x, size_t n, orc_cf32 *y, int threads);
/* correlator chain per frame, Scale::None both ways (benches.rs:410-416) */
int orc_correlate_frames(const orc_cf32 *sig_freq, size_t fft_len,
                         orc_cf32 *frames, size_t nframes);

/* ---- sampling: src/sampling.rs ----------------------------------------- */
/* :7-24.  Writes n_src + (n_src-1)*n_between elements to dst (the reference
 * appends to a Vec; the caller owns the offset). compat_im != 0 reproduces
 * the reference's `im: x1.re + i*rate.1` (:19). Returns count, 0 if n_src==0
 * (the reference panics there, :23). */
size_t orc_interpolate(const orc_cf32 *src, size_t n_src, orc_cf32 *dst,
                       size_t n_between, int compat_im);
/* :28-42 / :49-62.  dst[i] = src[i*dec], dec = n_src/n_dst, generic element
 * size.  Returns 0, or -1 when n_src % n_dst != 0 (debug_assert :32-36). */
int orc_downsample(const void *src, size_t n_src, void *dst, size_t n_dst, size_t elem_size);
int orc_downsample_release(const void *src, size_t n_src, void *dst, size_t n_dst, size_t elem_size, int step_by);

/* ---- assert_evm!: src/lib.rs:26-49 ------------------------------------- */
/* Literal macro: returns -1 if every element passes, else index of first
 * failing element; -2 on length mismatch / non-negative limit.  NaN in `act`
 * is reported as a failure (the macro itself silently passes NaN). */
long   orc_assert_evm(const orc_cf32 *act, size_t n_act, const orc_cf32 *ref, size_t n_ref, double evm_limit_db);
/* worst per-element macro-scale value 10*log10(|a-r|/|r|) (dB, -inf if exact) */
double orc_evm_worst_macro_db(const orc_cf32 *act, const orc_cf32 *ref, size_t n);
/* conventional aggregate EVM 20*log10(||act-ref||_2 / ||ref||_2) */
double orc_evm_aggregate_db(const orc_cf32 *act, const orc_cf32 *ref, size_t n);
double orc_evm_aggregate_db_f64ref(const orc_cf32 *act, const orc_cf64 *ref, size_t n);

/* ---- modulation: src/modulation.rs ------------------ */
void orc_qpsk_modulate(const uint8_t *bits, size_t nbits, orc_cf32 *out);        /* :21-24,:87-92,:115-121 */
void orc_qpsk_demod_naive(const orc_cf32 *sym, size_t nsym, uint8_t *bits_out);  /* :33-56 (incl. `idx & 1u8 << 1` quirk) */
/* generic forms: bps = 1 (BPSK, :5-15, demod by the trait default :133-144) or 2 (QPSK);
 * table = NULL -> the generic tables (:77-92); compat = 0 emits (idx >> 1) & 1 for QPSK */
int orc_modulate(const uint8_t *bits, size_t nbits, int bps, const orc_cf32 *table, orc_cf32 *out);
int orc_demod_naive(const orc_cf32 *sym, size_t nsym, int bps, const orc_cf32 *table, int compat, uint8_t *bits_out);

/* ---- noise: src/noise.rs:29-59 around the build's counter-based generator ---- */
/* z = complex standard normal number `idx` of stream `seed` (awgn_restatement.inc) */
void orc_rng_cnormal(uint64_t seed, uint64_t idx, float *re, float *im);
/* Awgn::apply: s[i] += (z * scale) * scale, scale = sqrtf(power) (noise.rs:35,41-42,58) */
void orc_awgn_apply(orc_cf32 *signal, size_t n, float power, uint64_t seed, uint64_t offset);
void orc_awgn_fill(orc_cf32 *target, size_t n, float power, uint64_t seed, uint64_t offset);   /* noise.rs:61-65 */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]);   /* rounds: 7 (the generator's) or 10 */
void orc_rng_normal_pairs(const uint32_t *ab, size_t n, orc_cf32 *out);   /* the Box-Muller stage on explicit word pairs */

/* ---- deterministic synthetic input (the build's own generator) ---------- */
/* complex normal, unit power (sigma = 1/sqrt(2) per component), splitmix64 +
 * Box-Muller in f64, rounded to f32.  Seed 815 = noise.rs:6. */
/* seconds per call of a restated op on one of the reference's criterion shapes (bench.py's CPU-baseline leg) */
double orc_time_shape(int op, size_t n, size_t b, int reps);
void orc_synth_cnormal(uint64_t seed, orc_cf32 *out, size_t n);
/* 64-tap style windowed-sinc low-pass (Hamming, cutoff*fs), unit DC gain */
void orc_synth_lowpass_taps(size_t ntaps, double cutoff, orc_cf32 *taps);

#ifdef __cplusplus
}
#endif
#endif
