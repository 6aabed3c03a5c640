/*
 * aeth_oracle.c -- CPU restatement of the aether_primitives hot path.
 * TEST INFRASTRUCTURE ONLY (see aeth_oracle.h).  Build with
 *   gcc -O2 -ffp-contract=off  (no -ffast-math): Rust never contracts a*b+c,
 * so the f32 arithmetic here must not be fused either.
 */
#define _GNU_SOURCE
#include "aeth_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ======================================================================= */
/* element-wise VecOps                                                      */
/* ======================================================================= */

/* src/vecops.rs:94-97 -- c.scale(s) = (re*s, im*s) */
void orc_vec_scale(orc_cf32 *x, size_t n, float s)
{
    for (size_t i = 0; i < n; i++) { x[i].re = x[i].re * s; x[i].im = x[i].im * s; }
}

/* src/vecops.rs:99-112 -- `*a *= b` (num-complex 0.2 Mul: naive 4 mul 2 add) */
int orc_vec_mul(orc_cf32 *a, size_t na, const orc_cf32 *b, size_t nb)
{
    if (na != nb) return -1;      /* "Vectors must have same length" :100-104 */
    for (size_t i = 0; i < na; i++) {
        float re = a[i].re * b[i].re - a[i].im * b[i].im;
        float im = a[i].re * b[i].im + a[i].im * b[i].re;
        a[i].re = re; a[i].im = im;
    }
    return 0;
}

/* src/vecops.rs:114-125 -- `*a /= b` (num-complex 0.2 Div:
 * norm_sqr = c*c + d*d; re = (a*c + b*d)/norm_sqr; im = (b*c - a*d)/norm_sqr) */
int orc_vec_div(orc_cf32 *a, size_t na, const orc_cf32 *b, size_t nb)
{
    if (na != nb) return -1;
    for (size_t i = 0; i < na; i++) {
        float ns = b[i].re * b[i].re + b[i].im * b[i].im;
        float re = a[i].re * b[i].re + a[i].im * b[i].im;
        float im = a[i].im * b[i].re - a[i].re * b[i].im;
        a[i].re = re / ns; a[i].im = im / ns;
    }
    return 0;
}

/* src/vecops.rs:127-130 */
void orc_vec_conj(orc_cf32 *x, size_t n)
{
    for (size_t i = 0; i < n; i++) x[i].im = -x[i].im;
}

/* src/vecops.rs:132-142 */
int orc_vec_add(orc_cf32 *a, size_t na, const orc_cf32 *b, size_t nb)
{
    if (na != nb) return -1;
    for (size_t i = 0; i < na; i++) { a[i].re = a[i].re + b[i].re; a[i].im = a[i].im + b[i].im; }
    return 0;
}

/* src/vecops.rs:144-155 */
int orc_vec_sub(orc_cf32 *a, size_t na, const orc_cf32 *b, size_t nb)
{
    if (na != nb) return -1;
    for (size_t i = 0; i < na; i++) { a[i].re = a[i].re - b[i].re; a[i].im = a[i].im - b[i].im; }
    return 0;
}

/* src/vecops.rs:157-161 -- swap(x, x+mid), mid = len/2; odd len leaves the last alone */
void orc_vec_mirror(orc_cf32 *x, size_t n)
{
    size_t mid = n / 2;
    for (size_t i = 0; i < mid; i++) { orc_cf32 t = x[i]; x[i] = x[i + mid]; x[i + mid] = t; }
}

/* src/vecops.rs:163-172 */
int orc_vec_clone(orc_cf32 *a, size_t na, const orc_cf32 *b, size_t nb)
{
    if (na != nb) return -1;
    memcpy(a, b, na * sizeof(orc_cf32));
    return 0;
}

/* src/vecops.rs:174-177 */
void orc_vec_zero(orc_cf32 *x, size_t n)
{
    for (size_t i = 0; i < n; i++) { x[i].re = 0.0f; x[i].im = 0.0f; }
}

/* ======================================================================= */
/* Scale: src/fft.rs:22-37                                                  */
/* ======================================================================= */

float orc_scale_factor(int kind, size_t n, float x)
{
    switch (kind) {
    case ORC_SCALE_SN: return 1.0f / sqrtf((float)n);   /* (len as f32).sqrt().recip() :26 */
    case ORC_SCALE_N:  return 1.0f / (float)n;          /* (len as f32).recip()        :30 */
    case ORC_SCALE_X:  return x;                        /* :33-35 */
    default:           return 1.0f;
    }
}

void orc_scale_apply(int kind, float x, orc_cf32 *data, size_t n)
{
    if (kind == ORC_SCALE_NONE) return;                 /* :24 -- no pass at all */
    orc_vec_scale(data, n, orc_scale_factor(kind, n, x));
}

/* ======================================================================= */
/* FFT                                                                      */
/* ======================================================================= */

#define REAL float
#define CPLX orc_cf32
#define NAME(x) f32_##x
#include "fft_template.inc"
#undef REAL
#undef CPLX
#undef NAME

#define REAL double
#define CPLX orc_cf64
#define NAME(x) f64_##x
#include "fft_template.inc"
#undef REAL
#undef CPLX
#undef NAME

/* Cfft: src/fft.rs:134-159 -- plan + tmp of 2*len */
struct orc_fft_plan {
    f32_plan *p;
    orc_cf32 *tmp;     /* 2*len, as Cfft.tmp (fft.rs:141,155) */
    size_t len;
};

orc_fft_plan *orc_fft_plan_create(size_t n)
{
    orc_fft_plan *h = (orc_fft_plan *)calloc(1, sizeof(*h));
    if (!h) return NULL;
    h->p = f32_plan_new(n);
    h->tmp = (orc_cf32 *)calloc(2 * (n ? n : 1), sizeof(orc_cf32));
    h->len = n;
    return h;
}

void orc_fft_plan_destroy(orc_fft_plan *h)
{
    if (!h) return;
    f32_plan_free(h->p);
    free(h->tmp);
    free(h);
}

size_t orc_fft_plan_len(const orc_fft_plan *h) { return h->len; }

void orc_fft_process(orc_fft_plan *h, const orc_cf32 *in, orc_cf32 *out, int sign)
{
    if (h->len == 0) return;
    f32_rec(h->p, 0, h->len, in, 1, out, sign);
}

/* fwd/bwd: fft.rs:162-182 -- tmp[..len] <- input; process(tmp, output); scale(output) */
int orc_cfft_outofplace(orc_fft_plan *h, const orc_cf32 *in, size_t n_in,
                        orc_cf32 *out, int sign, int scale_kind, float x)
{
    if (n_in != h->len) return -1;           /* "Input and FFT must be the same length" */
    memcpy(h->tmp, in, h->len * sizeof(orc_cf32));
    orc_fft_process(h, h->tmp, out, sign);
    orc_scale_apply(scale_kind, x, out, h->len);
    return 0;
}

/* ifwd/ibwd: fft.rs:184-204 */
int orc_cfft_inplace(orc_fft_plan *h, orc_cf32 *io, size_t n, int sign, int scale_kind, float x)
{
    if (n != h->len) return -1;
    memcpy(h->tmp, io, h->len * sizeof(orc_cf32));
    orc_fft_process(h, h->tmp, io, sign);
    orc_scale_apply(scale_kind, x, io, h->len);
    return 0;
}

/* tfwd/tbwd: fft.rs:206-230 -- result lives in tmp[len..] */
const orc_cf32 *orc_cfft_tmp(orc_fft_plan *h, const orc_cf32 *in, size_t n_in,
                             int sign, int scale_kind, float x)
{
    if (n_in != h->len) return NULL;
    memcpy(h->tmp, in, h->len * sizeof(orc_cf32));
    orc_fft_process(h, h->tmp, h->tmp + h->len, sign);
    orc_scale_apply(scale_kind, x, h->tmp + h->len, h->len);
    return h->tmp + h->len;
}

void orc_fft_f64(const orc_cf64 *in, orc_cf64 *out, size_t n, int sign)
{
    if (n == 0) return;
    f64_plan *p = f64_plan_new(n);
    f64_rec(p, 0, n, in, 1, out, sign);
    f64_plan_free(p);
}

void orc_dft_naive_f64(const orc_cf64 *in, orc_cf64 *out, size_t n, int sign)
{
    for (size_t k = 0; k < n; k++) {
        long double sr = 0, si = 0;
        for (size_t j = 0; j < n; j++) {
            size_t e = (size_t)(((unsigned long long)j * k) % n);
            double a = (sign > 0 ? 2.0 : -2.0) * M_PI * (double)e / (double)n;
            double c = cos(a), s = sin(a);
            sr += in[j].re * c - in[j].im * s;
            si += in[j].re * s + in[j].im * c;
        }
        out[k].re = (double)sr; out[k].im = (double)si;
    }
}

/* ======================================================================= */
/* FIR                                                                      */
/* ======================================================================= */

void orc_fir_direct_f64(const orc_cf32 *h, size_t ntaps, const orc_cf32 *hist,
                        const orc_cf32 *x, size_t n, orc_cf64 *y)
{
    for (size_t i = 0; i < n; i++) {
        double sr = 0, si = 0;
        for (size_t k = 0; k < ntaps; k++) {
            double xr, xi;
            if (i >= k) { xr = x[i - k].re; xi = x[i - k].im; }
            else if (hist) {
                /* hist[j] = x[j - (ntaps-1)], j in [0, ntaps-1) */
                size_t back = k - i;                 /* 1..ntaps-1 */
                const orc_cf32 *hv = &hist[(ntaps - 1) - back];
                xr = hv->re; xi = hv->im;
            } else continue;
            double hr = h[k].re, hi = h[k].im;
            sr += hr * xr - hi * xi;
            si += hr * xi + hi * xr;
        }
        y[i].re = sr; y[i].im = si;
    }
}

/* one overlap-save block: the reference's chain rfft -> vec_mul -> rifft
 * (benches/benches.rs:410-416) with Scale::N on the way back */
static void ols_block(orc_fft_plan *plan, const orc_cf32 *Hf, size_t fft_len, orc_cf32 *blk)
{
    orc_cfft_inplace(plan, blk, fft_len, ORC_SIGN_REF_FWD, ORC_SCALE_NONE, 0.0f);
    orc_vec_mul(blk, fft_len, Hf, fft_len);
    orc_cfft_inplace(plan, blk, fft_len, ORC_SIGN_REF_BWD, ORC_SCALE_N, 0.0f);
}

static void ols_fill_block(orc_cf32 *blk, size_t fft_len, size_t ntaps, size_t hop, const orc_cf32 *hist,
                           const orc_cf32 *x, size_t n, size_t out0)
{
    /* the block holds exactly the samples its hop outputs depend on,
     * x[out0-(ntaps-1) .. out0+hop), and zeros elsewhere: a block's result is then a
     * function of those samples alone, whatever stream or shard it is cut from */
    const size_t ov = ntaps - 1;
    for (size_t j = 0; j < fft_len; j++) {
        long long idx = (long long)out0 - (long long)ov + (long long)j;
        if (j >= ov + hop) { blk[j].re = 0; blk[j].im = 0; }
        else if (idx < 0) {
            if (hist) blk[j] = hist[(long long)ov + idx];
            else { blk[j].re = 0; blk[j].im = 0; }
        } else if ((size_t)idx < n) blk[j] = x[idx];
        else { blk[j].re = 0; blk[j].im = 0; }
    }
}

static orc_cf32 *ols_taps_freq(orc_fft_plan *plan, const orc_cf32 *h, size_t ntaps, size_t fft_len)
{
    orc_cf32 *Hf = (orc_cf32 *)calloc(fft_len, sizeof(orc_cf32));
    memcpy(Hf, h, ntaps * sizeof(orc_cf32));
    orc_cfft_inplace(plan, Hf, fft_len, ORC_SIGN_REF_FWD, ORC_SCALE_NONE, 0.0f);
    return Hf;
}

int orc_fir_ols_f32(const orc_cf32 *h, size_t ntaps, size_t fft_len, size_t hop,
                    const orc_cf32 *hist, const orc_cf32 *x, size_t n, orc_cf32 *y)
{
    if (ntaps == 0 || ntaps > fft_len || hop == 0 || hop > fft_len - ntaps + 1) return -1;
    orc_fft_plan *plan = orc_fft_plan_create(fft_len);
    orc_cf32 *Hf = ols_taps_freq(plan, h, ntaps, fft_len);
    orc_cf32 *blk = (orc_cf32 *)malloc(fft_len * sizeof(orc_cf32));
    for (size_t out0 = 0; out0 < n; out0 += hop) {
        ols_fill_block(blk, fft_len, ntaps, hop, hist, x, n, out0);
        ols_block(plan, Hf, fft_len, blk);
        size_t cnt = (n - out0 < hop) ? n - out0 : hop;
        memcpy(y + out0, blk + (ntaps - 1), cnt * sizeof(orc_cf32));
    }
    free(blk); free(Hf);
    orc_fft_plan_destroy(plan);
    return 0;
}

typedef struct {
    const orc_cf32 *h, *x; orc_cf32 *y;
    size_t ntaps, fft_len, hop, n, blk0, blk1;
} ols_job;

static void *ols_worker(void *arg)
{
    ols_job *j = (ols_job *)arg;
    orc_fft_plan *plan = orc_fft_plan_create(j->fft_len);
    orc_cf32 *Hf = ols_taps_freq(plan, j->h, j->ntaps, j->fft_len);
    orc_cf32 *blk = (orc_cf32 *)malloc(j->fft_len * sizeof(orc_cf32));
    for (size_t b = j->blk0; b < j->blk1; b++) {
        size_t out0 = b * j->hop;
        ols_fill_block(blk, j->fft_len, j->ntaps, j->hop, NULL, j->x, j->n, out0);
        ols_block(plan, Hf, j->fft_len, blk);
        size_t cnt = (j->n - out0 < j->hop) ? j->n - out0 : j->hop;
        memcpy(j->y + out0, blk + (j->ntaps - 1), cnt * sizeof(orc_cf32));
    }
    free(blk); free(Hf);
    orc_fft_plan_destroy(plan);
    return NULL;
}

int orc_fir_ols_f32_mt(const orc_cf32 *h, size_t ntaps, size_t fft_len, size_t hop,
                       const orc_cf32 *x, size_t n, orc_cf32 *y, int threads)
{
    if (ntaps == 0 || ntaps > fft_len || hop == 0 || hop > fft_len - ntaps + 1) return -1;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    size_t nblk = (n + hop - 1) / hop;
    pthread_t tid[256];
    ols_job jobs[256];
    for (int t = 0; t < threads; t++) {
        ols_job *j = &jobs[t];
        j->h = h; j->x = x; j->y = y; j->ntaps = ntaps; j->fft_len = fft_len; j->hop = hop; j->n = n;
        j->blk0 = nblk * (size_t)t / (size_t)threads;
        j->blk1 = nblk * (size_t)(t + 1) / (size_t)threads;
        if (threads == 1) ols_worker(j);
        else pthread_create(&tid[t], NULL, ols_worker, j);
    }
    if (threads > 1) for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
    return 0;
}

/* benches/benches.rs:410-416: input.vec_rfft(fft, None).vec_mul(&sig).vec_rifft(fft, None) */
int orc_correlate_frames(const orc_cf32 *sig, size_t fft_len, orc_cf32 *frames, size_t nframes)
{
    orc_fft_plan *plan = orc_fft_plan_create(fft_len);
    for (size_t f = 0; f < nframes; f++) {
        orc_cf32 *blk = frames + f * fft_len;
        orc_cfft_inplace(plan, blk, fft_len, ORC_SIGN_REF_FWD, ORC_SCALE_NONE, 0.0f);
        orc_vec_mul(blk, fft_len, sig, fft_len);
        orc_cfft_inplace(plan, blk, fft_len, ORC_SIGN_REF_BWD, ORC_SCALE_NONE, 0.0f);
    }
    orc_fft_plan_destroy(plan);
    return 0;
}

/* ======================================================================= */
/* sampling: src/sampling.rs                                                */
/* ======================================================================= */

size_t orc_interpolate(const orc_cf32 *src, size_t n_src, orc_cf32 *dst,
                       size_t n_between, int compat_im)
{
    if (n_src == 0) return 0;                       /* reference: unwrap() panics, :23 */
    size_t o = 0;
    const float div = (float)(n_between + 1);       /* (n_between + 1) as f32, :12-13 */
    for (size_t w = 0; w + 1 < n_src; w++) {        /* src.windows(2), :8 */
        orc_cf32 x1 = src[w], x2 = src[w + 1];
        float r0 = (x2.re - x1.re) / div;
        float r1 = (x2.im - x1.im) / div;
        for (size_t i = 0; i <= n_between; i++) {   /* (0..=n_between).map(|i| i as f32), :16 */
            float fi = (float)i;
            dst[o].re = x1.re + fi * r0;            /* :18 */
            dst[o].im = (compat_im ? x1.re : x1.im) + fi * r1;   /* :19 (sic: x1.re) */
            o++;
        }
    }
    dst[o++] = src[n_src - 1];                      /* :23 */
    return o;
}

int orc_downsample(const void *src, size_t n_src, void *dst, size_t n_dst, size_t elem_size)
{
    if (n_dst == 0) return -1;                      /* division by zero panic in the reference */
    if (n_src % n_dst != 0) return -1;              /* debug_assert_eq!, :32-36 */
    size_t dec = n_src / n_dst;                     /* :38 */
    const unsigned char *s = (const unsigned char *)src;
    unsigned char *d = (unsigned char *)dst;
    for (size_t i = 0; i < n_dst; i++)              /* *c = src[i * dec], :39-41 */
        memcpy(d + i * elem_size, s + i * dec * elem_size, elem_size);
    return 0;
}

/* The same functions as a RELEASE build compiles them (`cargo bench`: benches/benches.rs:113,130 runs 8096 -> 512):
 * the debug_assert_eq! of :32-36 / :53-57 is gone, dec floors.  step_by != 0: downsample_sb (:58-61), whose
 * step_by(0) panics when src is shorter than dst. */
int orc_downsample_release(const void *src, size_t n_src, void *dst, size_t n_dst, size_t elem_size, int step_by)
{
    if (n_dst == 0) return -1;                      /* src.len() / dst.len(): division by zero panics in any build */
    size_t dec = n_src / n_dst;                     /* :38 / :58 */
    if (step_by && dec == 0) return -1;             /* step_by(0): "assertion failed: step != 0" */
    const unsigned char *s = (const unsigned char *)src;
    unsigned char *d = (unsigned char *)dst;
    for (size_t i = 0; i < n_dst; i++) {            /* *c = src[i * dec] (:39-41); zip with step_by(dec) (:59-61) */
        if (i * dec >= n_src) return -1;            /* slice index out of bounds (an empty src) */
        memcpy(d + i * elem_size, s + i * dec * elem_size, elem_size);
    }
    return 0;
}

/* ======================================================================= */
/* assert_evm!: src/lib.rs:26-49                                            */
/* ======================================================================= */

/* num-complex norm() = hypot(re, im) */
static float cnormf(float re, float im) { return hypotf(re, im); }

long orc_assert_evm(const orc_cf32 *act, size_t n_act, const orc_cf32 *ref, size_t n_ref, double db)
{
    if (n_act != n_ref) return -2;                 /* :33 */
    if (!(db < 0.0)) return -2;                    /* :34 */
    /* `re.norm() * 10f64.powf(db/10) as f32` -- the cast binds to the powf result, :38 */
    const float lim_fac = (float)pow(10.0, db / 10.0);
    for (size_t i = 0; i < n_act; i++) {
        if (isnan(act[i].re) || isnan(act[i].im)) return (long)i;   /* oracle addition: NaN reject */
        float evm = cnormf(act[i].re - ref[i].re, act[i].im - ref[i].im);   /* :37 */
        float limit = cnormf(ref[i].re, ref[i].im) * lim_fac;              /* :38 */
        if (evm > limit) return (long)i;                                    /* :40 */
    }
    return -1;
}

double orc_evm_worst_macro_db(const orc_cf32 *act, const orc_cf32 *ref, size_t n)
{
    double worst = -INFINITY;
    for (size_t i = 0; i < n; i++) {
        double e = hypot((double)act[i].re - ref[i].re, (double)act[i].im - ref[i].im);
        double r = hypot((double)ref[i].re, (double)ref[i].im);
        if (e == 0.0) continue;
        double v = (r == 0.0) ? INFINITY : 10.0 * log10(e / r);
        if (v > worst) worst = v;
    }
    return worst;
}

double orc_evm_aggregate_db(const orc_cf32 *act, const orc_cf32 *ref, size_t n)
{
    double pe = 0, pr = 0;
    for (size_t i = 0; i < n; i++) {
        double dr = (double)act[i].re - ref[i].re, di = (double)act[i].im - ref[i].im;
        pe += dr * dr + di * di;
        pr += (double)ref[i].re * ref[i].re + (double)ref[i].im * ref[i].im;
    }
    if (pe == 0.0) return -INFINITY;
    if (pr == 0.0) return INFINITY;
    return 10.0 * log10(pe / pr);      /* == 20*log10(rms err / rms ref) */
}

double orc_evm_aggregate_db_f64ref(const orc_cf32 *act, const orc_cf64 *ref, size_t n)
{
    double pe = 0, pr = 0;
    for (size_t i = 0; i < n; i++) {
        double dr = (double)act[i].re - ref[i].re, di = (double)act[i].im - ref[i].im;
        pe += dr * dr + di * di;
        pr += ref[i].re * ref[i].re + ref[i].im * ref[i].im;
    }
    if (pe == 0.0) return -INFINITY;
    if (pr == 0.0) return INFINITY;
    return 10.0 * log10(pe / pr);
}

/* ======================================================================= */
/* modulation: src/modulation.rs                                            */
/* ======================================================================= */

/* GENERIC_QPSK_TABLE :87-92 */
static const orc_cf32 QPSK[4] = { {1.0f, 1.0f}, {-1.0f, 1.0f}, {1.0f, -1.0f}, {-1.0f, -1.0f} };

/* modulate :115-121 with [cf32;4]::index :21-24 -> ((bits[1] << 1) + bits[0]) */
void orc_qpsk_modulate(const uint8_t *bits, size_t nbits, orc_cf32 *out)
{
    size_t ns = nbits / 2;
    for (size_t s = 0; s < ns; s++) {
        unsigned idx = (unsigned)(((bits[2 * s + 1] << 1) + bits[2 * s]) & 0xff);
        out[s] = QPSK[idx & 3];
    }
}

/* demod_naive for [cf32;4] :33-56: min squared distance by min_by (of equal elements the first
 * stays; an unordered pair replaces the running minimum -- see orc_demod_generic); emits idx & 1 and
 * `idx & 1u8 << 1` == idx & 2 (value 0 or 2 -- precedence quirk, :54) */
void orc_qpsk_demod_naive(const orc_cf32 *sym, size_t nsym, uint8_t *bits_out)
{
    for (size_t s = 0; s < nsym; s++) {
        int best = 0; float bd = 0;
        for (int i = 0; i < 4; i++) {
            float dr = sym[s].re - QPSK[i].re, di = sym[s].im - QPSK[i].im;
            float d = dr * dr + di * di;
            if (i == 0 || !(bd <= d)) { best = i; bd = d; }      /* min_by: see orc_demod_generic */
        }
        bits_out[2 * s] = (uint8_t)(best & 1);
        bits_out[2 * s + 1] = (uint8_t)(best & 2);
    }
}

/* GENERIC_BPSK_TABLE :77 */
static const orc_cf32 BPSK[2] = { {1.0f, 1.0f}, {-1.0f, -1.0f} };

/* trait Modulation, DEFAULT methods (modulation.rs:94-149) for a symbol table of 2^bps entries, bps = 3..8 */
static int orc_modulate_generic(const uint8_t *bits, size_t nbits, int bps, const orc_cf32 *table, orc_cf32 *out)
{
    if (!table || nbits % (size_t)bps) return -1;
    for (size_t s = 0; s < nbits / (size_t)bps; s++) {
        size_t idx = 0;
        for (int i = 0; i < bps; i++) idx += ((size_t)bits[s * (size_t)bps + i] % 2) << i;      /* :107-110 */
        out[s] = table[idx];                                                                    /* :115-121 */
    }
    return 0;
}

static int orc_demod_generic(const orc_cf32 *sym, size_t nsym, int bps, const orc_cf32 *table, int compat, uint8_t *bits_out)
{
    if (!table) return -1;
    int ncand = compat ? bps * 2 : (1 << bps);                                                  /* :135 */
    if (ncand > (1 << bps)) ncand = 1 << bps;
    for (size_t s = 0; s < nsym; s++) {
        int best = 0; float bd = 0;
        for (int i = 0; i < ncand; i++) {
            float dr = sym[s].re - table[i].re, di = sym[s].im - table[i].im;                   /* :136 */
            float d = dr * dr + di * di;                                                        /* :137 */
            /* :139-140 min_by(|d, e| d.partial_cmp(e).unwrap_or(Greater)): the running minimum is replaced when the comparison
             * says Greater -- a strictly smaller distance, or an unordered pair (NaN on either side); of equal
             * distances the first stays.  A NaN sample therefore decodes as the LAST candidate. */
            if (i == 0 || !(bd <= d)) { best = i; bd = d; }
        }
        for (int i = 0; i < bps; i++) bits_out[s * (size_t)bps + i] = (uint8_t)((best >> i) & 1);   /* :143 */
    }
    return 0;
}

int orc_modulate(const uint8_t *bits, size_t nbits, int bps, const orc_cf32 *table, orc_cf32 *out)
{
    if (bps >= 3 && bps <= 8) return orc_modulate_generic(bits, nbits, bps, table, out);
    if (bps != 1 && bps != 2) return -1;
    if (nbits % (size_t)bps) return -1;            /* chunks() would hand index() a short chunk */
    const orc_cf32 *t = table ? table : (bps == 1 ? BPSK : QPSK);
    for (size_t s = 0; s < nbits / (size_t)bps; s++) {
        unsigned idx = (bps == 1) ? (bits[s] & 1u)                                   /* :9-12  */
                                  : (((bits[2 * s + 1] & 1u) << 1) + (bits[2 * s] & 1u));   /* :21-24 */
        out[s] = t[idx];                                                             /* :115-121 */
    }
    return 0;
}

int orc_demod_naive(const orc_cf32 *sym, size_t nsym, int bps, const orc_cf32 *table, int compat, uint8_t *bits_out)
{
    if (bps >= 3 && bps <= 8) return orc_demod_generic(sym, nsym, bps, table, compat, bits_out);
    if (bps != 1 && bps != 2) return -1;
    const orc_cf32 *t = table ? table : (bps == 1 ? BPSK : QPSK);
    const int ncand = bps == 1 ? 2 : 4;            /* trait default scans BITS_PER_SYMBOL*2 (:135) = 2 for BPSK */
    for (size_t s = 0; s < nsym; s++) {
        int best = 0; float bd = 0;
        for (int i = 0; i < ncand; i++) {
            float dr = sym[s].re - t[i].re, di = sym[s].im - t[i].im;
            float d = dr * dr + di * di;
            if (i == 0 || !(bd <= d)) { best = i; bd = d; }      /* min_by: see orc_demod_generic */
        }
        if (bps == 1) bits_out[s] = (uint8_t)(best & 1);                             /* :143 */
        else {
            bits_out[2 * s] = (uint8_t)(best & 1);                                   /* :53 */
            bits_out[2 * s + 1] = (uint8_t)(compat ? (best & 2) : ((best >> 1) & 1)); /* :54 */
        }
    }
    return 0;
}

/* ======================================================================= */
/* noise: src/noise.rs                                                      */
/* ======================================================================= */
#include "awgn_restatement.inc"

void orc_rng_cnormal(uint64_t seed, uint64_t idx, float *re, float *im) { aeth_rng_cnormal(seed, idx, re, im); }

void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4])
{
    aeth_philox4x32(rounds, ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { orc_philox4x32(ctr, key, 10, out); }

/* the Box-Muller stage on explicit word pairs: out[i] = normal of (ab[2i], ab[2i+1]) */
void orc_rng_normal_pairs(const uint32_t *ab, size_t n, orc_cf32 *out)
{
    for (size_t i = 0; i < n; i++) aeth_rng_normal_pair(ab[2 * i], ab[2 * i + 1], &out[i].re, &out[i].im);
}

/* Awgn::fill (noise.rs:61-65): push next() until the capacity is reached; next() scales once (noise.rs:39-43) */
void orc_awgn_fill(orc_cf32 *target, size_t n, float power, uint64_t seed, uint64_t offset)
{
    const float scale = sqrtf(power);                           /* noise.rs:35 */
    for (size_t i = 0; i < n; i++) {
        float zr, zi;
        aeth_rng_cnormal(seed, offset + i, &zr, &zi);
        target[i].re = zr * scale;                              /* noise.rs:41 */
        target[i].im = zi * scale;                              /* noise.rs:42 */
    }
}

void orc_awgn_apply(orc_cf32 *signal, size_t n, float power, uint64_t seed, uint64_t offset)
{
    const float scale = sqrtf(power);                           /* noise.rs:35 */
    for (size_t i = 0; i < n; i++) {
        float zr, zi;
        aeth_rng_cnormal(seed, offset + i, &zr, &zi);
        signal[i].re = signal[i].re + (zr * scale) * scale;     /* noise.rs:41 then :58 */
        signal[i].im = signal[i].im + (zi * scale) * scale;     /* noise.rs:42 then :58 */
    }
}

/* ======================================================================= */
/* timing of the reference's criterion shapes (benches/benches.rs) for bench.py's CPU-baseline leg:     */
/* seconds per iteration of ONE call of the restated op on the shape, buffers set up outside the clock  */
/* as criterion's iter_with_setup does.  op: 0 vec_mul(n) :37, 1 vec_scale(n) :48, 2 vec_clone(n) :59,  */
/* 3 interpolate(n, n_between = b) :76-92, 4 downsample release build (n -> b) :99-113, 5 Cfft ifwd SN  */
/* (n) :294-310, 6 Cfft fwd SN copy (n) :340-357, 7 correlator chain rfft -> vec_mul -> rifft (n)        */
/* :388-420, 8 qpsk modulate (n bits) :210-223, 9 qpsk demod_naive (n symbols) :245-260                   */
/* ======================================================================= */
#include <time.h>
static double orc_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

double orc_time_shape(int op, size_t n, size_t b, int reps)
{
    if (n == 0 || reps < 1) return -1.0;
    const size_t big = (op == 3 ? n + (n - 1) * b : n) + 16;
    orc_cf32 *x = (orc_cf32 *)malloc(big * sizeof(orc_cf32)), *y = (orc_cf32 *)malloc(big * sizeof(orc_cf32));
    uint8_t *bits = (uint8_t *)malloc(2 * n + 16);
    if (!x || !y || !bits) { free(x); free(y); free(bits); return -1.0; }
    for (size_t i = 0; i < big; i++) { x[i].re = 1.0f; x[i].im = 1.0f; y[i].re = 1.0f; y[i].im = 1.0f; }
    for (size_t i = 0; i < 2 * n; i++) bits[i] = (uint8_t)((i * 2654435761u >> 13) & 1u);
    orc_fft_plan *plan = (op >= 5 && op <= 7) ? orc_fft_plan_create(n) : NULL;
    volatile float sink = 0.f;
    double t0 = orc_now();
    for (int r = 0; r < reps; r++) {
        switch (op) {
        case 0: orc_vec_mul(x, n, y, n); break;
        case 1: orc_vec_scale(x, n, 1.0f); break;
        case 2: orc_vec_clone(x, n, y, n); break;
        case 3: sink += (float)orc_interpolate(x, n, y, b, 1); break;
        case 4: orc_downsample_release(x, n, y, b, sizeof(orc_cf32), 0); break;
        case 5: orc_cfft_inplace(plan, x, n, ORC_SIGN_REF_FWD, ORC_SCALE_SN, 0.0f); break;
        case 6: orc_cfft_outofplace(plan, x, n, y, ORC_SIGN_REF_FWD, ORC_SCALE_SN, 0.0f); break;
        case 7:
            orc_cfft_inplace(plan, x, n, ORC_SIGN_REF_FWD, ORC_SCALE_NONE, 0.0f);
            orc_vec_mul(x, n, y, n);
            orc_cfft_inplace(plan, x, n, ORC_SIGN_REF_BWD, ORC_SCALE_NONE, 0.0f);
            if (!(x[0].re < 1e30f && x[0].re > -1e30f)) for (size_t i = 0; i < n; i++) { x[i].re = 1.0f; x[i].im = 1.0f; }   /* keep the data finite */
            break;
        case 8: orc_qpsk_modulate(bits, n, y); break;
        case 9: orc_qpsk_demod_naive(x, n, bits); break;
        default: break;
        }
        sink += x[r % n].re + y[r % n].im;
    }
    const double el = orc_now() - t0;
    (void)sink;
    if (plan) orc_fft_plan_destroy(plan);
    free(x); free(y); free(bits);
    return el / (double)reps;
}

/* ======================================================================= */
/* synthetic input                                                          */
/* ======================================================================= */

static uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_synth_cnormal(uint64_t seed, orc_cf32 *out, size_t n)
{
    uint64_t s = seed;
    const double sigma = 0.70710678118654752440;
    for (size_t i = 0; i < n; i++) {
        double u1 = ((double)(splitmix64(&s) >> 11) + 1.0) * (1.0 / 9007199254740993.0);
        double u2 = (double)(splitmix64(&s) >> 11) * (1.0 / 9007199254740992.0);
        double r = sqrt(-2.0 * log(u1)) * sigma;
        out[i].re = (float)(r * cos(2.0 * M_PI * u2));
        out[i].im = (float)(r * sin(2.0 * M_PI * u2));
    }
}

void orc_synth_lowpass_taps(size_t ntaps, double cutoff, orc_cf32 *taps)
{
    double sum = 0;
    double *t = (double *)malloc(sizeof(double) * ntaps);
    for (size_t k = 0; k < ntaps; k++) {
        double m = (double)k - (double)(ntaps - 1) / 2.0;
        double sinc = (m == 0.0) ? 2.0 * cutoff : sin(2.0 * M_PI * cutoff * m) / (M_PI * m);
        double w = ntaps > 1 ? 0.54 - 0.46 * cos(2.0 * M_PI * (double)k / (double)(ntaps - 1)) : 1.0;   /* one tap: no window */
        t[k] = sinc * w; sum += t[k];
    }
    for (size_t k = 0; k < ntaps; k++) { taps[k].re = (float)(t[k] / sum); taps[k].im = 0.0f; }
    free(t);
}
