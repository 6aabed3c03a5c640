/*
 * aether_hip.h -- C ABI of the MI355X (gfx950) backend for the cf32 hot path of
 * razorheadfx/aether_primitives: VecOps / Fft / FIR / sampling.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.
 * The reference is a Rust crate; a maintainer binds these symbols from an
 * `extern "C"` block and implements the crate's own traits on top (rust/ in
 * this repo holds that binding; INTEGRATION.md walks through it).  Each entry
 * point cites the reference interface (path:line in the upstream tree) it
 * replaces.
 *
 * Conventions
 *  - aeth_cf32 is bit-identical to the crate's cf32 = Complex<f32>, repr(C)
 *    (src/lib.rs:8-12) and to HIP float2.
 *  - Every function returns AETH_OK (0) or a negative AETH_E_* code and never
 *    unwinds.  aeth_last_error() returns a thread-local message.  The
 *    reference's convention is panic-on-misuse (assert_eq!, e.g.
 *    src/vecops.rs:100-104, src/fft.rs:163-167); the binding re-raises
 *    AETH_E_LEN as a panic with the reference's message text, which
 *    aeth_last_error() carries verbatim.
 *  - "dev" functions take DEVICE pointers, are ordered on the context's HIP
 *    stream and return without waiting (aeth_ctx_sync to wait).
 *    "host" functions take HOST slices, stage through the context's pinned
 *    buffers and return when the result is back in the caller's slice --
 *    the literal one-frame-per-call trait semantics (PCIe-bound; use the dev
 *    flavour on the hot path).
 *  - A context/plan is not thread-safe (every reference method takes
 *    `&mut self`), but may move between threads; distinct contexts are
 *    independent (one device + one stream each).
 *  - Buffers: device pointers must be 8-byte aligned (AETH_E_ALIGN otherwise);
 *    16-byte alignment enables the widest loads.
 */
#ifndef AETHER_HIP_H
#define AETHER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AETH_API __attribute__((visibility("default")))

/* src/lib.rs:8-12 */
typedef struct { float re, im; } aeth_cf32;

typedef struct aeth_ctx aeth_ctx;       /* device + stream + staging buffers      */
typedef struct aeth_fft aeth_fft;       /* replaces Cfft (src/fft.rs:134-159)     */
typedef struct aeth_fir aeth_fir;       /* gives Fir<T> (src/fir.rs:3-22) a body  */
typedef struct aeth_event aeth_event;   /* hipEvent on the context's stream       */
typedef struct aeth_pool aeth_pool;     /* replaces Pool<T> (src/pool.rs:71-160) for pinned host buffers */

enum {
    AETH_OK = 0,
    AETH_E_LEN = -1,          /* the reference's assert_eq! length panics          */
    AETH_E_ARG = -2,          /* null pointer, bad enum, zero length where illegal */
    AETH_E_ALIGN = -3,        /* device pointer not 8-byte aligned                  */
    AETH_E_HIP = -4,          /* a HIP runtime call failed (message has details)   */
    AETH_E_NOMEM = -5,
    AETH_E_UNSUPPORTED = -6   /* e.g. an FFT length the backend cannot plan        */
};

/* enum Scale, src/fft.rs:6-18 */
enum { AETH_SCALE_NONE = 0, AETH_SCALE_SN = 1, AETH_SCALE_N = 2, AETH_SCALE_X = 3 };

/* Sign of the DFT exponent: out[k] = sum_n in[n] * exp(sign * 2*pi*i*n*k/N).
 * The reference plans Fft::fwd with FFTplanner::new(true) and Fft::bwd with
 * FFTplanner::new(false) (src/fft.rs:148,150); in rustfft 3.x the argument is
 * `inverse`, so fwd carries the +j exponent.  The binding maps
 * fwd/ifwd/tfwd -> AETH_SIGN_REF_FWD and bwd/ibwd/tbwd -> AETH_SIGN_REF_BWD in
 * exactly one place. */
enum { AETH_SIGN_REF_FWD = +1, AETH_SIGN_REF_BWD = -1 };

/* Tuning knobs.  The library reads these environment integers ONLY in a process started with AETH_TUNING=1; each
 * chooses between behaviours that ship (measured routes and shapes that were not kept are not in the library at all:
 * csrc/aeth_internal.h, lab_int):
 *   AETH_FIR_GRID_FIRST    sixteenths of the resident grid for a fused-FIR launch that starts a chain or runs alone
 *                          with the overlap lane on (default 16)
 *   AETH_FIR_GRID_CHAINED  the same for a launch that runs beside its predecessor (default 12); values above 16
 *                          oversubscribe: the workgroups without a slot start as earlier ones finish (measured
 *                          20 / 24 / 32: within noise of 12, profiles/r04_k20_trace.txt)
 *   AETH_FIR_SPREAD        0 / 1: force the burst / spread form of the next-window prefetch (default: spread for a
 *                          lone launch, burst beside another)
 *   AETH_NT                0 / 1: force plain / non-temporal accesses on streamed operands (default: by size)
 *   AETH_PIPE_THREADS      host copy threads of the stream pipeline (default: 3/8 of the cores the process may use -- affinity mask and
 *                          cgroup CPU quota --, 2..12)
 *   AETH_PIPE_MIXED        1: a pageable input next to a pinned output downloads directly into the caller's memory
 *                          (default 0: both sides staged -- the mixed form measured 2.4 x slower downloads)
 *   AETH_SYNC_SPIN_US      how long aeth_ctx_sync polls both queues before it blocks (default 2000) */
AETH_API const char *aeth_last_error(void);
AETH_API int aeth_version(void);                     /* 0x00MMmmpp */
AETH_API int aeth_device_count(int *count);

/* ---- context ----------------------------------------------------------- */
AETH_API int aeth_ctx_create(int device, aeth_ctx **out);
/* borrow an existing hipStream_t (e.g. torch's current stream); not destroyed */
AETH_API int aeth_ctx_create_on_stream(int device, void *hip_stream, aeth_ctx **out);
AETH_API int aeth_ctx_destroy(aeth_ctx *ctx);
AETH_API int aeth_ctx_sync(aeth_ctx *ctx);
/* Overlap lane (no reference counterpart: the reference is synchronous; this is the device analogue of keeping
 * two stages of src/pipeline.rs:52-137 busy).  When enabled, consecutive aeth_fir_exec calls whose buffers do not
 * touch each other alternate between two HIP queues of the context, so that the end of one launch overlaps the
 * start of the next; every other call (and aeth_ctx_sync / aeth_event_record) is ordered behind both queues, so
 * results are those of one in-order stream.  Off by default; refused for contexts on a borrowed stream. */
AETH_API int aeth_ctx_set_overlap(aeth_ctx *ctx, int enable);
AETH_API int aeth_ctx_overlap(const aeth_ctx *ctx);   /* 1 if enabled AND in use (0 while aeth_ctx_stream has parked it) */
/* hipStream_t of the context, for interop (torch ExternalStream, the caller's own copies and kernels).  Handing it out
 * parks the overlap lane: every later launch stays on this stream, so work the caller enqueues on it is ordered behind
 * the library's, until the caller re-arms the lane with aeth_ctx_set_overlap(ctx, 1). */
AETH_API void *aeth_ctx_stream(aeth_ctx *ctx);
AETH_API int aeth_ctx_device(const aeth_ctx *ctx);

/* ---- device memory (caller-owned; the library never keeps host pointers) -- */
AETH_API int aeth_dev_alloc(aeth_ctx *ctx, size_t bytes, void **dptr);
AETH_API int aeth_dev_free(aeth_ctx *ctx, void *dptr);
AETH_API int aeth_upload(aeth_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);   /* waits */
AETH_API int aeth_download(aeth_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes); /* waits */
AETH_API int aeth_copy_dev(aeth_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);  /* async */

/* ---- timing on the context's stream (bench harness) ---------------------- */
AETH_API int aeth_event_create(aeth_ctx *ctx, aeth_event **out);
AETH_API int aeth_event_destroy(aeth_event *ev);
AETH_API int aeth_event_record(aeth_event *ev);
AETH_API int aeth_event_sync(aeth_event *ev);
AETH_API int aeth_event_elapsed_ms(aeth_event *start, aeth_event *stop, float *ms);

/* ---- VecOps, device flavour: trait VecOps, src/vecops.rs:39-89 ------------ */
/* All in place on `self_`.  Binary ops take both lengths so the reference's
 * "Vectors must have same length" assert (e.g. :100-104) lives on this side. */
AETH_API int aeth_vec_scale (aeth_ctx *ctx, aeth_cf32 *self_, size_t n, float scale);          /* :94-97   */
AETH_API int aeth_vec_mul   (aeth_ctx *ctx, aeth_cf32 *self_, size_t n, const aeth_cf32 *other, size_t n_other); /* :99-112  */
AETH_API int aeth_vec_div   (aeth_ctx *ctx, aeth_cf32 *self_, size_t n, const aeth_cf32 *other, size_t n_other); /* :114-125 */
AETH_API int aeth_vec_conj  (aeth_ctx *ctx, aeth_cf32 *self_, size_t n);                       /* :127-130 */
AETH_API int aeth_vec_add   (aeth_ctx *ctx, aeth_cf32 *self_, size_t n, const aeth_cf32 *other, size_t n_other); /* :132-142 */
AETH_API int aeth_vec_sub   (aeth_ctx *ctx, aeth_cf32 *self_, size_t n, const aeth_cf32 *other, size_t n_other); /* :144-155 */
AETH_API int aeth_vec_mirror(aeth_ctx *ctx, aeth_cf32 *self_, size_t n);                       /* :157-161 */
AETH_API int aeth_vec_clone (aeth_ctx *ctx, aeth_cf32 *self_, size_t n, const aeth_cf32 *other, size_t n_other); /* :163-172 */
AETH_API int aeth_vec_zero  (aeth_ctx *ctx, aeth_cf32 *self_, size_t n);                       /* :174-177 */
/* vec_mirror applied to each of `batch` consecutive frames of `frame_len`
 * (the `chunks_mut(fft_len).for_each(|c| c.vec_rfft(..).vec_mirror())` idiom,
 * src/util/plot.rs:59-61) */
AETH_API int aeth_vec_mirror_frames(aeth_ctx *ctx, aeth_cf32 *self_, size_t frame_len, size_t batch);
/* vec_mul with ONE right-hand side shared by `batch` consecutive frames of `frame_len` (frames[f][j] *= sig[j]):
 * the middle step of `c.vec_rfft(..).vec_mul(&sig).vec_rifft(..)` over chunks_mut(fft_len)
 * (benches/benches.rs:410-416); n_sig != frame_len -> AETH_E_LEN with the reference's text (:100-104) */
AETH_API int aeth_vec_mul_frames(aeth_ctx *ctx, aeth_cf32 *frames, size_t frame_len, size_t batch,
                                 const aeth_cf32 *sig, size_t n_sig);
/* vec_mutate (:179-182) takes a Rust closure and stays on the host side of the binding. */
/* A CHAIN of the element-wise methods above in one pass over memory: `v.vec_add(&a).vec_mul(&b).vec_conj()` (the
 * reference's chaining, src/vecops.rs:12-38; BASELINE config 1) reads `self_` once, every binary link's operand once,
 * and writes `self_` once -- one launch and 32 B per sample for that chain instead of three launches and 64 B.  Each
 * link is its own kernel's arithmetic, rounded link by link: bit-identical to the separate calls.  Per link the checks
 * of its own entry point ("Vectors must have same length"); an operand that overlaps self_ is refused (AETH_E_ARG).
 * vec_mirror (a permutation) and vec_mutate (a closure) are not links.  Any number of steps (8 per pass). */
enum { AETH_VEC_SCALE = 0, AETH_VEC_MUL = 1, AETH_VEC_DIV = 2, AETH_VEC_CONJ = 3, AETH_VEC_ADD = 4, AETH_VEC_SUB = 5,
       AETH_VEC_CLONE = 6, AETH_VEC_ZERO = 7 };
typedef struct aeth_vec_step { int op; const aeth_cf32 *other_dev; size_t n_other; float scale; } aeth_vec_step;
AETH_API int aeth_vec_chain(aeth_ctx *ctx, aeth_cf32 *self_, size_t n, const aeth_vec_step *steps, size_t n_steps);

/* ---- VecOps, host-slice flavour (synchronous) ----------------------------- */
AETH_API int aeth_host_vec_scale (aeth_ctx *ctx, aeth_cf32 *self_, size_t n, float scale);
AETH_API int aeth_host_vec_mul   (aeth_ctx *ctx, aeth_cf32 *self_, size_t n, const aeth_cf32 *other, size_t n_other);
AETH_API int aeth_host_vec_div   (aeth_ctx *ctx, aeth_cf32 *self_, size_t n, const aeth_cf32 *other, size_t n_other);
AETH_API int aeth_host_vec_conj  (aeth_ctx *ctx, aeth_cf32 *self_, size_t n);
AETH_API int aeth_host_vec_add   (aeth_ctx *ctx, aeth_cf32 *self_, size_t n, const aeth_cf32 *other, size_t n_other);
AETH_API int aeth_host_vec_sub   (aeth_ctx *ctx, aeth_cf32 *self_, size_t n, const aeth_cf32 *other, size_t n_other);
AETH_API int aeth_host_vec_mirror(aeth_ctx *ctx, aeth_cf32 *self_, size_t n);
AETH_API int aeth_host_vec_clone (aeth_ctx *ctx, aeth_cf32 *self_, size_t n, const aeth_cf32 *other, size_t n_other);
AETH_API int aeth_host_vec_zero  (aeth_ctx *ctx, aeth_cf32 *self_, size_t n);

/* ---- Scale: src/fft.rs:22-37 ---------------------------------------------- */
/* factor exactly as the reference computes it in f32: SN -> (n as f32).sqrt().recip(),
 * N -> (n as f32).recip(), X -> x, None -> 1 (and no pass at all). */
AETH_API float aeth_scale_factor(int scale_kind, size_t n, float x);
AETH_API int aeth_scale_apply(aeth_ctx *ctx, int scale_kind, float x, aeth_cf32 *data_dev, size_t n);

/* ---- Fft: trait Fft src/fft.rs:48-77, struct Cfft :134-235 ----------------- */
/* Cfft::with_len(len) (:147-158).  max_batch sizes the plan's device scratch
 * (>= 2*len*max_batch, mirroring Cfft.tmp :141,155); exec grows it on demand. */
AETH_API int aeth_fft_create(aeth_ctx *ctx, size_t len, size_t max_batch, aeth_fft **out);
AETH_API int aeth_fft_destroy(aeth_fft *plan);
AETH_API size_t aeth_fft_len(const aeth_fft *plan);                                   /* Fft::len :232-234 */
/* text name of the kernel path chosen for this length ("stockham_pow2", ...) */
AETH_API const char *aeth_fft_algorithm(const aeth_fft *plan);
/* `batch` frames of len() each, device pointers, out == in => in-place
 * (fwd/bwd :162-182, ifwd/ibwd :184-204).  n_in is the TOTAL element count of
 * `in` and must equal batch*len ("Input and FFT must be the same length"). */
AETH_API int aeth_fft_exec(aeth_fft *plan, const aeth_cf32 *in, size_t n_in, aeth_cf32 *out,
                           size_t batch, int sign, int scale_kind, float x);
/* The same followed by vec_mirror on every frame (`chunks_mut(fft_len).for_each(|c| c.vec_rfft(&mut fft, s)
 * .vec_mirror())`, src/util/plot.rs:59-61): for the register-resident power-of-two lengths the swap of the halves
 * is folded into the transform's store addresses (no second pass over memory), other lengths run the two steps. */
AETH_API int aeth_fft_exec_mirrored(aeth_fft *plan, const aeth_cf32 *in_dev, size_t n_in, aeth_cf32 *out_dev,
                                    size_t batch, int sign, int scale_kind, float x);
/* Per frame: the transform, then sampling::interpolate(&frame, &mut dst, n_between) (src/sampling.rs:7-24) -- BASELINE
 * config 5's chain in one call.  dst receives batch frames of len + (len-1)*n_between samples; `in` is not modified.
 * The spectrum goes through the plan's temp (Cfft.tmp); bit-identical to aeth_fft_exec + aeth_interpolate_frames. */
AETH_API int aeth_fft_exec_interpolate(aeth_fft *plan, const aeth_cf32 *in_dev, size_t n_in, size_t batch, int sign,
                                       int scale_kind, float x, aeth_cf32 *dst_dev, size_t dst_cap, size_t n_between,
                                       int compat_im, size_t *n_written);
/* VecOps::vec_fft / vec_ifft (src/vecops.rs:184-196): in place on a whole slice with a plan of its length.  The reference
 * builds a fresh Cfft on every call; the context keeps the plans these calls have built (the eight most recently used
 * lengths, 2^24 points in total at most; aeth_ctx_trim frees them), so a repeated length plans once.  sign: AETH_SIGN_REF_FWD for vec_fft, _BWD for vec_ifft. */
AETH_API int aeth_vec_fft(aeth_ctx *ctx, aeth_cf32 *x_dev, size_t n, int sign, int scale_kind, float x);
AETH_API int aeth_host_vec_fft(aeth_ctx *ctx, aeth_cf32 *x_host, size_t n, int sign, int scale_kind, float x);
/* host slices, one frame per call: the literal trait methods. out may equal in. */
AETH_API int aeth_fft_exec_host(aeth_fft *plan, const aeth_cf32 *in, size_t n_in,
                                aeth_cf32 *out, size_t n_out, int sign, int scale_kind, float x);
/* tfwd/tbwd (:206-230): transform into the plan's internal temp and lend it.
 * *view points to host memory valid until the next call on this plan. */
AETH_API int aeth_fft_exec_tmp_host(aeth_fft *plan, const aeth_cf32 *in, size_t n_in,
                                    int sign, int scale_kind, float x, const aeth_cf32 **view);
/* device flavour of the same: *view_dev is the plan's device temp */
AETH_API int aeth_fft_exec_tmp(aeth_fft *plan, const aeth_cf32 *in, size_t n_in, size_t batch,
                               int sign, int scale_kind, float x, const aeth_cf32 **view_dev);

/* ---- frequency-domain multiply chain: benches/benches.rs:410-416 ----------- */
/* frames.vec_rfft(fft, s_fwd).vec_mul(sig).vec_rifft(fft, s_bwd) per frame, fused into
 * one kernel (the "correlator inplace" benchmark; also BASELINE config 4).
 * `sig_dev` has fft_len elements.  In place on `frames_dev`. */
AETH_API int aeth_fft_mul_ifft(aeth_fft *plan, aeth_cf32 *frames_dev, size_t n_total, size_t batch,
                               const aeth_cf32 *sig_dev, size_t n_sig,
                               int scale_kind_fwd, float x_fwd, int scale_kind_bwd, float x_bwd);
/* The same chain followed by Modulation::demod_naive on its output (examples/modem.rs:28-31; BASELINE config 4's
 * receive side): only the bit bytes are written, `frames` is not modified.  BPSK / QPSK (table_host NULL = the generic
 * tables, `compat` as in aeth_demod_naive).  For plan lengths 1024 .. 4096 the decisions are taken in the chain's
 * registers; other lengths run the two steps through the plan's temp.  Same bits as aeth_fft_mul_ifft +
 * aeth_demod_naive. */
AETH_API int aeth_fft_mul_ifft_demod(aeth_fft *plan, const aeth_cf32 *frames_dev, size_t n_total, size_t batch,
                                     const aeth_cf32 *sig_dev, size_t n_sig, int scale_kind_fwd, float x_fwd,
                                     int scale_kind_bwd, float x_bwd, int bits_per_symbol,
                                     const aeth_cf32 *table_host, uint8_t *bits_out_dev, size_t nbits_out, int compat);

/* ---- FIR (src/fir.rs:3-22 holds taps + scratch but no filter method) ------- */
/* y[n] = sum_{k<ntaps} taps[k] * x[n-k] by overlap-save built from the
 * reference's own chain rfft -> vec_mul -> rifft(Scale::N).  taps are host
 * memory, copied.  fft_len must be a supported power of two >= 2*ntaps. */
AETH_API int aeth_fir_create(aeth_ctx *ctx, const aeth_cf32 *taps_host, size_t ntaps,
                             size_t fft_len, aeth_fir **out);
AETH_API int aeth_fir_destroy(aeth_fir *fir);
AETH_API size_t aeth_fir_ntaps(const aeth_fir *fir);
AETH_API size_t aeth_fir_fft_len(const aeth_fir *fir);
AETH_API size_t aeth_fir_hop(const aeth_fir *fir);      /* outputs per block (<= fft_len-ntaps+1) */
/* n outputs for n inputs.  hist_dev: NULL => zero initial state, else the
 * ntaps-1 samples preceding in_dev[0] (x[-(ntaps-1)] .. x[-1]).  The output range must not touch the input range
 * or the history (AETH_E_ARG): blocks run concurrently and read windows that reach into their neighbours', so ANY
 * overlap -- not only out == in -- would read samples already overwritten.  (Rust's &[T] / &mut [T] make that
 * unrepresentable on the reference side; the C ABI has to refuse it.) */
AETH_API int aeth_fir_exec(aeth_fir *fir, const aeth_cf32 *hist_dev, const aeth_cf32 *in_dev,
                           size_t n, aeth_cf32 *out_dev);
/* The filter followed by sampling::downsample (src/sampling.rs:28-42) in one pass: out[i] = y[i * dec] with
 * dec = n / n_out.  n % n_out != 0 -> AETH_E_ARG "Only even decimations are supported" (:32-36).  Bit-identical
 * to aeth_fir_exec + aeth_downsample; the output write traffic drops by dec.  fft_len 1024 .. 4096. */
AETH_API int aeth_fir_exec_decim(aeth_fir *fir, const aeth_cf32 *hist_dev, const aeth_cf32 *in_dev, size_t n,
                                 aeth_cf32 *out_dev, size_t n_out);
AETH_API int aeth_fir_exec_host(aeth_fir *fir, const aeth_cf32 *hist_host, const aeth_cf32 *in_host,
                                size_t n, aeth_cf32 *out_host);

/* ---- pinned host buffers: src/pool.rs:43-221 -------------------------------------------------- */
/* The reference's object pool ("useful for large buffers and other time expensive objects", :9-10) with pinned
 * (hipHostMalloc) elements of elem_bytes each: the maker allocates one element, the resetter optionally zeroes it.
 * Samples produced INTO pool elements (file reads, receivers, generators) cross PCIe with true asynchronous copies
 * and no staging; the library never page-locks memory it did not allocate unless asked to (aeth_host_register).
 * Thread-safe like the reference's Arc<Mutex<..>> (take / give_back from any thread). */
enum { AETH_POOL_ZERO_ON_RETURN = 1 };                                  /* resetter: memset 0 (pool.rs:47,175-178) */
AETH_API int aeth_pool_create(aeth_ctx *ctx, size_t elem_bytes, size_t initial_len, int flags, aeth_pool **out); /* pool::make :43-69 */
AETH_API int aeth_pool_destroy(aeth_pool *pool);              /* refused (AETH_E_ARG) while elements are checked out   */
AETH_API int aeth_pool_take(aeth_pool *pool, void **buf);     /* Pool::take :78-97: *buf = NULL when the pool is empty */
AETH_API int aeth_pool_take_or_make(aeth_pool *pool, void **buf);       /* Pool::take_or_make :115-132 (grows the pool) */
AETH_API int aeth_pool_give_back(aeth_pool *pool, void *buf); /* Elem::drop -> give_back :175-208                      */
AETH_API size_t aeth_pool_len(aeth_pool *pool);               /* Pool::len :138-140: elements currently checked in     */
AETH_API size_t aeth_pool_cap(aeth_pool *pool);               /* Pool::cap :157-159: elements the pool owns            */
AETH_API size_t aeth_pool_elem_bytes(const aeth_pool *pool);
/* Explicit opt-in for memory the caller owns (no reference counterpart): page-lock [ptr, ptr + bytes) so that the
 * host pipeline copies from / to it directly.  The range must start on a page boundary and cover whole pages
 * (AETH_E_ALIGN) and must not touch a pool element or a range registered before (AETH_E_ARG); every runtime return
 * code is checked.  The caller keeps it alive and mapped until aeth_host_unregister has returned AETH_OK. */
AETH_API int aeth_host_register(aeth_ctx *ctx, void *ptr, size_t bytes);
AETH_API int aeth_host_unregister(aeth_ctx *ctx, void *ptr);
AETH_API int aeth_host_is_pinned(const void *ptr, size_t bytes);        /* 1: inside one pool element / registered range */

/* Host-resident stream through the device at PCIe rate (SURVEY 8f "next" #4): chunks through five stages -- copy-in
 * (caller slice -> pinned pool element, host threads) | upload | compute | download (three HIP streams, three device
 * slots handed on by events) | copy-out (host threads) -- so that both copy engines run back to back: the counterpart
 * of the reference's thread-per-stage pipeline over pooled buffers (src/pipeline.rs:52-137, src/pool.rs:43-221).
 *
 * The reference's pipeline takes any closure as a stage (src/pipeline.rs:24-41 `add_stage<F: FnMut(O) -> U>`,
 * :123-137 `new`); a closure cannot cross this boundary (as for vec_mutate), so the compute stage is described by an
 * aeth_stream_op: one of the library's device ops applied chunk by chunk.  Input and output differ per op:
 *   AETH_STREAM_FIR                 fir                      n cf32 in -> n cf32 out (hop-aligned chunks; = aeth_fir_exec)
 *   AETH_STREAM_FFT                 fft, sign, scale_kind_fwd / x_fwd          frames of len cf32 -> the same (= aeth_fft_exec)
 *   AETH_STREAM_FFT_MUL_IFFT        fft, sig_dev, both scales                  frames -> frames (= aeth_fft_mul_ifft)
 *   AETH_STREAM_FFT_MUL_IFFT_DEMOD  ... + bits_per_symbol, table_host, compat  8 B in -> bits_per_symbol BYTES out per sample
 *   AETH_STREAM_FFT_INTERPOLATE     fft, sign, scale, n_between, compat (= compat_im)   len in -> len + (len-1)*n_between out per frame
 *   AETH_STREAM_MODULATE_AWGN       bits_per_symbol, table_host, x_fwd = noise power, seed, offset   BIT BYTES in -> symbols out (= aeth_modulate_awgn;
 *                                   the one op whose input is not cf32: in_host holds n_in bytes; in a chain it has to be the first stage)
 *   AETH_STREAM_FIR_DECIM           fir, n_between = dec (must divide aeth_fir_hop and n_in)   n in -> n / dec out (= aeth_fir_exec_decim)
 * Every op's output is bit-identical to its device flavour on the whole slice.  n_in counts input samples, n_out
 * output ELEMENTS of the op's type and must equal aeth_stream_out_count(ctx, op, n_in) (AETH_E_LEN otherwise; the frame
 * ops need whole frames: "Input and FFT must be the same length").  The two host ranges must not overlap (AETH_E_ARG).
 *
 * A side that is already page-locked (aeth_host_is_pinned: a pool element, a registered range) skips its host stage
 * and is copied from / to directly; caller memory is never registered by these calls.  stats->pinned = 1 * (in direct)
 * + 2 * (out direct); an input that needs staging takes the output through the host stage as well (measured: the
 * mixed form is the slow one; AETH_PIPE_MIXED below).  chunk_samples = 0 picks 32 MiB on the larger side per chunk
 * (an eighth of a short stream, at least 1 MiB), rounded to whole hops / frames.
 *
 * What a context keeps between calls: the three stage streams, three device slots per side, three pinned staging
 * elements per side (only for pageable caller memory) and the copy threads, sized by the largest chunk seen.  A slot is
 * at most 64 MiB (a larger chunk_samples is split internally; one frame of more than that still gets its slot), so the
 * retained memory is bounded by 6 x 64 MiB of device memory and 6 x 64 MiB of pinned memory; aeth_ctx_trim gives all
 * of it back (and the device scratch of the host-slice flavours), aeth_ctx_destroy does the same. */
typedef struct { double seconds, samples, chunks, pinned; } aeth_pipe_stats;
enum { AETH_STREAM_FIR = 0, AETH_STREAM_FFT = 1, AETH_STREAM_FFT_MUL_IFFT = 2, AETH_STREAM_FFT_MUL_IFFT_DEMOD = 3,
       AETH_STREAM_FFT_INTERPOLATE = 4, AETH_STREAM_FIR_DECIM = 5, AETH_STREAM_MODULATE_AWGN = 6 };
typedef struct aeth_stream_op {
    int kind;                          /* AETH_STREAM_*                                                          */
    aeth_fir *fir;                     /* FIR                                                                    */
    aeth_fft *fft;                     /* the frame ops: frames of aeth_fft_len(fft) samples                     */
    const aeth_cf32 *sig_dev;          /* MUL_IFFT, MUL_IFFT_DEMOD: the multiplier, DEVICE memory, n_sig = len   */
    size_t n_sig;
    int sign;                          /* FFT, FFT_INTERPOLATE: AETH_SIGN_REF_FWD / _BWD                         */
    int scale_kind_fwd; float x_fwd;   /* Scale of the (forward) transform                                       */
    int scale_kind_bwd; float x_bwd;   /* MUL_IFFT, MUL_IFFT_DEMOD: Scale of the way back                        */
    int bits_per_symbol;               /* MUL_IFFT_DEMOD: 1 (BPSK) or 2 (QPSK)                                   */
    const aeth_cf32 *table_host;       /*   symbol table (host), NULL = the generic tables                       */
    int compat;                        /*   as aeth_demod_naive; FFT_INTERPOLATE: compat_im of aeth_interpolate  */
    size_t n_between;                  /* FFT_INTERPOLATE; FIR_DECIM: the decimation                             */
    uint64_t seed, offset;             /* MODULATE_AWGN: noise stream and the position of the first symbol in it  */
} aeth_stream_op;
AETH_API size_t aeth_stream_out_count(aeth_ctx *ctx, const aeth_stream_op *op, size_t n_in);   /* 0 for a bad op */
AETH_API int aeth_stream_host(aeth_ctx *ctx, const aeth_stream_op *op, const void *in_host, size_t n_in,
                              void *out_host, size_t n_out, size_t chunk_samples, aeth_pipe_stats *stats);
/* The same run with the per-stage report of the reference's pipeline (src/pipeline.rs:89-114: items processed, rate and
 * "Utilisation" = time active / time elapsed, per stage): seconds each of the stages was busy -- upload, compute, download
 * from timed events around every stage operation, the two host stages from the wall clock between hand-over and completion
 * of each chunk.  Utilisation of a stage = active_x / seconds. */
typedef struct { double seconds, samples, chunks, pinned, active_upload, active_kernel, active_download,
                 active_copy_in, active_copy_out; /* the two host stages (0 for a side copied directly) */ } aeth_pipe_util;
AETH_API int aeth_stream_host_util(aeth_ctx *ctx, const aeth_stream_op *op, const void *in_host, size_t n_in,
                                   void *out_host, size_t n_out, size_t chunk_samples, aeth_pipe_util *util);
/* Several ops as ONE compute stage -- `pipeline::new(..).add_stage(a).add_stage(b)` (src/pipeline.rs:24-41): stage i's output
 * is stage i + 1's input on the device, only the first stage sees host data and only the last one's output goes back (so
 * e.g. FIR -> FFT frames -> correlate + demod moves 8 B up and 2 B down per sample).  1 .. 8 ops; a filter can only be the
 * first stage, a stage that emits bits only the last; n_in has to be a whole number of the chain's granule (the smallest
 * count every stage takes in whole hops / frames).  Bit-identical to the ops' device flavours applied one after the
 * other.  stats and util may each be NULL. */
AETH_API size_t aeth_stream_chain_out_count(aeth_ctx *ctx, const aeth_stream_op *ops, size_t n_ops, size_t n_in);
AETH_API int aeth_stream_host_chain(aeth_ctx *ctx, const aeth_stream_op *ops, size_t n_ops, const void *in_host, size_t n_in,
                                    void *out_host, size_t n_out, size_t chunk_samples, aeth_pipe_stats *stats,
                                    aeth_pipe_util *util);
/* AETH_STREAM_FIR with the filter as the only argument (output bit-identical to aeth_fir_exec_host on the whole slice) */
AETH_API int aeth_fir_stream_host(aeth_fir *fir, const aeth_cf32 *in_host, size_t n, aeth_cf32 *out_host,
                                  size_t chunk_samples, aeth_pipe_stats *stats);
AETH_API int aeth_fir_stream_host_util(aeth_fir *fir, const aeth_cf32 *in_host, size_t n, aeth_cf32 *out_host,
                                       size_t chunk_samples, aeth_pipe_util *util);
/* Releases what the context retains between calls (see above); the next call that needs it creates it again. */
AETH_API int aeth_ctx_trim(aeth_ctx *ctx);
/* Test support (tests/test_gpu_pool.py): the n-th pinned staging element the pipeline takes from now on fails right
 * after it has been taken (once), so that the give-back of every error path can be checked. */
AETH_API void aeth_test_fail_staging_after(int n);

/* ---- raw sample files (SURVEY 8f "next" #3): src/util/file.rs:12-107 ------------------------ */
/* The reference's binary files are header-less native-endian dumps of back-to-back structs;
 * for cf32 that is exactly the byte layout of a device buffer. */
AETH_API int aeth_file_count_structs(const char *path, size_t elem_size, size_t *count);            /* :12-25  */
AETH_API int aeth_file_read(const char *path, size_t offset_structs, void *dst_host, size_t n, size_t elem_size);  /* BinaryReader::read :46-57 */
AETH_API int aeth_file_write(const char *path, const void *src_host, size_t n, size_t elem_size, int append);      /* binary_writer + write :83-109 */
/* raw cf32 file -> one of the pipeline's ops -> raw file of its output type (cf32, or bit bytes for the demodulating stage)
 * through the pipeline above (both files mapped; the two paths must not name the same file) */
AETH_API int aeth_stream_file(aeth_ctx *ctx, const aeth_stream_op *op, const char *in_path, const char *out_path,
                              size_t chunk_samples, aeth_pipe_stats *stats);
/* raw cf32 file -> FIR -> raw cf32 file */
AETH_API int aeth_fir_stream_file(aeth_fir *fir, const char *in_path, const char *out_path, size_t chunk_samples,
                                  aeth_pipe_stats *stats);

/* ---- sampling: src/sampling.rs ---------------------------------------------- */
/* interpolate (:7-24): writes n_src + (n_src-1)*n_between elements to dst
 * (the Rust wrapper reserves that much spare Vec capacity, passes its end, then
 * set_len's: the reference APPENDS, :17,:23).  compat_im != 0 reproduces
 * `im: x1.re + i*rate.1` (:19).  n_src == 0 -> AETH_E_LEN (reference panics, :23). */
AETH_API int aeth_interpolate(aeth_ctx *ctx, const aeth_cf32 *src_dev, size_t n_src,
                              aeth_cf32 *dst_dev, size_t dst_capacity, size_t n_between,
                              int compat_im, size_t *n_written);
/* the same for `batch` independent frames of frame_len (BASELINE config 5) */
AETH_API int aeth_interpolate_frames(aeth_ctx *ctx, const aeth_cf32 *src_dev, size_t frame_len,
                                     size_t batch, aeth_cf32 *dst_dev, size_t dst_capacity,
                                     size_t n_between, int compat_im, size_t *n_written);
AETH_API int aeth_host_interpolate(aeth_ctx *ctx, const aeth_cf32 *src, size_t n_src,
                                   aeth_cf32 *dst, size_t dst_capacity, size_t n_between,
                                   int compat_im, size_t *n_written);
/* downsample / downsample_sb (:28-42 / :49-62): dst[i] = src[i*(n_src/n_dst)],
 * generic T: Copy via elem_size (1,2,4,8,16 bytes).  n_src % n_dst != 0 ->
 * AETH_E_LEN with "Only even decimations are supported" (:32-36). */
AETH_API int aeth_downsample(aeth_ctx *ctx, const void *src_dev, size_t n_src,
                             void *dst_dev, size_t n_dst, size_t elem_size);
AETH_API int aeth_host_downsample(aeth_ctx *ctx, const void *src, size_t n_src,
                                  void *dst, size_t n_dst, size_t elem_size);
/* The two entry points above are the reference as `cargo build` / `cargo test` compile it (DEBUG build): the
 * divisibility check of :32-36 is a debug_assert_eq! and panics there.  The two below are the reference as
 * `cargo build --release` / `cargo bench` compile it (RELEASE build, the one its own benchmark runs at
 * benches/benches.rs:113,130 with the shape 8096 -> 512): the assert is compiled out, dec = n_src / n_dst FLOORS and
 * dst[i] = src[i*dec] for every i < n_dst (:38-41).  What still panics in a release build is an error here too
 * (AETH_E_LEN): n_dst == 0 (division by zero), n_src == 0 (src[0] out of bounds); n_src < n_dst gives dec = 0, which
 * downsample runs as dst[i] = src[0] and downsample_sb (step_by != 0) refuses, as step_by(0) panics (:58-61). */
AETH_API int aeth_downsample_release(aeth_ctx *ctx, const void *src_dev, size_t n_src,
                                     void *dst_dev, size_t n_dst, size_t elem_size, int step_by);
AETH_API int aeth_host_downsample_release(aeth_ctx *ctx, const void *src, size_t n_src,
                                          void *dst, size_t n_dst, size_t elem_size, int step_by);

/* ---- modulation (SURVEY 8f "next" #1): src/modulation.rs --------------------------- */
/* Modulation::modulate (:115-121): one symbol per `bits_per_symbol` input bytes (each byte is
 * one bit, taken modulo 2 as the trait's default index() does, :107); index =
 * bits[0] for BPSK (:9-12), (bits[1] << 1) + bits[0] for QPSK (:21-24); out[s] = table[index].
 * table_host: 2 or 4 symbols, NULL = GENERIC_BPSK_TABLE / GENERIC_QPSK_TABLE (:77-92).
 * nbits must be a multiple of bits_per_symbol (AETH_E_LEN) and n_out == nbits / bits_per_symbol. */
AETH_API int aeth_modulate(aeth_ctx *ctx, const uint8_t *bits_dev, size_t nbits, int bits_per_symbol,
                           const aeth_cf32 *table_host, aeth_cf32 *out_dev, size_t n_out);
/* modulate followed by Awgn::apply on the fresh symbols (examples/modem.rs:19-26) in one pass over memory:
 * out[s] = table[index] + (z * scale) * scale with z = position offset + s of stream `seed` (see aeth_awgn_apply).
 * BPSK / QPSK tables; bit-identical to aeth_modulate + aeth_awgn_apply. */
AETH_API int aeth_modulate_awgn(aeth_ctx *ctx, const uint8_t *bits_dev, size_t nbits, int bits_per_symbol,
                                const aeth_cf32 *table_host, aeth_cf32 *out_dev, size_t n_out, float power,
                                uint64_t seed, uint64_t offset);
/* Modulation::demod_naive: nearest table symbol by squared distance, folded exactly as the reference's
 * min_by(|d, e| d.partial_cmp(e).unwrap_or(Ordering::Greater)) (:46, :139): of equal distances the FIRST stays, a
 * strictly smaller one replaces it, and so does an unordered pair -- a sample with a NaN component decodes as the
 * last candidate scanned.  compat != 0 reproduces the QPSK specialisation's
 * output exactly (:33-56): it pushes `idx & 1` and `idx & 1u8 << 1` == idx & 2, i.e. the
 * second bit comes out as 0 or 2; compat == 0 emits (idx >> 1) & 1.  BPSK follows the
 * trait default (:133-144).  bits_out_dev receives nsym * bits_per_symbol bytes. */
AETH_API int aeth_demod_naive(aeth_ctx *ctx, const aeth_cf32 *sym_dev, size_t nsym, int bits_per_symbol,
                              const aeth_cf32 *table_host, uint8_t *bits_out_dev, size_t nbits_out, int compat);

/* ---- noise (SURVEY 8f "next" #1): src/noise.rs ------------------------------------- */
/* Awgn::apply (:53-59) on a device-resident signal: s[i] += next().scale(scale) with
 * scale = sqrt(power) (:35) and next() = (z.re * scale, z.im * scale) (:39-43) -- the
 * reference scales twice, so the noise amplitude is proportional to `power`; reproduced.
 * z comes from the library's own counter-based generator (Philox4x32-7 + Box-Muller,
 * position `offset + i` of stream `seed`; DEFAULT seed of the reference is 815, :6): the
 * reference's StdRng stream is not reproducible, results agree bit for bit with the CPU
 * restatement of THIS generator only.  Consecutive calls continue a stream by passing
 * offset += n, as the reference's generator object would. */
AETH_API int aeth_awgn_apply(aeth_ctx *ctx, aeth_cf32 *signal_dev, size_t n, float power, uint64_t seed,
                             uint64_t offset);
/* Awgn::fill / Awgn::iter (src/noise.rs:61-84): target[i] = next() = (z.re * scale, z.im * scale) -- scaled
 * ONCE, unlike apply -- for positions offset .. offset+n of stream `seed`.  The reference fills a Vec up to its
 * capacity; here the capacity is `n`. */
AETH_API int aeth_awgn_fill(aeth_ctx *ctx, aeth_cf32 *target_dev, size_t n, float power, uint64_t seed,
                            uint64_t offset);
/* The generator's integer stage by itself (replaces rand::StdRng, src/noise.rs:2-4,23,33): out[i][0..3] =
 * Philox4x32-R of counter ctr_key[i][0..3] under key ctr_key[i][4..5] (Salmon et al., SC'11), R = 7 (what the
 * generator draws with since round 4: the smallest Crush-resistant round count of this width) or 10; the Random123
 * known-answer vectors of both are in tests/golden/philox4x32_{7,10}_kat.json.  Device pointers, n x 6 and n x 4 words. */
AETH_API int aeth_rng_philox4x32(aeth_ctx *ctx, const uint32_t *ctr_key_dev, size_t n, int rounds, uint32_t *out_dev);
AETH_API int aeth_rng_philox4x32_10(aeth_ctx *ctx, const uint32_t *ctr_key_dev, size_t n, uint32_t *out_dev);
/* The generator's floating-point stage by itself (replaces rand_distr::Normal, src/noise.rs:3,39-43): out[i] = the
 * complex standard normal the generator makes of the 32-bit word pair (ab[i][0], ab[i][1]) -- radius from the first
 * word (u = ((a >> 8) | 1) / 2^24, r = sqrt(-2 ln u)), angle from the second.  Exists so that the stage can be pinned
 * over its WHOLE radius argument (all 2^24 values of a >> 8: the device takes r from v_rsq_f32 plus one correcting
 * step, the oracle from sqrtf) instead of on the samples a stream happens to draw; tests/test_gpu_modulation.py does
 * that.  Device pointers, n x 2 words in, n samples out. */
AETH_API int aeth_rng_normal_pairs(aeth_ctx *ctx, const uint32_t *ab_dev, size_t n, aeth_cf32 *out_dev);

#ifdef __cplusplus
}
#endif
#endif /* AETHER_HIP_H */
