//! `extern "C"` declarations: one to one with include/aether_hip.h.
#![allow(non_camel_case_types)]
use aether_primitives::cf32;
use std::os::raw::{c_char, c_float, c_int, c_void};

#[repr(C)] pub struct aeth_ctx { _p: [u8; 0] }
#[repr(C)] pub struct aeth_fft { _p: [u8; 0] }
#[repr(C)] pub struct aeth_fir { _p: [u8; 0] }
#[repr(C)] pub struct aeth_event { _p: [u8; 0] }
#[repr(C)] pub struct aeth_pool { _p: [u8; 0] }
pub const AETH_POOL_ZERO_ON_RETURN: c_int = 1;
/// aeth_pipe_stats: what the three-stage host-stream pipeline reports
#[repr(C)] #[derive(Default, Clone, Copy)]
pub struct aeth_pipe_stats { pub seconds: f64, pub samples: f64, pub chunks: f64, pub pinned: f64 }
/// aeth_vec_step: one link of aeth_vec_chain (op = AETH_VEC_*)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct aeth_vec_step { pub op: c_int, pub other_dev: *const cf32, pub n_other: usize, pub scale: c_float }
pub const AETH_VEC_SCALE: c_int = 0; pub const AETH_VEC_MUL: c_int = 1; pub const AETH_VEC_DIV: c_int = 2; pub const AETH_VEC_CONJ: c_int = 3;
pub const AETH_VEC_ADD: c_int = 4; pub const AETH_VEC_SUB: c_int = 5; pub const AETH_VEC_CLONE: c_int = 6; pub const AETH_VEC_ZERO: c_int = 7;
/// aeth_stream_op: the compute stage of the host pipeline (src/pipeline.rs:24-41 takes a closure; a closure cannot
/// cross the C ABI, so the stage is one of the library's device ops, described field by field as in aether_hip.h)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct aeth_stream_op { pub kind: c_int, pub fir: *mut aeth_fir, pub fft: *mut aeth_fft, pub sig_dev: *const cf32, pub n_sig: usize,
                            pub sign: c_int, pub scale_kind_fwd: c_int, pub x_fwd: c_float, pub scale_kind_bwd: c_int, pub x_bwd: c_float,
                            pub bits_per_symbol: c_int, pub table_host: *const cf32, pub compat: c_int, pub n_between: usize,
                            pub seed: u64, pub offset: u64 }
pub const AETH_STREAM_FIR: c_int = 0;
pub const AETH_STREAM_FFT: c_int = 1;
pub const AETH_STREAM_FFT_MUL_IFFT: c_int = 2;
pub const AETH_STREAM_FFT_MUL_IFFT_DEMOD: c_int = 3;
pub const AETH_STREAM_FFT_INTERPOLATE: c_int = 4;
pub const AETH_STREAM_FIR_DECIM: c_int = 5;
pub const AETH_STREAM_MODULATE_AWGN: c_int = 6;
/// aeth_pipe_util: the same plus the seconds each stage (upload, kernel, download) was active -- the per-stage
/// utilisation report of src/pipeline.rs:89-114
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct aeth_pipe_util { pub seconds: f64, pub samples: f64, pub chunks: f64, pub pinned: f64,
                            pub active_upload: f64, pub active_kernel: f64, pub active_download: f64,
                            pub active_copy_in: f64, pub active_copy_out: f64 }

pub const AETH_OK: c_int = 0;
pub const AETH_E_LEN: c_int = -1;
pub const AETH_SCALE_NONE: c_int = 0;
pub const AETH_SCALE_SN: c_int = 1;
pub const AETH_SCALE_N: c_int = 2;
pub const AETH_SCALE_X: c_int = 3;
/// The ONE place where the reference's method names meet an exponent sign:
/// `Cfft::with_len` plans `fwd` with `FFTplanner::new(true)` (rustfft "inverse", +j)
/// and `bwd` with `FFTplanner::new(false)` (src/fft.rs:148,150).
pub const AETH_SIGN_REF_FWD: c_int = 1;
pub const AETH_SIGN_REF_BWD: c_int = -1;

// cf32 = Complex<f32> is repr(C) {re, im} (src/lib.rs:8-12) == aeth_cf32
extern "C" {
    pub fn aeth_last_error() -> *const c_char;
    pub fn aeth_version() -> c_int;
    pub fn aeth_device_count(count: *mut c_int) -> c_int;
    pub fn aeth_ctx_create(device: c_int, out: *mut *mut aeth_ctx) -> c_int;
    pub fn aeth_ctx_create_on_stream(device: c_int, hip_stream: *mut c_void, out: *mut *mut aeth_ctx) -> c_int;
    pub fn aeth_ctx_stream(ctx: *mut aeth_ctx) -> *mut c_void;
    pub fn aeth_ctx_device(ctx: *const aeth_ctx) -> c_int;
    pub fn aeth_copy_dev(ctx: *mut aeth_ctx, dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
    pub fn aeth_event_create(ctx: *mut aeth_ctx, out: *mut *mut aeth_event) -> c_int;
    pub fn aeth_event_destroy(ev: *mut aeth_event) -> c_int;
    pub fn aeth_event_record(ev: *mut aeth_event) -> c_int;
    pub fn aeth_event_sync(ev: *mut aeth_event) -> c_int;
    pub fn aeth_event_elapsed_ms(start: *mut aeth_event, stop: *mut aeth_event, ms: *mut c_float) -> c_int;
    pub fn aeth_ctx_destroy(ctx: *mut aeth_ctx) -> c_int;
    pub fn aeth_ctx_sync(ctx: *mut aeth_ctx) -> c_int;
    pub fn aeth_ctx_set_overlap(ctx: *mut aeth_ctx, enable: c_int) -> c_int;
    pub fn aeth_ctx_overlap(ctx: *const aeth_ctx) -> c_int;
    pub fn aeth_dev_alloc(ctx: *mut aeth_ctx, bytes: usize, dptr: *mut *mut c_void) -> c_int;
    pub fn aeth_dev_free(ctx: *mut aeth_ctx, dptr: *mut c_void) -> c_int;
    pub fn aeth_upload(ctx: *mut aeth_ctx, dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
    pub fn aeth_download(ctx: *mut aeth_ctx, dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;

    pub fn aeth_vec_scale(ctx: *mut aeth_ctx, s: *mut cf32, n: usize, scale: c_float) -> c_int;
    pub fn aeth_vec_mul(ctx: *mut aeth_ctx, s: *mut cf32, n: usize, o: *const cf32, no: usize) -> c_int;
    pub fn aeth_vec_mul_frames(ctx: *mut aeth_ctx, frames: *mut cf32, frame_len: usize, batch: usize, sig: *const cf32, n_sig: usize) -> c_int;
    pub fn aeth_vec_div(ctx: *mut aeth_ctx, s: *mut cf32, n: usize, o: *const cf32, no: usize) -> c_int;
    pub fn aeth_vec_conj(ctx: *mut aeth_ctx, s: *mut cf32, n: usize) -> c_int;
    pub fn aeth_vec_add(ctx: *mut aeth_ctx, s: *mut cf32, n: usize, o: *const cf32, no: usize) -> c_int;
    pub fn aeth_vec_sub(ctx: *mut aeth_ctx, s: *mut cf32, n: usize, o: *const cf32, no: usize) -> c_int;
    pub fn aeth_vec_mirror(ctx: *mut aeth_ctx, s: *mut cf32, n: usize) -> c_int;
    pub fn aeth_vec_clone(ctx: *mut aeth_ctx, s: *mut cf32, n: usize, o: *const cf32, no: usize) -> c_int;
    pub fn aeth_vec_zero(ctx: *mut aeth_ctx, s: *mut cf32, n: usize) -> c_int;
    pub fn aeth_vec_mirror_frames(ctx: *mut aeth_ctx, s: *mut cf32, frame_len: usize, batch: usize) -> c_int;
    // host-slice flavours: the literal `impl VecOps for [cf32]` semantics, one H2D + D2H per call
    pub fn aeth_host_vec_scale(ctx: *mut aeth_ctx, s: *mut cf32, n: usize, scale: c_float) -> c_int;
    pub fn aeth_host_vec_mul(ctx: *mut aeth_ctx, s: *mut cf32, n: usize, o: *const cf32, no: usize) -> c_int;
    pub fn aeth_host_vec_div(ctx: *mut aeth_ctx, s: *mut cf32, n: usize, o: *const cf32, no: usize) -> c_int;
    pub fn aeth_host_vec_conj(ctx: *mut aeth_ctx, s: *mut cf32, n: usize) -> c_int;
    pub fn aeth_host_vec_add(ctx: *mut aeth_ctx, s: *mut cf32, n: usize, o: *const cf32, no: usize) -> c_int;
    pub fn aeth_host_vec_sub(ctx: *mut aeth_ctx, s: *mut cf32, n: usize, o: *const cf32, no: usize) -> c_int;
    pub fn aeth_host_vec_mirror(ctx: *mut aeth_ctx, s: *mut cf32, n: usize) -> c_int;
    pub fn aeth_host_vec_clone(ctx: *mut aeth_ctx, s: *mut cf32, n: usize, o: *const cf32, no: usize) -> c_int;
    pub fn aeth_host_vec_zero(ctx: *mut aeth_ctx, s: *mut cf32, n: usize) -> c_int;
    pub fn aeth_scale_factor(kind: c_int, n: usize, x: c_float) -> c_float;
    pub fn aeth_scale_apply(ctx: *mut aeth_ctx, kind: c_int, x: c_float, data: *mut cf32, n: usize) -> c_int;

    pub fn aeth_fft_create(ctx: *mut aeth_ctx, len: usize, max_batch: usize, out: *mut *mut aeth_fft) -> c_int;
    pub fn aeth_fft_destroy(plan: *mut aeth_fft) -> c_int;
    pub fn aeth_fft_len(plan: *const aeth_fft) -> usize;
    pub fn aeth_fft_algorithm(plan: *const aeth_fft) -> *const c_char;
    pub fn aeth_fft_exec_tmp(plan: *mut aeth_fft, inp: *const cf32, n_in: usize, batch: usize, sign: c_int,
                             scale_kind: c_int, x: c_float, view: *mut *const cf32) -> c_int;
    pub fn aeth_fft_exec(plan: *mut aeth_fft, inp: *const cf32, n_in: usize, out: *mut cf32, batch: usize,
                         sign: c_int, scale_kind: c_int, x: c_float) -> c_int;
    pub fn aeth_fft_exec_mirrored(plan: *mut aeth_fft, inp: *const cf32, n_in: usize, out: *mut cf32, batch: usize, sign: c_int, kind: c_int, x: c_float) -> c_int;
    pub fn aeth_fft_exec_interpolate(plan: *mut aeth_fft, inp: *const cf32, n_in: usize, batch: usize, sign: c_int, kind: c_int, x: c_float,
                                     dst: *mut cf32, dst_cap: usize, n_between: usize, compat_im: c_int, n_written: *mut usize) -> c_int;
    pub fn aeth_fft_exec_host(plan: *mut aeth_fft, inp: *const cf32, n_in: usize, out: *mut cf32, n_out: usize,
                              sign: c_int, scale_kind: c_int, x: c_float) -> c_int;
    pub fn aeth_fft_exec_tmp_host(plan: *mut aeth_fft, inp: *const cf32, n_in: usize, sign: c_int,
                                  scale_kind: c_int, x: c_float, view: *mut *const cf32) -> c_int;
    pub fn aeth_fft_mul_ifft(plan: *mut aeth_fft, frames: *mut cf32, n_total: usize, batch: usize,
                             sig: *const cf32, n_sig: usize, kf: c_int, xf: c_float, kb: c_int, xb: c_float) -> c_int;
    pub fn aeth_fft_mul_ifft_demod(plan: *mut aeth_fft, frames: *const cf32, n_total: usize, batch: usize, sig: *const cf32, n_sig: usize,
                                   kind_fwd: c_int, x_fwd: c_float, kind_bwd: c_int, x_bwd: c_float, bps: c_int,
                                   table: *const cf32, bits_out: *mut u8, nbits_out: usize, compat: c_int) -> c_int;

    pub fn aeth_fir_create(ctx: *mut aeth_ctx, taps: *const cf32, ntaps: usize, fft_len: usize,
                           out: *mut *mut aeth_fir) -> c_int;
    pub fn aeth_fir_destroy(fir: *mut aeth_fir) -> c_int;
    pub fn aeth_fir_exec(fir: *mut aeth_fir, hist: *const cf32, inp: *const cf32, n: usize, out: *mut cf32) -> c_int;
    pub fn aeth_fir_exec_decim(fir: *mut aeth_fir, hist: *const cf32, inp: *const cf32, n: usize, out: *mut cf32, n_out: usize) -> c_int;
    pub fn aeth_fir_exec_host(fir: *mut aeth_fir, hist: *const cf32, inp: *const cf32, n: usize, out: *mut cf32) -> c_int;
    pub fn aeth_fir_ntaps(fir: *const aeth_fir) -> usize;
    pub fn aeth_fir_fft_len(fir: *const aeth_fir) -> usize;
    pub fn aeth_fir_hop(fir: *const aeth_fir) -> usize;
    pub fn aeth_pool_create(ctx: *mut aeth_ctx, elem_bytes: usize, initial_len: usize, flags: c_int, out: *mut *mut aeth_pool) -> c_int;
    pub fn aeth_pool_destroy(pool: *mut aeth_pool) -> c_int;
    pub fn aeth_pool_take(pool: *mut aeth_pool, buf: *mut *mut c_void) -> c_int;
    pub fn aeth_pool_take_or_make(pool: *mut aeth_pool, buf: *mut *mut c_void) -> c_int;
    pub fn aeth_pool_give_back(pool: *mut aeth_pool, buf: *mut c_void) -> c_int;
    pub fn aeth_pool_len(pool: *mut aeth_pool) -> usize;
    pub fn aeth_pool_cap(pool: *mut aeth_pool) -> usize;
    pub fn aeth_pool_elem_bytes(pool: *const aeth_pool) -> usize;
    pub fn aeth_host_register(ctx: *mut aeth_ctx, ptr: *mut c_void, bytes: usize) -> c_int;
    pub fn aeth_host_unregister(ctx: *mut aeth_ctx, ptr: *mut c_void) -> c_int;
    pub fn aeth_host_is_pinned(ptr: *const c_void, bytes: usize) -> c_int;
    pub fn aeth_vec_fft(ctx: *mut aeth_ctx, x: *mut cf32, n: usize, sign: c_int, kind: c_int, x_scale: c_float) -> c_int;
    pub fn aeth_host_vec_fft(ctx: *mut aeth_ctx, x: *mut cf32, n: usize, sign: c_int, kind: c_int, x_scale: c_float) -> c_int;
    pub fn aeth_vec_chain(ctx: *mut aeth_ctx, self_: *mut cf32, n: usize, steps: *const aeth_vec_step, n_steps: usize) -> c_int;
    pub fn aeth_stream_out_count(ctx: *mut aeth_ctx, op: *const aeth_stream_op, n_in: usize) -> usize;
    pub fn aeth_stream_host(ctx: *mut aeth_ctx, op: *const aeth_stream_op, inp: *const c_void, n_in: usize, out: *mut c_void,
                            n_out: usize, chunk: usize, stats: *mut aeth_pipe_stats) -> c_int;
    pub fn aeth_stream_host_util(ctx: *mut aeth_ctx, op: *const aeth_stream_op, inp: *const c_void, n_in: usize, out: *mut c_void,
                                 n_out: usize, chunk: usize, util: *mut aeth_pipe_util) -> c_int;
    pub fn aeth_stream_chain_out_count(ctx: *mut aeth_ctx, ops: *const aeth_stream_op, n_ops: usize, n_in: usize) -> usize;
    pub fn aeth_stream_host_chain(ctx: *mut aeth_ctx, ops: *const aeth_stream_op, n_ops: usize, inp: *const c_void, n_in: usize,
                                  out: *mut c_void, n_out: usize, chunk: usize, stats: *mut aeth_pipe_stats, util: *mut aeth_pipe_util) -> c_int;
    pub fn aeth_ctx_trim(ctx: *mut aeth_ctx) -> c_int;
    pub fn aeth_test_fail_staging_after(n: c_int);
    pub fn aeth_fir_stream_host(fir: *mut aeth_fir, inp: *const cf32, n: usize, out: *mut cf32, chunk: usize,
                                stats: *mut aeth_pipe_stats) -> c_int;
    pub fn aeth_fir_stream_host_util(fir: *mut aeth_fir, inp: *const cf32, n: usize, out: *mut cf32, chunk: usize,
                                     util: *mut aeth_pipe_util) -> c_int;
    pub fn aeth_stream_file(ctx: *mut aeth_ctx, op: *const aeth_stream_op, in_path: *const c_char, out_path: *const c_char, chunk: usize,
                            stats: *mut aeth_pipe_stats) -> c_int;
    pub fn aeth_fir_stream_file(fir: *mut aeth_fir, in_path: *const c_char, out_path: *const c_char, chunk: usize,
                                stats: *mut aeth_pipe_stats) -> c_int;
    pub fn aeth_file_count_structs(path: *const c_char, elem_size: usize, count: *mut usize) -> c_int;
    pub fn aeth_file_read(path: *const c_char, first_elem: usize, dst: *mut c_void, n: usize, elem_size: usize) -> c_int;
    pub fn aeth_file_write(path: *const c_char, src: *const c_void, n: usize, elem_size: usize, append: c_int) -> c_int;

    pub fn aeth_interpolate(ctx: *mut aeth_ctx, src: *const cf32, n_src: usize, dst: *mut cf32, cap: usize,
                            n_between: usize, compat_im: c_int, n_written: *mut usize) -> c_int;
    pub fn aeth_interpolate_frames(ctx: *mut aeth_ctx, src: *const cf32, frame_len: usize, batch: usize, dst: *mut cf32,
                                   cap: usize, n_between: usize, compat_im: c_int, n_written: *mut usize) -> c_int;
    pub fn aeth_downsample(ctx: *mut aeth_ctx, src: *const c_void, n_src: usize, dst: *mut c_void, n_dst: usize,
                           elem_size: usize) -> c_int;

    pub fn aeth_host_interpolate(ctx: *mut aeth_ctx, src: *const cf32, n_src: usize, dst: *mut cf32, cap: usize,
                                 n_between: usize, compat_im: c_int, n_written: *mut usize) -> c_int;
    pub fn aeth_modulate(ctx: *mut aeth_ctx, bits: *const u8, nbits: usize, bits_per_symbol: c_int,
                         table: *const cf32, out: *mut cf32, n_out: usize) -> c_int;
    pub fn aeth_modulate_awgn(ctx: *mut aeth_ctx, bits: *const u8, nbits: usize, bps: c_int, table: *const cf32, out: *mut cf32,
                              n_out: usize, power: c_float, seed: u64, offset: u64) -> c_int;
    pub fn aeth_demod_naive(ctx: *mut aeth_ctx, sym: *const cf32, nsym: usize, bits_per_symbol: c_int,
                            table: *const cf32, bits_out: *mut u8, nbits_out: usize, compat: c_int) -> c_int;
    pub fn aeth_awgn_apply(ctx: *mut aeth_ctx, signal: *mut cf32, n: usize, power: c_float, seed: u64, offset: u64) -> c_int;
    pub fn aeth_awgn_fill(ctx: *mut aeth_ctx, target: *mut cf32, n: usize, power: c_float, seed: u64, offset: u64) -> c_int;
    pub fn aeth_rng_philox4x32_10(ctx: *mut aeth_ctx, ctr_key: *const u32, n: usize, out: *mut u32) -> c_int;
    pub fn aeth_rng_philox4x32(ctx: *mut aeth_ctx, ctr_key: *const u32, n: usize, rounds: c_int, out: *mut u32) -> c_int;
    pub fn aeth_rng_normal_pairs(ctx: *mut aeth_ctx, ab: *const u32, n: usize, out: *mut aeth_cf32) -> c_int;
    pub fn aeth_host_downsample(ctx: *mut aeth_ctx, src: *const c_void, n_src: usize, dst: *mut c_void,
                                n_dst: usize, elem_size: usize) -> c_int;
    pub fn aeth_downsample_release(ctx: *mut aeth_ctx, src: *const c_void, n_src: usize, dst: *mut c_void, n_dst: usize,
                                   elem_size: usize, step_by: c_int) -> c_int;
    pub fn aeth_host_downsample_release(ctx: *mut aeth_ctx, src: *const c_void, n_src: usize, dst: *mut c_void,
                                        n_dst: usize, elem_size: usize, step_by: c_int) -> c_int;
}

/// Error convention: the reference panics (assert_eq!); the C ABI returns a code and a
/// thread-local message that carries the reference's panic text verbatim.
pub fn check(rc: c_int) {
    if rc != AETH_OK {
        let msg = unsafe { std::ffi::CStr::from_ptr(aeth_last_error()) }.to_string_lossy().into_owned();
        panic!("{}", msg);
    }
}
