//! aether_hip -- MI355X backend behind aether_primitives' own traits.
//!
//! * `HipFft` implements `aether_primitives::fft::Fft` (src/fft.rs:48-77), so it plugs
//!   into `vec_rfft` / `vec_rifft` (generic over `impl Fft`, src/vecops.rs:83,88) exactly
//!   where `Cfft` does: `data.vec_rfft(&mut hip_fft, Scale::SN)`.
//! * `DeviceVec` keeps samples in HBM and offers the `VecOps` method set with the same
//!   names and chaining (`&mut self -> &mut Self`), one kernel launch per call.
//! * `sampling::{interpolate, downsample}` have the reference's signatures.
//!
//! Not compiled in this pipeline (no Rust toolchain): see Cargo.toml.
pub mod ffi;

use aether_primitives::cf32;
use aether_primitives::fft::{Fft, Scale};
use ffi::*;
use std::marker::PhantomData;
use std::os::raw::{c_int, c_void};
use std::ptr;

fn scale_args(s: Scale) -> (i32, f32) {
    match s {
        Scale::None => (AETH_SCALE_NONE, 0.0),
        Scale::SN => (AETH_SCALE_SN, 0.0),
        Scale::N => (AETH_SCALE_N, 0.0),
        Scale::X(x) => (AETH_SCALE_X, x),
    }
}

/// One device + one HIP stream.  `Send` (may move between pipeline threads like a
/// pooled `Cfft`, src/pool.rs:69-71), not `Sync` (every method takes `&mut self`).
///
/// Device objects (`DeviceVec`, plans) borrow the context SHARED, so several can live at once; the
/// context itself is `!Sync` (raw pointer), which keeps all of them on one thread at a time as the
/// C side requires.
pub struct Context { h: *mut aeth_ctx }
unsafe impl Send for Context {}
impl Context {
    pub fn new(device: i32) -> Context {
        let mut h = ptr::null_mut();
        check(unsafe { aeth_ctx_create(device, &mut h) });
        Context { h }
    }
    pub fn sync(&self) { check(unsafe { aeth_ctx_sync(self.h) }) }
    /// Consecutive independent `Fir::filter` launches alternate between two HIP queues.
    pub fn set_overlap(&self, enable: bool) { check(unsafe { aeth_ctx_set_overlap(self.h, enable as c_int) }) }
}
impl Drop for Context { fn drop(&mut self) { unsafe { aeth_ctx_destroy(self.h); } } }

/// Replaces `Cfft` (src/fft.rs:134-235).  Borrows its `Context` for `'c`: the plan's Drop (and every call)
/// goes through the context's stream, so the context must outlive it -- the borrow checker enforces what the
/// C side requires.  Not `Send`: the plan shares the context's one stream and staging buffers with every other
/// object of that context, so it stays on the context's thread (the reference's `Cfft` is `Send` because it owns
/// everything it touches; a whole `Context` with its plans can still move as one unit).
pub struct HipFft<'c> { h: *mut aeth_fft, _ctx: PhantomData<&'c Context> }
impl<'c> HipFft<'c> {
    /// `Cfft::with_len` (src/fft.rs:147)
    pub fn with_len(ctx: &'c Context, len: usize) -> HipFft<'c> {
        let mut h = ptr::null_mut();
        check(unsafe { aeth_fft_create(ctx.h, len, 1, &mut h) });
        HipFft { h, _ctx: PhantomData }
    }
    fn host(&mut self, input: *const cf32, n_in: usize, output: *mut cf32, n_out: usize, sign: i32, s: Scale) {
        let (k, x) = scale_args(s);
        check(unsafe { aeth_fft_exec_host(self.h, input, n_in, output, n_out, sign, k, x) });
    }
    fn tmp(&mut self, input: &[cf32], sign: i32, s: Scale) -> &[cf32] {
        let (k, x) = scale_args(s);
        let mut view: *const cf32 = ptr::null();
        check(unsafe { aeth_fft_exec_tmp_host(self.h, input.as_ptr(), input.len(), sign, k, x, &mut view) });
        // borrow tied to &mut self: valid until the next call on this plan (src/fft.rs:68,73)
        unsafe { std::slice::from_raw_parts(view, self.len()) }
    }
}
impl<'c> Drop for HipFft<'c> { fn drop(&mut self) { unsafe { aeth_fft_destroy(self.h); } } }

impl<'c> Fft for HipFft<'c> {
    fn fwd(&mut self, input: &[cf32], output: &mut [cf32], s: Scale) {
        self.host(input.as_ptr(), input.len(), output.as_mut_ptr(), output.len(), AETH_SIGN_REF_FWD, s)
    }
    fn bwd(&mut self, input: &[cf32], output: &mut [cf32], s: Scale) {
        self.host(input.as_ptr(), input.len(), output.as_mut_ptr(), output.len(), AETH_SIGN_REF_BWD, s)
    }
    fn ifwd(&mut self, input: &mut [cf32], s: Scale) {
        let (p, n) = (input.as_mut_ptr(), input.len());
        self.host(p as *const cf32, n, p, n, AETH_SIGN_REF_FWD, s)     // in == out: in place on the C side
    }
    fn ibwd(&mut self, input: &mut [cf32], s: Scale) {
        let (p, n) = (input.as_mut_ptr(), input.len());
        self.host(p as *const cf32, n, p, n, AETH_SIGN_REF_BWD, s)
    }
    fn tfwd(&mut self, input: &[cf32], s: Scale) -> &[cf32] { self.tmp(input, AETH_SIGN_REF_FWD, s) }
    fn tbwd(&mut self, input: &[cf32], s: Scale) -> &[cf32] { self.tmp(input, AETH_SIGN_REF_BWD, s) }
    fn len(&self) -> usize { unsafe { aeth_fft_len(self.h) } }
}

/// Device-resident `[cf32]` with the `VecOps` method set (src/vecops.rs:39-89).
pub struct DeviceVec<'c> { ctx: &'c Context, p: *mut cf32, n: usize }
impl<'c> DeviceVec<'c> {
    pub fn from_slice(ctx: &'c Context, host: &[cf32]) -> DeviceVec<'c> {
        let mut p: *mut c_void = ptr::null_mut();
        check(unsafe { aeth_dev_alloc(ctx.h, host.len() * 8, &mut p) });
        check(unsafe { aeth_upload(ctx.h, p, host.as_ptr() as *const c_void, host.len() * 8) });
        DeviceVec { ctx, p: p as *mut cf32, n: host.len() }
    }
    pub fn to_vec(&mut self) -> Vec<cf32> {
        let mut v = vec![cf32::default(); self.n];
        check(unsafe { aeth_download(self.ctx.h, v.as_mut_ptr() as *mut c_void, self.p as *const c_void, self.n * 8) });
        v
    }
    pub fn len(&self) -> usize { self.n }
    pub fn vec_scale(&mut self, s: f32) -> &mut Self { check(unsafe { aeth_vec_scale(self.ctx.h, self.p, self.n, s) }); self }
    pub fn vec_mul(&mut self, o: &DeviceVec) -> &mut Self { check(unsafe { aeth_vec_mul(self.ctx.h, self.p, self.n, o.p, o.n) }); self }
    pub fn vec_div(&mut self, o: &DeviceVec) -> &mut Self { check(unsafe { aeth_vec_div(self.ctx.h, self.p, self.n, o.p, o.n) }); self }
    pub fn vec_conj(&mut self) -> &mut Self { check(unsafe { aeth_vec_conj(self.ctx.h, self.p, self.n) }); self }
    pub fn vec_mirror(&mut self) -> &mut Self { check(unsafe { aeth_vec_mirror(self.ctx.h, self.p, self.n) }); self }
    pub fn vec_clone(&mut self, o: &DeviceVec) -> &mut Self { check(unsafe { aeth_vec_clone(self.ctx.h, self.p, self.n, o.p, o.n) }); self }
    pub fn vec_zero(&mut self) -> &mut Self { check(unsafe { aeth_vec_zero(self.ctx.h, self.p, self.n) }); self }
    pub fn vec_add(&mut self, o: &DeviceVec) -> &mut Self { check(unsafe { aeth_vec_add(self.ctx.h, self.p, self.n, o.p, o.n) }); self }
    pub fn vec_sub(&mut self, o: &DeviceVec) -> &mut Self { check(unsafe { aeth_vec_sub(self.ctx.h, self.p, self.n, o.p, o.n) }); self }
    /// closures cannot cross the FFI: D2H, apply in order, H2D (slow by design)
    pub fn vec_mutate(&mut self, f: impl FnMut(&mut cf32)) -> &mut Self {
        let mut v = self.to_vec();
        v.iter_mut().for_each(f);
        check(unsafe { aeth_upload(self.ctx.h, self.p as *mut c_void, v.as_ptr() as *const c_void, self.n * 8) });
        self
    }
    /// the chained element-wise methods as ONE pass over memory (`aeth_vec_chain`; bit-identical to the separate calls):
    /// `v.fused().vec_add(&a).vec_mul(&b).vec_conj().run();` -- the operands are borrowed until `run`, so the borrow
    /// checker rules out an operand that aliases `self` exactly as it does for the separate calls
    pub fn fused<'v>(&'v mut self) -> Chain<'v, 'c> { Chain { v: self, steps: Vec::new() } }
    /// device frames through a plan: `len()` may be a multiple of `fft.len()` (a batch)
    pub fn vec_rfft(&mut self, fft: &mut HipFft, s: Scale) -> &mut Self { self.exec(fft, AETH_SIGN_REF_FWD, s) }
    pub fn vec_rifft(&mut self, fft: &mut HipFft, s: Scale) -> &mut Self { self.exec(fft, AETH_SIGN_REF_BWD, s) }
    fn exec(&mut self, fft: &mut HipFft, sign: i32, s: Scale) -> &mut Self {
        let (k, x) = scale_args(s);
        let batch = if fft.len() > 0 { self.n / fft.len() } else { 0 };
        check(unsafe { aeth_fft_exec(fft.h, self.p, self.n, self.p, batch, sign, k, x) });
        self
    }
}
impl<'c> Drop for DeviceVec<'c> { fn drop(&mut self) { unsafe { aeth_dev_free(self.ctx.h, self.p as *mut c_void); } } }

/// links recorded by `DeviceVec::fused()`
pub struct Chain<'v, 'c> { v: &'v mut DeviceVec<'c>, steps: Vec<aeth_vec_step> }
impl<'v, 'c> Chain<'v, 'c> {
    fn un(mut self, op: c_int, scale: f32) -> Self { self.steps.push(aeth_vec_step { op, other_dev: ptr::null(), n_other: 0, scale }); self }
    fn bin(mut self, op: c_int, o: &'v DeviceVec) -> Self { self.steps.push(aeth_vec_step { op, other_dev: o.p, n_other: o.n, scale: 0.0 }); self }
    pub fn vec_scale(self, s: f32) -> Self { self.un(AETH_VEC_SCALE, s) }
    pub fn vec_conj(self) -> Self { self.un(AETH_VEC_CONJ, 0.0) }
    pub fn vec_zero(self) -> Self { self.un(AETH_VEC_ZERO, 0.0) }
    pub fn vec_mul(self, o: &'v DeviceVec) -> Self { self.bin(AETH_VEC_MUL, o) }
    pub fn vec_div(self, o: &'v DeviceVec) -> Self { self.bin(AETH_VEC_DIV, o) }
    pub fn vec_add(self, o: &'v DeviceVec) -> Self { self.bin(AETH_VEC_ADD, o) }
    pub fn vec_sub(self, o: &'v DeviceVec) -> Self { self.bin(AETH_VEC_SUB, o) }
    pub fn vec_clone(self, o: &'v DeviceVec) -> Self { self.bin(AETH_VEC_CLONE, o) }
    pub fn run(self) { check(unsafe { aeth_vec_chain(self.v.ctx.h, self.v.p, self.v.n, self.steps.as_ptr(), self.steps.len()) }); }
}

pub mod sampling {
    use super::*;
    /// `sampling::interpolate` (src/sampling.rs:7-24): APPENDS to `dst`.
    pub fn interpolate(ctx: &Context, src: &[cf32], dst: &mut Vec<cf32>, n_between: usize) {
        assert!(!src.is_empty());                              // the reference unwrap()s src.last()
        let add = src.len() + (src.len() - 1) * n_between;
        dst.reserve(add);
        let mut written = 0usize;
        let tail = unsafe { dst.as_mut_ptr().add(dst.len()) };
        check(unsafe { aeth_host_interpolate(ctx.h, src.as_ptr(), src.len(), tail, add, n_between, 1, &mut written) });
        unsafe { dst.set_len(dst.len() + written) };
    }
    /// `sampling::downsample<T: Copy>` (src/sampling.rs:28-42)
    /// The reference's divisibility check is a `debug_assert_eq!` (:32-36): a debug build of this crate binds the
    /// checked entry point, a release build the one that floors the ratio, exactly as the reference itself behaves
    /// under `cargo test` and `cargo bench` (benches/benches.rs:113,130: 8096 -> 512).
    pub fn downsample<T: Copy>(ctx: &Context, src: &[T], dst: &mut [T]) {
        let (s, d, e) = (src.as_ptr() as *const c_void, dst.as_mut_ptr() as *mut c_void, std::mem::size_of::<T>());
        if cfg!(debug_assertions) { check(unsafe { aeth_host_downsample(ctx.h, s, src.len(), d, dst.len(), e) }); }
        else { check(unsafe { aeth_host_downsample_release(ctx.h, s, src.len(), d, dst.len(), e, 0) }); }
    }
    /// `sampling::downsample_sb<T: Copy>` (src/sampling.rs:49-62): the step_by variant
    pub fn downsample_sb<T: Copy>(ctx: &Context, src: &[T], dst: &mut [T]) {
        let (s, d, e) = (src.as_ptr() as *const c_void, dst.as_mut_ptr() as *mut c_void, std::mem::size_of::<T>());
        if cfg!(debug_assertions) { check(unsafe { aeth_host_downsample(ctx.h, s, src.len(), d, dst.len(), e) }); }
        else { check(unsafe { aeth_host_downsample_release(ctx.h, s, src.len(), d, dst.len(), e, 1) }); }
    }
}

/// Overlap-save FIR behind `fir::Fir`'s constructor shape (src/fir.rs:3-22 stores taps and a scratch but has
/// no filter method; this one filters).
/// Like `HipFft`, borrows its `Context` (Drop goes through the context) and stays on the context's thread.
pub struct Fir<'c> { h: *mut aeth_fir, _ctx: PhantomData<&'c Context> }
impl<'c> Fir<'c> {
    pub fn new(ctx: &'c Context, taps: &[cf32], fft_len: usize) -> Fir<'c> {
        let mut h = ptr::null_mut();
        check(unsafe { aeth_fir_create(ctx.h, taps.as_ptr(), taps.len(), fft_len, &mut h) });
        Fir { h, _ctx: PhantomData }
    }
    /// device-resident stream: y[n] = sum_k taps[k] x[n-k], zero initial state
    pub fn filter(&mut self, x: &DeviceVec, y: &mut DeviceVec) {
        assert_eq!(x.n, y.n, "Vectors must have same length");
        check(unsafe { aeth_fir_exec(self.h, ptr::null(), x.p, x.n, y.p) });
    }
    /// host-resident stream through the three-stage H2D | kernel | D2H pipeline (src/pipeline.rs counterpart)
    pub fn filter_stream(&mut self, x: &[cf32], y: &mut [cf32]) -> aeth_pipe_stats {
        assert_eq!(x.len(), y.len(), "Vectors must have same length");
        let mut st = aeth_pipe_stats::default();
        check(unsafe { aeth_fir_stream_host(self.h, x.as_ptr(), x.len(), y.as_mut_ptr(), 0, &mut st) });
        st
    }
    /// the same run, printing the reference pipeline's per-stage report (src/pipeline.rs:101-108) for the device stages
    /// and, when the slices had to be staged through the context's pinned pool, the two host stages
    pub fn filter_stream_report(&mut self, x: &[cf32], y: &mut [cf32]) -> aeth_pipe_util {
        assert_eq!(x.len(), y.len(), "Vectors must have same length");
        let mut u = aeth_pipe_util::default();
        check(unsafe { aeth_fir_stream_host_util(self.h, x.as_ptr(), x.len(), y.as_mut_ptr(), 0, &mut u) });
        for (name, active) in [("copy-in", u.active_copy_in), ("upload", u.active_upload), ("kernel", u.active_kernel),
                               ("download", u.active_download), ("copy-out", u.active_copy_out)].iter() {
            if name.starts_with("copy") && *active == 0.0 { continue; }
            println!("Stage: {:15} : Processed {} in {:3.3}s ({:9.2}/s); Utilisation: {:3.2}%",
                     name, u.chunks as u64, u.seconds, u.chunks / u.seconds, active / u.seconds * 100.0);
        }
        u
    }
}
impl<'c> Drop for Fir<'c> { fn drop(&mut self) { unsafe { aeth_fir_destroy(self.h); } } }

/// The device counterpart of `pipeline::new().add_stage(..)` (src/pipeline.rs:24-41, :123-137): five fixed stages --
/// copy-in | upload | compute | download | copy-out -- whose compute stage is one of the library's device ops (a closure
/// cannot cross the C ABI).  `run` takes host slices and returns what `aeth_stream_host` reports.
pub struct Stage<'c> { op: aeth_stream_op, ctx: &'c Context }
impl<'c> Stage<'c> {
    fn blank(ctx: &'c Context, kind: std::os::raw::c_int) -> Stage<'c> {
        Stage { ctx, op: aeth_stream_op { kind, fir: ptr::null_mut(), fft: ptr::null_mut(), sig_dev: ptr::null(), n_sig: 0, sign: 0,
                                          scale_kind_fwd: 0, x_fwd: 0.0, scale_kind_bwd: 0, x_bwd: 0.0, bits_per_symbol: 0,
                                          table_host: ptr::null(), compat: 0, n_between: 0, seed: 0, offset: 0 } }
    }
    pub fn fir(ctx: &'c Context, f: &Fir<'c>) -> Stage<'c> { let mut s = Stage::blank(ctx, AETH_STREAM_FIR); s.op.fir = f.h; s }
    /// `Fft::fwd` over `chunks_mut(fft_len)` (src/util/plot.rs:59-61)
    pub fn fft(ctx: &'c Context, f: &HipFft<'c>, scale: Scale) -> Stage<'c> {
        let (k, x) = scale_args(scale);
        let mut s = Stage::blank(ctx, AETH_STREAM_FFT); s.op.fft = f.h; s.op.sign = AETH_SIGN_REF_FWD; s.op.scale_kind_fwd = k; s.op.x_fwd = x; s
    }
    /// `vec_rfft -> vec_mul(&sig) -> vec_rifft` per frame (benches/benches.rs:410-416)
    pub fn mul_chain(ctx: &'c Context, f: &HipFft<'c>, sig: &DeviceVec, s_fwd: Scale, s_bwd: Scale) -> Stage<'c> {
        let (kf, xf) = scale_args(s_fwd); let (kb, xb) = scale_args(s_bwd);
        let mut s = Stage::blank(ctx, AETH_STREAM_FFT_MUL_IFFT); s.op.fft = f.h; s.op.sig_dev = sig.p; s.op.n_sig = sig.n;
        s.op.scale_kind_fwd = kf; s.op.x_fwd = xf; s.op.scale_kind_bwd = kb; s.op.x_bwd = xb; s
    }
    /// the chain, then `Modulation::demod_naive` (examples/modem.rs:28-31): `bits_per_symbol` bytes out per sample
    pub fn correlate_demod(ctx: &'c Context, f: &HipFft<'c>, sig: &DeviceVec, bits_per_symbol: usize) -> Stage<'c> {
        let mut s = Stage::blank(ctx, AETH_STREAM_FFT_MUL_IFFT_DEMOD); s.op.fft = f.h; s.op.sig_dev = sig.p; s.op.n_sig = sig.n;
        s.op.bits_per_symbol = bits_per_symbol as std::os::raw::c_int; s.op.compat = 1; s
    }
    /// the transform, then `sampling::interpolate` per frame (src/sampling.rs:7-24)
    pub fn fft_interpolate(ctx: &'c Context, f: &HipFft<'c>, n_between: usize, scale: Scale) -> Stage<'c> {
        let (k, x) = scale_args(scale);
        let mut s = Stage::blank(ctx, AETH_STREAM_FFT_INTERPOLATE); s.op.fft = f.h; s.op.sign = AETH_SIGN_REF_FWD;
        s.op.scale_kind_fwd = k; s.op.x_fwd = x; s.op.n_between = n_between; s.op.compat = 1; s
    }
    pub fn out_count(&self, n_in: usize) -> usize { unsafe { aeth_stream_out_count(self.ctx.h, &self.op, n_in) } }
    /// cf32 in, `T` out (`cf32`, or `u8` for the demodulating stage); `out.len()` must be `out_count(x.len())`
    pub fn run<T: Copy>(&self, x: &[cf32], out: &mut [T]) -> aeth_pipe_stats {
        assert_eq!(out.len(), self.out_count(x.len()), "Vectors must have same length");
        let mut st = aeth_pipe_stats::default();
        check(unsafe { aeth_stream_host(self.ctx.h, &self.op, x.as_ptr() as *const _, x.len(), out.as_mut_ptr() as *mut _, out.len(), 0, &mut st) });
        st
    }
}

/// Pinned host buffers behind the reference's `pool::Pool<T>` (src/pool.rs:43-221): `take`, `take_or_make`, `len`,
/// `cap`; an `Elem` derefs into `[cf32]` and goes back to the pool on drop.  Elements are page-locked once by the
/// library, so `Fir::filter_stream` copies from / to them directly (no staging, no registration of caller memory).
pub struct PinnedPool<'c> { h: *mut aeth_pool, n: usize, _ctx: PhantomData<&'c Context> }
pub struct PinnedElem<'p, 'c> { pool: &'p PinnedPool<'c>, p: *mut cf32 }
impl<'c> PinnedPool<'c> {
    /// `pool::make(initial_len, maker, resetter)` with maker = one pinned buffer of `n` samples
    pub fn make(ctx: &'c Context, n: usize, initial_len: usize, zero_on_return: bool) -> PinnedPool<'c> {
        let mut h = ptr::null_mut();
        check(unsafe { aeth_pool_create(ctx.h, n * std::mem::size_of::<cf32>(), initial_len,
                                        if zero_on_return { AETH_POOL_ZERO_ON_RETURN } else { 0 }, &mut h) });
        PinnedPool { h, n, _ctx: PhantomData }
    }
    pub fn take(&self) -> Option<PinnedElem<'_, 'c>> {
        let mut p = ptr::null_mut();
        check(unsafe { aeth_pool_take(self.h, &mut p) });
        if p.is_null() { None } else { Some(PinnedElem { pool: self, p: p as *mut cf32 }) }
    }
    pub fn take_or_make(&self) -> PinnedElem<'_, 'c> {
        let mut p = ptr::null_mut();
        check(unsafe { aeth_pool_take_or_make(self.h, &mut p) });
        PinnedElem { pool: self, p: p as *mut cf32 }
    }
    pub fn len(&self) -> usize { unsafe { aeth_pool_len(self.h) } }
    pub fn cap(&self) -> usize { unsafe { aeth_pool_cap(self.h) } }
}
impl<'c> Drop for PinnedPool<'c> { fn drop(&mut self) { unsafe { aeth_pool_destroy(self.h); } } }
impl<'p, 'c> Drop for PinnedElem<'p, 'c> { fn drop(&mut self) { unsafe { aeth_pool_give_back(self.pool.h, self.p as *mut c_void); } } }
impl<'p, 'c> std::ops::Deref for PinnedElem<'p, 'c> {
    type Target = [cf32];
    fn deref(&self) -> &[cf32] { unsafe { std::slice::from_raw_parts(self.p, self.pool.n) } }
}
impl<'p, 'c> std::ops::DerefMut for PinnedElem<'p, 'c> {
    fn deref_mut(&mut self) -> &mut [cf32] { unsafe { std::slice::from_raw_parts_mut(self.p, self.pool.n) } }
}

/// `modulation::Modulation` for the generic BPSK/QPSK tables (src/modulation.rs:5-149) on device buffers,
/// and `noise::Awgn::apply` (src/noise.rs:53-59).
pub mod modem {
    use super::*;
    /// `m.modulate(&bits)`: one u8 per bit in, `bits.len() / bits_per_symbol` symbols out
    pub fn modulate(ctx: &Context, bits_dev: *const u8, nbits: usize, bits_per_symbol: i32, out: &mut DeviceVec) {
        check(unsafe { aeth_modulate(ctx.h, bits_dev, nbits, bits_per_symbol, ptr::null(), out.p, out.n) });
    }
    /// `m.demod_naive(..)`; `compat` reproduces the reference's `idx & 1u8 << 1` output (src/modulation.rs:54)
    pub fn demod_naive(ctx: &Context, sym: &DeviceVec, bits_per_symbol: i32, bits_dev: *mut u8, nbits: usize, compat: bool) {
        check(unsafe { aeth_demod_naive(ctx.h, sym.p, sym.n, bits_per_symbol, ptr::null(), bits_dev, nbits, compat as i32) });
    }
    /// `noise::new(power, seed).apply(&mut signal)` -- the double scaling of the reference is kept
    pub fn awgn_apply(ctx: &Context, signal: &mut DeviceVec, power: f32, seed: u64, offset: u64) {
        check(unsafe { aeth_awgn_apply(ctx.h, signal.p, signal.n, power, seed, offset) });
    }
}
