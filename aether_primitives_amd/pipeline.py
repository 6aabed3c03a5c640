"""Host-resident streams through the device, stage by stage: the mirror of the reference's `pipeline` module
(src/pipeline.rs:24-41 `add_stage`, :123-137 `new`, per-stage report :89-114) over `aeth_stream_host`.

The reference builds a pipeline from closures; the device's compute stage is one of the library's ops instead
(include/aether_hip.h, aeth_stream_op).  Copy-in, upload, download and copy-out are the other four stages."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check
from .fft import Scale, SIGN_REF_FWD

FIR, FFT, FFT_MUL_IFFT, FFT_MUL_IFFT_DEMOD, FFT_INTERPOLATE, FIR_DECIM, MODULATE_AWGN = range(7)


class _Op(C.Structure):
    # struct aeth_stream_op, field by field
    _fields_ = [("kind", C.c_int), ("fir", C.c_void_p), ("fft", C.c_void_p), ("sig_dev", C.c_void_p), ("n_sig", C.c_size_t),
                ("sign", C.c_int), ("scale_kind_fwd", C.c_int), ("x_fwd", C.c_float), ("scale_kind_bwd", C.c_int),
                ("x_bwd", C.c_float), ("bits_per_symbol", C.c_int), ("table_host", C.c_void_p), ("compat", C.c_int),
                ("n_between", C.c_size_t), ("seed", C.c_uint64), ("offset", C.c_uint64)]


class _Stats(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("seconds", "samples", "chunks", "pinned")]


class _Util(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("seconds", "samples", "chunks", "pinned", "active_upload", "active_kernel",
                                          "active_download", "active_copy_in", "active_copy_out")]


class Stage:
    """The compute stage of a host pipeline: an op descriptor plus whatever keeps its operands alive."""

    in_dtype = np.complex64

    def __init__(self, ctx, op, out_dtype, keep=()):
        self.ctx, self.op, self.out_dtype, self._keep = ctx, op, out_dtype, keep

    @staticmethod
    def fir(f):
        return Stage(f.ctx, _Op(kind=FIR, fir=f.h), np.complex64, (f,))

    @staticmethod
    def modulate_awgn(ctx, bits_per_symbol, power, seed, offset=0, table=None):
        """Modulation::modulate, then Awgn::apply on the fresh symbols (examples/modem.rs:19-26): BIT BYTES in, symbols out"""
        tab = None if table is None else np.ascontiguousarray(table, dtype=np.complex64)
        st = Stage(ctx, _Op(kind=MODULATE_AWGN, bits_per_symbol=bits_per_symbol, table_host=None if tab is None else tab.ctypes.data,
                            x_fwd=float(np.float32(power)), seed=int(seed), offset=int(offset)), np.complex64, (tab,))
        st.in_dtype = np.uint8
        return st

    @staticmethod
    def fir_decim(f, dec):
        """the filter, then sampling::downsample (sampling.rs:28-42) in its store: n in, n / dec out"""
        return Stage(f.ctx, _Op(kind=FIR_DECIM, fir=f.h, n_between=dec), np.complex64, (f,))

    @staticmethod
    def fft(plan, scale=Scale.NONE, sign=SIGN_REF_FWD):
        """Fft::fwd / bwd over chunks_mut(fft_len) (src/util/plot.rs:59-61)"""
        return Stage(plan.ctx, _Op(kind=FFT, fft=plan.h, sign=sign, scale_kind_fwd=scale.kind, x_fwd=scale.x), np.complex64, (plan,))

    @staticmethod
    def mul_chain(plan, sig, s_fwd=Scale.NONE, s_bwd=Scale.NONE):
        """vec_rfft -> vec_mul(&sig) -> vec_rifft per frame (benches/benches.rs:410-416); sig is a DeviceVec"""
        return Stage(plan.ctx, _Op(kind=FFT_MUL_IFFT, fft=plan.h, sig_dev=sig._p(), n_sig=sig.n, scale_kind_fwd=s_fwd.kind, x_fwd=s_fwd.x,
                                   scale_kind_bwd=s_bwd.kind, x_bwd=s_bwd.x), np.complex64, (plan, sig))

    @staticmethod
    def correlate_demod(plan, sig, bits_per_symbol=2, table=None, compat=True, s_fwd=Scale.NONE, s_bwd=Scale.NONE):
        """the chain, then Modulation::demod_naive (examples/modem.rs:28-31): bits_per_symbol bytes out per sample"""
        tab = None if table is None else np.ascontiguousarray(table, dtype=np.complex64)
        return Stage(plan.ctx, _Op(kind=FFT_MUL_IFFT_DEMOD, fft=plan.h, sig_dev=sig._p(), n_sig=sig.n, scale_kind_fwd=s_fwd.kind,
                                   x_fwd=s_fwd.x, scale_kind_bwd=s_bwd.kind, x_bwd=s_bwd.x, bits_per_symbol=bits_per_symbol,
                                   table_host=None if tab is None else tab.ctypes.data, compat=1 if compat else 0),
                     np.uint8, (plan, sig, tab))

    @staticmethod
    def fft_interpolate(plan, n_between, scale=Scale.NONE, sign=SIGN_REF_FWD, compat_im=True):
        """the transform, then sampling::interpolate per frame (src/sampling.rs:7-24; BASELINE config 5)"""
        return Stage(plan.ctx, _Op(kind=FFT_INTERPOLATE, fft=plan.h, sign=sign, scale_kind_fwd=scale.kind, x_fwd=scale.x,
                                   n_between=n_between, compat=1 if compat_im else 0), np.complex64, (plan,))

    def out_count(self, n_in):
        return _lib.load().aeth_stream_out_count(self.ctx.h, C.byref(self.op), n_in)


_STAGE_NAMES = ("copy-in", "upload", "compute", "download", "copy-out")


def run(stage, x, out=None, chunk=0, report=False):
    """x (host cf32 array) through copy-in | upload | stage | download | copy-out; returns (out, stats).  report=True
    adds the seconds each stage was active and `lines` in the format of the reference's report (pipeline.rs:101-108)."""
    lib = _lib.load()
    x = np.ascontiguousarray(x, dtype=stage.in_dtype)
    n_out = stage.out_count(x.size)
    if out is None:
        out = np.empty(n_out, stage.out_dtype)
    assert out.dtype == stage.out_dtype and out.flags["C_CONTIGUOUS"]
    if report:
        u = _Util()
        check(lib.aeth_stream_host_util(stage.ctx.h, C.byref(stage.op), x.ctypes.data_as(C.c_void_p), x.size,
                                        out.ctypes.data_as(C.c_void_p), out.size, chunk, C.byref(u)))
        st = {k: getattr(u, k) for k, _ in _Util._fields_}
        act = (u.active_copy_in, u.active_upload, u.active_kernel, u.active_download, u.active_copy_out)
        st["lines"] = [f"Stage: {name:15} : Processed {int(u.chunks)} in {u.seconds:3.3f}s ({u.chunks / u.seconds:9.2f}/s); "
                       f"Utilisation: {a / u.seconds * 100.0:3.2f}%"
                       for name, a in zip(_STAGE_NAMES, act) if not (name.startswith("copy") and a == 0)] if u.seconds > 0 else []
        return out, st
    s = _Stats()
    check(lib.aeth_stream_host(stage.ctx.h, C.byref(stage.op), x.ctypes.data_as(C.c_void_p), x.size,
                               out.ctypes.data_as(C.c_void_p), out.size, chunk, C.byref(s)))
    return out, {k: getattr(s, k) for k, _ in _Stats._fields_}


def run_file(stage, in_path, out_path, chunk=0):
    """raw cf32 file (src/util/file.rs format) -> stage -> raw file of the stage's output type; returns the stats"""
    import os
    s = _Stats()
    check(_lib.load().aeth_stream_file(stage.ctx.h, C.byref(stage.op), os.fsencode(in_path), os.fsencode(out_path), chunk, C.byref(s)))
    return {k: getattr(s, k) for k, _ in _Stats._fields_}


def run_chain(stages, x, out=None, chunk=0, report=False):
    """`pipeline::new(..).add_stage(a).add_stage(b)` (src/pipeline.rs:24-41): several stages as ONE compute stage -- stage
    i's output is stage i + 1's input on the device; only `x` goes up and only the last stage's output comes back.
    Returns (out, stats); report=True adds the per-stage busy seconds and `lines` as `run` does."""
    lib = _lib.load()
    ctx = stages[0].ctx
    x = np.ascontiguousarray(x, dtype=stages[0].in_dtype)
    ops = (_Op * len(stages))(*[s.op for s in stages])
    n_out = lib.aeth_stream_chain_out_count(ctx.h, ops, len(stages), x.size)
    dt = stages[-1].out_dtype
    if out is None:
        out = np.empty(n_out, dt)
    assert out.dtype == dt and out.flags["C_CONTIGUOUS"]
    s, u = _Stats(), _Util()
    check(lib.aeth_stream_host_chain(ctx.h, ops, len(stages), x.ctypes.data_as(C.c_void_p), x.size, out.ctypes.data_as(C.c_void_p),
                                     out.size, chunk, None if report else C.byref(s), C.byref(u) if report else None))
    if not report:
        return out, {k: getattr(s, k) for k, _ in _Stats._fields_}
    st = {k: getattr(u, k) for k, _ in _Util._fields_}
    act = (u.active_copy_in, u.active_upload, u.active_kernel, u.active_download, u.active_copy_out)
    st["lines"] = [f"Stage: {name:15} : Processed {int(u.chunks)} in {u.seconds:3.3f}s ({u.chunks / u.seconds:9.2f}/s); "
                   f"Utilisation: {a / u.seconds * 100.0:3.2f}%"
                   for name, a in zip(_STAGE_NAMES, act) if not (name.startswith("copy") and a == 0)] if u.seconds > 0 else []
    return out, st
