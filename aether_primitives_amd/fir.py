"""FIR by overlap-save (the reference's Fir<T>, src/fir.rs:3-22, has no filter
method; the build defines it from the chain at benches/benches.rs:410-416)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check
from .context import DeviceVec


class Fir:
    def __init__(self, ctx, taps, fft_len=2048):
        self.ctx = ctx
        self._lib = _lib.load()
        taps = np.ascontiguousarray(taps, dtype=np.complex64)
        h = C.c_void_p()
        check(self._lib.aeth_fir_create(ctx.h, taps.ctypes.data_as(C.c_void_p), taps.size, fft_len, C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self._lib.aeth_fir_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def ntaps(self): return self._lib.aeth_fir_ntaps(self.h)
    @property
    def fft_len(self): return self._lib.aeth_fir_fft_len(self.h)
    @property
    def hop(self): return self._lib.aeth_fir_hop(self.h)

    def filter(self, x, out=None, hist=None):
        """y[n] = sum_k taps[k] x[n-k]; zero initial state unless `hist` (ntaps-1 samples)."""
        if isinstance(x, DeviceVec):
            if out is None:
                out = DeviceVec(self.ctx, x.n)
            hp = hist._p() if hist is not None else None
            check(self._lib.aeth_fir_exec(self.h, hp, x._p(), x.n, out._p()))
            return out
        x = np.ascontiguousarray(x, dtype=np.complex64)
        if out is None:
            out = np.empty_like(x)
        hp = None
        if hist is not None:
            hist = np.ascontiguousarray(hist, dtype=np.complex64)
            assert hist.size == self.ntaps - 1
            hp = hist.ctypes.data_as(C.c_void_p)
        check(self._lib.aeth_fir_exec_host(self.h, hp, x.ctypes.data_as(C.c_void_p), x.size,
                                           out.ctypes.data_as(C.c_void_p)))
        return out

    def filter_decim(self, x, dec, out=None, hist=None):
        """filter, keep every dec-th output: sampling::downsample(&fir(x), &mut out) (sampling.rs:28-42) in one launch"""
        n_out = x.n // dec if dec else 0
        if out is None:
            out = DeviceVec(self.ctx, n_out)
        hp = hist._p() if hist is not None else None
        check(self._lib.aeth_fir_exec_decim(self.h, hp, x._p(), x.n, out._p(), out.n))
        return out

    def filter_file(self, in_path, out_path, chunk=0):
        """raw cf32 file -> FIR -> raw cf32 file (util::file format), through the three-stage stream pipeline."""
        import os

        class _Stats(C.Structure):
            _fields_ = [("seconds", C.c_double), ("samples", C.c_double), ("chunks", C.c_double), ("pinned", C.c_double)]
        st = _Stats()
        check(self._lib.aeth_fir_stream_file(self.h, os.fsencode(in_path), os.fsencode(out_path), chunk, C.byref(st)))
        return {"seconds": st.seconds, "samples": st.samples, "chunks": st.chunks, "pinned": st.pinned}

    def filter_stream(self, x, out=None, chunk=0, report=False):
        """Host array through the device in hop-aligned chunks: copy-in | upload | kernel | download | copy-out (PCIe-rate
        path).  Arrays that live in pinned pool elements (aether_primitives_amd.pool) or registered ranges are copied
        from / to directly; anything else is staged through the context's own pinned pool by host threads.
        Returns (y, stats) with stats = dict(seconds, samples, chunks, pinned); report=True adds the seconds each stage
        was active and `lines`, the reference pipeline's per-stage report (pipeline.rs:101-108)."""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        if out is None:
            out = np.empty_like(x)
        assert out.dtype == np.complex64 and out.flags["C_CONTIGUOUS"] and out.size == x.size
        if report:
            class _Util(C.Structure):
                _fields_ = [(k, C.c_double) for k in ("seconds", "samples", "chunks", "pinned", "active_upload", "active_kernel",
                                                      "active_download", "active_copy_in", "active_copy_out")]
            u = _Util()
            check(self._lib.aeth_fir_stream_host_util(self.h, x.ctypes.data_as(C.c_void_p), x.size,
                                                      out.ctypes.data_as(C.c_void_p), chunk, C.byref(u)))
            st = {k: getattr(u, k) for k, _ in _Util._fields_}
            stages = [("copy-in", u.active_copy_in), ("upload", u.active_upload), ("kernel", u.active_kernel),
                      ("download", u.active_download), ("copy-out", u.active_copy_out)]
            # an empty stream runs nothing: no rates to print (seconds == 0)
            st["lines"] = [f"Stage: {name:15} : Processed {int(u.chunks)} in {u.seconds:3.3f}s ({u.chunks / u.seconds:9.2f}/s); "
                           f"Utilisation: {act / u.seconds * 100.0:3.2f}%"
                           for name, act in stages if not (name.startswith("copy") and act == 0)] if u.seconds > 0 else []
            return out, st

        class _Stats(C.Structure):
            _fields_ = [("seconds", C.c_double), ("samples", C.c_double), ("chunks", C.c_double), ("pinned", C.c_double)]
        st = _Stats()
        check(self._lib.aeth_fir_stream_host(self.h, x.ctypes.data_as(C.c_void_p), x.size,
                                             out.ctypes.data_as(C.c_void_p), chunk, C.byref(st)))
        return out, {"seconds": st.seconds, "samples": st.samples, "chunks": st.chunks, "pinned": st.pinned}
