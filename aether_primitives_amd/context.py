"""Context + the VecOps mirror (reference: src/vecops.rs:39-89).

`DeviceVec` is the device-resident receiver: the same chainable
method set as the trait, each call one stream-ordered kernel launch, no host
traffic.  `HostVec` is the literal host-slice receiver: every call stages
through the GPU and returns with the numpy array updated (one H2D + D2H per
call, as a drop-in `impl VecOps for [cf32]` would).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


def _c64(a):
    a = np.ascontiguousarray(a, dtype=np.complex64)
    return a


class Context:
    """One device + one HIP stream (aeth_ctx).  Not thread-safe, like `&mut self`."""

    def __init__(self, device=0, stream=None):
        self._lib = _lib.load()
        h = C.c_void_p()
        if stream is None:
            check(self._lib.aeth_ctx_create(device, C.byref(h)))
        else:
            check(self._lib.aeth_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(h)))
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self._lib.aeth_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self._lib.aeth_ctx_sync(self.h))

    def trim(self):
        """Give back what the context retains between calls (host pipeline: stage streams, device slots, pinned staging
        elements, copy threads; the device scratch of the host-slice flavours)."""
        check(self._lib.aeth_ctx_trim(self.h))

    def set_overlap(self, enable=True):
        """Let consecutive independent `Fir.filter` launches alternate between two HIP queues (the end of one
        launch then runs beside the start of the next); everything else stays ordered as on one stream."""
        check(self._lib.aeth_ctx_set_overlap(self.h, 1 if enable else 0))

    @property
    def overlap(self):
        return bool(self._lib.aeth_ctx_overlap(self.h))

    @property
    def stream(self):
        """hipStream_t of the context for interop.  SIDE EFFECT: handing the stream out parks the overlap lane (every
        later launch stays on this one stream, `overlap` reads False) until `set_overlap(True)` re-arms it."""
        return self._lib.aeth_ctx_stream(self.h)

    # ---- device memory ----
    def alloc(self, nbytes):
        p = C.c_void_p()
        check(self._lib.aeth_dev_alloc(self.h, nbytes, C.byref(p)))
        return p.value

    def free(self, ptr):
        check(self._lib.aeth_dev_free(self.h, C.c_void_p(ptr)))

    def upload(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        check(self._lib.aeth_upload(self.h, C.c_void_p(dptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def download(self, dptr, arr):
        assert arr.flags["C_CONTIGUOUS"]
        check(self._lib.aeth_download(self.h, arr.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), arr.nbytes))

    # ---- constructors ----
    def vec(self, host_array):
        """Upload a complex64 array -> DeviceVec."""
        a = _c64(host_array).reshape(-1)
        v = DeviceVec(self, a.size)
        if a.size:
            self.upload(v.ptr, a)
        return v

    def empty(self, n):
        return DeviceVec(self, n)

    # ---- events (bench) ----
    def event(self):
        return Event(self)


class Event:
    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        check(ctx._lib.aeth_event_create(ctx.h, C.byref(h)))
        self.h = h

    def record(self):
        check(self.ctx._lib.aeth_event_record(self.h))

    def sync(self):
        check(self.ctx._lib.aeth_event_sync(self.h))

    def elapsed_ms(self, stop):
        ms = C.c_float()
        check(self.ctx._lib.aeth_event_elapsed_ms(self.h, stop.h, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if self.h:
                self.ctx._lib.aeth_event_destroy(self.h)
                self.h = None
        except Exception:
            pass


class DeviceVec:
    """Device-resident `[cf32]` with the VecOps method set (src/vecops.rs:39-89)."""

    def __init__(self, ctx, n, ptr=None, offset=0, owner=None):
        self.ctx = ctx
        self.n = int(n)
        self._owner = owner
        if ptr is None:
            self._base = ctx.alloc(max(self.n, 1) * 8)
            self.ptr = self._base
            self._owns = True
        else:
            self._base = ptr
            self.ptr = ptr + offset * 8
            self._owns = False

    def __len__(self):
        return self.n

    def __del__(self):
        try:
            if self._owns and self._base and self.ctx.h:
                self.ctx.free(self._base)
                self._base = None
        except Exception:
            pass

    def slice(self, start, stop):
        """Borrowed sub-slice `self[start..stop]` (keeps the parent alive)."""
        assert 0 <= start <= stop <= self.n
        return DeviceVec(self.ctx, stop - start, ptr=self.ptr, offset=start, owner=self)

    def to_host(self):
        out = np.empty(self.n, np.complex64)
        if self.n:
            self.ctx.download(self.ptr, out)
        return out

    def _p(self):
        return C.c_void_p(self.ptr)

    def fused(self):
        """A chain of the element-wise methods as ONE pass over memory (`aeth_vec_chain`): record the links with the
        usual names, `run()` (or leaving a `with` block) executes them -- bit-identical to the separate calls:
            v.fused().vec_add(a).vec_mul(b).vec_conj().run()         # BASELINE config 1: one launch, 32 B per sample"""
        return _Chain(self)

    def _other(self, other):
        if isinstance(other, DeviceVec):
            return other
        return self.ctx.vec(other)      # host slice given: stage it (AsRef<[cf32]>)

    # ---- trait VecOps ----
    def vec_scale(self, scale):                                   # vecops.rs:41
        check(self.ctx._lib.aeth_vec_scale(self.ctx.h, self._p(), self.n, float(np.float32(scale)))); return self

    def vec_mul(self, other):                                     # vecops.rs:44
        o = self._other(other); check(self.ctx._lib.aeth_vec_mul(self.ctx.h, self._p(), self.n, o._p(), o.n)); return self

    def vec_div(self, other):                                     # vecops.rs:47
        o = self._other(other); check(self.ctx._lib.aeth_vec_div(self.ctx.h, self._p(), self.n, o._p(), o.n)); return self

    def vec_conj(self):                                           # vecops.rs:50
        check(self.ctx._lib.aeth_vec_conj(self.ctx.h, self._p(), self.n)); return self

    def vec_mirror(self):                                         # vecops.rs:54
        check(self.ctx._lib.aeth_vec_mirror(self.ctx.h, self._p(), self.n)); return self

    def vec_mul_frames(self, sig, frame_len=None):
        """every frame of `frame_len` samples *= sig (one launch; benches.rs:410-416's vec_mul per chunk)"""
        o = self._other(sig); L = frame_len or o.n
        check(self.ctx._lib.aeth_vec_mul_frames(self.ctx.h, self._p(), L, self.n // L if L else 0, o._p(), o.n)); return self

    def vec_mirror_frames(self, frame_len):
        check(self.ctx._lib.aeth_vec_mirror_frames(self.ctx.h, self._p(), frame_len, self.n // frame_len)); return self

    def vec_clone(self, other):                                   # vecops.rs:58
        o = self._other(other); check(self.ctx._lib.aeth_vec_clone(self.ctx.h, self._p(), self.n, o._p(), o.n)); return self

    def vec_zero(self):                                           # vecops.rs:61
        check(self.ctx._lib.aeth_vec_zero(self.ctx.h, self._p(), self.n)); return self

    def vec_mutate(self, f):                                      # vecops.rs:64
        """Closure per element, in order: cannot cross the FFI -> D2H, apply, H2D (slow by design)."""
        h = self.to_host()
        for i in range(h.size):
            r = f(h[i])
            if r is not None:
                h[i] = r
        if self.n:
            self.ctx.upload(self.ptr, h)
        return self

    def vec_add(self, other):                                     # vecops.rs:67
        o = self._other(other); check(self.ctx._lib.aeth_vec_add(self.ctx.h, self._p(), self.n, o._p(), o.n)); return self

    def vec_sub(self, other):                                     # vecops.rs:70
        o = self._other(other); check(self.ctx._lib.aeth_vec_sub(self.ctx.h, self._p(), self.n, o._p(), o.n)); return self

    def vec_fft(self, scale):                                     # vecops.rs:74, :185-189 (the reference plans per call; the context caches)
        check(self.ctx._lib.aeth_vec_fft(self.ctx.h, self._p(), self.n, +1, scale.kind, scale.x)); return self

    def vec_ifft(self, scale):                                    # vecops.rs:78, :191-196
        check(self.ctx._lib.aeth_vec_fft(self.ctx.h, self._p(), self.n, -1, scale.kind, scale.x)); return self

    def vec_rfft(self, fft, scale):                               # vecops.rs:83, :198-202
        fft.ifwd(self, scale); return self

    def vec_rifft(self, fft, scale):                              # vecops.rs:88, :203-207
        fft.ibwd(self, scale); return self


class _VecStep(C.Structure):
    # struct aeth_vec_step
    _fields_ = [("op", C.c_int), ("other_dev", C.c_void_p), ("n_other", C.c_size_t), ("scale", C.c_float)]


class _Chain:
    """links recorded for DeviceVec.fused(); AETH_VEC_*: scale 0, mul 1, div 2, conj 3, add 4, sub 5, clone 6, zero 7"""

    def __init__(self, vec):
        self.vec, self.steps, self._keep = vec, [], []

    def _bin(self, op, other):
        o = self.vec._other(other)
        self._keep.append(o)
        self.steps.append(_VecStep(op, o.ptr, o.n, 0.0)); return self

    def vec_scale(self, s): self.steps.append(_VecStep(0, None, 0, float(np.float32(s)))); return self
    def vec_mul(self, o): return self._bin(1, o)
    def vec_div(self, o): return self._bin(2, o)
    def vec_conj(self): self.steps.append(_VecStep(3, None, 0, 0.0)); return self
    def vec_add(self, o): return self._bin(4, o)
    def vec_sub(self, o): return self._bin(5, o)
    def vec_clone(self, o): return self._bin(6, o)
    def vec_zero(self): self.steps.append(_VecStep(7, None, 0, 0.0)); return self

    def run(self):
        arr = (_VecStep * max(len(self.steps), 1))(*self.steps)
        check(self.vec.ctx._lib.aeth_vec_chain(self.vec.ctx.h, self.vec._p(), self.vec.n, arr, len(self.steps)))
        self.steps, self._keep = [], []
        return self.vec

    def __enter__(self): return self
    def __exit__(self, et, ev, tb):
        if et is None: self.run()


class HostVec:
    """Host slice receiver: numpy complex64 array mutated in place through the GPU."""

    def __init__(self, ctx, array):
        assert isinstance(array, np.ndarray) and array.dtype == np.complex64 and array.flags["C_CONTIGUOUS"]
        self.ctx = ctx
        self.a = array

    def _p(self):
        return self.a.ctypes.data_as(C.c_void_p)

    def _bin(self, name, other):
        o = _c64(other)
        check(getattr(self.ctx._lib, name)(self.ctx.h, self._p(), self.a.size, o.ctypes.data_as(C.c_void_p), o.size))
        return self

    def vec_scale(self, s):
        check(self.ctx._lib.aeth_host_vec_scale(self.ctx.h, self._p(), self.a.size, float(np.float32(s)))); return self

    def vec_mul(self, o): return self._bin("aeth_host_vec_mul", o)
    def vec_div(self, o): return self._bin("aeth_host_vec_div", o)
    def vec_add(self, o): return self._bin("aeth_host_vec_add", o)
    def vec_sub(self, o): return self._bin("aeth_host_vec_sub", o)
    def vec_clone(self, o): return self._bin("aeth_host_vec_clone", o)

    def vec_conj(self):
        check(self.ctx._lib.aeth_host_vec_conj(self.ctx.h, self._p(), self.a.size)); return self

    def vec_mirror(self):
        check(self.ctx._lib.aeth_host_vec_mirror(self.ctx.h, self._p(), self.a.size)); return self

    def vec_zero(self):
        check(self.ctx._lib.aeth_host_vec_zero(self.ctx.h, self._p(), self.a.size)); return self

    def vec_mutate(self, f):
        for i in range(self.a.size):
            r = f(self.a[i])
            if r is not None:
                self.a[i] = r
        return self

    def vec_fft(self, scale):
        check(self.ctx._lib.aeth_host_vec_fft(self.ctx.h, self._p(), self.a.size, +1, scale.kind, scale.x)); return self

    def vec_ifft(self, scale):
        check(self.ctx._lib.aeth_host_vec_fft(self.ctx.h, self._p(), self.a.size, -1, scale.kind, scale.x)); return self

    def vec_rfft(self, fft, scale):
        fft.ifwd(self.a, scale); return self

    def vec_rifft(self, fft, scale):
        fft.ibwd(self.a, scale); return self
