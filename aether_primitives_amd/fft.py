"""`Scale`, and `HipFft` -- the `impl Fft` that plugs in where the reference's
`Cfft` does (trait Fft: src/fft.rs:48-77; Cfft: src/fft.rs:134-235).

Every method accepts either host slices (numpy complex64: the literal trait
signature, one frame per call, synchronous) or `DeviceVec`s (device-resident;
`len(vec)` may be a multiple of `len()` = a batch of frames, stream-ordered).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check
from .context import DeviceVec

# exponent sign bound to the reference's method names in exactly one place:
# Cfft::with_len plans `fwd` with FFTplanner::new(true) (= rustfft "inverse", +j)
# and `bwd` with FFTplanner::new(false) (src/fft.rs:148,150).
SIGN_REF_FWD = +1
SIGN_REF_BWD = -1


class Scale:
    """enum Scale (src/fft.rs:6-18)."""
    __slots__ = ("kind", "x")

    def __init__(self, kind, x=0.0):
        self.kind = kind
        self.x = float(np.float32(x))

    @staticmethod
    def X(x):
        return Scale(3, x)

    def factor(self, n):
        return np.float32(_lib.load().aeth_scale_factor(self.kind, n, self.x))

    def scale(self, data):
        """Scale::scale (src/fft.rs:22-37) on a DeviceVec."""
        check(_lib.load().aeth_scale_apply(data.ctx.h, self.kind, self.x, data._p(), data.n))

    def __repr__(self):
        return ["Scale.None", "Scale.SN", "Scale.N", f"Scale.X({self.x})"][self.kind]


Scale.NONE = Scale(0)
Scale.SN = Scale(1)
Scale.N = Scale(2)


def _is_dev(x):
    return isinstance(x, DeviceVec)


class HipFft:
    """Fixed-length complex FFT plan on the GPU (replaces Cfft::with_len, src/fft.rs:147)."""

    def __init__(self, ctx, length, max_batch=1):
        self.ctx = ctx
        self._lib = _lib.load()
        h = C.c_void_p()
        check(self._lib.aeth_fft_create(ctx.h, length, max_batch, C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self._lib.aeth_fft_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def len(self):                                                # fft.rs:232-234
        return self._lib.aeth_fft_len(self.h)

    @property
    def algorithm(self):
        return self._lib.aeth_fft_algorithm(self.h).decode()

    # ---- generic exec with explicit sign (the C ABI's shape) ----
    def exec(self, inp, out, sign, s=Scale.NONE):
        if _is_dev(inp):
            n = self.len()
            if out.n != inp.n:
                raise _lib.LengthMismatch(_lib.E_LEN, "Output and FFT must be the same length")
            # batch = whole frames in the slice; a ragged slice fails the C side's length assert
            check(self._lib.aeth_fft_exec(self.h, inp._p(), inp.n, out._p(), inp.n // n, sign, s.kind, s.x))
            return out
        assert inp.dtype == np.complex64 and out.dtype == np.complex64
        check(self._lib.aeth_fft_exec_host(self.h, inp.ctypes.data_as(C.c_void_p), inp.size,
                                           out.ctypes.data_as(C.c_void_p), out.size, sign, s.kind, s.x))
        return out

    # ---- trait Fft ----
    def fwd(self, inp, out, s):                                   # fft.rs:51, :162-171
        return self.exec(inp, out, SIGN_REF_FWD, s)

    def bwd(self, inp, out, s):                                   # fft.rs:55, :173-182
        return self.exec(inp, out, SIGN_REF_BWD, s)

    def ifwd(self, inp, s):                                       # fft.rs:59, :184-193
        return self.exec(inp, inp, SIGN_REF_FWD, s)

    def ibwd(self, inp, s):                                       # fft.rs:63, :195-204
        return self.exec(inp, inp, SIGN_REF_BWD, s)

    def rfft_mirror(self, frames, s=Scale.NONE, out=None):
        """every frame: vec_rfft(self, s) then vec_mirror() (util/plot.rs:59-61), one call; in place unless out="""
        out = frames if out is None else out
        n = self.len()
        check(self._lib.aeth_fft_exec_mirrored(self.h, frames._p(), frames.n, out._p(), frames.n // n if n else 0,
                                               SIGN_REF_FWD, s.kind, s.x))
        return out

    def rfft_interpolate(self, frames, dst, n_between, s=Scale.NONE, compat_im=True):
        """every frame: vec_rfft(self, s), then sampling::interpolate(frame, dst, n_between) (sampling.rs:7-24) appended
        frame after frame into dst; `frames` stays as it was.  Returns the number of samples written."""
        n = self.len()
        w = C.c_size_t(0)
        check(self._lib.aeth_fft_exec_interpolate(self.h, frames._p(), frames.n, frames.n // n if n else 0, SIGN_REF_FWD,
                                                  s.kind, s.x, dst._p(), dst.n, n_between, 1 if compat_im else 0, C.byref(w)))
        return w.value

    def _tmp(self, inp, sign, s):
        view = C.c_void_p()
        if _is_dev(inp):
            n = self.len()
            check(self._lib.aeth_fft_exec_tmp(self.h, inp._p(), inp.n, inp.n // n if n else 0, sign, s.kind, s.x,
                                              C.byref(view)))
            return DeviceVec(self.ctx, inp.n, ptr=view.value, owner=self)   # borrow of the plan's temp
        check(self._lib.aeth_fft_exec_tmp_host(self.h, inp.ctypes.data_as(C.c_void_p), inp.size, sign, s.kind, s.x,
                                               C.byref(view)))
        buf = (C.c_float * (2 * self.len())).from_address(view.value)   # len() complex values
        return np.frombuffer(buf, dtype=np.complex64)             # valid until the next call on this plan

    def tfwd(self, inp, s):                                       # fft.rs:68, :206-217
        return self._tmp(inp, SIGN_REF_FWD, s)

    def tbwd(self, inp, s):                                       # fft.rs:73, :219-230
        return self._tmp(inp, SIGN_REF_BWD, s)

    # ---- benches/benches.rs:410-416 fused: frames.vec_rfft(s).vec_mul(sig).vec_rifft(s) ----
    def mul_chain(self, frames, sig, s_fwd=Scale.NONE, s_bwd=Scale.NONE):
        n = self.len()
        check(self._lib.aeth_fft_mul_ifft(self.h, frames._p(), frames.n, frames.n // n if n else 0, sig._p(), sig.n,
                                          s_fwd.kind, s_fwd.x, s_bwd.kind, s_bwd.x))
        return frames
