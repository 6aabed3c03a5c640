"""sampling::{interpolate, downsample, downsample_sb} (reference: src/sampling.rs:7-62)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check
from .context import DeviceVec


def interpolate(ctx, src, dst, n_between, compat_im=True, frame_len=None):
    """Linear interpolation, `n_between` samples between each pair (sampling.rs:7-24).

    Host flavour: `src` numpy complex64, `dst` a python list-like "Vec" that is
    APPENDED to (as the reference does) -- pass a list, get it extended.
    Device flavour: `src`/`dst` DeviceVec; writes from dst[0]; returns count.
    `compat_im` reproduces the reference's `im: x1.re + i*rate.1` (sampling.rs:19).
    """
    lib = _lib.load()
    n_written = C.c_size_t()
    if isinstance(src, DeviceVec):
        if frame_len is None:
            check(lib.aeth_interpolate(ctx.h, src._p(), src.n, dst._p(), dst.n, n_between, int(compat_im),
                                       C.byref(n_written)))
        else:
            check(lib.aeth_interpolate_frames(ctx.h, src._p(), frame_len, src.n // frame_len, dst._p(), dst.n,
                                              n_between, int(compat_im), C.byref(n_written)))
        return n_written.value
    src = np.ascontiguousarray(src, dtype=np.complex64)
    need = src.size + (src.size - 1) * n_between if src.size else 0
    tmp = np.empty(max(need, 1), np.complex64)
    check(lib.aeth_host_interpolate(ctx.h, src.ctypes.data_as(C.c_void_p), src.size,
                                    tmp.ctypes.data_as(C.c_void_p), tmp.size, n_between, int(compat_im),
                                    C.byref(n_written)))
    dst.extend(tmp[:n_written.value].tolist())
    return n_written.value


def _downsample(ctx, src, dst, release, step_by):
    lib = _lib.load()
    if isinstance(src, DeviceVec):
        if release:
            check(lib.aeth_downsample_release(ctx.h, src._p(), src.n, dst._p(), dst.n, 8, int(step_by)))
        else:
            check(lib.aeth_downsample(ctx.h, src._p(), src.n, dst._p(), dst.n, 8))
        return dst
    assert src.dtype == dst.dtype and src.flags["C_CONTIGUOUS"] and dst.flags["C_CONTIGUOUS"]
    sp, dp = src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p)
    if release:
        check(lib.aeth_host_downsample_release(ctx.h, sp, src.size, dp, dst.size, src.dtype.itemsize, int(step_by)))
    else:
        check(lib.aeth_host_downsample(ctx.h, sp, src.size, dp, dst.size, src.dtype.itemsize))
    return dst


def downsample(ctx, src, dst, release=False):
    """dst[i] = src[i * (len(src)/len(dst))] for any Copy element type (sampling.rs:28-42).
    release=False is the reference's debug build (uneven sizes panic: debug_assert_eq!, :32-36); release=True its
    release build, where the assert is compiled out and the ratio floors (benches/benches.rs:113: 8096 -> 512)."""
    return _downsample(ctx, src, dst, release, False)


def downsample_sb(ctx, src, dst, release=False):
    """The step_by variant (sampling.rs:49-62): same results; a release build panics on step_by(0) when src is shorter
    than dst, where `downsample` broadcasts src[0]."""
    return _downsample(ctx, src, dst, release, True)
