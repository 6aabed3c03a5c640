"""Awgn (reference: src/noise.rs:1-84) on device-resident signals.

The noise power semantics follow the reference (amplitude proportional to `power`,
noise.rs:41-42,58); the random stream is the library's own counter-based generator."""
import numpy as np

from . import _lib
from ._lib import check

DEFAULT_RNG_SEED = 815                       # noise.rs:6


class Awgn:
    def __init__(self, ctx, power=1.0, seed=DEFAULT_RNG_SEED):     # noise.rs:29-37
        self.ctx, self.power, self.seed = ctx, float(np.float32(power)), int(seed)
        self.offset = 0                      # position in the stream, advances like the reference's generator state
        self._lib = _lib.load()

    def set_power(self, power):              # noise.rs:47-50
        self.power = float(np.float32(power))

    def apply(self, signal):                 # noise.rs:53-59
        check(self._lib.aeth_awgn_apply(self.ctx.h, signal._p(), signal.n, self.power, self.seed, self.offset))
        self.offset += signal.n
        return signal


    def fill(self, target):                  # noise.rs:61-65 (capacity = the vector's length)
        check(self._lib.aeth_awgn_fill(self.ctx.h, target._p(), target.n, self.power, self.seed, self.offset))
        self.offset += target.n
        return target

    def iter(self, chunk=4096):              # noise.rs:67-84: an endless stream of next()
        """Generator of cf32 noise samples, drawn from the device `chunk` at a time."""
        buf = self.ctx.empty(chunk)
        while True:
            for v in self.fill(buf).to_host():
                yield v


def philox4x32_10(ctx, counters, keys):
    return philox4x32(ctx, counters, keys, 10)


def philox4x32(ctx, counters, keys, rounds=7):
    """The generator's integer stage on the device (7 rounds; 10 for the other published known answers): counters
    (n, 4) and keys (n, 2) uint32 -> (n, 4) uint32."""
    import ctypes as C
    c = np.ascontiguousarray(counters, np.uint32).reshape(-1, 4); k = np.ascontiguousarray(keys, np.uint32).reshape(-1, 2)
    n = c.shape[0]
    ck = np.ascontiguousarray(np.concatenate([c, k], axis=1))
    lib = _lib.load()
    din, dout = C.c_void_p(), C.c_void_p()
    check(lib.aeth_dev_alloc(ctx.h, ck.nbytes, C.byref(din))); check(lib.aeth_dev_alloc(ctx.h, 16 * n, C.byref(dout)))
    try:
        check(lib.aeth_upload(ctx.h, din, ck.ctypes.data_as(C.c_void_p), ck.nbytes))
        check(lib.aeth_rng_philox4x32(ctx.h, din, n, rounds, dout))
        out = np.empty((n, 4), np.uint32)
        check(lib.aeth_download(ctx.h, out.ctypes.data_as(C.c_void_p), dout, out.nbytes))
    finally:
        lib.aeth_dev_free(ctx.h, din); lib.aeth_dev_free(ctx.h, dout)
    return out


def normal_pairs(ctx, a, b):
    """The generator's floating-point stage on the device: word pairs (a[i], b[i]) uint32 -> complex64 standard normals."""
    import ctypes as C
    ab = np.ascontiguousarray(np.stack([np.asarray(a, np.uint32), np.asarray(b, np.uint32)], axis=1))
    n = ab.shape[0]
    lib = _lib.load()
    din, dout = C.c_void_p(), C.c_void_p()
    check(lib.aeth_dev_alloc(ctx.h, max(ab.nbytes, 8), C.byref(din))); check(lib.aeth_dev_alloc(ctx.h, max(8 * n, 8), C.byref(dout)))
    try:
        out = np.empty(n, np.complex64)
        if n:
            check(lib.aeth_upload(ctx.h, din, ab.ctypes.data_as(C.c_void_p), ab.nbytes))
            check(lib.aeth_rng_normal_pairs(ctx.h, din, n, dout))
            check(lib.aeth_download(ctx.h, out.ctypes.data_as(C.c_void_p), dout, out.nbytes))
    finally:
        lib.aeth_dev_free(ctx.h, din); lib.aeth_dev_free(ctx.h, dout)
    return out


def generator(ctx):                          # noise.rs:8-11
    return Awgn(ctx, 1.0, DEFAULT_RNG_SEED)


def new(ctx, power, seed):                   # noise.rs:13-16
    return Awgn(ctx, power, seed)
