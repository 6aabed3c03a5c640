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


def generator(ctx):                          # noise.rs:8-11
    return Awgn(ctx, 1.0, DEFAULT_RNG_SEED)


def new(ctx, power, seed):                   # noise.rs:13-16
    return Awgn(ctx, power, seed)
