"""trait Modulation for the generic BPSK / QPSK tables (reference: src/modulation.rs:5-149),
device-resident.  Bits are one byte per bit, as in the reference."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check
from .context import DeviceVec

GENERIC_BPSK_TABLE = np.array([1 + 1j, -1 - 1j], np.complex64)                    # modulation.rs:77
GENERIC_QPSK_TABLE = np.array([1 + 1j, -1 + 1j, 1 - 1j, -1 - 1j], np.complex64)   # modulation.rs:87-92


class DeviceBits:
    """u8-per-bit buffer in HBM."""

    def __init__(self, ctx, n, host=None):
        self.ctx, self.n = ctx, int(n)
        self.ptr = ctx.alloc(max(self.n, 1))
        if host is not None:
            ctx.upload(self.ptr, np.ascontiguousarray(host, np.uint8))

    def to_host(self):
        out = np.empty(self.n, np.uint8)
        if self.n:
            self.ctx.download(self.ptr, out)
        return out

    def __del__(self):
        try:
            if self.ptr and self.ctx.h:
                self.ctx.free(self.ptr); self.ptr = None
        except Exception:
            pass


class _Modulation:
    BITS_PER_SYMBOL = 0
    table = None

    def __init__(self, ctx, table=None):
        self.ctx = ctx
        self._lib = _lib.load()
        if table is not None:
            self.table = np.ascontiguousarray(table, np.complex64)

    def bits_per_symbol(self):                                  # modulation.rs:146-148
        return self.BITS_PER_SYMBOL

    def symbol(self, idx):                                      # modulation.rs:13-15 / :26-28
        return self.table[idx]

    def modulate(self, bits, out=None):                         # modulation.rs:115-121 (out= : modulate_into, :124-131)
        if not isinstance(bits, DeviceBits):
            bits = DeviceBits(self.ctx, len(bits), bits)
        if out is None:
            out = DeviceVec(self.ctx, bits.n // self.BITS_PER_SYMBOL)
        check(self._lib.aeth_modulate(self.ctx.h, C.c_void_p(bits.ptr), bits.n, self.BITS_PER_SYMBOL,
                                      self.table.ctypes.data_as(C.c_void_p), out._p(), out.n))
        return out

    def modulate_awgn(self, bits, awgn, out=None):              # examples/modem.rs:19-26: modulate, then awgn.apply, one pass
        if not isinstance(bits, DeviceBits):
            bits = DeviceBits(self.ctx, len(bits), bits)
        if out is None:
            out = DeviceVec(self.ctx, bits.n // self.BITS_PER_SYMBOL)
        check(self._lib.aeth_modulate_awgn(self.ctx.h, C.c_void_p(bits.ptr), bits.n, self.BITS_PER_SYMBOL,
                                           self.table.ctypes.data_as(C.c_void_p), out._p(), out.n, awgn.power, awgn.seed,
                                           awgn.offset))
        awgn.offset += out.n
        return out

    def correlate_demod(self, fft, frames, sig, s_fwd=None, s_bwd=None, compat=True, out=None):
        """frames.vec_rfft(fft).vec_mul(sig).vec_rifft(fft) per frame, then demod_naive: only the bits are written
        (examples/modem.rs:28-31 behind the correlator of benches.rs:410-416); `frames` stays as it was."""
        from .fft import Scale
        s_fwd = s_fwd or Scale.NONE; s_bwd = s_bwd or Scale.NONE
        if out is None:
            out = DeviceBits(self.ctx, frames.n * self.BITS_PER_SYMBOL)
        n = fft.len()
        check(self._lib.aeth_fft_mul_ifft_demod(fft.h, frames._p(), frames.n, frames.n // n if n else 0, sig._p(), sig.n,
                                                s_fwd.kind, s_fwd.x, s_bwd.kind, s_bwd.x, self.BITS_PER_SYMBOL,
                                                self.table.ctypes.data_as(C.c_void_p), C.c_void_p(out.ptr), out.n,
                                                1 if compat else 0))
        return out

    def demod_naive(self, symbols, compat=True, out=None):      # modulation.rs:33-56 / :133-144
        if out is None:
            out = DeviceBits(self.ctx, symbols.n * self.BITS_PER_SYMBOL)
        check(self._lib.aeth_demod_naive(self.ctx.h, symbols._p(), symbols.n, self.BITS_PER_SYMBOL,
                                         self.table.ctypes.data_as(C.c_void_p), C.c_void_p(out.ptr), out.n,
                                         1 if compat else 0))
        return out


class Bpsk(_Modulation):
    BITS_PER_SYMBOL = 1
    table = GENERIC_BPSK_TABLE


class Qpsk(_Modulation):
    BITS_PER_SYMBOL = 2
    table = GENERIC_QPSK_TABLE


def table(ctx, symbols):
    """Any implementor of trait Modulation backed by a table of 2^k symbols, k = 1 .. 8: the trait's default
    index / modulate / demod_naive (modulation.rs:94-149)."""
    symbols = np.ascontiguousarray(symbols, np.complex64)
    k = int(symbols.size).bit_length() - 1
    if symbols.size != 1 << k or not 1 <= k <= 8:
        raise ValueError("a symbol table holds 2, 4, ... 256 entries")
    cls = type(f"Table{symbols.size}", (_Modulation,), {"BITS_PER_SYMBOL": k})
    return cls(ctx, symbols)


def bpsk(ctx):                                                  # modulation.rs:61-63
    return Bpsk(ctx)


def qpsk(ctx):                                                  # modulation.rs:66-68
    return Qpsk(ctx)
