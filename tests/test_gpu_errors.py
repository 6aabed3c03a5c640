"""GPU: argument validation and edge cases of the C ABI (the reference panics; the ABI returns codes)."""
import ctypes as C

import numpy as np
import pytest

import aether_primitives_amd as ap
from aether_primitives_amd import _lib, HipFft, Fir, Scale
from helpers import bits_equal, rand_c64

pytestmark = pytest.mark.gpu


def test_fft_plan_limits(ctx):
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.aeth_fft_create(ctx.h, 0, 1, C.byref(h)) == _lib.E_ARG
    assert lib.aeth_fft_create(ctx.h, (1 << 24) + 1, 1, C.byref(h)) == _lib.E_UNSUPPORTED
    assert lib.aeth_fft_create(None, 8, 1, C.byref(h)) == _lib.E_ARG
    f = HipFft(ctx, 1)                                           # len 1: identity (times scale)
    x = rand_c64(1, 5)
    d = ctx.vec(x); f.ifwd(d, Scale.X(2.0))
    assert bits_equal(d.to_host(), (x * np.float32(2)).astype(np.complex64))


def test_empty_inputs_are_noops(ctx):
    e = ctx.empty(0)
    for op in ("vec_conj", "vec_mirror", "vec_zero"):
        getattr(e, op)()
    e.vec_scale(2.0); e.vec_add(ctx.empty(0)); e.vec_mul(ctx.empty(0))
    f = HipFft(ctx, 64)
    f.exec(ctx.empty(0), ctx.empty(0), +1)                        # batch 0
    fir = Fir(ctx, np.ones(3, np.complex64), 16)
    assert fir.filter(ctx.empty(0)).n == 0
    assert fir.filter(np.zeros(0, np.complex64)).size == 0


def test_bad_arguments(ctx):
    lib = _lib.load()
    f = HipFft(ctx, 64)
    v = ctx.vec(rand_c64(1, 64))
    assert lib.aeth_fft_exec(f.h, v._p(), 64, v._p(), 1, 0, 0, 0.0) == _lib.E_ARG       # sign must be +-1
    assert lib.aeth_fft_exec(f.h, v._p(), 64, v._p(), 1, 1, 7, 0.0) == _lib.E_ARG       # scale kind
    assert lib.aeth_fft_exec(f.h, None, 64, v._p(), 1, 1, 0, 0.0) == _lib.E_ARG         # null pointer
    assert lib.aeth_fft_exec(f.h, C.c_void_p(v.ptr + 4), 64, v._p(), 1, 1, 0, 0.0) == _lib.E_ALIGN
    assert lib.aeth_vec_scale(ctx.h, C.c_void_p(v.ptr + 4), 1, 1.0) == _lib.E_ALIGN
    assert b"aligned" in lib.aeth_last_error()
    assert lib.aeth_downsample(ctx.h, v._p(), 64, v._p(), 8, 3) == _lib.E_ARG           # elem_size 3
    # 0 % n == 0 passes the reference's divisibility assert; dec = 0 then indexes an empty slice and panics (sampling.rs:39-41)
    assert lib.aeth_downsample(ctx.h, v._p(), 0, v._p(), 8, 8) == _lib.E_LEN
    assert b"empty src" in lib.aeth_last_error()
    fd = Fir(ctx, np.ones(8, np.complex64), 2048)
    assert lib.aeth_fir_exec_decim(fd.h, None, v._p(), 0, v._p(), 4) == _lib.E_LEN
    assert lib.aeth_modulate(ctx.h, v._p(), 4, 3, None, v._p(), 1) == _lib.E_ARG          # 3 bits per symbol: needs a table
    assert lib.aeth_modulate(ctx.h, v._p(), 9, 9, v._p(), v._p(), 1) == _lib.E_UNSUPPORTED  # 9 bits per symbol
    with pytest.raises(ap.AetherError):
        Fir(ctx, np.zeros(0, np.complex64), 64)
    with pytest.raises(ap.AetherError):
        ap.Context(99)


def test_minimal_fir_geometry(ctx, oracle):
    x = rand_c64(5, 1000)
    for taps, n in ((np.array([2 - 1j], np.complex64), 2), (rand_c64(1, 2), 4), (rand_c64(2, 8), 16)):
        y = Fir(ctx, taps, n).filter(ctx.vec(x)).to_host()
        assert oracle.evm_db(y, oracle.fir_direct_f64(taps, x)) <= -120


def test_two_contexts_are_independent(ctx):
    c2 = ap.Context(0)
    a, b = ctx.vec(rand_c64(1, 1 << 16)), c2.vec(rand_c64(2, 1 << 16))
    for _ in range(10):
        a.vec_scale(1.0); b.vec_conj().vec_conj()
    ctx.sync(); c2.sync()
    assert bits_equal(a.to_host(), rand_c64(1, 1 << 16)) and bits_equal(b.to_host(), rand_c64(2, 1 << 16))
    del b
    c2.close()


def test_borrowed_stream_context(ctx):
    """aeth_ctx_create_on_stream: run on a caller's stream (here: another context's) without owning it."""
    c2 = ap.Context(0, stream=ctx.stream)
    v = c2.vec(rand_c64(3, 4096)); v.vec_conj()
    c2.sync()
    assert bits_equal(v.to_host(), np.conj(rand_c64(3, 4096)))
    del v
    c2.close()
    ctx.vec(rand_c64(1, 8)).vec_conj(); ctx.sync()               # the lender's stream is still alive


def test_plan_temps_live_on_the_plans_device():
    """A context on device 1 used from a process whose current device is 0: tfwd / tbwd and correlate + demod grow the
    plan's temp on first use, and that allocation has to land on device 1 (fft_ensure_tmp sets the plan's device)."""
    import ctypes as C
    from aether_primitives_amd import _lib, modulation
    from aether_primitives_amd.fft import Scale
    from helpers import rand_c64, bits_equal
    n = C.c_int()
    _lib.check(_lib.load().aeth_device_count(C.byref(n)))
    if n.value < 2:
        pytest.skip("needs two visible GPUs")
    c0, c1 = ap.Context(0), ap.Context(1)                      # the current device stays 0 (ctx_make restores it)
    x = rand_c64(3, 512 * 4)
    for c in (c0, c1):
        f = ap.HipFft(c, 512, max_batch=1)                     # temp sized for one frame: four frames force a regrow
        t = f.tfwd(c.vec(x), Scale.SN)
        c.sync()
    f0, f1 = ap.HipFft(c0, 512), ap.HipFft(c1, 512)
    a, b = f0.tfwd(c0.vec(x), Scale.SN), f1.tfwd(c1.vec(x), Scale.SN)
    assert bits_equal(a.to_host(), b.to_host())
    q0, q1 = modulation.qpsk(c0), modulation.qpsk(c1)
    sig = rand_c64(4, 512)
    r0 = q0.correlate_demod(f0, c0.vec(x), c0.vec(sig)).to_host()       # 512 < 1024: the chain through the plan's temp
    r1 = q1.correlate_demod(f1, c1.vec(x), c1.vec(sig)).to_host()
    assert (r0 == r1).all()
    c0.close(); c1.close()
