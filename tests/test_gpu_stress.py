"""GPU: seeded randomized sweep over lengths, batches, signs, scales, in/out-of-place and
FIR geometries -- every case against f64 truth (aggregate EVM <= -120 dB)."""
import numpy as np
import pytest

import aether_primitives_amd as ap
from aether_primitives_amd import HipFft, Fir, Scale
from helpers import bits_equal, rand_c64

pytestmark = pytest.mark.gpu


def test_fft_random_sweep(ctx, oracle):
    rng = np.random.default_rng(2026)
    pool = ([2 ** k for k in range(1, 15)] + [3, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 18, 20, 21, 24, 25, 27, 30, 36,
            48, 50, 63, 75, 96, 100, 125, 144, 200, 243, 250, 343, 360, 500, 512 * 3, 625, 729, 1000, 1331, 2000,
            2187, 2401, 3000, 3375, 4000, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 101, 257, 641, 4097])
    plans = {}
    for it in range(120):
        n = int(rng.choice(pool))
        batch = int(rng.choice([1, 2, 3, 5, 8, 17, 64, 129]))
        if n * batch > (1 << 21):
            batch = max(1, (1 << 21) // n)
        sign = int(rng.choice([-1, 1]))
        s = [Scale.NONE, Scale.SN, Scale.N, Scale.X(0.5)][int(rng.integers(4))]
        f = plans.setdefault(n, HipFft(ctx, n))
        x = rand_c64(1000 + it, n * batch)
        d = ctx.vec(x)
        if rng.integers(2):
            f.exec(d, d, sign, s); got = d.to_host()
        else:
            o = ctx.empty(x.size); f.exec(d, o, sign, s); got = o.to_host()
            assert bits_equal(d.to_host(), x)
        truth = oracle.fft_f64_frames(x.astype(np.complex128), n, sign) * float(s.factor(n))
        e = oracle.evm_db(got, truth)
        assert e <= -120.0, (n, batch, sign, repr(s), f.algorithm, e)


def test_fir_random_sweep(ctx, oracle):
    rng = np.random.default_rng(7)
    for it in range(40):
        fft_len = int(rng.choice([16, 32, 64, 128, 256, 512, 1024, 2048, 4096]))
        ntaps = int(rng.integers(1, fft_len // 2 + 1))
        n = int(rng.integers(1, 60000))
        h = rand_c64(it, ntaps, scale=1.0 / np.sqrt(ntaps))
        x = rand_c64(500 + it, n)
        f = Fir(ctx, h, fft_len)
        use_hist = bool(rng.integers(2)) and ntaps > 1
        hist = rand_c64(900 + it, ntaps - 1) if use_hist else None
        y = f.filter(ctx.vec(x), hist=ctx.vec(hist) if use_hist else None).to_host()
        truth = oracle.fir_direct_f64(h, x, hist=hist)
        if n >= 4 * ntaps:
            e = oracle.evm_db(y, truth)
            assert e <= -120.0, (fft_len, ntaps, n, use_hist, e)
        else:
            bound = 8 * 2.0 ** -23 * float(np.abs(np.concatenate([x, hist if use_hist else x[:0]])).max()) * float(np.abs(h).sum())
            assert np.abs(y - truth).max() <= bound, (fft_len, ntaps, n)


def test_two_host_threads_two_contexts(oracle):
    """SURVEY 8b threading: a context/plan is not thread-safe but distinct contexts run concurrently from
    distinct host threads (each has its own stream, plans, staging and thread-local error text)."""
    import threading
    taps = oracle.synth_lowpass_taps(64, 0.25)
    x = [rand_c64(900 + t, (1 << 18) + 77 * t) for t in range(2)]
    want = []
    c0 = ap.Context(0)
    for t in range(2):                                       # single-threaded reference run
        f = ap.Fir(c0, taps, 2048); h = ap.HipFft(c0, 1024)
        y = f.filter(c0.vec(x[t])).to_host()
        m = (x[t].size // 1024) * 1024
        s = c0.vec(x[t][:m]); h.ifwd(s, ap.Scale.SN)
        want.append((y, s.to_host()))
    got, errs = [None, None], []

    def work(t):
        try:
            c = ap.Context(0)
            f = ap.Fir(c, taps, 2048); h = ap.HipFft(c, 1024)
            m = (x[t].size // 1024) * 1024
            for _ in range(20):
                y = f.filter(c.vec(x[t])).to_host()
                s = c.vec(x[t][:m]); h.ifwd(s, ap.Scale.SN)
                sp = s.to_host()
                with pytest.raises(ap.LengthMismatch):       # error text is per thread
                    c.vec(x[t][:5]).vec_add(c.vec(x[t][:4]))
            got[t] = (y, sp)
            c.close()
        except Exception as e:                               # noqa: BLE001 - reported below
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    for t in range(2):
        assert bits_equal(got[t][0], want[t][0]) and bits_equal(got[t][1], want[t][1])
    c0.close()


def test_plans_release_their_device_memory(ctx, oracle):
    """create / run / destroy every kind of plan many times: free device memory returns to where it was"""
    import ctypes
    import gc
    hip = ctypes.CDLL("libamdhip64.so")                      # the runtime libaether_hip.so already loaded

    def free_bytes():
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipDeviceSynchronize() == 0
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value
    taps = oracle.synth_lowpass_taps(64, 0.25)

    def cycle():
        c = ap.Context(0)
        for n in (2048, 1000, 6000, 65536, 4099, 8192):          # pow2, mixed, mixed in place, four-step, bluestein, 8192
            f = HipFft(c, n, max_batch=4)
            x = c.vec(rand_c64(n, 4 * n)); f.ifwd(x, Scale.SN); f.ibwd(x, Scale.SN)
            del f, x
        fir = Fir(c, taps, 2048)
        y = fir.filter(c.vec(rand_c64(5, 1 << 16)))
        out, _ = fir.filter_stream(rand_c64(6, 1 << 16))
        del fir, y, out
        c.sync(); c.close()
        gc.collect()

    cycle()
    free0 = free_bytes()
    for _ in range(20):
        cycle()
    free1 = free_bytes()
    assert free0 - free1 < (64 << 20), f"{(free0 - free1) >> 20} MiB lost over 20 create/destroy cycles"
