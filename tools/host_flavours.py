#!/usr/bin/env python3
"""The literal host-slice trait calls (H2D -> kernel -> D2H inside one synchronous call) on large pageable slices:
ms per call and GB/s of slice bytes moved.  AETH_TUNING=1 AETH_PIN_MIN_KIB=1000000000 shows the unpinned form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from bench import lowpass_taps, FFT_LEN

ctx = ap.Context(0)
rng = np.random.default_rng(1)

def best(f, reps=4):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts[1:])

for n in (1 << 16, 1 << 20, 1 << 22, 1 << 24, 1 << 26):
    x = rng.standard_normal(2 * n, dtype=np.float32).view(np.complex64)
    y = rng.standard_normal(2 * n, dtype=np.float32).view(np.complex64)
    hx = ap.HostVec(ctx, x.copy())
    t = best(lambda: hx.vec_mul(y))
    row = [f"vec_mul {t * 1e3:8.3f} ms {24 * n / t / 1e9:5.1f} GB/s"]
    t = best(lambda: hx.vec_conj())
    row.append(f"vec_conj {t * 1e3:8.3f} ms {16 * n / t / 1e9:5.1f} GB/s")
    if n <= 1 << 24:
        f = ap.HipFft(ctx, n)
        out = np.empty_like(x)
        t = best(lambda: f.fwd(x, out, ap.Scale.NONE))
        row.append(f"fft.fwd {t * 1e3:8.3f} ms {16 * n / t / 1e9:5.1f} GB/s")
    fir = ap.Fir(ctx, lowpass_taps(), FFT_LEN)
    fo = np.empty_like(x)
    t = best(lambda: fir.filter(x, fo))
    row.append(f"fir {t * 1e3:8.3f} ms {16 * n / t / 1e9:5.1f} GB/s = {n / t / 1e9:5.2f} GS/s")
    print(f"n = 2^{n.bit_length() - 1:2d}  " + "   ".join(row), flush=True)
