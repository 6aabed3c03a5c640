#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-slice FIR pipeline (aeth_fir_stream_host): 64 Mi / 256 Mi samples, several chunk
sizes; output checked against the device-resident one-shot run (bit-identical)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from bench import lowpass_taps, FFT_LEN

ctx = ap.Context(0)
fir = ap.Fir(ctx, lowpass_taps(), FFT_LEN)
for n in (64 << 20, 256 << 20):
    rng = np.random.default_rng(5)
    x = rng.standard_normal(2 * n, dtype=np.float32).view(np.complex64)
    ref = None
    if n == 64 << 20:
        ref = fir.filter(ctx.vec(x)).to_host()
    for chunk in (1 << 20, 2 << 20, 4 << 20, 8 << 20, 16 << 20):
        best = None
        for rep in range(3):
            y, st = fir.filter_stream(x, chunk=chunk)
            if best is None or st["seconds"] < best["seconds"]: best = st
        same = "" if ref is None else ("  bit-identical" if np.array_equal(y.view(np.uint32), ref.view(np.uint32)) else "  MISMATCH")
        if chunk == 4 << 20:
            for l in fir.filter_stream(x, chunk=chunk, report=True)[1]["lines"]: print("      " + l)
        print(f"n = {n >> 20:4d} Mi  chunk {chunk >> 20:3d} Mi samples ({chunk >> 17:4d} MiB)  {best['seconds'] * 1e3:8.2f} ms  "
              f"{n / best['seconds'] / 1e9:5.2f} GS/s  {8 * n / best['seconds'] / 1e9:5.1f} GB/s per direction  pinned={int(best['pinned'])}{same}", flush=True)
