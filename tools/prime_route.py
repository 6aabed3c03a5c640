#!/usr/bin/env python3
"""Lengths with a prime factor 17..61, three routes side by side: the LDS mixed-radix kernel with the generic O(r^2)
pass (AETH_MIXED_BIGR=0), the same kernel with its register butterflies for 11..23, and the route the library takes
(one-launch chirp-z kernel, or the ragged table where the length has a row).  256 MiB operands, out of place; EVM
against numpy's f64 transform."""
import os, sys, statistics
os.environ['AETH_TUNING'] = '1'
os.environ.setdefault('AETH_LAB_LIB', '1')   # these knobs exist only in the lab build: make -C aether_primitives_amd/csrc LAB=1
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from aether_primitives_amd import Scale
ctx = ap.Context(0); e0, e1 = ctx.event(), ctx.event()
lens = [int(a) for a in sys.argv[1:]] or [17, 19, 23, 34, 37, 51, 61, 68, 85, 119, 136, 187, 221, 289, 323, 391, 527, 629, 731, 901, 1003, 1156, 1411, 1700, 1938, 2006, 2047]
total = 1 << 25
for n in lens:
    batch = max(1, total // n)
    x = (np.random.default_rng(n).standard_normal(2 * n * batch, dtype=np.float32)).view(np.complex64)
    src = ctx.vec(x); dst = ctx.empty(n * batch)
    row = []
    for blu, bigr in ((0, 0), (0, 1), (1, 1)):
        os.environ['AETH_FFT_PRIME_BLU'] = str(blu)
        os.environ['AETH_MIXED_BIGR'] = str(bigr)
        f = ap.HipFft(ctx, n, max_batch=batch)
        f.fwd(src, dst, Scale.NONE); ctx.sync()
        got = dst.to_host()[: n * min(batch, 64)].reshape(-1, n)
        truth = np.fft.ifft(x[: n * min(batch, 64)].reshape(-1, n).astype(np.complex128), axis=1) * n
        evm = 20 * np.log10(np.linalg.norm(got - truth) / np.linalg.norm(truth))
        ts = []
        for r in range(4):
            e0.record()
            for i in range(10): f.fwd(src, dst, Scale.NONE)
            e1.record(); ctx.sync(); ts.append(e0.elapsed_ms(e1) / 10)
        t = statistics.median(ts[1:])
        row.append((f.algorithm, t, evm))
        del f
    cols = "   ".join(f"{r[0]:14s} {16*n*batch/r[1]/1e9:5.2f} TB/s ({r[2]:6.1f} dB)" for r in row)
    print(f"N={n:5d} batch={batch:8d}  {cols}", flush=True)
