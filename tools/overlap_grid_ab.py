#!/usr/bin/env python3
"""A/B in one process, alternating: grid of the fused FIR launch under the overlap lane (AETH_FIR_OVERLAP_GRID 0 = full,
1 = three quarters, 2 = three quarters for chained launches only), regions of K = 20 and K = 200 launches from an
idle device, medians over many alternating rounds."""
import os, sys, time, statistics
os.environ['AETH_TUNING'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from aether_primitives_amd._lib import check
from bench import lowpass_taps, FFT_LEN, STREAM, synth_stream

ctx = ap.Context(0); ctx.set_overlap(True)
fir = ap.Fir(ctx, lowpass_taps(), FFT_LEN)
ns = 6
ins = [ctx.vec(synth_stream(100 + s, STREAM)) for s in range(ns)]
outs = [ctx.empty(STREAM) for s in range(ns)]
ex = fir._lib.aeth_fir_exec
args = [(fir.h, None, ins[k]._p(), STREAM, outs[k]._p()) for k in range(ns)]
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2:
    for i in range(64): check(ex(*args[i % ns]))
    ctx.sync()
res = {(m, K): [] for m in (0, 1, 2) for K in (20, 200)}
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 15):
    for K in (20, 200):
        for m in (0, 1, 2):
            os.environ['AETH_FIR_OVERLAP_GRID'] = str(m)
            for i in range(8): check(ex(*args[i % ns]))
            ctx.sync()
            t0 = time.perf_counter()
            for i in range(K): check(ex(*args[i % ns]))
            ctx.sync()
            res[(m, K)].append((time.perf_counter() - t0) * 1e6)
for K in (20, 200):
    base = statistics.median(res[(0, K)])
    for m in (0, 1, 2):
        v = statistics.median(res[(m, K)])
        print(f"K = {K:3d}  mode {m}: wall median {v:9.1f} us  ({v / K:6.2f} per launch, {STREAM * K / v / 1e3:6.1f} GS/s)  {100 * (v / base - 1):+5.2f} % vs full grid")
