#!/usr/bin/env python3
"""Fixed cost of a timed region of K chained C3 launches (two-queue overlap lane vs one queue): event and wall time
for K = 1 .. 200 from an idle, synchronised device, best of 5.  T(K) = a K + b; b is what a K = 20 run pays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from aether_primitives_amd._lib import check
from bench import lowpass_taps, FFT_LEN, STREAM, synth_stream

ctx = ap.Context(0)
fir = ap.Fir(ctx, lowpass_taps(), FFT_LEN)
ns = 6
ins = [ctx.vec(synth_stream(100 + s, STREAM)) for s in range(ns)]
outs = [ctx.empty(STREAM) for s in range(ns)]
ex = fir._lib.aeth_fir_exec
args = [(fir.h, None, ins[k]._p(), STREAM, outs[k]._p()) for k in range(ns)]
idle_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
for overlap in (True, False):
    ctx.set_overlap(overlap)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        for i in range(64): check(ex(*args[i % ns]))
        ctx.sync()
    rows = []
    for K in (1, 2, 4, 10, 20, 40, 100, 200):
        best_ev, best_wall = 1e9, 1e9
        for rep in range(5):
            for i in range(8): check(ex(*args[i % ns]))
            ctx.sync()
            if idle_ms: time.sleep(idle_ms * 1e-3)
            e0, e1 = ctx.event(), ctx.event()
            t0 = time.perf_counter()
            e0.record()
            for i in range(K): check(ex(*args[i % ns]))
            e1.record(); ctx.sync()
            wall = (time.perf_counter() - t0) * 1e6
            ev = e0.elapsed_ms(e1) * 1e3
            best_ev = min(best_ev, ev); best_wall = min(best_wall, wall)
        rows.append((K, best_ev, best_wall))
    a = (rows[-1][1] - rows[-2][1]) / (rows[-1][0] - rows[-2][0])
    print(f"{'two queues' if overlap else 'one queue '}  idle before region {idle_ms} ms   slope {a:.2f} us/launch")
    for K, ev, wall in rows:
        print(f"   K = {K:3d}   events {ev:9.1f} us ({ev / K:6.2f}/launch, fixed {ev - a * K:6.1f})   wall {wall:9.1f} us (fixed {wall - a * K:6.1f})")
