#!/usr/bin/env python3
"""Kernel time of the fused FIR vs stream length (in whole rounds of 1024 blocks):
separates the per-launch fixed cost (prologue + pipeline fill + drain) from the
steady-state rate."""
import os, sys, statistics
os.environ.setdefault('AETH_TUNING', '1')   # enables the library's AETH_* tuning knobs
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aether_primitives_amd as ap
from bench import synth_stream, lowpass_taps
ctx = ap.Context(0)
fir = ap.Fir(ctx, lowpass_taps(), 2048)
hop = fir.hop
e0, e1 = ctx.event(), ctx.event()
big = 1024 * 20 * hop
x = synth_stream(1, big)
ins = [ctx.vec(x) for _ in range(3)]
outs = [ctx.empty(big) for _ in range(3)]
prev = None
for rounds in (1, 2, 3, 4, 6, 8, 9, 12, 16, 20):
    n = 1024 * rounds * hop
    ts = []
    for rep in range(5):
        for i in range(3): fir.filter(ins[i % 3].slice(0, n), out=outs[i % 3].slice(0, n))
        ctx.sync(); e0.record()
        for i in range(20): fir.filter(ins[i % 3].slice(0, n), out=outs[i % 3].slice(0, n))
        e1.record(); ctx.sync()
        ts.append(e0.elapsed_ms(e1) / 20 * 1e3)
    t = statistics.median(ts)
    inc = "" if prev is None else f"  +{(t - prev[1]) / (rounds - prev[0]):5.2f} us/round"
    print(f"rounds={rounds:2d} blocks={1024*rounds:6d}  {t:7.2f} us  ({n / t / 1e3:6.1f} GS/s){inc}")
    prev = (rounds, t)
