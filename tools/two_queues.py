#!/usr/bin/env python3
"""What an application that keeps TWO HIP queues fed gets out of the fused FIR kernel: consecutive launches on
alternating contexts overlap the drain of one with the fill of the next.  (bench.py does not do this: its
roofline line is per kernel launch on one stream.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aether_primitives_amd as ap
from bench import synth_stream, lowpass_taps, STREAM

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ctxs = [ap.Context(0) for _ in range(nq)]
firs = [ap.Fir(c, lowpass_taps(), 2048) for c in ctxs]
# device buffers are plain pointers: allocate through context 0, use from every queue
ins = [ctxs[0].vec(synth_stream(815 + i, STREAM)) for i in range(6)]
outs = [ctxs[0].empty(STREAM) for _ in range(6)]
def step(i):
    q = i % nq
    firs[q].filter(ap.context.DeviceVec(ctxs[q], STREAM, ptr=ins[i % 6].ptr), out=ap.context.DeviceVec(ctxs[q], STREAM, ptr=outs[i % 6].ptr))
for i in range(1200): step(i)
for c in ctxs: c.sync()
t0 = time.perf_counter()
for i in range(steps): step(i)
for c in ctxs: c.sync()
el = time.perf_counter() - t0
print(f"{nq} queue(s): {el / steps * 1e6:.2f} us per 16 Mi-sample launch, {STREAM * steps / el / 1e9:.1f} GS/s, {16 * STREAM * steps / el / 8e12 * 100:.1f}% of 8 TB/s")
