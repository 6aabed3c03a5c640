#!/usr/bin/env python3
"""Per-launch time of the fused FIR kernel from a cold (idle) device: an event every G launches,
no host sync in between.  Shows the power-management transient the first tens of ms of load see.
usage: transient.py [launches] [group] [idle_seconds_before] [pre_launches]
(pre_launches > 0: run that many launches first, THEN idle, then measure: how long a gap re-arms the transient)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aether_primitives_amd as ap
from bench import synth_stream, lowpass_taps, STREAM
total = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 20
idle = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
pre = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ctx = ap.Context(0)
ins = [ctx.vec(synth_stream(815 + i, STREAM)) for i in range(4)]
outs = [ctx.empty(STREAM) for _ in range(4)]
fir = ap.Fir(ctx, lowpass_taps(), 2048)
ev = [ctx.event() for _ in range(total // G + 1)]
for i in range(pre): fir.filter(ins[i % 4], out=outs[i % 4])
ctx.sync(); time.sleep(idle)
ev[0].record()
for g in range(total // G):
    for i in range(G): fir.filter(ins[(g * G + i) % 4], out=outs[(g * G + i) % 4])
    ev[g + 1].record()
ctx.sync()
t = 0.0
line = []
for g in range(total // G):
    ms = ev[g].elapsed_ms(ev[g + 1]); t += ms
    line.append(f"{t:.1f}ms:{ms / G * 1e3:.1f}")
print(f"idle {idle}s after {pre} launches; elapsed:us/launch  " + " ".join(line[:40]))
print("... " + " ".join(line[40::max(1, (len(line) - 40) // 12)]))
