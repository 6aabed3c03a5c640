#!/usr/bin/env python3
"""Tuning helper: A/B the fused FIR kernel (C3 geometry) under several settings of
the AETH_FIR_* knobs, interleaved rounds in ONE process (cdna guide rule 24).
usage: tune_fir.py [steps] [stream_len] -- "K=V K=V" "K=V" ...   (each quoted string = one variant)"""
import os, sys, statistics
os.environ.setdefault('AETH_TUNING', '1')   # enables the library's AETH_* tuning knobs
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aether_primitives_amd as ap
from bench import synth_stream, lowpass_taps, STREAM

args = sys.argv[1:]
variants = [""]
if "--" in args:
    i = args.index("--"); variants = args[i + 1:] or [""]; args = args[:i]
steps = int(args[0]) if len(args) > 0 else 40
n = int(args[1]) if len(args) > 1 else STREAM
ctx = ap.Context(0)
fir = ap.Fir(ctx, lowpass_taps(), 2048)
ns = 5
ins = [ctx.vec(synth_stream(815 + i, n)) for i in range(ns)]
outs = [ctx.empty(n) for _ in range(ns)]
e0, e1 = ctx.event(), ctx.event()
res = {v: [] for v in variants}
KEYS = ["AETH_FIR_GRID_FIRST", "AETH_FIR_GRID_CHAINED", "AETH_FIR_SPREAD", "AETH_NT"]     # the knobs the library still reads (aether_hip.h)
for rnd in range(6):
    for v in variants:
        for k in KEYS: os.environ.pop(k, None)
        for kv in v.split():
            k, val = kv.split("="); os.environ[k] = val
        for i in range(4): fir.filter(ins[i % ns], out=outs[i % ns])
        ctx.sync(); e0.record()
        for i in range(steps): fir.filter(ins[i % ns], out=outs[i % ns])
        e1.record(); ctx.sync()
        res[v].append(e0.elapsed_ms(e1) / steps)
for v in variants:
    r = res[v][1:]
    med, mn = statistics.median(r), min(r)
    print(f"[{v or 'default':40s}] n={n} median {med*1e3:6.1f} us  min {mn*1e3:6.1f} us  {n/med/1e6:6.1f} GS/s  {16*n/med/1e6/8000*100:5.1f}% of 8TB/s")
