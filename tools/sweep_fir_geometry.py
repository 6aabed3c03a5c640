#!/usr/bin/env python3
"""Fused FIR kernel across block geometries (fft_len x ntaps), 16 Mi samples, settled state:
checks that no (fft_len, ntaps) pair falls off the HBM-bound curve of the headline geometry."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from bench import synth_stream, STREAM
from oracle import pyoracle as orc

ctx = ap.Context(0)
n = STREAM
ins = [ctx.vec(synth_stream(815 + i, n)) for i in range(4)]
outs = [ctx.empty(n) for _ in range(4)]
e0, e1 = ctx.event(), ctx.event()
for fft_len in (64, 128, 256, 512, 1024, 2048, 4096):
    for ntaps in sorted({8, 64, fft_len // 4, fft_len // 2}):
        if 2 * ntaps > fft_len: continue
        fir = ap.Fir(ctx, orc.synth_lowpass_taps(ntaps, 0.25), fft_len)
        for i in range(600): fir.filter(ins[i % 4], out=outs[i % 4])      # settle
        ts = []
        for r in range(3):
            ctx.sync(); e0.record()
            for i in range(100): fir.filter(ins[i % 4], out=outs[i % 4])
            e1.record(); ctx.sync(); ts.append(e0.elapsed_ms(e1) / 100)
        t = statistics.median(ts)
        eff = fir.hop / fft_len
        print(f"fft_len {fft_len:5d} ntaps {ntaps:5d} hop {fir.hop:5d} ({eff*100:4.1f}% of the window is output): "
              f"{t*1e3:7.1f} us  {n/t/1e6:6.1f} GS/s  {16*n/t/1e6/80:5.1f}% of 8 TB/s", flush=True)
