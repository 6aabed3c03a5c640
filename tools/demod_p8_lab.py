#!/usr/bin/env python3
"""Lab (needs `make -C aether_primitives_amd/csrc LAB=1`): the fused correlate + demod kernel of N = 2048 built with 256
lanes x 8 points (radices 8.8.8.4, three exchanges per transform) instead of 128 lanes x 16 points (16.16.8, two
exchanges) -- 146 registers instead of 248, so three waves per SIMD instead of two, or four with the register count
forced to 128 (2 spills).  The kernel is bound by its transform time at two waves per SIMD: does the
occupancy buy more than the third exchange costs?  One process per setting (AETH_DEMOD_P8 = 0 / 1 / 4); each times the
call on 16384 frames, alone and alternating over the context's two queues, and compares its bits with the oracle's on
64 frames.   python3 tools/demod_p8_lab.py  ->  gpurun_out/demod_p8_lab.txt"""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import numpy as np
    import aether_primitives_amd as ap
    from aether_primitives_amd import modulation
    from oracle import pyoracle as orc
    ctx = ap.Context(0)
    N, frames = 2048, 16384
    q = modulation.qpsk(ctx)
    f = ap.HipFft(ctx, N, max_batch=frames)
    rng = np.random.default_rng(5)
    sigh = (0.3 * (rng.standard_normal(N) + 1j * rng.standard_normal(N))).astype(np.complex64)
    sig = ctx.vec(sigh)
    nb = 3
    x = [ctx.vec((rng.standard_normal(N * frames) + 1j * rng.standard_normal(N * frames)).astype(np.complex64)) for _ in range(nb)]
    out = [modulation.DeviceBits(ctx, 2 * N * frames) for _ in range(nb)]
    res = []
    for lane in (False, True):
        ctx.set_overlap(lane)
        e0, e1 = ctx.event(), ctx.event()
        for i in range(30): q.correlate_demod(f, x[i % nb], sig, out=out[i % nb])
        ts = []
        for r in range(5):
            ctx.sync(); e0.record()
            for i in range(60): q.correlate_demod(f, x[i % nb], sig, out=out[i % nb])
            e1.record(); ctx.sync(); ts.append(e0.elapsed_ms(e1) / 60 * 1e3)
        res.append(min(ts))
    ctx.set_overlap(False)
    small = x[0].slice(0, 64 * N)
    got = q.correlate_demod(f, small, sig).to_host()
    xs = small.to_host()
    want = orc.demod_naive(orc.correlate_frames(sigh, xs), 2, compat=True)
    diff = int((got != want).sum())
    print(f"AETH_DEMOD_P8={os.environ.get('AETH_DEMOD_P8', '0'):2s}   one queue {res[0]:6.1f} us   two queues {res[1]:6.1f} us   per 2^25 samples;"
          f"   bits differing from the oracle's chain on 64 frames: {diff} of {got.size}", flush=True)
    sys.exit(0)

for v in ("0", "1", "4", "0", "1", "4"):
    e = dict(os.environ, AETH_TUNING="1", AETH_LAB_LIB="1", AETH_DEMOD_P8=v)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=e, cwd=ROOT)
