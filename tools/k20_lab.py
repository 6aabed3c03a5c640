#!/usr/bin/env python3
"""Single-shot timed regions of K chained C3 launches, exactly as bench.py brackets them (sync | K steps | sync),
for several launch-grid policies of the overlap lane, alternating in ONE process (interleaved rounds).

  AETH_TUNING=1 python3 tools/k20_lab.py [K=20] [rounds=15]

Per configuration: median / min / mean wall time of the region and the GS/s each gives.  `first` / `chained` are the
sixteenths of the resident grid the first launch of a chain / a launch beside its predecessor takes
(AETH_FIR_GRID_FIRST / AETH_FIR_GRID_CHAINED); ev = HIP events recorded inside the region as well."""
import os, sys, time
os.environ.setdefault("AETH_TUNING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import aether_primitives_amd as ap
from aether_primitives_amd._lib import check
from bench import lowpass_taps, FFT_LEN, STREAM, synth_stream

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 15
WARM = 5
ctx = ap.Context(0)
fir = ap.Fir(ctx, lowpass_taps(), FFT_LEN)
ns = 6
ins = [ctx.vec(synth_stream(100 + s, STREAM)) for s in range(ns)]
outs = [ctx.empty(STREAM) for s in range(ns)]
ex = fir._lib.aeth_fir_exec
args = [(fir.h, None, ins[k]._p(), STREAM, outs[k]._p()) for k in range(ns)]

# (label, overlap, first, chained, events[, AETH_SYNC_SPIN_US])
configs = [
    ("2q first16 ch12", True, 16, 12, False),
    ("2q first16 ch12 spin0", True, 16, 12, False, 0),          # round 4: aeth_ctx_sync blocks on each queue instead of polling both
    ("2q first16 ch12 spin50", True, 16, 12, False, 50),
    ("2q first16 ch12 ev", True, 16, 12, True),
    ("2q first12 ch12", True, 12, 12, False),
    ("2q first10 ch12", True, 10, 12, False),
    ("2q first8  ch12", True, 8, 12, False),
    ("2q first12 ch11", True, 12, 11, False),
    ("2q first12 ch13", True, 12, 13, False),
    ("2q first14 ch12", True, 14, 12, False),
    ("1q", False, 16, 16, False),
    # oversubscribed grids: more workgroups than fit at once -- the ones without a slot start as their predecessor's
    # finish, so the LAST launch of a chain can fill the whole machine while it drains
    ("2q first16 ch16", True, 16, 16, False),
    ("2q first16 ch20", True, 16, 20, False),
    ("2q first16 ch24", True, 16, 24, False),
    ("2q first16 ch32", True, 16, 32, False),
    ("2q first24 ch24", True, 24, 24, False),
    ("2q first32 ch32", True, 32, 32, False),
    ("2q first16 ch48", True, 16, 48, False),
]
if len(sys.argv) > 3:
    configs = [c for c in configs if any(a in c[0] for a in sys.argv[3:])]


STAMPS = []


def region(overlap, first, chained, events, spin=2000, k=None):
    k = K if k is None else k
    os.environ["AETH_SYNC_SPIN_US"] = str(spin)
    os.environ["AETH_FIR_GRID_FIRST"] = str(first)
    os.environ["AETH_FIR_GRID_CHAINED"] = str(chained)
    ctx.set_overlap(overlap)
    for i in range(WARM):
        check(ex(*args[i % ns]))
    e0 = e1 = None
    if events:
        e0, e1 = ctx.event(), ctx.event()
    torch.cuda.synchronize(); ctx.sync()
    t0 = time.perf_counter()
    if events: e0.record()
    for i in range(k):
        check(ex(*args[(WARM + i) % ns]))
    t1 = time.perf_counter()
    if events: e1.record()
    ctx.sync()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    STAMPS.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6, (t3 - t0) * 1e6))
    return (t3 - t0) * 1e6


# settle (load-onset power transient)
ctx.set_overlap(True)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15:
    for i in range(50): check(ex(*args[i % ns]))
    ctx.sync()
res = {c[0]: [] for c in configs}
for r in range(ROUNDS):
    for c in configs:
        res[c[0]].append(region(*c[1:]))
print(f"K = {K}, {ROUNDS} interleaved rounds, single-shot regions (wall, us)")
print(f"{'config':24} {'median':>9} {'min':>9} {'mean':>9} {'max':>9}   GS/s(median)  GS/s(min)")
for c in configs:
    v = np.array(res[c[0]])
    print(f"{c[0]:24} {np.median(v):9.1f} {v.min():9.1f} {v.mean():9.1f} {v.max():9.1f}   {STREAM * K / np.median(v) / 1e3:10.1f} {STREAM * K / v.min() / 1e3:10.1f}")

# where the region's wall time goes on the host side, default policy: enqueue done | ctx.sync done | torch sync done
STAMPS.clear()
for r in range(10):
    region(True, 16, 12, False)
a = np.median(np.array(STAMPS), axis=0)
print(f"host stamps (median of 10, us from t0): {K} launches enqueued {a[0]:.1f} | aeth_ctx_sync returned {a[1]:.1f} | torch.cuda.synchronize returned {a[2]:.1f}")
t0 = time.perf_counter()
for i in range(200): torch.cuda.synchronize()
print(f"torch.cuda.synchronize on an idle device: {(time.perf_counter() - t0) / 200 * 1e6:.2f} us;", end=" ")
t0 = time.perf_counter()
for i in range(200): ctx.sync()
print(f"ctx.sync on an idle context: {(time.perf_counter() - t0) / 200 * 1e6:.2f} us")
# fixed cost vs marginal cost of a launch: regions of different length, same protocol
print("K sweep (default policy, median of 7 single-shot regions, wall us):")
rows = []
for k in (1, 2, 5, 10, 20, 40, 100, 200):
    v = [region(True, 16, 12, False, 2000, k) for _ in range(7)]
    rows.append((k, float(np.median(v)), float(np.min(v))))
slope = (rows[-1][1] - rows[-2][1]) / (rows[-1][0] - rows[-2][0])
for k, med, mn in rows:
    print(f"   K = {k:3d}   median {med:9.1f}  min {mn:9.1f}   per launch {med / k:6.2f}   fixed (median - {slope:.2f} K) {med - slope * k:6.1f}")
