#!/usr/bin/env python3
"""Tuning helper: A/B the four-step FFT under settings of the AETH_4S_* knobs, interleaved
rounds in ONE process.  Every variant's output is compared bit for bit with the default
build's (the knobs may only change scheduling, never arithmetic -- AETH_4S_NOTW excepted).
usage: tune_4step.py [batch] [N] -- "K=V K=V" "K=V" ...   (each quoted string = one variant)"""
import os, sys, statistics
os.environ.setdefault('AETH_TUNING', '1')   # enables the library's AETH_* tuning knobs
os.environ.setdefault('AETH_LAB_LIB', '1')   # these knobs exist only in the lab build: make -C aether_primitives_amd/csrc LAB=1
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from aether_primitives_amd import Scale

args = sys.argv[1:]
variants = [""]
if "--" in args:
    i = args.index("--"); variants = args[i + 1:] or [""]; args = args[:i]
batch = int(args[0]) if len(args) > 0 else 512
N = int(args[1]) if len(args) > 1 else 65536
KEYS = ["AETH_4S_NOTW", "AETH_4S_GROUP_MIB", "AETH_4S_COLG"]

def setenv(v):
    for k in KEYS: os.environ.pop(k, None)
    for kv in v.split():
        k, val = kv.split("="); os.environ[k] = val

ctx = ap.Context(0); e0, e1 = ctx.event(), ctx.event()
rng = np.random.default_rng(0)
pat = rng.standard_normal(2 * N * min(batch, 64), dtype=np.float32).view(np.complex64)
x = np.tile(pat, (batch + 63) // 64)[:N * batch]
src = ctx.vec(x)
A = [ctx.empty(N * batch) for _ in range(3)]
f = ap.HipFft(ctx, N)

# reference: default settings
setenv("")
A[0].vec_clone(src); f.ifwd(A[0], Scale.SN); ctx.sync()
ref = A[0].to_host()
for v in variants:
    setenv(v)
    A[1].vec_clone(src); f.ifwd(A[1], Scale.SN); ctx.sync()
    got = A[1].to_host()
    same = bool((got.view(np.uint32) == ref.view(np.uint32)).all())
    f.fwd(src, A[2], Scale.SN); ctx.sync()
    oop = A[2].to_host()
    same2 = bool((oop.view(np.uint32) == ref.view(np.uint32)).all())
    print(f"[{v or 'default':44s}] in-place identical: {same}   out-of-place identical: {same2}", flush=True)

res = {v: [] for v in variants}
for i in range(3): A[i].vec_clone(src)
for rnd in range(5):
    for v in variants:
        setenv(v)
        for i in range(3): f.ifwd(A[i % 3], Scale.SN)
        ctx.sync(); e0.record()
        for i in range(20): f.ifwd(A[i % 3], Scale.SN)
        e1.record(); ctx.sync(); res[v].append(e0.elapsed_ms(e1) / 20)
for v in variants:
    t = statistics.median(res[v][1:])
    print(f"[{v or 'default':44s}] N={N} batch={batch}: {t*1e3:8.1f} us  {N*batch/t/1e6:7.1f} GS/s  {16*N*batch/t/1e6/80:5.1f}% of 8TB/s (16 B/sample)")
