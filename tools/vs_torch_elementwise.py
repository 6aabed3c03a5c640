#!/usr/bin/env python3
"""Element-wise cf32 ops beside PyTorch's own kernels on the same device (256 MiB operands, rotating buffers).
(torch first: its bundled HIP runtime then serves the whole process.)"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap

n = 1 << 25
xs = [torch.randn(n, dtype=torch.complex64, device="cuda") for _ in range(3)]
ys = [torch.randn(n, dtype=torch.complex64, device="cuda") for _ in range(3)]
ctx = ap.Context(0); e0, e1 = ctx.event(), ctx.event()
host = np.ones(n, np.complex64)
A = [ctx.vec(host) for _ in range(3)]; B = [ctx.vec(host) for _ in range(3)]

def t_torch(fn, reps=20):
    for _ in range(30): fn(0)
    torch.cuda.synchronize(); ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(reps): fn(i)
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / reps)
    return statistics.median(ts)

def t_mine(fn, reps=20):
    for i in range(30): fn(i)
    ctx.sync(); ts = []
    for _ in range(4):
        e0.record()
        for i in range(reps): fn(i)
        e1.record(); ctx.sync(); ts.append(e0.elapsed_ms(e1) / reps)
    return statistics.median(ts)

rows = [("add (24 B)", 24, lambda i: xs[i % 3].add_(ys[i % 3]), lambda i: A[i % 3].vec_add(B[i % 3])),
        ("mul (24 B)", 24, lambda i: xs[i % 3].mul_(ys[i % 3]), lambda i: A[i % 3].vec_mul(B[i % 3])),
        ("clone / copy (16 B)", 16, lambda i: xs[i % 3].copy_(ys[i % 3]), lambda i: A[i % 3].vec_clone(B[i % 3])),
        ("scale (16 B)", 16, lambda i: xs[i % 3].mul_(1.0001), lambda i: A[i % 3].vec_scale(1.0001)),
        ("conj (16 B)", 16, lambda i: torch.conj_physical_(xs[i % 3]), lambda i: A[i % 3].vec_conj()),
        ("zero (8 B)", 8, lambda i: xs[i % 3].zero_(), lambda i: A[i % 3].vec_zero())]
print(f"{'op':22s} {'torch':>20s} {'this library':>20s}   ratio")
for name, b, ft, fm in rows:
    a, m = t_torch(ft), t_mine(fm)
    print(f"{name:22s} {a*1e3:7.1f} us {b*n/a/1e9:5.2f} TB/s {m*1e3:7.1f} us {b*n/m/1e9:5.2f} TB/s   {a/m:5.2f}x")
