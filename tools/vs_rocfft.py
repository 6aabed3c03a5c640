#!/usr/bin/env python3
"""The hand-written kernels beside the vendor library on the same device and data: hipFFT / rocFFT (called
directly through libhipfft.so) against this library for batched transforms and for the transform * filter * inverse chain.
(torch is imported first: its bundled HIP runtime then serves the whole process.)"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from aether_primitives_amd import Scale

import ctypes
dev = torch.device("cuda", 0)
hipfft = ctypes.CDLL("libhipfft.so")                       # the vendor library called directly: no framework in between
HIPFFT_C2C, HIPFFT_FORWARD, HIPFFT_BACKWARD = 0x29, -1, 1

class VendorPlan:
    def __init__(self, n, batch):
        self.h = ctypes.c_void_p()
        rc = hipfft.hipfftPlan1d(ctypes.byref(self.h), ctypes.c_int(n), ctypes.c_int(HIPFFT_C2C), ctypes.c_int(batch))
        assert rc == 0, rc
        # torch's current stream is the default stream here
    def exec(self, src, dst, direction):
        rc = hipfft.hipfftExecC2C(self.h, ctypes.c_void_p(src), ctypes.c_void_p(dst), ctypes.c_int(direction))
        assert rc == 0, rc

ctx = ap.Context(0)
e0, e1 = ctx.event(), ctx.event()

def t_torch(fn, reps=20, rounds=4):
    for _ in range(30): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / reps)
    return statistics.median(ts)

def t_mine(fn, reps=20, rounds=4):
    for i in range(30): fn(i)
    ctx.sync(); ts = []
    for _ in range(rounds):
        e0.record()
        for i in range(reps): fn(i)
        e1.record(); ctx.sync(); ts.append(e0.elapsed_ms(e1) / reps)
    return statistics.median(ts)

total = 1 << 25                                            # 32 Mi samples = 256 MiB per buffer
rng = np.random.default_rng(0)
host = (rng.standard_normal(2 * total, dtype=np.float32) * 0.7).view(np.complex64)
xs = [torch.from_numpy(host).to(dev) for _ in range(3)]
ys = [torch.empty_like(xs[0]) for _ in range(3)]
mine = [ctx.vec(host) for _ in range(3)]
mine_out = [ctx.empty(total) for _ in range(3)]
print(f"{'case':44s} {'hipFFT / rocFFT':>22s} {'this library':>22s}   ratio")
sizes = [int(a) for a in sys.argv[1:]] or [100, 512, 1000, 1024, 2048, 4096, 8192, 65536]
for N in sizes:
    m = (total // N) * N; batch = m // N
    k = [0]
    vp = VendorPlan(N, batch)
    def f_t():
        i = k[0] % 3; k[0] += 1
        vp.exec(xs[i].data_ptr(), ys[i].data_ptr(), HIPFFT_FORWARD)
    f = ap.HipFft(ctx, N, max_batch=batch)
    def f_m(i): f.bwd(mine[i % 3].slice(0, m), mine_out[i % 3].slice(0, m), Scale.NONE)      # -j exponent, like torch.fft.fft
    a, b = t_torch(f_t), t_mine(f_m)
    print(f"fft N={N:6d} x {batch:7d} out of place".ljust(44) + f" {a*1e3:8.1f} us {16*m/a/1e9:6.2f} TB/s  {b*1e3:8.1f} us {16*m/b/1e9:6.2f} TB/s   {a/b:5.2f}x", flush=True)
if len(sys.argv) > 1: sys.exit(0)
# transform * filter * inverse on 2048-point frames (the correlator chain / one overlap-save block per frame)
N = 2048; batch = total // N
H = torch.from_numpy((rng.standard_normal(2 * N, dtype=np.float32)).view(np.complex64)).to(dev)
k = [0]
vp = VendorPlan(N, batch)
def c_t():
    i = k[0] % 3; k[0] += 1
    vp.exec(xs[i].data_ptr(), ys[i].data_ptr(), HIPFFT_FORWARD)
    ys[i].view(batch, N).mul_(H)                           # one element-wise launch (torch), scale of the inverse left out
    vp.exec(ys[i].data_ptr(), ys[i].data_ptr(), HIPFFT_BACKWARD)
f = ap.HipFft(ctx, N, max_batch=batch); sig = ctx.vec(H.cpu().numpy())
def c_m(i): f.mul_chain(mine[i % 3], sig)
a, b = t_torch(c_t), t_mine(c_m)
print(f"fft * H * ifft, N=2048 x {batch} frames".ljust(44) + f" {a*1e3:8.1f} us {16*total/a/1e9:6.2f} TB/s  {b*1e3:8.1f} us {16*total/b/1e9:6.2f} TB/s   {a/b:5.2f}x")
