#!/usr/bin/env python3
"""Per-kernel bandwidth survey (HBM-honest sizes): every op of the hot path, achieved
algorithmic GB/s against the 8 TB/s spec peak.  Rotates over buffers > Infinity Cache."""
import os, sys, statistics, json
os.environ.setdefault('AETH_TUNING', '1')
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aether_primitives_amd as ap
from aether_primitives_amd import Scale, sampling

ctx = ap.Context(0)
e0, e1 = ctx.event(), ctx.event()


def timeit(fn, reps=20, rounds=5, settle_ms=40.0):
    # carry the device through its load-onset power transient first (tools/transient.py): a kernel that
    # draws more than the previous one runs 10-25 % slower for ~25 ms before it settles
    import time
    t0 = time.perf_counter(); k = 0
    while (time.perf_counter() - t0) * 1e3 < settle_ms:
        for i in range(10): fn(k + i)
        k += 10; ctx.sync()
    ts = []
    for _ in range(rounds):
        fn(0); fn(1); ctx.sync(); e0.record()
        for i in range(reps): fn(i)
        e1.record(); ctx.sync(); ts.append(e0.elapsed_ms(e1) / reps)
    return statistics.median(ts)


rows = []
def report(name, nbytes, ms, samples=None):
    gbs = nbytes / ms / 1e6
    extra = f"  {samples / ms / 1e6:8.1f} GS/s" if samples else ""
    print(f"{name:46s} {ms*1e3:9.1f} us  {gbs:8.1f} GB/s  {gbs/80:5.1f}% of 8TB/s{extra}", flush=True)
    rows.append({"op": name, "us": ms * 1e3, "GBps": gbs, "frac_of_8TBps": gbs / 8000})


rng = np.random.default_rng(0)
n = 1 << 25                       # 32 Mi samples = 256 MiB per vector
NB = 4
host = (rng.standard_normal(2 * n, dtype=np.float32) * 0.7).view(np.complex64)
A = [ctx.vec(host) for _ in range(NB)]
B = [ctx.vec(host[::-1].copy()) for _ in range(NB)]
for op, b in (("vec_add", 24), ("vec_sub", 24), ("vec_mul", 24), ("vec_div", 24), ("vec_clone", 16)):
    report(f"{op} n=2^25", b * n, timeit(lambda i: getattr(A[i % NB], op)(B[i % NB])), n)
for op, b in (("vec_conj", 16), ("vec_mirror", 16), ("vec_zero", 8)):
    report(f"{op} n=2^25", b * n, timeit(lambda i: getattr(A[i % NB], op)()), n)
report("vec_scale n=2^25", 16 * n, timeit(lambda i: A[i % NB].vec_scale(1.0001)), n)
report("vec_mirror_frames(2048) n=2^25", 16 * n, timeit(lambda i: A[i % NB].vec_mirror_frames(2048)), n)
# C1 shape: 4096-sample chain (launch-bound)
v, a, b = ctx.vec(host[:4096]), ctx.vec(host[4096:8192]), ctx.vec(host[8192:12288])
report("C1 chain add->mul->conj n=4096 (3 launches)", 64 * 4096, timeit(lambda i: v.vec_add(a).vec_mul(b).vec_conj(), reps=50), 4096)

for N in (512, 1024, 2048, 4096, 12, 100, 126, 143, 960, 1000, 1536, 3600, 6000, 10000, 16384, 20480, 30720):
    f = ap.HipFft(ctx, N)
    m = (n // N) * N
    report(f"fft ifwd N={N} batch={m // N} ({f.algorithm})", 16 * m, timeit(lambda i: f.ifwd(A[i % NB].slice(0, m), Scale.SN)), m)
f = ap.HipFft(ctx, 2048)
# round 4: the chained methods in one pass (aeth_vec_chain) -- BASELINE config 1's chain on HBM-sized operands and on 4096 samples
c3 = [ctx.vec(host[:n]) for _ in range(3)]
report("chain add->mul->conj, three calls (64 B/sample)", 64 * n, timeit(lambda i: A[i % NB].vec_add(c3[0]).vec_mul(c3[1]).vec_conj()), n)
report("chain add->mul->conj, fused (32 B/sample)", 32 * n, timeit(lambda i: A[i % NB].fused().vec_add(c3[0]).vec_mul(c3[1]).vec_conj().run()), n)
s4 = [ctx.vec(host[:4096]) for _ in range(3)]
report("C1 literal: chain on 4096 samples, three calls", 64 * 4096, timeit(lambda i: s4[0].vec_add(s4[1]).vec_mul(s4[2]).vec_conj(), reps=200), 4096)
report("C1 literal: chain on 4096 samples, fused", 32 * 4096, timeit(lambda i: s4[0].fused().vec_add(s4[1]).vec_mul(s4[2]).vec_conj().run(), reps=200), 4096)
report("fft fwd (out of place) N=2048", 16 * n, timeit(lambda i: f.fwd(A[i % NB], B[i % NB], Scale.SN)), n)
c2 = [ctx.vec(host[:1 << 20]) for _ in range(2)]
report("C2 literal: fft-2048 ifwd on 1 Mi samples (8 MiB)", 16 * (1 << 20), timeit(lambda i: f.ifwd(c2[i % 2], Scale.SN), reps=50), 1 << 20)
sig = ctx.vec(host[:2048])
report("correlator chain N=2048 (fused) batch=16384", 16 * n, timeit(lambda i: f.mul_chain(A[i % NB], sig)), n)
for N in (8192, 65536, 1 << 20):
    f2 = ap.HipFft(ctx, N)
    for bt in ((64, n // N) if N == 65536 else (n // N,)):
        m = bt * N
        report(f"fft ifwd N={N} batch={bt} ({f2.algorithm})", 16 * m, timeit(lambda i: f2.ifwd(A[i % NB].slice(0, m), Scale.SN)), m)
f3 = ap.HipFft(ctx, 4099)
m = (n // 4099 // 8) * 4099
report(f"fft ifwd N=4099 batch={m // 4099} (bluestein)", 16 * m, timeit(lambda i: f3.ifwd(A[i % NB].slice(0, m), Scale.SN), reps=5), m)

# sampling
S = 1 << 22
dst = [ctx.empty(S * 10) for _ in range(2)]
report("interpolate n_between=9, 4 Mi in -> 40 Mi out", 8 * S + 80 * S, timeit(lambda i: sampling.interpolate(ctx, A[i % NB].slice(0, S), dst[i % 2], 9)), S)
report("interpolate frames(65536) n_between=9", 8 * S + 80 * S, timeit(lambda i: sampling.interpolate(ctx, A[i % NB].slice(0, S), dst[i % 2], 9, frame_len=65536)), S)
for dec in (2, 8, 30):
    nd = n // dec
    d = ctx.empty(nd)
    report(f"downsample dec={dec} (n_dst={nd})", 16 * nd, timeit(lambda i: sampling.downsample(ctx, A[i % NB].slice(0, nd * dec), d)), nd)
small = ctx.vec(host[:30720]); d1 = ctx.empty(1024)
report("downsample 30720->1024 (reference bench shape)", 16 * 1024, timeit(lambda i: sampling.downsample(ctx, small, d1), reps=100), 1024)
# modulation / noise
from aether_primitives_amd import modulation, noise
nsym = 1 << 25
q = modulation.qpsk(ctx)
bits = modulation.DeviceBits(ctx, 2 * nsym, rng.integers(0, 2, 2 * nsym, dtype=np.uint8))
outb = modulation.DeviceBits(ctx, 2 * nsym)
report("qpsk modulate 2^25 symbols", 10 * nsym, timeit(lambda i: q.modulate(bits, out=A[i % NB])), nsym)
report("qpsk demod_naive 2^25 symbols", 10 * nsym, timeit(lambda i: q.demod_naive(A[i % NB], out=outb)), nsym)
g = noise.new(ctx, 0.01, 815)
report("awgn apply 2^25 samples", 16 * nsym, timeit(lambda i: g.apply(A[i % NB])), nsym)
# round 2: fused / new entry points
report("qpsk modulate_awgn (fused) 2^25 symbols", 10 * nsym, timeit(lambda i: q.modulate_awgn(bits, g, out=A[i % NB])), nsym)
report("correlate + demod (fused) N=2048 batch=16384", 10 * n, timeit(lambda i: q.correlate_demod(f, A[i % NB], sig, out=outb)), n)
report("awgn fill 2^25 samples", 8 * nsym, timeit(lambda i: g.fill(A[i % NB])), nsym)
sig100 = ctx.vec(host[:100]); m100 = (n // 100) * 100
report("vec_mul_frames N=100 (one launch, 335544 frames)", 16 * m100, timeit(lambda i: A[i % NB].slice(0, m100).vec_mul_frames(sig100)), m100)
fir = ap.Fir(ctx, host[:64] * 0.1, 2048)
half = n // 2
for dec in (4, 16):
    dd = ctx.empty(half // dec)
    report(f"fir + decimating store dec={dec} (16 Mi in)", 8 * half + 8 * half // dec, timeit(lambda i: fir.filter_decim(A[i % NB].slice(0, half), dec, out=dd)), half)
report("fft + mirror epilogue N=2048", 16 * n, timeit(lambda i: f.rfft_mirror(A[i % NB], Scale.SN)), n)
for N in (17, 85, 1700, 61, 323, 1003, 2006, 2176, 4199):
    fp = ap.HipFft(ctx, N); m = (n // N) * N
    report(f"fft ifwd N={N} batch={m // N} ({fp.algorithm})", 16 * m, timeit(lambda i: fp.ifwd(A[i % NB].slice(0, m), Scale.SN)), m)
json.dump(rows, open("gpurun_out/kernel_survey.json", "w"), indent=1)
