import os, sys, statistics
os.environ.setdefault('AETH_TUNING', '1')   # enables the library's AETH_* tuning knobs
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from aether_primitives_amd import Scale
ctx = ap.Context(0); e0, e1 = ctx.event(), ctx.event()
N = 65536; batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = (np.random.default_rng(0).standard_normal(2 * N * batch, dtype=np.float32)).view(np.complex64)
A = [ctx.vec(x) for _ in range(3)]
f = ap.HipFft(ctx, N)
ts = []
for r in range(5):
    for i in range(3): f.ifwd(A[i % 3], Scale.SN)
    ctx.sync(); e0.record()
    for i in range(20): f.ifwd(A[i % 3], Scale.SN)
    e1.record(); ctx.sync(); ts.append(e0.elapsed_ms(e1) / 20)
t = statistics.median(ts)
print(f"[{os.environ.get('AETH_4S_NOTW','')}] N={N} batch={batch}: {t*1e3:.1f} us  {N*batch/t/1e6:.1f} GS/s  {16*N*batch/t/1e6/80:.1f}% of 8TB/s (16 B/sample)")
