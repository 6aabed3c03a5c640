#!/usr/bin/env python3
"""PCIe-inclusive rate of the host pipeline (aeth_fir_stream_host) for streams of 1 Mi ... 256 Mi samples and the
three kinds of caller memory:
   pool       both slices live in elements of an aeth_pool (pinned once by the library): copied directly
   pageable   plain numpy memory: staged through the context's pinned pool by the copy threads (five stages)
   mixed      pageable input, pool output
Best of 5 per case (the first uses of a freshly pinned 2 GiB element pay a one-off mapping cost of 8-19 ms each), default chunking (and a chunk sweep for 64 Mi); output checked bit for bit against the
device-resident one-shot run for the sizes that fit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from aether_primitives_amd import pool
from bench import lowpass_taps, FFT_LEN

ctx = ap.Context(0)
fir = ap.Fir(ctx, lowpass_taps(), FFT_LEN)
NMAX = 256 << 20
pp = pool.Pool(ctx, NMAX * 8, initial_len=2)
ein, eout = pp.take(), pp.take()
rng = np.random.default_rng(5)
xbig = rng.standard_normal(2 * NMAX, dtype=np.float32).view(np.complex64)
ein.array(np.complex64)[:] = xbig
ybig = np.empty(NMAX, np.complex64)


def run(kind, n, chunk=0, report=False):
    x = ein.array(np.complex64, n) if kind == "pool" else xbig[:n]
    y = ybig[:n] if kind == "pageable" else eout.array(np.complex64, n)
    best = None
    for rep in range(5):
        _, st = fir.filter_stream(x, out=y, chunk=chunk)
        if best is None or st["seconds"] < best["seconds"]: best = st
    if report:
        best = dict(best, lines=fir.filter_stream(x, out=y, chunk=chunk, report=True)[1]["lines"])
    return y, best


print("# default chunking (4 Mi samples, an eighth of the stream if that is less, at least 128 Ki)")
for n in (1 << 20, 4 << 20, 16 << 20, 64 << 20, 256 << 20):
    ref = fir.filter(ctx.vec(xbig[:n])).to_host() if n <= (64 << 20) else None
    for kind in ("pool", "pageable", "mixed"):
        y, st = run(kind, n)
        same = "" if ref is None else ("  bit-identical" if np.array_equal(y.view(np.uint32), ref.view(np.uint32)) else "  MISMATCH")
        print(f"n = {n >> 20:4d} Mi  {kind:9s} chunks {int(st['chunks']):3d}  {st['seconds'] * 1e3:8.2f} ms  {n / st['seconds'] / 1e9:5.2f} GS/s  "
              f"{8 * n / st['seconds'] / 1e9:5.1f} GB/s per direction  pinned={int(st['pinned'])}{same}", flush=True)
print("# chunk sweep, 64 Mi samples; indented: the per-stage report (format of src/pipeline.rs:101-108) of the run above it")
for kind in ("pool", "pageable"):
    for chunk in (1 << 20, 2 << 20, 4 << 20, 8 << 20, 16 << 20):
        y, st = run(kind, 64 << 20, chunk, report=(chunk == 4 << 20))
        print(f"n =   64 Mi  {kind:9s} chunk {chunk >> 20:3d} Mi samples  {st['seconds'] * 1e3:8.2f} ms  {(64 << 20) / st['seconds'] / 1e9:5.2f} GS/s  "
              f"{8 * (64 << 20) / st['seconds'] / 1e9:5.1f} GB/s per direction", flush=True)
        for l in st.get("lines", []): print("      " + l)
y = None                              # the arrays lent by the elements keep them checked out
ein.close(); eout.close(); pp.close()
