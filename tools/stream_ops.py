#!/usr/bin/env python3
"""PCIe-inclusive rate of the host pipeline (aeth_stream_host) for every op its compute stage can be, from pool elements
(copied directly) and from pageable numpy memory (staged both ways):
   fir                     64 taps, FFT-2048                8 B in, 8 B out per sample
   fft                     2048-point frames, Scale::SN     8 in, 8 out
   mul_chain               rfft -> mul -> rifft             8 in, 8 out (in place on the device slot)
   correlate_demod (QPSK)  the chain, then demod_naive      8 in, 2 out
   fft_interpolate         2048-point frames, n_between 9   8 in, ~80 out
   fir_decim               the filter, every 4th output kept 8 in, 2 out
   modem chain             modulate + AWGN -> correlate + demod (two chained stages): 2 bit bytes in, 2 out per symbol
Best of 3, default chunking; the first 4 Mi input samples of every run are compared bit for bit with the op's device
flavour.  -> profiles/r04_stream_host.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from aether_primitives_amd import Scale, pool, pipeline, modulation
from bench import lowpass_taps, FFT_LEN

ctx = ap.Context(0)
N = FFT_LEN
NIN = 64 << 20
fir = ap.Fir(ctx, lowpass_taps(), N)
plan = ap.HipFft(ctx, N, max_batch=4096)
rng = np.random.default_rng(5)
sig = ctx.vec(rng.standard_normal(2 * N, dtype=np.float32).view(np.complex64))
xbig = rng.standard_normal(2 * NIN, dtype=np.float32).view(np.complex64)
pin = pool.Pool(ctx, NIN * 8, initial_len=1)
ein = pin.take()
ein.array(np.complex64)[:] = xbig
NB = 9


def device_flavour(name, x):
    if name == "modem":
        from aether_primitives_amd import noise
        m = modulation.qpsk(ctx)
        return m.correlate_demod(plan, m.modulate_awgn(x, noise.new(ctx, 0.01, 815)), sig).to_host()
    d = ctx.vec(x)
    if name == "fir_decim": return fir.filter_decim(d, 4).to_host()
    if name == "fir": return fir.filter(d).to_host()
    if name == "fft": plan.ifwd(d, Scale.SN); return d.to_host()
    if name == "mul_chain": plan.mul_chain(d, sig); return d.to_host()
    if name == "correlate_demod": return modulation.qpsk(ctx).correlate_demod(plan, d, sig).to_host()
    o = ctx.empty((N + (N - 1) * NB) * (x.size // N)); plan.rfft_interpolate(d, o, NB, Scale.SN); return o.to_host()


ops = [("fir", pipeline.Stage.fir(fir), NIN), ("fft", pipeline.Stage.fft(plan, Scale.SN), NIN),
       ("mul_chain", pipeline.Stage.mul_chain(plan, sig), NIN), ("correlate_demod", pipeline.Stage.correlate_demod(plan, sig, 2), NIN),
       ("fft_interpolate", pipeline.Stage.fft_interpolate(plan, NB, Scale.SN), 16 << 20),
       ("fir_decim", pipeline.Stage.fir_decim(fir, 4), NIN)]
bits_big = rng.integers(0, 2, 2 * NIN, dtype=np.uint8)
pbits = pool.Pool(ctx, bits_big.nbytes, initial_len=1); ebits = pbits.take(); ebits.array(np.uint8)[:] = bits_big
print(f"# {'op':16s} {'memory':9s} {'in':>7s} {'out':>9s} chunks       ms    GS/s   GB/s up  GB/s down   check (first 4 Mi samples)")
chain = [pipeline.Stage.modulate_awgn(ctx, 2, 0.01, 815), pipeline.Stage.correlate_demod(plan, sig, 2)]
ops.append(("modem", chain, 2 * NIN))                       # n counts INPUT elements: bit bytes
for name, st, n in ops:
    is_chain = isinstance(st, list)
    if is_chain:
        stages, st = st, st[-1]
        n_out = n
        runner = lambda x, out: pipeline.run_chain(stages, x, out=out)
    else:
        runner = lambda x, out: pipeline.run(st, x, out=out)
    n_out = n_out if is_chain else st.out_count(n)
    obytes = n_out * np.dtype(st.out_dtype).itemsize
    pout = pool.Pool(ctx, obytes, initial_len=1)
    eout = pout.take()
    ypage = np.empty(n_out, st.out_dtype)
    want = device_flavour(name, bits_big[:8 << 20] if is_chain else xbig[:4 << 20])
    for kind in ("pool", "pageable"):
        x = (ebits.array(np.uint8, n) if kind == "pool" else bits_big[:n]) if is_chain else (ein.array(np.complex64, n) if kind == "pool" else xbig[:n])
        y = eout.array(st.out_dtype, n_out) if kind == "pool" else ypage
        best = None
        for rep in range(3):
            _, s = runner(x, y)
            if best is None or s["seconds"] < best["seconds"]: best = s
        k = want.size
        same = "bit-identical" if np.array_equal(y[:k].view(np.uint8), want.view(np.uint8)) else "MISMATCH"
        if name == "fir_decim": same = "bit-identical" if np.array_equal(y[:k - 4096].view(np.uint8), want[:k - 4096].view(np.uint8)) else "MISMATCH"
        if name == "fir": same = "bit-identical" if np.array_equal(y[:k - 4096].view(np.uint8), want[:k - 4096].view(np.uint8)) else "MISMATCH"   # the 4 Mi run's last block sees zeros where the long stream has samples
        ibytes = n * (1 if is_chain else 8)
        print(f"  {name:16s} {kind:9s} {n >> 20:4d} Mi {n_out / (1 << 20):7.1f} Mi {int(best['chunks']):5d} {best['seconds'] * 1e3:9.2f} {n / best['seconds'] / 1e9:7.2f} "
              f"{ibytes / best['seconds'] / 1e9:9.1f} {obytes / best['seconds'] / 1e9:10.1f}   {same}  pinned={int(best['pinned'])}", flush=True)
    _, rep = (pipeline.run_chain(stages, bits_big[:n], out=ypage, report=True) if is_chain else pipeline.run(st, xbig[:n], out=ypage, report=True))
    for l in rep["lines"]: print("        " + l)
    x = y = None                      # the arrays lent by the elements keep them checked out
    eout.close(); pout.close()
    ctx.trim()
ebits.close(); pbits.close(); ein.close(); pin.close()
