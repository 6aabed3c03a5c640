#!/usr/bin/env python3
"""Soak: randomized FFT / FIR / element-wise cases against the oracle for a given number of seconds, one
process.  Prints a progress line every ~20 s; exits non-zero on the first mismatch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import aether_primitives_amd as ap
from aether_primitives_amd import HipFft, Fir, Scale
from oracle import pyoracle as orc
from helpers import rand_c64, bits_equal

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
ctx = ap.Context(0)
smooth = sorted({2 ** a * 3 ** b * 5 ** c for a in range(15) for b in range(10) for c in range(7)
                 if 3 <= 2 ** a * 3 ** b * 5 ** c <= 20480} - {2 ** k for k in range(14)})     # the ragged table
seven = sorted({2 ** a * 3 ** b * 5 ** c * 7 ** d for a in range(13) for b in range(8) for c in range(6) for d in range(1, 5)
                if 2 ** a * 3 ** b * 5 ** c * 7 ** d <= 4096})
eleven = sorted({2 ** a * 3 ** b * 5 ** c * 7 ** d * 11 ** e * 13 ** f for a in range(12) for b in range(7) for c in range(5)
                 for d in range(4) for e in range(3) for f in range(3)
                 if e + f >= 1 and 2 ** a * 3 ** b * 5 ** c * 7 ** d * 11 ** e * 13 ** f <= 2048})
seventeen = sorted({17 ** g * m for g in (1, 2) for m in range(1, 121) if 17 ** g * m <= 2048
                    and all(m % q for q in (19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 101, 103, 107, 109, 113))})
def _lpf(n):
    best, f = 1, 2
    while f * f <= n:
        while n % f == 0: best, n = f, n // f
        f += 1
    return n if n > 1 else best
nineteen = [n for n in range(19, 2049) if _lpf(n) in (19, 23)]
lengths = [2 ** k for k in range(1, 17)] + smooth + seven + eleven + seventeen + nineteen + [
    17, 19, 34, 67, 127, 134, 323, 257, 509, 1009, 2039, 4099, 4116, 7203, 8190, 9604, 23040, 30720, 100000,
    # round 2: register butterflies 11 .. 23 in the LDS kernel, chirp-z with a free convolution length, four-step splits
    2176, 2431, 3553, 4199, 6647, 6859, 7429, 8177, 4583, 5003, 6007, 8191, 10007, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21]
plans, firs = {}, {}
t0 = last = time.time(); it = 0; worst = -400.0
while time.time() - t0 < secs:
    it += 1
    kind = rng.integers(8)
    if kind == 0:                                            # FFT, random batch / sign / scale / placement
        n = int(rng.choice(lengths)); batch = int(rng.choice([1, 2, 3, 7, 16, 33, 128]))
        if n * batch > (1 << 22): batch = max(1, (1 << 22) // n)
        sign = int(rng.choice([-1, 1])); s = [Scale.NONE, Scale.SN, Scale.N, Scale.X(0.25)][int(rng.integers(4))]
        if n not in plans: plans[n] = HipFft(ctx, n, max_batch=128)
        x = rand_c64(int(rng.integers(1 << 30)), n * batch)
        d = ctx.vec(x)
        if rng.integers(2): plans[n].exec(d, d, sign, s); got = d.to_host()
        else: o = ctx.empty(n * batch); plans[n].exec(d, o, sign, s); got = o.to_host()
        fac = orc.scale_factor(s.kind, n, s.x)
        ref = np.concatenate([orc.fft_f64(x[i * n:(i + 1) * n].astype(np.complex128), sign) for i in range(batch)]) * fac
        e = orc.evm_db(got, ref); worst = max(worst, e)
        if not e <= -115.0: print(f"FFT mismatch n={n} batch={batch} sign={sign} scale={s}: EVM {e:.1f} dB"); sys.exit(1)
    elif kind == 1:                                          # FIR, random geometry and length
        fft_len = int(rng.choice([256, 512, 1024, 2048, 4096])); ntaps = int(rng.choice([1, 2, 8, 33, 64, fft_len // 4]))
        key = (fft_len, ntaps)
        taps = orc.synth_lowpass_taps(ntaps, 0.2)
        if key not in firs: firs[key] = Fir(ctx, taps, fft_len)
        n = int(rng.integers(1, 1 << 19))
        x = rand_c64(int(rng.integers(1 << 30)), n)
        got = firs[key].filter(ctx.vec(x)).to_host()
        ref = orc.fir_direct_f64(taps, x)
        e = orc.evm_db(got, ref) if n >= 64 else -200.0
        if not e <= -110.0:
            # a stream much shorter than the filter only sees the filter's leading tail: outputs of 1e-4 beside block
            # rounding noise of 1e-7 x the input.  That is conditioning, not a defect, if the oracle's own f32 chain
            # is no better (the parity tests' rule: within 8 dB of the oracle's error)
            eo = orc.evm_db(orc.fir_ols_f32(taps, x, fft_len, firs[key].hop), ref)
            if not e <= eo + 8.0: print(f"FIR mismatch fft_len={fft_len} ntaps={ntaps} n={n}: EVM {e:.1f} dB (oracle's f32 chain: {eo:.1f} dB)"); sys.exit(1)
        else: worst = max(worst, e)
    elif kind == 3:                                          # interpolate / downsample, bit-exact
        from aether_primitives_amd import sampling
        S = int(rng.integers(2, 1 << 16)); nb = int(rng.choice([0, 1, 2, 3, 9, 15, 100]))
        src = rand_c64(int(rng.integers(1 << 30)), S)
        dst = ctx.empty(S + (S - 1) * nb)
        compat = bool(rng.integers(2))
        sampling.interpolate(ctx, ctx.vec(src), dst, nb, compat_im=compat)
        if not bits_equal(dst.to_host(), orc.interpolate(src, nb, compat_im=compat)): print(f"interpolate mismatch S={S} nb={nb}"); sys.exit(1)
        dec = int(rng.choice([1, 2, 3, 8, 30])); nd = int(rng.integers(1, 1 << 15))
        big = rand_c64(int(rng.integers(1 << 30)), nd * dec)
        out = ctx.empty(nd); sampling.downsample(ctx, ctx.vec(big), out)
        if not bits_equal(out.to_host(), big[::dec][:nd]): print(f"downsample mismatch nd={nd} dec={dec}"); sys.exit(1)
    elif kind == 4:                                          # modulate -> awgn -> demod, bit-exact against the oracle
        from aether_primitives_amd import modulation, noise
        bps = int(rng.choice([1, 2])); nsym = int(rng.integers(1, 1 << 17)); seed = int(rng.integers(1 << 30))
        bits = rng.integers(0, 2, nsym * bps, dtype=np.uint8)
        m = modulation.qpsk(ctx) if bps == 2 else modulation.bpsk(ctx)
        tx = m.modulate(modulation.DeviceBits(ctx, bits.size, bits))
        ref = orc.modulate(bits, bps)
        if not bits_equal(tx.to_host(), ref): print(f"modulate mismatch bps={bps} nsym={nsym}"); sys.exit(1)
        noise.new(ctx, 0.04, seed).apply(tx)
        ref = orc.awgn_apply(ref, 0.04, seed=seed)
        if not bits_equal(tx.to_host(), ref): print(f"awgn mismatch nsym={nsym} seed={seed}"); sys.exit(1)
        compat = bool(rng.integers(2))
        got = m.demod_naive(tx, compat=compat).to_host()
        if not (got == orc.demod_naive(ref, bps, compat=compat)).all(): print(f"demod mismatch bps={bps} nsym={nsym}"); sys.exit(1)
        # the fused form and the fill, at a random stream position (odd ones take the per-sample draw) and alignment
        off = int(rng.integers(0, 1 << 40)); lead = int(rng.integers(0, 2))
        g = noise.new(ctx, 0.04, seed); g.offset = off
        buf = ctx.empty(nsym + lead)
        fused = m.modulate_awgn(modulation.DeviceBits(ctx, bits.size, bits), g, out=buf.slice(lead, lead + nsym))
        if not bits_equal(fused.to_host(), orc.awgn_apply(orc.modulate(bits, bps), 0.04, seed=seed, offset=off)): print(f"modulate_awgn mismatch bps={bps} nsym={nsym} off={off} lead={lead}"); sys.exit(1)
        g2 = noise.new(ctx, 0.04, seed); g2.offset = off
        g2.fill(buf.slice(lead, lead + nsym))
        if not bits_equal(buf.slice(lead, lead + nsym).to_host(), orc.awgn_fill(nsym, 0.04, seed, off)): print(f"awgn fill mismatch nsym={nsym} off={off} lead={lead}"); sys.exit(1)
    elif kind == 5:                                          # round 4: a random chain of links in one pass (aeth_vec_chain), bit-exact
        n = int(rng.integers(1, 1 << 20)); nl = int(rng.integers(1, 20))
        ops = [np.complex64(2) + rand_c64(int(rng.integers(1 << 30)), n + 3) for _ in range(3)]
        dev = [ctx.vec(o) for o in ops]
        start = rand_c64(int(rng.integers(1 << 30)), n + 3)
        off = int(rng.integers(0, 3))
        dv = ctx.vec(start); ch = dv.slice(off, off + n).fused(); want = start[off:off + n].copy()
        for _ in range(nl):
            k = int(rng.integers(8)); j = int(rng.integers(3)); oo = int(rng.integers(0, 3))
            o, do = ops[j][oo:oo + n], dev[j].slice(oo, oo + n)
            if k == 0: ch.vec_scale(0.9); want = orc.vec_scale(want, 0.9)
            elif k == 1: ch.vec_mul(do); want = orc.vec_mul(want, o)
            elif k == 2: ch.vec_div(do); want = orc.vec_div(want, o)
            elif k == 3: ch.vec_conj(); want = orc.vec_conj(want)
            elif k == 4: ch.vec_add(do); want = orc.vec_add(want, o)
            elif k == 5: ch.vec_sub(do); want = orc.vec_sub(want, o)
            elif k == 6: ch.vec_clone(do); want = o.copy()
            else: ch.vec_zero(); want = np.zeros(n, np.complex64)
        ch.run()
        exp = start.copy(); exp[off:off + n] = want
        g, e_ = dv.to_host().view(np.float32), exp.view(np.float32)
        nan = np.isnan(e_)
        if not ((np.isnan(g) == nan).all() and (g.view(np.uint32)[~nan] == e_.view(np.uint32)[~nan]).all()): print(f"chain mismatch n={n} links={nl} off={off}"); sys.exit(1)
    elif kind in (6, 7):                                     # round 4: a host stream through a random pipeline stage, bit-exact against the device call
        from aether_primitives_amd import pipeline, modulation
        N = int(rng.choice([1024, 2048, 4096])); frames = int(rng.integers(1, 400))
        if N not in plans: plans[N] = HipFft(ctx, N, max_batch=128)
        f = plans[N]; x = rand_c64(int(rng.integers(1 << 30)), N * frames)
        sigh = rand_c64(int(rng.integers(1 << 30)), N, scale=0.3); sig = ctx.vec(sigh)
        chunk = int(rng.choice([0, N, N * 3, N * 64]))
        which = int(rng.integers(4))
        if which == 0:
            y, _ = pipeline.run(pipeline.Stage.fft(f, Scale.SN), x, chunk=chunk); d = ctx.vec(x); f.ifwd(d, Scale.SN); ok = bits_equal(y, d.to_host())
        elif which == 1:
            y, _ = pipeline.run(pipeline.Stage.mul_chain(f, sig), x, chunk=chunk); d = ctx.vec(x); f.mul_chain(d, sig); ok = bits_equal(y, d.to_host())
        elif which == 2:
            bps = int(rng.choice([1, 2])); m = modulation.qpsk(ctx) if bps == 2 else modulation.bpsk(ctx)
            y, _ = pipeline.run(pipeline.Stage.correlate_demod(f, sig, bps), x, chunk=chunk); ok = np.array_equal(y, m.correlate_demod(f, ctx.vec(x), sig).to_host())
        else:
            nb = int(rng.choice([1, 3, 9])); y, _ = pipeline.run(pipeline.Stage.fft_interpolate(f, nb, Scale.N), x, chunk=chunk)
            o = ctx.empty((N + (N - 1) * nb) * frames); f.rfft_interpolate(ctx.vec(x), o, nb, Scale.N); ok = bits_equal(y, o.to_host())
        if not ok: print(f"pipeline mismatch stage={which} N={N} frames={frames} chunk={chunk}"); sys.exit(1)
        if it % 50 == 0: ctx.trim()
    else:                                                    # element-wise chain, bit-exact
        n = int(rng.integers(1, 1 << 20))
        a, b = rand_c64(int(rng.integers(1 << 30)), n), rand_c64(int(rng.integers(1 << 30)), n) + np.complex64(2)
        v = ctx.vec(a); v.vec_add(ctx.vec(b)).vec_mul(ctx.vec(b)).vec_conj().vec_div(ctx.vec(b)).vec_scale(0.5)
        want = orc.vec_scale(orc.vec_div(orc.vec_conj(orc.vec_mul(orc.vec_add(a.copy(), b), b)), b), 0.5)
        if not bits_equal(v.to_host(), want): print(f"element-wise mismatch n={n}"); sys.exit(1)
    if time.time() - last > 20:
        last = time.time(); print(f"{it} cases, {time.time() - t0:.0f} s, worst EVM {worst:.1f} dB", flush=True)
print(f"soak ok: {it} cases in {time.time() - t0:.0f} s, worst EVM {worst:.1f} dB")
