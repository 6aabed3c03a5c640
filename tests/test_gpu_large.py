"""GPU parity at sizes whose BYTE offsets cross 4 GiB (and, for interpolate, whose element
count crosses 2^31): the "maximum sizes" edge of the path.  288 GB of HBM makes such
operands ordinary, and 32-bit index arithmetic in a kernel would only show here.

The operands are built on the device by tiling a 2^22-sample seeded pattern (no multi-GiB
host arrays); results are checked on slices -- the head, the 4 GiB crossing, the ragged
tail -- against the oracle run on the same slice.  Element-wise ops bit-exact
(src/vecops.rs:94-177, src/sampling.rs:7-42); FFT/FIR within the -80 dB aggregate EVM
of the headline metric (src/lib.rs:26-49).
"""
import numpy as np
import pytest

import aether_primitives_amd as ap
from aether_primitives_amd import Scale, sampling, HipFft
from helpers import bits_equal, rand_c64

pytestmark = pytest.mark.gpu

P = 1 << 22                       # pattern period (samples)
CROSS = 1 << 29                   # sample index whose byte offset is 4 GiB


def tiled(ctx, pattern_dev, n):
    """DeviceVec of n samples = the pattern repeated (device-to-device clones)."""
    v = ctx.empty(n)
    for o in range(0, n, P):
        m = min(P, n - o)
        v.slice(o, o + m).vec_clone(pattern_dev.slice(0, m))
    return v


def host_slice(pat, lo, hi):
    idx = np.arange(lo, hi) % P
    return pat[idx]


def windows(n, w=4096):
    """(lo, hi) slices worth checking for an n-sample vector"""
    out = [(0, w), (CROSS - w, min(CROSS + w, n)), (max(n - w, 0), n)]
    return [(lo, hi) for lo, hi in out if 0 <= lo < hi <= n]


@pytest.mark.parametrize("op", ["vec_add", "vec_mul", "vec_div", "vec_conj", "vec_scale", "vec_mirror"])
def test_vecops_beyond_4gib(ctx, oracle, op):
    n = CROSS + (1 << 20) + 3
    pa, pb = rand_c64(11, P), rand_c64(12, P) + np.complex64(3)
    da, db = ctx.vec(pa), ctx.vec(pb)
    a = tiled(ctx, da, n)
    b = tiled(ctx, db, n) if op in ("vec_add", "vec_mul", "vec_div") else None
    if op == "vec_scale": a.vec_scale(0.37)
    elif b is not None: getattr(a, op)(b)
    else: getattr(a, op)()
    ctx.sync()
    mid = n // 2
    for lo, hi in windows(n):
        got = a.slice(lo, hi).to_host()
        if op == "vec_mirror":
            # swap(x, x+mid): output[i] = input[i+mid] for i < mid, input[i-mid] above (vecops.rs:157-161);
            # an odd length leaves the last element in place
            idx = np.arange(lo, hi)
            src = np.where(idx < mid, idx + mid, idx - mid)
            if n % 2: src = np.where(idx == n - 1, idx, src)
            want = pa[src % P]
        else:
            x = host_slice(pa, lo, hi).copy()
            if op == "vec_scale": want = oracle.vec_scale(x, 0.37)
            elif op == "vec_conj": want = oracle.vec_conj(x)
            else: want = getattr(oracle, op)(x, host_slice(pb, lo, hi))
        assert bits_equal(got, want), (op, lo, hi)


def test_fft2048_batch_beyond_4gib(ctx, oracle):
    N = 2048
    batch = CROSS // N + 5                       # frames straddle the 4 GiB byte offset
    pat = rand_c64(21, P)
    x = tiled(ctx, ctx.vec(pat), N * batch)
    f = ap.HipFft(ctx, N)
    f.ifwd(x, Scale.SN)
    ctx.sync()
    cf = oracle.Cfft(N)
    for fr in (0, CROSS // N - 1, CROSS // N, batch - 1):
        got = x.slice(fr * N, (fr + 1) * N).to_host()
        ref = oracle.fft_f64(host_slice(pat, fr * N, (fr + 1) * N).astype(np.complex128), +1) / np.sqrt(N)
        assert oracle.evm_db(got, ref) <= -80.0, fr
        assert oracle.evm_db(got, cf.fwd(host_slice(pat, fr * N, (fr + 1) * N), 1)) <= -80.0, fr
    # round trip of the whole buffer returns the pattern (checked on slices)
    f.ibwd(x, Scale.SN)
    ctx.sync()
    for lo, hi in windows(N * batch):
        assert oracle.evm_db(x.slice(lo, hi).to_host(), host_slice(pat, lo, hi)) <= -80.0


def test_fir_stream_beyond_4gib(ctx, oracle):
    n = CROSS + 3 * 1984 + 17
    taps = oracle.synth_lowpass_taps(64, 0.25)
    pat = rand_c64(31, P)
    x = tiled(ctx, ctx.vec(pat), n)
    y = ctx.empty(n)
    fir = ap.Fir(ctx, taps, 2048)
    fir.filter(x, out=y)
    ctx.sync()
    for lo, hi in windows(n, 8192):
        h0 = max(lo - 63, 0)
        seg = host_slice(pat, h0, hi)
        ref = oracle.fir_direct_f64(taps, seg)[lo - h0:]
        got = y.slice(lo, hi).to_host()
        if lo == 0:
            ref = oracle.fir_direct_f64(taps, host_slice(pat, 0, hi))
        assert oracle.evm_db(got, ref) <= -80.0, (lo, hi)


def test_interpolate_output_beyond_4gib(ctx, oracle):
    # 32-bit-index kernel with byte offsets past 4 GiB: 2^26+1 inputs x 9 between -> ~2^29.3 outputs
    S = (1 << 26) + 1
    nb = 9
    pat = rand_c64(41, P)
    src = tiled(ctx, ctx.vec(pat), S)
    Lo = S + (S - 1) * nb
    dst = ctx.empty(Lo)
    assert sampling.interpolate(ctx, src, dst, nb) == Lo
    ctx.sync()
    for w0 in (0, (CROSS // (nb + 1)) - 50, S - 200):
        w1 = min(w0 + 200, S)
        want = oracle.interpolate(host_slice(pat, w0, w1), nb)
        # the oracle closes its slice with the raw last sample (sampling.rs:22-23); inside the
        # stream that position is an ordinary i = 0 output, so it is compared only at the true end
        k = want.size if w1 == S else want.size - 1
        got = dst.slice(w0 * (nb + 1), w0 * (nb + 1) + k).to_host()
        assert bits_equal(got, want[:k]), w0


def test_interpolate_beyond_2_31_outputs(ctx, oracle):
    # more than 2^31 outputs (16 GiB): the 64-bit kernel
    S = (1 << 27) + 7
    nb = 16
    pat = rand_c64(42, P)
    src = tiled(ctx, ctx.vec(pat), S)
    Lo = S + (S - 1) * nb
    assert Lo > (1 << 31)
    dst = ctx.empty(Lo)
    assert sampling.interpolate(ctx, src, dst, nb) == Lo
    ctx.sync()
    for w0 in (0, (1 << 31) // (nb + 1) - 50, (1 << 32) // (nb + 1) - 50 if (1 << 32) < Lo else 1000, S - 150):
        w1 = min(w0 + 150, S)
        want = oracle.interpolate(host_slice(pat, w0, w1), nb)
        # the oracle closes its slice with the raw last sample (sampling.rs:22-23); inside the
        # stream that position is an ordinary i = 0 output, so it is compared only at the true end
        k = want.size if w1 == S else want.size - 1
        got = dst.slice(w0 * (nb + 1), w0 * (nb + 1) + k).to_host()
        assert bits_equal(got, want[:k]), w0


def test_downsample_source_beyond_4gib(ctx, oracle):
    n_dst = (1 << 24) + 1
    dec = 40
    n_src = n_dst * dec                         # 5.4 GiB of source
    pat = rand_c64(51, P)
    src = tiled(ctx, ctx.vec(pat), n_src)
    dst = ctx.empty(n_dst)
    sampling.downsample(ctx, src, dst)
    ctx.sync()
    for lo in (0, CROSS // dec - 100, n_dst - 300):
        hi = min(lo + 300, n_dst)
        want = pat[(np.arange(lo, hi) * dec) % P]
        assert bits_equal(dst.slice(lo, hi).to_host(), want), lo


@pytest.mark.parametrize("logn", [23, 24])
def test_fourstep_largest_lengths(ctx, oracle, logn):
    """2^23 and 2^24 points (the deep form: 128 / 256 columns over rows of 65536 points that are four-step transforms
    themselves, then a transpose) in place and out of place against the f64 truth (advertised but untested)."""
    n = 1 << logn
    x = rand_c64(logn, n)
    f = HipFft(ctx, n)
    assert f.algorithm == "fourstep_pow2"
    truth = np.fft.ifft(x.astype(np.complex128)) * np.sqrt(float(n))      # reference fwd = +j exponent; Scale::SN
    d = ctx.vec(x); f.ifwd(d, Scale.SN)
    assert oracle.evm_db(d.to_host(), truth) <= -120
    o = ctx.empty(n); f.fwd(ctx.vec(x), o, Scale.SN)
    assert bits_equal(o.to_host(), d.to_host())
    back = ctx.vec(d.to_host()); f.ibwd(back, Scale.SN)
    assert oracle.evm_db(back.to_host(), x) <= -120
    if logn == 23:
        # a batch of frames through the deep form (columns over all frames, nested rows, transpose per frame): every
        # frame comes out with the bits of the single-frame run
        y = rand_c64(99, n)
        two = ctx.vec(np.concatenate([x, y])); f.fwd(two, two, Scale.SN)
        got = two.to_host()
        assert bits_equal(got[:n], d.to_host())
        e = ctx.vec(y); f.ifwd(e, Scale.SN)
        assert bits_equal(got[n:], e.to_host())


def test_c4_channel_at_baseline_size(ctx, oracle):
    """One channel of BASELINE config 4 at its full size (4096 frames of 2048: 8.4 M QPSK symbols, noise power 0.01):
    the two fused calls bench.py makes -- modulate_awgn, correlate_demod -- against (i) the four separate device calls,
    bit for bit (symbols and decided bits), (ii) the oracle's chain on the transmit side bit for bit (8.4 M noisy
    symbols: table lookup + the generator + the reference's double scaling), (iii) the oracle's receive chain: its FFT
    rounds differently from the device's, so a decision may differ only where the oracle's own distances to the two
    candidates are within rounding of each other -- and at this noise level none should."""
    from aether_primitives_amd import modulation, noise
    N, frames = 2048, 4096
    n = N * frames
    bits = np.random.default_rng(815).integers(0, 2, 2 * n, dtype=np.uint8)
    q = modulation.qpsk(ctx)
    dbits = modulation.DeviceBits(ctx, 2 * n, bits)
    f = HipFft(ctx, N, max_batch=frames)
    ref = np.zeros(N, np.complex64)
    ref[:4] = np.conj(np.array([-1 + 1j, 0, 1 - 1j, 1 - 1j], np.complex64))       # benches.rs:394-405
    sig = ctx.vec(ref)
    tx = q.modulate_awgn(dbits, noise.new(ctx, 0.01, 815))
    rx = q.correlate_demod(f, tx, sig).to_host()
    # (i) the separate calls on the device
    tx2 = q.modulate(dbits); noise.new(ctx, 0.01, 815).apply(tx2)
    assert bits_equal(tx.to_host(), tx2.to_host())
    f.mul_chain(tx2, sig)
    assert (rx == q.demod_naive(tx2).to_host()).all()
    # (ii) transmit side against the oracle
    txo = oracle.awgn_apply(oracle.modulate(bits, 2), 0.01, seed=815)
    assert bits_equal(tx.to_host(), txo)
    # (iii) receive side against the oracle
    yo = oracle.correlate_frames(ref, txo)
    want = oracle.demod_naive(yo, 2, compat=True)
    bad = np.flatnonzero(rx != want)
    if bad.size:
        sym = np.unique(bad // 2)
        tab = np.array([1 + 1j, -1 + 1j, 1 - 1j, -1 - 1j], np.complex64)
        d = np.abs(yo[sym, None] - tab[None, :]) ** 2
        d.sort(axis=1)
        assert sym.size <= 8 and ((d[:, 1] - d[:, 0]) <= 1e-4 * d[:, 1]).all(), (sym.size, d[:4])
    # size-independent: the same bits again at another stream position decode to the same decisions only by chance
    tx3 = q.modulate_awgn(dbits, noise.new(ctx, 0.01, 816))
    assert not bits_equal(tx3.to_host()[:4096], txo[:4096])


def test_c5_job_at_baseline_size(ctx, oracle):
    """BASELINE config 5 at its full size on one GPU (512 frames of 65536 samples, forward FFT Scale::SN, then
    sampling::interpolate with n_between = 9: 335 M output samples): the one call bench.py makes against the two trait-level
    calls, bit for bit over the whole output (compared 64 frames at a time); two frames -- the first and the last --
    against the oracle: the transform within -120 dB of the f64 transform, the interpolation of the device's spectrum bit
    for bit."""
    n, frames, nb = 65536, 512, 9
    Lo = n + (n - 1) * nb
    pat = rand_c64(815, P)
    pd = ctx.vec(pat)
    x = tiled(ctx, pd, n * frames)                       # frame f = pattern[(f * n) % P ...]: 64 distinct frames, repeated
    x.slice(0, n).vec_scale(0.5)                         # ... and the first and last frame made different from their copies
    x.slice(n * (frames - 1), n * frames).vec_conj()
    f = HipFft(ctx, n, max_batch=frames)
    one = ctx.empty(Lo * frames)
    assert f.rfft_interpolate(x, one, nb, Scale.SN) == Lo * frames
    spec = ctx.empty(n * frames); spec.vec_clone(x); f.ifwd(spec, Scale.SN)
    two = ctx.empty(Lo * frames)
    sampling.interpolate(ctx, spec, two, nb, frame_len=n)
    step = 64
    for f0 in range(0, frames, step):
        a = one.slice(Lo * f0, Lo * (f0 + step)).to_host(); b = two.slice(Lo * f0, Lo * (f0 + step)).to_host()
        assert bits_equal(a, b), f"frames {f0}..{f0 + step}"
    for fr in (0, frames - 1):
        xin = x.slice(n * fr, n * (fr + 1)).to_host()
        sp = spec.slice(n * fr, n * (fr + 1)).to_host()
        truth = oracle.fft_f64_frames(xin.astype(np.complex128), n, +1) / np.sqrt(float(n))
        assert oracle.evm_db(sp, truth) <= -120.0
        assert bits_equal(one.slice(Lo * fr, Lo * (fr + 1)).to_host(), oracle.interpolate(sp, nb))
