"""GPU parity: HipFft behind the reference's Fft trait (src/fft.rs:48-77, :134-235).

Tolerances (stated once, used below):
  * the reference's own FFT tests are replayed with the reference's own
    assert_evm! thresholds (-80 / -72 'dB' of the macro's scale, src/lib.rs:36-47);
  * on random spectra a different f32 FFT cannot match rustfft bit for bit, so
    parity is aggregate EVM 20*log10(|err|/|ref|): north_star asks <= -80 dB vs the
    reference; we require <= -120 dB vs the f64 ground truth AND vs the f32 oracle,
    and that the GPU error vs truth is within 8 dB of the oracle own (floor -140 dB).
"""
import numpy as np
import pytest

import aether_primitives_amd as ap
from aether_primitives_amd import Scale, HipFft
from helpers import expand, load_kat, bits_equal, rand_c64

pytestmark = pytest.mark.gpu
KAT = load_kat()
TOL_DB = -120.0

POW2 = [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096]


def _largest_prime_factor(n):
    best, f = 1, 2
    while f * f <= n:
        while n % f == 0:
            best, n = f, n // f
        f += 1
    return n if n > 1 else best


# the ragged register-resident table: every length 2^a 3^b 5^c up to 20480 that is no power of two, 8192, 16384, and every
# length 2^a 3^b 5^c 7^d up to 4096 with d >= 1
SMOOTH = sorted(({2 ** a * 3 ** b * 5 ** c for a in range(15) for b in range(10) for c in range(7)
                  if 3 <= 2 ** a * 3 ** b * 5 ** c <= 20480} - {2 ** k for k in range(15)}) | {8192, 16384}
                | {2 ** a * 3 ** b * 5 ** c * 7 ** d for a in range(13) for b in range(8) for c in range(6) for d in range(1, 5)
                   if 2 ** a * 3 ** b * 5 ** c * 7 ** d <= 4096}
                # ... and every 13-smooth length up to 2048 with a factor 11 or 13
                | {2 ** a * 3 ** b * 5 ** c * 7 ** d * 11 ** e * 13 ** f for a in range(12) for b in range(7) for c in range(5)
                   for d in range(4) for e in range(3) for f in range(3)
                   if e + f >= 1 and 2 ** a * 3 ** b * 5 ** c * 7 ** d * 11 ** e * 13 ** f <= 2048}
                # ... and every 17-smooth length up to 2048 with a factor 17 that has a decomposition of at most 34 points per lane
                | ({17 ** g * m for g in (1, 2) for m in range(1, 121)
                    if 17 ** g * m <= 2048 and all(m % q for q in (19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 101, 103, 107, 109, 113))}
                   - {578, 748})
                # ... and the 23-smooth lengths up to 2048 whose largest prime factor is 19 or 23 (15 of the 143 have no row)
                | ({n for n in range(19, 2049) if _largest_prime_factor(n) in (19, 23)}
                   - {418, 437, 506, 529, 646, 665, 722, 759, 782, 828, 836, 874, 897, 1012, 1058}))
MIXED = [1, 17, 19, 34, 61, 289, 323, 2079, 4095, 4116, 7203, 8190,           # LDS ping-pong kernel
         # ... with its register butterflies for 11 .. 23 (beyond the one-launch chirp-z kernel's 2048), alone and mixed
         # with a radix that still takes the O(r^2) pass (37)
         2176, 2431, 3553, 4199, 6647, 6859, 7429, 8177]
BIG = [8192, 32768, 65536, 1 << 17, 1 << 18, 1 << 20, 1 << 21, 1 << 22]   # 8192: one workgroup; above: four-step, every column-group width
ODD = [67, 97, 127, 134, 1009, 4099, 5000, 6000, 10007]
# above 8192 with two factors of at most 8192: transposes around the batched transforms of the factors
TWO_FACTOR = [8232, 9604, 21000, 23040, 24000, 30720, 100000, 196608, 1000000, 2048 * 2025]


def _truth(oracle, x, n, sign):
    return oracle.fft_f64_frames(x.astype(np.complex128), n, sign)


def _check(oracle, got, x, n, sign, factor=1.0):
    truth = _truth(oracle, x, n, sign) * float(factor)
    e_gpu = oracle.evm_db(got, truth)
    orc = oracle.Cfft(n).frames(x, sign).astype(np.complex64)
    e_orc = oracle.evm_db((orc.astype(np.complex128) * float(factor)).astype(np.complex64), truth)
    assert e_gpu <= TOL_DB, f"N={n}: GPU vs f64 truth {e_gpu:.1f} dB"
    assert e_gpu <= max(e_orc, -140.0) + 8.0, f"N={n}: GPU {e_gpu:.1f} dB vs oracle {e_orc:.1f} dB"
    return e_gpu


# ---- the reference's own tests ---------------------------------------------------
def test_reference_doctest_fft128(ctx):
    """src/fft.rs:93-117 via DeviceVec, then again via host slices."""
    for host in (False, True):
        data = expand(KAT["fft128_ones_fwd"]["self"])
        if host:
            v = ap.HostVec(ctx, data.copy()); get = lambda: v.a
        else:
            v = ctx.vec(data); get = v.to_host
        v.vec_fft(Scale.NONE)                                                     # fresh plan, fft.rs:98
        ap.assert_evm(get(), expand(KAT["fft128_ones_fwd"]["expect"]), -80.0)     # off-DC bins exactly 0
        f = HipFft(ctx, 128)
        f.ibwd(v.a if host else v, Scale.N)                                       # fft.rs:110-111
        ap.assert_evm(get(), expand(KAT["fft128_back_n"]["expect"]), -80.0)
        v.vec_rfft(f, Scale.SN).vec_scale(2.0).vec_rifft(f, Scale.SN)             # fft.rs:116
        ap.assert_evm(get(), expand(KAT["fft128_sn_scale2_sn"]["expect"]), -72.0)


def test_reference_roundtrip_100(ctx):
    """src/vecops.rs:443-463: N = 100 = 2^2 5^2, const (1,1), SN both ways, default -80."""
    v = expand(KAT["vec_fft_roundtrip_100"]["self"])
    c = ctx.vec(v)
    c.vec_fft(Scale.SN).vec_ifft(Scale.SN)
    ap.assert_evm(c.to_host(), v)
    c = ctx.vec(v); f = HipFft(ctx, 100)
    assert f.algorithm in ("stockham_mixed", "stockham_mixed_reg", "stockham_mixed_ragged")
    c.vec_rfft(f, Scale.SN).vec_rifft(f, Scale.SN)
    ap.assert_evm(c.to_host(), v)
    h = v.copy(); ap.HostVec(ctx, h).vec_rfft(f, Scale.SN).vec_rifft(f, Scale.SN)
    ap.assert_evm(h, v)


# ---- parity on random spectra ------------------------------------------------------
@pytest.mark.parametrize("n", POW2 + MIXED + SMOOTH)
def test_fft_vs_truth_small(ctx, oracle, n):
    f = HipFft(ctx, n)
    assert f.len() == n
    if n in SMOOTH:
        assert f.algorithm == "stockham_mixed_ragged"
    for sign, batch in ((+1, 1), (-1, 1), (+1, 7), (-1, 130 if n <= 8192 else 9)):
        x = rand_c64(1000 * n + batch + sign, n * batch)
        out = ctx.empty(n * batch)
        f.exec(ctx.vec(x), out, sign)
        _check(oracle, out.to_host(), x, n, sign)
        if batch == 7:
            d = ctx.vec(x); f.exec(d, d, sign)                   # in place
            assert bits_equal(d.to_host(), out.to_host())


@pytest.mark.parametrize("n", BIG + ODD + TWO_FACTOR)
def test_fft_vs_truth_big(ctx, oracle, n):
    f = HipFft(ctx, n)
    if n in TWO_FACTOR:
        assert f.algorithm == "fourstep_mixed"
    for sign, batch in ((+1, 1), (-1, 3)):
        x = rand_c64(n + batch, n * batch)
        out = ctx.empty(n * batch)
        f.exec(ctx.vec(x), out, sign)
        _check(oracle, out.to_host(), x, n, sign)
        d = ctx.vec(x); f.exec(d, d, sign)                       # in place
        assert bits_equal(d.to_host(), out.to_host())


@pytest.mark.parametrize("n", [3, 7, 11, 12, 25, 100, 120, 143, 196, 480, 1000, 2002, 2401, 3125, 3600, 6000, 7500, 10000, 16384, 20480])
def test_ragged_streaming_batch_matches_small_batches(ctx, n):
    """Batches beyond the cache take the non-temporal instantiation and the persistent grid wraps several times;
    the bits must be those of the same frames transformed a handful at a time."""
    f = HipFft(ctx, n)
    assert f.algorithm == "stockham_mixed_ragged"
    batch = (9 << 20) // n + 3                                   # > 128 MiB of traffic, ragged last group
    x = rand_c64(77 + n, n * batch)
    big = ctx.empty(n * batch)
    f.exec(ctx.vec(x), big, +1)
    got = big.to_host().reshape(batch, n)
    for lo in (0, batch // 2 - 2, batch - 5):
        part = x.reshape(batch, n)[lo:lo + 5].reshape(-1)
        out = ctx.empty(5 * n)
        f.exec(ctx.vec(part), out, +1)
        assert bits_equal(out.to_host().reshape(5, n), got[lo:lo + 5]), (n, lo)


def test_algorithms_chosen(ctx):
    assert HipFft(ctx, 2048).algorithm == "stockham_pow2"
    assert HipFft(ctx, 100).algorithm == "stockham_mixed_ragged"
    assert HipFft(ctx, 126).algorithm == "stockham_mixed_ragged"
    assert HipFft(ctx, 143).algorithm == "stockham_mixed_ragged"
    assert HipFft(ctx, 1700).algorithm == "stockham_mixed_ragged"      # 4 * 25 * 17: radix 17 in registers
    assert HipFft(ctx, 578).algorithm == "bluestein"                   # 2 * 17^2: no register decomposition in the table
    assert HipFft(ctx, 323).algorithm == "stockham_mixed_ragged"     # 17 * 19: radix 17 and 19 in registers
    assert HipFft(ctx, 437).algorithm == "bluestein"                 # 19 * 23: no row; one-launch chirp-z beats the O(r^2) prime pass
    assert HipFft(ctx, 2 * 2057).algorithm == "stockham_mixed"       # 4114 = 2 * 11^2 * 17: too long for the one-launch kernel
    assert HipFft(ctx, 65536).algorithm == "fourstep_pow2"
    assert HipFft(ctx, 4099).algorithm == "bluestein"
    assert HipFft(ctx, 10000).algorithm == "stockham_mixed_ragged"
    assert HipFft(ctx, 16384).algorithm == "stockham_mixed_ragged"
    assert HipFft(ctx, 30000).algorithm == "fourstep_mixed"
    assert HipFft(ctx, 2 * 10007).algorithm == "bluestein"


@pytest.mark.parametrize("n", [8, 100, 2048, 65536, 97])
def test_trait_methods_agree(ctx, oracle, n):
    """fwd/bwd (copy), ifwd/ibwd (in place), tfwd/tbwd (plan's temp) give the same
    bits; fwd leaves its input alone (fft.rs:49-50); sign binding: fwd=+j, bwd=-j."""
    x = rand_c64(n, n)
    f = HipFft(ctx, n)
    for fwd, ifwd, tfwd, sign in ((f.fwd, f.ifwd, f.tfwd, +1), (f.bwd, f.ibwd, f.tbwd, -1)):
        for s in (Scale.NONE, Scale.SN, Scale.N, Scale.X(0.37)):
            inp, out = ctx.vec(x), ctx.empty(n)
            fwd(inp, out, s)
            assert bits_equal(inp.to_host(), x)
            ref = out.to_host()
            _check(oracle, ref, x, n, sign, s.factor(n))
            io = ctx.vec(x); ifwd(io, s); assert bits_equal(io.to_host(), ref)
            assert bits_equal(tfwd(ctx.vec(x), s).to_host(), ref)
            # host-slice flavours
            ho = np.empty(n, np.complex64); fwd(x, ho, s); assert bits_equal(ho, ref)
            hio = x.copy(); ifwd(hio, s); assert bits_equal(hio, ref)
            assert bits_equal(np.array(tfwd(x, s)), ref)


def test_scale_is_applied_to_unscaled_result_like_the_reference(ctx):
    """Reference: process() then a separate vec_scale pass (fft.rs:169-170): scaled = fl(fl(X) * s)."""
    n = 2048
    x = rand_c64(5, n)
    f = HipFft(ctx, n)
    raw = ctx.empty(n); f.fwd(ctx.vec(x), raw, Scale.NONE)
    for s in (Scale.SN, Scale.N, Scale.X(3.3)):
        sc = ctx.empty(n); f.fwd(ctx.vec(x), sc, s)
        assert bits_equal(sc.to_host(), ctx.vec(raw.to_host()).vec_scale(s.factor(n)).to_host())


def test_length_assert(ctx):
    f = HipFft(ctx, 128)
    with pytest.raises(ap.LengthMismatch, match="Input and FFT must be the same length"):
        f.ifwd(ctx.vec(rand_c64(1, 127)), Scale.NONE)
    with pytest.raises(ap.LengthMismatch, match="Input and FFT must be the same length"):
        f.ifwd(rand_c64(1, 129), Scale.NONE)
    with pytest.raises(ap.LengthMismatch):
        f.fwd(ctx.vec(rand_c64(1, 128)), ctx.empty(64), Scale.NONE)
    with pytest.raises(ap.LengthMismatch):
        f.tfwd(rand_c64(1, 100), Scale.NONE)


def test_c2_full_size_properties(ctx, oracle):
    """BASELINE config 2: FFT-2048 fwd + ifwd over a 1 M-sample stream (512 frames)."""
    n, frames = 2048, 512
    x = oracle.synth_cnormal(815, n * frames)
    f = HipFft(ctx, n, max_batch=frames)
    d = ctx.vec(x)
    out = ctx.empty(x.size)
    f.fwd(d, out, Scale.SN)
    X = out.to_host()
    _check(oracle, X, x, n, +1, Scale.SN.factor(n))
    # Parseval per frame with 1/sqrt(N) scaling
    pin = (np.abs(x.astype(np.complex128)) ** 2).reshape(frames, n).sum(1)
    pout = (np.abs(X.astype(np.complex128)) ** 2).reshape(frames, n).sum(1)
    assert np.max(np.abs(pout / pin - 1)) < 1e-5
    # in-place variant, then back: round trip
    f.ifwd(d, Scale.SN); assert bits_equal(d.to_host(), X)
    f.ibwd(d, Scale.SN)
    assert oracle.evm_db(d.to_host(), x) <= TOL_DB
    # linearity: F(a x + y) = a F(x) + F(y)
    y = oracle.synth_cnormal(816, n * frames)
    z = (np.float32(0.5) * x + y).astype(np.complex64)
    Fz = ctx.empty(z.size); f.fwd(ctx.vec(z), Fz, Scale.SN)
    Fy = ctx.empty(z.size); f.fwd(ctx.vec(y), Fy, Scale.SN)
    lin = 0.5 * X.astype(np.complex128) + Fy.to_host().astype(np.complex128)
    assert oracle.evm_db(Fz.to_host(), lin) <= TOL_DB


def test_fft_then_mirror_frames_is_fftshift(ctx, oracle):
    """util/plot.rs:59-61: chunks_mut(fft_len).for_each(|c| c.vec_rfft(..).vec_mirror())."""
    n, frames = 256, 9
    x = rand_c64(4, n * frames)
    f = HipFft(ctx, n)
    d = ctx.vec(x); f.ifwd(d, Scale.SN); d.vec_mirror_frames(n)
    ref = ctx.vec(x); f.ifwd(ref, Scale.SN)
    exp = np.fft.fftshift(ref.to_host().reshape(frames, n), axes=1).reshape(-1)
    assert bits_equal(d.to_host(), exp)


def test_c5_full_size_properties(ctx, oracle):
    """BASELINE config 5: batched 65536-point FFT (Scale::SN) + 10x linear interpolation
    (n_between = 9), 64 frames.  Oracle on a few frames, size-independent properties on all."""
    from aether_primitives_amd import sampling
    n, frames, nb = 65536, 64, 9
    x = oracle.synth_cnormal(815, n * frames)
    f = HipFft(ctx, n, max_batch=frames)
    assert f.algorithm == "fourstep_pow2"
    d = ctx.vec(x)
    f.ifwd(d, Scale.SN)
    X = d.to_host()
    for fr in (0, 17, 63):                                   # oracle + f64 truth on single frames
        seg = slice(fr * n, (fr + 1) * n)
        truth = oracle.fft_f64(x[seg].astype(np.complex128), +1) / np.sqrt(float(n))
        assert oracle.evm_db(X[seg], truth) <= TOL_DB
    # Parseval on every frame
    pin = (np.abs(x.astype(np.complex128)) ** 2).reshape(frames, n).sum(1)
    pout = (np.abs(X.astype(np.complex128)) ** 2).reshape(frames, n).sum(1)
    assert np.max(np.abs(pout / pin - 1)) < 1e-5
    # round trip
    back = ctx.vec(X); f.ibwd(back, Scale.SN)
    assert oracle.evm_db(back.to_host(), x) <= TOL_DB
    # interpolate every frame independently: 655351 outputs per frame, bit-exact vs the oracle on 3 frames,
    # and the kept samples (every 10th output) must be the FFT outputs' real parts verbatim
    Lo = n + (n - 1) * nb
    out = ctx.empty(Lo * frames)
    assert sampling.interpolate(ctx, d, out, nb, frame_len=n) == Lo * frames
    Y = out.to_host().reshape(frames, Lo)
    for fr in (0, 31, 63):
        assert bits_equal(Y[fr], oracle.interpolate(X[fr * n:(fr + 1) * n], nb))
    assert bits_equal(np.ascontiguousarray(Y[:, ::nb + 1].real), np.ascontiguousarray(X.reshape(frames, n).real))
    assert bits_equal(np.ascontiguousarray(Y[:, -1]), np.ascontiguousarray(X.reshape(frames, n)[:, -1]))


@pytest.mark.parametrize("n,batch", [(2, 5), (4, 3), (16, 100), (64, 33), (256, 7), (512, 9), (1024, 5), (2048, 17), (4096, 3), (8192, 2),
                                     (100, 6), (1536, 4), (65536, 2)])
def test_fft_with_mirror_epilogue(ctx, n, batch):
    """c.vec_rfft(fft, s).vec_mirror() per frame (util/plot.rs:59-61) in one call: bit-identical to the two steps"""
    x = rand_c64(n, n * batch)
    f = HipFft(ctx, n, max_batch=batch)
    two = ctx.vec(x); f.ifwd(two, Scale.SN); two.vec_mirror_frames(n)
    one = ctx.vec(x); f.rfft_mirror(one, Scale.SN)
    assert bits_equal(one.to_host(), two.to_host())
    o = ctx.empty(n * batch)
    f.rfft_mirror(ctx.vec(x), Scale.NONE, out=o)
    t = ctx.vec(x); f.ifwd(t, Scale.NONE); t.vec_mirror_frames(n)
    assert bits_equal(o.to_host(), t.to_host())


@pytest.mark.parametrize("n,batch,nb", [(65536, 3, 9), (65536, 1, 1), (16384, 5, 4), (32768, 2, 2), (131072, 2, 9), (262144, 1, 3),
                                        (2048, 4, 9), (100, 3, 2), (524288, 1, 1)])
def test_fft_then_interpolate_in_one_call(ctx, n, batch, nb):
    """BASELINE config 5's chain in one call: per frame vec_rfft(Scale::SN) then sampling::interpolate
    (sampling.rs:7-24); bit-identical to the two calls, input left untouched."""
    from aether_primitives_amd import sampling
    x = rand_c64(n + nb, n * batch)
    f = HipFft(ctx, n, max_batch=batch)
    Lo = n + (n - 1) * nb
    for compat in (True, False):
        two_x = ctx.vec(x); f.ifwd(two_x, Scale.SN)
        two = ctx.empty(Lo * batch); sampling.interpolate(ctx, two_x, two, nb, frame_len=n, compat_im=compat)
        src = ctx.vec(x)
        one = ctx.vec(np.full(Lo * batch + 3, 5 - 5j, np.complex64))
        wrote = f.rfft_interpolate(src, one.slice(0, Lo * batch), nb, Scale.SN, compat_im=compat)
        h = one.to_host()
        assert wrote == Lo * batch and bits_equal(h[:Lo * batch], two.to_host())
        assert (h[Lo * batch:] == 5 - 5j).all() and bits_equal(src.to_host(), x)      # nothing past dst, input untouched


def test_one_shot_vec_fft_reuses_the_contexts_plans(ctx, oracle):
    """vec_fft / vec_ifft (vecops.rs:184-196) plan per call in the reference; here the context keeps the plans: results are
    those of a fresh plan, more lengths than the cache holds still work, a trimmed context plans again, and the second
    call of a length is much cheaper than the first"""
    import time
    for n in (100, 128, 2048, 6000, 17, 1000, 4096, 512, 960, 3600, 100, 2048):     # 10 distinct lengths > 8 cached, repeats
        x = rand_c64(n, n)
        got = ctx.vec(x).vec_fft(Scale.SN).to_host()
        d = ctx.vec(x); HipFft(ctx, n).ifwd(d, Scale.SN)
        assert bits_equal(got, d.to_host()), n
        back = ctx.vec(got).vec_ifft(Scale.SN).to_host()
        assert oracle.evm_db(back, x.astype(np.complex128)) <= -120
        h = ap.HostVec(ctx, x.copy()); h.vec_fft(Scale.SN)
        assert bits_equal(h.a, got), n
    v = ctx.vec(rand_c64(1, 3000))
    ctx.sync(); t0 = time.perf_counter(); v.vec_fft(Scale.NONE); ctx.sync(); first = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(20): v.vec_fft(Scale.NONE)
    ctx.sync(); later = (time.perf_counter() - t0) / 20
    assert later < first / 3, (first, later)
    ctx.trim()
    assert bits_equal(ctx.vec(rand_c64(5, 128)).vec_fft(Scale.N).to_host(), ctx.vec(rand_c64(5, 128)).vec_fft(Scale.N).to_host())
