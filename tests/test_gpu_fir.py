"""GPU parity: fused overlap-save FIR and the frequency-domain multiply chain.

The reference has no FIR (src/fir.rs:3-22 is a stub); the filter is DEFINED by
the reference's chain rfft -> vec_mul -> rifft (benches/benches.rs:410-416) run
as overlap-save, so parity is pinned by mathematics: direct convolution in f64
(oracle.fir_direct_f64) and the oracle's f32 restatement of the chain.
Tolerance: aggregate EVM <= -120 dB (north_star: <= -80 dB)."""
import numpy as np
import pytest

import aether_primitives_amd as ap
from aether_primitives_amd import Scale, HipFft, Fir
from helpers import bits_equal, rand_c64

pytestmark = pytest.mark.gpu
TOL_DB = -120.0


@pytest.fixture(scope="module")
def taps(oracle):
    return oracle.synth_lowpass_taps(64, 0.25)


def test_geometry(ctx, taps):
    f = Fir(ctx, taps, 2048)
    assert (f.ntaps, f.fft_len, f.hop) == (64, 2048, 1984)      # hop = floor64(N - M + 1)
    with pytest.raises(ap.AetherError):
        Fir(ctx, taps, 64)                                       # fft_len < 2*ntaps
    with pytest.raises(ap.AetherError):
        Fir(ctx, taps, 3000)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1983, 1984, 1985, 2048, 5000, 123457])
def test_fir_vs_direct_convolution(ctx, oracle, taps, n):
    x = rand_c64(n, n)
    f = Fir(ctx, taps, 2048)
    y = f.filter(ctx.vec(x)).to_host()
    truth = oracle.fir_direct_f64(taps, x)
    ols = oracle.fir_ols_f32(taps, x, 2048, f.hop)
    if n >= 63:
        assert oracle.evm_db(y, truth) <= TOL_DB
        assert oracle.evm_db(y, ols) <= TOL_DB
    else:
        # a handful of start-up outputs: FFT convolution errors scale with the block, not the
        # sample, so use an absolute bound (a few f32 ulps of max|x| * sum|h|)
        bound = 4 * 2.0 ** -23 * float(np.abs(x).max()) * float(np.abs(taps).sum())
        assert np.abs(y - truth).max() <= bound and np.abs(y - ols).max() <= bound
    # host-slice flavour gives the same bits
    assert bits_equal(f.filter(x), y)


@pytest.mark.parametrize("fft_len,ntaps", [(16, 3), (64, 17), (256, 64), (512, 100), (1024, 64), (4096, 64), (4096, 1500)])
def test_fir_other_geometries(ctx, oracle, fft_len, ntaps):
    h = rand_c64(ntaps, ntaps, scale=0.3)                         # complex taps
    x = rand_c64(fft_len, 20011)
    f = Fir(ctx, h, fft_len)
    y = f.filter(ctx.vec(x)).to_host()
    assert oracle.evm_db(y, oracle.fir_direct_f64(h, x)) <= TOL_DB


def test_impulse_response_and_dc_gain(ctx, taps):
    f = Fir(ctx, taps, 2048)
    x = np.zeros(6000, np.complex64); x[0] = 1; x[3000] = 1j
    y = f.filter(ctx.vec(x)).to_host()
    assert np.abs(y[:64] - taps).max() < 1e-6 and np.abs(y[64:3000]).max() < 1e-6
    assert np.abs(y[3000:3064] - 1j * taps).max() < 1e-6
    ones = np.ones(10000, np.complex64)
    y = f.filter(ctx.vec(ones)).to_host()
    assert np.abs(y[63:] - 1).max() < 1e-5                        # unit DC gain after the transient


def test_history_continues_a_stream(ctx, oracle, taps):
    x = rand_c64(1, 50000)
    f = Fir(ctx, taps, 2048)
    whole = f.filter(ctx.vec(x)).to_host()
    cut = 20001
    d = ctx.vec(x)
    second = f.filter(d.slice(cut, x.size), hist=d.slice(cut - 63, cut)).to_host()
    truth = oracle.fir_direct_f64(taps, x)
    assert oracle.evm_db(second, truth[cut:]) <= TOL_DB
    assert oracle.evm_db(second, whole[cut:]) <= TOL_DB
    assert bits_equal(f.filter(x[cut:], hist=x[cut - 63:cut]), second)
    # without history the head differs (zero initial state), the rest agrees
    nohist = f.filter(d.slice(cut, x.size)).to_host()
    assert oracle.evm_db(nohist[63:], whole[cut + 63:]) <= TOL_DB
    assert oracle.evm_db(nohist[:63], whole[cut:cut + 63]) > -40


def test_hop_aligned_shards_are_bit_identical(ctx, taps):
    """shard boundaries aligned to the hop + (ntaps-1)-sample history read from the
    source => every shard runs exactly the blocks of the 1-GPU run => identical bits for any G."""
    from aether_primitives_amd.sharding import fir_shard
    n = 300000
    x = rand_c64(11, n)
    f = Fir(ctx, taps, 2048)
    whole = f.filter(ctx.vec(x)).to_host()
    for world in (2, 3, 8):
        parts = []
        for r in range(world):
            s = fir_shard(n, f.hop, f.ntaps, r, world)
            hist = ctx.vec(x[s["hist_lo"]:s["in_lo"]]) if s["in_lo"] else None
            parts.append(f.filter(ctx.vec(x[s["in_lo"]:s["out_hi"]]), hist=hist).to_host())
        assert bits_equal(np.concatenate(parts), whole), world


def test_fir_is_linear_and_shift_invariant(ctx, oracle, taps):
    f = Fir(ctx, taps, 2048)
    x, z = rand_c64(2, 40000), rand_c64(3, 40000)
    fx, fz = f.filter(ctx.vec(x)).to_host(), f.filter(ctx.vec(z)).to_host()
    comb = (np.float32(2.0) * x + z).astype(np.complex64)
    fc = f.filter(ctx.vec(comb)).to_host()
    assert oracle.evm_db(fc, 2.0 * fx.astype(np.complex128) + fz) <= TOL_DB
    shifted = np.concatenate([np.zeros(777, np.complex64), x])[:40000]
    fs = f.filter(ctx.vec(shifted)).to_host()
    assert oracle.evm_db(fs[777:], fx[:40000 - 777].astype(np.complex128)) <= TOL_DB


def test_fir_rejects_in_place(ctx, taps):
    f = Fir(ctx, taps, 2048)
    d = ctx.vec(rand_c64(1, 4096))
    with pytest.raises(ap.AetherError):
        f.filter(d, out=d)


def test_fir_rejects_any_overlap_of_output_and_input(ctx, taps):
    """not only out == in: blocks run concurrently and read windows their neighbours may already have overwritten, so a
    partially overlapping output (out = in + 100) is refused too -- and an output that touches the history, and the
    same for the decimating flavour.  Adjacent ranges (out starts where in ends) are fine."""
    f = Fir(ctx, taps, 2048)
    n = 6000
    big = ctx.vec(rand_c64(2, 3 * n))
    x = big.slice(0, n)
    for lo in (100, n - 1, 1):
        with pytest.raises(ap.AetherError, match="overlaps"):
            f.filter(x, out=big.slice(lo, lo + n))
    with pytest.raises(ap.AetherError, match="overlaps"):
        f.filter(big.slice(100, 100 + n), out=big.slice(0, n))          # out in front of in, reaching into it
    hist = big.slice(2 * n, 2 * n + 63)
    with pytest.raises(ap.AetherError, match="overlaps"):
        f.filter(x, out=big.slice(2 * n - n + 40, 2 * n + 40), hist=hist)   # out ends inside the history
    with pytest.raises(ap.AetherError, match="overlaps"):
        f.filter_decim(x, 4, out=big.slice(n - 10, n - 10 + n // 4))
    y = f.filter(x, out=big.slice(n, 2 * n))                            # adjacent: allowed
    assert bits_equal(y.to_host(), f.filter(x).to_host())


def test_c3_full_size(ctx, oracle, taps):
    """BASELINE config 3: 64 taps over 16 M samples (8456 blocks of 1984)."""
    n = 1 << 24
    x = oracle.synth_cnormal(815, n)
    f = Fir(ctx, taps, 2048)
    y = f.filter(ctx.vec(x)).to_host()
    ref = oracle.fir_ols_f32(taps, x, 2048, f.hop, threads=8)
    assert oracle.evm_db(y, ref) <= TOL_DB
    # f64 truth on windows: the very start, block seams, the ragged end
    for lo in (0, 1984 - 100, 1984 * 4000 - 50, n - 5000):
        hi = min(lo + 5000, n)
        hist = x[lo - 63:lo] if lo else None
        truth = oracle.fir_direct_f64(taps, x[lo:hi], hist=hist)
        assert oracle.evm_db(y[lo:hi], truth) <= TOL_DB
    # energy check over everything: low-pass at 0.25 fs keeps about half the power of white noise
    p = float(np.mean(np.abs(y[::7].astype(np.complex128)) ** 2))
    assert 0.4 < p < 0.6


@pytest.mark.parametrize("n", [512, 1024, 2048])
def test_correlator_chain(ctx, oracle, n):
    """benches/benches.rs:386-420: input = sig pattern repeated, sig = conj(pattern) zero-padded."""
    pat = np.array([-1 + 1j, 0, 1 - 1j, 1 - 1j], np.complex64)
    inp = np.tile(pat, n // 4)
    sig = np.zeros(n, np.complex64); sig[:4] = np.conj(pat)
    frames = np.concatenate([inp, rand_c64(n, n * 5)])
    f = HipFft(ctx, n)
    d = ctx.vec(frames)
    f.mul_chain(d, ctx.vec(sig))
    got = d.to_host()
    ref = oracle.correlate_frames(sig, frames)
    assert oracle.evm_db(got, ref) <= TOL_DB
    # the three separate trait calls give the same answer as the fused kernel (tolerance: different op order)
    e = ctx.vec(frames)
    sigd = ctx.vec(sig)
    for k in range(6):
        fr = e.slice(k * n, (k + 1) * n)
        fr.vec_rfft(f, Scale.NONE).vec_mul(sigd).vec_rifft(f, Scale.NONE)
    assert oracle.evm_db(got, e.to_host()) <= TOL_DB
    with pytest.raises(ap.LengthMismatch):
        f.mul_chain(d, ctx.vec(sig[:-1]))


@pytest.mark.parametrize("n,batch", [(100, 4), (100, 5000), (12, 70001), (8192, 9), (6000, 130)])
def test_correlator_chain_generic_length(ctx, oracle, n, batch):
    """lengths the fused kernel does not cover: fft -> ONE broadcast multiply launch -> ifft, any batch"""
    frames = rand_c64(1, n * batch); sig = rand_c64(2, n)
    f = HipFft(ctx, n, max_batch=batch)
    d = ctx.vec(frames); f.mul_chain(d, ctx.vec(sig))
    assert oracle.evm_db(d.to_host(), oracle.correlate_frames(sig, frames)) <= TOL_DB


@pytest.mark.parametrize("n,batch", [(1, 1), (7, 3), (100, 4096), (2048, 513), (300, 70000)])
def test_vec_mul_frames_bit_exact(ctx, oracle, n, batch):
    """frames[f] *= sig in one launch == vec_mul frame by frame (vecops.rs:99-112), bit for bit"""
    frames = rand_c64(3, n * batch); sig = rand_c64(4, n)
    got = ctx.vec(frames).vec_mul_frames(ctx.vec(sig)).to_host()
    exp = (frames.reshape(batch, n).copy())
    for f in range(min(batch, 64)):                       # the oracle frame by frame on a sample of frames ...
        assert bits_equal(got.reshape(batch, n)[f], oracle.vec_mul(exp[f], sig))
    tiled = oracle.vec_mul(frames, np.tile(sig, batch))   # ... and the whole thing against one long vec_mul
    assert bits_equal(got, tiled)
    with pytest.raises(AssertionError, match="Vectors must have same length"):
        ctx.vec(frames).vec_mul_frames(ctx.vec(np.concatenate([sig, sig[:1]])), frame_len=n)
    # frames and signal on odd 8-byte slots (no 16-byte accesses), even and odd frame lengths alike
    pad = ctx.vec(np.concatenate([frames[:1], frames])); spad = ctx.vec(np.concatenate([sig[:1], sig]))
    pad.slice(1, 1 + n * batch).vec_mul_frames(spad.slice(1, 1 + n), frame_len=n)
    assert bits_equal(pad.to_host()[1:], tiled) and bits_equal(pad.to_host()[:1], frames[:1])


@pytest.mark.parametrize("n,chunk", [(1000, 1984), (1984 * 7 + 5, 1984 * 2), (3_000_000, 1 << 20), (1 << 23, 0)])
def test_host_stream_pipeline_bit_identical(ctx, taps, n, chunk):
    """the double-buffered H2D | kernel | D2H pipeline over hop-aligned chunks gives
    exactly the bits of the one-shot run (every chunk runs the blocks the whole stream would)."""
    x = rand_c64(n, n)
    f = Fir(ctx, taps, 2048)
    y, st = f.filter_stream(x, chunk=chunk)
    assert bits_equal(y, f.filter(x))
    assert st["samples"] == n and st["chunks"] >= 1 and st["seconds"] > 0


def test_host_stream_pipeline_stage_report(ctx, taps):
    """the per-stage report of src/pipeline.rs:89-114 (items, rate, utilisation = active / elapsed) for the
    upload, kernel and download stages; same bits as the plain run, every stage busy for part of the time and never for
    longer than the run took (no thresholds on the utilisation itself: that is the box's business, not the test's)."""
    n = 24 << 20
    x = rand_c64(7, n)
    f = Fir(ctx, taps, 2048)
    y, st = f.filter_stream(x, chunk=1 << 20, report=True)
    assert bits_equal(y, f.filter_stream(x, chunk=1 << 20)[0])
    assert st["chunks"] == -(-n // (((1 << 20) + f.hop - 1) // f.hop * f.hop))
    for k in ("active_upload", "active_kernel", "active_download", "active_copy_in", "active_copy_out"):
        assert 0 < st[k] <= st["seconds"] * 1.05, (k, st)          # pageable numpy memory: both host stages ran
    assert st["pinned"] == 0
    assert len(st["lines"]) == 5 and all(l.startswith("Stage: ") and "Utilisation:" in l for l in st["lines"])


@pytest.mark.parametrize("n,dec", [(1984 * 4, 2), (30720, 30), (1 << 20, 8), (3_000_000, 3), (123456, 64), (1 << 24, 16)])
def test_fir_decimating_store(ctx, oracle, taps, n, dec):
    """aeth_fir_exec_decim == fir -> sampling::downsample (sampling.rs:28-42), bit for bit, one launch"""
    from aether_primitives_amd import sampling
    x = rand_c64(n % 1000 + dec, n)
    f = Fir(ctx, taps, 2048)
    d = ctx.vec(x)
    full = f.filter(d)
    ref = sampling.downsample(ctx, full, ctx.empty(n // dec))
    got = f.filter_decim(d, dec)
    assert got.n == n // dec and bits_equal(got.to_host(), ref.to_host())
    # nothing past the decimated output was touched
    guard = ctx.vec(np.full(n // dec + 8, 7 - 7j, np.complex64))
    f.filter_decim(d, dec, out=guard.slice(0, n // dec))
    assert (guard.to_host()[n // dec:] == 7 - 7j).all()


def test_fir_decim_rejects_uneven(ctx, taps):
    f = Fir(ctx, taps, 2048)
    with pytest.raises(ap.AetherError, match="Only even decimations"):
        f.filter_decim(ctx.vec(rand_c64(1, 7000)), 3, out=ctx.empty(2333))       # 7000 % 2333 != 0
