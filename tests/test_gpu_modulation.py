"""GPU parity: Modulation::{modulate, demod_naive} for the generic BPSK/QPSK tables
(reference src/modulation.rs:5-149; SURVEY 8f 'next' #1), bit-exact, and BASELINE config 4's
chain QPSK mod -> AWGN -> FFT-2048 correlate -> hard demod on the device."""
import numpy as np
import pytest

import aether_primitives_amd as ap
from aether_primitives_amd import HipFft, Scale, modulation, noise
from helpers import expand, load_kat, bits_equal, rand_c64

pytestmark = pytest.mark.gpu
KAT = load_kat()


def test_reference_qpsk_table(ctx):
    c = KAT["qpsk_table"]                                        # modulation.rs:174-181
    out = modulation.qpsk(ctx).modulate(np.array(c["bits"], np.uint8)).to_host()
    assert bits_equal(out, expand(c["expect"]))
    out = modulation.bpsk(ctx).modulate(np.array([0, 1, 0, 1], np.uint8)).to_host()      # :157-172
    assert bits_equal(out, np.array([1 + 1j, -1 - 1j, 1 + 1j, -1 - 1j], np.complex64))


@pytest.mark.parametrize("bps", [1, 2])
@pytest.mark.parametrize("n", [0, 2, 100, 8000, 1 << 20])
def test_modulate_demod_bit_exact(ctx, oracle, bps, n):
    rng = np.random.default_rng(n + bps)
    bits = rng.integers(0, 2, n, dtype=np.uint8)
    m = modulation.qpsk(ctx) if bps == 2 else modulation.bpsk(ctx)
    sym = m.modulate(bits)
    assert bits_equal(sym.to_host(), oracle.modulate(bits, bps))
    noisy = (sym.to_host() + rand_c64(n, sym.n, scale=0.8)).astype(np.complex64)
    for compat in (True, False):
        got = m.demod_naive(ctx.vec(noisy), compat=compat).to_host()
        assert (got == oracle.demod_naive(noisy, bps, compat=compat)).all()
    # noiseless round trip (the reference's naive_demod test, modulation.rs:183-196, with real bits)
    back = m.demod_naive(sym, compat=False).to_host()
    assert (back == bits).all()


def test_demod_quirk_and_ties(ctx, oracle):
    q = modulation.qpsk(ctx)
    d = q.demod_naive(ctx.vec(modulation.GENERIC_QPSK_TABLE)).to_host()
    assert d.tolist() == [0, 0, 1, 0, 0, 2, 1, 2]               # `idx & 1u8 << 1` (modulation.rs:54)
    ties = np.array([0, 1j, 1, -1, -1j, 0.5 + 0.5j], np.complex64)     # equidistant points: first minimum wins
    assert (q.demod_naive(ctx.vec(ties)).to_host() == oracle.demod_naive(ties, 2)).all()
    with pytest.raises(ap.LengthMismatch):
        q.modulate(np.array([0, 1, 1], np.uint8))               # not a multiple of BITS_PER_SYMBOL


@pytest.mark.parametrize("n,offset", [(1, 0), (2, 0), (7, 0), (7, 3), (4096, 0), (100001, 5), (1 << 20, 1 << 33)])
def test_awgn_apply_bit_exact(ctx, oracle, n, offset):
    """Awgn::apply (noise.rs:53-59) around the build's counter-based generator: bit for bit
    against its CPU restatement, incl. odd stream offsets, odd lengths, misaligned slices."""
    x = rand_c64(n, n + 1)
    for power in (1.0, 0.01):
        d = ctx.vec(x)
        g = noise.new(ctx, power, 815); g.offset = offset
        g.apply(d.slice(0, n))
        assert bits_equal(d.to_host()[:n], oracle.awgn_apply(x[:n], power, 815, offset))
        assert bits_equal(d.to_host()[n:], x[n:])                       # nothing past the slice moved
        d = ctx.vec(x); g = noise.new(ctx, power, 815); g.offset = offset
        g.apply(d.slice(1, n + 1))                                      # 8- but not 16-byte aligned view
        assert bits_equal(d.to_host()[1:], oracle.awgn_apply(x[1:], power, 815, offset))


def test_awgn_stream_continues_and_statistics(ctx, oracle):
    n = 1 << 21
    z = ctx.vec(np.zeros(n, np.complex64))
    g = noise.generator(ctx)                                            # power 1, seed 815 (noise.rs:8-11)
    g.apply(z.slice(0, 1000)); g.apply(z.slice(1000, 1001)); g.apply(z.slice(1001, n))   # three calls = one stream
    h = z.to_host()
    assert bits_equal(h, oracle.awgn_apply(np.zeros(n, np.complex64), 1.0, 815, 0))
    assert abs(h.real.mean()) < 3e-3 and abs(h.imag.mean()) < 3e-3
    assert abs(h.real.var() - 1) < 5e-3 and abs(h.imag.var() - 1) < 5e-3
    assert abs(np.corrcoef(h.real, h.imag)[0, 1]) < 3e-3
    # the reference scales twice (noise.rs:41-42, 58): amplitude ~ power, variance ~ power^2
    z2 = ctx.vec(np.zeros(n, np.complex64)); noise.new(ctx, 0.01, 1).apply(z2)
    assert abs(z2.to_host().real.var() / 1e-4 - 1) < 1e-2
    other = ctx.vec(np.zeros(16, np.complex64)); noise.new(ctx, 1.0, 816).apply(other)
    assert not bits_equal(other.to_host(), h[:16])                      # different seed, different stream


def test_c4_chain(ctx, oracle):
    """QPSK mod -> AWGN (power 0.01, examples/modem.rs:25) -> per 2048-frame rfft * conj-reference
    -> rifft -> hard demod.  Correlating against a unit impulse reference leaves the frame
    scaled by N, so the demodulated bits must equal the transmitted ones."""
    n, frames = 2048, 64
    rng = np.random.default_rng(815)
    bits = rng.integers(0, 2, 2 * n * frames, dtype=np.uint8)
    q = modulation.qpsk(ctx)
    tx = q.modulate(bits)
    noise.new(ctx, 0.01, 815).apply(tx)                                 # examples/modem.rs:25
    ref = np.zeros(n, np.complex64); ref[0] = 1
    f = HipFft(ctx, n)
    sig = ctx.vec(ref); f.ifwd(sig, Scale.NONE); sig.vec_conj()            # conj of the reference spectrum
    rx = ctx.vec(tx.to_host())
    f.mul_chain(rx, sig)
    got = q.demod_naive(rx, compat=False).to_host()
    assert (got == bits).all()
    # the demod kernel agrees bit for bit with the oracle on the very same correlator output
    assert (q.demod_naive(rx).to_host() == oracle.demod_naive(rx.to_host(), 2)).all()
