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
    b = KAT["generic_bpsk"]                                      # :157-172
    out = modulation.bpsk(ctx).modulate(np.array(b["bits"], np.uint8)).to_host()
    assert bits_equal(out, expand(b["expect"]))


def test_reference_naive_demod(ctx):
    """modulation.rs:184-196: bits from gen_range(0u8, 1u8) -- all zero whatever the seed -- through qpsk().modulate and
    demod_naive (the QPSK specialisation, compat output) come back equal."""
    c = KAT["naive_demod"]
    q = modulation.qpsk(ctx)
    bits = np.array(c["bits"], np.uint8)
    for _seed in c["seeds"]:
        sym = q.modulate(bits)
        assert bits_equal(sym.to_host(), expand(c["expect"]))
        assert q.demod_naive(sym, compat=True).to_host().tolist() == c["expect_bits"]


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


def test_demod_unordered_distances_follow_min_by(ctx, oracle):
    """min_by(|d, e| d.partial_cmp(e).unwrap_or(Ordering::Greater)) (modulation.rs:46, :139) replaces the running
    minimum whenever the comparison says Greater: on a strictly smaller distance and on an unordered pair.  A sample
    with a NaN component makes every distance NaN, so it decodes as the LAST candidate scanned (QPSK: index 3, BPSK:
    index 1); equal infinite distances keep the first."""
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    sym = np.array([complex(nan, 0.3), complex(-0.2, nan), complex(nan, nan), complex(inf, inf), complex(-inf, 0.0),
                    0.7 + 0.7j, complex(0.1, -0.9)], np.complex64)
    q = modulation.qpsk(ctx)
    got = q.demod_naive(ctx.vec(sym), compat=False).to_host().reshape(-1, 2)
    assert got[:3].tolist() == [[1, 1]] * 3                     # index 3
    assert got[3].tolist() == [0, 0] and got[4].tolist() == [0, 0]   # all distances +inf: Equal, the first stays
    assert got[5].tolist() == [0, 0] and got[6].tolist() == [0, 1]
    assert (q.demod_naive(ctx.vec(sym)).to_host() == oracle.demod_naive(sym, 2)).all()
    b = modulation.bpsk(ctx)
    assert b.demod_naive(ctx.vec(sym)).to_host().tolist()[:5] == [1, 1, 1, 0, 0]
    assert (b.demod_naive(ctx.vec(sym)).to_host() == oracle.demod_naive(sym, 1)).all()
    # a generic 3-bit table: the trait default scans BITS_PER_SYMBOL * 2 = 6 candidates, a NaN sample takes the sixth
    tab = np.exp(2j * np.pi * np.arange(8) / 8).astype(np.complex64)
    g = modulation.table(ctx, tab)
    want = oracle.demod_naive(sym, 3, table=tab)
    assert (g.demod_naive(ctx.vec(sym)).to_host() == want).all()
    assert want[:3].tolist() == [1, 0, 1]                       # index 5 = 0b101, least significant bit first


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


@pytest.mark.parametrize("bps", [3, 4, 6, 8])
def test_generic_tables_follow_the_trait_defaults(ctx, oracle, bps):
    """trait Modulation's default methods for any table type (modulation.rs:94-149): 8-PSK, 16/64/256-QAM-like
    tables; modulate is a gather, demod_naive scans BITS_PER_SYMBOL*2 candidates in compat mode (:135) and all of
    them otherwise; bit i of the index goes to output byte i (:143)."""
    rng = np.random.default_rng(bps)
    m = 1 << bps
    tab = (rng.standard_normal(m) + 1j * rng.standard_normal(m)).astype(np.complex64)
    mod = modulation.table(ctx, tab)
    assert mod.bits_per_symbol() == bps
    nsym = 50_003
    bits = rng.integers(0, 256, nsym * bps, dtype=np.uint8)          # any byte: index() takes bit % 2 (:109)
    tx = mod.modulate(bits)
    assert bits_equal(tx.to_host(), oracle.modulate(bits, bps, tab))
    noisy = (tx.to_host() + 0.05 * (rng.standard_normal(nsym) + 1j * rng.standard_normal(nsym))).astype(np.complex64)
    for compat in (True, False):
        got = mod.demod_naive(ctx.vec(noisy), compat=compat).to_host()
        assert (got == oracle.demod_naive(noisy, bps, tab, compat=compat)).all()
    # with every candidate scanned and little noise the transmitted bits come back
    back = mod.demod_naive(ctx.vec(tx.to_host()), compat=False).to_host()
    assert (back == (bits % 2)).all()
    with pytest.raises(ap.LengthMismatch):
        mod.modulate(np.zeros(bps + 1, np.uint8))


@pytest.mark.parametrize("rounds", [7, 10])
def test_philox_known_answers_on_the_device(ctx, oracle, rounds):
    """The generator's integer stage as the GPU runs it (seven rounds; ten, the other published set, too), against the
    Random123 known-answer vectors (fixtures) and, on random counters/keys, against the oracle's independent restatement."""
    import json, os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", f"philox4x32_{rounds}_kat.json")))["cases"]
    ctr = np.array([[int(x, 16) for x in c["counter"]] for c in kat], np.uint32)
    key = np.array([[int(x, 16) for x in c["key"]] for c in kat], np.uint32)
    got = noise.philox4x32(ctx, ctr, key, rounds)
    for g, c in zip(got, kat):
        assert [f"{v:08x}" for v in g] == c["expected"], c["name"]
    rng = np.random.default_rng(5)
    ctr = rng.integers(0, 2 ** 32, (5000, 4), dtype=np.uint64).astype(np.uint32)
    key = rng.integers(0, 2 ** 32, (5000, 2), dtype=np.uint64).astype(np.uint32)
    got = noise.philox4x32(ctx, ctr, key, rounds)
    for i in range(0, 5000, 37):
        assert (got[i] == oracle.philox4x32(ctr[i], key[i], rounds)).all()
    if rounds == 10:
        assert (noise.philox4x32_10(ctx, ctr[:8], key[:8]) == got[:8]).all()
    with pytest.raises(ap.AetherError):
        noise.philox4x32(ctx, ctr[:1], key[:1], 8)


def test_box_muller_stage_over_its_whole_radius_argument(ctx, oracle):
    """The generator's floating-point stage on explicit words (aeth_rng_normal_pairs): every one of the 2^24 values
    the radius can be drawn from -- the device takes r = sqrt(-2 ln u) from v_rsq_f32 plus one correcting step, the
    oracle from sqrtf: they must agree on ALL of them, not on the ones a stream happens to hit -- each with its own
    angle word; then the corners of the angle (0, +-pi/2, pi, the words around them) and empty input."""
    rng = np.random.default_rng(3)
    step = 1 << 22
    for k0 in range(0, 1 << 24, step):
        a = (np.arange(k0, k0 + step, dtype=np.uint64) << 8).astype(np.uint32) | rng.integers(0, 256, step, dtype=np.uint32)
        b = rng.integers(0, 2 ** 32, step, dtype=np.uint64).astype(np.uint32)
        got = noise.normal_pairs(ctx, a, b)
        assert np.isfinite(got.view(np.float32)).all()
        assert bits_equal(got, oracle.rng_normal_pairs(a, b)), f"radius words {k0:#x}.."
    corners = np.array([0, 1, 0x3fffffff, 0x40000000, 0x40000001, 0x7fffffff, 0x80000000, 0x80000001, 0xbfffffff,
                        0xc0000000, 0xc0000001, 0xffffffff], np.uint32)
    a = np.full(corners.size, 0x12345678, np.uint32)
    got = noise.normal_pairs(ctx, a, corners)
    assert bits_equal(got, oracle.rng_normal_pairs(a, corners))
    r = np.sqrt(-2 * np.log(((0x12345678 >> 8) | 1) / 2.0 ** 24))
    th = np.pi * corners.astype(np.int32).astype(np.float64) / 2.0 ** 31
    assert np.abs(got - r * np.exp(1j * th)).max() < 5e-6
    assert noise.normal_pairs(ctx, np.zeros(0, np.uint32), np.zeros(0, np.uint32)).size == 0


def test_generator_fixture_on_the_device(ctx):
    """tests/golden/generator_v3.json: the committed samples of the build's own generator, drawn by the device"""
    import json, os
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "generator_v3.json")))
    for s in fx["streams"]:
        g = noise.new(ctx, s["power"], s["seed"]); g.offset = s["offset"]
        got = g.fill(ctx.empty(s["n"])).to_host().view(np.uint32)
        assert [f"{v:08x}" for v in got] == s["fill"], (s["seed"], s["offset"])
    a = np.array([int(x, 16) for x in fx["pairs"]["a"]], np.uint32); b = np.array([int(x, 16) for x in fx["pairs"]["b"]], np.uint32)
    assert [f"{v:08x}" for v in noise.normal_pairs(ctx, a, b).view(np.uint32)] == fx["pairs"]["normal"]


@pytest.mark.parametrize("n,offset", [(1, 0), (2, 1), (7, 0), (4097, 3), (1 << 20, 1 << 33)])
def test_awgn_fill_bit_exact(ctx, oracle, n, offset):
    """Awgn::fill / iter (noise.rs:61-84): next() scaled once; stream positions continue across calls"""
    for power in (1.0, 0.01):
        g = noise.new(ctx, power, 815); g.offset = offset
        d = ctx.vec(np.full(n + 1, 9 + 9j, np.complex64))
        g.fill(d.slice(1, n + 1))                                    # 8- but not 16-byte aligned
        h = d.to_host()
        assert h[0] == 9 + 9j and bits_equal(h[1:], oracle.awgn_fill(n, power, 815, offset))
        assert g.offset == offset + n
    it = noise.new(ctx, 0.5, 7).iter(chunk=64)
    first = np.array([next(it) for _ in range(150)], np.complex64)  # crosses two chunk boundaries
    assert bits_equal(first, oracle.awgn_fill(150, 0.5, 7, 0))


def test_awgn_distribution_ks_and_tails(ctx):
    """2 x 2^24 normals from the device generator: Kolmogorov-Smirnov against N(0,1), tail counts beyond 3, 4 and
    5 sigma inside 5-sigma binomial bands, the Box-Muller radius against chi-square(2), and no serial correlation."""
    from scipy import stats
    n = 1 << 24
    z = noise.new(ctx, 1.0, 20261004).fill(ctx.empty(n)).to_host()
    v = np.concatenate([z.real, z.imag]).astype(np.float64)
    assert np.isfinite(v).all()
    d, _ = stats.kstest(v[::4], "norm")                              # 8.4 M values: sorting all 33 M adds nothing
    assert d < 1.36 / np.sqrt(v.size / 4) * 1.5                      # 5 % critical value, with headroom for f32 rounding
    for k in (3.0, 4.0, 5.0):
        p = 2 * stats.norm.sf(k)
        cnt = int((np.abs(v) > k).sum())
        exp, sd = v.size * p, np.sqrt(v.size * p * (1 - p))
        assert abs(cnt - exp) <= 5 * sd + 1, (k, cnt, exp)
    assert np.abs(v).max() < 6.2                                      # 24-bit uniforms: |z| <= sqrt(2 ln 2^24) = 5.77
    r2 = (z.real.astype(np.float64) ** 2 + z.imag.astype(np.float64) ** 2)[::8]
    d2, _ = stats.kstest(r2, "chi2", args=(2,))
    assert d2 < 1.36 / np.sqrt(r2.size) * 1.5
    ph = np.angle(z[::8].astype(np.complex128))
    d3, _ = stats.kstest((ph + np.pi) / (2 * np.pi), "uniform")
    assert d3 < 1.36 / np.sqrt(ph.size) * 1.5
    x = z.real[: 1 << 22].astype(np.float64)
    for lag in (1, 2, 3, 64):
        assert abs(np.corrcoef(x[:-lag], x[lag:])[0, 1]) < 5 / np.sqrt(x.size)
    assert abs(np.corrcoef(z.real[: 1 << 22], z.imag[: 1 << 22])[0, 1]) < 5 / np.sqrt(1 << 22)


@pytest.mark.parametrize("bps,nsym,offset", [(2, 1, 0), (2, 7, 3), (2, 4096, 0), (1, 100001, 5), (2, 1 << 20, 1 << 33), (1, 2, 1)])
def test_modulate_awgn_fused_bit_exact(ctx, oracle, bps, nsym, offset):
    """modulate + Awgn::apply in one pass (examples/modem.rs:19-26) == the two calls, bit for bit; the stream
    position advances like the generator's state"""
    rng = np.random.default_rng(nsym)
    bits = rng.integers(0, 2, nsym * bps, dtype=np.uint8)
    mod = modulation.qpsk(ctx) if bps == 2 else modulation.bpsk(ctx)
    for power in (1.0, 0.01):
        g1 = noise.new(ctx, power, 815); g1.offset = offset
        two = mod.modulate(bits); g1.apply(two)
        g2 = noise.new(ctx, power, 815); g2.offset = offset
        buf = ctx.vec(np.full(nsym + 2, 3 + 3j, np.complex64))
        mod.modulate_awgn(bits, g2, out=buf.slice(1, nsym + 1))        # 8- but not 16-byte aligned destination
        h = buf.to_host()
        assert bits_equal(h[1:nsym + 1], two.to_host()) and h[0] == 3 + 3j and h[-1] == 3 + 3j
        assert g2.offset == g1.offset == offset + nsym
        assert bits_equal(mod.modulate_awgn(bits, noise.new(ctx, power, 815)).to_host(),
                          oracle.awgn_apply(oracle.modulate(bits, bps), power, 815, 0))


@pytest.mark.parametrize("n,frames,bps", [(2048, 64, 2), (1024, 33, 2), (4096, 5, 1), (2048, 3, 1), (256, 40, 2), (100, 12, 2)])
def test_correlate_then_demod_in_one_call(ctx, oracle, n, frames, bps):
    """rfft * sig -> rifft -> demod_naive with only the bits written: same bits as the two calls (both compat modes),
    input frames untouched; lengths outside 1024..4096 take the two-step path"""
    rng = np.random.default_rng(n + frames)
    bits = rng.integers(0, 2, bps * n * frames, dtype=np.uint8)
    mod = modulation.qpsk(ctx) if bps == 2 else modulation.bpsk(ctx)
    tx = mod.modulate_awgn(bits, noise.new(ctx, 0.3, 815))             # strong noise: decisions near the boundaries too
    x = tx.to_host()
    f = HipFft(ctx, n, max_batch=frames)
    sig = ctx.vec(rand_c64(9, n))
    ref = ctx.vec(x); f.mul_chain(ref, sig)
    for compat in (True, False):
        want = mod.demod_naive(ref, compat=compat).to_host()
        got = mod.correlate_demod(f, tx, sig, compat=compat).to_host()
        assert (got == want).all()
    assert bits_equal(tx.to_host(), x)
    # scaled variant (Scale::SN both ways)
    ref2 = ctx.vec(x); f.mul_chain(ref2, sig, Scale.SN, Scale.SN)
    assert (mod.correlate_demod(f, tx, sig, Scale.SN, Scale.SN).to_host() == mod.demod_naive(ref2).to_host()).all()


def test_correlate_demod_with_a_custom_table(ctx):
    """The fused chain's decisions for a QPSK table that is NOT of the separable form {(a,c), (b,c), (a,d), (b,d)} (a
    rotated constellation: the shared-squares path must not be taken) and for a separable one with unequal levels, with
    values on the decision boundaries, NaN and Inf among the samples: same bits as mul_chain + demod_naive."""
    n, frames = 2048, 8
    f = HipFft(ctx, n, max_batch=frames)
    x = rand_c64(5, n * frames, scale=1.5)
    x[::97] = 0                                                    # exact ties between all four candidates
    x[5] = complex(np.nan, 1.0); x[6] = complex(np.inf, -np.inf)
    ident = np.zeros(n, np.complex64); ident[:] = 1.0 / n          # sig = 1/N everywhere: the chain returns the frame itself ...
    for tab in (np.array([1 + 0j, 0 + 1j, 0 - 1j, -1 + 0j], np.complex64),                     # rotated by 45 degrees: not separable
                np.array([2 + 1j, -0.5 + 1j, 2 - 3j, -0.5 - 3j], np.complex64)):               # separable, unequal levels
        mod = modulation.table(ctx, tab)
        for sig in (ctx.vec(rand_c64(9, n)), ctx.vec(ident)):
            tx, ref = ctx.vec(x), ctx.vec(x)
            f.mul_chain(ref, sig)
            for compat in (True, False):
                want = mod.demod_naive(ref, compat=compat).to_host()
                got = mod.correlate_demod(f, tx, sig, compat=compat).to_host()
                assert (got == want).all()


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
