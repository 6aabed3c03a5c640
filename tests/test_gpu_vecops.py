"""GPU parity: element-wise VecOps through the C ABI vs the oracle, bit-exact
(reference: src/vecops.rs:94-177; tolerance semantics src/lib.rs:36-47 -- at
-80 'dB' the macro is tighter than 1 ulp, so bit-exact is the honest bar)."""
import numpy as np
import pytest

import aether_primitives_amd as ap
from helpers import expand, load_kat, bits_equal, rand_c64

pytestmark = pytest.mark.gpu
KAT = load_kat()

BINARY = ["vec_add", "vec_sub", "vec_mul", "vec_div", "vec_clone"]
UNARY = ["vec_conj", "vec_mirror", "vec_zero"]


def _apply(ctx, c):
    v = ctx.vec(expand(c["self"]))
    op = c["op"]
    if op == "vec_scale": v.vec_scale(c["arg"])
    elif op in BINARY: getattr(v, op)(ctx.vec(expand(c["other"])))
    else: getattr(v, op)()
    return v.to_host()


@pytest.mark.parametrize("cid", ["vec_scale", "vec_mul", "vec_div", "vec_conj", "vec_add", "vec_sub",
                                 "vec_mirror", "vec_clone", "vec_zero"])
def test_reference_kat(ctx, cid):
    c = KAT[cid]
    ap.assert_evm(_apply(ctx, c), expand(c["expect"]), c["evm_db"])


def test_reference_doctest_chain(ctx):
    c = KAT["vecops_doctest_chain"]
    twos, ones = ctx.vec(expand(c["twos"])), ctx.vec(expand(c["ones"]))
    v = ctx.vec(expand(c["self"]))

    def im_minus_one(z):
        return np.complex64(complex(z.real, -1.0))
    (v.vec_div(twos).vec_mul(twos).vec_zero().vec_add(ones).vec_sub(twos).vec_clone(ones)
      .vec_mutate(im_minus_one).vec_conj().vec_mirror())
    ap.assert_evm(v.to_host(), expand(c["expect"]), c["evm_db"])


def test_reference_vec_mutate(ctx):
    c = KAT["vec_mutate"]
    v = ctx.vec(expand(c["self"]))
    state = {"x": 0}

    def f(z):
        r = np.complex64(z * np.float32(state["x"])); state["x"] += 1; return r
    v.vec_mutate(f)
    ap.assert_evm(v.to_host(), expand(c["expect"]), c["evm_db"])


# sizes: empty, 1, odd, ragged around the 2-sample vector width and the 4x unroll,
# C1 (4096), a streaming size
SIZES = [0, 1, 2, 3, 7, 100, 1023, 1024, 1025, 4096, 65537, (1 << 20) + 3]


@pytest.mark.parametrize("n", SIZES)
def test_binary_ops_bit_exact(ctx, oracle, n):
    a, b = rand_c64(10 + n, n), rand_c64(20 + n, n)
    for op in BINARY:
        got = getattr(ctx.vec(a), op)(ctx.vec(b)).to_host()
        assert bits_equal(got, getattr(oracle, op)(a, b)), op


@pytest.mark.parametrize("n", SIZES)
def test_unary_ops_bit_exact(ctx, oracle, n):
    a = rand_c64(30 + n, n)
    for op in UNARY:
        assert bits_equal(getattr(ctx.vec(a), op)().to_host(), getattr(oracle, op)(a)), op
    for s in (2.0, 0.1, -3.5e-3):
        assert bits_equal(ctx.vec(a).vec_scale(s).to_host(), oracle.vec_scale(a, s))


def test_misaligned_views_bit_exact(ctx, oracle):
    """Slices starting on an odd sample are 8- but not 16-byte aligned."""
    a, b = rand_c64(1, 5001), rand_c64(2, 5001)
    for sa, sb in [(1, 1), (1, 0), (0, 1), (3, 2)]:
        n = 4000
        da, db = ctx.vec(a), ctx.vec(b)
        da.slice(sa, sa + n).vec_mul(db.slice(sb, sb + n))
        expect = a.copy(); expect[sa:sa + n] = oracle.vec_mul(a[sa:sa + n], b[sb:sb + n])
        assert bits_equal(da.to_host(), expect)          # and nothing outside the slice moved
        dm = ctx.vec(a); dm.slice(sa, sa + n - 1).vec_mirror()
        expect = a.copy(); expect[sa:sa + n - 1] = oracle.vec_mirror(a[sa:sa + n - 1])
        assert bits_equal(dm.to_host(), expect)


def test_special_values_bit_exact(ctx, oracle):
    sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3.4e38, 1.0, -1.0], np.float32)
    re, im = np.meshgrid(sp, sp)
    a = (re + 1j * im).astype(np.complex64).reshape(-1)
    a.real, a.imag = re.reshape(-1), im.reshape(-1)
    b = np.roll(a, 7)
    for op in ["vec_add", "vec_sub", "vec_mul", "vec_div"]:
        got = getattr(ctx.vec(a), op)(ctx.vec(b)).to_host()
        exp = getattr(oracle, op)(a, b)
        # NaN payload/sign is not pinned by IEEE; compare NaN-ness there and bits elsewhere
        gn, en = np.isnan(got.view(np.float32)), np.isnan(exp.view(np.float32))
        assert (gn == en).all(), op
        assert (got.view(np.uint32)[~gn] == exp.view(np.uint32)[~en]).all(), op
    assert bits_equal(ctx.vec(a).vec_conj().to_host(), oracle.vec_conj(a))      # sign-bit flip incl. -0.0, NaN


def test_length_mismatch_raises_with_reference_message(ctx):
    a, b = ctx.vec(rand_c64(1, 10)), ctx.vec(rand_c64(2, 9))
    for op in BINARY:
        with pytest.raises(ap.LengthMismatch, match="Vectors must have same length"):
            getattr(a, op)(b)
    assert bits_equal(a.to_host(), rand_c64(1, 10))      # untouched


def test_c1_chain_4096(ctx, oracle):
    """BASELINE config 1: add -> mul -> conj on 4096-sample vectors."""
    v, a, b = (oracle.synth_cnormal(815 + i, 4096) for i in range(3))
    got = ctx.vec(v).vec_add(ctx.vec(a)).vec_mul(ctx.vec(b)).vec_conj().to_host()
    assert bits_equal(got, oracle.vec_conj(oracle.vec_mul(oracle.vec_add(v, a), b)))


def test_host_slice_flavour(ctx, oracle):
    a, b = rand_c64(5, 3001), rand_c64(6, 3001)
    h = a.copy()
    ap.HostVec(ctx, h).vec_add(b).vec_mul(b).vec_conj().vec_scale(0.5).vec_mirror()
    exp = oracle.vec_mirror(oracle.vec_scale(oracle.vec_conj(oracle.vec_mul(oracle.vec_add(a, b), b)), 0.5))
    assert bits_equal(h, exp)
    with pytest.raises(ap.LengthMismatch):
        ap.HostVec(ctx, h).vec_sub(b[:-1])
    z = a.copy(); ap.HostVec(ctx, z).vec_zero(); assert not z.any()
    c = a.copy(); ap.HostVec(ctx, c).vec_clone(b); assert bits_equal(c, b)
    d = a.copy(); ap.HostVec(ctx, d).vec_div(b); assert bits_equal(d, oracle.vec_div(a, b))


def test_mirror_frames(ctx, oracle):
    x = rand_c64(9, 8 * 2048)
    got = ctx.vec(x).vec_mirror_frames(2048).to_host()
    exp = np.concatenate([oracle.vec_mirror(f) for f in x.reshape(8, 2048)])
    assert bits_equal(got, exp)
    assert bits_equal(ctx.vec(x[:7 * 5]).vec_mirror_frames(5).to_host(),
                      np.concatenate([oracle.vec_mirror(f) for f in x[:35].reshape(7, 5)]))


def test_scale_kat_and_apply(ctx):
    for cid in ["scale_none", "scale_sn", "scale_n", "scale_x2"]:
        c = KAT[cid]
        v = ctx.vec(expand(c["self"]))
        s = {"None": ap.Scale.NONE, "SN": ap.Scale.SN, "N": ap.Scale.N}.get(c["scale"]) or ap.Scale.X(c["x"])
        s.scale(v)
        ap.assert_evm(v.to_host(), expand(c["expect"]), c["evm_db"])


# ---- the chained methods as one pass over memory (aeth_vec_chain) -----------------------------------------------------
def _apply_chain(oracle, v, links):
    for name, arg in links:
        v = getattr(oracle, name)(v, arg) if arg is not None else getattr(oracle, name)(v)
    return v


@pytest.mark.parametrize("n", [1, 2, 3, 255, 4096, 100003, 1 << 20])
def test_fused_chain_is_the_separate_calls_bit_for_bit(ctx, oracle, n):
    """BASELINE config 1's chain (add -> mul -> conj) and a longer one with every link kind, against the oracle applied
    link by link and against the same chain as separate device calls"""
    v, a, b, c = (rand_c64(s + n, n) for s in (1, 2, 3, 4))
    da, db, dc = ctx.vec(a), ctx.vec(b), ctx.vec(c)
    got = ctx.vec(v).fused().vec_add(da).vec_mul(db).vec_conj().run().to_host()
    assert bits_equal(got, oracle.vec_conj(oracle.vec_mul(oracle.vec_add(v, a), b)))
    assert bits_equal(got, ctx.vec(v).vec_add(da).vec_mul(db).vec_conj().to_host())
    links = [("vec_scale", 0.37), ("vec_div", b), ("vec_sub", c), ("vec_conj", None), ("vec_mul", a), ("vec_add", c), ("vec_scale", -2.5), ("vec_div", a)]
    dev = {id(a): da, id(b): db, id(c): dc}
    ch = ctx.vec(v).fused()
    for name, arg in links:
        ch = getattr(ch, name)(dev[id(arg)]) if isinstance(arg, np.ndarray) else (getattr(ch, name)(arg) if arg is not None else getattr(ch, name)())
    assert bits_equal(ch.run().to_host(), _apply_chain(oracle, v, links))


def test_fused_chain_longer_than_one_pass_and_overwriting_links(ctx, oracle):
    n = 5000
    v, a, b = rand_c64(1, n), rand_c64(2, n), rand_c64(3, n)
    da, db = ctx.vec(a), ctx.vec(b)
    ch = ctx.vec(v).fused()
    want = v
    for k in range(19):                                     # 19 links: three passes of at most eight
        if k % 3 == 0: ch = ch.vec_add(da); want = oracle.vec_add(want, a)
        elif k % 3 == 1: ch = ch.vec_scale(0.75); want = oracle.vec_scale(want, 0.75)
        else: ch = ch.vec_mul(db); want = oracle.vec_mul(want, b)
    assert bits_equal(ch.run().to_host(), want)
    # clone / zero in front: self is not even read
    assert bits_equal(ctx.vec(v).fused().vec_clone(da).vec_mul(db).run().to_host(), oracle.vec_mul(a, b))
    assert bits_equal(ctx.vec(v).fused().vec_zero().vec_add(db).run().to_host(), oracle.vec_add(np.zeros(n, np.complex64), b))
    assert bits_equal(ctx.vec(v).fused().vec_add(da).vec_zero().run().to_host(), np.zeros(n, np.complex64))
    with ctx.vec(v).fused() as f:                           # the `with` form runs on exit
        f.vec_conj()
    assert bits_equal(f.vec.to_host(), oracle.vec_conj(v))
    assert bits_equal(ctx.vec(v).fused().run().to_host(), v)            # an empty chain does nothing


def test_fused_chain_on_misaligned_views(ctx, oracle):
    a, b, c = rand_c64(1, 6001), rand_c64(2, 6001), rand_c64(3, 6001)
    for sa, sb, sc in [(1, 1, 1), (1, 0, 1), (0, 1, 0), (3, 2, 5)]:
        n = 5000
        da, db, dc = ctx.vec(a), ctx.vec(b), ctx.vec(c)
        da.slice(sa, sa + n).fused().vec_mul(db.slice(sb, sb + n)).vec_sub(dc.slice(sc, sc + n)).run()
        expect = a.copy(); expect[sa:sa + n] = oracle.vec_sub(oracle.vec_mul(a[sa:sa + n], b[sb:sb + n]), c[sc:sc + n])
        assert bits_equal(da.to_host(), expect)


def test_fused_chain_errors(ctx):
    v, short = ctx.vec(rand_c64(1, 100)), ctx.vec(rand_c64(2, 99))
    with pytest.raises(ap.LengthMismatch, match="Vectors must have same length"):
        v.fused().vec_conj().vec_mul(short).run()           # the link that would have panicked in the reference
    with pytest.raises(ap.AetherError, match="overlaps self"):
        v.fused().vec_add(v.slice(0, 100)).run()
    big = ctx.vec(rand_c64(3, 300))
    with pytest.raises(ap.AetherError, match="overlaps self"):
        big.slice(0, 200).fused().vec_mul(big.slice(100, 300)).run()
