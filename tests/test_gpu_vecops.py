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
