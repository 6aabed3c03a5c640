"""GPU parity: sampling::{interpolate, downsample} (reference src/sampling.rs:7-62), bit-exact."""
import numpy as np
import pytest

import aether_primitives_amd as ap
from aether_primitives_amd import sampling
from helpers import expand, load_kat, bits_equal, rand_c64

pytestmark = pytest.mark.gpu
KAT = load_kat()


@pytest.mark.parametrize("cid", ["interpolate_2_between", "interpolate_1_between"])
def test_interpolate_kat_host_appends(ctx, cid):
    c = KAT[cid]
    src = expand(c["self"])
    dst = [1 + 1j]                                   # the reference APPENDS (sampling.rs:17,23)
    n = sampling.interpolate(ctx, src, dst, c["n_between"])
    assert n == src.size + (src.size - 1) * c["n_between"] and len(dst) == n + 1
    assert bits_equal(np.array(dst[1:], np.complex64), expand(c["expect"]))


@pytest.mark.parametrize("n_src,nb", [(1, 3), (2, 0), (2, 1), (400, 3), (1024, 4), (2048, 4), (65536, 9), (100001, 2)])
@pytest.mark.parametrize("compat", [True, False])
def test_interpolate_bit_exact(ctx, oracle, n_src, nb, compat):
    src = rand_c64(n_src + nb, n_src)
    d = ctx.empty(n_src + (n_src - 1) * nb)
    n = sampling.interpolate(ctx, ctx.vec(src), d, nb, compat_im=compat)
    assert n == d.n
    assert bits_equal(d.to_host(), oracle.interpolate(src, nb, compat_im=compat))


def test_interpolate_frames_bit_exact(ctx, oracle):
    S, B, nb = 4096, 5, 9
    src = rand_c64(77, S * B)
    Lo = S + (S - 1) * nb
    d = ctx.empty(Lo * B)
    assert sampling.interpolate(ctx, ctx.vec(src), d, nb, frame_len=S) == Lo * B
    exp = np.concatenate([oracle.interpolate(f, nb) for f in src.reshape(B, S)])
    assert bits_equal(d.to_host(), exp)


def test_interpolate_errors(ctx):
    with pytest.raises(ap.LengthMismatch):           # empty src: the reference panics (sampling.rs:23)
        sampling.interpolate(ctx, ctx.empty(0), ctx.empty(4), 2)
    with pytest.raises(ap.LengthMismatch):           # destination too small
        sampling.interpolate(ctx, ctx.vec(rand_c64(1, 10)), ctx.empty(5), 2)


@pytest.mark.parametrize("cid", ["downsample_21_v_7", "downsample_16_v_4"])
def test_downsample_kat(ctx, cid):
    c = KAT[cid]
    src = np.array(c["src_ints"], np.int32)
    dst = np.zeros(c["n_dst"], np.int32)
    sampling.downsample(ctx, src, dst)
    assert dst.tolist() == c["expect_ints"]
    dst[:] = 0
    sampling.downsample_sb(ctx, src, dst)
    assert dst.tolist() == c["expect_ints"]


def test_downsample_rejects_uneven(ctx):
    c = KAT["downsample_7_v_3_fail"]
    with pytest.raises(ap.LengthMismatch, match=c["expect_error"]):
        sampling.downsample(ctx, np.array(c["src_ints"], np.int32), np.zeros(3, np.int32))


@pytest.mark.parametrize("n_src,n_dst", [(30720, 1024), (8096, 506), (1 << 20, 1 << 10), (7, 7), (4096, 1)])
def test_downsample_cf32_bit_exact(ctx, oracle, n_src, n_dst):
    src = rand_c64(n_src, n_src)
    d = ctx.empty(n_dst)
    sampling.downsample(ctx, ctx.vec(src), d)
    assert bits_equal(d.to_host(), oracle.downsample(src, n_dst))


@pytest.mark.parametrize("cid", ["bench_downsample_30720_1024", "bench_downsample_8096_512"])
def test_downsample_release_build_bench_shapes(ctx, oracle, cid):
    """The two shapes the reference benchmarks (benches/benches.rs:100-130; criterion = release build, where the
    debug_assert of sampling.rs:32-36 is gone and 8096 -> 512 runs with dec = 15): the reference's constant input,
    then seeded data against the oracle, both variants, device and host flavour."""
    c = KAT[cid]
    src, n_dst = expand(c["self"]), c["n_dst"]
    for fn in (sampling.downsample, sampling.downsample_sb):
        d = ctx.empty(n_dst)
        fn(ctx, ctx.vec(src), d, release=True)
        assert bits_equal(d.to_host(), expand(c["expect"]))
        rnd = rand_c64(n_dst, src.size)
        fn(ctx, ctx.vec(rnd), d, release=True)
        assert bits_equal(d.to_host(), oracle.downsample(rnd, n_dst, release=True, step_by=fn is sampling.downsample_sb))
        assert bits_equal(d.to_host(), rnd[::c["dec"]][:n_dst])
        h = np.zeros(n_dst, np.complex64)
        fn(ctx, rnd, h, release=True)
        assert bits_equal(h, d.to_host())
    if src.size % n_dst:
        with pytest.raises(ap.LengthMismatch, match="Only even decimations"):       # the debug build of the same call
            sampling.downsample(ctx, ctx.vec(src), ctx.empty(n_dst))


def test_downsample_release_build_edges(ctx, oracle):
    src = np.arange(7, dtype=np.int32)
    for sb, fn in ((False, sampling.downsample), (True, sampling.downsample_sb)):
        dst = np.zeros(3, np.int32)
        fn(ctx, src, dst, release=True)                          # 7 -> 3 panics in the debug build only (:162-169)
        assert dst.tolist() == oracle.downsample(src, 3, release=True, step_by=sb).tolist() == [0, 2, 4]
        with pytest.raises(ap.LengthMismatch):
            fn(ctx, src, np.zeros(0, np.int32), release=True)    # division by zero
        with pytest.raises(ap.LengthMismatch):
            fn(ctx, src[:0], np.zeros(4, np.int32), release=True)
    dst = np.full(5, -1, np.int32)
    sampling.downsample(ctx, src[1:3], dst, release=True)        # dec = 0: every dst[i] = src[0]
    assert dst.tolist() == [1] * 5
    with pytest.raises(ap.LengthMismatch, match="step_by"):
        sampling.downsample_sb(ctx, src[1:3], dst, release=True)


@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.float32, np.float64, np.complex128])
def test_downsample_generic_element(ctx, oracle, dtype):
    rng = np.random.default_rng(3)
    src = (rng.standard_normal(6000) * 100).astype(dtype)
    dst = np.zeros(300, dtype)
    sampling.downsample(ctx, src, dst)
    assert (dst.view(np.uint8) == oracle.downsample(src, 300).view(np.uint8)).all()
