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


@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.float32, np.float64, np.complex128])
def test_downsample_generic_element(ctx, oracle, dtype):
    rng = np.random.default_rng(3)
    src = (rng.standard_normal(6000) * 100).astype(dtype)
    dst = np.zeros(300, dtype)
    sampling.downsample(ctx, src, dst)
    assert (dst.view(np.uint8) == oracle.downsample(src, 300).view(np.uint8)).all()
