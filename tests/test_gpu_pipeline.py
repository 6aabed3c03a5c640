"""GPU parity: the host pipeline with a generic compute stage (src/pipeline.rs:24-41 `add_stage`, :123-137 `new`).

Every op the stage can be -- FIR, batched FFT frames, the correlator chain, correlate + demod (8 B in, 2 B out),
FFT + interpolate (1 in, 10 out) -- streamed from HOST memory chunk by chunk must give exactly the bits of its device
flavour on the whole slice: pageable memory (staged through the context's pinned pools), pool elements (direct) and
chunk sizes that do and do not divide the stream."""
import ctypes as C

import numpy as np
import pytest

import aether_primitives_amd as ap
from aether_primitives_amd import Scale, HipFft, Fir, modulation, pipeline, pool, sampling
from helpers import bits_equal, rand_c64

pytestmark = pytest.mark.gpu
N = 2048


@pytest.fixture(scope="module")
def plan(ctx):
    return HipFft(ctx, N, max_batch=64)


@pytest.fixture(scope="module")
def sig(ctx):
    return ctx.vec(rand_c64(5, N, scale=0.3))


@pytest.mark.parametrize("frames,chunk", [(1, 0), (37, 0), (37, N * 5), (600, 0), (600, N * 64)])
def test_fft_frames_stream(ctx, plan, frames, chunk):
    x = rand_c64(frames, frames * N)
    want = ctx.vec(x)
    plan.ifwd(want, Scale.SN)
    y, st = pipeline.run(pipeline.Stage.fft(plan, Scale.SN), x, chunk=chunk)
    assert bits_equal(y, want.to_host()) and st["samples"] == x.size
    if chunk:
        assert st["chunks"] == -(-x.size // chunk)


@pytest.mark.parametrize("frames,chunk", [(3, 0), (200, N * 33)])
def test_correlator_chain_stream_runs_in_place_on_the_slot(ctx, plan, sig, frames, chunk):
    x = rand_c64(100 + frames, frames * N)
    want = ctx.vec(x)
    plan.mul_chain(want, sig)
    y, _ = pipeline.run(pipeline.Stage.mul_chain(plan, sig), x, chunk=chunk)
    assert bits_equal(y, want.to_host())
    y2, st = pipeline.run(pipeline.Stage.mul_chain(plan, sig, Scale.SN, Scale.SN), x, chunk=chunk, report=True)
    want = ctx.vec(x)
    plan.mul_chain(want, sig, Scale.SN, Scale.SN)
    assert bits_equal(y2, want.to_host()) and len(st["lines"]) == 5 and "compute" in st["lines"][2]


@pytest.mark.parametrize("bps", [1, 2])
@pytest.mark.parametrize("frames,chunk", [(5, 0), (300, N * 40)])
def test_correlate_demod_stream_8_bytes_in_bits_out(ctx, plan, sig, bps, frames, chunk):
    x = rand_c64(7 * bps + frames, frames * N)
    q = modulation.qpsk(ctx) if bps == 2 else modulation.bpsk(ctx)
    want = q.correlate_demod(plan, ctx.vec(x), sig).to_host()
    y, st = pipeline.run(pipeline.Stage.correlate_demod(plan, sig, bps), x, chunk=chunk)
    assert y.dtype == np.uint8 and y.size == bps * x.size and np.array_equal(y, want)


def test_correlate_demod_stream_custom_table(ctx, plan, sig):
    x = rand_c64(77, 20 * N)
    tab = np.array([0.5 + 1j, -1 + 0.25j, 0.75 - 1j, -0.5 - 0.5j], np.complex64)       # not separable
    q = modulation.table(ctx, tab)
    want = q.correlate_demod(plan, ctx.vec(x), sig, compat=False).to_host()
    y, _ = pipeline.run(pipeline.Stage.correlate_demod(plan, sig, 2, table=tab, compat=False), x, chunk=N * 7)
    assert np.array_equal(y, want)


@pytest.mark.parametrize("n,nb,frames,chunk", [(2048, 9, 40, 0), (2048, 9, 40, 2048 * 7), (4096, 3, 9, 4096 * 2), (65536, 9, 6, 65536 * 4)])
def test_fft_interpolate_stream_one_in_ten_out(ctx, n, nb, frames, chunk):
    p = HipFft(ctx, n, max_batch=frames)
    x = rand_c64(n + nb, frames * n)
    out_len = (n + (n - 1) * nb) * frames
    want = ctx.empty(out_len)
    p.rfft_interpolate(ctx.vec(x), want, nb, Scale.SN)
    st = pipeline.Stage.fft_interpolate(p, nb, Scale.SN)
    assert st.out_count(x.size) == out_len
    y, s = pipeline.run(st, x, chunk=chunk)
    assert y.size == out_len and bits_equal(y, want.to_host())


def test_fir_through_the_generic_entry_point(ctx):
    f = Fir(ctx, rand_c64(1, 64, scale=0.2), 2048)
    x = rand_c64(11, 1984 * 70 + 5)
    y, _ = pipeline.run(pipeline.Stage.fir(f), x, chunk=1984 * 16)
    assert bits_equal(y, f.filter(x)) and bits_equal(y, f.filter_stream(x)[0])


@pytest.mark.parametrize("dec,blocks,chunk", [(4, 70, 1984 * 16), (31, 12, 0), (64, 200, 1984 * 33)])
def test_fir_with_decimating_store_as_a_stage(ctx, dec, blocks, chunk):
    """the filter followed by sampling::downsample in the kernel's store (aeth_fir_exec_decim) over a host stream:
    8 B in, 8 / dec B out per sample; the decimation has to divide the hop so that every chunk starts on a kept sample"""
    f = Fir(ctx, rand_c64(1, 64, scale=0.2), 2048)
    n = 1984 * blocks + dec * 5                              # a ragged last chunk that is still a multiple of dec
    x = rand_c64(dec + blocks, n)
    y, st = pipeline.run(pipeline.Stage.fir_decim(f, dec), x, chunk=chunk)
    want = f.filter_decim(ctx.vec(x), dec).to_host()
    assert y.size == n // dec and bits_equal(y, want) and bits_equal(y, f.filter(x)[::dec])
    with pytest.raises(ap.AetherError, match="divide"):
        pipeline.run(pipeline.Stage.fir_decim(f, 3), x[: 1984 * 3])          # 3 does not divide 1984
    with pytest.raises(ap.AetherError, match="Only even decimations"):
        pipeline.run(pipeline.Stage.fir_decim(f, dec), x[: n - 1])


def test_stream_from_and_into_pool_elements(ctx, plan, sig):
    """pinned on both sides: no host stage (stats: pinned == 3); the demodulator's byte output lands in a pool element"""
    frames = 64
    x = rand_c64(3, frames * N)
    p = pool.Pool(ctx, x.nbytes, initial_len=2)
    with p.take() as ein, p.take() as eout:
        xin = ein.array(np.complex64)
        xin[:] = x
        bits = eout.array(np.uint8)[: 2 * x.size]
        _, st = pipeline.run(pipeline.Stage.correlate_demod(plan, sig, 2), xin, out=bits, chunk=N * 10)
        want = modulation.qpsk(ctx).correlate_demod(plan, ctx.vec(x), sig).to_host()
        assert st["pinned"] == 3 and np.array_equal(bits, want)
        del xin, bits, _                                           # `_` is the output array run() returned
    p.close()


def test_stream_argument_errors(ctx, plan, sig):
    lib = plan._lib
    x = rand_c64(1, 3 * N + 1)                                            # not whole frames
    with pytest.raises(ap.AetherError, match="Input and FFT must be the same length"):
        pipeline.run(pipeline.Stage.fft(plan), x, out=np.empty(3 * N + 1, np.complex64))
    x = rand_c64(1, 3 * N)
    with pytest.raises(ap.LengthMismatch):
        pipeline.run(pipeline.Stage.fft(plan), x, out=np.empty(3 * N - 1, np.complex64))
    buf = np.zeros(4 * N, np.complex64)
    with pytest.raises(ap.AetherError, match="overlap"):                   # out starts inside in
        pipeline.run(pipeline.Stage.fft(plan), buf[: 3 * N], out=buf[N:])
    other = ap.Context(0)
    with pytest.raises(ap.AetherError, match="another context"):
        st = pipeline.Stage.fft(plan); st.ctx = other
        pipeline.run(st, x)
    other.close()
    bad = pipeline.Stage.fft(plan); bad.op.kind = 17
    assert bad.out_count(N) == 0
    del lib


def test_a_failed_staging_take_leaves_nothing_checked_out():
    """every error path of the staging take gives its elements back: after a forced failure the next, LARGER run (which
    has to destroy and rebuild the staging pools) still works -- round 3 left them checked out and the pool refused"""
    c = ap.Context(0)
    f = Fir(c, rand_c64(1, 64, scale=0.2), 2048)
    x = rand_c64(2, 1984 * 30)
    want = f.filter(x)
    lib = f._lib
    for nth in (1, 2, 3, 4, 5, 6):                                          # 3 input + 3 output elements per run
        lib.aeth_test_fail_staging_after(nth)
        with pytest.raises(ap.AetherError, match="forced failure"):
            f.filter_stream(x, chunk=1984 * 10)
        y, _ = f.filter_stream(x, chunk=1984 * 10)
        assert bits_equal(y, want)
    big = rand_c64(3, 1984 * 300)
    y, _ = f.filter_stream(big, chunk=1984 * 100)                          # larger elements: the pools are rebuilt
    assert bits_equal(y, f.filter(big))
    lib.aeth_test_fail_staging_after(0)
    c.close()


def test_trim_gives_back_and_the_next_run_rebuilds(ctx, plan):
    free0 = _free_bytes()
    x = rand_c64(4, 512 * N)
    want, _ = pipeline.run(pipeline.Stage.fft(plan, Scale.SN), x)
    held = free0 - _free_bytes()
    ctx.trim()
    assert _free_bytes() >= free0 - (1 << 20) or held == 0                 # the slots went back to the device
    again, _ = pipeline.run(pipeline.Stage.fft(plan, Scale.SN), x)
    assert bits_equal(again, want)
    ctx.trim()


def test_a_huge_chunk_is_split_so_that_a_slot_stays_bounded(ctx, plan):
    """chunk_samples beyond 64 MiB per slot is split internally: what the context retains stays bounded"""
    frames = (80 << 20) // (N * 8)                                          # an 80 MiB stream asked for as ONE chunk
    x = rand_c64(9, frames * N)
    y, st = pipeline.run(pipeline.Stage.fft(plan, Scale.NONE), x, chunk=x.size)
    assert st["chunks"] == 2
    want = ctx.vec(x); plan.ifwd(want, Scale.NONE)
    assert bits_equal(y, want.to_host())
    ctx.trim()


def _free_bytes():
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


# ---- several ops as one compute stage: pipeline::new(..).add_stage(a).add_stage(b) (pipeline.rs:24-41) -----------------
def test_chain_fir_then_fft_then_correlate_demod(ctx, plan, sig):
    """8 B up, 2 B down per sample; the intermediates never leave the device; bits of the three device calls in a row"""
    f = Fir(ctx, rand_c64(1, 64, scale=0.2), 2048)
    granule = 1984 * 2048 // np.gcd(1984, 2048)              # whole hops AND whole frames: lcm(1984, 2048) = 63488
    n = granule * 9
    x = rand_c64(5, n)
    stages = [pipeline.Stage.fir(f), pipeline.Stage.fft(plan, Scale.SN), pipeline.Stage.correlate_demod(plan, sig, 2)]
    d = f.filter(ctx.vec(x)); plan.ifwd(d, Scale.SN)
    want = modulation.qpsk(ctx).correlate_demod(plan, d, sig).to_host()
    for chunk in (0, granule, granule * 4):
        y, st = pipeline.run_chain(stages, x, chunk=chunk)
        assert y.dtype == np.uint8 and np.array_equal(y, want), chunk
    y, st = pipeline.run_chain(stages, x, report=True)
    assert np.array_equal(y, want) and len(st["lines"]) == 5
    with pytest.raises(ap.AetherError, match="granules"):
        pipeline.run_chain(stages, x[: granule + 2048])       # whole frames, but not whole hops of the filter's chunks


@pytest.mark.parametrize("frames,chunk", [(7, 0), (300, N * 32)])
def test_chain_with_in_place_stages(ctx, plan, sig, frames, chunk):
    """the correlator chain runs in place: first (on the slot), last (in scratch, copied to the download buffer), alone"""
    x = rand_c64(frames, frames * N)
    # in place first, out of place last: correlator chain -> FFT + interpolate
    o = ctx.empty((N + (N - 1) * 2) * frames); d = ctx.vec(x); plan.mul_chain(d, sig); plan.rfft_interpolate(d, o, 2, Scale.SN)
    y, _ = pipeline.run_chain([pipeline.Stage.mul_chain(plan, sig), pipeline.Stage.fft_interpolate(plan, 2, Scale.SN)], x, chunk=chunk)
    assert bits_equal(y, o.to_host())
    # out of place first, in place last: FFT frames -> correlator chain
    d = ctx.vec(x); plan.ifwd(d, Scale.SN); plan.mul_chain(d, sig, Scale.NONE, Scale.N)
    y, _ = pipeline.run_chain([pipeline.Stage.fft(plan, Scale.SN), pipeline.Stage.mul_chain(plan, sig, Scale.NONE, Scale.N)], x, chunk=chunk)
    assert bits_equal(y, d.to_host())
    # every stage in place
    d = ctx.vec(x); plan.mul_chain(d, sig); plan.mul_chain(d, sig, Scale.SN, Scale.SN)
    y, _ = pipeline.run_chain([pipeline.Stage.mul_chain(plan, sig), pipeline.Stage.mul_chain(plan, sig, Scale.SN, Scale.SN)], x, chunk=chunk)
    assert bits_equal(y, d.to_host())
    # a chain of one is the plain call
    y, _ = pipeline.run_chain([pipeline.Stage.fft(plan, Scale.SN)], x, chunk=chunk)
    assert bits_equal(y, pipeline.run(pipeline.Stage.fft(plan, Scale.SN), x)[0])


def test_chain_rules(ctx, plan, sig):
    f = Fir(ctx, rand_c64(1, 64, scale=0.2), 2048)
    x = rand_c64(1, 63488)
    with pytest.raises(ap.AetherError, match="first stage"):
        pipeline.run_chain([pipeline.Stage.fft(plan), pipeline.Stage.fir(f)], x)
    with pytest.raises(ap.AetherError, match="last stage"):
        pipeline.run_chain([pipeline.Stage.correlate_demod(plan, sig, 2), pipeline.Stage.fft(plan)], x)
    small = HipFft(ctx, 512)
    y, _ = pipeline.run_chain([pipeline.Stage.fft(plan, Scale.SN), pipeline.Stage.fft(small, Scale.SN)], x[: 2048 * 6])   # 2048-frames, then 512-frames
    d = ctx.vec(x[: 2048 * 6]); plan.ifwd(d, Scale.SN); small.ifwd(d, Scale.SN)
    assert bits_equal(y, d.to_host())


@pytest.mark.parametrize("bps", [1, 2])
def test_modem_as_a_host_pipeline(ctx, plan, sig, bps):
    """BASELINE config 4's channel end to end over host memory (examples/modem.rs:15-32 behind the correlator): bit bytes up,
    modulate + AWGN -> correlate -> demodulate on the device, bit bytes down.  The noise is addressed by stream position,
    so every chunking draws what the one-shot device calls draw."""
    from aether_primitives_amd import noise
    frames = 90
    bits = np.random.default_rng(3).integers(0, 2, bps * N * frames, dtype=np.uint8)
    m = modulation.qpsk(ctx) if bps == 2 else modulation.bpsk(ctx)
    tx = m.modulate_awgn(bits, noise.new(ctx, 0.01, 4711))
    want_tx = tx.to_host()
    want = m.correlate_demod(plan, tx, sig).to_host()
    mod = pipeline.Stage.modulate_awgn(ctx, bps, 0.01, 4711)
    for chunk in (0, bps * N * 7, bps * N * 64):
        y, _ = pipeline.run(mod, bits, chunk=chunk)                      # the source stage alone: bits in, symbols out
        assert y.dtype == np.complex64 and bits_equal(y, want_tx), chunk
        rx, _ = pipeline.run_chain([mod, pipeline.Stage.correlate_demod(plan, sig, bps)], bits, chunk=chunk)
        assert rx.dtype == np.uint8 and rx.size == bits.size and np.array_equal(rx, want), chunk
    # a stream that starts in the middle of the noise stream: offset = symbols already drawn (odd offsets too)
    off = 12345
    tail = m.modulate_awgn(bits, _Awgn(0.01, 4711, off, ctx)).to_host()
    y, _ = pipeline.run(pipeline.Stage.modulate_awgn(ctx, bps, 0.01, 4711, offset=off), bits, chunk=bps * N * 5)
    assert bits_equal(y, tail)
    with pytest.raises(ap.AetherError, match="first stage"):
        pipeline.run_chain([pipeline.Stage.fft(plan), mod], rand_c64(1, N))


def _Awgn(power, seed, offset, ctx):
    from aether_primitives_amd import noise
    a = noise.new(ctx, power, seed); a.offset = offset
    return a
