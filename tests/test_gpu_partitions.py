"""GPU: the node-level partitions of BASELINE configs 4 and 5 through the real kernels with two ranks.

Two processes on the box's GPU (gloo as the control plane only, as in tests/test_gpu_overlap.py): C4's channels are
dealt out with `channel_of`, C5's frames with `frame_shard`; each rank runs the same calls bench.py --workload c4 / c5
makes on its share, and what rank 0 gathers equals a one-process run of the whole job bit for bit (C4: the decided
bits; C5: the interpolated samples).  No collective touches the data path."""
import os

import numpy as np
import pytest

import aether_primitives_amd as ap
from helpers import rand_c64

pytestmark = pytest.mark.gpu

N = 2048
C4_CHANNELS, C4_FRAMES = 8, 24
C5_FRAMES, C5_LEN, C5_NB = 6, 65536, 9


def c4_channel(ctx, c, q, f, sig):
    """channel c of C4: bits (seed 815 + c) -> QPSK + AWGN (stream 815 + c) -> FFT-2048 correlate -> hard demod"""
    from aether_primitives_amd import modulation, noise
    n = N * C4_FRAMES
    bits = np.random.default_rng(815 + c).integers(0, 2, 2 * n, dtype=np.uint8)
    tx = q.modulate_awgn(modulation.DeviceBits(ctx, 2 * n, bits), noise.new(ctx, 0.01, 815 + c))
    return q.correlate_demod(f, tx, sig).to_host()


def c5_frames(ctx, f, lo, cnt):
    """frames [lo, lo + cnt) of C5: 65536-point forward FFT (Scale::SN) + 10x linear interpolation, one call"""
    from aether_primitives_amd import Scale
    x = rand_c64(815, C5_FRAMES * C5_LEN)[lo * C5_LEN:(lo + cnt) * C5_LEN]
    out = ctx.empty((C5_LEN + (C5_LEN - 1) * C5_NB) * cnt)
    f.rfft_interpolate(ctx.vec(x), out, C5_NB, Scale.SN)
    return out.to_host()


def _setup(ctx):
    from aether_primitives_amd import modulation
    ref = np.zeros(N, np.complex64)
    ref[:4] = np.conj(np.array([-1 + 1j, 0, 1 - 1j, 1 - 1j], np.complex64))
    return modulation.qpsk(ctx), ap.HipFft(ctx, N, max_batch=C4_FRAMES), ctx.vec(ref)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from aether_primitives_amd.sharding import channel_of, frame_shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)      # control plane only; both ranks share GPU 0
    ctx = ap.Context(0)
    mod, f2048, sig = _setup(ctx)
    dist.barrier()
    mine = {c: c4_channel(ctx, c, mod, f2048, sig) for c in channel_of(rank, world, C4_CHANNELS)}
    lo, cnt = frame_shard(C5_FRAMES, rank, world)
    f64k = ap.HipFft(ctx, C5_LEN, max_batch=max(cnt, 1))
    shard = c5_frames(ctx, f64k, lo, cnt)
    parts = [None] * world
    dist.all_gather_object(parts, (mine, lo, shard))                  # the check, not the product: nothing here feeds a kernel
    if rank == 0:
        chans = {}
        for m, _, _ in parts:
            chans.update(m)
        c4_ok = sorted(chans) == list(range(C4_CHANNELS)) and all(
            np.array_equal(chans[c], c4_channel(ctx, c, mod, f2048, sig)) for c in range(C4_CHANNELS))
        whole = c5_frames(ctx, ap.HipFft(ctx, C5_LEN, max_batch=C5_FRAMES), 0, C5_FRAMES)
        got = np.concatenate([s for _, _, s in sorted(parts, key=lambda t: t[1])])
        c5_ok = got.size == whole.size and bool((got.view(np.uint32) == whole.view(np.uint32)).all())
        q.put((c4_ok, c5_ok, [sorted(m) for m, _, _ in parts], [(l, s.size) for _, l, s in parts]))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_run_the_c4_and_c5_partitions_through_the_kernels(ctx):
    import multiprocessing as mp      # not torch.multiprocessing: importing torch here would bring a second HIP runtime into the test process
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 31700 + (os.getpid() % 2000)
    procs = [mpc.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(280)
    assert all(p.exitcode == 0 for p in procs)
    c4_ok, c5_ok, chan_split, frame_split = q.get(timeout=5)
    assert chan_split == [[0, 2, 4, 6], [1, 3, 5, 7]]                 # channel c -> rank c % 2
    per = C5_LEN + (C5_LEN - 1) * C5_NB
    assert frame_split == [(0, 3 * per), (3, 3 * per)]                # contiguous frame ranges
    assert c4_ok, "C4: gathered bits differ from the one-process run"
    assert c5_ok, "C5: gathered samples differ from the one-process run"


def test_c4_channels_are_distinct_and_decodable(ctx):
    """sanity of the job itself: channels carry different bits, and at noise power 0.01 the correlate -> demod chain is
    deterministic for a given channel"""
    mod, f, sig = _setup(ctx)
    a, b = c4_channel(ctx, 0, mod, f, sig), c4_channel(ctx, 1, mod, f, sig)
    assert a.shape == b.shape == (2 * N * C4_FRAMES,) and not np.array_equal(a, b)
    assert np.array_equal(a, c4_channel(ctx, 0, mod, f, sig))
