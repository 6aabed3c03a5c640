"""CPU, 2 processes over gloo: the N>1 path of the bench -- contiguous shards, no
data-path collective, barrier + MAX-over-ranks timing -- and the property that makes
it safe: hop-aligned FIR shards reproduce the single-process output BIT FOR BIT."""
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from aether_primitives_amd.sharding import channel_of, fir_shard, frame_shard  # noqa: E402


def test_frame_shard_covers_everything():
    for n, w in [(512, 8), (513, 8), (7, 8), (0, 2), (8456, 3)]:
        spans = [frame_shard(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (a, ca), (b, _) in zip(spans, spans[1:]):
            assert a + ca == b
    assert channel_of(1, 8, 8) == [1] and channel_of(0, 2, 8) == [0, 2, 4, 6]


def test_fir_shard_alignment():
    n, hop, m = (1 << 24), 1984, 64
    prev = 0
    for r in range(8):
        s = fir_shard(n, hop, m, r, 8)
        assert s["out_lo"] == prev and s["out_lo"] % hop == 0
        assert s["hist_lo"] == max(s["out_lo"] - 63, 0)
        prev = s["out_hi"]
    assert prev == n


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from oracle import pyoracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    n, hop = 200000, 1984
    taps = orc.synth_lowpass_taps(64, 0.25)
    x = orc.synth_cnormal(815, n)                      # every rank can regenerate the source stream
    s = fir_shard(n, hop, 64, rank, world)
    hist = x[s["hist_lo"]:s["in_lo"]] if s["in_lo"] > 0 else None
    dist.barrier()
    t0 = time.perf_counter()
    y = orc.fir_ols_f32(taps, x[s["in_lo"]:s["out_hi"]], 2048, hop, hist=hist)   # stand-in for the kernel
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)          # the bench's timing rule: MAX over ranks
    # gather only to CHECK the result (the data path itself needs no collective)
    parts = [None] * world
    dist.all_gather_object(parts, (s["out_lo"], y))
    if rank == 0:
        full = np.concatenate([p for _, p in sorted(parts, key=lambda t: t[0])])
        ref = orc.fir_ols_f32(taps, x, 2048, hop)
        q.put((bool((full.view(np.uint32) == ref.view(np.uint32)).all()), float(el.item()), full.size))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_fir_shards_bit_identical():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(150)
    assert all(p.exitcode == 0 for p in procs)
    same, max_time, n = q.get(timeout=5)
    assert same and n == 200000 and max_time > 0
