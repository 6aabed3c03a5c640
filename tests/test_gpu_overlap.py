"""GPU: the context's overlap lane (aeth_ctx_set_overlap, include/aether_hip.h) and the multi-rank path through the
real kernel.  Consecutive independent FIR launches alternate between two HIP queues; whatever is enqueued around them
must see the results of one in-order stream."""
import os
import sys

import numpy as np
import pytest

import aether_primitives_amd as ap
from aether_primitives_amd import Fir
from helpers import bits_equal, rand_c64

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def octx():
    c = ap.Context(0)
    c.set_overlap(True)
    assert c.overlap
    yield c
    c.close()


def test_overlap_refused_on_borrowed_stream(ctx):
    b = ap.Context(0, stream=ctx.stream)
    with pytest.raises(ap.AetherError):
        b.set_overlap(True)
    assert not b.overlap
    b.close()


def test_handing_the_stream_out_parks_the_lane(octx, oracle):
    """aeth_ctx_stream gives the main stream to code the library does not see (here: a second context that borrows it,
    standing in for torch ops on an ExternalStream).  From then on every launch stays on that stream, so foreign work
    enqueued on it is ordered behind later launches too; aeth_ctx_set_overlap(1) re-arms the lane."""
    taps = oracle.synth_lowpass_taps(64, 0.25)
    n = 1 << 20
    f = Fir(octx, taps, 2048)
    xs = [rand_c64(60 + i, n) for i in range(4)]
    ins = [octx.vec(x) for x in xs]
    outs = [octx.empty(n) for _ in xs]
    want = [oracle.fir_ols_f32(taps, x, 2048, f.hop) for x in xs]
    foreign = ap.Context(0, stream=octx.stream)                # the hand-over
    lib = f._lib
    for rep in range(3):
        for i in range(4):
            f.filter(ins[i], out=outs[i])                      # would alternate lanes if the lane were still in use
        # foreign work on the borrowed stream, no sync, no library call on octx in between: reads what launches 2 and 3 wrote
        from aether_primitives_amd._lib import check
        check(lib.aeth_vec_add(foreign.h, outs[3]._p(), n, outs[2]._p(), n))
        foreign.sync()
        got = np.empty(n, np.complex64)
        foreign.download(outs[3].ptr, got)
        assert oracle.evm_db(got, oracle.vec_add(want[3], want[2])) <= -120
    foreign.close()
    octx.set_overlap(True)                                     # re-armed: chained launches use both queues again
    for i in range(4):
        f.filter(ins[i], out=outs[i])
    for i in range(4):
        assert oracle.evm_db(outs[i].to_host(), want[i]) <= -120


def test_independent_launches_match_the_in_order_run(ctx, octx, oracle):
    taps = oracle.synth_lowpass_taps(64, 0.25)
    n = 1 << 20
    xs = [rand_c64(100 + i, n) for i in range(5)]
    f_ref, f_ov = Fir(ctx, taps, 2048), Fir(octx, taps, 2048)
    ref = [f_ref.filter(ctx.vec(x)).to_host() for x in xs]
    ins = [octx.vec(x) for x in xs]
    outs = [octx.empty(n) for _ in xs]
    for rep in range(3):                                   # back to back, no sync in between: lanes alternate
        for i in range(5):
            f_ov.filter(ins[i], out=outs[i])
    for i in range(5):
        assert bits_equal(outs[i].to_host(), ref[i])


def test_dependent_calls_stay_ordered(octx, oracle):
    """Launch k+1 reads what launch k wrote (and later overwrites what k read): the lane must serialise them."""
    taps = oracle.synth_lowpass_taps(64, 0.25)
    n = 1 << 19
    x = rand_c64(7, n)
    f = Fir(octx, taps, 2048)
    a, b, c = octx.vec(x), octx.empty(n), octx.empty(n)
    f.filter(a, out=b)          # b = h * x
    f.filter(b, out=c)          # reads b: conflict with the predecessor's output
    f.filter(c, out=a)          # writes a, the first launch's input, and reads c
    y1 = oracle.fir_ols_f32(taps, x, 2048, f.hop)
    y2 = oracle.fir_ols_f32(taps, y1, 2048, f.hop)
    y3 = oracle.fir_ols_f32(taps, y2, 2048, f.hop)
    assert oracle.evm_db(b.to_host(), y1) <= -120 and oracle.evm_db(c.to_host(), y2) <= -120
    assert oracle.evm_db(a.to_host(), y3) <= -120
    # same chain on a plain context: identical bits
    p = ap.Context(0)
    fp = Fir(p, taps, 2048)
    pa, pb, pc = p.vec(x), p.empty(n), p.empty(n)
    fp.filter(pa, out=pb); fp.filter(pb, out=pc); fp.filter(pc, out=pa)
    assert bits_equal(pa.to_host(), a.to_host())
    p.close()


def test_other_ops_join_the_lane(octx, oracle):
    """An element-wise op right behind a chained launch reads that launch's output: it must wait for the aux queue."""
    taps = oracle.synth_lowpass_taps(64, 0.25)
    n = 1 << 20
    f = Fir(octx, taps, 2048)
    xs = [rand_c64(40 + i, n) for i in range(4)]
    ins = [octx.vec(x) for x in xs]
    outs = [octx.empty(n) for _ in xs]
    for rep in range(4):
        for i in range(4):
            f.filter(ins[i], out=outs[i])
        # outs[3] was produced on one lane, outs[2] on the other: combine both right away
        outs[3].vec_add(outs[2])
        got = outs[3].to_host()
        exp = oracle.vec_add(oracle.fir_ols_f32(taps, xs[3], 2048, f.hop), oracle.fir_ols_f32(taps, xs[2], 2048, f.hop))
        assert oracle.evm_db(got, exp) <= -120
    # events are recorded behind both lanes: the elapsed time covers every launch in between
    e0, e1 = octx.event(), octx.event()
    e0.record()
    for i in range(8):
        f.filter(ins[i % 4], out=outs[i % 4])
    e1.record(); octx.sync()
    assert e0.elapsed_ms(e1) > 8 * (n / 500e9) * 1e3 * 0.5      # at least half of what 8 launches need at 500 GS/s


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_call_sequences_match_a_plain_context(seed, oracle):
    """Random programs over a pool of buffers -- filters between random pairs (sometimes chained, sometimes clobbering
    an earlier input or output, sometimes on overlapping slices), element-wise ops, downloads, syncs -- give the same
    bits on a context with the overlap lane as on a plain one."""
    rng = np.random.default_rng(seed)
    n, nbuf = 1 << 18, 6
    taps = oracle.synth_lowpass_taps(64, 0.25)
    init = [rand_c64(1000 * seed + i, n) for i in range(nbuf)]
    prog = []
    for _ in range(120):
        k = rng.integers(10)
        if k < 6:
            i, j = rng.choice(nbuf, 2, replace=False)
            if rng.integers(4) == 0:                     # overlapping views of one buffer as input of one call, output of the next
                lo = int(rng.integers(0, n // 2)); prog.append(("fir_slice", int(i), int(j), lo, lo + n // 2))
            else:
                prog.append(("fir", int(i), int(j)))
        elif k < 8:
            i, j = rng.choice(nbuf, 2, replace=False); prog.append(("add", int(i), int(j)))
        elif k == 8:
            prog.append(("scale", int(rng.integers(nbuf)), float(rng.uniform(0.5, 1.5))))
        else:
            prog.append(("peek", int(rng.integers(nbuf))))

    def run(c):
        f = Fir(c, taps, 2048)
        b = [c.vec(x) for x in init]
        peeks = []
        for op in prog:
            if op[0] == "fir": f.filter(b[op[1]], out=b[op[2]])
            elif op[0] == "fir_slice": f.filter(b[op[1]].slice(op[3], op[4]), out=b[op[2]].slice(op[3], op[4]))
            elif op[0] == "add": b[op[1]].vec_add(b[op[2]])
            elif op[0] == "scale": b[op[1]].vec_scale(op[2])
            else: peeks.append(b[op[1]].slice(0, 64).to_host())
        return [v.to_host() for v in b], peeks

    plain = ap.Context(0)
    lane = ap.Context(0); lane.set_overlap(True)
    want, wp = run(plain)
    got, gp = run(lane)
    for a, b in zip(want, got):
        assert bits_equal(a, b)
    for a, b in zip(wp, gp):
        assert bits_equal(a, b)
    plain.close(); lane.close()


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from oracle import pyoracle as orc
    from aether_primitives_amd.sharding import fir_shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)      # control plane only; both ranks share GPU 0
    c = ap.Context(0)
    c.set_overlap(True)
    n = 3_000_000
    taps = orc.synth_lowpass_taps(64, 0.25)
    x = orc.synth_cnormal(815, n)
    f = Fir(c, taps, 2048)
    s = fir_shard(n, f.hop, 64, rank, world)
    hist = c.vec(x[s["hist_lo"]:s["in_lo"]]) if s["in_lo"] > 0 else None
    dist.barrier()
    y = f.filter(c.vec(x[s["in_lo"]:s["out_hi"]]), hist=hist).to_host()       # the HIP kernel on this rank's shard
    parts = [None] * world
    dist.all_gather_object(parts, (s["out_lo"], y))
    if rank == 0:
        full = np.concatenate([p for _, p in sorted(parts, key=lambda t: t[0])])
        whole = f.filter(c.vec(x)).to_host()                                   # one launch over the whole stream
        q.put((bool((full.view(np.uint32) == whole.view(np.uint32)).all()), full.size,
               float(orc.evm_db(full, orc.fir_ols_f32(taps, x, 2048, f.hop)))))
    dist.barrier()
    c.close()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_shard_one_stream_through_the_kernel(ctx):
    """N > 1 path with the real kernel: two processes (gloo for the check only), each filters its hop-aligned shard
    on the GPU; the gathered output equals the single-launch output bit for bit."""
    import multiprocessing as mp      # not torch.multiprocessing: importing torch here would bring a second HIP runtime into the test process
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 29600 + (os.getpid() % 2000)
    procs = [mpc.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(280)
    assert all(p.exitcode == 0 for p in procs)
    same, n, evm = q.get(timeout=5)
    assert same and n == 3_000_000 and evm <= -120
