#!/usr/bin/env python3
"""BASELINE config 5's transform (65536-point FFT, Scale::SN) under the lab shapes of aeth_fft_big.hip, interleaved
A/B in one process, 512 frames (one GPU's job) and 64 frames (an 8-GPU shard):
   default            two launches over the whole batch (step A, step B)
   groups g           the same, group by group through a g-MiB work buffer (AETH_4S_GROUP_MIB)
   parts              frame halves on two HIP streams (AETH_4S_PARTS=2): one half's step B beside the other's step A
                      -- round 3's shape; its code left the library in round 4 (numbers: profiles/r03_c5.json), the knob is a no-op now
   parts + groups g   g-MiB groups alternating between the two streams
Every shape's output is compared bit for bit with the default's.  One shape only (for rocprofv3 --pmc runs):
   AETH_TUNING=1 python3 tools/c5_shapes.py --only "parts+groups 64" --batch 512 --launches 20"""
import argparse, os, sys, time
os.environ.setdefault("AETH_TUNING", "1")
os.environ.setdefault('AETH_LAB_LIB', '1')   # these knobs exist only in the lab build: make -C aether_primitives_amd/csrc LAB=1
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from aether_primitives_amd import Scale
from bench import synth_stream

ap_ = argparse.ArgumentParser()
ap_.add_argument("--only", default=None)
ap_.add_argument("--batch", type=int, default=0)
ap_.add_argument("--launches", type=int, default=30)
ap_.add_argument("--rounds", type=int, default=5)
args = ap_.parse_args()

SHAPES = [("default", 1, 0), ("groups 128", 1, 128), ("groups 64", 1, 64), ("parts", 2, 0), ("parts+groups 128", 2, 128),
          ("parts+groups 64", 2, 64), ("parts+groups 32", 2, 32), ("parts+groups 16", 2, 16)]
if args.only:
    SHAPES = [s for s in SHAPES if s[0] == args.only]
N = 65536
ctx = ap.Context(0)


def set_shape(parts, gmib):
    os.environ["AETH_4S_PARTS"] = str(parts)
    os.environ["AETH_4S_GROUP_MIB"] = str(gmib)


for batch in ([args.batch] if args.batch else [512, 64]):
    n = N * batch
    f = ap.HipFft(ctx, N, max_batch=batch)
    nbuf = 3 if batch > 128 else 8                      # rotate past the 256 MiB cache
    ins = [ctx.vec(synth_stream(815 + i, n)) for i in range(nbuf)]
    outs = [ctx.empty(n) for _ in range(nbuf)]
    set_shape(1, 0)
    f.fwd(ins[0], outs[0], Scale.SN); ref = outs[0].to_host()
    res = {s[0]: [] for s in SHAPES}
    for name, parts, gmib in SHAPES:
        set_shape(parts, gmib)
        if gmib and gmib * (1 << 20) >= n * 8 and parts == 1:
            continue
        f.fwd(ins[0], outs[1], Scale.SN)
        same = np.array_equal(outs[1].to_host().view(np.uint32), ref.view(np.uint32))
        print(f"batch {batch:4d}  {name:18s} output {'bit-identical' if same else 'DIFFERS'}", flush=True)
    for r in range(args.rounds):
        for name, parts, gmib in SHAPES:
            set_shape(parts, gmib)
            for i in range(5): f.fwd(ins[i % nbuf], outs[i % nbuf], Scale.SN)
            ctx.sync()
            e0, e1 = ctx.event(), ctx.event()
            e0.record()
            for i in range(args.launches): f.fwd(ins[i % nbuf], outs[i % nbuf], Scale.SN)
            e1.record(); ctx.sync()
            res[name].append(e0.elapsed_ms(e1) / args.launches * 1e3)
    print(f"== {batch} frames x 65536 points ({n * 8 >> 20} MiB in, {n * 8 >> 20} MiB out), {args.rounds} interleaved rounds x {args.launches} transforms ==")
    for name, _, _ in SHAPES:
        v = np.array(res[name])
        if v.size:
            med = float(np.median(v))
            print(f"   {name:18s} median {med:8.1f} us  min {v.min():8.1f} us   {16 * n / med / 1e6:6.2f} TB/s of 16 B/sample = {16 * n / med / 1e6 / 8 * 100:5.1f} % of 8 TB/s")
    del ins, outs, f
