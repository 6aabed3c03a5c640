#!/usr/bin/env python3
"""bench.py -- headline benchmark of the cf32 hot path on MI355X.

Metric (BASELINE.json): GSamples/s of cf32 through the FFT-2048 + 64-tap FIR
chain, and % of HBM roofline on one GPU.  Workload = BASELINE config 3: a 64-tap
FIR by overlap-save (FFT-2048) over a 16 Mi-sample cf32 stream; one "step" =
one pass over one stream = ONE launch of the fused kernel.  Streams rotate
through a working set larger than the 256 MiB Infinity Cache so every step
reads its input from HBM.  Frames are independent, so N GPUs run N
independent streams (weak scaling, no collective on the data path).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FFT_LEN, NTAPS = 2048, 64
STREAM = 1 << 24               # 16 Mi samples (BASELINE config 3)
BYTES_PER_SAMPLE = 16          # algorithmic: 8 B read + 8 B written per output sample


def synth_stream(seed, n):
    """complex normal, unit power, deterministic (seed 815 = noise.rs:6)."""
    rng = np.random.default_rng(seed)
    out = np.empty(n, np.complex64)
    v = out.view(np.float32)
    chunk = 1 << 22
    for i in range(0, 2 * n, chunk):
        v[i:i + chunk] = rng.standard_normal(min(chunk, 2 * n - i), dtype=np.float32) * np.float32(0.70710678)
    return out


def lowpass_taps(ntaps=NTAPS, cutoff=0.25):
    k = np.arange(ntaps, dtype=np.float64)
    m = k - (ntaps - 1) / 2.0
    sinc = np.where(m == 0, 2 * cutoff, np.sin(2 * np.pi * cutoff * m) / (np.pi * np.where(m == 0, 1, m)))
    w = 0.54 - 0.46 * np.cos(2 * np.pi * k / (ntaps - 1))
    t = sinc * w
    return (t / t.sum()).astype(np.complex64)


def cpu_baseline(budget_s=12.0):
    """Oracle (CPU restatement of the reference chain) on a bounded sample of the same workload."""
    from oracle import pyoracle as orc
    n = 1 << 22
    x = orc.synth_cnormal(815, n)
    taps = orc.synth_lowpass_taps(NTAPS, 0.25)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16)      # the GPU box's CPU share for one GPU
    res = {}
    for label, th in (("1", 1), ("T", cores)):
        done, t0 = 0, time.perf_counter()
        while True:
            orc.fir_ols_f32(taps, x, FFT_LEN, FFT_LEN - NTAPS + 1 - ((FFT_LEN - NTAPS + 1) % 64), threads=th)
            done += n
            el = time.perf_counter() - t0
            if el >= budget_s / 2:
                break
        res[label] = done / el / 1e9
        res[label + "_n"] = done
    return {"value": round(res["T"], 5), "unit": "GSamples/s", "cores": cores, "kind": "port",
            "value_1thread": round(res["1"], 5),
            "sample": f"oracle/aeth_oracle.c overlap-save chain (rfft->vec_mul->rifft, f32) over "
                      f"{res['T_n'] >> 20} Mi samples on {cores} threads / {res['1_n'] >> 20} Mi on 1 thread, "
                      f"same taps and FFT-2048 geometry as the GPU run"}


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC
    summary (profiles/rNN_summary.json: separate FETCH_SIZE / WRITE_SIZE passes over this
    same command, FETCH_SIZE doubled per MI355X_MICROARCH.md).  None if no profile is there."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json"))):
        try:
            d = json.load(open(f))
            if "hbm_bytes_per_launch" in d:
                best = (os.path.basename(f), d["hbm_bytes_per_launch"])
        except Exception:
            pass
    return best


class Ranks:
    """One process per GPU (torch.distributed.run sets RANK/LOCAL_RANK/WORLD_SIZE); RCCL is used for the
    barrier and the MAX-over-ranks of the elapsed time only -- the data path has no exchange step."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            # the process count is the truth (for N > 1 launch through
            # `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`)
            if self.rank == 0 and args.gpus != 1:
                print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}; reporting n_gpus={self.world}", file=sys.stderr)
            args.gpus = self.world
        import torch
        self.torch = torch
        self.dist = None
        if self.world > 1 or "RANK" in os.environ:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.cuda.set_device(self.local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank))
            self.dist = dist

    def barrier(self, ctx):
        if self.dist is not None:
            t = self.torch.zeros(1, device="cuda")
            self.dist.all_reduce(t)
        self.torch.cuda.synchronize()
        ctx.sync()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = self.torch.tensor([seconds], device="cuda", dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def settle(step, ctx, ms):
    """Untimed launches of `step` for `ms` milliseconds: carries the device through its load-onset power
    transient (tools/transient.py: after >= 5 ms of idle ~2 ms at full speed, then 10-25 % slower for
    ~25 ms, then steady).  Returns the number of launches."""
    n, t0 = 0, time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for i in range(50):
            step(n + i)
        n += 50
        ctx.sync()
    return n


def side_workload(args):
    """BASELINE configs 2, 4 and 5 (device-resident, synthetic); same JSON shape, no roofline claim.
    Under torch.distributed.run every rank runs its own channel (C4) / frame shard (C2, C5) on its own GPU."""
    ranks = Ranks(args)
    import aether_primitives_amd as ap
    from aether_primitives_amd import Scale, sampling, modulation, noise
    ctx = ap.Context(ranks.local_rank)
    N = 2048
    if args.workload == "c2":
        n = 1 << 20                                             # 512 frames: 8 MiB, cache-resident by definition
        f = ap.HipFft(ctx, N, max_batch=n // N)
        bufs = [(ctx.vec(synth_stream(815 + i, n)), ctx.empty(n)) for i in range(4)]
        def step(i):
            a, b = bufs[i % 4]; f.fwd(a, b, Scale.SN); f.ifwd(a, Scale.SN)      # benches.rs:305-306,352-353
        samples, name, bytes_ = 2 * n, "C2: FFT-2048 fwd (copy) + ifwd (in place) on a 1 Mi-sample stream", 32 * n
    elif args.workload == "c5":
        frames, nb = 64, 9
        n = 65536 * frames
        f = ap.HipFft(ctx, 65536, max_batch=frames)
        bufs = [(ctx.vec(synth_stream(815 + i, n)), ctx.empty((65536 + 65535 * nb) * frames)) for i in range(3)]
        def step(i):
            a, o = bufs[i % 3]; f.ifwd(a, Scale.SN); sampling.interpolate(ctx, a, o, nb, frame_len=65536)
        samples, name, bytes_ = n, "C5: 64 x 65536-point FFT (Scale::SN) + 10x linear interpolation", 104 * n
    else:
        frames = 4096
        n = N * frames
        rng = np.random.default_rng(815 + ranks.rank)       # one independent channel per rank
        q = modulation.qpsk(ctx)
        f = ap.HipFft(ctx, N, max_batch=frames)
        ref = np.zeros(N, np.complex64); ref[:4] = np.conj(np.array([-1 + 1j, 0, 1 - 1j, 1 - 1j], np.complex64))
        sig = ctx.vec(ref)
        bits = [modulation.DeviceBits(ctx, 2 * n, rng.integers(0, 2, 2 * n, dtype=np.uint8)) for _ in range(2)]
        awgn = noise.new(ctx, 0.01, 815 + ranks.rank)
        txs = [ctx.empty(n) for _ in range(2)]
        rxb = [modulation.DeviceBits(ctx, 2 * n) for _ in range(2)]
        def step(i):
            tx = q.modulate(bits[i % 2], out=txs[i % 2]); awgn.apply(tx); f.mul_chain(tx, sig)
            q.demod_naive(tx, out=rxb[i % 2])
        samples, name, bytes_ = n, "C4 (one channel per GPU): QPSK mod -> AWGN -> FFT-2048 correlate -> hard demod, 4096 frames", 52 * n
    ranks.barrier(ctx)
    nsettle = settle(step, ctx, args.settle_ms)
    for i in range(args.warmup): step(i)
    ranks.barrier(ctx); t0 = time.perf_counter()
    for i in range(args.steps): step(args.warmup + i)
    ctx.sync(); el = ranks.max_over_ranks(time.perf_counter() - t0)
    if ranks.rank == 0:
        print(json.dumps({"metric": "GSamples/s cf32", "value": round(samples * args.steps * args.gpus / el / 1e9, 3),
                          "unit": "GSamples/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                          "settle_launches": nsettle, "ms_per_step": round(el / args.steps * 1e3, 5),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": name},
                          "algorithmic_GBps_per_gpu": round(bytes_ * args.steps / el / 1e9, 1)}), flush=True)
    ranks.close()


def main():
    ap_ = argparse.ArgumentParser()
    ap_.add_argument("--gpus", type=int, default=1)
    ap_.add_argument("--steps", type=int, default=2000)
    ap_.add_argument("--warmup", type=int, default=200)
    ap_.add_argument("--settle-ms", type=float, default=50.0,
                     help="untimed launches of the same step before the warm-up so that the device has left its "
                          "load-onset power transient (tools/transient.py: ~25 ms after any idle gap >= 5 ms)")
    ap_.add_argument("--streams", type=int, default=6, help="rotating working set, x 256 MiB (in+out) each")
    ap_.add_argument("--no-cpu-baseline", action="store_true")
    ap_.add_argument("--no-two-queues", action="store_true",
                     help="skip the extra leg that issues the same launches alternately on two HIP queues (informational)")
    ap_.add_argument("--workload", default="c3", choices=["c3", "c2", "c4", "c5"],
                     help="c3 (default) = the headline config; the others are BASELINE configs 2, 4, 5 for the record")
    args = ap_.parse_args()
    if args.workload != "c3":
        return side_workload(args)

    ranks = Ranks(args)
    rank, local_rank = ranks.rank, ranks.local_rank

    import aether_primitives_amd as ap
    ctx = ap.Context(local_rank)
    fir = ap.Fir(ctx, lowpass_taps(), FFT_LEN)

    nstreams = max(1, args.streams)
    ins, outs = [], []
    for s in range(nstreams):
        ins.append(ctx.vec(synth_stream(815 + 1000 * rank + s, STREAM)))
        outs.append(ctx.empty(STREAM))

    def step(i):
        fir.filter(ins[i % nstreams], out=outs[i % nstreams])

    def barrier():
        ranks.barrier(ctx)

    ev0, ev1 = ctx.event(), ctx.event()
    barrier()                  # RCCL sets its communicator up lazily: have that idle gap here, not next to the timed region
    # Load-onset transient: after >= 5 ms of idle the GPU runs ~2 ms at full speed, then 10-25 % slower for
    # ~25 ms while its power management settles (profiles/r01_fmi_duration_vs_time.json).  A stream processor
    # lives in the settled state, so reach it before the warm-up; nothing below is skipped or shortened.
    nsettle = settle(step, ctx, args.settle_ms)
    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step(args.warmup + i)
    ev1.record()
    ctx.sync()
    ranks.torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kern_ms = ev0.elapsed_ms(ev1) / max(args.steps, 1)       # per launch, on the kernel's own stream
    elapsed = ranks.max_over_ranks(elapsed)
    barrier()
    # a plain device copy of the same bytes, same buffers, right behind the timed region (the
    # fraction is reported against the spec peak AND against what a copy reaches on this device)
    c0, c1 = ctx.event(), ctx.event()
    for i in range(20):
        outs[i % nstreams].vec_clone(ins[i % nstreams])
    c0.record()
    for i in range(200):
        outs[i % nstreams].vec_clone(ins[i % nstreams])
    c1.record(); ctx.sync()
    copy_gbs = BYTES_PER_SAMPLE * STREAM / (c0.elapsed_ms(c1) / 200 * 1e-3) / 1e9
    # Informational extra leg (rank 0, one GPU): the same launches issued alternately on TWO queues.  The drain of
    # one launch then overlaps the fill of the next, which `value` above -- one queue, one launch at a time, the
    # setting the per-kernel roofline needs -- does not show.
    two_q = None
    if args.gpus == 1 and not args.no_two_queues:
        ctx2 = ap.Context(local_rank)
        fir2 = ap.Fir(ctx2, lowpass_taps(), FFT_LEN)
        views = [(ap.context.DeviceVec(ctx2, STREAM, ptr=ins[i].ptr), ap.context.DeviceVec(ctx2, STREAM, ptr=outs[i].ptr))
                 for i in range(nstreams)]
        def step2(i):
            if i & 1: fir2.filter(views[i % nstreams][0], out=views[i % nstreams][1])
            else: step(i)
        for i in range(400): step2(i)
        ctx.sync(); ctx2.sync()
        tq = time.perf_counter()
        for i in range(args.steps): step2(i)
        ctx.sync(); ctx2.sync()
        tq = time.perf_counter() - tq
        two_q = {"queues": 2, "value": round(float(STREAM) * args.steps / tq / 1e9, 3), "unit": "GSamples/s",
                 "pct_of_hbm_roofline": round(100.0 * STREAM * args.steps / tq * BYTES_PER_SAMPLE / (HBM_PEAK_GBS * 1e9), 2),
                 "note": "same kernel and buffers, consecutive launches alternate over two HIP queues (drain/fill overlap); "
                         "not the reported value"}
        del fir2, views
        ctx2.close()

    if rank == 0:
        total_samples = float(STREAM) * args.steps * args.gpus
        value = total_samples / elapsed / 1e9
        achieved = BYTES_PER_SAMPLE * STREAM / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "GSamples/s cf32 (FFT-2048 + 64-tap FIR chain)",
            "value": round(value, 3), "unit": "GSamples/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "settle_launches": nsettle, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C3: 64-tap FIR via overlap-save (FFT-2048) on 16 Mi cf32 samples per step "
                                   "(one fused kernel launch), device-resident, rotating over "
                                   f"{nstreams} stream pairs = {nstreams * 256} MiB",
                       "fft_len": FFT_LEN, "ntaps": NTAPS, "hop": fir.hop, "samples_per_step": STREAM,
                       "parallelism": f"{args.gpus} independent stream(s), one per GPU, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "device_copy_GBps": round(copy_gbs, 1), "frac_of_device_copy": round(achieved / copy_gbs, 4),
                         "traffic": (measured_traffic() or (None, None))[1],
                         "traffic_source": (measured_traffic() or (None, None))[0],
                         "kernel": "fmi_kernel<Cfg<2048,16,16,16,8>>", "kernel_ms": round(kern_ms, 5),
                         "bytes_per_launch": BYTES_PER_SAMPLE * STREAM},
            "pct_of_hbm_roofline": round(100.0 * value / args.gpus * BYTES_PER_SAMPLE / HBM_PEAK_GBS, 2),
        }
        if two_q is not None:
            line["two_queues"] = two_q
        if args.gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)

    ranks.close()


if __name__ == "__main__":
    main()
