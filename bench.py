#!/usr/bin/env python3
"""bench.py -- headline benchmark of the cf32 hot path on MI355X.

Metric (BASELINE.json): GSamples/s of cf32 through the FFT-2048 + 64-tap FIR
chain, and % of HBM roofline on one GPU.  Workload = BASELINE config 3: a 64-tap
FIR by overlap-save (FFT-2048) over a 16 Mi-sample cf32 stream; one "step" =
one pass over one stream = ONE launch of the fused kernel.  Streams rotate
through a working set larger than the 256 MiB Infinity Cache so every step
reads its input from HBM.  Frames are independent, so N GPUs run N
independent streams (weak scaling, no collective on the data path).

  python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: starts its own N ranks)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(budget_s=16.0):
    """The CPU restatement of the reference chain (oracle/aeth_oracle.c: rfft -> vec_mul -> rifft per overlap-save
    block, f32) on a bounded sample of the same workload, on this box's host cores: 1 thread (the reference's own
    per-call path is single-threaded) and T = every core this process may run on.  Timed
    on a `-O3 -march=native -ffp-contract=off` build made on this machine; the portable -O2 build if that fails."""
    import ctypes as C
    from oracle import pyoracle as orc
    n = 1 << 22
    x = orc.synth_cnormal(815, n)
    taps = orc.synth_lowpass_taps(NTAPS, 0.25)
    y = np.empty_like(x)
    hop = FFT_LEN - NTAPS + 1 - ((FFT_LEN - NTAPS + 1) % 64)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nat, flags = orc.native_fir_lib()
    if nat is not None:
        def run(th):
            rc = nat.orc_fir_ols_f32_mt(taps.ctypes.data, taps.size, FFT_LEN, hop, x.ctypes.data, n, y.ctypes.data, th)
            assert rc == 0
    else:
        flags = "-O2 -ffp-contract=off (portable build; " + flags + ")"
        def run(th):
            orc.fir_ols_f32(taps, x, FFT_LEN, hop, threads=th)
    # T: every core this process may run on, unless fewer threads are faster (a GPU box hands one GPU's job a
    # share of a big host: 256 visible hardware threads, far fewer schedulable at once) -- short sweep, best kept
    cores = min(cores, 256)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    cands = sorted({c for c in (8, 16, 32, 64, 128, cores, quota or cores) if c and c <= cores})
    best_t, best_r = cores, 0.0
    run(cores)                                     # page the buffers in
    for c in cands:
        t0 = time.perf_counter(); reps = 0
        while time.perf_counter() - t0 < 1.0:
            run(c); reps += 1
        r = reps * n / (time.perf_counter() - t0)
        if r > best_r:
            best_t, best_r = c, r
    res = {}
    for label, th in (("1", 1), ("T", best_t)):
        done, t0 = 0, time.perf_counter()
        while True:
            run(th)
            done += n
            el = time.perf_counter() - t0
            if el >= budget_s / 2:
                break
        res[label] = done / el / 1e9
        res[label + "_n"] = done
    # kind: "port" in the bench contract's vocabulary (reference | port); SURVEY 8d's name for it is the label
    return {"value": round(res["T"], 5), "unit": "GSamples/s", "cores": best_t, "kind": "port",
            "label": "c++-restatement-of-rust-path",
            "value_1thread": round(res["1"], 5), "cpu_model": cpu_model(), "cores_visible": cores,
            "cgroup_cpu_quota": quota, "threads_tried": cands, "build": "gcc " + flags,
            "sample": f"oracle/aeth_oracle.c overlap-save chain (rfft->vec_mul->rifft, f32) over "
                      f"{res['T_n'] >> 20} Mi samples on {best_t} threads / {res['1_n'] >> 20} Mi on 1 thread, "
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
            # never report a line whose n_gpus is not the number of ranks that ran
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}: launch "
                             f"`python bench.py --gpus {args.gpus}` (it starts its own ranks) or pass --gpus {self.world}")
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

    def ranks_seen(self):
        """Number of ranks the collective backend actually spans (the line's n_gpus must equal it)."""
        if self.dist is None:
            return 1
        t = self.torch.ones(1, device="cuda")
        self.dist.all_reduce(t)
        return int(round(float(t.item())))

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
    C4 and C5 are the node-level jobs BASELINE.json names, partitioned with aether_primitives_amd.sharding (the
    helpers tests/test_dist_shard.py validates): C4 = 8 independent channels, channel c on rank c % N; C5 = 512
    frames of 65536 samples, contiguous frame ranges -- fixed total work, so "scaling" is strong.  C2 is one
    cache-resident 1 Mi-sample stream per rank (weak)."""
    ranks = Ranks(args)
    import aether_primitives_amd as ap
    from aether_primitives_amd import Scale, sampling, modulation, noise
    from aether_primitives_amd.sharding import channel_of, frame_shard
    ctx = ap.Context(ranks.local_rank)
    N = 2048
    scaling = "strong"
    # --as-rank R --of W (one GPU): run rank R's share of a W-rank job and report the throughput W such GPUs would
    # give together -- a MODEL of the strong-scaling curve (no communication exists to model; labelled in the line)
    modelled = args.as_rank is not None
    if modelled:
        assert args.gpus == 1 and args.of >= 1 and 0 <= args.as_rank < args.of, "--as-rank R --of W needs --gpus 1 and R < W"
    p_rank, p_world = (args.as_rank, args.of) if modelled else (ranks.rank, args.gpus)
    if args.workload == "c2":
        scaling = "weak"
        n = 1 << 20                                             # 512 frames: 8 MiB, cache-resident by definition
        # The two launches of a step are 3 us of traffic each: launch-latency-bound.  Steps work on independent buffer
        # pairs, so they are dealt alternately to two contexts = two HIP queues of this GPU (as for C4): the launch
        # latency of one step runs beside the kernels of the other.  --c2-one-queue: round 3's single context.
        nq = 1 if args.c2_one_queue else 2
        ctxs = [ctx] + [ap.Context(ranks.local_rank) for _ in range(nq - 1)]
        plans = [ap.HipFft(c, N, max_batch=n // N) for c in ctxs]
        bufs = [(ctxs[i % nq].vec(synth_stream(815 + i, n)), ctxs[i % nq].empty(n)) for i in range(4)]
        # f.fwd(a, b, Scale.SN); f.ifwd(a, Scale.SN) (benches.rs:305-306,352-353) as the two C-ABI calls they are, argument
        # tuples built once: the host side of a step is two ctypes calls and nothing else (the interpreter is not the product)
        from aether_primitives_amd._lib import check as _check
        from aether_primitives_amd.fft import SIGN_REF_FWD
        _fx = plans[0]._lib.aeth_fft_exec
        _c2 = []
        for i in range(4):
            a, b = bufs[i]; f = plans[i % nq]
            _c2.append(((f.h, a._p(), a.n, b._p(), a.n // N, SIGN_REF_FWD, Scale.SN.kind, Scale.SN.x),
                        (f.h, a._p(), a.n, a._p(), a.n // N, SIGN_REF_FWD, Scale.SN.kind, Scale.SN.x)))
        def step(i):
            fw, iw = _c2[i % 4]
            _check(_fx(*fw)); _check(_fx(*iw))

        class _All:
            def sync(self_):
                for c in ctxs: c.sync()
        ctx = _All()
        job_samples = 2 * n * args.gpus
        name, bytes_ = f"C2: FFT-2048 fwd (copy) + ifwd (in place) on a 1 Mi-sample stream per GPU, {nq} queue(s) per GPU", 32 * n
        shard = "one stream per rank"
    elif args.workload == "c5":
        total_frames, nb = 512, 9
        lo, frames = frame_shard(total_frames, p_rank, p_world)
        n = 65536 * frames
        f = ap.HipFft(ctx, 65536, max_batch=max(frames, 1))
        nbuf = 2 if frames > 128 else 3
        bufs = [(ctx.vec(synth_stream(815 + lo + 7919 * i, n)), ctx.empty((65536 + 65535 * nb) * frames)) for i in range(nbuf)]
        if args.c5_unfused:
            def step(i):                                        # the two trait-level calls, spectrum through HBM
                a, o = bufs[i % nbuf]; f.ifwd(a, Scale.SN); sampling.interpolate(ctx, a, o, nb, frame_len=65536)
        else:
            def step(i):                                        # one call (spectrum through the plan's temp)
                a, o = bufs[i % nbuf]; f.rfft_interpolate(a, o, nb, Scale.SN)
        job_samples = 65536 * total_frames
        name = "C5: 512 x 65536-point FFT (Scale::SN) + 10x linear interpolation, frame-sharded" + (
            " (two calls)" if args.c5_unfused else " (aeth_fft_exec_interpolate)")
        bytes_ = 104 * n
        shard = f"frame_shard: rank {p_rank} of {p_world} owns frames [{lo}, {lo + frames}) of {total_frames}"
    else:
        # Work units = (channel, frame range): this rank's channels, each cut into as many frame ranges as it takes to
        # have at least two units (frames are independent: the noise stream is addressed by position, so a cut channel
        # decides the same bits).  The units are dealt alternately to TWO contexts = two HIP queues of this GPU, each
        # running its units' two fused calls in order: the VALU-bound generator of one unit runs beside the
        # memory-bound correlator of another (tools/c4_lab.py, profiles/r04_c4_lab.json: 144 -> 163 GS/s on one GPU).
        n_channels, frames = 8, 4096
        mine = channel_of(p_rank, p_world, n_channels)
        cuts = 1 if len(mine) >= 2 or args.c4_one_queue else 2
        nq = 1 if args.c4_one_queue else 2
        ctxs = [ctx] + [ap.Context(ranks.local_rank) for _ in range(nq - 1)]
        ref = np.zeros(N, np.complex64); ref[:4] = np.conj(np.array([-1 + 1j, 0, 1 - 1j, 1 - 1j], np.complex64))
        per = [dict(ctx=c, q=modulation.qpsk(c), f=ap.HipFft(c, N, max_batch=frames // cuts), sig=c.vec(ref), units=[]) for c in ctxs]
        k = 0
        for ch in mine:                                         # channel ch: its own bits and noise seed (815 + ch)
            rng = np.random.default_rng(815 + ch)
            allbits = rng.integers(0, 2, 2 * N * frames, dtype=np.uint8)
            for cut in range(cuts):
                fr = frames // cuts
                n_u = N * fr
                h = per[k % nq]; k += 1
                awgn = noise.new(h["ctx"], 0.01, 815 + ch)
                h["units"].append(dict(bits=modulation.DeviceBits(h["ctx"], 2 * n_u, allbits[2 * n_u * cut: 2 * n_u * (cut + 1)]),
                                       awgn=awgn, pos=n_u * cut, tx=h["ctx"].empty(n_u), rx=modulation.DeviceBits(h["ctx"], 2 * n_u)))
        n = N * frames
        rounds = max(len(h["units"]) for h in per)
        if args.c4_unfused:
            def step(i):                                        # four launches per unit: 52 B of traffic per sample
                for r in range(rounds):
                    for h in per:
                        if r < len(h["units"]):
                            u = h["units"][r]; u["awgn"].offset = u["pos"]
                            tx = h["q"].modulate(u["bits"], out=u["tx"]); u["awgn"].apply(tx); h["f"].mul_chain(tx, h["sig"]); h["q"].demod_naive(tx, out=u["rx"])
        else:
            def step(i):                                        # two launches: modulate+AWGN, correlate+demod: 20 B per sample
                for r in range(rounds):
                    for h in per:
                        if r < len(h["units"]):
                            u = h["units"][r]; u["awgn"].offset = u["pos"]
                            tx = h["q"].modulate_awgn(u["bits"], u["awgn"], out=u["tx"]); h["q"].correlate_demod(h["f"], tx, h["sig"], out=u["rx"])

        class _All:                                             # what the harness syncs: every queue of this rank
            def sync(self_):
                for c in ctxs: c.sync()
        ctx = _All()
        job_samples = n * n_channels
        name = "C4: 8 channels x (QPSK mod -> AWGN -> FFT-2048 correlate -> hard demod), 4096 frames each" + (
            " (four calls)" if args.c4_unfused else " (modulate_awgn + mul_ifft_demod)") + f", {nq} queue(s) per GPU"
        # unfused: modulate 8 W + awgn 8 R + 8 W + correlate 8 R + 8 W + demod 8 R (+ 2 x 2 B of bits) = 52 B/sample;
        # fused: modulate_awgn 2 R + 8 W, correlate + demod 8 R + 2 W = 20 B/sample
        bytes_ = (52 if args.c4_unfused else 20) * n * len(mine)
        shard = f"channel_of: rank {p_rank} of {p_world} runs channels {mine}" + (f", each cut into {cuts} frame ranges" if cuts > 1 else "")
    ranks.barrier(ctx)
    seen = ranks.ranks_seen()
    nsettle = settle(step, ctx, args.settle_ms)
    for i in range(args.warmup): step(i)
    ranks.barrier(ctx); t0 = time.perf_counter()
    for i in range(args.steps): step(args.warmup + i)
    ctx.sync(); el = ranks.max_over_ranks(time.perf_counter() - t0)
    if ranks.rank == 0:
        assert seen == args.gpus, f"collective spans {seen} ranks, --gpus says {args.gpus}"
        extra = {"modelled": f"one GPU ran rank {p_rank}'s share of a {p_world}-GPU job; value = the job's samples / this rank's time "
                             f"(the ranks share nothing, so the job ends when its slowest rank does)", "as_rank": p_rank, "of": p_world} if modelled else {}
        print(json.dumps({"metric": "GSamples/s cf32", "value": round(job_samples * args.steps / el / 1e9, 3), **extra,
                          "unit": "GSamples/s", "n_gpus": seen, "steps": args.steps, "warmup": args.warmup,
                          "settle_launches": nsettle, "ms_per_step": round(el / args.steps * 1e3, 5),
                          "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": name, "partition": shard},
                          "algorithmic_GBps_rank0": round(bytes_ * args.steps / el / 1e9, 1)}), flush=True)
    ranks.close()


def criterion_shapes(args):
    """The reference's criterion shapes (benches/benches.rs:30-67 vec ops @ 2048, :92 interpolate, :113,130 downsample,
    :310-377 FFT 512/1024/2048, :410-420 correlator, :223,260 mod/demod), ONE frame per call as the reference runs
    them: the oracle's CPU time (this leg is bench.py's CPU-baseline use of oracle/), the literal host flavour (host
    slice in, H2D -> kernel -> D2H, host slice out: the drop-in trait call) and the device flavour (operands resident,
    one launch; `sync` = call + wait, `pipelined` = 200 calls back to back / 200).  Then, for FFT-2048 and the
    correlator chain, the batch size from which each flavour beats `batch` CPU calls."""
    import ctypes as C
    import aether_primitives_amd as ap
    from aether_primitives_amd import Scale, sampling, modulation
    from oracle import pyoracle
    ctx = ap.Context(0)
    lib = ctx._lib
    L, flags = pyoracle.native_fir_lib()
    if L is not None:
        L.orc_time_shape.restype = C.c_double; L.orc_time_shape.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_int]
        cpu = lambda op, n, b=0: min(L.orc_time_shape(op, n, b, 2000) for _ in range(3)) * 1e6
    else:
        flags = "portable -O2 build"
        cpu = lambda op, n, b=0: min(pyoracle.time_shape(op, n, b, 2000) for _ in range(3)) * 1e6

    def t_sync(fn, reps=200):
        for _ in range(10): fn()
        ctx.sync(); best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps): fn(); ctx.sync()
            best = min(best, (time.perf_counter() - t0) / reps)
        return best * 1e6

    def t_pipe(fn, reps=200):
        for _ in range(10): fn()
        ctx.sync(); best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps): fn()
            ctx.sync(); best = min(best, (time.perf_counter() - t0) / reps)
        return best * 1e6

    rows = []
    def row(name, ref, cpu_us, host_fn, dev_fn):
        r = {"shape": name, "reference": ref, "cpu_us": round(cpu_us, 3), "host_flavour_us": round(t_sync(host_fn, 100), 2),
             "device_sync_us": round(t_sync(dev_fn), 2), "device_pipelined_us": round(t_pipe(dev_fn), 2)}
        rows.append(r); print(r, flush=True)

    ones = lambda n: np.full(n, 1 + 1j, np.complex64)
    # vec ops @ 2048 (benches.rs:30-67)
    a, b = ap.HostVec(ctx, ones(2048)), ap.HostVec(ctx, ones(2048))
    da, db = ctx.vec(ones(2048)), ctx.vec(ones(2048))
    row("vec_mul 2048", "benches.rs:37", cpu(0, 2048), lambda: a.vec_mul(b.a), lambda: da.vec_mul(db))
    row("vec_scale 2048", "benches.rs:48", cpu(1, 2048), lambda: a.vec_scale(1.0), lambda: da.vec_scale(1.0))
    row("vec_clone 2048", "benches.rs:59", cpu(2, 2048), lambda: a.vec_clone(b.a), lambda: da.vec_clone(db))
    # interpolate (len, n_between): the bench passes n_between = 4 whatever the tuple says (benches.rs:88)
    for n in (1024, 2048, 400):
        src = (np.arange(n) + 0j).astype(np.complex64); dsrc = ctx.vec(src); ddst = ctx.empty(n + (n - 1) * 4)
        hdst = np.empty(n + (n - 1) * 4, np.complex64); nw = C.c_size_t()
        host = lambda: lib.aeth_host_interpolate(ctx.h, src.ctypes.data_as(C.c_void_p), n, hdst.ctypes.data_as(C.c_void_p), hdst.size, 4, 1, C.byref(nw))
        row(f"interpolate {n} x5", "benches.rs:76-92", cpu(3, n, 4), host, lambda: sampling.interpolate(ctx, dsrc, ddst, 4))
    # downsample, release build (benches.rs:99-130)
    for n, m in ((30720, 1024), (8096, 512)):
        src = ones(n); dst = np.empty(m, np.complex64); dsrc = ctx.vec(src); ddst = ctx.empty(m)
        row(f"downsample {n} -> {m}", "benches.rs:113,130", cpu(4, n, m), lambda: sampling.downsample(ctx, src, dst, release=True),
            lambda: sampling.downsample(ctx, dsrc, ddst, release=True))
    # FFT in place / copy, Scale::SN (benches.rs:294-377), correlator chain (:388-420)
    for n in (512, 1024, 2048):
        f = ap.HipFft(ctx, n); x = ones(n); y = np.empty(n, np.complex64); dx = ctx.vec(x); dy = ctx.empty(n)
        sigd = ctx.vec(ones(n))
        row(f"fft ifwd SN {n}", "benches.rs:294-310", cpu(5, n), lambda: f.ifwd(x, Scale.SN), lambda: f.ifwd(dx, Scale.SN))
        row(f"fft fwd SN copy {n}", "benches.rs:340-357", cpu(6, n), lambda: f.fwd(x, y, Scale.SN), lambda: f.fwd(dx, dy, Scale.SN))
        hx = ap.HostVec(ctx, x); hs = ap.HostVec(ctx, ones(n))
        row(f"correlator chain {n}", "benches.rs:388-420", cpu(7, n),
            lambda: hx.vec_rfft(f, Scale.NONE).vec_mul(hs.a).vec_rifft(f, Scale.NONE), lambda: f.mul_chain(dx, sigd))
    # modulation (benches.rs:210-278)
    q = modulation.qpsk(ctx)
    for n in (100, 8000):
        bits = np.random.default_rng(n).integers(0, 2, n - n % 2, dtype=np.uint8); dbits = modulation.DeviceBits(ctx, bits.size, bits)
        dsym = ctx.empty(bits.size // 2); dout = modulation.DeviceBits(ctx, 2 * n); syms = ctx.vec(ones(n))
        row(f"qpsk modulate {n} bits", "benches.rs:210-223", cpu(8, n), lambda: q.modulate(bits).to_host(), lambda: q.modulate(dbits, out=dsym))
        row(f"qpsk demod_naive {n} symbols", "benches.rs:245-260", cpu(9, n), lambda: q.demod_naive(syms, out=dout).to_host(), lambda: q.demod_naive(syms, out=dout))
    # from which batch does the GPU win?  FFT-2048 ifwd(SN) and the correlator chain on `batch` frames in ONE call
    cross = []
    f = ap.HipFft(ctx, 2048, max_batch=4096); sigd = ctx.vec(ones(2048))
    c_fft, c_chain = cpu(5, 2048), cpu(7, 2048)
    for batch in (1, 2, 4, 8, 16, 64, 256, 1024, 4096):
        x = np.tile(ones(2048), batch); dx = ctx.vec(x); hv = np.empty_like(x)
        def host_fft():                                     # upload, one batched launch, download: what a caller with host data pays
            ctx.upload(dx.ptr, x); f.ifwd(dx, Scale.SN); ctx.download(dx.ptr, hv)
        def host_chain():
            ctx.upload(dx.ptr, x); f.mul_chain(dx, sigd); ctx.download(dx.ptr, hv)
        r = {"batch": batch, "cpu_fft_us": round(c_fft * batch, 1), "gpu_fft_device_us": round(t_sync(lambda: f.ifwd(dx, Scale.SN), 50), 2),
             "gpu_fft_host_data_us": round(t_sync(host_fft, 30), 2), "cpu_chain_us": round(c_chain * batch, 1),
             "gpu_chain_device_us": round(t_sync(lambda: f.mul_chain(dx, sigd), 50), 2), "gpu_chain_host_data_us": round(t_sync(host_chain, 30), 2)}
        cross.append(r); print(r, flush=True)
    out = {"cpu": f"oracle/aeth_oracle.c (c++-restatement-of-rust-path), 1 thread, gcc {flags}, {cpu_model()}", "one_frame_per_call": rows,
           "batched_fft2048": cross}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "criterion_shapes.json"), "w"), indent=1)


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as a CHILD process
    (torch.distributed.run) before this process touches torch or the GPU, and leave with its return code."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


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
    ap_.add_argument("--no-overlap", action="store_true",
                     help="run the timed region on ONE HIP queue (aeth_ctx_set_overlap off): every launch then waits for "
                          "the previous one to drain; this is also the mode to profile per-kernel durations in")
    ap_.add_argument("--no-single-queue-leg", action="store_true",
                     help="skip the extra leg that repeats the timed steps on one queue (per-launch kernel time)")
    ap_.add_argument("--as-rank", type=int, default=None, help="--workload c4|c5 on ONE GPU: run this rank's share of an --of W rank job (modelled scaling)")
    ap_.add_argument("--of", type=int, default=1)
    ap_.add_argument("--c2-one-queue", action="store_true", help="--workload c2 on one context / one HIP queue per GPU (round 3's layout)")
    ap_.add_argument("--c4-one-queue", action="store_true", help="--workload c4 on one context / one HIP queue per GPU (round 3's layout)")
    ap_.add_argument("--c4-unfused", action="store_true", help="--workload c4 as four calls per channel (modulate, apply, mul_chain, demod)")
    ap_.add_argument("--c5-unfused", action="store_true", help="--workload c5 as two calls (fft, then interpolate)")
    ap_.add_argument("--workload", default="c3", choices=["c3", "c2", "c4", "c5", "criterion"],
                     help="c3 (default) = the headline config; the others are BASELINE configs 2, 4, 5 for the record")
    args = ap_.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    if args.workload == "criterion":
        return criterion_shapes(args)
    if args.workload != "c3":
        return side_workload(args)

    ranks = Ranks(args)
    rank, local_rank = ranks.rank, ranks.local_rank

    import aether_primitives_amd as ap
    ctx = ap.Context(local_rank)
    fir = ap.Fir(ctx, lowpass_taps(), FFT_LEN)
    # Two HIP queues inside the context (aeth_ctx_set_overlap): consecutive steps touch different streams, so the
    # library lets launch k+1 start while launch k drains; anything else on the context (events, copies, syncs)
    # is ordered behind both queues.  --no-overlap = one queue, one launch at a time.
    overlap = not args.no_overlap
    ctx.set_overlap(overlap)

    nstreams = max(1, args.streams)
    ins, outs = [], []
    for s in range(nstreams):
        ins.append(ctx.vec(synth_stream(815 + 1000 * rank + s, STREAM)))
        outs.append(ctx.empty(STREAM))

    # one step = one aeth_fir_exec through the C ABI; the argument tuples are built once so that the host side of a
    # step is the ctypes call and nothing else (the first launch of a timed region waits on exactly that)
    from aether_primitives_amd._lib import check as _check
    _exec = fir._lib.aeth_fir_exec
    _args = [(fir.h, None, ins[k]._p(), STREAM, outs[k]._p()) for k in range(nstreams)]

    def step(i):
        _check(_exec(*_args[i % nstreams]))

    def barrier():
        ranks.barrier(ctx)

    def timed(steps, first, events=False):
        """barrier + sync | `steps` launches | sync: wall seconds (this rank); with events=True HIP events are recorded
        around the launches as well (behind both queues) and their elapsed ms is returned too"""
        e0, e1 = (ctx.event(), ctx.event()) if events else (None, None)
        barrier()
        t0 = time.perf_counter()
        if events:
            e0.record()
        for i in range(steps):
            step(first + i)
        if events:
            e1.record()
        ctx.sync()
        ranks.torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        return wall, (e0.elapsed_ms(e1) if events else None)

    barrier()                  # RCCL sets its communicator up lazily: have that idle gap here, not next to the timed region
    seen = ranks.ranks_seen()
    # Load-onset transient: after >= 5 ms of idle the GPU runs ~2 ms at full speed, then 10-25 % slower for
    # ~25 ms while its power management settles (profiles/r01_fmi_duration_vs_time.json).  A stream processor
    # lives in the settled state, so reach it before the warm-up; nothing below is skipped or shortened.
    nsettle = settle(step, ctx, args.settle_ms)
    for i in range(args.warmup):
        step(i)
    # THE timed region (`value`, `ms_per_step`, `roofline.achieved` / `frac` all come from this one clock): barrier +
    # sync | exactly K steps | sync + barrier, wall time, MAX over ranks.  Nothing but the K launches is enqueued inside.
    elapsed, _ = timed(args.steps, args.warmup)
    elapsed = ranks.max_over_ranks(elapsed)
    # The same K steps again with HIP events recorded around them on the context's stream (behind both queues):
    # `roofline.frac_events`, the device-side view of the same region without the host's launch and wake-up latency.
    for i in range(args.warmup):
        step(i)
    _, ev_ms = timed(args.steps, args.warmup, events=True)
    step_ms = ev_ms / max(args.steps, 1)

    # Extra leg (rank 0's GPU, untimed for `value`): the same steps on ONE queue.  Per-launch kernel time as
    # rocprofv3 --kernel-trace sees it (profiles/): with two queues the dispatches overlap and their individual
    # durations no longer add up to the wall time.
    single = None
    if overlap and not args.no_single_queue_leg:
        ctx.set_overlap(False)
        for i in range(max(args.warmup, 5)):
            step(i)
        w1, e1 = timed(args.steps, args.warmup, events=True)
        ctx.set_overlap(True)
        k_ms = e1 / max(args.steps, 1)
        single = {"queues": 1, "kernel_ms": round(k_ms, 5),
                  "value": round(float(STREAM) * args.steps / w1 / 1e9, 3), "unit": "GSamples/s",
                  "achieved_GBps": round(BYTES_PER_SAMPLE * STREAM / (k_ms * 1e-3) / 1e9, 1),
                  "frac": round(BYTES_PER_SAMPLE * STREAM / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                  "note": "one launch at a time: fill + drain of every launch exposed; this is the duration a "
                          "kernel trace reports per dispatch"}
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

    if rank == 0:
        assert seen == args.gpus, f"collective spans {seen} ranks, --gpus says {args.gpus}"
        total_samples = float(STREAM) * args.steps * args.gpus
        value = total_samples / elapsed / 1e9
        # per GPU, from the SAME clock as `value`: algorithmic bytes of one launch / (wall time of the region / K)
        achieved = BYTES_PER_SAMPLE * STREAM / (elapsed / args.steps) / 1e9
        achieved_ev = BYTES_PER_SAMPLE * STREAM / (step_ms * 1e-3) / 1e9
        traffic = measured_traffic() or (None, None)
        line = {
            "metric": "GSamples/s cf32 (FFT-2048 + 64-tap FIR chain)",
            "value": round(value, 3), "unit": "GSamples/s", "n_gpus": seen, "steps": args.steps,
            "warmup": args.warmup, "settle_launches": nsettle, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C3: 64-tap FIR via overlap-save (FFT-2048) on 16 Mi cf32 samples per step "
                                   "(one fused kernel launch), device-resident, rotating over "
                                   f"{nstreams} stream pairs = {nstreams * 256} MiB",
                       "fft_len": FFT_LEN, "ntaps": NTAPS, "hop": fir.hop, "samples_per_step": STREAM,
                       "queues_per_gpu": 2 if overlap else 1,
                       "launch_overlap": ("consecutive steps (independent streams) alternate between the context's two "
                                          "HIP queues, at most two launches in flight (aeth_ctx_set_overlap)")
                                         if overlap else "none: one queue, one launch at a time",
                       "parallelism": f"{args.gpus} independent stream(s), one per GPU, no collective"},
            # frac == value / n_gpus * 16 B / 8 TB/s: same clock as `value` (the wall time of the timed region)
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "level": "step: algorithmic bytes of one launch / (wall time of the timed region / steps), per GPU" + (
                             "; two launches overlap, see per_dispatch_* for the duration of one dispatch" if overlap else ""),
                         "achieved_events": round(achieved_ev, 1), "frac_events": round(achieved_ev / HBM_PEAK_GBS, 4),
                         "events_note": "the same K steps repeated with HIP events recorded around them on the context's "
                                        "stream, behind both queues (device-side time: no host launch / wake-up latency)",
                         "device_copy_GBps": round(copy_gbs, 1), "frac_of_device_copy": round(achieved / copy_gbs, 4),
                         "traffic": traffic[1], "traffic_source": traffic[0],
                         "kernel": "fmi_kernel<Cfg<2048,16,16,16,8>> (V_PRIO|V_XOR; +V_SPREAD for a launch that runs alone)", "step_ms_events": round(step_ms, 5),
                         "bytes_per_launch": BYTES_PER_SAMPLE * STREAM},
        }
        if single is not None:
            line["single_queue"] = single
            # the same kernel one dispatch at a time (what rocprofv3 --kernel-trace --stats averages): kept inside the
            # roofline object too, so that nobody reads the step-level fraction as a per-dispatch one
            line["roofline"]["per_dispatch_ms"] = single["kernel_ms"]
            line["roofline"]["per_dispatch_frac"] = single["frac"]
        if args.gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)

    ranks.close()


if __name__ == "__main__":
    main()
