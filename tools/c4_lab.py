#!/usr/bin/env python3
"""BASELINE config 4 on ONE GPU (8 channels x 4096 frames of 2048: QPSK mod -> AWGN -> FFT-2048 correlate -> hard demod)
under different orders / queue layouts of the same two fused calls per channel, interleaved A/B in one process:
   interleaved   per channel: modulate_awgn, correlate_demod; one context, one queue (round 3's bench order)
   phased        the 8 modulate_awgn calls, then the 8 correlate_demod calls; one queue
   phased+lane   the same with the context's overlap lane on: consecutive correlate_demod calls alternate queues
   two-ctx       channels dealt to two contexts (two HIP queues), each running its channels interleaved: the
                 VALU-bound generator of one channel beside the memory-bound correlator of another
   two-ctx+lane  two contexts, each phased with its lane on
Every layout's decided bits are compared with the first layout's.
   python3 tools/c4_lab.py [rounds=7] [steps=20]"""
import os, sys, time, json
os.environ.setdefault("AETH_TUNING", "1")          # the grid sweep at the end uses AETH_FIR_GRID_FIRST
import numpy as np
sys.path.insert(0, ".")
import aether_primitives_amd as ap
from aether_primitives_amd import modulation, noise

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N, frames, nch = 2048, 4096, 8
n = N * frames
ref = np.zeros(N, np.complex64); ref[:4] = np.conj(np.array([-1 + 1j, 0, 1 - 1j, 1 - 1j], np.complex64))


class Half:
    def __init__(self, chans):
        self.ctx = ap.Context(0)
        self.q = modulation.qpsk(self.ctx)
        self.f = ap.HipFft(self.ctx, N, max_batch=frames)
        self.sig = self.ctx.vec(ref)
        self.ch = []
        for c in chans:
            rng = np.random.default_rng(815 + c)
            self.ch.append((modulation.DeviceBits(self.ctx, 2 * n, rng.integers(0, 2, 2 * n, dtype=np.uint8)),
                            noise.new(self.ctx, 0.01, 815 + c), self.ctx.empty(n), modulation.DeviceBits(self.ctx, 2 * n)))

    def interleaved(self):
        for bits, awgn, txb, rxb in self.ch:
            tx = self.q.modulate_awgn(bits, awgn, out=txb); self.q.correlate_demod(self.f, tx, self.sig, out=rxb)

    def phased(self):
        for bits, awgn, txb, rxb in self.ch: self.q.modulate_awgn(bits, awgn, out=txb)
        for bits, awgn, txb, rxb in self.ch: self.q.correlate_demod(self.f, txb, self.sig, out=rxb)

    def bits(self):
        return [rxb.to_host().copy() for _, _, _, rxb in self.ch]


one = Half(range(nch))
two = [Half(range(0, nch, 2)), Half(range(1, nch, 2))]


def run_one(kind, lane):
    one.ctx.set_overlap(lane)
    getattr(one, kind)()


def run_two(kind, lane):
    for h in two: h.ctx.set_overlap(lane)
    # issue alternately so that neither queue waits for the host
    if kind == "interleaved":
        for k in range(nch // 2):
            for h in two:
                bits, awgn, txb, rxb = h.ch[k]
                h.q.modulate_awgn(bits, awgn, out=txb); h.q.correlate_demod(h.f, txb, h.sig, out=rxb)
    else:
        for k in range(nch // 2):
            for h in two:
                bits, awgn, txb, rxb = h.ch[k]; h.q.modulate_awgn(bits, awgn, out=txb)
        for k in range(nch // 2):
            for h in two:
                bits, awgn, txb, rxb = h.ch[k]; h.q.correlate_demod(h.f, txb, h.sig, out=rxb)


def sync_all():
    one.ctx.sync()
    for h in two: h.ctx.sync()


layouts = [("interleaved", lambda: run_one("interleaved", False)), ("phased", lambda: run_one("phased", False)),
           ("phased+lane", lambda: run_one("phased", True)), ("two-ctx", lambda: run_two("interleaved", False)),
           ("two-ctx phased", lambda: run_two("phased", False)), ("two-ctx+lane", lambda: run_two("phased", True))]

# parity of the layouts
base = None


def rewind():
    for h in [one] + two:
        for _, awgn, _, _ in h.ch: awgn.offset = 0              # every layout draws the same noise


for name, fn in layouts:
    rewind(); fn(); sync_all()
    if name.startswith("two"):
        got = [None] * nch
        for hi, h in enumerate(two):
            for k, b in enumerate(h.bits()): got[hi + 2 * k] = b
    else:
        got = one.bits()
    if base is None: base = got
    same = all(np.array_equal(a, b) for a, b in zip(base, got))
    print(f"{name:16s} decided bits {'identical' if same else 'DIFFER'}", flush=True)

t = {name: [] for name, _ in layouts}
for r in range(rounds + 1):
    for name, fn in layouts:
        for _ in range(3): fn()
        sync_all(); t0 = time.perf_counter()
        for _ in range(steps): fn()
        sync_all(); el = time.perf_counter() - t0
        if r: t[name].append(el / steps)
rows = {}
print(f"== C4 on one GPU, {rounds} interleaved rounds x {steps} steps (8 channels x {n} samples per step) ==")
for name, _ in layouts:
    v = sorted(t[name]); med = v[len(v) // 2]
    rows[name] = {"ms_per_step": round(med * 1e3, 4), "GS_per_s": round(nch * n / med / 1e9, 1), "frac_of_20B_line": round(nch * n * 20 / med / 8e12, 3)}
    print(f"  {name:16s} {med * 1e3:8.3f} ms  {nch * n / med / 1e9:7.1f} GS/s   {nch * n * 20 / med / 8e12:.3f} of the 20 B/sample line")
# Can the two kernels SHARE compute units?  The correlator holds 248 VGPRs per lane: two of its waves fill a SIMD's
# register file and no generator wave (40 VGPRs) fits beside them.  With fewer correlator workgroups resident (a
# fraction of the full grid, AETH_FIR_GRID_FIRST sixteenths; needs the overlap switch on, the lane itself never
# forms here: a modulate call sits between two correlate calls) some SIMDs keep room for generator waves.
print("== two contexts, interleaved, correlator grid in sixteenths of the resident grid ==")
for g in (16, 14, 12, 10, 8, 6):
    os.environ["AETH_FIR_GRID_FIRST"] = str(g)
    ts = []
    for r in range(rounds):
        for _ in range(3): run_two("interleaved", True)
        sync_all(); t0 = time.perf_counter()
        for _ in range(steps): run_two("interleaved", True)
        sync_all(); ts.append((time.perf_counter() - t0) / steps)
    med = sorted(ts)[len(ts) // 2]
    rows[f"two-ctx grid {g}/16"] = {"ms_per_step": round(med * 1e3, 4), "GS_per_s": round(nch * n / med / 1e9, 1)}
    print(f"  grid {g:2d}/16        {med * 1e3:8.3f} ms  {nch * n / med / 1e9:7.1f} GS/s")
os.environ["AETH_FIR_GRID_FIRST"] = "16"
json.dump(rows, open("gpurun_out/c4_lab.json", "w"), indent=1)
