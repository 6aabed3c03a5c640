#!/usr/bin/env python3
"""The generator kernels under the SQ counters: 60 launches each of awgn fill, awgn apply and modulate_awgn on 2^25
samples (run it under rocprofv3 --pmc ...; `--summary dir` condenses the csv of the passes into per-launch averages).
   rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_VALU SQ_LEVEL_WAVES \\
       --output-format csv -d gpurun_out/gen_pmc/p1 -- python3 tools/gen_counters.py
   python3 tools/gen_counters.py --summary gpurun_out/gen_pmc"""
import csv, glob, json, os, sys
if len(sys.argv) > 2 and sys.argv[1] == "--summary":
    res = {}
    for f in sorted(glob.glob(os.path.join(sys.argv[2], "p*", "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            name = next((n for n in ("awgn_fill_kernel", "awgn_apply_kernel", "modulate_awgn_kernel") if n in k), None)
            if not name: continue
            a = res.setdefault(name, {}).setdefault(row["Counter_Name"], [0.0, 0])
            a[0] += float(row["Counter_Value"]); a[1] += 1
    out = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in res.items()}
    for k, d in out.items():
        if "SQ_WAVE_CYCLES" in d and "SQ_ACTIVE_INST_VALU" in d:
            d["valu_issue_share_of_wave_cycles"] = round(d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"], 4)
        if "SQ_WAVE_CYCLES" in d and "SQ_WAIT_INST_ANY" in d:
            d["waiting_share_of_wave_cycles"] = round(d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 4)
        if "SQ_INSTS_VALU" in d and "SQ_WAVES" in d:
            d["valu_instructions_per_wave"] = round(d["SQ_INSTS_VALU"] / d["SQ_WAVES"], 1)
    print(json.dumps(out, indent=1, sort_keys=True))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from aether_primitives_amd import modulation, noise
ctx = ap.Context(0)
n = 1 << 25
g = noise.new(ctx, 0.5, 815)
x = ctx.empty(n)
q = modulation.qpsk(ctx)
bits = modulation.DeviceBits(ctx, 2 * n, np.random.default_rng(1).integers(0, 2, 2 * n, dtype=np.uint8))
for i in range(60): g.fill(x)
for i in range(60): g.apply(x)
for i in range(60): q.modulate_awgn(bits, g, out=x)
ctx.sync()
