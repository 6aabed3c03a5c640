#!/usr/bin/env python3
"""MODELLED strong-scaling curve of BASELINE configs 4 and 5 from shard-size runs on ONE GPU (no 8-GPU node is the
builder's to measure): `bench.py --workload W --as-rank 0 --of N` runs rank 0's share of an N-GPU job -- the path has no
exchange step, so nothing else of the N-GPU job is missing -- and the job's throughput is its samples over that time.
C3 (the headline) scales weakly by definition: every GPU filters its own streams, nothing to model.
   python3 tools/scaling_model.py [steps=200]   ->  gpurun_out/scaling_model.json  (copy to profiles/r04_scaling_model.json)"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "200"
out = {"label": "modelled, one GPU", "how": __doc__.split("\n   python3")[0], "rows": []}
for wl in ("c4", "c5"):
    base = None
    for n in (1, 2, 4, 8):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--as-rank", "0", "--of", str(n),
                            "--steps", steps, "--warmup", "20", "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode or not line:
            print(p.stdout, p.stderr); raise SystemExit(f"{wl} of {n} failed")
        d = json.loads(line[-1])
        if base is None: base = d["value"]
        row = {"workload": wl, "gpus_modelled": n, "rank0_ms_per_step": d["ms_per_step"], "job_GSps": d["value"],
               "speedup_vs_1": round(d["value"] / base, 3), "efficiency": round(d["value"] / base / n, 3), "partition": d["config"]["partition"]}
        out["rows"].append(row)
        print(row, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "scaling_model.json"), "w"), indent=1)
