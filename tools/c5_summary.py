#!/usr/bin/env python3
"""gpurun_out/c5_<tag>/ (tools/c5_profile.sh) -> profiles/<tag>_c5.json: bench lines of BASELINE config 5, per-kernel
durations and HBM bytes (FETCH_SIZE doubled on gfx950: MI355X_MICROARCH.md, HBM section) of its three kernels."""
import csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = f"gpurun_out/c5_{tag}"
out = {"tag": tag, "workload": "C5: 512 frames x 65536-point FFT (Scale::SN) + sampling::interpolate(n_between = 9), one GPU",
       "algorithmic_bytes_per_sample": {"fft": 16, "interpolate": 88, "chain": 104}, "samples_per_step": 512 * 65536}
for k in ("bench_one_call", "bench_two_calls"):
    p = os.path.join(src, k + ".json")
    if os.path.exists(p):
        try: out[k] = json.loads(open(p).read().strip().splitlines()[-1])
        except Exception: pass
def newest(pat):
    f = sorted(glob.glob(os.path.join(src, pat)), key=os.path.getmtime)
    return f[-1] if f else None
kern = {}
st = newest("trace/*/*_kernel_stats.csv")
if st:
    for r in csv.DictReader(open(st)):
        n = r["Name"]
        key = "fourstep_cols" if "fourstep_cols" in n else "fourstep_rows" if "fourstep_rows" in n else "interpolate" if "interpolate_kernel" in n else None
        if key: kern.setdefault(key, {})["avg_us"] = float(r["AverageNs"]) / 1e3; kern[key]["calls"] = int(r["Calls"])
for cname, pat, mult in (("FETCH_SIZE", "fetch/*/*_counter_collection.csv", 2 * 1024), ("WRITE_SIZE", "write/*/*_counter_collection.csv", 1024)):
    f = newest(pat)
    if not f: continue
    acc = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = "fourstep_cols" if "fourstep_cols" in n else "fourstep_rows" if "fourstep_rows" in n else "interpolate" if "interpolate_kernel" in n else None
        if key and r["Counter_Name"] == cname: acc.setdefault(key, []).append(float(r["Counter_Value"]))
    for key, v in acc.items():
        v = v[-20:]
        kern.setdefault(key, {})["hbm_read_MB" if cname == "FETCH_SIZE" else "hbm_write_MB"] = sum(v) / len(v) * mult / 1e6
n = 512 * 65536
for key, alg in (("fourstep_cols", 16 * n), ("fourstep_rows", 16 * n), ("interpolate", 88 * n)):
    if key in kern:
        k = kern[key]
        k["algorithmic_MB (this kernel's own reads + writes)"] = alg / 1e6
        if "avg_us" in k: k["own_traffic_TBps"] = alg / k["avg_us"] / 1e6
out["kernels"] = kern
if all(k in kern and "avg_us" in kern[k] for k in ("fourstep_cols", "fourstep_rows", "interpolate")):
    t = sum(kern[k]["avg_us"] for k in ("fourstep_cols", "fourstep_rows", "interpolate"))
    out["chain_us_sum_of_kernels"] = t
    out["chain_algorithmic_TBps (104 B/sample)"] = 104 * n / t / 1e6
    out["fft_frac_of_8TBps (16 B/sample over both passes)"] = 16 * n / (kern["fourstep_cols"]["avg_us"] + kern["fourstep_rows"]["avg_us"]) / 1e6 / 8
out["tried_and_not_kept"] = {
    "frame groups through a cache-sized work buffer (AETH_4S_GROUP_MIB, tools/tune_4step.py)": "512 x 65536 FFT: 176 us as two launches; 185 / 202 / 222 / 308 us with groups of 128 / 64 / 32 / 16 MiB",
    "interpolation fused into step B (17 rows per workgroup, interpolated runs written instead of X)": "chain 875 us (1058 / 1231 us in two other store arrangements) against 653 us for the separate kernels; bit-identical output",
    "one persistent launch with per-XCD tickets (round 1)": "292-374 us with the dependency waits against 207-212 us as two launches"}
os.makedirs("profiles", exist_ok=True)
json.dump(out, open(f"profiles/{tag}_c5.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("bench")}, indent=1))
