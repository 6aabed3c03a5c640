#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (rocprofv3 csv) into profiles/<tag>_*.{csv,json}.

HBM traffic follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are
collected in separate --pmc passes, are reported in KiB, and on gfx950 FETCH_SIZE
counts 64 B per 128-B request of a wide coalesced stream, i.e. it reads HALF the
fetched bytes -> doubled here; WRITE_SIZE is exact for streaming stores."""
import csv, glob, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
KERNEL = "fmi_kernel"


def one(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)     # newest run
    return f[-1] if f else None


NTIMED = 1000
out = {"tag": tag, "command_one_queue": "python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-single-queue-leg --no-overlap",
       "command_two_queues": "python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-single-queue-leg"}
st = one("trace1q/*/*_kernel_stats.csv")
if st:
    rows = list(csv.DictReader(open(st)))
    with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
    # (a lone launch runs the V_SPREAD instantiation, a launch beside its predecessor the burst one: take the row
    # that carries the timed launches)
    for r in sorted((r for r in rows if KERNEL in r["Name"]), key=lambda r: int(r["Calls"])):
        out["kernel"] = r["Name"]; out["calls"] = int(r["Calls"])
        out["avg_ns"] = float(r["AverageNs"]); out["min_ns"] = float(r["MinNs"]); out["max_ns"] = float(r["MaxNs"])
for key, pat in (("FETCH_SIZE", "fetch/*/*_counter_collection.csv"), ("WRITE_SIZE", "write/*/*_counter_collection.csv")):
    f = one(pat)
    if not f:
        continue
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == key]
    if vals:
        vals = vals[-NTIMED:]                    # the timed launches (settle + warm-up come first)
        out[key + "_KiB_per_launch_raw"] = sum(vals) / len(vals)
for name in ("trace1q", "trace2q", "fetch", "write"):
    p = os.path.join(src, f"{name}_bench.json")
    if os.path.exists(p):
        try:
            out[f"bench_line_{name}"] = json.loads(open(p).read().strip().splitlines()[-1])
        except Exception:
            pass
if "FETCH_SIZE_KiB_per_launch_raw" in out and "WRITE_SIZE_KiB_per_launch_raw" in out:
    rd = out["FETCH_SIZE_KiB_per_launch_raw"] * 1024 * 2        # gfx950 correction: x2 (see docstring)
    wr = out["WRITE_SIZE_KiB_per_launch_raw"] * 1024
    out["hbm_read_bytes_per_launch"] = rd
    out["hbm_write_bytes_per_launch"] = wr
    out["hbm_bytes_per_launch"] = rd + wr
    out["algorithmic_bytes_per_launch"] = 16 * (1 << 24)
    out["traffic_over_algorithmic"] = (rd + wr) / (16 * (1 << 24))
# per-dispatch durations, one queue: the last NTIMED dispatches are the timed region
tr = one("trace1q/*/*_kernel_trace.csv")
if tr:
    rows = [r for r in csv.DictReader(open(tr)) if KERNEL in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if rows:
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows][-NTIMED:]
        out["one_queue_timed_launches_avg_us"] = sum(durs) / len(durs)
        out["one_queue_timed_launches_min_us"] = min(durs)
        out["one_queue_timed_launches_max_us"] = max(durs)
# two queues: dispatches overlap, so the per-dispatch duration is no longer the step time; the step time is the
# start-to-start (= end-to-end) interval of consecutive dispatches
tr = one("trace2q/*/*_kernel_trace.csv")
if tr:
    rows = [r for r in csv.DictReader(open(tr)) if KERNEL in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-NTIMED:]
    if len(rows) > 10:
        st_ = [int(r["Start_Timestamp"]) for r in rows]; en = sorted(int(r["End_Timestamp"]) for r in rows)
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        out["two_queues"] = {
            "dispatches": len(rows),
            "per_dispatch_duration_avg_us": sum(durs) / len(durs),
            "start_to_start_avg_us": (st_[-1] - st_[0]) / 1e3 / (len(rows) - 1),
            "end_to_end_avg_us": (en[-1] - en[0]) / 1e3 / (len(rows) - 1),
            "queues_seen": sorted({r.get("Queue_Id", "?") for r in rows}),
            "avg_dispatches_in_flight": sum(durs) * 1e3 / max(en[-1] - st_[0], 1),
            "note": "two dispatches in flight: each one's own duration spans the tail of its predecessor and the "
                    "head of its successor, so durations sum to more than the wall time; throughput = one launch "
                    "per end_to_end interval"}
json.dump(out, open(f"profiles/{tag}_summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("bench_line")}, indent=1))
