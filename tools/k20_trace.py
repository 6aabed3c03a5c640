#!/usr/bin/env python3
"""Where the fixed cost of a K = 20 region goes: kernel dispatches and HIP runtime calls of one `bench.py --steps 20
--warmup 5` run on ONE time base (rocprofv3 --kernel-trace --hip-runtime-trace, csv).
   rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d gpurun_out/prof_k20 -- \
       python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-queue-leg
   python3 tools/k20_trace.py gpurun_out/prof_k20
The timed region is found as the 20 fmi_kernel dispatches behind the longest launch-free gap that follows the settle
launches (the barrier in front of the region)."""
import csv, glob, os, sys
d = sys.argv[1]
kt = sorted(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
ht = sorted(glob.glob(os.path.join(d, "*", "*hip_api_trace.csv")), key=os.path.getmtime)[-1]
K = [r for r in csv.DictReader(open(kt)) if "fmi_kernel" in r["Kernel_Name"]]
K.sort(key=lambda r: int(r["Start_Timestamp"]))
st = [int(r["Start_Timestamp"]) for r in K]; en = [int(r["End_Timestamp"]) for r in K]
H = sorted(csv.DictReader(open(ht)), key=lambda r: int(r["Start_Timestamp"]))
launch = [r for r in H if "LaunchKernel" in r["Function"]]
# regions: groups of dispatches separated by an idle gap of > 30 us between one's end and the next one's start
groups, cur = [], [0]
for i in range(1, len(K)):
    if st[i] - max(en[:i][-3:]) > 30000: groups.append(cur); cur = []
    cur.append(i)
groups.append(cur)
print("dispatch groups (count):", [len(g) for g in groups][:12], "...")
for g in groups:
    if len(g) != 20: continue
    a, b = g[0], g[-1]
    t_first, t_last_end = st[a], max(en[i] for i in g)
    # the launch calls of this region: the 20 hipLaunchKernel calls that precede the dispatches
    ls = [r for r in launch if int(r["Start_Timestamp"]) < t_first + 2_000_000 and int(r["Start_Timestamp"]) > en[a - 1]]
    l0 = int(ls[0]["Start_Timestamp"]) if ls else None
    after = [r for r in H if int(r["End_Timestamp"]) > t_last_end][:4]
    print(f"region of 20 dispatches: first launch call -> first kernel start {(t_first - l0) / 1e3 if l0 else float('nan'):7.1f} us")
    print(f"   first kernel start -> last kernel end {(t_last_end - t_first) / 1e3:8.1f} us   (20 x steady step would be {20 * 46.85:.0f})")
    print("   start-to-start intervals (us):", " ".join(f"{(st[i + 1] - st[i]) / 1e3:.1f}" for i in g[:-1]))
    print("   durations (us):              ", " ".join(f"{(en[i] - st[i]) / 1e3:.1f}" for i in g))
    print("   last kernel end -> return of the call that sees it:")
    for r in after:
        print(f"      {r['Function']:28s} start {(int(r['Start_Timestamp']) - t_last_end) / 1e3:8.1f} us  end {(int(r['End_Timestamp']) - t_last_end) / 1e3:8.1f} us after the last kernel's end")
    if l0:
        print(f"   launch calls: the 20 take {(int(ls[min(19, len(ls) - 1)]['End_Timestamp']) - l0) / 1e3:.1f} us of host time")
