#!/usr/bin/env python3
"""Per-launch averages of the counters collected by tools/sq_run.sh for the fused FIR kernel (fmi_kernel dispatches
only; the first 40 dispatches -- correctness checks and settle -- are skipped).  Prints one JSON object."""
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for f in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    acc, cnt = {}, {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "fmi_kernel" not in row.get("Kernel_Name", ""):
                continue
            name = row["Counter_Name"]; did = int(row["Dispatch_Id"])
            if did < 40:
                continue
            acc[name] = acc.get(name, 0.0) + float(row["Counter_Value"]); cnt[name] = cnt.get(name, 0) + 1
    for k in acc:
        res[k] = acc[k] / cnt[k]
print(json.dumps(res, indent=1, sort_keys=True))
