#!/usr/bin/env python3
"""Mean per launch of arbitrary rocprofv3 --pmc counters for the step kernel.
usage: pmc_counters.py OUT.json DIR [DIR...] (each DIR: one `rocprofv3 --pmc A B C -d DIR -- python3
bench.py --steps 64 --warmup 0 --no-cpu-baseline --no-rollout` pass); the last 64 launches count.
PMC_KERNEL=<substring of the kernel name> selects the kernel (default: whichever step kernel ran)."""
import csv
import glob
import json
import os
import sys

out_path, dirs = sys.argv[1], sys.argv[2:]
tag, k = os.environ.get("PMC_KERNEL", ""), 64
TAGS = [tag] if tag else ["step_window1_kernel<24>", "step_pair_kernel<24>", "step_kernel64<24, false>"]
res = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                hit = [t for t in TAGS if t in row["Kernel_Name"]]
                if not hit:
                    continue
                tag = hit[0]
                key = (row["Counter_Name"], int(row["Dispatch_Id"]))
                acc[key] = acc.get(key, 0.0) + float(row["Counter_Value"])
        for name in sorted({n for n, _ in acc}):
            vals = [v for (n, i), v in sorted(acc.items(), key=lambda kv: kv[0][1]) if n == name][-k:]
            res[name] = sum(vals) / len(vals)
json.dump({"kernel": tag, "launches": k, "mean_per_launch": res}, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
