#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV passes for one kernel.
Usage: python tools/pmc_summary.py <dir with pmc_*/run_counter_collection.csv> <kernel substring> [prefix]"""
import collections
import csv
import glob
import json
import os
import sys

root, kern = sys.argv[1], sys.argv[2]
prefix = sys.argv[3] if len(sys.argv) > 3 else "pmc_"
out = {"kernel": kern, "counters": {}, "duration_ms": []}
for d in sorted(glob.glob(os.path.join(root, prefix + "*"))):
    if not os.path.isdir(d) or (prefix == "pmc_" and os.path.basename(d).startswith("pmc_cd_")):
        continue
    f = os.path.join(d, "run_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg, launches = collections.defaultdict(float), set()
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            launches.add(r["Dispatch_Id"])
    for k, v in agg.items():
        out["counters"][k] = v / max(1, len(launches))
    for r in csv.DictReader(open(os.path.join(d, "run_kernel_trace.csv"))):
        if kern in r["Kernel_Name"]:
            out["duration_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print(json.dumps(out, indent=1))
