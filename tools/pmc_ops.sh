#!/bin/bash
# HBM-side traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of the op-level benchmarks.
# Usage: tools/pmc_ops.sh <tag> <bench_ops section ...>     (outputs gpurun_out/<tag>/)
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for sec in "$@"; do
  for P in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/${sec}_$P -o run -- \
      python tools/bench_ops.py $sec > /dev/null 2> $out/${sec}_$P.err
  done
done
python - "$out" "$@" <<'PY'
import collections, csv, glob, json, os, sys
out, secs = sys.argv[1], sys.argv[2:]
res = {}
for sec in secs:
    per = collections.defaultdict(lambda: {"n": 0})
    for P in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(os.path.join(out, "%s_%s" % (sec, P), "**", "run_counter_collection.csv"), recursive=True)
        if not f:
            continue
        agg, cnt = collections.defaultdict(float), collections.defaultdict(set)
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == P and "mvp::" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[k] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
        for k in agg:
            per[k][P + "_KB_per_launch"] = agg[k] / len(cnt[k]); per[k]["n"] = len(cnt[k])
        t = glob.glob(os.path.join(out, "%s_%s" % (sec, P), "**", "run_kernel_trace.csv"), recursive=True)
        if t:
            dur = collections.defaultdict(list)
            for r in csv.DictReader(open(t[0])):
                if "mvp::" in r["Kernel_Name"]:
                    dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            for k, v in dur.items():
                per[k]["avg_us"] = sum(v) / len(v)
    res[sec] = per
json.dump(res, open(os.path.join(out, "pmc_ops.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
