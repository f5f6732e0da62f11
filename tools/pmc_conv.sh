#!/bin/bash
# PMC passes over the three 1x1-convolution kernels at one shape.  Usage: tools/pmc_conv.sh <tag> B Cin Cout L
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/pmc_$i -o run -- python tools/run_conv_pass.py "$@" 2 > $out/pmc_$i.log 2>&1
  f=$(find $out/pmc_$i -name run_counter_collection.csv | head -1)
  [ -n "$f" ] && cp $f $out/pmc_$i/run_counter_collection.csv.flat 2>/dev/null
done
python - "$out" <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d): continue
    f = glob.glob(os.path.join(d, "**", "run_counter_collection.csv"), recursive=True)
    t = glob.glob(os.path.join(d, "**", "run_kernel_trace.csv"), recursive=True)
    if not f: continue
    agg, cnt = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        if "pointwise" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void mvp::", "")
            agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
    for (k, c), v in agg.items():
        res[k][c] = v / len(cnt[(k, c)])
    if t:
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(t[0])):
            if "pointwise" in r["Kernel_Name"]:
                dur[r["Kernel_Name"].split("(")[0].replace("void mvp::", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in dur.items():
            res[k]["avg_us"] = sum(v) / len(v)
json.dump(res, open(os.path.join(out, "pmc_conv.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
