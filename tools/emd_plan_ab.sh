#!/bin/bash
# A/B of the lean kernel's tiered cluster widths (env knobs), default library:  emd_plan_ab.sh outdir "round every heavy" ...
out=gpurun_out/$1; shift; mkdir -p $out
export MVP_BENCH_REPS=6
python tools/emd_variant_hash.py 2>&1 | grep -E "headline|uniform|dups|blob|forced" | cut -c1-100 > $out/hash_plan.txt
MVP_EMD_SPLIT=1 python tools/emd_variant_hash.py 2>&1 | grep -E "headline|uniform|dups|blob|forced" | cut -c1-100 > $out/hash_fixed.txt
cat $out/hash_*.txt | sort | uniq -c
for rep in 1 2; do
  echo "split=1: $(MVP_EMD_SPLIT=1 python tools/bench_emd_one.py 64 16384 0.004 3000 2>&1 | tail -1)"
  for cfg in "$@"; do read r e h <<< "$cfg"
    echo "round=$r every=$e heavy=$h: $(MVP_EMD_PLAN_ROUND=$r MVP_EMD_PLAN_EVERY=$e MVP_EMD_PLAN_HEAVY=$h python tools/bench_emd_one.py 64 16384 0.004 3000 2>&1 | tail -1)"
  done
done | tee $out/time.txt
