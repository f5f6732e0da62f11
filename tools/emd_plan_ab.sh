#!/bin/bash
# (the MVP_EMD_* knobs are read by libmvpops_hooks.so only: make -C mvp_benchmark_amd/csrc hooks)
# A/B of the tiered launch's width patterns (env knob), default library:  emd_plan_ab.sh outdir "8,4,4,4,4,4,2,2" ...
out=gpurun_out/$1; shift; mkdir -p $out
export MVP_BENCH_REPS=6
for cfg in "$@"; do
  MVP_EMD_PLAN_WIDTHS=$cfg python tools/emd_variant_hash.py 2>&1 | grep -E "headline" | cut -c1-60
done | sort | uniq -c
for rep in 1 2; do
  echo "split=1: $(MVP_EMD_SPLIT=1 python tools/bench_emd_one.py 64 16384 0.004 3000 mvp_benchmark_amd/libmvpops_hooks.so 2>&1 | tail -1)"
  for cfg in "$@"; do
    echo "widths=$cfg: $(MVP_EMD_PLAN_WIDTHS=$cfg python tools/bench_emd_one.py 64 16384 0.004 3000 mvp_benchmark_amd/libmvpops_hooks.so 2>&1 | tail -1)"
  done
done | tee $out/time.txt
