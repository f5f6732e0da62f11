#!/bin/bash
# (the MVP_EMD_* knobs are read by libmvpops_hooks.so only: make -C mvp_benchmark_amd/csrc hooks)
export MVP_BENCH_REPS=6
for pr in 150 200 250 300 400; do
  echo "PLAN_ROUND=$pr: $(MVP_EMD_PLAN_ROUND=$pr python tools/bench_emd_one.py 64 16384 0.004 3000 mvp_benchmark_amd/libmvpops_hooks.so 2>&1 | tail -1)"
done
for pe in 2000 1400 1000 700; do
  echo "PLAN_EVERY=$pe: $(MVP_EMD_PLAN_EVERY=$pe python tools/bench_emd_one.py 64 16384 0.004 3000 mvp_benchmark_amd/libmvpops_hooks.so 2>&1 | tail -1)"
done
echo "default again: $(python tools/bench_emd_one.py 64 16384 0.004 3000 2>&1 | tail -1)"
