#!/bin/bash
# cycle counters of the EMD tail kernel (profiling build), per cache width
mkdir -p gpurun_out
{
for d in 0 5 20; do
  echo "== delta $d"
  MVP_EMD_TAIL_DELTA=$d timeout 300 python tools/bench_emd_one.py 64 16384 0.004 3000 mvp_benchmark_amd/libmvpops_prof.so 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r2b_emd_prof.txt 2>&1
cat gpurun_out/r2b_emd_prof.txt
