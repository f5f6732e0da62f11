#!/bin/bash
# Round 4: GPU check of the LDS-resident EMD tail (parity tests, then timings against split = 2, then the profile build).
tag=${1:-r4a}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_emd_resident.py -x -q > gpurun_out/${tag}_resident_tests.txt 2>&1
tail -5 gpurun_out/${tag}_resident_tests.txt
{
for n in 1024 2048 4096; do
  for split in 2 3; do
    echo "## split=$split"
    MVP_EMD_SPLIT=$split MVP_BENCH_REPS=5 timeout 300 python tools/bench_emd_one.py 64 $n 0.004 3000 mvp_benchmark_amd/libmvpops_hooks.so   # (the release library reads no MVP_EMD_* variable)
  done
done
for cap in 8 16 32; do
  echo "## cap=$cap"
  for n in 1024 2048 4096; do
    MVP_EMD_RESIDENT_CAP=$cap MVP_BENCH_REPS=5 timeout 300 python tools/bench_emd_one.py 64 $n 0.004 3000 mvp_benchmark_amd/libmvpops_hooks.so   # (the release library reads no MVP_EMD_* variable)
  done
done
if [ -f mvp_benchmark_amd/libmvpops_prof.so ]; then
  for n in 1024 2048 4096; do
    echo "## profile build n=$n"
    timeout 300 python tools/bench_emd_one.py 64 $n 0.004 3000 mvp_benchmark_amd/libmvpops_prof.so 2>&1 | grep -E "resident|W="
  done
fi
} > gpurun_out/${tag}_resident_times.txt 2>&1
grep -v amdgpu.ids gpurun_out/${tag}_resident_times.txt
