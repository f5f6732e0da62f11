#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "emd_tail or emd_headline" > gpurun_out/r2f_emd_tests.txt 2>&1
tail -3 gpurun_out/r2f_emd_tests.txt
{
for v in "MVP_EMD_TAIL=0" "MVP_EMD_TAIL=1 MVP_EMD_TAIL_CLUSTER=2 MVP_EMD_TAIL_DELTA=3" "MVP_EMD_TAIL=1 MVP_EMD_TAIL_CLUSTER=4 MVP_EMD_TAIL_DELTA=3" "MVP_EMD_TAIL=1 MVP_EMD_TAIL_CLUSTER=4 MVP_EMD_TAIL_DELTA=5" "MVP_EMD_TAIL=1 MVP_EMD_TAIL_CLUSTER=4 MVP_EMD_TAIL_DELTA=2"; do
  echo "== $v"
  env $v timeout 300 python tools/bench_emd_one.py 64 16384 0.004 3000 2>&1 | grep -v amdgpu.ids
done
echo "== prof W4 delta 3"
MVP_EMD_TAIL=1 MVP_EMD_TAIL_DELTA=3 timeout 300 python tools/bench_emd_one.py 64 16384 0.004 3000 mvp_benchmark_amd/libmvpops_prof.so 2>&1 | grep -v amdgpu.ids | grep -E "cloud 0|W="
} > gpurun_out/r2f_emd_bench.txt 2>&1
cat gpurun_out/r2f_emd_bench.txt
timeout 900 python -m pytest tests/test_registration.py tests/test_gpu_ddp.py -x -q -m gpu > gpurun_out/r2f_misc_tests.txt 2>&1
tail -15 gpurun_out/r2f_misc_tests.txt
