#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "emd" > gpurun_out/r2c_emd_tests.txt 2>&1
tail -5 gpurun_out/r2c_emd_tests.txt
{
for v in "MVP_EMD_TAIL=0" "MVP_EMD_TAIL_CLUSTER=1" "MVP_EMD_TAIL_CLUSTER=2" "MVP_EMD_TAIL_CLUSTER=4" "MVP_EMD_TAIL_CLUSTER=4 MVP_EMD_TAIL_DELTA=0" "MVP_EMD_TAIL_CLUSTER=4 MVP_EMD_TAIL_DELTA=3" "MVP_EMD_TAIL_CLUSTER=4 MVP_EMD_TAIL_DELTA=8"; do
  echo "== $v"
  env $v timeout 300 python tools/bench_emd_one.py 64 16384 0.004 3000 2>&1 | grep -v amdgpu.ids
done
echo "== prof W4"
timeout 300 python tools/bench_emd_one.py 64 16384 0.004 3000 mvp_benchmark_amd/libmvpops_prof.so 2>&1 | grep -v amdgpu.ids
for n in 1024 2048 4096 8192; do timeout 300 python tools/bench_emd_one.py 64 $n 0.004 3000 2>&1 | grep -v amdgpu.ids; done
timeout 300 python tools/bench_emd_one.py 32 16384 0.004 3000 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r2c_emd_bench.txt 2>&1
cat gpurun_out/r2c_emd_bench.txt
