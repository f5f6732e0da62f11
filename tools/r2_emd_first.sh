#!/bin/bash
# first GPU check of the EMD tail kernel: parity tests, then timings per variant
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "emd" > gpurun_out/r2a_emd_tests.txt 2>&1
tail -5 gpurun_out/r2a_emd_tests.txt
{
for v in "MVP_EMD_TAIL=0" "MVP_EMD_TAIL=1 MVP_EMD_TAIL_DELTA=0" "MVP_EMD_TAIL=1 MVP_EMD_TAIL_DELTA=2" "MVP_EMD_TAIL=1 MVP_EMD_TAIL_DELTA=5" "MVP_EMD_TAIL=1 MVP_EMD_TAIL_DELTA=10"; do
  echo "== $v"
  env $v timeout 300 python tools/bench_emd_one.py 64 16384 0.004 3000
done
for n in 1024 2048 4096 8192; do timeout 300 python tools/bench_emd_one.py 64 $n 0.004 3000; done
timeout 300 python tools/bench_emd_one.py 32 16384 0.004 3000
timeout 300 python tools/bench_emd_one.py 64 16384 0.005 50
} > gpurun_out/r2a_emd_bench.txt 2>&1
cat gpurun_out/r2a_emd_bench.txt
