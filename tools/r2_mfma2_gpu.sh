#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_harness.py -x -q -m gpu -k "pointwise" > gpurun_out/r2j_mfma_tests.txt 2>&1
tail -8 gpurun_out/r2j_mfma_tests.txt
timeout 600 python tools/bench_pointwise_mfma.py > gpurun_out/r2j_bench_pointwise_mfma.txt 2>&1
cat gpurun_out/r2j_bench_pointwise_mfma.txt
timeout 600 python tools/bench_models.py > gpurun_out/r2j_bench_models.txt 2>&1; cat gpurun_out/r2j_bench_models.txt
