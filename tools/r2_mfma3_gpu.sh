#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_harness.py -x -q -m gpu > gpurun_out/r2l_harness_tests.txt 2>&1
tail -4 gpurun_out/r2l_harness_tests.txt
timeout 600 python tools/bench_models.py > gpurun_out/r2l_bench_models.txt 2>&1; cat gpurun_out/r2l_bench_models.txt
