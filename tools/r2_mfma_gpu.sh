#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_harness.py -x -q -m gpu -k "pointwise or uniform_and_repulsion_loss_values" > gpurun_out/r2g_mfma_tests.txt 2>&1
tail -15 gpurun_out/r2g_mfma_tests.txt
timeout 600 python tools/bench_pointwise_mfma.py > gpurun_out/r2g_bench_pointwise_mfma.txt 2>&1
cat gpurun_out/r2g_bench_pointwise_mfma.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ddp.py -x -q -m gpu -k "scatter or sampler or query_and_group or cfg3 or chamfer_sorted or fps_mid" > gpurun_out/r2g_misc_tests.txt 2>&1
tail -8 gpurun_out/r2g_misc_tests.txt
timeout 600 python tools/bench_ops.py grad > gpurun_out/r2g_bench_ops_grad.txt 2>&1; cat gpurun_out/r2g_bench_ops_grad.txt
