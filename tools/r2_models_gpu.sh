#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_harness.py -x -q -m gpu > gpurun_out/r2h_harness_tests.txt 2>&1
tail -6 gpurun_out/r2h_harness_tests.txt
timeout 600 python tools/bench_models.py > gpurun_out/r2h_bench_models.txt 2>&1; cat gpurun_out/r2h_bench_models.txt
timeout 600 python - > gpurun_out/r2h_bench_models_nomfma.txt 2>&1 <<'PY'
import sys, runpy
sys.path.insert(0, ".")
import mvp_benchmark_amd.pointwise as pw
pw.USE_MFMA = False
sys.argv = ["tools/bench_models.py"]
runpy.run_path("tools/bench_models.py", run_name="__main__")
PY
cat gpurun_out/r2h_bench_models_nomfma.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ddp.py -x -q -m gpu -k "scatter or cfg3" > gpurun_out/r2h_misc_tests.txt 2>&1
tail -4 gpurun_out/r2h_misc_tests.txt
