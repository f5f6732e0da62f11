#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/bench_pointwise_mfma.py 2>&1 | head -14 > gpurun_out/r2m_bk16.txt; cat gpurun_out/r2m_bk16.txt
MVP_LIB=mvp_benchmark_amd/libmvpops_bk32.so timeout 600 python tools/bench_pointwise_mfma.py 2>&1 | head -14 > gpurun_out/r2m_bk32.txt; cat gpurun_out/r2m_bk32.txt
