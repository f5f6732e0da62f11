#!/bin/bash
# registration + DDP + bench launcher checks on the GPU box
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_registration.py tests/test_gpu_ddp.py -x -q -m gpu > gpurun_out/r2e_misc_tests.txt 2>&1
tail -15 gpurun_out/r2e_misc_tests.txt
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r2e_bench.json 2> gpurun_out/r2e_bench.err
tail -c 3000 gpurun_out/r2e_bench.json; tail -3 gpurun_out/r2e_bench.err
timeout 600 python bench.py --workload vrcnet_train --steps 5 --warmup 2 > gpurun_out/r2e_bench_vrcnet.json 2>> gpurun_out/r2e_bench.err
cat gpurun_out/r2e_bench_vrcnet.json
