#!/bin/bash
# Round-4 artefacts that depend on the pointwise / model code (the EMD sources and their PMC passes are unchanged since
# tools/r4_artifacts.sh ran): the -m gpu suite, smoke, the default bench, the other workloads, model steps.  Outputs: gpurun_out/r4y/*.
set -u
out=gpurun_out/r4y
mkdir -p $out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > $out/gpu_tests.txt 2>&1; tail -4 $out/gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- \
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side > $out/bench_under_rocprof.json 2> $out/trace.err
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats.csv \;
timeout 600 python bench.py --workload pcn_eval --steps 20 --warmup 3 > $out/bench_pcn_eval.json 2>> $out/bench.err
timeout 600 python bench.py --workload vrcnet_train --steps 20 --warmup 3 > $out/bench_vrcnet_train.json 2>> $out/bench.err
timeout 600 python tools/bench_models.py > $out/bench_models.txt 2>&1
MVP_MFMA_TRAIN=0 timeout 600 python tools/bench_models.py >> $out/bench_models.txt 2>&1
timeout 300 python tools/profile_kernels.py vrcnet ecg > $out/vrcnet_kernels.txt 2>&1
timeout 300 python tools/bench_conv_passes.py vrcnet > $out/conv_passes_vrcnet.txt 2>&1
timeout 300 python tools/bench_conv_passes.py ecg > $out/conv_passes_ecg.txt 2>&1
timeout 300 python tools/bench_conv_passes.py vrcnet 1 > $out/conv_passes_vrcnet_skinny.txt 2>&1
rm -rf $out/trace
ls $out
