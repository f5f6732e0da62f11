#!/bin/bash
# Round 6, second half: artefacts at HEAD in one gpurun call (the EMD sources are those of tools/r6_artifacts.sh's PMC passes:
# profiles/traffic.json stays current).  Outputs: gpurun_out/r6b/*.
set -u
out=gpurun_out/r6b
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -c 200 $out/bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- \
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side > $out/bench_under_rocprof.json 2> $out/trace.err
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats.csv \; ; rm -rf $out/trace
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o vrc -- \
  python bench.py --workload vrcnet_train --steps 10 --warmup 3 > $out/vrcnet_under_rocprof.json 2>> $out/trace.err
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/vrcnet_kernel_stats.csv \; ; rm -rf $out/trace
timeout 600 python bench.py --workload vrcnet_train --steps 20 --warmup 3 > $out/bench_vrcnet_train.json 2>> $out/bench.err
timeout 600 python bench.py --workload pcn_eval --steps 20 --warmup 3 > $out/bench_pcn_eval.json 2>> $out/bench.err
{ MVP_BENCH_REPS=10 timeout 400 python tools/ab_vrcnet.py singleton_sk=0,fused_activations=0 singleton_sk=1,fused_activations=1 vrcnet 6
  MVP_BENCH_REPS=10 timeout 400 python tools/ab_vrcnet.py singleton_sk=0 singleton_sk=1 vrcnet 5
  MVP_BENCH_REPS=10 timeout 400 python tools/ab_vrcnet.py fused_activations=0 fused_activations=1 vrcnet 5; } > $out/vrcnet_ab.txt 2>&1
timeout 600 python tools/bench_models.py > $out/bench_models.txt 2>&1
timeout 300 python tools/bench_conv_passes.py > $out/conv_passes_vrcnet.txt 2>&1
timeout 300 python tools/profile_kernels.py vrcnet > $out/vrcnet_kernels.txt 2>&1
timeout 300 python tools/dispatch_sites.py vrcnet > $out/vrcnet_dispatch_sites.txt 2>&1
timeout 600 python tools/fuzz_pointwise.py 300 1 ex > $out/fuzz_pointwise_ex.txt 2>&1
ls $out
