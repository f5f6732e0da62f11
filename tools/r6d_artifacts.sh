#!/bin/bash
# Round 6, last session (few-bidders rounds + single-bidder chain: EMD sources changed): the whole GPU suite, the PMC passes of the EMD kernels
# (-> profiles/traffic.json via tools/make_traffic.py), the default bench + its kernel trace, the EMD sweep and surfaces.
# One gpurun call; outputs gpurun_out/r6d/*.
# Two stages (the bench line reads profiles/traffic.json, which stage 1's counters produce: tools/make_traffic.py in between):
#   tools/r6d_artifacts.sh pmc | tools/r6d_artifacts.sh bench
set -u
out=gpurun_out/r6d
mkdir -p $out
export TMPDIR=/tmp
stage=${1:-pmc}
if [ $stage = pmc ]; then
timeout 1500 python -m pytest tests -q -m gpu > $out/gpu_tests.txt 2>&1; tail -2 $out/gpu_tests.txt
for P in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  name=$(echo $P | cut -d" " -f1)
  timeout 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/pmc_$name -o run -- \
    python tools/run_op.py emd 1 > /dev/null 2> $out/pmc_$name.err
done
for k in emd_auction_kernel emd_lean_kernel emd_lean_tiers_kernel; do
  python tools/pmc_summary.py $out $k > $out/pmc_$k.json
done
rm -rf $out/pmc_*/
ls $out; exit 0
fi
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -c 200 $out/bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- \
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side > $out/bench_under_rocprof.json 2> $out/trace.err
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats.csv \; ; rm -rf $out/trace
{ for n in 1024 2048 4096 8192 16384; do MVP_BENCH_REPS=5 timeout 300 python tools/bench_emd_one.py 64 $n 0.004 3000; done
  MVP_BENCH_REPS=5 timeout 300 python tools/bench_emd_one.py 32 16384 0.004 3000
  MVP_BENCH_REPS=5 timeout 300 python tools/bench_emd_one.py 64 16384 0.005 50
  MVP_BENCH_REPS=5 timeout 300 python tools/bench_emd_one.py 64 2048 0.005 50; } > $out/bench_emd_sweep.txt 2>&1
timeout 900 python tools/emd_surfaces.py 64 16384 > $out/emd_surfaces_16384.txt 2>&1
timeout 600 python tools/emd_surfaces.py 64 2048 > $out/emd_surfaces_2048.txt 2>&1
timeout 600 python bench.py --workload pcn_eval --steps 20 --warmup 3 > $out/bench_pcn_eval.json 2>> $out/bench.err
ls $out
