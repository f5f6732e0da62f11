#!/bin/bash
# Round-6 artefacts at HEAD, one gpurun call: the -m gpu suite, the default bench (+ under rocprofv3 with the kernel
# trace), PMC passes of the EMD kernels, the other workloads and the op benches.  Outputs: gpurun_out/r6z/*.
set -u
out=gpurun_out/r6z
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- \
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side > $out/bench_under_rocprof.json 2> $out/trace.err
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats.csv \;
for P in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  name=$(echo $P | cut -d" " -f1)
  timeout 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/pmc_$name -o run -- \
    python tools/run_op.py emd 1 > /dev/null 2> $out/pmc_$name.err
  timeout 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/pmcs_$name -o run -- \
    python tools/run_op.py emd 1 64 1024 > /dev/null 2>> $out/pmc_$name.err
done
for k in emd_auction_kernel emd_lean_kernel emd_lean_tiers_kernel; do
  python tools/pmc_summary.py $out $k > $out/pmc_$k.json
done
python tools/pmc_summary.py $out emd_resident_kernel pmcs_ > $out/pmc_emd_resident_kernel_n1024.json
timeout 600 python bench.py --workload pcn_eval --steps 20 --warmup 3 > $out/bench_pcn_eval.json 2>> $out/bench.err
timeout 600 python bench.py --workload vrcnet_train --steps 20 --warmup 3 > $out/bench_vrcnet_train.json 2>> $out/bench.err
{ for n in 1024 2048 4096 8192 16384; do MVP_BENCH_REPS=5 timeout 300 python tools/bench_emd_one.py 64 $n 0.004 3000; done
  MVP_BENCH_REPS=5 timeout 300 python tools/bench_emd_one.py 32 16384 0.004 3000
  MVP_BENCH_REPS=5 timeout 300 python tools/bench_emd_one.py 64 16384 0.005 50
  MVP_BENCH_REPS=5 timeout 300 python tools/bench_emd_one.py 64 2048 0.005 50; } > $out/bench_emd_sweep.txt 2>&1
timeout 600 python tools/bench_models.py > $out/bench_models.txt 2>&1
timeout 900 python tools/emd_surfaces.py 64 16384 > $out/emd_surfaces_16384.txt 2>&1
timeout 600 python tools/emd_surfaces.py 64 2048 > $out/emd_surfaces_2048.txt 2>&1
rm -rf $out/trace $out/pmc_*/ $out/pmcs_*/
ls $out
