#!/bin/bash
# PMC passes (separate rocprofv3 runs, --kernel-trace only beside --pmc) of ONE auction on a realistic pair: 32 clouds of 16384
# points, prediction = chair-like ground truth + noise 0.03 (cfg 2's EMD: the pcn_eval step's).  -> gpurun_out/r6d_real/pmc_*.json
set -u
out=gpurun_out/r6d_real; mkdir -p $out; export TMPDIR=/tmp
for P in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  name=$(echo $P | cut -d" " -f1)
  MVP_BENCH_REPS=2 MVP_BENCH_SHAPE=chair:0.03 timeout 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/pmc_$name -o run -- \
    python tools/bench_emd_one.py 32 16384 0.004 3000 > /dev/null 2> $out/pmc_$name.err
done
for k in emd_auction_kernel emd_lean_kernel; do python tools/pmc_summary.py $out $k > $out/pmc_$k.json; done
rm -rf $out/pmc_*/
ls $out
