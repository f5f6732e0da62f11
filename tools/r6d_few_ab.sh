#!/bin/bash
# Same-box A/B of the rounds of at most 16 bidders (emd_lean_round_few.inc, MVP_EMD_FEW): the shipped library against
# make -C mvp_benchmark_amd/csrc variant NAME=nofew DEFS=-DMVP_EMD_FEW=0 FILES=emd_lean.hip.  Two passes, order swapped.
out=gpurun_out/r6d_few_ab.txt; mkdir -p gpurun_out; : > $out
export MVP_BENCH_REPS=4
one() {  # lib-name b n [shape]
  lib=mvp_benchmark_amd/libmvpops_$1.so; [ $1 = default ] && lib=mvp_benchmark_amd/libmvpops.so
  echo "$1: $(MVP_BENCH_SHAPE=$4 python tools/bench_emd_one.py $2 $3 0.004 3000 $lib 2>&1 | tail -1) ${4:-uniform}" >> $out
}
for order in "nofew default" "default nofew"; do
  for v in $order; do one $v 64 16384; done
  for shape in chair:0.03 chair:0.01 sphere:0.03 chair:indep; do for v in $order; do one $v 64 16384 $shape; done; done
  for v in $order; do one $v 32 16384 chair:0.03; done
  for shape in "" chair:0.03; do for v in $order; do one $v 64 8192 $shape; done; done
done
cat $out
