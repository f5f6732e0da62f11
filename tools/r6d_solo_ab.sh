#!/bin/bash
# Same-box A/B of the resident tail's single-bidder chain (emd_resident.h, MVP_RES_SOLO): the shipped library against
# a build with -DMVP_RES_SOLO=0 (make -C mvp_benchmark_amd/csrc variant NAME=nosolo DEFS=-DMVP_RES_SOLO=0 FILES="emd_lean.hip emd_resident.hip").
# Two passes, the order of the two libraries swapped in the second one.
out=gpurun_out/r6d_solo_ab.txt; mkdir -p gpurun_out; : > $out
export MVP_BENCH_REPS=5
one() {  # lib-name n [shape]
  lib=mvp_benchmark_amd/libmvpops_$1.so; [ $1 = default ] && lib=mvp_benchmark_amd/libmvpops.so
  echo "$1: $(MVP_BENCH_SHAPE=$3 python tools/bench_emd_one.py 64 $2 0.004 3000 $lib 2>&1 | tail -1) ${3:-uniform}" >> $out
}
for order in "nosolo default" "default nosolo"; do
  for n in 1024 2048 4096; do for shape in "" chair:0.03 chair:0.01 sphere:0.03; do for v in $order; do one $v $n $shape; done; done; done
done
cat $out
