#!/bin/bash
# Same-box A/B of the list positions per wave of the few-bidders rounds (MVP_EMD_FEWPOS 1 / 2 / 3: the cluster collapses
# to member 0 at 16 / 32 / 48 persons) against the build without those rounds: result digests first, then times.
out=gpurun_out/r6d_fewpos_ab.txt; mkdir -p gpurun_out; : > $out
libof() { [ $1 = default ] && echo mvp_benchmark_amd/libmvpops.so || echo mvp_benchmark_amd/libmvpops_$1.so; }
for v in "$@"; do python tools/emd_variant_hash.py $(libof $v) 2>&1 | grep -E "headline|uniform|dups|blob|forced" | cut -c1-100 > gpurun_out/hash_$v.txt; done
echo "digests (count = libraries that agree):" >> $out; cat gpurun_out/hash_*.txt | sort | uniq -c >> $out
export MVP_BENCH_REPS=4
one() {  # lib-name b n [shape]
  echo "$1: $(MVP_BENCH_SHAPE=$4 python tools/bench_emd_one.py $2 $3 0.004 3000 $(libof $1) 2>&1 | tail -1) ${4:-uniform}" >> $out
}
for pass in 1 2; do
  for v in "$@"; do one $v 64 16384; done
  for shape in chair:0.03 chair:0.01 sphere:0.03; do for v in "$@"; do one $v 64 16384 $shape; done; done
  for v in "$@"; do one $v 32 16384; done
  for shape in "" chair:0.03; do for v in "$@"; do one $v 64 8192 $shape; done; done
done
cat $out
