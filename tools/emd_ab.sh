#!/bin/bash
# usage: ab.sh outdir lib1 lib2 ...   (names without libmvpops_ prefix; "default" = libmvpops.so)
out=gpurun_out/$1; shift; mkdir -p $out
export MVP_BENCH_REPS=6
for v in "$@"; do
  lib=mvp_benchmark_amd/libmvpops_$v.so; [ $v = default ] && lib=mvp_benchmark_amd/libmvpops.so
  python tools/emd_variant_hash.py $lib 2>&1 | grep -E "headline|uniform|dups|blob|forced" | cut -c1-100 > $out/hash_$v.txt
done
for rep in 1 2; do for v in "$@"; do
  lib=mvp_benchmark_amd/libmvpops_$v.so; [ $v = default ] && lib=mvp_benchmark_amd/libmvpops.so
  echo "$v: $(python tools/bench_emd_one.py 64 16384 0.004 3000 $lib 2>&1 | tail -1)" >> $out/time.txt
done; done
cat $out/hash_*.txt | sort | uniq -c; cat $out/time.txt
