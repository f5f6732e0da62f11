out=gpurun_out/r6d_prefew_ab.txt; : > $out
libof() { [ $1 = default ] && echo mvp_benchmark_amd/libmvpops.so || echo mvp_benchmark_amd/libmvpops_$1.so; }
for v in prefew default; do python tools/emd_variant_hash.py $(libof $v) 2>&1 | grep -E "headline|uniform|dups|blob|forced" | cut -c1-100 > gpurun_out/hash2_$v.txt; done
echo "digests (count = libraries that agree):" >> $out; cat gpurun_out/hash2_*.txt | sort | uniq -c >> $out
export MVP_BENCH_REPS=6
for pass in 1 2 3; do for v in prefew default; do
  echo "$v: $(python tools/bench_emd_one.py 64 16384 0.004 3000 $(libof $v) 2>&1 | tail -1)" >> $out
done; done
for v in prefew default; do echo "$v: $(python tools/bench_emd_one.py 64 8192 0.004 3000 $(libof $v) 2>&1 | tail -1)" >> $out; echo "$v: $(python tools/bench_emd_one.py 64 16384 0.005 50 $(libof $v) 2>&1 | tail -1)" >> $out; done
cat $out
