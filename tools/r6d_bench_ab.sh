#!/bin/bash
# The default bench line's timed region (the driver's clouds) on two libraries alternately, same box:
# mvp_benchmark_amd/libmvpops_prefew.so (built from commit a32dde1: before the few-bidders rounds) against the shipped one.
out=gpurun_out/r6d_bench_ab.txt; mkdir -p gpurun_out; : > $out
L=mvp_benchmark_amd/libmvpops.so
cp $L /tmp/shipped.so
for pass in 1 2 3; do for v in prefew shipped; do
  if [ $v = prefew ]; then cp mvp_benchmark_amd/libmvpops_prefew.so $L; else cp /tmp/shipped.so $L; fi
  python bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-side 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v: step %.3f ms, cd + f1 %.3f, emd %.3f' % (d['ms_per_step'], d['extra']['cd_f1_ms'], d['extra']['emd_ms']))" >> $out
done; done
cp /tmp/shipped.so $L
cat $out
