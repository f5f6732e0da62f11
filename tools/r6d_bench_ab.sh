#!/bin/bash
# The default bench line's timed region (the driver's clouds) on several libraries alternately, same box:
#   tools/r6d_bench_ab.sh name ...     (mvp_benchmark_amd/libmvpops_<name>.so; "shipped" = the library in place)
out=gpurun_out/r6d_bench_ab.txt; mkdir -p gpurun_out; : > $out
L=mvp_benchmark_amd/libmvpops.so
cp $L /tmp/shipped.so
for pass in 1 2 3; do for v in "$@"; do
  if [ $v = shipped ]; then cp /tmp/shipped.so $L; else cp mvp_benchmark_amd/libmvpops_$v.so $L; fi
  python bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-side 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v: step %.3f ms, cd + f1 %.3f, emd %.3f' % (d['ms_per_step'], d['extra']['cd_f1_ms'], d['extra']['emd_ms']))" >> $out
done; done
cp /tmp/shipped.so $L
cat $out
