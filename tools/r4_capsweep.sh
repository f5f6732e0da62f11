# (the MVP_EMD_* knobs are read by libmvpops_hooks.so only: make -C mvp_benchmark_amd/csrc hooks)
timeout 900 python -m pytest tests/test_gpu_emd_resident.py -x -q 2>&1 | tail -2
for cap in 16 32; do
  echo "## cap=$cap"
  for n in 1024 2048 4096; do
    MVP_EMD_RESIDENT_CAP=$cap MVP_BENCH_REPS=4 timeout 300 python tools/bench_emd_one.py 64 $n 0.004 3000 mvp_benchmark_amd/libmvpops_hooks.so 2>&1 | grep W=
  done
done
for sp in 3 2; do echo "## split=$sp"; for n in 1024 2048 4096; do MVP_EMD_SPLIT=$sp MVP_BENCH_REPS=4 timeout 300 python tools/bench_emd_one.py 64 $n 0.004 3000 mvp_benchmark_amd/libmvpops_hooks.so 2>&1 | grep W=; done; done
for n in 1024 4096; do timeout 300 python tools/bench_emd_one.py 64 $n 0.004 3000 mvp_benchmark_amd/libmvpops_prof.so 2>&1 | grep -E "resident cloud 0 wave 0|W="; done
