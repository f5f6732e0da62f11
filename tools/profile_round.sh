#!/bin/bash
# Round profile on the GPU box: kernel trace of bench.py + PMC passes for the
# dominant kernel.  Usage: tools/profile_round.sh <tag>   (outputs gpurun_out/<tag>/)
set -u
tag=${1:-r1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- \
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side > $out/bench_traced.json 2> $out/trace.err
for P in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  name=$(echo $P | cut -d" " -f1)
  timeout 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/pmc_$name -o run -- \
    python tools/run_op.py emd 1 > /dev/null 2> $out/pmc_$name.err
  timeout 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/pmc_cd_$name -o run -- \
    python tools/run_op.py cd 1 > /dev/null 2>> $out/pmc_$name.err
done
find $out -name "*.csv" | head -40
