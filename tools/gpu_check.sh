#!/bin/bash
# Everything the driver runs at round end, on the gpurun box: the -m gpu suite, smoke(), the default bench.
# Usage: tools/gpu_check.sh <tag>      (outputs gpurun_out/<tag>_*)
tag=${1:-check}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/${tag}_gpu_tests.txt 2>&1
tail -5 gpurun_out/${tag}_gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.txt 2>&1; tail -2 gpurun_out/${tag}_smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 1500 gpurun_out/${tag}_bench.json
