#!/bin/bash
# round 4, last check of the final code: the whole GPU suite and the default bench line
set -x
cd /root/repo
mkdir -p gpurun_out/r4
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r4/final2_gpu_suite.log
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r4/final2_bench_c2.json
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3
