#!/bin/bash
# round 4, closing run: the whole GPU suite, the bench line, rocprofv3 evidence for C2 / C3 / C4
set -x
cd /root/repo
mkdir -p gpurun_out/r4
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee gpurun_out/r4/final_gpu_suite.log
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r4/final_bench_c2.json
timeout 600 python scripts/gemm_clock.py 47 100000 200 20 2>&1 | tee gpurun_out/r4/final_clock_bt4_d200.log
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/gpu_r4_profiles.sh > gpurun_out/r4/final_profiles.log 2>&1
tail -5 gpurun_out/r4/final_profiles.log
