#!/bin/bash
# round 4, run N: bt4 with at most one odd-step stage per tile
set -x
cd /root/repo
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_bigtile.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r4/n_bt4_tests.log
timeout 300 python scripts/gemm_clock.py 47 100000 200 20 2>&1 | tee gpurun_out/r4/clock9_bt4_d200.log
timeout 300 python scripts/gemm_clock.py 47 60000 512 20 2>&1 | tee gpurun_out/r4/clock9_bt4_d512.log
timeout 300 python scripts/gemm_clock.py 47 100000 600 10 2>&1 | tee gpurun_out/r4/clock9_bt4_d600.log
timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -3 | tee gpurun_out/r4/n_bench.log
