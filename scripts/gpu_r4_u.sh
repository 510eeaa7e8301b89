#!/bin/bash
set -x
cd /root/repo
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_bigtile.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r4/u_tests.log
timeout 300 python scripts/gemm_clock.py 47 100000 200 20 2>&1 | tee gpurun_out/r4/u_clock_bt4_d200.log
timeout 300 python scripts/gemm_clock.py 47 60000 512 20 2>&1 | tee gpurun_out/r4/u_clock_bt4_d512.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu --no-extra 2>&1 | tail -1 > gpurun_out/r4/u_bench.json
python -c "
import json; j=json.load(open('gpurun_out/r4/u_bench.json')); print(j['ms_per_step'], j['roofline']['frac'], j['roofline']['avg_kernel_ms'])"
