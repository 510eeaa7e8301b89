#!/bin/bash
set -x
cd /root/repo
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_gpu_scoring.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r4/t_tests.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu 2>&1 | tail -1 > gpurun_out/r4/t_bench.json
python -c "
import json; j=json.load(open('gpurun_out/r4/t_bench.json')); print(j['ms_per_step'], j['roofline']['frac'], j['fit']['output_ms'], j.get('transform'))"
