#!/bin/bash
# round 4, run R: K4 with T resident in registers
set -x
cd /root/repo
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_scoring.py -m gpu -x -q -k "transform" 2>&1 | tail -8 | tee gpurun_out/r4/r_tests.log
for v in 0 8 0 8; do
PLDA_TRANSFORM_VARIANT=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('transform variant $v', j.get('transform'))" | tee -a gpurun_out/r4/r_bench.log
done
