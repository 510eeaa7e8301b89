#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_fit.py tests/test_gpu_fullsize.py tests/test_gpu_spd_inverse.py tests/test_gpu_golden.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r4/em_tests.log
timeout 900 python scripts/stress_parity3.py 40 5000 2>&1 | tail -3 | tee gpurun_out/r4/em_stress.log
for v in 0 2 0 2; do
PLDA_EM_VARIANT=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-extra 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); f=j['fit']; print('em variant $v', {k:f[k] for k in ('stats_ms','em_ms','output_ms','fit_wall_s','em_iters_per_s')})" | tee -a gpurun_out/r4/em_bench.log
done
