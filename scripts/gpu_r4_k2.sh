#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4
PLDA_GEMM64_VARIANT=6 timeout 900 python -m pytest tests/test_gpu_fit.py tests/test_gpu_gemm64.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r4/k2_tests.log
rm -f gpurun_out/r4/k2_bench.log
for v in 0 6 0 6; do
PLDA_GEMM64_VARIANT=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-extra 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); f=j['fit']; print('gemm64 variant $v', f['stats_ms'], [ (s['name'][:28],s['ms']) for s in f['stages'] if 'K2' in s['name'] or 'K1' in s['name']])" | tee -a gpurun_out/r4/k2_bench.log
done
