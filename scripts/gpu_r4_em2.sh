#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests/test_gpu_fit.py tests/test_gpu_fullsize.py tests/test_gpu_api_edges.py tests/test_gpu_golden.py tests/test_gpu_pipeline.py tests/test_gpu_comm.py tests/test_gpu_comm_procs.py tests/test_gpu_hostio.py -m gpu -x -q 2>&1 | grep -v 'version\|Hostname\|Librccl' | tail -5 | tee gpurun_out/r4/em2_tests.log
timeout 900 python scripts/stress_parity3.py 30 6000 2>&1 | tail -2 | tee gpurun_out/r4/em2_stress.log
rm -f gpurun_out/r4/em2_bench.log
for i in 1 2 3; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-extra 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); f=j['fit']; print({k:f[k] for k in ('stats_ms','em_ms','output_ms','fit_wall_s','em_iters_per_s')})" | tee -a gpurun_out/r4/em2_bench.log
done
