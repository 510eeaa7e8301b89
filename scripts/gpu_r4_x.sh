#!/bin/bash
set -x
cd /root/repo
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_gpu_scoring.py tests/test_gpu_api_edges.py tests/test_gpu_pipeline.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r4/x_tests.log
timeout 600 python scripts/transform_stream_probe.py 100000,200000,400000,800000 0,6,0 2>&1 | tee gpurun_out/r4/x_probe.log
PLDA_TRANSFORM_VARIANT=0 timeout 600 python scripts/transform_probe.py 2>&1 | tail -12 | tee gpurun_out/r4/x_probe2.log
