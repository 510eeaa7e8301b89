#!/bin/bash
set -x
cd /root/repo
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_scoring.py -m gpu -x -q -k "transform" 2>&1 | tail -8 | tee gpurun_out/r4/s_tests.log
timeout 900 python scripts/transform_stream_probe.py 2>&1 | tee gpurun_out/r4/s_probe.log
