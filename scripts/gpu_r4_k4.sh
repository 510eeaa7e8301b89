set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_scoring.py -m gpu -x -q -k "transform" 2>&1 | tail -6 | tee gpurun_out/r4/k4_tests.log
SWEEP_VARIANTS=0,8 timeout 600 python scripts/transform_sweep.py 2>&1 | tail -30 | tee gpurun_out/r4/k4_sweep.log
