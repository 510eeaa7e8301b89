set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_gpu_bigtile.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r4/g_bt4_tests.log
timeout 300 python scripts/gemm_clock.py 47 100000 200 20 2>&1 | tee gpurun_out/r4/clock6_bt4_d200.log
timeout 300 python scripts/gemm_clock.py 47 60000 512 20 2>&1 | tee gpurun_out/r4/clock6_bt4_d512.log
timeout 300 python scripts/gemm_timeline4.py 32768 200 2>&1 | tee gpurun_out/r4/g_timeline_d200.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee gpurun_out/r4/gpu_suite_2.log
