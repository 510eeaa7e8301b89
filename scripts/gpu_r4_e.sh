# round 4: bt4 (tile queue, fringe scratch) + syrk_blk: parity, stamps, K2 timing
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_gpu_bigtile.py -m gpu -x -q -k "one_wave or bit_identical" 2>&1 | tail -15 | tee gpurun_out/r4/e_bt4_tests.log
timeout 300 python scripts/gemm_clock.py 47 100000 200 20 2>&1 | tee gpurun_out/r4/clock4_bt4_d200.log
timeout 300 python scripts/gemm_clock.py 47 60000 512 20 2>&1 | tee gpurun_out/r4/clock4_bt4_d512.log
timeout 300 python scripts/gemm_sweep.py 0,40 100000 200 5 2>&1 | tee gpurun_out/r4/e_sweep_d200.log
timeout 200 python scripts/gemm_soak.py 40 100000 200 30 2>&1 | tee gpurun_out/r4/e_soak40_d200.log
timeout 200 python scripts/gemm_soak.py 0 100000 200 30 2>&1 | tee gpurun_out/r4/e_soak0_d200.log
timeout 900 python -m pytest tests/test_gpu_gemm64.py tests/test_gpu_fit.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r4/e_syrk_tests.log
timeout 300 python scripts/k2_size_probe.py 2>&1 | tail -20 | tee gpurun_out/r4/e_k2_probe.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "c3" 2>&1 | tail -15 | tee gpurun_out/r4/e_fullsize_c3.log
