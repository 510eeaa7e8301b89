# round 4: bt4 (cheaper tile fetch) + syrk_blk fix
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_gpu_bigtile.py -m gpu -x -q -k "one_wave or bit_identical" 2>&1 | tail -15 | tee gpurun_out/r4/f_bt4_tests.log
timeout 300 python scripts/gemm_clock.py 47 100000 200 20 2>&1 | tee gpurun_out/r4/clock5_bt4_d200.log
timeout 300 python scripts/gemm_clock.py 37 100000 200 20 2>&1 | tee gpurun_out/r4/clock5_bt2_d200.log
timeout 300 python scripts/gemm_clock.py 47 60000 512 20 2>&1 | tee gpurun_out/r4/clock5_bt4_d512.log
timeout 900 python -m pytest tests/test_gpu_gemm64.py tests/test_gpu_fit.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r4/f_syrk_tests.log
timeout 300 python scripts/k2_size_probe.py 2>&1 | tail -12 | tee gpurun_out/r4/f_k2_probe.log
