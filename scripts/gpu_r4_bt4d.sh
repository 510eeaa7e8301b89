# round 4: bt4 with the tile queue (dynamic schedule): parity, per-workgroup stamps
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_gpu_bigtile.py -m gpu -x -q -k "one_wave or bit_identical" 2>&1 | tail -15 | tee gpurun_out/r4/bt4d_tests.log
timeout 300 python scripts/gemm_clock.py 47 100000 200 20 2>&1 | tee gpurun_out/r4/clock3_bt4_d200.log
timeout 300 python scripts/gemm_clock.py 47 60000 512 20 2>&1 | tee gpurun_out/r4/clock3_bt4_d512.log
timeout 300 python scripts/gemm_sweep.py 0,40 100000 200 5 2>&1 | tee gpurun_out/r4/bt4d_sweep_d200.log
timeout 300 python scripts/gemm_sweep.py 0,40 60000 512 4 2>&1 | tee gpurun_out/r4/bt4d_sweep_d512.log
timeout 200 python scripts/gemm_soak.py 40 100000 200 40 2>&1 | tee gpurun_out/r4/bt4d_soak_d200.log
timeout 200 python scripts/gemm_soak.py 0 100000 200 40 2>&1 | tee gpurun_out/r4/bt2d_soak_d200.log
