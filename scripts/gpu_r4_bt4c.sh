# round 4: bt4 with staggered DMA + in-kernel edges: parity, cycles per tile, then the whole GPU suite
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_gpu_bigtile.py -m gpu -x -q -k "one_wave or bit_identical" 2>&1 | tail -15 | tee gpurun_out/r4/bt4c_tests.log
timeout 300 python scripts/gemm_clock.py 47 100000 200 20 2>&1 | tee gpurun_out/r4/clock2_bt4_d200.log
timeout 300 python scripts/gemm_clock.py 37 100000 200 20 2>&1 | tee gpurun_out/r4/clock2_bt2_d200.log
timeout 300 python scripts/gemm_clock.py 47 60000 512 20 2>&1 | tee gpurun_out/r4/clock2_bt4_d512.log
timeout 300 python scripts/gemm_sweep.py 0,40 100000 200 5 2>&1 | tee gpurun_out/r4/bt4c_sweep_d200.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/r4/gpu_suite_1.log
