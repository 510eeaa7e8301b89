# round 4: first run of the one-wave-per-SIMD trials GEMM (variant 40): parity, then interleaved sweeps
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_gpu_bigtile.py -m gpu -x -q -k "one_wave or bit_identical" 2>&1 | tail -15 | tee gpurun_out/r4/bt4_tests.log
timeout 300 python scripts/gemm_sweep.py 0,40,44,45,46,34,35,36 100000 200 5 2>&1 | tee gpurun_out/r4/bt4_sweep_d200.log
timeout 300 python scripts/gemm_sweep.py 0,40 60000 512 4 2>&1 | tee gpurun_out/r4/bt4_sweep_d512.log
timeout 300 python scripts/gemm_sweep.py 0,40 80000 256 4 2>&1 | tee gpurun_out/r4/bt4_sweep_d256.log
