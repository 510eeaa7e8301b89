# round 4: timeline + bounding arms of the one-wave-per-SIMD trials GEMM
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 300 python scripts/gemm_timeline4.py 32768 200 2>&1 | tee gpurun_out/r4/bt4_timeline_d200.log
timeout 300 python scripts/gemm_timeline4.py 32768 512 2>&1 | tee gpurun_out/r4/bt4_timeline_d512.log
timeout 300 python scripts/gemm_sweep.py 0,40,44,45,46 100000 200 5 2>&1 | tee gpurun_out/r4/bt4_sweep2_d200.log
timeout 300 python scripts/gemm_sweep.py 0,40,44,45,46 60000 512 4 2>&1 | tee gpurun_out/r4/bt4_sweep2_d512.log
timeout 200 python scripts/gemm_soak.py 40 100000 200 40 2>&1 | tee gpurun_out/r4/bt4_soak_d200.log
timeout 200 python scripts/gemm_soak.py 0 100000 200 40 2>&1 | tee gpurun_out/r4/bt2_soak_d200.log
