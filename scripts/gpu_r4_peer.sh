set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_gpu_comm_procs.py tests/test_gpu_comm.py -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/r4/peer_tests.log
