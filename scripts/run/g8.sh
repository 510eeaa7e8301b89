cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_scoring.py tests/test_gpu_comm_procs.py tests/test_gpu_gemm64.py tests/test_gpu_bf16x3.py -x -q -s 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -15
