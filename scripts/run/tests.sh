cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6/gputests.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r6/gputests.txt | tail -5
