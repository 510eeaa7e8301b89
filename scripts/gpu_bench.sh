set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -3
timeout 1200 python bench.py --steps 5 --warmup 1 > gpurun_out/bench.log 2>gpurun_out/bench.err; tail -5 gpurun_out/bench.err; cat gpurun_out/bench.log
