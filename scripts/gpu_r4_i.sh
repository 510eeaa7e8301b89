set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --transport peer --steps 2 --warmup 1 --rows 20000 --no-cpu > gpurun_out/r4/bench_2ranks_one_gpu_peer.json 2> gpurun_out/r4/bench_2ranks_one_gpu_peer.err; tail -c 2500 gpurun_out/r4/bench_2ranks_one_gpu_peer.json; tail -3 gpurun_out/r4/bench_2ranks_one_gpu_peer.err
