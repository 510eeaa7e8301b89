#!/bin/bash
# round 3, first GPU pass: bench at N = 1, the bench's own N > 1 path with two ranks on the one GPU (host transport),
# and the emulate-ranks logic check
set -x
mkdir -p gpurun_out/r3a
python bench.py --steps 10 --warmup 2 > gpurun_out/r3a/bench_c2.json 2> gpurun_out/r3a/bench_c2.err
tail -c 6000 gpurun_out/r3a/bench_c2.json
tail -5 gpurun_out/r3a/bench_c2.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --transport host --rows 20000 --no-cpu > gpurun_out/r3a/bench_2ranks_host.json 2> gpurun_out/r3a/bench_2ranks_host.err
tail -c 3000 gpurun_out/r3a/bench_2ranks_host.json
tail -5 gpurun_out/r3a/bench_2ranks_host.err
python bench.py --steps 2 --warmup 1 --emulate-ranks 4 --no-cpu --no-extra > gpurun_out/r3a/bench_emu4.json 2> gpurun_out/r3a/bench_emu4.err
tail -c 1500 gpurun_out/r3a/bench_emu4.json
tail -5 gpurun_out/r3a/bench_emu4.err
