#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4
export MASTER_ADDR=127.0.0.1
for T in host peer; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --transport $T --rows 20000 --no-cpu 2>&1 | tail -1 > gpurun_out/r4/two_$T.json
python -c "
import json; j=json.load(open('gpurun_out/r4/two_$T.json')); print('$T', j['n_gpus'], j['value'], j.get('multi_gpu',{}).get('cross_rank_checksum_ok'), j.get('gather_inclusive',{}).get('gathered_blocks_bit_identical_to_their_owners'), list(j.get('gather_inclusive',{}).keys())[:4])" 2>&1 | tail -2
done
