#!/bin/bash
# Round-3 evidence, one box, one call: the whole GPU suite, rocprofv3 profiles of C2 (kernel trace + PMC passes,
# scripts/gpu_profile.sh), the un-profiled bench lines C2 / C3 / C4-on-one-GPU, the bench's own N > 1 path with two ranks on
# this one GPU (host transport) and by emulation, the transform sweep and the GEMM sweep with its bounding arms.
# Output under gpurun_out/r3/; the small summaries are copied to profiles/ (list in profiles/README.md).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3
mkdir -p $O
python -m pytest tests -q -m gpu --durations=8 > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
ROUND=r03 CONFIG=C2 bash scripts/gpu_profile.sh > $O/profile_C2.log 2>&1
python bench.py --steps 20 --warmup 3 > $O/r03_bench_c2.json 2> $O/bench_c2.err
python bench.py --config C3 --steps 10 --warmup 2 > $O/r03_bench_c3.json 2> $O/bench_c3.err
python bench.py --config C4 --steps 3 --warmup 1 --no-cpu > $O/r03_bench_c4_one_gpu.json 2> $O/bench_c4.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --transport host --rows 20000 --no-cpu > $O/r03_bench_2ranks_one_gpu_host_transport.json 2> $O/bench_2ranks.err
python bench.py --steps 3 --warmup 1 --emulate-ranks 8 --no-cpu --no-extra > $O/r03_bench_emulate8.json 2> $O/bench_emu8.err
SWEEP_VARIANTS=0,2,7,1 timeout 600 python scripts/transform_sweep.py > $O/r03_transform_sweep.txt 2>&1
( timeout 300 python scripts/gemm_sweep.py 0,34,35,36,32,20 100000 200; timeout 300 python scripts/gemm_sweep.py 0,34,35,36,32 65536 512 ) > $O/r03_gemm_sweep.txt 2>&1
bash scripts/gpu_r3_emtrace.sh 200 > $O/r03_em_kernel_stats.txt 2>&1
for f in r03_bench_c2 r03_bench_c3 r03_bench_c4_one_gpu r03_bench_2ranks_one_gpu_host_transport r03_bench_emulate8; do echo "== $f"; tail -c 700 $O/$f.json; echo; done
tail -9 $O/r03_transform_sweep.txt | cut -c1-200; cat $O/r03_gemm_sweep.txt | grep variant
