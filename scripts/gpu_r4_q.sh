#!/bin/bash
# round 4, run Q: GetOutput work (deep reflector prefetch, QL chain without LDS round trip, no per-level memsets, pinned model copies, no EM|GetOutput sync)
set -x
cd /root/repo
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_eig.py tests/test_gpu_fit.py tests/test_gpu_lda.py tests/test_gpu_api_edges.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r4/q_tests.log
for i in 1 2; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-extra 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); f=j['fit']; print({k:f[k] for k in ('stats_ms','em_ms','output_ms','fit_wall_s','em_iters_per_s')}); print([ (s['name'],s['ms']) for s in f['stages'] if 'getoutput' in s['name']])" | tee -a gpurun_out/r4/q_bench.log
done
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r4/q_prof -o q -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu --no-extra > /root/repo/gpurun_out/r4/q_prof.log 2>&1 )
find gpurun_out/r4/q_prof -name "*.db" -delete
