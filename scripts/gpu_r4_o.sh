#!/bin/bash
# round 4, run O: one-pass scoring prep (prep_side_kernel)
set -x
cd /root/repo
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_scoring.py tests/test_gpu_bigtile.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r4/o_tests.log
for v in 0 1 0 1; do
PLDA_PREP_VARIANT=$v timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('prep variant $v', j['ms_per_step'], j['roofline']['avg_kernel_ms'], j['roofline']['frac'], j['value'])" | tee -a gpurun_out/r4/o_bench.log
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r4/o_prof -o o -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu > /root/repo/gpurun_out/r4/o_prof.log 2>&1
cd /root/repo
python scripts/summarize_prof.py gpurun_out/r4/o_prof 2>&1 | head -30 | tee gpurun_out/r4/o_prof_summary.txt
find gpurun_out/r4/o_prof -name "*.csv" ! -name "*kernel_stats*" -delete; find gpurun_out/r4/o_prof -name "*.db" -delete
