#!/bin/bash
set -x
cd /root/repo
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r4/p_prof -o p -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu --no-extra > /root/repo/gpurun_out/r4/p_prof.log 2>&1 )
python scripts/summarize_prof.py gpurun_out/r4/p_prof 2>&1 | head -34 | tee gpurun_out/r4/p_prof_summary.txt
find gpurun_out/r4/p_prof -name "*.csv" -size +1M -delete; find gpurun_out/r4/p_prof -name "*.db" -delete
