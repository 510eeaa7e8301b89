#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
rm -rf gpurun_out/r4/t_prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/r4/t_prof -o t -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu --no-extra > /root/repo/gpurun_out/r4/t_prof.log 2>&1 )
find gpurun_out/r4/t_prof -name "*.db" -delete
