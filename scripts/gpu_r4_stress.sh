#!/bin/bash
# round 4: randomised parity sweeps with fresh seeds on the final code
cd /root/repo
mkdir -p gpurun_out/r4
timeout 1500 python scripts/stress_parity.py 300 4000 2>&1 | tail -4 | tee gpurun_out/r4/stress1.log
timeout 1500 python scripts/stress_parity2.py 200 4000 2>&1 | tail -4 | tee gpurun_out/r4/stress2.log
timeout 1500 python scripts/stress_parity3.py 40 4000 2>&1 | tail -4 | tee gpurun_out/r4/stress3.log
