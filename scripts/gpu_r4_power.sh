# round 4: is the trials GEMM limited by its schedule or by the chip's power management?
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/probe/mfma32_power_probe.hip -o /tmp/mfma32_power_probe 2>/dev/null
timeout 120 /tmp/mfma32_power_probe 1.5 2>&1 | tee gpurun_out/r4/mfma32_power_probe.log
timeout 300 python scripts/gemm_clock.py 37 100000 200 30 2>&1 | tee gpurun_out/r4/clock_bt2_d200.log
timeout 300 python scripts/gemm_clock.py 47 100000 200 30 2>&1 | tee gpurun_out/r4/clock_bt4_d200.log
timeout 300 python scripts/gemm_clock.py 47 60000 512 20 2>&1 | tee gpurun_out/r4/clock_bt4_d512.log
