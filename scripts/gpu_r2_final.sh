# Round-2 closing evidence, one box, one call: rocprofv3 profiles of C2 and C3 (kernel trace + PMC passes,
# scripts/gpu_profile.sh), then the un-profiled bench lines C2 / C3 / C4-on-one-GPU and the configuration probes
# (scripts/gpu_r2_refresh.sh), the eigensolver probe and the GEMM sweep.  Output under gpurun_out/; the small
# summaries are copied to profiles/ by hand (list in profiles/README.md).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
ROUND=r02 CONFIG=C2 bash scripts/gpu_profile.sh > gpurun_out/final/profile_C2.log 2>&1
ROUND=r02 CONFIG=C3 bash scripts/gpu_profile.sh > gpurun_out/final/profile_C3.log 2>&1
bash scripts/gpu_r2_refresh.sh > gpurun_out/final/refresh.log 2>&1
timeout 300 python scripts/eig_probe.py 200:0 200:1 512:0 512:1 > gpurun_out/final/eig_probe.txt 2>&1
( timeout 300 python scripts/gemm_sweep.py 0,32,20 100000 200; timeout 300 python scripts/gemm_sweep.py 0,32 65536 512 ) > gpurun_out/final/gemm_sweep.txt 2>&1
tail -c 600 gpurun_out/final/bench_c2.json; echo; tail -3 gpurun_out/final/eig_probe.txt; tail -6 gpurun_out/final/gemm_sweep.txt
