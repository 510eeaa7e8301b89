# Round-2 closing evidence: profiles of C2 and C3 (rocprofv3 trace + PMC), bench lines C2 / C3 / C4-on-one-GPU,
# the eigensolver probe.  Output under gpurun_out/; the small summaries are copied to profiles/ by hand.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
ROUND=r02 CONFIG=C2 bash scripts/gpu_profile.sh > gpurun_out/final/profile_C2.log 2>&1
ROUND=r02 CONFIG=C3 bash scripts/gpu_profile.sh > gpurun_out/final/profile_C3.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/final/bench_c2.json 2> gpurun_out/final/bench_c2.err
timeout 600 python bench.py --config C3 --steps 10 --warmup 2 > gpurun_out/final/bench_c3.json 2> gpurun_out/final/bench_c3.err
timeout 900 python bench.py --config C4 --steps 3 --warmup 1 --no-cpu > gpurun_out/final/bench_c4.json 2> gpurun_out/final/bench_c4.err
timeout 300 python scripts/eig_probe.py 200:0 200:1 512:0 512:1 > gpurun_out/final/eig_probe.txt 2>&1
tail -c 600 gpurun_out/final/bench_c2.json; echo; tail -3 gpurun_out/final/eig_probe.txt
