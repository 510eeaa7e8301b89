# Bench lines and configuration probes only (no rocprofv3 passes): for changes that do not touch the trials kernel.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/final/bench_c2.json 2> gpurun_out/final/bench_c2.err
timeout 600 python bench.py --config C3 --steps 10 --warmup 2 > gpurun_out/final/bench_c3.json 2> gpurun_out/final/bench_c3.err
timeout 900 python bench.py --config C4 --steps 3 --warmup 1 --no-cpu > gpurun_out/final/bench_c4.json 2> gpurun_out/final/bench_c4.err
timeout 900 python scripts/gpu_configs.py > gpurun_out/final/configs.json 2> gpurun_out/final/configs.err
timeout 300 python scripts/eer_probe.py > gpurun_out/final/eer_probe.json 2> gpurun_out/final/eer_probe.err
timeout 300 python scripts/transform_probe.py > gpurun_out/final/transform_probe.json 2> gpurun_out/final/transform_probe.err
tail -c 400 gpurun_out/final/bench_c2.json; echo; tail -c 300 gpurun_out/final/configs.json
