cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_scoring.py tests/test_gpu_fullsize.py tests/test_gpu_api_edges.py tests/test_gpu_golden.py -x -q -k "norm or c5 or golden" 2>&1 | tail -4
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/r6/bench_zn.json 2> gpurun_out/r6/bench_zn.err; tail -2 gpurun_out/r6/bench_zn.err
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r6/bench_zn.json").read().strip().splitlines()[-1])
print(json.dumps(r.get("znorm_stats"))[:1500])
print(json.dumps(r.get("fit_d1024"))[:2500])
PY
PLDA_ZNORM_VARIANT=3 python scripts/znorm_probe.py 2>&1 | tail -5
python scripts/znorm_probe.py 2>&1 | tail -5
