cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r6/bench_c2.json 2> gpurun_out/r6/bench_c2.err; tail -3 gpurun_out/r6/bench_c2.err
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r6/bench_c2.json").read().strip().splitlines()[-1])
print({k:r[k] for k in ("value","ms_per_step","fit_em_iters_per_s","fit_skewed_em_iters_per_s") if k in r})
print(json.dumps(r.get("fit_skewed"),indent=1))
print(r["fit"]["em_form"], r["fit"]["em_ms"], r["roofline"]["frac"])
PY
( python scripts/fit_groups_probe.py; PLDA_EM_VARIANT=3 python scripts/fit_groups_probe.py; PLDA_EM_VARIANT=4 python scripts/fit_groups_probe.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/fit_groups.txt; cat gpurun_out/r6/fit_groups.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
