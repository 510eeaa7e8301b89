cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
( python scripts/fit_groups_probe.py; PLDA_EM_VARIANT=3 python scripts/fit_groups_probe.py; PLDA_EM_VARIANT=4 python scripts/fit_groups_probe.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/fit_groups.txt; cat gpurun_out/r6/fit_groups.txt
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6/gputests.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r6/gputests.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r6/bench_c2_final.json 2> gpurun_out/r6/bench_c2_final.err; tail -2 gpurun_out/r6/bench_c2_final.err
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r6/bench_c2_final.json").read().strip().splitlines()[-1])
print({k:r.get(k) for k in ("value","ms_per_step","fit_em_iters_per_s","fit_skewed_em_iters_per_s")}, r["roofline"]["frac"], r["roofline"].get("traffic"), r["roofline"].get("clock",{}).get("shader_clock_MHz"))
print(r["fit"]["stats_ms"], r["fit"]["em_ms"], r["fit"]["output_ms"], r["fit_skewed"]["em_ms"], r["fit_skewed"]["cpu_oracle"]["gpu_vs_oracle_after_one_iteration"])
print(r.get("znorm_stats",{}).get("ms"), r.get("cpu_baseline",{}).get("value"), r.get("fit_d1024",{}).get("fit_wall_ms"))
PY
