cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
timeout 900 python bench.py --config C3 --steps 3 --warmup 1 --no-extra > gpurun_out/r6/bench_c3.json 2> gpurun_out/r6/bench_c3.err; tail -2 gpurun_out/r6/bench_c3.err
timeout 900 python bench.py --config C4 --steps 2 --warmup 1 --no-extra --no-cpu > gpurun_out/r6/bench_c4.json 2> gpurun_out/r6/bench_c4.err; tail -2 gpurun_out/r6/bench_c4.err
python - <<'PY'
import json
for c in ("c3","c4"):
    try:
        r=json.loads(open("gpurun_out/r6/bench_%s.json"%c).read().strip().splitlines()[-1])
        f=r["fit"]; print(c, r["value"], r["ms_per_step"], r["roofline"]["frac"], {k:f[k] for k in ("stats_ms","em_ms","output_ms","em_iters_per_s","em_form")}, r.get("oracle_check",{}).get("max_abs_err"))
        cb=r.get("cpu_baseline",{}).get("fit_em"); print(cb)
    except Exception as e: print(c, "ERR", e)
PY
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6/gputests.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r6/gputests.txt | tail -3
