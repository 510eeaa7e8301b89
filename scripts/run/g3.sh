cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fit.py -x -q 2>&1 | tail -5
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fitgroups_r6 -o t -- python scripts/fit_groups_probe.py 2>&1 | grep -v "^[WE]2026" | tail -6
f=$(find gpurun_out/fitgroups_r6 -name "*kernel_trace.csv" | head -1)
echo "G=36:"; python scripts/em_iter_trace.py $f 4
echo "G=1:"; python scripts/em_iter_trace.py $f 10
for r in 32 48 96; do echo rows $r; PLDA_EM_SYRK_ROWS=$r python scripts/fit_groups_probe.py 2>&1 | tail -4; done
