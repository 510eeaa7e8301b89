cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fit.py -x -q -k "matches_oracle or extended or fewer or large_dim" 2>&1 | tail -3
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fitgroups_r6 -o t -- python scripts/fit_groups_probe.py 2>&1 | grep -v "^[WE]2026" | tail -6
f=$(find gpurun_out/fitgroups_r6 -name "*kernel_trace.csv" | head -1)
echo "G=36:"; python scripts/em_iter_trace.py $f 4
echo "G=1:"; python scripts/em_iter_trace.py $f 10
