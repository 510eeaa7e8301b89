cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fitgroups_r6 -o t -- python scripts/fit_groups_probe.py 2>&1 | grep -v "^[WE]2026" | tail -6
f=$(find gpurun_out/fitgroups_r6 -name "*kernel_trace.csv" | head -1)
echo "G=40:"; python scripts/em_iter_trace.py $f 1
echo "G=36:"; python scripts/em_iter_trace.py $f 4
echo "G=1:"; python scripts/em_iter_trace.py $f 10
