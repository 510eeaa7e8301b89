cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_fit.py -x -q 2>&1 | tail -4
rm -rf gpurun_out/fitgroups_r6
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fitgroups_r6 -o t -- python scripts/fit_groups_probe.py 2>&1 | grep -v "^[WE]2026" | tail -5
f=$(find gpurun_out/fitgroups_r6 -name "*kernel_trace.csv" | head -1)
echo "G=36:"; python scripts/em_iter_trace.py $f 4 | cut -c1-100
echo "G=12:"; python scripts/em_iter_trace.py $f 7 | cut -c1-100
python scripts/fit_groups_probe.py 2>&1 | tail -4
