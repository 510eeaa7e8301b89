cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_fit.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -5
for v in "A=1"; do
echo "=== $v"
rm -rf gpurun_out/fitgroups_r6
env $v rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fitgroups_r6 -o t -- python scripts/fit_groups_probe.py 2>&1 | grep -v "^[WE]2026" | tail -4
f=$(find gpurun_out/fitgroups_r6 -name "*kernel_trace.csv" | head -1)
echo "G=36:"; python scripts/em_iter_trace.py $f 4 | cut -c1-80
echo "G=1:"; python scripts/em_iter_trace.py $f 10 | cut -c1-80
done
