# round 2, first GPU call: parity of the big-tile kernels, A/B of the trials-GEMM generations, timeline
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2a
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bigtile.py -x -q -m gpu > $OUT/bigtile_tests.log 2>&1; echo "bigtile rc=$?" >> $OUT/bigtile_tests.log
tail -15 $OUT/bigtile_tests.log
timeout 300 python scripts/gemm_sweep.py 0,28 100000 200 6 > $OUT/sweep_d200.log 2>&1; cat $OUT/sweep_d200.log
timeout 300 python scripts/gemm_sweep.py 0,28 65536 512 4 > $OUT/sweep_d512.log 2>&1; cat $OUT/sweep_d512.log
timeout 300 python scripts/gemm_sweep.py 0,28 100000 152 4 > $OUT/sweep_d152.log 2>&1; cat $OUT/sweep_d152.log
timeout 300 python scripts/gemm_timeline.py 32768 200 > $OUT/timeline_d200.log 2>&1; tail -60 $OUT/timeline_d200.log
