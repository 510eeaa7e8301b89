# rocprofv3 passes for the bench workload: (1) kernel trace + stats, (2)/(3) HBM byte counters
# in their own runs (TCC slots: FETCH_SIZE and WRITE_SIZE cannot share a pass).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
ARGS="${BENCH_ARGS:---steps 3 --warmup 1 --no-cpu --no-extra}"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err )
( cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/fetch_bench.json 2> $OUT/fetch.err )
( cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/write_bench.json 2> $OUT/write.err )
find $OUT -type f | head -50
python $GRAFT_REPO_ROOT/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt; cat $OUT/traffic_trials_gemm.json
# keep only small artefacts for the merge back
find $OUT -name "*.csv" -size +4M -delete
find $OUT -name "*.db" -delete
