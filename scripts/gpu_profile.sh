# rocprofv3 passes for a bench workload, files keyed to the round:
#   (1) kernel trace + stats, (2)/(3) HBM byte counters in their own runs (TCC slots: FETCH_SIZE and WRITE_SIZE
#   cannot share a pass), (4) SQ counters of the dominant kernel.
# usage: ROUND=r02 [CONFIG=C2] bash scripts/gpu_profile.sh    -> gpurun_out/prof_<ROUND>_<CONFIG>/ ; copy the
# small summaries into profiles/ afterwards.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ROUND=${ROUND:-r02}
CONFIG=${CONFIG:-C2}
case $CONFIG in C2) SHAPE=100000,100000,200;; C3) SHAPE=10000,1000000,512;; C4) SHAPE=40000,1200000,256;; esac
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${ROUND}_${CONFIG}
rm -rf $OUT; mkdir -p $OUT
ARGS="${BENCH_ARGS:---config $CONFIG --steps 4 --warmup 1 --no-cpu --no-extra}"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/${ROUND}_${CONFIG}_bench_under_trace.json 2> $OUT/trace.err )
( cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/fetch.err )
( cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/write.err )
python $GRAFT_REPO_ROOT/scripts/summarize_prof.py $OUT ${ROUND}_${CONFIG} $SHAPE > $OUT/${ROUND}_${CONFIG}_rocprofv3_summary.txt 2>&1
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/${ROUND}_${CONFIG}_kernel_stats.csv; done
for SET in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  tag=$(echo $SET | tr ' ' '_' | cut -c1-40)
  ( cd /tmp && timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/sq_$tag -o p -- python $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/sq_$tag.err )
  python - <<PY >> $OUT/${ROUND}_${CONFIG}_pmc_trials_gemm.txt
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/sq_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trials_gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob("$OUT/sq_$tag/**/*kernel_trace.csv", recursive=True):
    dur += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "trials_gemm" in r["Kernel_Name"]]
for k, v in agg.items():
    print("%-28s launches=%d  steady-state avg=%.6e" % (k, len(v), sum(v[1:]) / max(len(v) - 1, 1)))
print("kernel ms (this pass):", ["%.3f" % d for d in dur])
PY
done
cat $OUT/${ROUND}_${CONFIG}_rocprofv3_summary.txt | head -60; cat $OUT/${ROUND}_${CONFIG}_pmc_trials_gemm.txt; cat $OUT/*traffic*.json
# keep only the small artefacts for the merge back
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
