# PMC snapshot of the trials GEMM (product path) on the C2 shape: MFMA pipe utilisation, wait
# breakdown, LDS conflicts.  Counters in their own runs with --kernel-trace only.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
for SET in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS"; do
  tag=$(echo $SET | tr ' ' '_' | cut -c1-40)
  ( cd /tmp && timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/scripts/gemm_sweep.py 0 100000 200 2 > $OUT/$tag.log 2>&1 )
  python - <<PY >> $OUT/pmc_summary.txt
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trials_gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob("$OUT/$tag/**/*kernel_trace.csv", recursive=True):
    dur += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "trials_gemm" in r["Kernel_Name"]]
for k, v in agg.items():
    print("%-28s launches=%d  steady-state avg=%.6e" % (k, len(v), sum(v[1:]) / max(len(v) - 1, 1)))
print("kernel ms (this pass):", ["%.3f" % d for d in dur])
PY
done
cat $OUT/pmc_summary.txt
find $OUT -name "*.csv" -delete
