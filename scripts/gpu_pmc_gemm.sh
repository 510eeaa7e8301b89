set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -iE "MFMA|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY|SQ_WAIT_ANY|SQ_ACTIVE_INST_ANY|MfmaUtil|SQ_INSTS_MFMA|SQ_INST_CYCLES_VMEM|LDS_BANK_CONFLICT" | head -40 > $OUT/counters.txt
cat $OUT/counters.txt | cut -c1-160
for SET in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $SET | tr ' ' '_' | cut -c1-40)
  ( cd /tmp && timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/scripts/gemm_sweep.py 0 100000 200 2 > $OUT/$tag.log 2>&1 )
  python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/$tag/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "trials_gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, "n=%d" % len(v), "avg=%.6e" % (sum(v)/len(v)))
    # kernel duration from the trace rows of the same run
for f in glob.glob("$OUT/$tag/**/*kernel_trace.csv", recursive=True):
    d = [ (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in csv.DictReader(open(f)) if "trials_gemm" in r["Kernel_Name"]]
    print("kernel ms", d)
PY
done
find $OUT -name "*.csv" -size +1M -delete
