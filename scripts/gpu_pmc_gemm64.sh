# PMC snapshot of the fp64 GEMM (gemm_f64_kernel<..., 128>) on the LDA decision shape (100k x 5000 x 200).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc64
rm -rf $OUT; mkdir -p $OUT
for SET in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS"; do
  tag=$(echo $SET | tr ' ' '_' | cut -c1-40)
  ( cd /tmp && timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_configs.py lda > $OUT/$tag.log 2>&1 )
  python - <<PY >> $OUT/pmc_summary.txt
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_f64_kernel<true, true, 128>" in r["Kernel_Name"] and int(r.get("Grid_Size", "0") or 0) > 5000000:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("%-28s launches=%d  avg=%.6e" % (k, len(v), sum(v) / max(len(v), 1)))
PY
done
cat $OUT/pmc_summary.txt
find $OUT -name "*.csv" -delete
