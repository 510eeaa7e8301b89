cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for SET in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $SET | tr ' ' '_' | cut -c1-30)
  rm -rf gpurun_out/em_pmc_$tag
  ( cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/em_pmc_$tag -o p -- python $GRAFT_REPO_ROOT/scripts/fit_groups_probe.py > /dev/null 2>&1 )
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/em_pmc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for name in ("em_xtb_kernel", "em_rows_kernel", "syrk_tri_kernel", "em_syrk_reduce_mstep", "spd_inverse_mfma"):
            if name in k:
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in agg.items():
    # the G = 36 case: launches 60..90 of 120 per kernel (4 cases x 3 fits x 10 iterations); take those
    print(name, {c: "%.4g" % (sum(v[60:90]) / max(len(v[60:90]), 1)) for c, v in d.items()}, "launches", {c: len(v) for c, v in d.items()})
PY
done
