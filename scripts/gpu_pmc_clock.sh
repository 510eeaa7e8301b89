cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=/tmp/pmcc; rm -rf $OUT; mkdir -p $OUT
for V in 9 11 21; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/v$V -o p -- python $GRAFT_REPO_ROOT/scripts/gemm_sweep.py $V 60000 800 2 > $OUT/v$V.log 2>&1 )
  python - <<PY
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("$OUT/v$V/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trials_gemm" in r["Kernel_Name"]:
            rows[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
dur = {}
for f in glob.glob("$OUT/v$V/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trials_gemm" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k in sorted(rows, key=int):
    c = rows[k]; ms = dur.get(k, float("nan"))
    gui = c.get("GRBM_GUI_ACTIVE", 0); sqb = c.get("SQ_BUSY_CYCLES", 0); mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    print("variant $V dispatch %s: %.3f ms  GUI/8=%.4e (%.3f GHz)  SQ_BUSY/32=%.4e (%.3f GHz)  MFMA busy/1024=%.4e -> util vs GUI %.1f%%" % (
        k, ms, gui / 8, gui / 8 / ms / 1e6, sqb / 32, sqb / 32 / ms / 1e6, mf / 1024, 100 * (mf / 1024) / (gui / 8)))
PY
done
