# kernel trace of the last dispatches of scripts/score_size_curve.py calls.  TRACES="D MxNt;D MxNt" EXTRA=mixed
cd /tmp; export TMPDIR=/tmp
IFS=";" read -ra LIST <<< "${TRACES:-200 8192x8192;512 256x256;200 4096x4096}"
for T in "${LIST[@]}"; do
 set -- $T
 OUT=$GRAFT_REPO_ROOT/gpurun_out/curve_trace_$1_$2; rm -rf $OUT
 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $GRAFT_REPO_ROOT/scripts/score_size_curve.py $1 - $2 $EXTRA > $OUT.log 2>&1
 python - <<PY
import csv,glob
rows=[]
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r in csv.DictReader(open(f))]
rows.sort()
# the last call's kernels: take the last 14 dispatches
last=rows[-16:]
t0=last[0][0]
print("== $1 $2 (last dispatches: start us, dur us, gap before us)")
prev=None
for a,b,n in last:
    print("  %9.2f %8.2f %7.2f  %s" % ((a-t0)/1e3,(b-a)/1e3,((a-prev)/1e3 if prev else 0),n)); prev=b
PY
 tail -2 $OUT.log
done
