cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/fitgroups_trace; rm -rf $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $GRAFT_REPO_ROOT/scripts/fit_groups_probe.py > $OUT.log 2>&1
python - <<PY
import csv,glob
rows=[]
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:64], r.get("Grid_Size_X", "")) for r in csv.DictReader(open(f))]
rows.sort()
# last fit = G=40; find the G=36 fit: take the 3rd-from-last group of fits. Simpler: print the last fit's kernels (G=40)
# locate last 'set_identity2' occurrence
idx=[i for i,r in enumerate(rows) if "set_identity2" in r[2]]
i0=idx[-1]-12
t0=rows[i0][0]; prev=None
agg={}
for a,b,n,g in rows[i0:]:
    agg.setdefault(n,[0,0.0]); agg[n][0]+=1; agg[n][1]+=(b-a)/1e3
print("last fit: kernels by total us")
for n,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:16]:
    print("  %8.1f us  x%4d  %s" % (t,c,n))
print("total span us:", (rows[-1][1]-t0)/1e3)
# one EM iteration in the middle: find spd kernels
sp=[i for i,r in enumerate(rows) if i>=i0 and "spd_block" in r[2] or (i>=i0 and "sweep_mfma" in r[2])]
if len(sp)>5:
    a0=sp[4]; a1=sp[5]
    prev=None
    for a,b,n,g in rows[a0:a1]:
        print("   %8.2f dur %7.2f gap %6.2f grid %8s %s" % ((a-rows[a0][0])/1e3,(b-a)/1e3,((a-prev)/1e3 if prev else 0),g,n)); prev=b
PY
