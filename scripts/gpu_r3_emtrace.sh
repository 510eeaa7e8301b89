#!/bin/bash
# kernel trace of an EM-dominated fit (D = 200 by default): per-kernel average durations
cd /tmp && export TMPDIR=/tmp
D=${1:-200}
rm -rf /tmp/emtr && mkdir -p /tmp/emtr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/emtr -o t -- python $GRAFT_REPO_ROOT/scripts/em_probe.py $D > /tmp/emtr/log.txt 2>&1
tail -2 /tmp/emtr/log.txt
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/emtr/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("%-70s %8s %10s %8s" % ("kernel", "calls", "avg_us", "pct"))
for r in rows[:22]:
    print("%-70s %8s %10.2f %8s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
