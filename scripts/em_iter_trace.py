"""One EM iteration of the last fit in a rocprofv3 kernel trace, kernel by kernel (start offset, duration in us).
usage: python scripts/em_iter_trace.py <..._kernel_trace.csv> [fit index from the end, default 1] [iteration, default 5]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
itn = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ms = [i for i, r in enumerate(rows) if any(m in r["Kernel_Name"] for m in ("em_rows_mstep_kernel", "em_moment_mstep_kernel", "em_rank_reduce_mstep_kernel"))]
per_fit = 10
a, b = ms[len(ms) - back * per_fit + itn - 1], ms[len(ms) - back * per_fit + itn]
t0 = int(rows[a]["End_Timestamp"])
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %8.1f  %-60s grid=%s wg=%s lds=%s vgpr=%s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:60], r["Grid_Size_X"],
                                                        r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"]))
print("iteration: %.1f us" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3))
