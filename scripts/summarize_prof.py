"""Summarise rocprofv3 output dirs: per-kernel time (kernel-trace --stats) and per-kernel
FETCH_SIZE / WRITE_SIZE from the PMC passes (units and gfx950 correction per
MI355X_MICROARCH.md section HBM: counters are in KiB; FETCH_SIZE reads exactly 1/2 of a wide
coalesced stream on gfx950 and is doubled here; WRITE_SIZE is uncalibrated)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "rXX"            # round tag of the written files
shape = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [100000, 100000, 200]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, "**", pattern), recursive=True))


def short(name):
    name = name.split("(")[0]
    for pre in ("void plda::", "plda::"):
        if name.startswith(pre):
            name = name[len(pre):]
    return name[:70]


for f in find("*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print("  %-70s calls=%6s total_ms=%10.3f avg_us=%10.2f pct=%5s" % (
            short(r.get("Name", "")), r.get("Calls"), float(r.get("TotalDurationNs", 0)) / 1e6,
            float(r.get("AverageNs", 0)) / 1e3, r.get("Percentage")))

# steady-state duration of the dominant kernel from the kernel trace: the first launch of a process pays
# code load and clock ramp (34.0 vs 31.2 ms in round 1) and is left out of the average
import json
steady = {}
for f in find("*kernel_trace.csv"):
    if "pmc" in f:
        continue
    dur = defaultdict(list)
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    print("== steady-state kernel durations (first launch of each kernel excluded):", os.path.relpath(f, out))
    for k, v in sorted(dur.items(), key=lambda kv: -sum(d for _, d in kv[1]))[:12]:
        v.sort()
        ds = [d for _, d in v]
        ss = ds[1:] if len(ds) > 1 else ds
        print("  %-70s launches=%5d first_ms=%9.4f steady_avg_ms=%9.4f min_ms=%9.4f" % (k, len(ds), ds[0], sum(ss) / len(ss), min(ds)))
        if "trials_gemm" in k:
            steady[k] = {"launches": len(ds), "first_ms": ds[0], "steady_avg_ms": sum(ss) / len(ss), "min_ms": min(ds)}
if steady:
    json.dump(steady, open(os.path.join(out, "%s_trials_gemm_durations.json" % tag), "w"), indent=1)

for label, pat in (("FETCH_SIZE", "*fetch*counter_collection.csv"), ("WRITE_SIZE", "*write*counter_collection.csv")):
    for f in find(pat):
        print("== %s: %s" % (label, os.path.relpath(f, out)))
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != label:
                continue
            k = short(r.get("Kernel_Name", ""))
            agg[k][0] += 1
            agg[k][1] += float(r.get("Counter_Value", 0))
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
            kib = v / n
            corr = 2.0 if label == "FETCH_SIZE" else 1.0
            print("  %-70s launches=%6d avg_raw_KiB=%14.1f avg_bytes(corrected x%g)=%.4e" % (k, n, kib, corr, kib * 1024 * corr))


# filtered raw PMC rows of the dominant kernel (small enough to commit under profiles/)
for label, pat in (("fetch", "*fetch*counter_collection.csv"), ("write", "*write*counter_collection.csv")):
    for f in find(pat):
        rows = list(csv.reader(open(f)))
        hdr = rows[0]
        ki = hdr.index("Kernel_Name")
        keep = [hdr] + [r for r in rows[1:] if "trials_gemm" in r[ki]]
        csv.writer(open(os.path.join(out, "%s_pmc_%s_trials_gemm.csv" % (tag, label)), "w")).writerows(keep)


# machine-readable HBM traffic of the dominant kernel, per launch (consumed by bench.py)
names = set()
for f in find("*fetch*counter_collection.csv"):
    names |= {short(r.get("Kernel_Name", "")) for r in csv.DictReader(open(f)) if "trials_gemm" in r.get("Kernel_Name", "")}
traffic = {"shape": shape, "kernel": " + ".join(sorted(names)) or "?", "round": tag}
for label, pat in (("FETCH_SIZE", "*fetch*counter_collection.csv"), ("WRITE_SIZE", "*write*counter_collection.csv")):
    for f in find(pat):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                if r.get("Counter_Name") == label and "trials_gemm" in r.get("Kernel_Name", "")]
        if vals:
            vals = vals[1:] if len(vals) > 1 else vals          # steady state
            traffic[label + "_raw_KiB_avg"] = sum(vals) / len(vals)
            traffic[label + "_launches"] = len(vals)
if "FETCH_SIZE_raw_KiB_avg" in traffic and "WRITE_SIZE_raw_KiB_avg" in traffic:
    traffic["fetch_bytes_corrected_x2"] = traffic["FETCH_SIZE_raw_KiB_avg"] * 1024 * 2
    traffic["write_bytes"] = traffic["WRITE_SIZE_raw_KiB_avg"] * 1024
    traffic["hbm_bytes_per_launch"] = traffic["fetch_bytes_corrected_x2"] + traffic["write_bytes"]
    traffic["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KiB units; FETCH_SIZE doubled "
                       "(gfx950 reports 1/2 of a wide coalesced stream, MI355X_MICROARCH.md section HBM)")
    json.dump(traffic, open(os.path.join(out, "%s_traffic_trials_gemm.json" % tag), "w"), indent=1)
