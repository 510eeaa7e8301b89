#!/bin/bash
# round 4: SQ counters of the two K4 kernels (product, register-resident-T arm) at 800k x 200
cd /root/repo
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4/k4pmc
rm -rf $OUT; mkdir -p $OUT
for V in 0 6; do
for SET in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  tag=$(echo $SET | tr ' ' '_' | cut -c1-30)
  ( cd /tmp && PLDA_TRANSFORM_VARIANT=$V timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/v${V}_$tag -o p -- python /root/repo/scripts/transform_stream_probe.py 800000 $V > /dev/null 2> $OUT/v${V}_$tag.err )
done
done
python - <<'PY' | tee gpurun_out/r4/k4pmc_summary.txt
import csv, glob, collections, os
for V in (0, 6):
    agg = collections.defaultdict(list)
    for f in glob.glob("/root/repo/gpurun_out/r4/k4pmc/v%d_*/**/*counter_collection.csv" % V, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "transform_fused_kernel<13, 1" in k or "transform_treg_kernel" in k:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("PLDA_TRANSFORM_VARIANT=%d (800 000 x 200 rows, per launch, median over %d launches):" % (V, max(len(v) for v in agg.values()) if agg else 0))
    for k in sorted(agg):
        v = sorted(agg[k]); print("  %-28s %.4e" % (k, v[len(v) // 2]))
    if "SQ_INSTS_MFMA" in agg and "GRBM_GUI_ACTIVE" in agg:
        mf = sorted(agg["SQ_INSTS_MFMA"])[len(agg["SQ_INSTS_MFMA"]) // 2]; ga = sorted(agg["GRBM_GUI_ACTIVE"])[len(agg["GRBM_GUI_ACTIVE"]) // 2]
        print("  MFMA-busy by instruction count: %.3f (MFMAs x 64 cycles / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs)" % (mf * 64 / 1024 / (ga / 8)))
PY
find gpurun_out/r4/k4pmc -name "*.csv" -size +200k -delete; find gpurun_out/r4/k4pmc -name "*.db" -delete
