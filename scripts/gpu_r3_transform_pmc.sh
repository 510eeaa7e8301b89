#!/bin/bash
# SQ / GRBM counters of the transform kernel (K4) at a given shape: N D (default 98304 200 = three whole rounds)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
N=${1:-98304}; D=${2:-200}
OUT=/tmp/tfpmc; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/tf_target.py <<PY
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from plda_amd import MPlda
dev = torch.device("cuda", 0)
N, D = $N, $D
rng = np.random.default_rng(1)
q, _ = np.linalg.qr(rng.standard_normal((D, D)))
eng = MPlda(0)
eng.set_model(rng.random(D), q * (1.0 + rng.random(D))[:, None], np.sort(rng.random(D) * 4)[::-1].copy())
eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
X = torch.rand((N, D), dtype=torch.float64, device=dev)
U = torch.empty((N, D), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for _ in range(6):
    eng.transform_rows_dev(X.data_ptr(), N, D, None, 1, U.data_ptr())
torch.cuda.synchronize()
PY
for SET in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $SET | tr ' ' '_' | cut -c1-30)
  ( cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/$tag -o p -- python /tmp/tf_target.py > /dev/null 2> $OUT/$tag.err )
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "transform_fused" in r["Kernel_Name"] or "transform_dma" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob("$OUT/$tag/**/*kernel_trace.csv", recursive=True):
    dur += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "transform_fused" in r["Kernel_Name"] or "transform_dma" in r["Kernel_Name"]]
for k, v in agg.items():
    print("%-28s launches=%d  steady-state avg=%.6e" % (k, len(v), sum(v[1:]) / max(len(v) - 1, 1)))
print("kernel us (this pass):", ["%.1f" % d for d in dur])
PY
done
