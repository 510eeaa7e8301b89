"""Timing of the K4 kernels over the row count (fixed cost against cost per row): the product kernels (0), the
register-resident-T arm (PLDA_TRANSFORM_VARIANT=6) and its timing arms (10: no X DMA, 11: no MFMAs, 12: no stores, 13: all three)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from plda_amd import MPlda

dev = torch.device("cuda", 0)
D = 200
rng = np.random.default_rng(1)
q, _ = np.linalg.qr(rng.standard_normal((D, D)))
model = (rng.random(D), q * (0.5 + rng.random(D))[:, None], np.sort(rng.random(D) * 3 + 0.01)[::-1].copy())
rows = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "100000,200000,400000,800000".split(","))]
X = torch.randn((max(rows), D), dtype=torch.float64, device=dev)
O = torch.empty((max(rows), D), dtype=torch.float64, device=dev)
for variant in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "6", "10", "11", "12", "13"]):
    os.environ["PLDA_TRANSFORM_VARIANT"] = variant
    eng = MPlda(0)
    eng.set_model(*model)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    line = "variant %2s:" % variant
    for r in rows:
        for _ in range(3):
            eng.transform_rows_dev(X.data_ptr(), r, D, None, 2, O.data_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.transform_rows_dev(X.data_ptr(), r, D, None, 2, O.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += "  R=%7d %7.4f ms (%.3f of fp64 MFMA)" % (r, ms, 2.0 * r * D * D / ms / 1e9 / 78.6)
    print(line, flush=True)
    eng.set_stream(None)
