"""K4: rows/s and fp64-MFMA fraction of transform_rows_dev (mean-of-n rows -> PLDA space -> length norm) at the
BASELINE shapes.  2 N D^2 flop on v_mfma_f64_16x16x4_f64 (78.6 TFLOP/s), 16 N D bytes (in + out)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from plda_amd import MPlda  # noqa: E402

dev = torch.device("cuda", 0)
res = []
shapes = [("C2", 100_000, 200), ("C3", 1_000_000, 512), ("C4", 1_200_000, 256)]
if os.environ.get("PLDA_TRANSFORM_VARIANT") != "1":       # odd shapes too: column tails, row tails, K tails
    shapes += [("odd", 100_003, 129), ("odd", 50_001, 300), ("odd", 70_000, 77), ("odd", 33_333, 385)]
for (name, N, D) in shapes:
    rng = np.random.default_rng(1)
    q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    eng = MPlda(0)
    eng.set_model(rng.random(D), q * (1.0 + rng.random(D))[:, None], np.sort(rng.random(D) * 4)[::-1].copy())
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        eng.set_stream(stream.cuda_stream)
        X = torch.rand((N, D), dtype=torch.float64, device=dev)
        U = torch.empty((N, D), dtype=torch.float64, device=dev)
        for _ in range(2):
            eng.transform_rows_dev(X.data_ptr(), N, D, None, 1, U.data_ptr())
        stream.synchronize()
        eng.trace_enable(True)
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            eng.transform_rows_dev(X.data_ptr(), N, D, None, 1, U.data_ptr())
        stream.synchronize()
        dt = (time.perf_counter() - t0) / reps
        # numpy fp64 on a sample of rows (first, last, random)
        pick = np.unique(np.concatenate([np.arange(8), np.arange(N - 8, N), rng.integers(0, N, 200)]))
        m = eng.get_model()
        xs = X[torch.from_numpy(pick).to(dev)].cpu().numpy()
        tt = xs @ m["transform"].T + m["offset"] if "offset" in m else (xs - m["mean"]) @ m["transform"].T
        ref = tt * np.sqrt(D / (tt * tt / (m["psi"] + 1.0)).sum(1))[:, None]
        got = U[torch.from_numpy(pick).to(dev)].cpu().numpy()
        err = float(np.abs(got - ref).max() / np.abs(ref).max())
        spans = eng.trace_read(reset=True) if hasattr(eng, "trace_read") else None
    res.append({"config": name, "N": N, "D": D, "ms": dt * 1e3, "rows_per_s": N / dt,
                "TFLOPps": 2.0 * N * D * D / dt / 1e12, "frac_fp64_mfma_78.6": 2.0 * N * D * D / dt / 78.6e12,
                "GBps": 16.0 * N * D / dt / 1e9, "max_rel_err_vs_numpy": err,
                "event_ms": (spans[0]["ms"] / spans[0]["calls"]) if spans else None})
    del X, U
print(json.dumps(res, indent=1, default=str))
