"""K4 A/B in ONE process, interleaved: PLDA_TRANSFORM_VARIANT 0 (product: whole rounds + a tail launch of small blocks),
2 (no tail launch: every row in the persistent main launch, the round-2 shape), 1 (general GEMM + separate length-norm pass),
at the BASELINE shapes, a few odd ones and small calls.  HIP-event time of the stage span; fp64-MFMA fraction of 78.6 TFLOP/s."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from plda_amd import MPlda  # noqa: E402

dev = torch.device("cuda", 0)
shapes = [("C2", 100_000, 200), ("C4", 1_200_000, 256), ("C3", 1_000_000, 512), ("odd", 100_003, 129), ("odd", 50_001, 300),
          ("small", 2_000, 200), ("small", 300, 200), ("C2-mixed-n", 100_000, 200)]
variants = [v for v in os.environ.get("SWEEP_VARIANTS", "0,2,1").split(",")]
res = []
for (name, N, D) in shapes:
    rng = np.random.default_rng(1)
    q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    engs = {}
    for v in variants:
        os.environ["PLDA_TRANSFORM_VARIANT"] = v
        e = MPlda(0)
        e.set_model(rng.random(D), q * (1.0 + rng.random(D))[:, None], np.sort(rng.random(D) * 4)[::-1].copy())
        engs[v] = e
    os.environ.pop("PLDA_TRANSFORM_VARIANT")
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        X = torch.rand((N, D), dtype=torch.float64, device=dev)
        n = torch.randint(1, 6, (N,), dtype=torch.int32, device=dev) if name.endswith("mixed-n") else None
        outs = {v: torch.empty((N, D), dtype=torch.float64, device=dev) for v in variants}
        ms = {v: [] for v in variants}
        for e in engs.values():
            e.set_stream(stream.cuda_stream)
            e.trace_enable(True)
        for rep in range(4):
            for v in variants:
                e = engs[v]
                for _ in range(5):
                    e.transform_rows_dev(X.data_ptr(), N, D, n.data_ptr() if n is not None else None, 0 if n is not None else 1,
                                         outs[v].data_ptr())
                stream.synchronize()
                sp = e.trace_read(reset=True)
                if rep:
                    ms[v].append(sp[0]["ms"] / sp[0]["calls"])
        base = outs[variants[-1]]
        row = {"config": name, "N": N, "D": D}
        for v in variants:
            t = float(np.median(ms[v]))
            row["v%s_ms" % v] = round(t, 4)
            row["v%s_frac" % v] = round(2.0 * N * D * D / (t * 1e-3) / 78.6e12, 4)
            row["v%s_vs_last_max_rel" % v] = float(((outs[v] - base).abs().max() / base.abs().max()).item())
        res.append(row)
        print(json.dumps(row), flush=True)
    del X, outs
