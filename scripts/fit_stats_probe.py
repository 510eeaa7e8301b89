"""A/B of the statistics pass (K1a sort, K1 centroids, K2 scatter SYRK): symmetric kernel vs the general GEMM
(PLDA_GEMM64_VARIANT=2), with parity of the scatter against a float64 NumPy restatement on a subsample-free
small case and against each other at full size.  usage: fit_stats_probe.py [C2|C3]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
N, D, K = (100000, 200, 5000) if cfg == "C2" else (1000000, 512, 10000)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(2)
dX = torch.rand((N, D), dtype=torch.float64, device=dev, generator=g)
dy = torch.arange(N, device=dev, dtype=torch.int64) % K
res = {}
for variant in ("0", "2"):
    os.environ["PLDA_GEMM64_VARIANT"] = variant
    from plda_amd import MPlda
    e = MPlda(0)
    for _ in range(3):
        e.fit_stats_dev(dX.data_ptr(), N, D, dy.data_ptr(), K)
    t = []
    for _ in range(5):
        e.fit_stats_dev(dX.data_ptr(), N, D, dy.data_ptr(), K)
        t.append(e.fit_timings()["stats_ms"])
    S = torch.empty((D, D), dtype=torch.float64, device=dev)
    e.fit_get_stats_dev(None, None, S.data_ptr())
    res[variant] = (min(t), S.cpu().numpy())
    print("GEMM64_VARIANT=%s: statistics pass %.3f ms (min of 5)" % (variant, min(t)))
a, b = res["0"][1], res["2"][1]
print("scatter: symmetric kernel vs general kernel: max rel diff %.3e; asymmetry %.3e" % (np.abs(a - b).max() / np.abs(b).max(), np.abs(a - a.T).max()))
