"""K2 (scatter SYRK of the statistics pass) against the number of rows at D = 512 and 200: is its efficiency a matter of
memory (rows that fit the 256 MB MALL against rows that do not) or of the kernel?  HIP-event span of the stage.
(Round 3: 0.80 - 0.86 of the fp64 peak by the algorithmic count at D = 512 from 100k rows up, whatever the size: the
kernel; at D = 200 a fixed ~60 us of partial slabs, reduction and launch under a per-row cost that tends to 0.83.)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from plda_amd import MPlda
dev = torch.device("cuda", 0)
eng = MPlda(0)
eng.trace_enable(True)
for D in (512, 200):
    for N in (20000, 50000, 100000, 400000, 1000000):
        K = max(2, N // 100)
        X = torch.rand((N, D), dtype=torch.float64, device=dev)
        y = (torch.arange(N, device=dev) % K).to(torch.int64)
        torch.cuda.synchronize()
        for rep in range(3):
            eng.trace_read()
            eng.fit_stats_dev(X.data_ptr(), N, D, y.data_ptr(), K)
            torch.cuda.synchronize()
        sp = {s["name"]: s for s in eng.trace_read()}
        s = sp["fit.scatter_syrk (K2)"]
        print(D, N, "K2 %.3f ms  %.1f TFLOP/s algorithmic  (%.2f of 78.6)" % (s["ms"] / s["calls"], s["work"] / s["calls"] / (s["ms"] / s["calls"]) / 1e9, s["work"] / s["calls"] / (s["ms"] / s["calls"]) / 1e9 / 78.6))
        del X, y
