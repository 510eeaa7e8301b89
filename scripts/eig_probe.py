"""GPU: wall time of GetOutput inside a fit, per eigensolver variant (PLDA_EIG_VARIANT: 0 default dispatch,
1 block Jacobi, 2 register tridiagonalisation, 3 cooperative-rows tridiagonalisation).
    python scripts/eig_probe.py D[:variant] ...        e.g.  200:0 200:1 512:0"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from plda_amd import MPlda

dev = torch.device("cuda", 0)
for spec in sys.argv[1:] or ["200:0", "200:1", "512:0", "512:1"]:
    D, variant = (spec.split(":") + ["0"])[:2]
    D = int(D)
    os.environ["PLDA_EIG_VARIANT"] = variant
    N, K = (100000, 5000) if D <= 256 else (200000, 5000)
    eng = MPlda(0)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.rand((N, D), dtype=torch.float64, device=dev, generator=g)
    y = (torch.arange(N, device=dev) % K).to(torch.int64)
    torch.cuda.synchronize()
    best = None
    for rep in range(4):
        eng.fit_dev(X.data_ptr(), N, D, y.data_ptr(), K, 10)
        torch.cuda.synchronize()
        ft = eng.fit_timings()
        if best is None or ft["output_ms"] < best["output_ms"]:
            best = ft
    print("D=%d variant %s: stats %.3f em %.3f output %.3f ms" % (D, variant, best["stats_ms"], best["em_ms"], best["output_ms"]), flush=True)
    del eng, X, y
