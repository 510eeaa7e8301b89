"""GPU: wall time of GetOutput / of a whole fit with the direct eigensolver against block Jacobi
(PLDA_EIG_VARIANT=1), at the C2 and C3 fit shapes.  python scripts/eig_probe.py [D ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from plda_amd import MPlda

def run(D, N, K, variant):
    os.environ["PLDA_EIG_VARIANT"] = str(variant)
    eng = MPlda(0)
    dev = torch.device("cuda", 0)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream or None)
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.rand((N, D), dtype=torch.float64, device=dev, generator=g)
    y = (torch.arange(N, device=dev) % K).to(torch.int64)
    best = None
    for rep in range(4):
        eng.fit_dev(X.data_ptr(), N, D, y.data_ptr(), K, 10)
        torch.cuda.synchronize()
        ft = eng.fit_timings()
        if best is None or ft["output_ms"] < best["output_ms"]:
            best = ft
    m = eng.get_model()
    return best, m

for D in [int(a) for a in sys.argv[1:]] or [200, 512]:
    N, K = (100000, 5000) if D <= 256 else (1000000, 10000)
    a, ma = run(D, N, K, 0)
    b, mb = run(D, N, K, 1)
    dpsi = np.abs(ma["psi"] - mb["psi"]).max() / np.abs(mb["psi"]).max()
    print("D=%d  direct: stats %.3f em %.3f output %.3f ms | jacobi: output %.3f ms | psi rel diff %.2e" %
          (D, a["stats_ms"], a["em_ms"], a["output_ms"], b["output_ms"], dpsi), flush=True)
