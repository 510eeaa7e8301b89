"""The grouped EM's two closed forms (PLDA_EM_VARIANT 3 moments / 4 rows; 0 = chosen by shape) on skewed speaker counts at several
dimensions: statistics / EM / GetOutput ms.   usage: python scripts/em_forms_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plda_amd import MPlda
dev = torch.device("cuda", 0)
for D, N, K in ((512, 300000, 10000), (256, 200000, 7200), (64, 100000, 5000), (128, 100000, 5000)):
    rng = np.random.default_rng(D)
    nk = rng.integers(5, 61, K).astype(np.float64)
    nk = np.maximum(1, np.floor(nk * N / nk.sum())).astype(np.int64)
    while nk.sum() > N: nk[np.argmax(nk)] -= 1
    nk[: N - nk.sum()] += 1
    y = np.repeat(np.arange(K), nk)
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.rand((N, D), dtype=torch.float64, device=dev, generator=g)
    dy = torch.from_numpy(y.astype(np.int64)).to(dev)
    for v in ("0", "3", "4"):
        os.environ["PLDA_EM_VARIANT"] = v
        eng = MPlda(0)
        best = None
        for _ in range(3):
            eng.fit_dev(X.data_ptr(), N, D, dy.data_ptr(), K, 10); eng.synchronize()
            ft = eng.fit_timings()
            if best is None or ft["em_ms"] < best["em_ms"]: best = ft
        print("D=%d N=%d K=%d G=%d variant %s (%s): stats %.2f EM %.2f ms (%.3f ms/iter) GetOutput %.2f" % (
            D, N, K, len(np.unique(nk)), v, eng.fit_plan()["form"], best["stats_ms"], best["em_ms"], best["em_ms"] / 10, best["output_ms"]), flush=True)
        del eng
