"""fit on device-resident data when the speakers have DIFFERENT utterance counts (real data; SURVEY.md section 8d's second C2
labelling: n_k in [5, 60]): statistics / EM / GetOutput ms and EM iterations/s against the number G of distinct counts.
usage: [PLDA_EM_VARIANT=3|4] python scripts/fit_groups_probe.py   (3 / 4: the moment / the row form of the grouped EM always)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plda_amd import MPlda

dev = torch.device("cuda", 0)
N, D, K = 100000, 200, 5000
rng = np.random.default_rng(2)
X = torch.from_numpy(rng.random((N, D))).to(dev)
print("PLDA_EM_VARIANT =", os.environ.get("PLDA_EM_VARIANT", "0 (form chosen by shape)"))
for name, lo, hi in (("uniform 20", 20, 20), ("n_k in [15, 25]", 15, 25), ("n_k in [5, 60]", 5, 60), ("n_k in [1, 200]", 1, 200)):
    if lo == hi:
        y = np.arange(N) % K
    else:
        nk = rng.integers(lo, hi + 1, K).astype(np.float64)
        nk = np.maximum(1, np.floor(nk * N / nk.sum())).astype(np.int64)
        nk[: N - nk.sum()] += 1                       # (the rows the rounding left over, one each: no giant class)
        y = np.repeat(np.arange(K), nk)[:N]
        if y.shape[0] < N:
            y = np.concatenate([y, np.zeros(N - y.shape[0], np.int64)])
    G = len(np.unique(np.bincount(y, minlength=K)))
    dy = torch.from_numpy(y.astype(np.int64)).to(dev)
    eng = MPlda(0)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        t0 = time.perf_counter(); eng.fit_dev(X.data_ptr(), N, D, dy.data_ptr(), K, 10); eng.synchronize(); w = time.perf_counter() - t0
        ft = eng.fit_timings()
        if best is None or w < best[0]:
            best = (w, ft)
    w, ft = best
    print("%-18s G = %3d: fit %.2f ms wall | statistics %.2f | EM %.2f ms = %.0f iterations/s | GetOutput %.2f" % (
        name, G, w * 1e3, ft["stats_ms"], ft["em_ms"], ft["iters"] / (ft["em_ms"] * 1e-3), ft["output_ms"]))
