"""Whole fits (statistics, EM in both grouped forms, GetOutput) and norm() on fresh handles, many times, with another handle's
problem run and freed in between and the chip busy on a side stream every other run: W, B, psi, T^T T and the z-norm
statistics must equal the first run's bit for bit.
usage: python scripts/stress_fit.py [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plda_amd import MPlda

dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
side = torch.cuda.Stream(device=dev)
big = torch.rand((700, 700), dtype=torch.float32, device=dev)
bad = 0
cases = [("rows D=200", 20000, 200, 1000, (5, 60)), ("moments D=200", 20000, 200, 1000, None), ("rows D=72", 6000, 72, 400, (2, 30)),
         ("rows D=288", 9000, 288, 300, (10, 50)), ("moments D=512", 12000, 512, 200, None)]
for name, n, d, k, skew in cases:
    rng = np.random.default_rng(n + d)
    if skew:
        nk = rng.integers(skew[0], skew[1] + 1, k)
        y = np.repeat(np.arange(k), nk)[:n]
        n = y.shape[0]
        k = int(y.max()) + 1
    else:
        y = np.arange(n) % k
    x = rng.random((n, d)) + 0.4 * rng.standard_normal((k, d))[y]
    dX = torch.from_numpy(x).to(dev); dy = torch.from_numpy(y.astype(np.int64)).to(dev)
    bkg = torch.from_numpy(rng.random((3000, d))).to(dev)
    first = None
    for r in range(reps):
        other = MPlda(0)
        n2, d2, k2 = int(rng.integers(500, 8000)), int(rng.choice([64, 200, 256, 384])), int(rng.integers(5, 200))
        y2 = rng.integers(0, k2, n2); y2[:k2] = np.arange(k2)
        x2 = torch.from_numpy(1e3 * rng.standard_normal((n2, d2))).to(dev); dy2 = torch.from_numpy(y2.astype(np.int64)).to(dev)
        other.fit_dev(x2.data_ptr(), n2, d2, dy2.data_ptr(), k2, 2)
        other.synchronize()
        del other, x2, dy2
        eng = MPlda(0)
        eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        if r % 2 == 1:
            with torch.cuda.stream(side):
                for _ in range(int(rng.integers(1, 6))):
                    big2 = big @ big
        eng.fit_dev(dX.data_ptr(), n, d, dy.data_ptr(), k, 6)
        torch.cuda.synchronize()
        it = eng.fit_internals(); g = eng.get_model()
        models = torch.from_numpy(np.ascontiguousarray(g["transform"] @ x[:500].T).T.copy()).to(dev)
        zm = torch.empty(500, dtype=torch.float64, device=dev); zs = torch.empty(500, dtype=torch.float64, device=dev)
        eng.znorm_stats_dev(bkg.data_ptr(), 3000, 3000, d, models.data_ptr(), 500, zm.data_ptr(), zs.data_ptr())
        torch.cuda.synchronize()
        cur = dict(W=it["W"], B=it["B"], psi=g["psi"], TtT=g["transform"].T @ g["transform"], zm=zm.cpu().numpy(), zs=zs.cpu().numpy())
        form = eng.fit_plan()["form"]
        eng.set_stream(None)
        del eng
        if first is None:
            first = cur
        else:
            for key in cur:
                if not np.array_equal(first[key], cur[key]):
                    bad += 1
                    print("%s rep %d: %s differs, max rel %.3g" % (name, r, key, np.abs(cur[key] - first[key]).max() / np.abs(first[key]).max()), flush=True)
    print("%s (N=%d K=%d, EM form %s): %d runs" % (name, n, k, form, reps), flush=True)
print("mismatches:", bad)
