"""Randomised parity sweep, part 3: medium-size fits (many speakers, many distinct counts) vs the C oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_data          # noqa: E402
from oracle import binding as ob        # noqa: E402
from plda_amd import MPlda              # noqa: E402

ob.build()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # second argument: offset of every seed (fresh cases)
fails = []
for case in range(ncases):
    rng = np.random.default_rng(300 + case + seed0)
    d = int(rng.choice([48, 64, 100, 128, 200, 260]))
    k = int(rng.integers(50, 700))
    n = int(rng.integers(3 * k, 40 * k))
    iters = int(rng.integers(1, 6))
    x, y = make_data(400 + case + seed0, n, d, k, skew=True, scale_between=float(rng.choice([0.0, 0.3, 1.0])))
    eng = MPlda(0)
    t0 = time.perf_counter(); eng.fit(x, y, iters); tg = time.perf_counter() - t0
    t0 = time.perf_counter(); ref = ob.fit(x, y, iters); tc = time.perf_counter() - t0
    g = eng.get_model(); it = eng.fit_internals()
    e_psi = np.abs(g["psi"] - ref["psi"]).max() / max(ref["psi"].max(), 1e-12)
    e_w = np.abs(it["W"] - ref["W"]).max() / np.abs(ref["W"]).max()
    e_b = np.abs(it["B"] - ref["B"]).max() / max(np.abs(ref["B"]).max(), 1e-300)
    groups = len(np.unique(np.bincount(y.astype(np.int64))))
    line = "case %d N=%d D=%d K=%d groups=%d iters=%d: psi %.1e W %.1e B %.1e (gpu %.3fs, oracle %.1fs)" % (
        case, n, d, k, groups, iters, e_psi, e_w, e_b, tg, tc)
    print(line, flush=True)
    if not (e_psi < 1e-8 and e_w < 1e-8 and e_b < 1e-8):
        fails.append(line)
print("%d cases, %d failures" % (ncases, len(fails)))
