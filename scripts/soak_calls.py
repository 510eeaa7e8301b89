"""Soak of the per-call state machines of score_matrix_dev (tile-queue counters left at zero by the launch itself, tables of the
last six tile grids, coefficient / bucket-table caches, the polled count set): thousands of calls over a handful of shapes and
count kinds in random order on one handle, every output compared bit for bit with the first output of its (shape, kind).
usage: python scripts/soak_calls.py [calls] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plda_amd import MPlda

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
D = 200
q, _ = np.linalg.qr(rng.standard_normal((D, D)))
eng = MPlda(0)
eng.set_model(rng.random(D), q * (1.0 + rng.random(D))[:, None], np.sort(rng.random(D) * 4.0 + 0.05)[::-1].copy())
stream = torch.cuda.Stream(dev)
eng.set_stream(stream.cuda_stream)
shapes = [(300, 517), (4096, 4096), (6000, 7000), (8192, 8192), (12000, 12000), (11000, 13000), (2048, 60000), (13000, 11000), (700, 130000)]
nmax = 130000
U = torch.randn((nmax, D), dtype=torch.float64, device=dev)
V = torch.randn((nmax, D), dtype=torch.float64, device=dev)
counts = {"mixed5": torch.randint(1, 6, (nmax,), dtype=torch.int32, device=dev), "mixed3": torch.from_numpy(np.array([2, 9, 40], np.int32)[rng.integers(0, 3, nmax)]).to(dev)}
outs = {s: torch.empty(s, dtype=torch.float32, device=dev) for s in shapes}
first = {}
kernels = {}
bad = 0
torch.cuda.synchronize()
for c in range(calls):
    s = shapes[int(rng.integers(0, len(shapes)))]
    kind = str(rng.choice(["u1", "u7", "mixed5", "mixed3"]))
    dn = counts[kind].data_ptr() if kind.startswith("mixed") else None
    nu = {"u1": 1, "u7": 7}.get(kind, 0)
    o = outs[s]
    eng.score_matrix_dev(U.data_ptr(), dn, nu, s[0], V.data_ptr(), s[1], o.data_ptr(), s[1])
    key = (s, kind)
    k = eng.score_last_kernel()
    kernels[k] = kernels.get(k, 0) + 1
    if key not in first:
        eng.synchronize()
        first[key] = o.clone()
    elif c % 7 == 0 or c > calls - 40:
        eng.synchronize()
        if not torch.equal(o, first[key]):
            bad += 1
            print("MISMATCH call", c, key)
eng.synchronize()
print("soak: %d calls over %d (shape, counts) pairs, %d mismatches; kernels %s" % (calls, len(first), bad, kernels))
sys.exit(1 if bad else 0)
