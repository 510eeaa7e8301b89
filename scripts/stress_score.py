"""Trials matrices on fresh handles, many times, the chip busy on a side stream every other run: every matrix (uniform and mixed
enrol counts, z-normed; the three GEMM kernels by size) must equal the first run's bit for bit.
usage: python scripts/stress_score.py [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plda_amd import MPlda

dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
side = torch.cuda.Stream(device=dev)
big = torch.rand((900, 900), dtype=torch.float32, device=dev)
bad = 0
for name, m, nt, d in [("bt4 12k x 40k x 200", 12000, 40000, 200), ("bt2 8k x 8k x 200", 8192, 8192, 200), ("128^2 kernel 3k x 2k x 64", 3000, 2000, 64),
                       ("bt4 ragged 10001 x 50003 x 256", 10001, 50003, 256)]:
    rng = np.random.default_rng(m + nt)
    psi = np.sort(rng.random(d) + 0.05)[::-1].copy()
    U = torch.from_numpy(rng.standard_normal((m, d))).to(dev); V = torch.from_numpy(rng.standard_normal((nt, d))).to(dev)
    cnt = torch.from_numpy(rng.integers(1, 6, m).astype(np.int32)).to(dev)
    zm = torch.from_numpy(rng.standard_normal(m)).to(dev); zs = torch.from_numpy(0.5 + rng.random(m)).to(dev)
    out = torch.empty((m, nt), dtype=torch.float32, device=dev)
    first = {}
    for r in range(reps):
        eng = MPlda(0)
        eng.set_model(np.zeros(d), np.eye(d), psi)
        eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        for key, dn, nu, z in (("uniform", None, 1, False), ("mixed", cnt, 0, False), ("mixed+znorm", cnt, 0, True)):
            if r % 2 == 1:
                with torch.cuda.stream(side):
                    for _ in range(int(rng.integers(1, 5))):
                        big2 = big @ big
            out.fill_(float("nan"))
            eng.score_matrix_dev(U.data_ptr(), dn.data_ptr() if dn is not None else None, nu, m, V.data_ptr(), nt, out.data_ptr(), nt,
                                 zm.data_ptr() if z else None, zs.data_ptr() if z else None)
            torch.cuda.synchronize()
            if key not in first:
                first[key] = out.clone()
                assert torch.isfinite(first[key]).all()
            elif not torch.equal(first[key], out):
                bad += 1
                print("%s rep %d %s: differs in %d elements" % (name, r, key, int((first[key] != out).sum())), flush=True)
        eng.set_stream(None)
        del eng
    print("%s: %d runs x 3 kinds, kernel %s" % (name, reps, "?"), flush=True)
print("mismatches:", bad)
