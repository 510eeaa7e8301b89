"""GPU soak of the direct eigensolver: many decompositions in a row (cooperative tridiagonalisation with its
spin-wait all-gather), checking every result and that the block-Jacobi fallback was never needed.
    python scripts/eig_soak.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from plda_amd import MPlda

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
eng = MPlda(0)
rng = np.random.default_rng(0)
bad = 0
t0 = time.time()
for n in (200, 512, 1000, 161, 257):
    A = rng.standard_normal((n, n)); G = A + A.T
    ref = np.linalg.eigvalsh(G)[::-1]
    worst = 0.0
    for r in range(reps if n <= 512 else max(reps // 4, 1)):
        lam, V, used = eng.sym_eig(G, 0)
        err = np.abs(lam - ref).max() / np.abs(ref).max()
        worst = max(worst, err)
        if used != 2 or err > 1e-12:
            bad += 1
    print("n=%d: worst eigenvalue error %.2e, failures so far %d, %.1f s" % (n, worst, bad, time.time() - t0), flush=True)
print("soak done, failures:", bad)
sys.exit(1 if bad else 0)
