"""Latency of the reference-API one-trial call PLDA.score() (pldamodule.cpp:258-277) through the GPU."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liblda import PLDA
rng = np.random.default_rng(0)
n, d, k = 4000, 200, 100
y = (np.arange(n) % k).astype(np.uint64)
x = rng.random((n, d))
p = PLDA()
p.fit(x, y, 3)
enrol = p.transform(x[:1000], y[:1000])
test = p.transform(x[1000:1200], np.arange(200, dtype=np.uint64))
p.norm(x[2000:2300], enrol)
ids = list(enrol)
for reps in (200, 2000):
    t0 = time.perf_counter()
    for r in range(reps):
        p.score(ids[r % len(ids)], enrol[ids[r % len(ids)]], test[r % 200])
    dt = (time.perf_counter() - t0) / reps
print("score(): %.1f us per call" % (dt * 1e6))
t0 = time.perf_counter(); S = p.score_matrix(enrol, test); dt = time.perf_counter() - t0
print("score_matrix %dx%d: %.2f ms (%.3f us per trial incl. host packing)" % (S.shape[0], S.shape[1], dt * 1e3, dt * 1e6 / S.size))
