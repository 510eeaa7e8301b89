"""Host-side cost of the z-norm bookkeeping through liblda.PLDA at C5-like sizes (many enrol models): norm() and a z-normalised
score_matrix() call, with the library's own spans beside the wall time.  usage: python scripts/znorm_api_probe.py [models]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liblda import PLDA   # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
d, nt, nb = 200, 5000, 20000
rng = np.random.default_rng(0)
x = rng.random((4000, d)); y = (np.arange(4000) % 100).astype(np.uint64)
p = PLDA()
p.fit(x, y, 5)
xe = rng.random((m, d))
enrol = p.transform(xe, np.arange(m, dtype=np.uint64))
test = p.transform(rng.random((nt, d)), np.arange(nt, dtype=np.uint64))
cohort = rng.random((nb, d))
t0 = time.perf_counter(); p.norm(cohort, enrol); t_norm = time.perf_counter() - t0
ts = []
for _ in range(3):
    t0 = time.perf_counter(); S = p.score_matrix(enrol, test); ts.append(time.perf_counter() - t0)
t0 = time.perf_counter(); S0 = p.score_matrix(enrol, test, znorm=False); t_plain = time.perf_counter() - t0
print("%d models x %d tests, D = %d: norm(%d cohort rows) %.1f ms | score_matrix z-normed %.1f ms | without z-norm %.1f ms" % (
    m, nt, d, nb, t_norm * 1e3, min(ts) * 1e3, t_plain * 1e3))
