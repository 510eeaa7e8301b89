"""Wall time of the reference-sized calls through liblda.PLDA (NumPy in / out): the shapes of the reference's own tests and
README (tests/pldatest.py:10-11: 2000 x 10, 10 speakers; README.md:50-113 / BASELINE C1: 500 x 200, 2 speakers) -- what a
user switching from the reference sees first.  Launch-latency territory: reported as ms per call, min of 5."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liblda import PLDA   # noqa: E402


def best(fn, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


for name, n, d, k in (("pldatest 2000 x 10, 10 speakers", 2000, 10, 10), ("C1 500 x 200, 2 speakers", 500, 200, 2),
                      ("4000 x 200, 100 speakers", 4000, 200, 100)):
    rng = np.random.default_rng(1)
    x = rng.random((n, d))
    y = rng.integers(0, k, n).astype(np.uint64)
    p = PLDA()
    p.fit(x, y, 10)
    enrol = p.transform(x[: n // 2], y[: n // 2])
    test = p.transform(x[n // 2:], np.arange(n - n // 2, dtype=np.uint64))
    p.norm(x[: n // 4], enrol)
    ids = list(enrol)
    tv = list(test.values())
    print("%-34s fit(10 it) %.2f ms | transform %.2f ms | norm %.2f ms | score() %.1f us | score_matrix %dx%d %.2f ms" % (
        name, best(lambda: p.fit(x, y, 10)), best(lambda: p.transform(x[: n // 2], y[: n // 2])),
        best(lambda: p.norm(x[: n // 4], enrol)),
        best(lambda: [p.score(ids[i % len(ids)], enrol[ids[i % len(ids)]], tv[i % len(tv)]) for i in range(200)]) * 1e3 / 200,
        len(enrol), len(test), best(lambda: p.score_matrix(enrol, test))))
