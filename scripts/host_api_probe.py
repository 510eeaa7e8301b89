#!/usr/bin/env python
"""Where the time of the host-pointer API goes (NumPy in, NumPy out): cProfile of fit / transform at the C2 shape and the
library's own stage spans; run on the GPU box."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plda_amd import MPlda   # noqa: E402

rng = np.random.default_rng(2)
N, D, K = 100000, 200, 5000
X = rng.random((N, D))
y = (np.arange(N) % K).astype(np.uint64)
eng = MPlda(0)
eng.fit(X, y, 10)
eng.transform(X, y)
for name, fn in [("fit", lambda: eng.fit(X, y, 10)), ("transform", lambda: eng.transform(X, y)),
                 ("transform_array", lambda: eng.transform_array(X, 1))]:
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    print("== %s: %.2f ms" % (name, dt * 1e3))
    eng.trace_enable(True); eng.trace_read(reset=True)
    pr = cProfile.Profile(); pr.enable(); fn(); pr.disable()
    for sp in eng.trace_read(reset=True):
        print("   span %-40s %.3f ms" % (sp["name"], sp["ms"]))
    eng.trace_enable(False)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(8)
