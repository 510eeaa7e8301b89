"""Throughput of the sparse trial list (plda_score_pairs: scoring/scorePLDA.py:302-318's loop over a trials file) through the
host API, and its kernel alone (rocprofv3 / the engine's trace spans): P pairs drawn from M enrol models x Nt test vectors.
usage: python scripts/trial_list_probe.py [D]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plda_amd import MPlda   # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(0)
q, _ = np.linalg.qr(rng.standard_normal((D, D)))
eng = MPlda(0)
eng.set_model(rng.random(D), q * (1.0 + rng.random(D))[:, None], np.sort(rng.random(D) * 4.0 + 0.05)[::-1].copy())
for m, nt, p in ((1000, 10000, 10 ** 5), (5000, 50000, 10 ** 6), (5000, 50000, 10 ** 7), (40000, 200000, 10 ** 7)):
    U, V = rng.standard_normal((m, D)), rng.standard_normal((nt, D))
    counts = rng.integers(1, 6, m).astype(np.int32)
    e, t = rng.integers(0, m, p), rng.integers(0, nt, p)
    if p >= 10 ** 6:                      # trial files list a model's trials together
        e = np.sort(e)
    eng.score_trials((counts, U), (1, V), e[:1000], t[:1000])
    eng.trace_enable(True); eng.trace_read(reset=True)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); out = eng.score_trials((counts, U), (1, V), e, t); ts.append(time.perf_counter() - t0)
    spans = {s["name"]: s["ms"] / s["calls"] for s in eng.trace_read(reset=True)}
    eng.trace_enable(False)
    print("D=%d  %d x %d, %.0e pairs: %.1f ms per call = %.3g pairs/s through the host API; spans (ms): %s" % (
        D, m, nt, p, min(ts) * 1e3, p / min(ts), {k: round(v, 3) for k, v in spans.items()}))
