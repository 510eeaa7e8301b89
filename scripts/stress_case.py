"""Re-run one case of scripts/stress_parity.py with the stage-by-stage errors (hunting tool).
Usage: python scripts/stress_case.py CASE [SEED0]   |   python scripts/stress_case.py shape N D K ITERS BETWEEN
PLDA_STRESS_TRUTH=1 adds the distance of every W, B (device and C oracle) from the same EM in x87 extended
precision (oracle/plda_oracle_np.py:fit_wb_longdouble; seconds to minutes of NumPy at D >= 200)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_data          # noqa: E402
from oracle import binding as ob, plda_oracle_np as onp        # noqa: E402
from plda_amd import MPlda              # noqa: E402

ob.build()
if sys.argv[1] == "shape":
    n, d, k, iters = (int(v) for v in sys.argv[2:6])
    between, skew, case, seed0 = float(sys.argv[6]), False, 0, 0
else:
    case = int(sys.argv[1]); seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(1000 + case + seed0)
    d = int(rng.choice([1, 2, 3, 5, 8, 17, 31, 32, 33, 64, 100, 129, 200, 257, 300]))
    k = int(rng.integers(2, 40))
    n = int(max(k * 2, rng.integers(k + 1, 40 * k)))
    skew = bool(rng.integers(0, 2))
    iters = int(rng.integers(0, 7))
    between = float(rng.choice([0.0, 0.2, 1.0]))
x, y = make_data(5000 + case + seed0, n, d, k, skew=skew, scale_between=between)
print("case %d: N=%d D=%d K=%d skew=%s iters=%d between=%.1f" % (case, n, d, k, skew, iters, between))


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


truth = os.environ.get("PLDA_STRESS_TRUTH") == "1"


def dist(a, t):
    return float(np.abs(a.astype(np.longdouble) - t).max() / np.abs(t).max())


for it in range(max(iters - 2, 0), iters + 1):
    ref = ob.fit(x, y, it)
    line = "iters=%d cond(W)=%.2e cond(B)=%.2e" % (it, np.linalg.cond(ref["W"]), np.linalg.cond(ref["B"]))
    if truth:
        Wt, Bt = onp.fit_wb_longdouble(x, np.unique(y, return_inverse=True)[1], it)
        line += " | C oracle vs extended precision: W %.1e B %.1e" % (dist(ref["W"], Wt), dist(ref["B"], Bt))
    for env in ({}, {"PLDA_EM_VARIANT": "1"}):
        for kk in ("PLDA_EM_VARIANT", "PLDA_EIG_VARIANT"):
            os.environ.pop(kk, None)
        os.environ.update(env)
        eng = MPlda(0)
        eng.fit(x, y, it)
        g = eng.get_model(); fi = eng.fit_internals()
        # GetOutput redone on the host from the device's W and B: separates the EM error from GetOutput's
        st = {"sum": ref["mean"], "class_weight": 1.0}
        host = onp.get_output(st, fi["W"], fi["B"])
        e_host = np.abs(host["psi"] - ref["psi"]).max() / max(ref["psi"].max(), 1e-12)
        line += " | %s W %.1e B %.1e psi %.1e (host GetOutput of device W,B: %.1e) TtT %.1e" % (
            ",".join("%s=%s" % kv for kv in env.items()) or "default", rel(fi["W"], ref["W"]), rel(fi["B"], ref["B"]),
            np.abs(g["psi"] - ref["psi"]).max() / max(ref["psi"].max(), 1e-12), e_host,
            rel(g["transform"].T @ g["transform"], ref["transform"].T @ ref["transform"]))
        if truth:
            line += " vs extended precision: W %.1e B %.1e" % (dist(fi["W"], Wt), dist(fi["B"], Bt))
    print(line)
