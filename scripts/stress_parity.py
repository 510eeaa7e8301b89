"""Randomised parity sweep (GPU vs the fp64 oracles) over many seeded shapes -- a hunting tool run by hand
on the GPU box (`python scripts/stress_parity.py [n_cases]`); the fixed cases live in tests/."""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_data, score_tol          # noqa: E402
from oracle import binding as ob, lda_oracle_np as lo, plda_oracle_np as onp   # noqa: E402
from plda_amd import MPlda                          # noqa: E402
from plda_amd.lda import LDA                        # noqa: E402

ob.build()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # second argument: offset of every seed (fresh cases)
fails = []


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


for case in range(ncases):
    rng = np.random.default_rng(1000 + case + seed0)
    d = int(rng.choice([1, 2, 3, 5, 8, 17, 31, 32, 33, 64, 100, 129, 200, 257, 300]))
    k = int(rng.integers(2, 40))
    n = int(max(k * 2, rng.integers(k + 1, 40 * k)))
    skew = bool(rng.integers(0, 2))
    iters = int(rng.integers(0, 7))
    between = float(rng.choice([0.0, 0.2, 1.0]))
    tag = "case %d: N=%d D=%d K=%d skew=%s iters=%d between=%.1f" % (case, n, d, k, skew, iters, between)
    try:
        x, y = make_data(5000 + case + seed0, n, d, k, skew=skew, scale_between=between)
        eng = MPlda(0)
        eng.fit(x, y, iters)
        ref = ob.fit(x, y, iters)
        g = eng.get_model()
        e_psi = np.abs(g["psi"] - ref["psi"]).max() / max(ref["psi"].max(), 1e-12)
        e_tt = rel(g["transform"].T @ g["transform"], ref["transform"].T @ ref["transform"])
        # scoring with mixed / uniform counts, each side with its own transform (eigenvector signs)
        m = min(n, 150)
        labs = np.arange(m, dtype=np.uint64) if case % 2 else y[:m]
        got = eng.transform(x[:m], labs)
        _, rc, rv = ob.transform_groups(ref, x[:m], labs)
        ids = sorted(got)
        U = np.stack([got[i][1] for i in ids])
        cnt = np.array([got[i][0] for i in ids], np.int32)
        assert np.array_equal(cnt, rc)
        tt = eng.transform_array(x[-60:], 1)
        rt = onp.transform_ivector(ref, x[-60:], 1)
        S = eng.score_matrix((cnt, U), (np.ones(len(tt), np.int32), tt), znorm=False)
        Sr = ob.score_block(ref["psi"], rv, rc, rt)
        assert np.isfinite(Sr).all() and np.isfinite(S).all()
        tol = score_tol(Sr)
        bad = np.abs(S - Sr) > tol
        msg = []
        # two fp64 implementations of GetOutput differ by about cond(W) * eps (rank-deficient scatter: cond grows
        # ~40 x per EM iteration): the C and NumPy oracles themselves are 5e-7 apart in psi at cond = 1e9
        cw = float(np.linalg.cond(ref["W"])) if "W" in ref else 1.0
        if not (e_psi < max(1e-8, 20 * cw * 2.2e-16) and e_tt < max(1e-7, 200 * cw * 2.2e-16)):
            msg.append("fit psi %.2e TtT %.2e" % (e_psi, e_tt))
        if bad.any():
            msg.append("scores max err %.3e (tol %.3e)" % (np.abs(S - Sr).max(), float(np.min(tol))))
        # LDA on the same data
        if n - k > 2:
            counts = np.bincount(np.unique(y, return_inverse=True)[1])
            for solver in ("svd", "lsqr", "eigen"):
                if solver in ("lsqr", "eigen") and n - k < d + 2:
                    continue
                if solver == "eigen" and (k - 1 < d or counts.min() < 2):
                    continue          # below D + 1 classes the eigen solver's coef is basis-dependent
                lda = LDA(solver, engine=eng)
                lda.fit(x, y)
                lr = lo.fit(x, y, solver)
                e_lp = rel(lda.predict_log_proba(x[:50]), lo.predict_log_proba(lr, x[:50]))
                if not e_lp < 1e-7:
                    msg.append("lda %s log_proba %.2e" % (solver, e_lp))
        if msg:
            fails.append(tag + " -> " + "; ".join(msg))
    except Exception as ex:
        fails.append(tag + " -> EXC " + repr(ex)[:200] + " | " + traceback.format_exc().splitlines()[-3][:150])
print("%d cases, %d failures" % (ncases, len(fails)))
for f in fails:
    print(" ", f)
