"""CPU: the LDA oracle (oracle/lda_oracle_np.py) against outputs of the REFERENCE's own lda.py
(tests/golden/lda_*.npz, produced by tests/golden/make_lda_golden.py) -- the one row of this
repository whose parity is pinned by the reference itself."""
import glob
import os

import numpy as np
import pytest

from oracle import lda_oracle_np as lo

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "lda_*.npz")))


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def test_golden_files_present():
    assert len(GOLD) == 4


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[4:-4] for p in GOLD])
@pytest.mark.parametrize("solver", ["svd", "eigen", "lsqr"])
def test_oracle_matches_reference_outputs(path, solver):
    g = np.load(path)
    pri = g["priors_in"] if "priors_in" in g else None
    if solver + "_error" in g:                       # the reference raised (singular Sw, eigen solver)
        with pytest.raises(np.linalg.LinAlgError):
            lo.fit(g["X"], g["y"], solver, pri)
        return
    m = lo.fit(g["X"], g["y"], solver, pri)
    assert _rel(m["priors"], g[solver + "_priors"]) < 1e-14
    k, d = g[solver + "_coef"].shape
    well_defined = solver != "eigen" or k - 1 >= d or "binary" in path
    # with fewer than D+1 classes the eigen solver's coef depends on LAPACK's arbitrary basis of the
    # null space of Sb (every class-independent direction); log-probabilities do not
    assert _rel(lo.predict_log_proba(m, g["Xt"]), g[solver + "_log_proba"]) < 1e-10
    if well_defined:
        assert _rel(m["coef"], g[solver + "_coef"]) < 1e-10
        assert _rel(m["intercept"], g[solver + "_intercept"]) < 1e-10
        assert _rel(lo.decision_function(m, g["Xt"]), g[solver + "_decision"]) < 1e-10
        assert _rel(lo.predict_proba(m, g["Xt"]), g[solver + "_proba"]) < 1e-10
    assert lo.predict_proba(m, g["Xt"]).shape == g[solver + "_proba"].shape
    if solver == "svd":
        assert _rel(m["xbar"], g["svd_xbar"]) < 1e-13
        s, r = m["scalings"], g["svd_scalings"]
        assert s.shape == r.shape
        assert _rel(s @ s.T, r @ r.T) < 1e-9                       # sign-free comparison
    if solver == "eigen":
        assert _rel(m["explained_variance_ratio"], g["eigen_evr"]) < 1e-9
        lead = min(k - 1, d)
        s, r = m["scalings"][:, :lead], g["eigen_scalings"][:, :lead]
        assert np.abs(np.abs((s * r).sum(0)) - 1.0).max() < 1e-8   # unit columns, equal up to sign
        if well_defined:
            t = lo.transform(m, g["Xt"])
            assert _rel(np.abs(t), np.abs(g["eigen_transform"])) < 1e-8
            assert lo.transform(m, g["Xt"], 2).shape == g["eigen_transform2"].shape


def test_lsqr_has_no_transform():
    g = np.load(GOLD[0])
    m = lo.fit(g["X"], g["y"], "lsqr")
    with pytest.raises(NotImplementedError):
        lo.transform(m, g["Xt"])
