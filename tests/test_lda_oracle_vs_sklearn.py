"""CPU: a second, independent pin of the LDA oracle.  oracle/lda_oracle_np.py is already checked against outputs of the
reference's own python/liblda/lda.py (tests/golden/lda_*.npz); that file is a copy of scikit-learn's 2014
LinearDiscriminantAnalysis, and the scikit-learn of this image (1.7) still computes the same model for the 'svd' and
'lsqr' solvers -- so the oracle must agree with it as well.  ('eigen' is left out: later scikit-learn versions
changed the normalisation of its scalings.)"""
import numpy as np
import pytest

sklearn = pytest.importorskip("sklearn")
from sklearn.discriminant_analysis import LinearDiscriminantAnalysis  # noqa: E402

from oracle import lda_oracle_np as lo  # noqa: E402


@pytest.mark.parametrize("solver", ["svd", "lsqr"])
@pytest.mark.parametrize("n,d,k,seed", [(600, 12, 5, 0), (2000, 40, 17, 1), (300, 8, 3, 2)])
def test_oracle_matches_scikit_learn(solver, n, d, k, seed):
    rng = np.random.default_rng(seed)
    y = rng.integers(0, k, n)
    y[:k] = np.arange(k)                                   # every class present
    X = rng.standard_normal((n, d)) + rng.standard_normal((k, d))[y] * 1.5
    m = lo.fit(X, y, solver=solver)
    ref = LinearDiscriminantAnalysis(solver=solver).fit(X, y)
    np.testing.assert_allclose(m["priors"], ref.priors_, rtol=1e-13)
    np.testing.assert_allclose(m["means"], ref.means_, rtol=1e-12, atol=1e-13)
    Xt = rng.standard_normal((50, d))
    np.testing.assert_allclose(lo.decision_function(m, Xt), ref.decision_function(Xt), rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(lo.predict_log_proba(m, Xt), ref.predict_log_proba(Xt), rtol=1e-7, atol=1e-7)
    if solver == "svd":
        a, b = lo.transform(m, Xt), ref.transform(Xt)
        assert a.shape == b.shape
        np.testing.assert_allclose(np.abs(a), np.abs(b), rtol=1e-7, atol=1e-8)   # column signs are the SVD's choice
