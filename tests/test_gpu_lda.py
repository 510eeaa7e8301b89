"""GPU parity of the LDA row (plda_amd.lda.LDA, csrc/lda.hip) through the C ABI:
  * against the REFERENCE's own outputs (tests/golden/lda_*.npz, from python/liblda/lda.py),
  * against the NumPy oracle on larger seeded problems,
  * size-independent properties at a C2-sized problem.
Tolerances: fp64 everywhere; 1e-8 relative on coef / intercept / decision values (both SVDs of the
svd solver are taken through D x D Gram matrices on the GPU, which squares the condition number of
the whitening step), 1e-9 on log-probabilities.
"""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "lda_*.npz")))


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _lsqr_ok(g):
    # lstsq of a rank-deficient Sw is a minimum-norm solution whose cut-off is LAPACK's; compare only full rank
    n, d = g["X"].shape
    return n - len(np.unique(g["y"])) >= d


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[4:-4] for p in GOLD])
@pytest.mark.parametrize("solver", ["svd", "eigen", "lsqr"])
def test_lda_matches_reference_outputs(path, solver):
    from liblda import LDA
    g = np.load(path)
    pri = g["priors_in"].copy() if "priors_in" in g else None
    lda = LDA(solver, pri)
    if solver + "_error" in g:
        with pytest.raises(np.linalg.LinAlgError):
            lda.fit(g["X"], g["y"])
        return
    if solver == "lsqr" and not _lsqr_ok(g):
        pytest.skip("rank-deficient lstsq: minimum-norm cut-off is LAPACK's own")
    assert lda.fit(g["X"], g["y"]) is None
    k, d = g[solver + "_coef"].shape
    well_defined = solver != "eigen" or k - 1 >= d
    assert _rel(lda.priors, g[solver + "_priors"]) < 1e-14
    lp = lda.predict_log_proba(g["Xt"])
    assert lp.shape == g[solver + "_log_proba"].shape
    assert _rel(lp, g[solver + "_log_proba"]) < 1e-9
    assert lda.predict_proba(g["Xt"]).shape == g[solver + "_proba"].shape
    if well_defined:
        assert _rel(lda._coef, g[solver + "_coef"]) < 1e-8
        assert _rel(lda._intercept, g[solver + "_intercept"]) < 1e-8
        assert _rel(lda.decision_function(g["Xt"]), g[solver + "_decision"]) < 1e-8
        assert _rel(lda.predict_proba(g["Xt"]), g[solver + "_proba"]) < 1e-8
    if solver == "svd":
        assert _rel(lda._xbar, g["svd_xbar"]) < 1e-13
        assert lda._scalings.shape == g["svd_scalings"].shape
        assert _rel(lda._scalings @ lda._scalings.T, g["svd_scalings"] @ g["svd_scalings"].T) < 1e-8
    if solver == "eigen":
        assert _rel(lda.explained_variance_ratio_, g["eigen_evr"]) < 1e-9
        lead = min(k - 1, d)
        s, r = lda._scalings[:, :lead], g["eigen_scalings"][:, :lead]
        assert np.abs(np.abs((s * r).sum(0)) - 1.0).max() < 1e-8
        assert lda.transform(g["Xt"], 2).shape == g["eigen_transform2"].shape
        if well_defined:
            assert _rel(np.abs(lda.transform(g["Xt"])), np.abs(g["eigen_transform"])) < 1e-7


@pytest.mark.parametrize("solver", ["svd", "eigen", "lsqr"])
@pytest.mark.parametrize("n,d,k,given_priors", [(5000, 64, 80, False), (3000, 33, 7, True), (1500, 200, 30, False)])
def test_lda_matches_oracle(solver, n, d, k, given_priors):
    from oracle import lda_oracle_np as lo
    from liblda import LDA
    rng = np.random.default_rng(n + d + k)
    y = rng.integers(0, k, n) * 3 + 11                      # labels need not be dense or zero-based
    X = rng.random((n, d)) + 0.5 * rng.standard_normal((3 * k + 11, d))[y]
    Xt = rng.random((257, d))
    pri = rng.random(len(np.unique(y))) + 0.2 if given_priors else None
    ref = lo.fit(X, y, solver, pri)
    lda = LDA(solver, None if pri is None else pri.copy())
    lda.fit(X, y)
    well_defined = solver != "eigen" or len(ref["classes"]) - 1 >= d
    assert np.array_equal(lda._classes, ref["classes"])
    assert _rel(lda._means, ref["means"]) < 1e-13
    assert _rel(lda.predict_log_proba(Xt), lo.predict_log_proba(ref, Xt)) < 1e-9
    if well_defined:
        assert _rel(lda._coef, ref["coef"]) < 1e-8
        assert _rel(lda._intercept, ref["intercept"]) < 1e-8
        assert _rel(lda.predict_proba(Xt), lo.predict_proba(ref, Xt)) < 1e-8
    if solver == "svd":
        assert lda._scalings.shape == ref["scalings"].shape
        assert _rel(np.abs(lda.transform(Xt)), np.abs(lo.transform(ref, Xt))) < 1e-7
        assert lda.transform(Xt, 3).shape == (257, 3)


def test_lda_error_behaviour_and_persistence(tmp_path):
    from liblda import LDA
    rng = np.random.default_rng(5)
    X, y = rng.random((200, 9)), rng.integers(0, 4, 200)
    lda = LDA("svd")
    with pytest.raises(ValueError, match="not fitted yet"):        # lda.py:258-259
        lda.decision_function(X)
    lda.fit(X, y)
    with pytest.raises(ValueError, match="X has 5 features per sample; expecting 9"):   # lda.py:264-266
        lda.predict_log_proba(rng.random((3, 5)))
    lsqr = LDA("lsqr", engine=lda._eng)
    lsqr.fit(X, y)
    with pytest.raises(NotImplementedError, match="transform not implemented for 'lsqr'"):
        lsqr.transform(X)
    lda.fit(X, y)
    want = lda.predict_log_proba(X[:17])
    lda.save(str(tmp_path / "lda.npz"))
    other = LDA().load(str(tmp_path / "lda.npz"))
    np.testing.assert_array_equal(other.predict_log_proba(X[:17]), want)
    np.testing.assert_array_equal(other.transform(X[:5]), lda.transform(X[:5]))
    # one-sample predict, the shape scoring/scoreLDA.py:240-241 feeds
    one = lda.predict_log_proba(X[3][np.newaxis, :])[0]
    np.testing.assert_allclose(one, want[3], rtol=0, atol=1e-12)


def test_lda_c2_sized_properties():
    """100k x 200 background, 5k speakers (the C2 shape, scoring/scoreLDA.py's workload): rows of
    predict_log_proba are normalised log-probabilities, decision values are affine in X, and the
    prediction of a class centroid is that class."""
    import torch
    from liblda import LDA
    rng = np.random.default_rng(2)
    n, d, k = 100_000, 200, 5000
    y = np.arange(n) % k
    X = rng.random((n, d)) + 0.8 * rng.standard_normal((k, d))[y]
    lda = LDA("svd")
    lda.fit(X, y)
    assert lda._scalings.shape[1] == d                       # K - 1 > D: full rank
    Xt = X[:4096]
    lp = lda.predict_log_proba(Xt)
    assert lp.shape == (4096, k) and np.isfinite(lp).all() and (lp <= 0).all()
    assert np.abs(np.log(np.exp(lp).sum(1))).max() < 1e-10
    dec = lda.decision_function(Xt)
    assert _rel(dec - dec.max(1, keepdims=True) - np.log(np.exp(dec - dec.max(1, keepdims=True)).sum(1, keepdims=True)), lp) < 1e-10
    a, b = Xt[:512], Xt[512:1024]
    mix = lda.decision_function(0.25 * a + 0.75 * b)
    assert _rel(mix, 0.25 * dec[:512] + 0.75 * dec[512:1024]) < 1e-10
    cen = lda.predict_log_proba(lda._means[:1000])
    assert (cen.argmax(1) == np.arange(1000)).all()
    assert (lp.argmax(1) == y[:4096]).mean() > 0.95
    del torch


def test_lda_trial_list_scores_match_per_trial_loop(tmp_path):
    """scoring/scoreLDA.py:228-248: the batched trial-list scorer writes the same file as the
    reference's one-predict-per-trial loop run on the oracle."""
    import io
    from oracle import lda_oracle_np as lo
    from liblda import LDA
    from plda_amd.trials import parse_trial_ref, score_trial_list_lda
    rng = np.random.default_rng(77)
    spk = ["spk%02d" % i for i in range(9)]
    spktonum = {s: i for i, s in enumerate(sorted(spk))}
    labels = rng.integers(0, 9, 600)
    X = rng.random((600, 12)) + 0.7 * rng.standard_normal((9, 12))[labels]
    testtofeature = {"utt-%03d" % i: rng.random(12) + 0.7 * rng.standard_normal(12) for i in range(40)}
    lines = []
    for t in range(150):
        u = "utt-%03d" % rng.integers(0, 42)                     # two utterances are missing on purpose
        m = spk[rng.integers(0, 9)] if t % 37 else "ghost"
        lines.append("%s %s-%s %d\n" % (m, spk[rng.integers(0, 9)], u, rng.integers(0, 2)))
    ref_path = tmp_path / "test_ref"
    ref_path.write_text("".join(lines))
    refs = parse_trial_ref(str(ref_path))
    lda = LDA("svd")
    lda.fit(X, labels)
    got = io.StringIO()
    n, err = score_trial_list_lda(lda, refs, spktonum, testtofeature, got, chunk=16)
    model = lo.fit(X, labels, "svd")
    want, werr = io.StringIO(), 0
    for enrolemodel, vals in refs.items():
        if enrolemodel not in spktonum:
            werr += 1
            continue
        for testutt, targetmdl in vals:
            if testutt not in testtofeature:
                werr += 1
                continue
            score = lo.predict_log_proba(model, testtofeature[testutt][np.newaxis, :])[0]
            want.write("{} {}-{} {:.3f}\n".format(enrolemodel, targetmdl, testutt, score[spktonum[enrolemodel]]))
    assert err == werr and n == len(want.getvalue().splitlines())
    assert got.getvalue() == want.getvalue()
