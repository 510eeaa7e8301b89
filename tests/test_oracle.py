"""CPU: pins the oracle (oracle/plda_oracle.c) -- PARITY UNPINNED by the reference
(Kaldi absent, tests/pldatest.py pins no values), so the pins are: golden fixtures from
the independent NumPy restatement, closed-form known answers, and algebraic invariants
(SURVEY.md section 8c)."""
import numpy as np
import pytest

from conftest import GOLDEN_CASES, load_golden, make_data
from oracle import plda_oracle_np as onp


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_c_oracle_matches_golden(oracle, name):
    g = load_golden(name)
    x, y, d = g["X"], g["y"], int(g["D"])
    m = oracle.fit(x, y, int(g["iters"]))
    T, psi = m["transform"], m["psi"]
    b = g["TtT"].shape[0]
    assert np.abs(psi - g["psi"]).max() <= 1e-10 * g["psi"].max()
    assert _rel(m["mean"], g["mean"]) < 1e-12
    assert _rel((T.T @ T)[:b, :b], g["TtT"]) < 1e-9
    assert _rel((T.T @ np.diag(psi) @ T)[:b, :b], g["TtPsiT"]) < 1e-9
    assert _rel(m["W"][:b, :b], g["W"]) < 1e-10 and _rel(m["B"][:b, :b], g["B"]) < 1e-10
    tr = np.array([np.trace(T.T @ T), np.trace(T.T @ np.diag(psi) @ T), np.trace(m["W"]), np.trace(m["B"])])
    np.testing.assert_allclose(tr, g["traces"], rtol=1e-10)
    ne, nt = int(g["enrol_n"]), int(g["test_n"])
    el, ec, ev = oracle.transform_groups(m, x[:ne], y[:ne])
    np.testing.assert_array_equal(el, g["enrol_labels"])
    np.testing.assert_array_equal(ec, g["enrol_counts"])
    tv = np.stack([oracle.transform_ivector(m, r, 1) for r in x[ne:ne + nt]])
    S = oracle.score_block(psi, ev, ec, tv)
    np.testing.assert_allclose(S, g["scores"], rtol=1e-8, atol=1e-10)
    zm, zs = oracle.norm(m, g["bkg"], ev)
    np.testing.assert_allclose(zm, g["znorm_mean"], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(zs, g["znorm_std"], rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize("n,d,k,skew", [(60, 6, 5, True), (900, 17, 30, True), (400, 40, 2, False)])
def test_c_oracle_vs_numpy_restatement_stagewise(oracle, n, d, k, skew):
    x, y = make_data(n + d, n, d, k, skew=skew, scale_between=0.3)
    sc, sn = oracle.stats(x, y), onp.stats(x, y)
    for key in ("means", "scatter", "sum"):
        assert _rel(sc[key], sn[key]) < 1e-12, key
    assert abs(sc["class_weight"] - sn["class_weight"]) < 1e-12
    assert sc["example_weight"] == pytest.approx(k)            # w_k = 1/n_k  =>  sum w_k n_k = K
    W = B = np.eye(d)
    for _ in range(3):
        Wc, Bc = oracle.em_iter(sc, W, B)
        Wn, Bn = onp.em_iter(sn, W, B)
        assert _rel(Wc, Wn) < 1e-11 and _rel(Bc, Bn) < 1e-11
        W, B = Wn, Bn
    oc, on = oracle.get_output(sc, W, B), onp.get_output(sn, W, B)
    assert np.abs(oc["psi"] - on["psi"]).max() < 1e-11 * on["psi"].max()
    assert _rel(oc["transform"].T @ oc["transform"], on["transform"].T @ on["transform"]) < 1e-9


def test_getoutput_invariants(oracle):
    x, y = make_data(5, 800, 24, 40, skew=True, scale_between=0.5)
    m = oracle.fit(x, y, 7)
    T, psi, d = m["transform"], m["psi"], 24
    assert np.abs(T @ m["W"] @ T.T - np.eye(d)).max() < 1e-10
    assert np.abs(T @ m["B"] @ T.T - np.diag(psi)).max() < 1e-10
    assert (np.diff(psi) <= 0).all() and (psi >= 0).all()
    assert np.abs(m["offset"] + T @ m["mean"]).max() < 1e-12
    t = oracle.transform_ivector(m, x[3], 4)
    assert abs((t * t / (psi + 0.25)).sum() - d) < 1e-10        # length-norm target
    t_simple = oracle.transform_ivector(m, x[3], 4, simple_length_norm=True)
    assert abs(np.linalg.norm(t_simple) - np.sqrt(d)) < 1e-10
    t_raw = oracle.transform_ivector(m, x[3], 4, normalize_length=False)
    np.testing.assert_allclose(t_raw, T @ x[3] + m["offset"], rtol=1e-12)


def test_em_objective_non_decreasing(oracle):
    x, y = make_data(6, 300, 8, 12, skew=True, scale_between=0.7)
    st = oracle.stats(x, y)
    W = B = np.eye(8)
    prev = -np.inf
    for _ in range(8):
        W, B = oracle.em_iter(st, W, B)
        obj = oracle.objective(st, W, B)
        assert obj >= prev - 1e-12
        prev = obj


def test_llr_closed_form_d1(oracle):
    """D = 1 known answer: joint Gaussian of (enrol mean of n, test) in the PLDA space has
    covariance [[psi + 1/n, psi], [psi, psi + 1]]; the LLR is log N(v | u) - log N(v)."""
    from scipy.stats import norm
    for psi, n, u, v in [(2.0, 1, 0.7, -0.3), (0.5, 4, -1.2, 0.9), (10.0, 3, 2.0, 2.1), (1e-3, 7, 0.1, 0.2)]:
        cmean = psi / (psi + 1.0 / n) * u
        cvar = psi + 1.0 - psi * psi / (psi + 1.0 / n)
        expect = norm.logpdf(v, cmean, np.sqrt(cvar)) - norm.logpdf(v, 0.0, np.sqrt(psi + 1.0))
        assert abs(oracle.llr(np.array([psi]), np.array([u]), n, np.array([v])) - expect) < 1e-12


def test_llr_pair_form_equals_gemm_form(oracle):
    rng = np.random.default_rng(8)
    d = 31
    psi = np.sort(rng.random(d) * 5)[::-1].copy()
    U, V = rng.standard_normal((9, d)), rng.standard_normal((13, d))
    counts = rng.integers(1, 6, 9)
    pair = oracle.score_block(psi, U, counts, V)
    np.testing.assert_allclose(pair, onp.llr_matrix(psi, U, counts, V), rtol=1e-11, atol=1e-12)
    flip = np.where(rng.random(d) < 0.5, -1.0, 1.0)            # eigenvector sign ambiguity
    np.testing.assert_allclose(oracle.score_block(psi, U * flip, counts, V * flip), pair, rtol=0, atol=0)


def test_smooth(oracle):
    x, y = make_data(9, 400, 12, 20, scale_between=0.5)
    m = oracle.fit(x, y, 4)
    s0 = oracle.smooth(m, 0.0)
    np.testing.assert_array_equal(s0["psi"], m["psi"])
    np.testing.assert_array_equal(s0["transform"], m["transform"])
    s = oracle.smooth(m, 0.5)
    n = onp.smooth(m, 0.5)
    np.testing.assert_allclose(s["psi"], n["psi"], rtol=1e-14)
    np.testing.assert_allclose(s["transform"], n["transform"], rtol=1e-13)
    np.testing.assert_allclose(s["offset"], n["offset"], rtol=1e-11, atol=1e-13)


def test_dense_helpers(oracle):
    rng = np.random.default_rng(10)
    for d in (1, 2, 7, 40):
        a = rng.standard_normal((d, d + 3))
        spd = a @ a.T + 0.1 * np.eye(d)
        np.testing.assert_allclose(oracle.cholesky(spd), np.linalg.cholesky(spd), rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(oracle.invert(spd), np.linalg.inv(spd), rtol=1e-9, atol=1e-11)
        s, u = oracle.sym_eig(spd)
        np.testing.assert_allclose(np.sort(s), np.linalg.eigvalsh(spd), rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(u @ np.diag(s) @ u.T, spd, rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(u.T @ u, np.eye(d), atol=1e-12)


def test_fit_errors_and_edges(oracle):
    x = np.random.default_rng(0).random((10, 3))
    with pytest.raises(ValueError, match="Number of speakers is 1"):
        oracle.fit(x, np.zeros(10, np.uint64), 2)                # pldamodule.cpp:83-86
    with pytest.raises(RuntimeError):
        oracle.fit(x, np.array([0, 2] * 5, np.uint64), 2)        # not dense (label 1 unused)
    # iters = 0: W = B = I  =>  transform is orthogonal-free identity-like, psi = 1
    m = oracle.fit(x, (np.arange(10) % 2).astype(np.uint64), 0)
    np.testing.assert_allclose(m["psi"], 1.0)
    np.testing.assert_allclose(m["transform"].T @ m["transform"], np.eye(3), atol=1e-12)
    # ragged: transform of a single row / single group
    l, c, v = oracle.transform_groups(m, x[:1], np.array([7], np.uint64))
    assert list(l) == [7] and list(c) == [1] and v.shape == (1, 3)


def test_eer_restatement_properties():
    """oracle eer(): farfrr identities and the tie rule on a hand-checkable case."""
    thr, far, frr, e = onp.eer([0.1, 0.4, 0.35, 0.8], [0.3, 0.6, 0.9])
    # candidates: .1 | .2 .325 .375 .5 .7 .85 .9+1e-8 -> |FAR-FRR| = 1, .75, .417, .167, .083, .417, .75(?), 1
    assert (far, frr) == (0.25, 1.0 / 3.0) and thr == pytest.approx(0.5) and e == pytest.approx((0.25 + 1 / 3) / 2)
    rng = np.random.default_rng(0)
    neg, pos = rng.normal(-1, 1, 2000), rng.normal(1, 1, 300)
    thr, far, frr, e = onp.eer(neg, pos)
    n32, p32 = neg.astype(np.float32).astype(np.float64), pos.astype(np.float32).astype(np.float64)
    assert far == (n32 >= thr).mean() and frr == (p32 < thr).mean() and abs(far - frr) < 0.01


def test_em_recovers_a_generating_two_covariance_model(oracle):
    """A known-answer test that does not come from this repository's own restatement: data DRAWN from the
    two-covariance model  x_ki = mu + y_k + e_ki,  y_k ~ N(0, B*),  e_ki ~ N(0, W*)  with many balanced speakers.
    The estimator the reference drives (Kaldi PldaEstimator through pldamodule.cpp:94-106; with equal n_k the wrapper's
    1/n_k class weight is a constant) is maximum likelihood for exactly this model, so its fixed point must approach
    (mu, W*, B*) at the sampling rate 1/sqrt(K); psi must approach the generalised eigenvalues of (B*, W*)."""
    from scipy.linalg import eigh
    rng = np.random.default_rng(11)
    d, K, n = 5, 3000, 8
    a = rng.standard_normal((d, d)); Wt = a @ a.T / d + 0.5 * np.eye(d)
    b = rng.standard_normal((d, d)); Bt = 2.0 * (b @ b.T / d) + 0.2 * np.eye(d)
    mu = rng.standard_normal(d)
    yk = rng.multivariate_normal(np.zeros(d), Bt, K)
    x = mu + np.repeat(yk, n, axis=0) + rng.multivariate_normal(np.zeros(d), Wt, K * n)
    labels = np.repeat(np.arange(K, dtype=np.uint64), n)
    m = oracle.fit(x, labels, 40)
    assert np.abs(m["mean"] - mu).max() < 5.0 / np.sqrt(K)
    assert np.abs(m["W"] - Wt).max() < 0.06 * np.abs(Wt).max()           # K n = 24 000 samples of the within part
    assert np.abs(m["B"] - Bt).max() < 0.12 * np.abs(Bt).max()           # K = 3 000 samples of the between part
    ref_psi = np.sort(eigh(Bt, Wt, eigvals_only=True))[::-1]
    assert np.abs(m["psi"] - ref_psi).max() < 0.12 * ref_psi.max()
    # and the transform whitens the TRUE within-class covariance up to the same sampling error
    T = m["transform"]
    assert np.abs(T @ Wt @ T.T - np.eye(d)).max() < 0.1


def test_extended_precision_em_agrees_with_the_fp64_oracles():
    """oracle/plda_oracle_np.py:fit_wb_longdouble is the yardstick of the ill-conditioned GPU tests: on a
    well-conditioned problem it has to reproduce the fp64 restatements to rounding."""
    x, y = make_data(3, 300, 12, 20, skew=True, scale_between=0.5)
    _, dense = np.unique(y, return_inverse=True)
    W, B = onp.fit_wb_longdouble(x, dense, 4)
    m = onp.fit(x, dense, 4, return_wb=True)
    assert float(np.abs(W - m["W"]).max() / np.abs(m["W"]).max()) < 1e-13
    assert float(np.abs(B - m["B"]).max() / np.abs(m["B"]).max()) < 1e-13
