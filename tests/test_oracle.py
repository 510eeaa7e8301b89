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


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: checks of the oracle that do NOT go through SURVEY.md Appendix A's formulas (the C oracle, the NumPy oracle and
# the golden fixtures share one reading of Kaldi; these three share none of it).  They start from the two-covariance model
# itself -- x_ki = mu + y_k + e_ki, y_k ~ N(0, B), e_ki ~ N(0, W) -- and generic Gaussian algebra on the FULL n D-dimensional
# joint distribution of a class (big dense covariance matrices, NumPy / SciPy only).
# ---------------------------------------------------------------------------------------------------------------------
def _small_classes(seed, D, sizes):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((D, D)); Bt = A @ A.T / D + 0.3 * np.eye(D)
    C = rng.standard_normal((D, D)); Wt = C @ C.T / D + 0.5 * np.eye(D)
    mu = rng.standard_normal(D)
    y = np.repeat(np.arange(len(sizes)), sizes)
    cls = rng.multivariate_normal(np.zeros(D), Bt, len(sizes))
    x = mu + cls[y] + rng.multivariate_normal(np.zeros(D), Wt, len(y))
    return x, y.astype(np.uint64)


def _em_step_from_the_joint_gaussian(x, y, mu, W, B):
    """One EM iteration derived from the model alone.  E-step: for class k, (y_k, x_k1 .. x_kn) is jointly Gaussian;
    condition y_k on the observations with the generic formula E[y|x] = S_yx S_xx^-1 (x - mu), Cov = B - S_yx S_xx^-1 S_xy
    on the n D x n D covariance S_xx = I_n (x) W + 1 1^T (x) B.  M-step: B = sum_k w_k E[y y^T] / sum_k w_k,
    W = sum_k w_k sum_i E[(x_ki - mu - y)(x_ki - mu - y)^T] / sum_k w_k n_k, with the wrapper's class weight
    w_k = 1 / n_k (pldamodule.cpp:97)."""
    D = x.shape[1]
    Bs, Ws, bw, ww = np.zeros((D, D)), np.zeros((D, D)), 0.0, 0.0
    for k in np.unique(y):
        xk = x[y == k] - mu
        n = xk.shape[0]
        w = 1.0 / n
        Sxx = np.kron(np.eye(n), W) + np.kron(np.ones((n, n)), B)
        Syx = np.kron(np.ones((1, n)), B)                        # Cov(y, x_i) = B for every i
        G = Syx @ np.linalg.inv(Sxx)
        ey = G @ xk.reshape(-1)
        cy = B - G @ Syx.T
        Bs += w * (cy + np.outer(ey, ey)); bw += w
        for i in range(n):
            r = xk[i] - ey
            Ws += w * (cy + np.outer(r, r))
        ww += w * n
    return Ws / ww, Bs / bw


@pytest.mark.parametrize("sizes", [[4] * 12, [1, 2, 3, 5, 2, 7, 3, 3, 4, 6, 2, 5]])
def test_em_iteration_equals_the_exact_posterior_update(oracle, sizes):
    """oracle.em_iter (Kaldi's GetStatsFromIntraClass / GetStatsFromClassMeans / EstimateFromStats as restated) against
    the EM update computed from the joint Gaussian of each class by brute force: same W, B to 1e-10, from several
    starting points, balanced and unbalanced classes (the unbalanced ones exercise the 1 / n_k class weight and the
    count conventions of the M-step)."""
    D = 3
    x, y = _small_classes(5, D, sizes)
    st = oracle.stats(x, y)
    mu = st["sum"] / st["class_weight"]
    rng = np.random.default_rng(6)
    W, B = np.eye(D), np.eye(D)
    for it in range(4):
        Wo, Bo = oracle.em_iter(st, W, B)
        Wr, Br = _em_step_from_the_joint_gaussian(x, y, mu, W, B)
        assert _rel(Wo, Wr) < 1e-10 and _rel(Bo, Br) < 1e-10, (it, _rel(Wo, Wr), _rel(Bo, Br))
        # next starting point: the update, pushed off the EM path by a random SPD perturbation
        P = rng.standard_normal((D, D)) * 0.2
        W, B = Wo + P @ P.T, Bo + P.T @ P


def test_em_fixed_point_is_a_stationary_point_of_the_likelihood(oracle):
    """At the EM's fixed point the gradient of the (class-weighted) marginal log-likelihood sum_k w_k log p(x_k | mu, W, B)
    vanishes.  The likelihood is evaluated from the model's definition only: scipy's multivariate normal on the
    n D-dimensional covariance I (x) W + 1 1^T (x) B of every class; the gradient by central differences along symmetric
    directions.  After ONE iteration the same gradient is orders of magnitude larger (the test has teeth)."""
    from scipy.stats import multivariate_normal
    D, sizes = 2, [3, 5, 2, 4, 6, 3, 2, 5, 4, 3, 6, 2, 4, 5, 3, 4]
    x, y = _small_classes(9, D, sizes)
    st = oracle.stats(x, y)
    mu = st["sum"] / st["class_weight"]

    def loglik(W, B):
        tot = 0.0
        for k in np.unique(y):
            xk = x[y == k]
            n = xk.shape[0]
            S = np.kron(np.eye(n), W) + np.kron(np.ones((n, n)), B)
            tot += (1.0 / n) * multivariate_normal(np.tile(mu, n), S).logpdf(xk.reshape(-1))
        return tot

    def grad_norm(W, B, h=1e-5):
        g = []
        for M, which in ((W, 0), (B, 1)):
            for a in range(D):
                for b in range(a, D):
                    E = np.zeros((D, D)); E[a, b] = E[b, a] = 1.0
                    if which == 0: g.append((loglik(W + h * E, B) - loglik(W - h * E, B)) / (2 * h))
                    else: g.append((loglik(W, B + h * E) - loglik(W, B - h * E)) / (2 * h))
        return np.abs(np.array(g)).max()

    W, B = np.eye(D), np.eye(D)
    W1, B1 = oracle.em_iter(st, W, B)
    g_early = grad_norm(W1, B1)
    W, B = W1, B1
    for _ in range(30000):                    # (EM is slow on 16 small classes: ~10 000 iterations to 1e-14)
        Wn, Bn = oracle.em_iter(st, W, B)
        done = max(_rel(Wn, W), _rel(Bn, B)) < 1e-14
        W, B = Wn, Bn
        if done:
            break
    g_fixed = grad_norm(W, B)
    assert g_fixed < 1e-6, g_fixed
    assert g_early > 1e3 * g_fixed, (g_early, g_fixed)


@pytest.mark.parametrize("n", [1, 3, 10])
def test_length_normalisation_weights_are_the_models_variances(oracle, n):
    """Plda::TransformIvector's length normalisation divides by sqrt(sum_d t_d^2 / (psi_d + 1/n) / D).  Independent
    reading: in the model's own coordinates the mean of n utterances of a class is N(0, diag(psi + 1/n)), so that sum is a
    chi-square with D degrees of freedom over D -- its AVERAGE over draws from the model must be 1 (here 1 +- 4 sigma of
    the sampling error), for n = 1 and for n > 1 alike; a reading with the wrong variance (psi + 1, psi / n + 1, ...) fails
    for n > 1.  The factor is recovered from the oracle's outputs alone: normalised / un-normalised vector."""
    D, draws = 24, 4000
    rng = np.random.default_rng(40 + n)
    psi = np.sort(rng.random(D) * 3.0 + 0.05)[::-1].copy()
    q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    T = q * (1.0 + rng.random(D))[:, None]
    Tinv = np.linalg.inv(T)
    mean = rng.standard_normal(D)
    model = dict(mean=mean, transform=T, psi=psi, offset=-T @ mean)
    # class means of n utterances drawn in the model's coordinates, mapped back to the input space
    t = rng.standard_normal((draws, D)) * np.sqrt(psi + 1.0 / n)
    xs = t @ Tinv.T + mean
    inv_f2 = np.empty(draws)
    for j in range(draws):
        u = oracle.transform_ivector(model, xs[j], n, normalize_length=True)
        v = oracle.transform_ivector(model, xs[j], n, normalize_length=False)
        assert np.abs(v - t[j]).max() < 1e-9                     # the un-normalised transform is the model's coordinates
        inv_f2[j] = (v @ v) / (u @ u)                            # 1 / factor^2
        assert abs(np.sum(u * u / (psi + 1.0 / n)) - D) < 1e-9 * D
    sigma = np.sqrt(2.0 / D / draws)                             # std of a chi2_D / D average over `draws` draws
    assert abs(inv_f2.mean() - 1.0) < 4 * sigma, (inv_f2.mean(), sigma)


def test_det_restatement_is_consistent_with_farfrr_and_eer():
    """oracle det(): the thresholds run from the smallest to the largest score, FAR falls from 1, FRR rises from 0, every
    point IS farfrr at its threshold, and the curve brackets the EER the eer() restatement finds (scoring/eer.py:34-76)."""
    from oracle import plda_oracle_np as onp
    rng = np.random.default_rng(5)
    pos = rng.normal(1.5, 1.0, 4000).astype(np.float32); neg = rng.normal(-0.5, 1.0, 60000).astype(np.float32)
    thr, far, frr = onp.det(neg, pos, 100)
    assert thr[0] == min(pos.min(), neg.min()) and abs(thr[-1] - max(pos.max(), neg.max())) < 1e-9
    assert far[0] == 1.0 and frr[0] == 0.0 and (np.diff(far) <= 0).all() and (np.diff(frr) >= 0).all()
    for i in (0, 17, 50, 99):
        assert far[i] == (neg.astype(np.float64) >= thr[i]).mean() and frr[i] == (pos.astype(np.float64) < thr[i]).mean()
    t, f1, f2, e = onp.eer(neg, pos)
    j = int(np.searchsorted(thr, t))
    assert far[j - 1] >= f1 >= far[min(j, 99)] and frr[j - 1] <= f2 <= frr[min(j, 99)]
    d = onp.ppndf(np.array([0.0, 0.5, 1.0]))
    assert np.isfinite(d).all() and d[1] == 0.0 and abs(d[0] + d[2]) < 1e-5 and d[0] < -8.0      # (1 - eps rounds: not exactly symmetric)
