"""GPU: the symmetric eigensolver of GetOutput (csrc/eig_dc.hip: Householder tridiagonalisation + divide and
conquer + back-transformation; csrc/linalg.hip: block Jacobi) through plda_sym_eig, against numpy.linalg.eigh.
The reference reaches this step through Kaldi's SpMatrix::Eig inside PldaEstimator::GetOutput
(pldamodule.cpp:102-106); what GetOutput needs of it is: eigenvalues, an ORTHONORMAL set of eigenvectors
(any basis inside a cluster), small residual.  Hard cases: rank-deficient (the between-class covariance of fewer
speakers than dimensions), clusters, graded spectra, already-tridiagonal and diagonal input, extreme scales."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases(n, rng):
    A = rng.standard_normal((n, n))
    yield "gaussian", A + A.T
    B = rng.standard_normal((n, max(n // 3, 1)))
    yield "rank-deficient PSD", B @ B.T
    yield "identity", np.eye(n)
    yield "zero", np.zeros((n, n))
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.concatenate([np.ones(n // 2), np.full(n - n // 2, 2.0)])
    yield "two clusters", (q * lam) @ q.T
    yield "graded", (q * 10.0 ** (-np.arange(n) * 16.0 / n)) @ q.T
    yield "tridiagonal", (np.diag(np.abs(np.arange(n) - n // 2).astype(float)) + np.diag(np.ones(n - 1), 1)
                          + np.diag(np.ones(n - 1), -1))
    yield "diag + tiny coupling", np.diag(rng.random(n)) + 1e-12 * (A + A.T)
    yield "scaled 1e150", (A + A.T) * 1e150
    yield "scaled 1e-150", (A + A.T) * 1e-150


def _check(eng, name, G, method, expect_method=None, tol=5e-13):
    n = G.shape[0]
    G = 0.5 * (G + G.T)
    lam, V, used = eng.sym_eig(G, method)
    if expect_method is not None:
        assert used == expect_method, (name, n, used)
    ref = np.linalg.eigvalsh(G)[::-1]
    nrm = max(np.abs(ref).max(), 1e-300)
    assert np.all(np.diff(lam) <= 0), (name, n)
    e_val = np.abs(lam - ref).max() / nrm
    e_orth = np.abs(V @ V.T - np.eye(n)).max()
    Gs = G / nrm
    e_res = np.abs(V @ Gs - (lam / nrm)[:, None] * V).max()
    assert e_val < tol and e_orth < tol and e_res < tol, (name, n, method, e_val, e_orth, e_res)
    return e_val, e_orth, e_res


@pytest.mark.parametrize("n", [1, 2, 3, 5, 16, 17, 33, 64, 100, 200, 224, 256])
def test_direct_method_against_numpy(n):
    from plda_amd import MPlda
    eng = MPlda(0)
    rng = np.random.default_rng(n)
    for name, G in _cases(n, rng):
        _check(eng, name, G, method=2, expect_method=2)


@pytest.mark.parametrize("n", [161, 225, 257, 320, 512, 513, 1000, 1024, 1025, 1536, 2048])
def test_direct_method_large(n):
    """n > 160: the tridiagonalisation runs on ceil(n / 8) cooperating workgroups (register layouts of 7, 8, 16, 32 and
    64 elements per lane: the sizes straddle their limits; 2048 is the largest supported -- 256 workgroups, one per CU;
    above 1024 the row registers spill and only the direct method exists, the block Jacobi fallback stops at 1024)."""
    from plda_amd import MPlda
    eng = MPlda(0)
    rng = np.random.default_rng(n)
    for name, G in _cases(n, rng):
        _check(eng, name, G, method=2, expect_method=2, tol=2e-12)


@pytest.mark.parametrize("variant", ["2", "3"])
def test_both_tridiagonalisation_kernels(variant, monkeypatch):
    """PLDA_EIG_VARIANT=2: one workgroup, matrix in registers (n <= 256); 3: rows over cooperating workgroups."""
    monkeypatch.setenv("PLDA_EIG_VARIANT", variant)
    from plda_amd import MPlda
    eng = MPlda(0)
    for n in (3, 17, 100, 200, 256):
        rng = np.random.default_rng(1000 + n)
        for name, G in _cases(n, rng):
            _check(eng, name, G, method=2, expect_method=2)


def test_back_transformation_two_row_arm(monkeypatch):
    """PLDA_EIG_VARIANT=4: the default kernels with the two-rows-per-wave back-transformation of rounds 2-3 (the A/B arm of
    the one-row kernel with the eight-sums-at-once reduction that n <= 512 takes since round 4)."""
    monkeypatch.setenv("PLDA_EIG_VARIANT", "4")
    from plda_amd import MPlda
    eng = MPlda(0)
    for n in (9, 64, 200, 257, 512):
        rng = np.random.default_rng(2000 + n)
        for name, G in _cases(n, rng):
            _check(eng, name, G, method=2, expect_method=2, tol=2e-12)


@pytest.mark.parametrize("n", [5, 64, 200])
def test_jacobi_against_numpy(n):
    from plda_amd import MPlda
    eng = MPlda(0)
    rng = np.random.default_rng(100 + n)
    for name, G in _cases(n, rng):
        if name.startswith("scaled") or (name == "graded" and n > 64):
            # the Jacobi arm (fallback + warm starts of the per-iteration EM arm) is used unscaled and works on the
            # rows of G itself: with a spectrum graded over 16 decades the rows of the smallest eigenvalues are
            # rounding noise and its relative stopping test is never met (40 sweeps -> PLDA_E_NUMERIC).  The
            # direct method above covers these inputs.
            continue
        _check(eng, name, G, method=1, expect_method=1, tol=2e-11)   # stops at |cos| <= 4 eps sqrt(D) between rows


def test_default_dispatch_and_plda_like_input():
    """what fit uses: whitened between-class covariance of a PLDA-like problem, D = 200."""
    from plda_amd import MPlda
    eng = MPlda(0)
    rng = np.random.default_rng(7)
    D, K = 200, 1000
    W = np.cov(rng.standard_normal((D, 5000)))
    Bm = rng.standard_normal((D, K)) * (np.arange(D)[:, None] + 1.0) ** -1.0
    T1 = np.linalg.inv(np.linalg.cholesky(W))
    G = T1 @ (Bm @ Bm.T / K) @ T1.T
    _check(eng, "plda-like", G, method=0, expect_method=2)
    # fewer speakers than dimensions: D - K + 1 zero eigenvalues
    Bm = rng.standard_normal((D, 40))
    G = T1 @ (Bm @ Bm.T / 40) @ T1.T
    _check(eng, "plda-like rank 40", G, method=0, expect_method=2)


def test_non_finite_input_falls_back_or_fails_cleanly():
    from plda_amd import MPlda
    eng = MPlda(0)
    G = np.eye(8)
    G[2, 3] = G[3, 2] = np.nan
    with pytest.raises(Exception):
        eng.sym_eig(G, 2)
