"""GPU: the fp64 GEMM behind fit, GetOutput and the LDA (Kaldi / ATLAS dgemm in the reference, reached through
pldamodule.cpp:76-106) on its own, through plda_gemm_f64: every dispatch class -- one 16 x 16 tile per workgroup
(M, N, K <= 256), the panel kernel, the 64 x 64 and 128 x 128 tiles with and without split-K -- both operand layouts,
alpha / beta, batches, k-weights, ragged sizes, against NumPy in fp64."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(eng, rng, m, n, k, ta, tb, alpha, beta, batch=None, kw=False):
    sa = (k, m) if ta else (m, k)
    sb = (n, k) if tb else (k, n)
    if batch:
        sa, sb = (batch,) + sa, (batch,) + sb
    A, B = rng.standard_normal(sa), rng.standard_normal(sb)
    C0 = rng.standard_normal(((batch,) if batch else ()) + (m, n))
    w = rng.random(k) + 0.5 if kw else None
    got = eng.gemm_f64(A, B, alpha, beta, C0, ta, tb, w)
    opA = np.swapaxes(A, -1, -2) if ta else A
    opB = np.swapaxes(B, -1, -2) if tb else B
    if w is not None:
        opA = opA * w
    want = alpha * (opA @ opB) + beta * C0
    tol = 4e-16 * k * max(1.0, np.abs(opA).max() * np.abs(opB).max()) * abs(alpha) + 1e-15 * np.abs(beta * C0).max() + 1e-15
    assert np.abs(got - want).max() <= tol, (m, n, k, ta, tb, np.abs(got - want).max(), tol)


def test_small_products_every_layout():
    from plda_amd import MPlda
    eng = MPlda(0)
    rng = np.random.default_rng(0)
    for (m, n, k) in ((1, 1, 1), (3, 5, 2), (16, 16, 4), (17, 33, 5), (64, 64, 64), (200, 200, 200), (199, 201, 203),
                      (256, 256, 256), (255, 1, 256), (1, 256, 255), (40, 200, 13)):
        for ta in (False, True):
            for tb in (False, True):
                _check(eng, rng, m, n, k, ta, tb, 1.0, 0.0)
    _check(eng, rng, 200, 200, 200, False, True, -1.0, 1.0)
    _check(eng, rng, 100, 120, 77, True, False, 0.37, -2.5)


def test_batches_share_nothing():
    from plda_amd import MPlda
    eng = MPlda(0)
    rng = np.random.default_rng(1)
    _check(eng, rng, 200, 200, 200, False, False, 1.0, 0.0, batch=5)
    _check(eng, rng, 64, 48, 200, False, True, -1.0, 1.0, batch=35)
    _check(eng, rng, 300, 300, 300, True, False, 1.0, 0.0, batch=3)       # panel kernel, batched
    _check(eng, rng, 512, 512, 512, False, False, 1.0, 1.0, batch=2)


def test_deep_and_large_products():
    from plda_amd import MPlda
    eng = MPlda(0)
    rng = np.random.default_rng(2)
    _check(eng, rng, 200, 200, 5000, True, False, 1.0, 0.0)               # X^T X shape: panel kernel in chunks
    _check(eng, rng, 512, 512, 10000, True, False, 1.0, 0.0)              # tiles + split-K
    _check(eng, rng, 1000, 300, 700, False, True, 2.0, 0.5)
    _check(eng, rng, 130, 2000, 129, False, False, 1.0, 0.0)
    _check(eng, rng, 200, 200, 40000, True, False, 1.0, 0.0, kw=True)     # the scatter's weighted contraction
    _check(eng, rng, 260, 260, 3000, True, False, 1.0, 0.0, kw=True)


@pytest.mark.parametrize("d", [210, 256, 300, 320, 384, 450, 512])
def test_symmetric_products_one_read_kernel(d):
    """X^T diag(w) X with both operands the SAME array (what fit's statistics pass computes, pldamodule.cpp:94-98): for
    208 < D <= 512 the block kernel of round 4 (csrc/syrk_blk.inc: 64 x 64 blocks dealt to the waves of up to four
    workgroups, full rows staged once by LDS DMA).  Row counts around the 16-row stage and the group split (fewer
    stages than groups, ragged last stage, one row), with and without weights, alpha / beta; exactly symmetric output."""
    from plda_amd import MPlda
    eng = MPlda(0)
    rng = np.random.default_rng(d)
    for k in (2048, 2049, 2063, 5000, 40000):
        for kw in (False, True):
            X = rng.standard_normal((k, d))
            w = rng.random(k) + 0.5 if kw else None
            C0 = rng.standard_normal((d, d)); C0 = C0 + C0.T
            alpha, beta = (1.0, 0.0) if k != 5000 else (-0.5, 2.0)
            got = eng.gemm_f64(X, X, alpha, beta, C0, True, False, w)
            want = alpha * ((X.T * w) @ X if kw else X.T @ X) + beta * C0
            tol = 4e-16 * k * max(1.0, np.abs(X).max() ** 2) * abs(alpha) * 1.5 + 1e-15 * np.abs(beta * C0).max() + 1e-15
            assert np.abs(got - want).max() <= tol, (d, k, kw, np.abs(got - want).max(), tol)
            assert np.array_equal(got, got.T)
