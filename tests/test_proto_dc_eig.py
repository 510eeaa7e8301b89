"""CPU: the NumPy prototype of the device eigensolver (scripts/proto_dc_eig.py: Householder tridiagonalisation +
divide and conquer with dlaed2-style deflation, two-pole secular iteration, Gu-Eisenstat vectors) against
numpy.linalg.eigh on the hard cases.  csrc/eig_dc.hip implements the same rules (tests/test_gpu_eig.py checks the
kernels); this keeps the restatement they were developed against honest."""
import importlib.util
import os

import numpy as np
import pytest

_spec = importlib.util.spec_from_file_location(
    "proto_dc_eig", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "proto_dc_eig.py"))
proto = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(proto)


def _cases(n, rng):
    A = rng.standard_normal((n, n))
    yield "gaussian", A + A.T
    B = rng.standard_normal((n, max(n // 3, 1)))
    yield "rank-deficient", B @ B.T
    yield "identity", np.eye(n)
    yield "zero", np.zeros((n, n))
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    yield "two clusters", (q * np.concatenate([np.ones(n // 2), np.full(n - n // 2, 2.0)])) @ q.T
    yield "graded", (q * 10.0 ** (-np.arange(n) * 16.0 / n)) @ q.T
    yield "scaled 1e-150", (A + A.T) * 1e-150


@pytest.mark.parametrize("n", [1, 2, 5, 17, 33, 64, 100])
def test_prototype_matches_numpy(n):
    rng = np.random.default_rng(n)
    for name, G in _cases(n, rng):
        G = 0.5 * (G + G.T)
        lam, Z = proto.sym_eig(G)
        ref = np.linalg.eigvalsh(G)[::-1]
        nrm = max(np.abs(ref).max(), 1e-300)
        assert np.abs(lam - ref).max() / nrm < 1e-13, (name, n)
        assert np.abs(Z.T @ Z - np.eye(n)).max() < 1e-12, (name, n)
        assert np.abs(G @ Z - Z * lam[None, :]).max() / nrm < 1e-12, (name, n)


def test_secular_roots_interlace():
    """roots of 1 + rho sum z_i^2 / (d_i - lam): exactly one in each (d_j, d_j+1), the last in (d_k, d_k + rho |z|^2]."""
    rng = np.random.default_rng(3)
    d = np.sort(rng.random(40)) ; z = rng.standard_normal(40) ; z /= np.linalg.norm(z)
    org, mu, delta = proto.secular_roots(d, z, 0.7)
    lam = d[org] + mu
    assert (lam[:-1] > d[:-1]).all() and (lam[:-1] < d[1:]).all() and d[-1] < lam[-1] <= d[-1] + 0.7 + 1e-15
    ref = np.linalg.eigvalsh(np.diag(d) + 0.7 * np.outer(z, z))
    assert np.abs(lam - ref).max() < 1e-14
