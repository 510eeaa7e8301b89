"""GPU parity of the OPT-IN bf16x3 arm of the trials GEMM (PLDA_SCORE_DTYPE=bf16x3, csrc/score_bf16x3.inc): the fp32
contraction as three bf16 terms per operand, six v_mfma_f32_32x32x16_bf16 per 16 k.

Reference semantics as for every trials kernel: Plda::LogLikelihoodRatio through MPlda_score
(/root/reference/src/pldamodule.cpp:258-277), one call per trial from scoring/scorePLDA.py:302-318.  The arm is held to
the SAME tolerance against the fp64 oracle as the fp32 kernels (north_star: 1e-4 relative, `score_tol`), on its own --
not through agreement with the fp32 kernel: uniform and mixed (bucketed) enrol counts, z-normalised rows, ragged edges
in both dimensions, one and many tiles per workgroup, depths that are not a multiple of 16 (a zero k-oct is appended).
The default stays fp32 (north_star prescribes it); nothing here runs unless the variable is set at plda_create.
"""
import numpy as np
import pytest

from conftest import score_tol

pytestmark = pytest.mark.gpu


def _model(d, seed):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    T = q * (1.0 + rng.random(d))[:, None]
    psi = np.sort(rng.random(d) * 4.0 + 0.05)[::-1].copy()
    return rng.random(d), T, psi


def _engine(monkeypatch, d, seed=3, dtype="bf16x3"):
    from plda_amd import MPlda
    monkeypatch.delenv("PLDA_GEMM_VARIANT", raising=False)
    if dtype:
        monkeypatch.setenv("PLDA_SCORE_DTYPE", dtype)
    else:
        monkeypatch.delenv("PLDA_SCORE_DTYPE", raising=False)
    eng = MPlda(0)
    mean, T, psi = _model(d, seed)
    eng.set_model(mean, T, psi)
    return eng, psi


SHAPES = [(200, 300, 517), (64, 1024, 1024), (33, 257, 769), (200, 1, 700), (8, 513, 255), (24, 700, 2900), (512, 520, 600)]


@pytest.mark.parametrize("d,m,nt", SHAPES)
def test_bf16x3_uniform_and_mixed_vs_oracle(monkeypatch, oracle, d, m, nt):
    eng, psi = _engine(monkeypatch, d)
    rng = np.random.default_rng(900 + d + m)
    U, V = rng.standard_normal((m, d)), rng.standard_normal((nt, d))
    for n in (1, 7):
        ref = oracle.score_block(psi, U, n, V)
        got = eng.score_matrix((n, U), (1, V))
        assert eng.score_last_kernel() == "trials_gemm_bf16x3_kernel"
        assert (np.abs(got - ref) <= score_tol(ref)).all(), (n, np.abs(got - ref).max())
    counts = rng.integers(1, 6, m).astype(np.int32)
    ref = oracle.score_block(psi, U, counts, V)
    got = eng.score_matrix((counts, U), (1, V))
    assert (np.abs(got - ref) <= score_tol(ref)).all(), np.abs(got - ref).max()


@pytest.mark.parametrize("d,m,nt", [(72, 5300, 4200), (96, 1100, 19000), (200, 400, 35000)])
def test_bf16x3_patch_walks_bit_identical(monkeypatch, d, m, nt):
    """The arm's static walk of the 1024 x 2048 patches, along the rows (PLDA_GEMM_VARIANT=48) or down the columns (49; queue x
    owns columns x, x + 8, ..., the remainder dealt patch by patch -- score.hip: bt4_patch): the order only decides which
    workgroup computes a tile.  Shapes with 3, 10 and 18 patch columns, ragged edges."""
    rng = np.random.default_rng(d + nt)
    U, V = rng.standard_normal((m, d)), rng.standard_normal((nt, d))
    counts = rng.integers(1, 4, m).astype(np.int32)
    outs = []
    for variant in (48, 49):
        eng, _ = _engine(monkeypatch, d)
        monkeypatch.setenv("PLDA_GEMM_VARIANT", str(variant))
        from plda_amd import MPlda
        eng2 = MPlda(0)
        eng2.set_model(*_model(d, 3))
        outs.append((eng2.score_matrix((2, U), (1, V)), eng2.score_matrix((counts, U), (1, V))))
        assert eng2.score_last_kernel() == "trials_gemm_bf16x3_kernel"
    for a, b in zip(*outs):
        assert np.isfinite(a).all() and np.array_equal(a, b)


def test_bf16x3_znorm(monkeypatch, oracle):
    d, m, nt = 120, 300, 517
    eng, psi = _engine(monkeypatch, d)
    rng = np.random.default_rng(301)
    counts = rng.integers(1, 6, m).astype(np.int32)
    U, V = rng.standard_normal((m, d)), rng.standard_normal((nt, d))
    raw = oracle.score_block(psi, U, counts, V)
    zm, zs = raw.mean(1), raw.std(1)
    ids = np.arange(m, dtype=np.int64)
    eng._meanz = {int(k): float(v) for k, v in zip(ids, zm)}
    eng._stdvz = {int(k): float(v) for k, v in zip(ids, zs)}
    ref = oracle.score_block(psi, U, counts, V, zm, zs)
    got = eng.score_matrix((counts, U, ids), (1, V))
    assert (np.abs(got - ref) <= score_tol(ref)).all(), np.abs(got - ref).max()


@pytest.mark.parametrize("d,n_enrol,seed", [(200, 1, 21), (512, 100, 22), (256, None, 23)], ids=["C2_depth", "C3_depth", "C4_depth_mixed"])
def test_bf16x3_at_8192_squared(monkeypatch, d, n_enrol, seed):
    """The BASELINE depths on 8192 x 8192 (1024 tiles, four per workgroup) against the fp64 GEMM-form oracle, and beside the
    fp32 kernel's scores: the two differ by the order of the additions and three dropped sub-ulp products."""
    import torch
    from oracle import plda_oracle_np as onp
    dev = torch.device("cuda", 0)
    m = nt = 8192
    rng = np.random.default_rng(seed)
    U, V = rng.standard_normal((m, d)), rng.standard_normal((nt, d))
    n = rng.integers(1, 6, m).astype(np.int32) if n_enrol is None else n_enrol
    ref = onp.llr_matrix(_model(d, seed)[2], U, n, V)
    dU, dV = torch.from_numpy(U).to(dev), torch.from_numpy(V).to(dev)
    dn = torch.from_numpy(n).to(dev) if n_enrol is None else None
    outs = {}
    for dtype in ("bf16x3", None):
        eng, _ = _engine(monkeypatch, d, seed, dtype)
        eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        out = torch.full((m, nt), float("nan"), dtype=torch.float32, device=dev)
        eng.score_matrix_dev(dU.data_ptr(), dn.data_ptr() if dn is not None else None, 0 if dn is not None else int(n_enrol), m,
                             dV.data_ptr(), nt, out.data_ptr(), nt)
        torch.cuda.synchronize()
        outs[dtype] = out.cpu().numpy().astype(np.float64)
        eng.set_stream(None)
    got = outs["bf16x3"]
    assert np.isfinite(got).all()
    assert (np.abs(got - ref) <= score_tol(ref)).all(), np.abs(got - ref).max()
    assert np.abs(got - outs[None]).max() <= 0.5 * score_tol(ref).min(), np.abs(got - outs[None]).max()


def test_bf16x3_bad_value_is_refused(monkeypatch):
    from plda_amd import MPlda
    monkeypatch.setenv("PLDA_SCORE_DTYPE", "fp8")
    with pytest.raises(Exception, match="PLDA_SCORE_DTYPE"):
        MPlda(0)
