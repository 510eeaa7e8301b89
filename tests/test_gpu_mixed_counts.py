"""GPU parity of the trials matrix with MIXED enrol counts in the bucketed form (round 5).

Reference semantics: Plda::LogLikelihoodRatio(u, n_i, v) through MPlda_score
(/root/reference/src/pldamodule.cpp:258-277), one call per trial from scoring/scorePLDA.py:302-318, where
every enrol model brings its own utterance count n_i.  SURVEY.md Appendix A.5 turns the n-dependent part of
the LLR into a second operand half (GEMM depth 2 D); `csrc/score.hip` now carries it as one column-bias vector per
DISTINCT count: depth D + G - 1.  Checked here against the per-trial fp64 oracle `score_block`:

  * G = 1 (a count array whose values are all equal -> the uniform path), 2, 5 (BASELINE C4: n in 1..5), 9 (exactly one
    extra 8-k step), 10 (two), 40; counts at the edge of what a bucket takes (4095) and beyond it (-> depth-2D fallback);
    more distinct counts than pay at this dimension (-> fallback); PLDA_MIXED_VARIANT=1 (the depth-2D arm);
  * through every GEMM kernel (forced dispatch 20 / 30 / 40) -- same starting value and k order: same bits;
  * z-normalised rows (the map folded into r', s and the one-hot columns);
  * the host-pointer entry (counts found on the host), the device entry (counts found by the device pass), the sharded
    entry (found once per call), a prepared test side (plda_score_prepare_counts_dev) reused by calls whose counts are a
    subset of the prepared ones.
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


def _engine(monkeypatch, variant, d, seed=3, mixed_variant=None):
    from plda_amd import MPlda
    if variant is None:
        monkeypatch.delenv("PLDA_GEMM_VARIANT", raising=False)
    else:
        monkeypatch.setenv("PLDA_GEMM_VARIANT", str(variant))
    if mixed_variant is None:
        monkeypatch.delenv("PLDA_MIXED_VARIANT", raising=False)
    else:
        monkeypatch.setenv("PLDA_MIXED_VARIANT", str(mixed_variant))
    eng = MPlda(0)
    mean, T, psi = _model(d, seed)
    eng.set_model(mean, T, psi)
    return eng, psi


def _counts(rng, m, values):
    values = np.asarray(values, np.int32)
    c = values[rng.integers(0, len(values), m)]
    c[:len(values)] = values[:m]                   # every value present (m >= G in the cases below)
    return c.astype(np.int32)


# (name, distinct counts, expected depth as a function of d)
CASES = [
    ("G1", [4], lambda d: d),
    ("G2", [1, 3], lambda d: d + 1),
    ("G5_c4", [1, 2, 3, 4, 5], lambda d: d + 4),
    ("G9", list(range(1, 10)), lambda d: d + 8),
    ("G10", list(range(1, 11)), lambda d: d + 9),
    ("G40", list(range(1, 80, 2)), lambda d: d + 39),
    ("edge4095", [1, 17, 4095], lambda d: d + 2),
    ("over4095", [1, 17, 4096], lambda d: 2 * d),          # a count beyond a bucket's range: depth-2D form
]


@pytest.mark.parametrize("variant", [20, 30, 40])
@pytest.mark.parametrize("name,values,depth", CASES, ids=[c[0] for c in CASES])
def test_bucketed_counts_vs_oracle(monkeypatch, oracle, variant, name, values, depth):
    d, m, nt = 96, 300, 517
    eng, psi = _engine(monkeypatch, variant, d)
    rng = np.random.default_rng(500 + len(values))
    counts = _counts(rng, m, values)
    U, V = rng.standard_normal((m, d)), rng.standard_normal((nt, d))
    ref = oracle.score_block(psi, U, counts, V)
    got = eng.score_matrix((counts, U), (1, V))
    assert (np.abs(got - ref) <= score_tol(ref)).all(), (name, variant, np.abs(got - ref).max())
    if name != "G1":                                # (G1 goes through the wrapper's uniform shortcut)
        assert eng.score_last_shape()[2] == depth(d), (name, eng.score_last_shape())


@pytest.mark.parametrize("d,m,nt", [(256, 700, 300), (20, 1024, 1024), (8, 513, 255), (200, 600, 1100)])
def test_bucketed_shapes_and_kernels_agree(monkeypatch, oracle, d, m, nt):
    """C4-style counts on ragged shapes and small / large depths; the three GEMM kernels give the same bits; the
    depth-2D arm (PLDA_MIXED_VARIANT=1) agrees with the oracle too."""
    rng = np.random.default_rng(600 + d)
    counts = rng.integers(1, 6, m).astype(np.int32)
    U, V = rng.standard_normal((m, d)), rng.standard_normal((nt, d))
    outs = {}
    for variant in (20, 30, 40):
        eng, psi = _engine(monkeypatch, variant, d)
        outs[variant] = eng.score_matrix((counts, U), (1, V))
        assert eng.score_last_shape()[2] == d + 4
    ref = oracle.score_block(psi, U, counts, V)
    assert (np.abs(outs[20] - ref) <= score_tol(ref)).all(), np.abs(outs[20] - ref).max()
    assert np.array_equal(outs[30], outs[20]) and np.array_equal(outs[40], outs[20])
    eng, _ = _engine(monkeypatch, 20, d, mixed_variant=1)
    old = eng.score_matrix((counts, U), (1, V))
    assert eng.score_last_shape()[2] == 2 * d
    assert (np.abs(old - ref) <= score_tol(ref)).all(), np.abs(old - ref).max()


def test_too_many_distinct_counts_fall_back(monkeypatch, oracle):
    """G - 1 > max(D / 2, 8): the extra columns would not pay -> depth-2D form; > 64 distinct counts likewise."""
    rng = np.random.default_rng(7)
    for d, values in ((24, list(range(1, 15))), (200, list(range(1, 71)))):
        m, nt = 400, 300
        eng, psi = _engine(monkeypatch, 20, d)
        counts = _counts(rng, m, values)
        U, V = rng.standard_normal((m, d)), rng.standard_normal((nt, d))
        ref = oracle.score_block(psi, U, counts, V)
        got = eng.score_matrix((counts, U), (1, V))
        assert eng.score_last_shape()[2] == 2 * d
        assert (np.abs(got - ref) <= score_tol(ref)).all(), np.abs(got - ref).max()


@pytest.mark.parametrize("variant", [20, 40])
def test_bucketed_znorm(monkeypatch, oracle, variant):
    """z-norm (pldamodule.cpp:269-273) with mixed counts: s_i scales the A operand, the one-hot columns and q_0."""
    d, m, nt = 120, 300, 517
    eng, psi = _engine(monkeypatch, variant, d)
    rng = np.random.default_rng(301)
    counts = rng.integers(1, 6, m).astype(np.int32)
    U, V = rng.standard_normal((m, d)), rng.standard_normal((nt, d))
    raw = oracle.score_block(psi, U, counts, V)
    zm, zs = raw.mean(1), raw.std(1)
    zs[::17] = 0.0                                   # engine convention: std 0 -> row left un-normalised
    ids = np.arange(m, dtype=np.int64)
    eng._meanz = {int(k): float(v) for k, v in zip(ids, zm) if zs[k] != 0.0}
    eng._stdvz = {int(k): float(v) for k, v in zip(ids, zs) if zs[k] != 0.0}
    zs_o = np.where(zs == 0.0, 1.0, zs); zm_o = np.where(zs == 0.0, 0.0, zm)
    ref = oracle.score_block(psi, U, counts, V, zm_o, zs_o)
    got = eng.score_matrix((counts, U, ids), (1, V))
    assert eng.score_last_shape()[2] == d + 4
    assert (np.abs(got - ref) <= score_tol(ref)).all(), np.abs(got - ref).max()


def test_device_entry_and_prepared_counts(monkeypatch, oracle):
    """plda_score_matrix_dev finds the distinct counts on the device; a test side prepared for counts {1..6} is reused by
    calls bringing subsets of them ({1..5}, {2, 4}, {3} alone) and repacked for a call that brings a count outside; the
    depth-2D prepared form is used when that is what was prepared."""
    import torch
    dev = torch.device("cuda", 0)
    d, m, nt = 56, 700, 1300
    eng, psi = _engine(monkeypatch, None, d)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    rng = np.random.default_rng(21)
    Uh, Vh = rng.standard_normal((m, d)), rng.standard_normal((nt, d))
    U, V = torch.from_numpy(Uh).to(dev), torch.from_numpy(Vh).to(dev)

    def score(counts):
        dn = torch.from_numpy(np.ascontiguousarray(counts, np.int32)).to(dev)
        o = torch.empty((m, nt), dtype=torch.float32, device=dev)
        eng.score_matrix_dev(U.data_ptr(), dn.data_ptr(), 0, m, V.data_ptr(), nt, o.data_ptr(), nt)
        torch.cuda.synchronize()
        return o.cpu().numpy()

    sets = {"1..5": _counts(rng, m, [1, 2, 3, 4, 5]), "2,4": _counts(rng, m, [2, 4]), "3": np.full(m, 3, np.int32),
            "1..7": _counts(rng, m, list(range(1, 8)))}
    plain = {k: score(c) for k, c in sets.items()}
    for k, c in sets.items():
        ref = oracle.score_block(psi, Uh, c, Vh)
        assert (np.abs(plain[k] - ref) <= score_tol(ref)).all(), (k, np.abs(plain[k] - ref).max())
    assert eng.score_last_shape()[2] == d + 6
    # prepared for {1..6}: subsets reuse the packed side (depth d + 5 whatever the call brings), same scores within tolerance
    eng.score_prepare_counts_dev(V.data_ptr(), nt, [6, 5, 4, 3, 2, 1, 1, 3])
    for k in ("1..5", "2,4"):
        got = score(sets[k])
        assert eng.score_last_shape()[2] == d + 5, k
        ref = oracle.score_block(psi, Uh, sets[k], Vh)
        assert (np.abs(got - ref) <= score_tol(ref)).all(), (k, np.abs(got - ref).max())
    assert np.array_equal(score(sets["3"]), plain["3"])         # one distinct count: the uniform path, repacked
    eng.score_prepare_counts_dev(V.data_ptr(), nt, [1, 2, 3, 4, 5, 6])
    got = score(sets["1..7"])                                   # 7 is not among the prepared counts: packed for this call
    assert eng.score_last_shape()[2] == d + 6
    assert np.array_equal(got, plain["1..7"])
    # the depth-2D prepared form is what a later mixed call uses
    eng.score_prepare_dev(V.data_ptr(), nt, mixed_counts=True)
    got = score(sets["1..5"])
    assert eng.score_last_shape()[2] == 2 * d
    ref = oracle.score_block(psi, Uh, sets["1..5"], Vh)
    assert (np.abs(got - ref) <= score_tol(ref)).all()
    eng.score_unprepare()
    assert np.array_equal(score(sets["1..5"]), plain["1..5"])
    eng.set_stream(None)


def test_sharded_and_host_slabs_use_one_count_set(monkeypatch, oracle):
    """A blocked call packs the test side once: every block must see the count set of the WHOLE call, also a block whose
    own rows hold a single count.  Rows are ordered by count so that the first blocks are uniform."""
    import torch
    dev = torch.device("cuda", 0)
    d, m, nt = 40, 2100, 900
    eng, psi = _engine(monkeypatch, None, d)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    rng = np.random.default_rng(33)
    counts = np.sort(_counts(rng, m, [1, 2, 5, 9])).astype(np.int32)
    counts[:1024] = 1
    Uh, Vh = rng.standard_normal((m, d)), rng.standard_normal((nt, d))
    ref = oracle.score_block(psi, Uh, counts, Vh)
    U, V, dn = torch.from_numpy(Uh).to(dev), torch.from_numpy(Vh).to(dev), torch.from_numpy(counts).to(dev)
    one = torch.empty((m, nt), dtype=torch.float32, device=dev)
    eng.score_matrix_dev(U.data_ptr(), dn.data_ptr(), 0, m, V.data_ptr(), nt, one.data_ptr(), nt)
    for R in (1, 3):
        full = torch.full((m, nt), float("nan"), dtype=torch.float32, device=dev)
        for r in range(R):
            eng.comm_emulate(R, r)
            eng.score_matrix_sharded_dev(U.data_ptr(), dn.data_ptr(), 0, m, V.data_ptr(), nt, full.data_ptr(), nt, block_rows=256)
        eng.comm_emulate(1, 0)
        torch.cuda.synchronize()
        assert torch.equal(full, one), R
    got = one.cpu().numpy()
    assert (np.abs(got - ref) <= score_tol(ref)).all(), np.abs(got - ref).max()
    host = eng.score_matrix((counts, Uh), (1, Vh))
    assert np.array_equal(host, got)
    eng.set_stream(None)
