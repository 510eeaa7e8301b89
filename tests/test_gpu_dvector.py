"""GPU parity of the d-vector front-end against the NumPy restatement of
scoring/extractdvector.py:19-59 (tests may use the oracle package)."""
import numpy as np
import pytest

from oracle import plda_oracle_np as onp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 2e-7)])
@pytest.mark.parametrize("method", ["mean", "max", "var"])
@pytest.mark.parametrize("l2norm", [True, False])
def test_pool_matches_numpy(dtype, tol, method, l2norm):
    from plda_amd import dvector
    rng = np.random.default_rng(3)
    lens = [1, 2, 3, 17, 64, 257, 5, 1000]
    for d in (10, 16, 32, 64, 128, 256, 300, 1024):
        frames = (rng.standard_normal((sum(lens), d)) * 3).astype(dtype)
        off = np.concatenate([[0], np.cumsum(lens)])
        got = dvector.pool(frames, off, method, l2norm)
        ref = onp.dvector_pool(frames.astype(np.float64), off, method, l2norm)
        scale = np.abs(ref).max()
        assert got.shape == ref.shape and got.dtype == np.float64
        assert np.abs(got - ref).max() <= tol * max(scale, 1.0) * (50 if method == "var" and not l2norm else 1)


def test_reference_named_functions_and_ragged():
    from plda_amd import dvector
    rng = np.random.default_rng(4)
    utt = rng.random((37, 40))
    np.testing.assert_allclose(dvector.extractdvectormean(utt), onp.dvector_pool(utt, [0, 37], "mean")[0], rtol=1e-12)
    np.testing.assert_allclose(dvector.extractdvectormax(utt), onp.dvector_pool(utt, [0, 37], "max")[0], rtol=1e-12)
    np.testing.assert_allclose(dvector.extractdvectorvar(utt), onp.dvector_pool(utt, [0, 37], "var")[0], rtol=1e-10, atol=1e-15)
    utts = [rng.random((n, 24)) for n in (3, 1, 50)]
    got = dvector.pool_utterances(utts, "mean")
    for g, u in zip(got, utts):
        np.testing.assert_allclose(g, onp.dvector_pool(u, [0, len(u)], "mean")[0], rtol=1e-12)
    # empty utterance -> NaN row (np.mean of an empty slice); zero frame -> NaN like 0/0 in numpy
    got = dvector.pool(np.zeros((2, 8)), np.array([0, 0, 2]), "mean")
    assert np.isnan(got[0]).all() and np.isnan(got[1]).all()


def test_frontend_into_plda_end_to_end(oracle):
    """frames -> d-vectors -> fit -> score: the whole chain of scoring/scorePLDA.py on the GPU."""
    from liblda import PLDA
    from plda_amd import dvector
    rng = np.random.default_rng(6)
    spk = rng.standard_normal((12, 16))
    utts, labels = [], []
    for s in range(12):
        for _ in range(6):
            t = int(rng.integers(20, 60))
            utts.append(spk[s] + 0.5 * rng.standard_normal((t, 16)))
            labels.append(s)
    dv = dvector.pool_utterances(utts, "mean")
    ref_dv = np.stack([onp.dvector_pool(u, [0, len(u)], "mean")[0] for u in utts])
    np.testing.assert_allclose(dv, ref_dv, rtol=1e-11, atol=1e-13)
    y = np.array(labels, np.uint64)
    p = PLDA(); p.fit(dv, y, 5)
    ref = oracle.fit(ref_dv, y, 5)
    np.testing.assert_allclose(p._instance.get_model()["psi"], ref["psi"], rtol=1e-7, atol=1e-12)
