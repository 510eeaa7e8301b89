"""GPU EER vs the NumPy restatement of bob.measure's definition (oracle/plda_oracle_np.py:eer)."""
import numpy as np
import pytest

from oracle import plda_oracle_np as onp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from plda_amd import MPlda
    return MPlda(0)


@pytest.mark.parametrize("seed", list(range(24)))
@pytest.mark.parametrize("case", ["gauss", "ties", "separable", "inverted", "tiny", "wide"])
def test_eer_lists_match_restatement(eng, case, seed):
    from plda_amd import eer
    rng = np.random.default_rng(sum(map(ord, case)) + seed)   # (hash() of a str changes from process to process)
    if case == "gauss":
        pos, neg = rng.normal(2.0, 1.5, 5000), rng.normal(-1.0, 1.0, 200000)
    elif case == "ties":                         # heavy ties: quantised scores
        pos, neg = np.round(rng.normal(1.0, 1.0, 3000), 1), np.round(rng.normal(-0.5, 1.0, 50000), 1)
    elif case == "separable":
        pos, neg = rng.uniform(5, 6, 100), rng.uniform(-6, -5, 1000)
    elif case == "inverted":                     # worse than chance
        pos, neg = rng.normal(-2.0, 1.0, 400), rng.normal(2.0, 1.0, 4000)
    elif case == "tiny":
        pos, neg = np.array([0.5]), np.array([-0.25, 0.75])
    else:                                        # many exponents, both signs, zeros
        pos = np.concatenate([rng.normal(0, 1e-3, 500), rng.normal(50, 30, 500), [0.0, -0.0]])
        neg = np.concatenate([rng.normal(0, 1e-3, 5000), rng.normal(-50, 30, 5000), [0.0]])
    pos, neg = pos.astype(np.float32), neg.astype(np.float32)
    thr, far, frr, e = eer.eer_from_lists(eng, pos, neg)
    rthr, rfar, rfrr, re = onp.eer(neg, pos)
    assert (far, frr) == (rfar, rfrr), (far, frr, rfar, rfrr)
    assert thr == pytest.approx(rthr, rel=1e-12, abs=1e-300) and e == re
    # farfrr at the returned threshold reproduces the rates (bob.measure.farfrr)
    assert far == (neg.astype(np.float64) >= thr).mean() and frr == (pos.astype(np.float64) < thr).mean()


def test_eer_of_a_trials_matrix(eng, oracle):
    import torch
    from plda_amd import eer
    from conftest import make_data
    dev = torch.device("cuda", 0)
    x, y = make_data(51, 3000, 32, 60, scale_between=0.6)
    eng.fit(x, y, 5)
    tr = eng.transform_array(x, 1)
    U = torch.from_numpy(tr[:900]).to(dev); V = torch.from_numpy(tr[900:2900]).to(dev)
    S = torch.empty((900, 2000), dtype=torch.float32, device=dev)
    eng.score_matrix_dev(U.data_ptr(), None, 1, 900, V.data_ptr(), 2000, S.data_ptr(), 2000)
    eng.synchronize()
    es = torch.from_numpy(y[:900].astype(np.int64)).to(dev); ts = torch.from_numpy(y[900:2900].astype(np.int64)).to(dev)
    out = eer.eer_from_matrix_dev(eng, S.data_ptr(), 2000, 900, 2000, es.data_ptr(), ts.data_ptr())
    Sh = S.cpu().numpy()
    tgt = y[:900, None] == y[None, 900:2900]
    ref = onp.eer(Sh[~tgt], Sh[tgt])
    assert out[4] == tgt.sum() and out[5] == (~tgt).sum()
    assert tuple(out[1:4]) == ref[1:] and out[0] == pytest.approx(ref[0], rel=1e-12)
    assert 0.0 < out[3] < 0.5          # speaker structure => better than chance
    assert eer.format_line(out[1], out[2], out[0]).startswith("EER = ")
