"""GPU: the SPD inverse of the EM's E-step (Kaldi SpMatrix::Invert at ivector/plda.cc:436-447) on its own, through
plda_spd_inverse: the scalar sweep in registers (D <= 64), the block sweep on the fp64 matrix cores (D <= 256: 16
pivots per step, tiles of the triangle in MFMA accumulators) and the blocked whitening above, against
numpy.linalg.inv in the residual norm that the condition number allows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _spd(d, cond, seed):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    lam = np.exp(np.linspace(0.0, np.log(cond), d))
    a = (q * lam) @ q.T
    return 0.5 * (a + a.T)


@pytest.mark.parametrize("variant", ["0", "2", "1"])
@pytest.mark.parametrize("d", [1, 7, 16, 17, 64, 65, 80, 81, 96, 127, 128, 150, 192, 200, 207, 208, 209, 224, 240, 255, 256])
def test_small_inverse_every_block_count(monkeypatch, d, variant):
    from plda_amd import MPlda
    monkeypatch.setenv("PLDA_SWEEP_VARIANT", variant)
    eng = MPlda(0)
    for cond, seed in ((10.0, 1), (1e6, 2)):
        a = _spd(d, cond, seed + d)
        got = eng.spd_inverse(a)
        assert np.array_equal(got, got.T)
        res = np.abs(got @ a - np.eye(d)).max()
        assert res < 5e-16 * cond * d + 1e-13, (d, cond, res)
        want = np.linalg.inv(a)
        assert np.abs(got - want).max() <= 1e-15 * cond * d * np.abs(want).max() + 1e-14


def test_block_sweep_agrees_with_the_scalar_sweep(monkeypatch):
    """The block form is the same elimination grouped by 16 pivots: entries agree to rounding."""
    from plda_amd import MPlda
    a = _spd(200, 1e4, 5)
    monkeypatch.setenv("PLDA_SWEEP_VARIANT", "0")
    x0 = MPlda(0).spd_inverse(a)
    monkeypatch.setenv("PLDA_SWEEP_VARIANT", "2")
    x2 = MPlda(0).spd_inverse(a)
    assert np.abs(x0 - x2).max() <= 1e-12 * np.abs(x2).max()


@pytest.mark.parametrize("d", [257, 300, 512])
def test_blocked_inverse(d):
    from plda_amd import MPlda
    a = _spd(d, 1e3, d)
    got = MPlda(0).spd_inverse(a)
    assert np.abs(got @ a - np.eye(d)).max() < 1e-10


def test_not_positive_definite_is_an_error():
    from plda_amd import MPlda
    a = _spd(100, 10.0, 3)
    a[50, 50] = -1.0
    with pytest.raises(RuntimeError):
        MPlda(0).spd_inverse(a)
