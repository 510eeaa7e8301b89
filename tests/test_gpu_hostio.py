"""GPU: the host-pointer entry points a user of the reference's API actually calls (NumPy arrays in, NumPy arrays
out: pldamodule.cpp:42-277) -- the pinned ring + copy threads behind them (csrc/hostio.hip) must change nothing
but the speed: the pipelined trials matrix is bit-identical to the device-pointer call and to the serial arm
(PLDA_HOST_VARIANT=1), uploads through the ring give bit-identical fits and transforms, and the one-trial host
path of score() agrees with the GPU's fp64 trial-list kernel and with the oracle."""
import ctypes as C

import numpy as np
import pytest

from conftest import make_data

pytestmark = pytest.mark.gpu


def _model(d, seed=3):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    return rng.random(d), q * (1.0 + rng.random(d))[:, None], np.sort(rng.random(d) * 4.0 + 0.05)[::-1].copy()


@pytest.mark.parametrize("m,nt,mixed,znorm", [(2500, 40000, False, False),    # 7 slabs of 384 rows through 3 slots
                                              (700, 90000, True, True),       # 186-row slabs (not a multiple of 128)
                                              (300, 1000, False, True)])      # one slab
def test_pipelined_host_matrix_is_bit_identical(monkeypatch, m, nt, mixed, znorm):
    import torch
    from plda_amd import MPlda
    d = 48
    eng = MPlda(0)
    eng.set_model(*_model(d))
    rng = np.random.default_rng(m)
    U, V = rng.standard_normal((m, d)), rng.standard_normal((nt, d))
    n = rng.integers(1, 6, m).astype(np.int32) if mixed else None
    zm = rng.standard_normal(m) if znorm else None
    zs = 0.5 + rng.random(m) if znorm else None
    p = lambda a: a.ctypes.data if a is not None else None    # noqa: E731
    out = np.full((m, nt), np.nan, np.float32)
    eng._ck(eng._lib.plda_score_matrix(eng._h, p(U), p(n), 0 if mixed else 2, m, p(V), nt, p(zm), p(zs), p(out), nt))
    # device-pointer call on the same inputs
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(a).to(dev) if a is not None else None   # noqa: E731
    dU, dV, dn, dzm, dzs = t(U), t(V), t(n), t(zm), t(zs)
    ref = torch.empty((m, nt), dtype=torch.float32, device=dev)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.score_matrix_dev(dU.data_ptr(), dn.data_ptr() if mixed else None, 0 if mixed else 2, m, dV.data_ptr(), nt,
                         ref.data_ptr(), nt, dzmean=dzm.data_ptr() if znorm else None, dzstd=dzs.data_ptr() if znorm else None)
    torch.cuda.synchronize()
    eng.set_stream(None)
    assert np.array_equal(out, ref.cpu().numpy())
    # strided output: the padding columns are not touched
    ld = nt + 37
    wide = np.full((m, ld), -7.0, np.float32)
    eng._ck(eng._lib.plda_score_matrix(eng._h, p(U), p(n), 0 if mixed else 2, m, p(V), nt, p(zm), p(zs), p(wide), ld))
    assert np.array_equal(wide[:, :nt], out) and (wide[:, nt:] == -7.0).all()
    # the serial arm of rounds 1-2
    monkeypatch.setenv("PLDA_HOST_VARIANT", "1")
    old = MPlda(0)
    old.set_model(*_model(d))
    out1 = np.empty((m, nt), np.float32)
    old._ck(old._lib.plda_score_matrix(old._h, p(U), p(n), 0 if mixed else 2, m, p(V), nt, p(zm), p(zs), p(out1), nt))
    assert np.array_equal(out1, out)


def test_uploads_through_the_ring_change_nothing(monkeypatch):
    """fit / transform / norm on arrays large enough (> 4 MiB) to travel through the pinned ring against the serial arm."""
    from plda_amd import MPlda
    x, y = make_data(7, 40000, 64, 500, skew=True, scale_between=0.4)     # 20 MB of rows
    a = MPlda(0)
    a.fit(x, y, 3)
    monkeypatch.setenv("PLDA_HOST_VARIANT", "1")
    b = MPlda(0)
    b.fit(x, y, 3)
    ma, mb = a.get_model(), b.get_model()
    assert np.array_equal(ma["transform"], mb["transform"]) and np.array_equal(ma["psi"], mb["psi"])
    ta, tb = a.transform(x, y), b.transform(x, y)
    assert list(ta) == list(tb) == sorted(ta)
    assert all(ta[k][0] == tb[k][0] and np.array_equal(ta[k][1], tb[k][1]) for k in ta)
    ra, rb = a.transform_array(x, 1), b.transform_array(x, 1)            # 20 MB back through the ring
    assert np.array_equal(ra, rb)
    models = {int(k): ta[k] for k in list(ta)[:50]}
    a.norm(x, models); b.norm(x, models)
    assert a.znorm_stats() == b.znorm_stats()


def test_one_trial_host_path_matches_the_gpu_kernel_and_the_oracle(oracle):
    """plda_score_one (host mirror of psi) against plda_score_pairs (GPU, P = 1) and Plda::LogLikelihoodRatio as the
    oracle restates it; with and without z-norm statistics, several counts (the per-count cache), after the model
    changes (truncate, smooth: the cache must be invalidated)."""
    from plda_amd import MPlda
    d = 200
    x, y = make_data(11, 3000, d, 60, scale_between=0.5)
    eng = MPlda(0)
    eng.fit(x, y, 5)
    psi = eng.get_model()["psi"]          # the LLR is checked on the engine's own model: arithmetic only
    tr = eng.transform(x[:400], y[:400])
    keys = list(tr)
    eng.norm(x[1000:1300], {k: tr[k] for k in keys[:10]})
    zmean, zstd = eng.znorm_stats()
    worst = 0.0
    for i, k in enumerate(keys[:25]):
        for n in (tr[k][0], 1, 7):
            e = (n, tr[k][1]); tvec = tr[keys[(i * 7 + 3) % len(keys)]]
            got, dev = eng.score(k, e, tvec), eng.score_on_device(k, e, tvec)
            want = oracle.llr(psi, e[1], n, tvec[1])
            if k in zmean:
                want = (want - zmean[k]) / zstd[k]
            assert abs(got - dev) <= 1e-10 * max(1.0, abs(dev))      # (z-scores divide by a std of ~1e-4)
            worst = max(worst, abs(got - want) / max(1.0, abs(want)))
    assert worst < 1e-10, worst
    # model changes invalidate the per-count cache
    eng.truncate(150)
    tr2 = eng.transform(x[:400], y[:400])
    k = keys[0]
    assert abs(eng.score(k, tr2[k], tr2[keys[1]]) - eng.score_on_device(k, tr2[k], tr2[keys[1]])) < 1e-10
    eng.smooth(0.5)
    tr3 = eng.transform(x[:400], y[:400])
    assert abs(eng.score(k, tr3[k], tr3[keys[1]]) - eng.score_on_device(k, tr3[k], tr3[keys[1]])) < 1e-10
    with pytest.raises(ValueError):
        eng.score(k, tr[k], tr[keys[1]])            # 200-dim vectors against the 150-dim model
    with pytest.raises(RuntimeError):
        eng.score(k, (0, tr3[k][1]), tr3[keys[1]])  # num_examples must be > 0
