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


@pytest.mark.parametrize("m,nt,ld,k", [(1, 2, 2, 1), (3, 5, 7, 2), (2, 1023, 1023, 3), (257, 1023, 1023, 9),
                                       (300, 1025, 1028, 16), (7, 4099, 4100, 5), (5000, 3001, 3001, 40)])
def test_eer_of_ragged_matrices(eng, m, nt, ld, k):
    """The matrix passes give each workgroup a strip of 1024 columns and a slice of the rows, four rows at a time:
    widths that are not a multiple of 4 or of 1024, padded and unaligned leading dimensions, fewer rows than one
    group of four, quantised scores (ties across the two classes)."""
    import torch
    from plda_amd import eer
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(m * 7919 + nt)
    es = rng.integers(0, k, m); ts = rng.integers(0, k, nt)
    es[0] = ts[0] = 0                                   # at least one target ...
    if k > 1:
        ts[-1] = 1; es[-1] = 0                          # ... and one impostor
    tgt = es[:, None] == ts[None, :]
    if k == 1:
        tgt[0, -1] = False; ts = ts.copy(); ts[-1] = 7  # a speaker nobody enrolled
        tgt = es[:, None] == ts[None, :]
    Sh = (rng.standard_normal((m, ld)) * 8).astype(np.float32)
    Sh[:, :nt] += 12.0 * tgt
    if m * nt < 1_000_000:
        Sh = np.round(Sh * 4) / 4
    S = torch.from_numpy(Sh).to(dev)
    des, dts = torch.from_numpy(es).to(dev), torch.from_numpy(ts).to(dev)
    out = eer.eer_from_matrix_dev(eng, S.data_ptr(), ld, m, nt, des.data_ptr(), dts.data_ptr())
    sub = Sh[:, :nt]
    ref = onp.eer(sub[~tgt], sub[tgt])
    assert out[4] == tgt.sum() and out[5] == (~tgt).sum()
    assert tuple(out[1:4]) == ref[1:] and out[0] == pytest.approx(ref[0], rel=1e-12)


def _eer_rank(rank, world, port, q):
    import os
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from plda_amd import MPlda, eer
    from plda_amd.sharding import eer_sharded, init_comm, shard_rows
    dev = torch.device("cuda", 0)                      # both ranks share the one GPU of the test box
    rng = np.random.default_rng(17)                    # same data on every rank
    m, nt = 301, 997
    es, ts = rng.integers(0, 9, m), rng.integers(0, 9, nt)
    sc = np.round(rng.standard_normal((m, nt)) + 1.2 * (es[:, None] == ts[None, :]), 2).astype(np.float32)
    # world 3: two real slabs and an EMPTY one, which must still take part in the reductions
    spans = [shard_rows(m, 2, 0), shard_rows(m, 2, 1), (m, m)] if world == 3 else [shard_rows(m, world, r) for r in range(world)]
    a, b = spans[rank]
    eng = MPlda(0)
    init_comm(eng, transport="host")                   # the library's collectives over this gloo group (RCCL needs one GPU per rank)
    S = torch.from_numpy(sc[a:b].copy()).to(dev)
    e_l = torch.from_numpy(es[a:b].copy()).to(dev)
    t_all = torch.from_numpy(ts).to(dev)
    out = eer_sharded(eng, S, e_l, t_all)
    full = torch.from_numpy(sc).to(dev)
    e_all = torch.from_numpy(es).to(dev)
    ref = eer.eer_from_matrix_dev(MPlda(0), full.data_ptr(), nt, m, nt, e_all.data_ptr(), t_all.data_ptr())
    q.put((rank, bool(np.array_equal(out, ref)), out.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_eer_of_a_row_sharded_matrix(world):
    """EER over row slabs held by different ranks (plda_eer_matrix_comm_dev; the handle's collectives travel over a
    gloo group, all ranks on this box's GPU): counts are reduced between the histogram passes, nothing is gathered; the result equals the one-GPU EER of
    the assembled matrix on every rank, also when a rank owns no row."""
    import multiprocessing as mp
    import os
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_eer_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True] * world, res
    assert all(r[2] == res[0][2] for r in res)


def _eer_both(monkeypatch, S, ld, m, nt, des, dts, force=False):
    """(single-pass result, spans of the call, three-pass result) on the same matrix."""
    from plda_amd import MPlda, eer
    monkeypatch.setenv("PLDA_EER_VARIANT", "2" if force else "0")
    fast = MPlda(0)
    monkeypatch.setenv("PLDA_EER_VARIANT", "1")
    slow = MPlda(0)
    monkeypatch.delenv("PLDA_EER_VARIANT")
    fast.trace_enable(True)
    a = eer.eer_from_matrix_dev(fast, S.data_ptr(), ld, m, nt, des.data_ptr(), dts.data_ptr())
    names = [sp["name"] for sp in fast.trace_read()]
    b = eer.eer_from_matrix_dev(slow, S.data_ptr(), ld, m, nt, des.data_ptr(), dts.data_ptr())
    return a, names, b


def test_single_pass_eer_is_the_three_pass_eer(monkeypatch):
    """Round 5: large matrices take ONE full pass (pilot on every 32nd row -> key window -> counts below + the window's
    scores -> exact refinement on the lists).  20 000 x 20 000 Gaussian scores with 200 speakers (2e6 targets): the
    single pass must run (trace) and give the three-pass answer bit for bit -- threshold, FAR, FRR, EER, counts."""
    import torch
    dev = torch.device("cuda", 0)
    m = nt = 20000
    g = torch.Generator(device=dev); g.manual_seed(3)
    es = torch.randint(0, 200, (m,), device=dev, generator=g)
    ts = torch.randint(0, 200, (nt,), device=dev, generator=g)
    S = torch.randn((m, nt), dtype=torch.float32, device=dev, generator=g)
    S += 2.5 * (es[:, None] == ts[None, :]).float()
    torch.cuda.synchronize()          # (the engines run on their own non-blocking streams: torch's kernels must have finished writing S)
    a, names, b = _eer_both(monkeypatch, S, nt, m, nt, es, ts)
    assert "eer.window_pass" in names and "eer.three_passes" not in names, names
    assert np.array_equal(a, b), (a, b)
    assert 0.05 < a[3] < 0.2 and a[4] + a[5] == m * nt


@pytest.mark.parametrize("case", ["gauss", "ties", "coarse_ties", "separable", "few_targets", "ragged"])
def test_single_pass_eer_forced_on_small_and_awkward_inputs(monkeypatch, case):
    """PLDA_EER_VARIANT=2 sends every matrix with >= 256 rows through the single-pass form: whatever the pilot makes of
    it -- a window, or no usable window (heavy ties put the crossing key's neighbours outside any window; separable
    classes have no crossing inside the data; too few targets in the sample) -- the answer is the three-pass one, and the
    NumPy restatement's."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(sum(map(ord, case)))
    m, nt, k = (1500, 2100, 30) if case != "ragged" else (1031, 4099, 12)
    ld = nt if case != "ragged" else nt + 5
    es, ts = rng.integers(0, k, m), rng.integers(0, k, nt)
    if case == "few_targets":
        es[:] = np.arange(m) + 1000; ts[:] = np.arange(nt) + 5000; es[:40] = 7; ts[:25] = 7
    tgt = es[:, None] == ts[None, :]
    Sh = rng.standard_normal((m, ld)).astype(np.float32)
    Sh[:, :nt] += (8.0 if case == "separable" else 1.5) * tgt
    if case == "separable":
        Sh[:, :nt] = np.clip(Sh[:, :nt], -3, 3) + 8.0 * tgt
    if case == "ties":
        Sh = np.round(Sh, 2)
    if case == "coarse_ties":
        Sh = np.round(Sh * 2) / 2
    S = torch.from_numpy(Sh).to(dev)
    des, dts = torch.from_numpy(es).to(dev), torch.from_numpy(ts).to(dev)
    a, names, b = _eer_both(monkeypatch, S, ld, m, nt, des, dts, force=True)
    assert "eer.pilot" in names, names
    assert np.array_equal(a, b), (case, a, b)
    sub = Sh[:, :nt]
    ref = onp.eer(sub[~tgt], sub[tgt])
    assert tuple(a[1:4]) == ref[1:] and a[0] == pytest.approx(ref[0], rel=1e-12)


@pytest.mark.parametrize("case", ["uniform_windowed", "mixed_znorm_windowed", "three_pass_slabs", "big_uniform"])
def test_eer_without_the_matrix(monkeypatch, case):
    """plda_score_eer_dev (round 5): the EER of the trials between transformed enrol / test vectors with the scores held one
    row slab at a time -- what scoring/scorePLDA.py:302-318 -> scoring/eer.py:68-76 computes from M x Nt calls of MPlda_score.
    Identical (all six numbers) to scoring the matrix and taking plda_eer_matrix_dev of it: several slabs per pass
    (PLDA_EER_SLAB_ROWS), the single-pass form (forced at small sizes, natural at 20 000 x 20 000) and the three-pass form
    that re-scores the slabs per pass, uniform and mixed enrol counts, z-normalised rows."""
    import torch
    from plda_amd import MPlda, eer
    dev = torch.device("cuda", 0)
    d, m, nt, k = (48, 3000, 2500, 40) if case != "big_uniform" else (32, 20000, 20000, 200)
    rng = np.random.default_rng(sum(map(ord, case)))
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    model = (rng.random(d), q * (1.0 + rng.random(d))[:, None], np.sort(rng.random(d) * 4.0 + 0.05)[::-1].copy())
    es, ts = rng.integers(0, k, m), rng.integers(0, k, nt)
    spk = rng.standard_normal((k, d)) * 1.2                       # speaker structure, so that the EER is not 50 %
    U = torch.from_numpy(spk[es] + rng.standard_normal((m, d))).to(dev)
    V = torch.from_numpy(spk[ts] + rng.standard_normal((nt, d))).to(dev)
    des, dts = torch.from_numpy(es).to(dev), torch.from_numpy(ts).to(dev)
    mixed = case == "mixed_znorm_windowed"
    dn = torch.from_numpy(rng.integers(1, 6, m).astype(np.int32)).to(dev) if mixed else None
    zm = torch.from_numpy(rng.standard_normal(m) * 3.0).to(dev) if mixed else None
    zs = torch.from_numpy(rng.random(m) * 2.0 + 0.5).to(dev) if mixed else None
    monkeypatch.setenv("PLDA_EER_VARIANT", "1" if case == "three_pass_slabs" else ("0" if case == "big_uniform" else "2"))
    if case != "big_uniform":
        monkeypatch.setenv("PLDA_EER_SLAB_ROWS", "512")           # 6 slabs
    eng = MPlda(0)
    eng.set_model(*model)
    monkeypatch.setenv("PLDA_EER_VARIANT", "1")
    ref_eng = MPlda(0)
    ref_eng.set_model(*model)
    S = torch.empty((m, nt), dtype=torch.float32, device=dev)
    ptr = lambda t: t.data_ptr() if t is not None else None      # noqa: E731
    ref_eng.score_matrix_dev(U.data_ptr(), ptr(dn), 0 if mixed else 2, m, V.data_ptr(), nt, S.data_ptr(), nt, ptr(zm), ptr(zs))
    ref_eng.synchronize()
    ref = eer.eer_from_matrix_dev(ref_eng, S.data_ptr(), nt, m, nt, des.data_ptr(), dts.data_ptr())
    eng.trace_enable(True)
    got = eer.eer_from_operands_dev(eng, U.data_ptr(), ptr(dn), 0 if mixed else 2, m, V.data_ptr(), nt, des.data_ptr(), dts.data_ptr(),
                                    ptr(zm), ptr(zs))
    names = [sp["name"] for sp in eng.trace_read()]
    assert np.array_equal(got, ref), (case, got, ref)
    assert 0.0 < got[3] < 0.45 and got[4] + got[5] == m * nt
    if case == "three_pass_slabs":
        assert any(n.startswith("eer.three_passes") for n in names), names
    else:
        assert "eer.pilot" in names, names
    if case == "big_uniform":
        assert "eer.window_pass" in names and not any(n.startswith("eer.three_passes") for n in names), names


@pytest.mark.parametrize("n_points", [2, 100, 2047])
@pytest.mark.parametrize("case", ["gauss", "ties", "one_value"])
def test_det_points_match_restatement(eng, case, n_points):
    """The points of the DET curve scoring/eer.py:34-62 plots (bob.measure.plot.det(neg, pos, 100): farfrr at thresholds spread
    evenly over the score range, accumulated in float64): thresholds and both rates EQUAL to the NumPy restatement
    (oracle/plda_oracle_np.py:det) -- two-list form and labelled-matrix form (padded leading dimension)."""
    import torch
    from plda_amd import eer
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(sum(map(ord, case)) + n_points)
    m, nt, ld, k = 211, 333, 340, 7
    es, ts = rng.integers(0, k, m), rng.integers(0, k, nt)
    tgt = es[:, None] == ts[None, :]
    Sh = rng.standard_normal((m, ld)).astype(np.float32)
    Sh[:, :nt] += 2.0 * tgt
    if case == "ties":
        Sh = np.round(Sh * 2) / 2
    if case == "one_value":
        Sh[:] = 0.25
    sub = Sh[:, :nt]
    thr_r, far_r, frr_r = onp.det(sub[~tgt], sub[tgt], n_points)
    thr, far, frr = eer.det_from_lists(eng, sub[tgt], sub[~tgt], n_points)
    assert np.array_equal(thr, thr_r) and np.array_equal(far, far_r) and np.array_equal(frr, frr_r)
    S = torch.from_numpy(Sh).to(dev)
    des, dts = torch.from_numpy(es).to(dev), torch.from_numpy(ts).to(dev)
    thr, far, frr = eer.det_from_matrix_dev(eng, S.data_ptr(), ld, m, nt, des.data_ptr(), dts.data_ptr(), n_points)
    assert np.array_equal(thr, thr_r) and np.array_equal(far, far_r) and np.array_equal(frr, frr_r)
    assert far[0] == 1.0 and frr[0] == 0.0 and (np.diff(far) <= 0).all() and (np.diff(frr) >= 0).all()
    dev_far, dev_frr = eer.ppndf(far), eer.ppndf(frr)
    assert np.isfinite(dev_far).all() and np.isfinite(dev_frr).all()
    assert np.array_equal(dev_far, onp.ppndf(far_r))
