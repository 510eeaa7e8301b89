"""GPU, BASELINE.json full size (C2: 100k x 100k trials, D = 200): size-independent
properties of the trials matrix, since the oracle cannot finish 1e10 trials.

  * random trials agree with the fp64 trial-list kernel (itself oracle-checked at small
    size in test_gpu_scoring.py) within the 1e-4 tolerance;
  * tiling invariance: any sub-block scored on its own is BIT-IDENTICAL to the same
    block of the full matrix (fixed k-order fp32 FMA chains, position independent);
  * with n = 1 the LLR is symmetric in (enrol, test): S[i, j] == S[j, i] to fp32 rounding.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_c2_full_size_properties():
    import torch
    from plda_amd import MPlda
    dev = torch.device("cuda", 0)
    D, N = 200, 100000
    rng = np.random.default_rng(2)
    # a realistic model without a 100k fit: random orthogonal-ish transform, decaying psi
    q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    T = q * (1.0 + rng.random(D))[:, None]
    psi = np.sort(rng.random(D) * 4.0)[::-1].copy()
    eng = MPlda(0)
    eng.set_model(rng.random(D), T, psi)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    X = torch.from_numpy(rng.random((N, D))).to(dev)
    Ut = torch.empty((N, D), dtype=torch.float64, device=dev)
    eng.transform_rows_dev(X.data_ptr(), N, D, None, 1, Ut.data_ptr())
    out = torch.empty((N, N), dtype=torch.float32, device=dev)
    eng.score_matrix_dev(Ut.data_ptr(), None, 1, N, Ut.data_ptr(), N, out.data_ptr(), N)
    torch.cuda.synchronize()
    assert torch.isfinite(out[::997]).all()

    # (1) 4096 random trials vs the fp64 trial-list kernel
    P = 4096
    e = rng.integers(0, N, P); t = rng.integers(0, N, P)
    rows = np.unique(np.concatenate([e, t]))
    remap = {int(r): i for i, r in enumerate(rows)}
    Uh = Ut[torch.from_numpy(rows).to(dev)].cpu().numpy()
    ref = eng.score_trials((np.ones(len(rows), np.int32), Uh), (1, Uh),
                           np.array([remap[int(i)] for i in e]), np.array([remap[int(i)] for i in t]))
    got = out[torch.from_numpy(e).to(dev), torch.from_numpy(t).to(dev)].cpu().numpy().astype(np.float64)
    tol = 1e-4 * np.maximum(np.abs(ref), np.abs(ref).mean())
    assert (np.abs(got - ref) <= tol).all(), np.abs(got - ref).max()
    # ... and a 96 x 160 block (first, middle and last rows / columns) against the NumPy fp64 oracle itself
    from oracle import plda_oracle_np as onp
    er = np.r_[0:32, 50000:50032, N - 32:N]; tc = np.r_[0:64, 33333:33365, N - 64:N]
    oref = onp.llr_matrix(psi, Ut[torch.from_numpy(er).to(dev)].cpu().numpy(), 1, Ut[torch.from_numpy(tc).to(dev)].cpu().numpy())
    gblk = out[torch.from_numpy(er).to(dev)][:, torch.from_numpy(tc).to(dev)].cpu().numpy().astype(np.float64)
    assert (np.abs(gblk - oref) <= 1e-4 * np.maximum(np.abs(oref), np.abs(oref).mean())).all(), np.abs(gblk - oref).max()

    # (2) tiling invariance, bit-exact, at unaligned offsets
    for (r0, r1, c0, c1) in [(0, 300, 0, 500), (12345, 12345 + 777, 54321, 54321 + 1111), (N - 129, N, N - 257, N)]:
        blk = torch.empty((r1 - r0, c1 - c0), dtype=torch.float32, device=dev)
        eng.score_matrix_dev(Ut[r0:r1].data_ptr(), None, 1, r1 - r0, Ut[c0:c1].data_ptr(), c1 - c0,
                             blk.data_ptr(), c1 - c0)
        torch.cuda.synchronize()
        assert torch.equal(blk, out[r0:r1, c0:c1])

    # (3) symmetry for n = 1
    a = out[:4096, 50000:54096]
    b = out[50000:54096, :4096].T
    scale = float(a.abs().mean())
    assert float((a - b).abs().max()) <= 2e-5 * max(scale, 1.0)


def _model(d, seed):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    return rng.random(d), q * (1.0 + rng.random(d))[:, None], np.sort(rng.random(d) * 4.0 + 0.05)[::-1].copy()


def test_c5_full_size_fused_znorm_statistics():
    """BASELINE C5: 50 000 models x 200 000 cohort vectors (1e10 LLRs, nothing materialised) through the fused
    z-norm epilogue (MPlda_norm, pldamodule.cpp:196-256), then the 50k x 50k z-normalised trials matrix.
    Checked against the fp64 GEMM-form oracle on sampled models: mean / population std over the WHOLE cohort
    within 1e-10 (fp64 moments), and z-normalised trials within the score tolerance (1e-4: fp32 GEMM)."""
    import torch
    from plda_amd import MPlda
    from oracle import plda_oracle_np as onp
    dev = torch.device("cuda", 0)
    D, M, Nb = 200, 50000, 200000
    mean, T, psi = _model(D, 51)
    eng = MPlda(0)
    eng.set_model(mean, T, psi)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    g = torch.Generator(device=dev); g.manual_seed(9)
    bkg = torch.rand((Nb, D), dtype=torch.float64, device=dev, generator=g)
    models = torch.randn((M, D), dtype=torch.float64, device=dev, generator=g)
    zm = torch.empty(M, dtype=torch.float64, device=dev); zs = torch.empty(M, dtype=torch.float64, device=dev)
    eng.znorm_stats_dev(bkg.data_ptr(), Nb, Nb, D, models.data_ptr(), M, zm.data_ptr(), zs.data_ptr())
    torch.cuda.synchronize()
    sel = np.array([0, 1, 777, 25000, 49999])
    model = dict(mean=mean, transform=T, psi=psi, offset=-T @ mean)
    rm, rs = onp.norm(model, bkg.cpu().numpy(), models[torch.from_numpy(sel).to(dev)].cpu().numpy())
    gm, gs = zm[torch.from_numpy(sel).to(dev)].cpu().numpy(), zs[torch.from_numpy(sel).to(dev)].cpu().numpy()
    # (statistics by moments, fp64 throughout: 1e-10; the 1e-4 of north_star belongs to the fp32 trials below)
    assert (np.abs(gm - rm) <= 1e-10 * np.maximum(np.abs(rm), np.abs(rm).mean())).all(), np.abs(gm - rm).max()
    assert (np.abs(gs - rs) <= 1e-10 * rs).all(), (np.abs(gs - rs) / rs).max()
    # 50k x 50k z-normalised trials (default dispatch: the 256 x 256 kernel with the map folded into the operands)
    tests = torch.randn((M, D), dtype=torch.float64, device=dev, generator=g)
    out = torch.empty((M, M), dtype=torch.float32, device=dev)
    eng.score_matrix_dev(models.data_ptr(), None, 1, M, tests.data_ptr(), M, out.data_ptr(), M, zm.data_ptr(), zs.data_ptr())
    torch.cuda.synchronize()
    cols = np.random.default_rng(3).integers(0, M, 2048)
    tc = torch.from_numpy(cols).to(dev); ts = torch.from_numpy(sel).to(dev)
    raw = onp.llr_matrix(psi, models[ts].cpu().numpy(), 1, tests[tc].cpu().numpy())
    ref = (raw - zm[ts].cpu().numpy()[:, None]) / zs[ts].cpu().numpy()[:, None]
    got = out[ts][:, tc].cpu().numpy().astype(np.float64)
    tol = 1e-4 * np.maximum(np.abs(ref), np.abs(ref).mean())
    assert (np.abs(got - ref) <= tol).all(), (np.abs(got - ref) / tol).max()


def test_c3_and_c4_shard_full_size():
    """BASELINE C3 (10k models, n = 100, x 1M tests, D = 512) and one of C4's eight shards (5k models, n in 1..5, x 1.2M
    tests, D = 256, GEMM depth 512) at full size through default dispatch, checked on sampled rows x columns
    against the fp64 GEMM-form oracle."""
    import torch
    from plda_amd import MPlda
    from oracle import plda_oracle_np as onp
    dev = torch.device("cuda", 0)
    for (D, M, Nt, counts, seed) in [(512, 10000, 1000000, 100, 61), (256, 5000, 1200000, None, 62)]:
        mean, T, psi = _model(D, seed)
        eng = MPlda(0)
        eng.set_model(mean, T, psi)
        eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        g = torch.Generator(device=dev); g.manual_seed(seed)
        dU = torch.randn((M, D), dtype=torch.float64, device=dev, generator=g)
        dV = torch.randn((Nt, D), dtype=torch.float64, device=dev, generator=g)
        n = None
        if counts is None:
            n = torch.randint(1, 6, (M,), device=dev, dtype=torch.int32, generator=g)
        out = torch.empty((M, Nt), dtype=torch.float32, device=dev)
        eng.score_matrix_dev(dU.data_ptr(), n.data_ptr() if n is not None else None, counts or 0, M, dV.data_ptr(), Nt,
                             out.data_ptr(), Nt)
        torch.cuda.synchronize()
        rows = np.array([0, 255, 256, M // 2, M - 1])
        cols = np.unique(np.concatenate([np.random.default_rng(seed).integers(0, Nt, 3000), np.arange(Nt - 64, Nt)]))
        tr, tc = torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev)
        ref = onp.llr_matrix(psi, dU[tr].cpu().numpy(), n[tr].cpu().numpy() if n is not None else counts, dV[tc].cpu().numpy())
        got = out[tr][:, tc].cpu().numpy().astype(np.float64)
        tol = 1e-4 * np.maximum(np.abs(ref), np.abs(ref).mean())
        assert (np.abs(got - ref) <= tol).all(), (D, (np.abs(got - ref) / tol).max())
        del out, dU, dV
        torch.cuda.empty_cache()


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize("skewed", [False, True])
def test_c2_full_size_fit_matches_oracle(oracle, skewed):
    """BASELINE C2 fit at FULL size against oracle/plda_oracle.c: 100 000 x 200, 5 000 speakers -- balanced (20 per
    speaker: one EM group) and with speaker sizes in 5..60 (up to 56 distinct counts: the grouped EM's batch), 10
    iterations.  Every stage the reference's fit defines (pldamodule.cpp:76-106): counts exactly (K1a), means 1e-13
    (K1), offset scatter 1e-10 (K2), W / B / psi / T^T T 1e-8 (K3, K6, K7; observed ~1e-11)."""
    from plda_amd import MPlda
    N, D, K = 100000, 200, 5000
    rng = np.random.default_rng(2)
    x = rng.random((N, D))
    if skewed:
        c = rng.integers(5, 61, K)
        while c.sum() > N:                       # trim to exactly N rows, sizes stay in 5..60
            i = rng.integers(0, K, 4096)
            i = i[c[i] > 5][: int(c.sum() - N)]
            np.subtract.at(c, np.unique(i), 1)
        while c.sum() < N:
            i = rng.integers(0, K, 4096)
            i = i[c[i] < 60][: int(N - c.sum())]
            np.add.at(c, np.unique(i), 1)
        assert c.sum() == N and c.min() >= 5 and c.max() <= 60
        y = rng.permutation(np.repeat(np.arange(K), c)).astype(np.uint64)
    else:
        y = (np.arange(N) % K).astype(np.uint64)
    eng = MPlda(0)
    eng.fit(x, y, 10)
    it, g = eng.fit_internals(), eng.get_model()
    st = oracle.stats(x, y)
    assert np.array_equal(it["counts"], st["counts"])
    assert _rel(it["means"], st["means"]) < 1e-13
    assert _rel(it["scatter"], st["scatter"]) < 1e-10
    ref = oracle.fit(x, y, 10)
    assert _rel(it["W"], ref["W"]) < 1e-8 and _rel(it["B"], ref["B"]) < 1e-8, (_rel(it["W"], ref["W"]), _rel(it["B"], ref["B"]))
    assert np.abs(g["psi"] - ref["psi"]).max() <= 1e-8 * ref["psi"].max()
    assert _rel(g["mean"], ref["mean"]) < 1e-12
    T, Tr = g["transform"], ref["transform"]
    assert _rel(T.T @ T, Tr.T @ Tr) < 1e-8
    assert _rel(T.T @ np.diag(g["psi"]) @ T, Tr.T @ np.diag(ref["psi"]) @ Tr) < 1e-8
    assert np.abs(T @ it["W"] @ T.T - np.eye(D)).max() < 1e-9


def test_c3_full_size_statistics_pass():
    """BASELINE C3 statistics pass at FULL size: 1M x 512, 10 000 speakers (K1a label sort, K1 centroids, K2 super-tile
    SYRK with split-K) against NumPy fp64: counts exactly, means 1e-13, offset scatter
    X^T diag(1 / n_label) X - sum_k m_k m_k^T (pldamodule.cpp:94-98 with the wrapper's 1 / n_k weight) 1e-10."""
    import torch
    from plda_amd import MPlda
    dev = torch.device("cuda", 0)
    N, D, K = 1000000, 512, 10000
    g = torch.Generator(device=dev); g.manual_seed(3)
    dX = torch.rand((N, D), dtype=torch.float64, device=dev, generator=g)
    dy = torch.randint(0, K, (N,), device=dev, dtype=torch.int64, generator=g)
    dy[:K] = torch.arange(K, device=dev)                       # every speaker present: dense labels
    eng = MPlda(0)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    eng.fit_stats_dev(dX.data_ptr(), N, D, dy.data_ptr(), K)
    means = torch.empty((K, D), dtype=torch.float64, device=dev)
    counts = torch.empty((K,), dtype=torch.int64, device=dev)
    scatter = torch.empty((D, D), dtype=torch.float64, device=dev)
    eng.fit_get_stats_dev(means.data_ptr(), counts.data_ptr(), scatter.data_ptr())
    torch.cuda.synchronize()
    x, y = dX.cpu().numpy(), dy.cpu().numpy()
    del dX
    c = np.bincount(y, minlength=K)
    assert np.array_equal(counts.cpu().numpy(), c)
    order = np.argsort(y, kind="stable")
    sums = np.add.reduceat(x[order], np.r_[0, np.cumsum(c)[:-1]], axis=0)
    m = sums / c[:, None]
    assert _rel(means.cpu().numpy(), m) < 1e-13
    w = 1.0 / c[y]
    S = (x * w[:, None]).T @ x - m.T @ m
    assert _rel(scatter.cpu().numpy(), S) < 1e-10, _rel(scatter.cpu().numpy(), S)


def _speaker_sizes(rng, K, N, lo, hi, step):
    """K counts in lo..hi that are multiples of `step` (few distinct values: each one costs the per-class oracle an
    inversion) and sum to N exactly (the remainder goes to the last speakers, one utterance each)."""
    c = rng.integers(lo // step, hi // step + 1, K) * step
    diff = int(N - c.sum())
    i = 0
    while abs(diff) >= step:                       # whole steps first, spread over speakers that stay inside lo..hi
        s = step if diff > 0 else -step
        if lo <= c[i % K] + s <= hi:
            c[i % K] += s
            diff -= s
        i += 1
    c[0] += diff                                   # (|diff| < step: one more distinct value at most)
    assert c.sum() == N and c.min() >= 1
    return c


@pytest.mark.parametrize("N,D,K,seed,iters", [(1000000, 512, 10000, 31, 2), (1200000, 256, 7200, 32, 10)], ids=["C3", "C4_10_iterations"])
def test_c3_c4_full_size_fit_em_and_getoutput(oracle, N, D, K, seed, iters):
    """BASELINE C3 (1M x 512, 10 000 speakers; 2 EM iterations) and C4 (1.2M x 256, 7 200 speakers; the DEFAULT 10
    iterations of pldamodule.cpp:50, round-4 review weak 2: where a drift of the grouped EM's conditioning would show)
    fits at FULL size against the oracle (round-3 review, missing 3 / next 2): this is the part of pldamodule.cpp:100-106 that runs the
    blocked (D > 256) SPD inverse, the grouped EM with tens of distinct speaker sizes at K = 10 000 and the 4-wave
    tridiagonalisation.  Statistics against NumPy fp64 (counts exactly, means 1e-13, offset scatter 1e-10: the scalar C
    oracle would need minutes for N D^2 flop), then oracle.em_iter x iters and oracle.get_output on those statistics:
    W / B 1e-8, psi 1e-8 psi_max, T^T T and T^T Psi T 1e-8 (rotation-invariant), T W T^T = I 1e-9."""
    import torch
    from plda_amd import MPlda
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    c = _speaker_sizes(rng, K, N, N // K * 4 // 5, N // K * 6 // 5, 4)
    y = rng.permutation(np.repeat(np.arange(K), c)).astype(np.int64)
    g = torch.Generator(device=dev); g.manual_seed(seed)
    dX = torch.rand((N, D), dtype=torch.float64, device=dev, generator=g)
    # speaker structure, so that B is not numerically null: a per-speaker offset
    off = torch.randn((K, D), dtype=torch.float64, device=dev, generator=g) * 0.3
    dy = torch.from_numpy(y).to(dev)
    dX += off[dy]
    del off
    eng = MPlda(0)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    eng.fit_dev(dX.data_ptr(), N, D, dy.data_ptr(), K, iters)
    torch.cuda.synchronize()
    it, mdl = eng.fit_internals(), eng.get_model()
    x = dX.cpu().numpy()
    del dX
    torch.cuda.empty_cache()
    # ---- statistics (PldaStats::AddSamples with the wrapper's 1 / n_k weight, pldamodule.cpp:94-98) in NumPy fp64
    assert np.array_equal(it["counts"], c)
    order = np.argsort(y, kind="stable")
    sums = np.add.reduceat(x[order], np.r_[0, np.cumsum(c)[:-1]], axis=0)
    del order
    m = sums / c[:, None]
    del sums
    assert _rel(it["means"], m) < 1e-13
    x *= np.sqrt(1.0 / c[y])[:, None]
    S = x.T @ x - m.T @ m
    del x
    assert _rel(it["scatter"], S) < 1e-10, _rel(it["scatter"], S)
    st = dict(means=m, counts=c.astype(np.int64), scatter=S, sum=(m / c[:, None]).sum(0),
              class_weight=float((1.0 / c).sum()), example_weight=float(K))
    # ---- the EM iterations and GetOutput of the per-class oracle (oracle/plda_oracle.c) on those statistics
    W, B = np.eye(D), np.eye(D)
    for _ in range(iters):
        W, B = oracle.em_iter(st, W, B)
    assert _rel(it["W"], W) < 1e-8 and _rel(it["B"], B) < 1e-8, (_rel(it["W"], W), _rel(it["B"], B))
    ref = oracle.get_output(st, W, B)
    assert np.abs(mdl["psi"] - ref["psi"]).max() <= 1e-8 * ref["psi"].max(), np.abs(mdl["psi"] - ref["psi"]).max() / ref["psi"].max()
    assert _rel(mdl["mean"], ref["mean"]) < 1e-12
    T, Tr = mdl["transform"], ref["transform"]
    assert _rel(T.T @ T, Tr.T @ Tr) < 1e-8
    assert _rel(T.T @ np.diag(mdl["psi"]) @ T, Tr.T @ np.diag(ref["psi"]) @ Tr) < 1e-8
    assert np.abs(T @ it["W"] @ T.T - np.eye(D)).max() < 1e-9
