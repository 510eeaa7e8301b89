"""scripts/gpu_configs.py -- runs the single-GPU part of BASELINE.json configs C3, C4, C5
at full size (synthetic, seeded), checks size-independent properties against the fp64
trial-list kernel / invariants, and prints timings as JSON.  Parity-test cases, not bench
lines (bench.py measures C2)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plda_amd import MPlda

dev = torch.device("cuda", 0)
res = {}


def spot(eng, dU, counts, dT, out, rng, P=2048):
    """random trials of a device score matrix vs the fp64 trial-list kernel"""
    M, Nt = out.shape
    e = rng.integers(0, M, P); t = rng.integers(0, Nt, P)
    ue, ui = np.unique(e, return_inverse=True)
    ut, ti = np.unique(t, return_inverse=True)
    Uh = dU[torch.from_numpy(ue).to(dev)].cpu().numpy()
    Th = dT[torch.from_numpy(ut).to(dev)].cpu().numpy()
    ch = counts[ue] if isinstance(counts, np.ndarray) else np.full(len(ue), counts, np.int32)
    ref = eng.score_trials((ch.astype(np.int32), Uh), (1, Th), ui, ti, znorm=False)
    got = out[torch.from_numpy(e).to(dev), torch.from_numpy(t).to(dev)].cpu().numpy().astype(np.float64)
    tol = 1e-4 * np.maximum(np.abs(ref), np.abs(ref).mean())
    return bool((np.abs(got - ref) <= tol).all()), float(np.abs(got - ref).max()), float(np.abs(ref).mean())


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def c3():
    """1M x-vectors, D=512, 10k speakers; fit; 10k speaker models (n=100) x 1M tests."""
    N, D, K = 1_000_000, 512, 10_000
    rng = np.random.default_rng(3)
    eng = MPlda(0); eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    X = torch.from_numpy(rng.random((N, D))).to(dev)
    y = torch.from_numpy((np.arange(N) % K).astype(np.int64)).to(dev)
    eng.fit_dev(X.data_ptr(), N, D, y.data_ptr(), K, 10)
    t0 = time.perf_counter(); eng.fit_dev(X.data_ptr(), N, D, y.data_ptr(), K, 10); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ft = eng.fit_timings()
    m = eng.get_model(); it = eng.fit_internals()
    T, psi = m["transform"], m["psi"]
    inv1 = float(np.abs(T @ it["W"] @ T.T - np.eye(D)).max())
    inv2 = float(np.abs(T @ it["B"] @ T.T - np.diag(psi)).max())
    out = {"fit_wall_s": wall, "stats_ms": ft["stats_ms"], "em_ms_per_iter": ft["em_ms"] / 10, "output_ms": ft["output_ms"],
           "TWT_minus_I": inv1, "TBT_minus_psi": inv2, "psi_desc_nonneg": bool((np.diff(psi) <= 0).all() and (psi >= 0).all())}
    # enrol = speaker means (n = 100), test = all 1M vectors
    means = torch.from_numpy(it["means"]).to(dev)
    dU = torch.empty((K, D), dtype=torch.float64, device=dev)
    dT = torch.empty((N, D), dtype=torch.float64, device=dev)
    eng.transform_rows_dev(means.data_ptr(), K, D, None, 100, dU.data_ptr())
    eng.transform_rows_dev(X.data_ptr(), N, D, None, 1, dT.data_ptr())
    del X
    S = torch.empty((K, N), dtype=torch.float32, device=dev)
    dt = timed(lambda: eng.score_matrix_dev(dU.data_ptr(), None, 100, K, dT.data_ptr(), N, S.data_ptr(), N))
    ok, err, scale = spot(eng, dU, 100, dT, S, rng)
    out.update({"score_ms": dt * 1e3, "trials_per_s": K * N / dt, "tflops": 2 * D * K * N / dt / 1e12,
                "spot_ok": ok, "spot_max_err": err, "mean_abs_score": scale})
    # targetdim = 200 (build extension)
    eng.truncate(200)
    dU2 = torch.empty((K, 200), dtype=torch.float64, device=dev)
    dT2 = torch.empty((N, 200), dtype=torch.float64, device=dev)
    Xm = torch.from_numpy(np.random.default_rng(3).random((N, D))).to(dev)
    eng.transform_rows_dev(means.data_ptr(), K, D, None, 100, dU2.data_ptr())
    eng.transform_rows_dev(Xm.data_ptr(), N, D, None, 1, dT2.data_ptr())
    del Xm
    dt2 = timed(lambda: eng.score_matrix_dev(dU2.data_ptr(), None, 100, K, dT2.data_ptr(), N, S.data_ptr(), N))
    ok2, err2, _ = spot(eng, dU2, 100, dT2, S, rng)
    out.update({"targetdim200_score_ms": dt2 * 1e3, "targetdim200_trials_per_s": K * N / dt2, "targetdim200_spot_ok": ok2})
    return out


def c4_shard():
    """one rank's slab of C4: 5000 (= 40k / 8) enrol models with n in 1..5 x 1.2M tests, D=256 (GEMM depth 512)."""
    D, M, Nt = 256, 5000, 1_200_000
    rng = np.random.default_rng(4)
    q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    eng = MPlda(0); eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    eng.set_model(rng.random(D), q * (1 + rng.random(D))[:, None], np.sort(rng.random(D) * 5)[::-1].copy())
    counts = rng.integers(1, 6, M).astype(np.int32)
    dn = torch.from_numpy(counts).to(dev)
    E = torch.from_numpy(rng.random((M, D))).to(dev); V = torch.from_numpy(rng.random((Nt, D))).to(dev)
    dU = torch.empty((M, D), dtype=torch.float64, device=dev); dT = torch.empty((Nt, D), dtype=torch.float64, device=dev)
    eng.transform_rows_dev(E.data_ptr(), M, D, dn.data_ptr(), 0, dU.data_ptr())
    eng.transform_rows_dev(V.data_ptr(), Nt, D, None, 1, dT.data_ptr())
    S = torch.empty((M, Nt), dtype=torch.float32, device=dev)
    dt = timed(lambda: eng.score_matrix_dev(dU.data_ptr(), dn.data_ptr(), 0, M, dT.data_ptr(), Nt, S.data_ptr(), Nt))
    ok, err, scale = spot(eng, dU, counts, dT, S, rng)
    return {"score_ms": dt * 1e3, "trials_per_s": M * Nt / dt, "gemm_depth": 2 * D, "tflops": 4 * D * M * Nt / dt / 1e12,
            "spot_ok": ok, "spot_max_err": err, "mean_abs_score": scale}


def c5():
    """z-norm: 50k models (n=1) vs 200k cohort, fused (1e10 LLRs, no matrix), then 50k x 50k z-normed trials."""
    D, M, Nb = 200, 50_000, 200_000
    rng = np.random.default_rng(5)
    q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    eng = MPlda(0); eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    eng.set_model(rng.random(D), q * (1 + rng.random(D))[:, None], np.sort(rng.random(D) * 5)[::-1].copy())
    E = torch.from_numpy(rng.random((M, D))).to(dev)
    dU = torch.empty((M, D), dtype=torch.float64, device=dev)
    eng.transform_rows_dev(E.data_ptr(), M, D, None, 1, dU.data_ptr())
    bkg = torch.from_numpy(rng.random((Nb, D))).to(dev)
    zm = torch.empty(M, dtype=torch.float64, device=dev); zs = torch.empty(M, dtype=torch.float64, device=dev)
    dt = timed(lambda: eng.znorm_stats_dev(bkg.data_ptr(), Nb, 0, D, dU.data_ptr(), M, zm.data_ptr(), zs.data_ptr()), reps=2)
    # reference for 3 models over the full cohort with the fp64 trial-list kernel (roles as in MPlda_norm)
    cohort_t = torch.empty((Nb, D), dtype=torch.float64, device=dev)
    eng.transform_rows_dev(bkg.data_ptr(), Nb, D, None, Nb, cohort_t.data_ptr())
    Ch = cohort_t.cpu().numpy()
    sel = np.array([0, M // 2, M - 1])
    Mh = dU[torch.from_numpy(sel).to(dev)].cpu().numpy()
    okz, worst = True, 0.0
    for k, j in enumerate(sel):
        s = eng.score_trials((np.ones(Nb, np.int32), Ch), (1, Mh[k:k + 1]), np.arange(Nb), np.zeros(Nb, np.int64), znorm=False)
        rm, rs = s.mean(), s.std()
        gm, gs = float(zm[j]), float(zs[j])
        e1 = abs(gm - rm) / max(abs(rm), np.abs(s).mean()); e2 = abs(gs - rs) / rs
        worst = max(worst, e1, e2)
        okz = okz and e1 < 1e-4 and e2 < 1e-4
    out = {"znorm_ms": dt * 1e3, "llr_per_s": M * Nb / dt, "znorm_ok": bool(okz), "znorm_worst_rel_err": float(worst)}
    S = torch.empty((M, M), dtype=torch.float32, device=dev)
    dt2 = timed(lambda: eng.score_matrix_dev(dU.data_ptr(), None, 1, M, dU.data_ptr(), M, S.data_ptr(), M, zm.data_ptr(), zs.data_ptr()))
    out.update({"znormed_50kx50k_ms": dt2 * 1e3, "znormed_trials_per_s": M * M / dt2, "finite": bool(torch.isfinite(S[::499]).all())})
    return out


def c2_skew():
    """C2 with skewed speaker sizes (n_k in 5..60): fit on the GPU, invariants + oracle-free checks."""
    N, D, K = 100_000, 200, 5000
    rng = np.random.default_rng(22)
    sizes = rng.integers(5, 61, K)
    sizes = np.floor(sizes * (N / sizes.sum())).astype(np.int64); sizes[-1] += N - sizes.sum()
    y = np.repeat(np.arange(K), sizes); rng.shuffle(y)
    eng = MPlda(0); eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    X = torch.from_numpy(rng.random((N, D)) + 0.3 * rng.standard_normal((K, D))[y]).to(dev)
    yy = torch.from_numpy(y.astype(np.int64)).to(dev)
    eng.fit_dev(X.data_ptr(), N, D, yy.data_ptr(), K, 10)
    t0 = time.perf_counter(); eng.fit_dev(X.data_ptr(), N, D, yy.data_ptr(), K, 10); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ft = eng.fit_timings(); m = eng.get_model(); it = eng.fit_internals()
    T, psi = m["transform"], m["psi"]
    Xh = X.cpu().numpy()
    means_ref = np.zeros((K, D)); np.add.at(means_ref, y, Xh); means_ref /= sizes[:, None]
    return {"fit_wall_s": wall, "stats_ms": ft["stats_ms"], "em_ms_per_iter": ft["em_ms"] / 10, "distinct_n": int(len(np.unique(sizes))),
            "counts_ok": bool((it["counts"] == sizes).all()), "means_rel_err": float(np.abs(it["means"] - means_ref).max() / np.abs(means_ref).max()),
            "TWT_minus_I": float(np.abs(T @ it["W"] @ T.T - np.eye(D)).max()),
            "TBT_minus_psi": float(np.abs(T @ it["B"] @ T.T - np.diag(psi)).max()), "psi_max": float(psi[0])}


def pcie():
    """host-pointer plda_score_matrix (pageable numpy buffers): the PCIe-inclusive rate, never the bench value."""
    D, M = 200, 20_000
    rng = np.random.default_rng(9)
    q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    eng = MPlda(0)
    eng.set_model(rng.random(D), q, np.sort(rng.random(D) * 4)[::-1].copy())
    U = eng.transform_array(rng.random((M, D)), 1)
    eng.score_matrix((1, U[:512]), (1, U[:512]))
    t0 = time.perf_counter(); S = eng.score_matrix((1, U), (1, U)); dt = time.perf_counter() - t0
    return {"trials": M * M, "seconds": dt, "trials_per_s_incl_pcie": M * M / dt, "out_GB": M * M * 4 / 1e9}


def eer_full():
    """EER of a 100k x 100k trials matrix (5000 speakers, 20 utts each): 3 histogram passes over 40 GB."""
    from plda_amd import eer
    D, N, K = 200, 100_000, 5000
    rng = np.random.default_rng(8)
    eng = MPlda(0); eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    y = (np.arange(N) % K)
    X = rng.random((N, D)) + 0.25 * rng.standard_normal((K, D))[y]
    dX = torch.from_numpy(X).to(dev); dy = torch.from_numpy(y.astype(np.int64)).to(dev)
    eng.fit_dev(dX.data_ptr(), N, D, dy.data_ptr(), K, 5)
    U = torch.empty((N, D), dtype=torch.float64, device=dev)
    eng.transform_rows_dev(dX.data_ptr(), N, D, None, 1, U.data_ptr())
    S = torch.empty((N, N), dtype=torch.float32, device=dev)
    eng.score_matrix_dev(U.data_ptr(), None, 1, N, U.data_ptr(), N, S.data_ptr(), N)
    torch.cuda.synchronize()
    out = None
    def run():
        nonlocal out
        out = eer.eer_from_matrix_dev(eng, S.data_ptr(), N, N, N, dy.data_ptr(), dy.data_ptr())
    dt = timed(run, reps=2)
    # cross-check on a 6000 x 6000 corner with the same device code vs exact numpy counts at the returned threshold
    sub = S[:6000, :6000].cpu().numpy(); tgt = y[:6000, None] == y[None, :6000]
    o2 = eer.eer_from_matrix_dev(eng, S.data_ptr(), N, 6000, 6000, dy.data_ptr(), dy.data_ptr())
    far = float((sub[~tgt].astype(np.float64) >= o2[0]).mean()); frr = float((sub[tgt].astype(np.float64) < o2[0]).mean())
    return {"eer_ms": dt * 1e3, "GBps": 3 * N * N * 4 / dt / 1e9, "eer": float(out[3]), "far": float(out[1]), "frr": float(out[2]),
            "threshold": float(out[0]), "targets": float(out[4]), "impostors": float(out[5]),
            "corner_farfrr_consistent": bool(far == o2[1] and frr == o2[2])}


def frontend():
    """d-vector pooling: 200k utterances x 100 frames x 64 dims, float32 -> HBM GB/s."""
    import ctypes as C
    U, F, D = 200_000, 100, 64
    eng = MPlda(0); eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    frames = torch.randn((U * F, D), dtype=torch.float32, device=dev)
    off = torch.arange(0, (U + 1) * F, F, dtype=torch.int64, device=dev)
    out = torch.empty((U, D), dtype=torch.float64, device=dev)
    def run(method):
        rc = eng._lib.plda_dvector_pool_dev(eng._h, C.c_void_p(frames.data_ptr()), 0, U * F, D, C.c_void_p(off.data_ptr()), U,
                                            method, 1, C.c_void_p(out.data_ptr()))
        assert rc == 0
    res = {}
    for name, m in (("mean", 0), ("max", 1), ("var", 2)):
        dt = timed(lambda: run(m), reps=5)
        res[name + "_ms"] = dt * 1e3
        res[name + "_GBps"] = frames.numel() * 4 / dt / 1e9
    ref = torch.nn.functional.normalize(frames[:F * 3].double().view(3, F, D), dim=2).mean(1)
    run(0); torch.cuda.synchronize()
    res["check_max_err"] = float((out[:3] - ref).abs().max())
    res["bytes"] = frames.numel() * 4
    return res


def lda_c2():
    """LDA row at the C2 shape (scoring/scoreLDA.py's workload): svd fit on 100k x 200 / 5k speakers, then
    predict_log_proba of 100k test d-vectors against the 5k classes (5e8 scores)."""
    from plda_amd.lda import LDA
    N, D, K, Nt = 100_000, 200, 5000, 100_000
    rng = np.random.default_rng(2)
    y = np.arange(N) % K
    X = torch.from_numpy(rng.random((N, D)) + 0.8 * rng.standard_normal((K, D))[y]).to(dev)
    yy = torch.from_numpy(y.astype(np.int64)).to(dev)
    Xt = torch.from_numpy(rng.random((Nt, D))).to(dev)
    out = torch.empty((Nt, K), dtype=torch.float64, device=dev)
    res = {}
    for solver in ("svd", "eigen", "lsqr"):
        lda = LDA(solver)
        lda.fit_dev(X.data_ptr(), N, D, yy.data_ptr(), K)
        res["fit_%s_ms" % solver] = 1e3 * timed(lambda: lda.fit_dev(X.data_ptr(), N, D, yy.data_ptr(), K), 2)
    lda = LDA("svd"); lda.fit_dev(X.data_ptr(), N, D, yy.data_ptr(), K)
    for mode, name in ((0, "decision"), (1, "log_proba")):
        ms = 1e3 * timed(lambda: lda.predict_dev(Xt.data_ptr(), Nt, mode, out.data_ptr()), 3)
        res["%s_ms" % name] = ms
        res["%s_scores_per_s" % name] = Nt * K / (ms / 1e3)
    res["log_proba_fp64_tflops"] = 2.0 * D * Nt * K / (res["log_proba_ms"] / 1e3) / 1e12
    res["row_logsumexp_max_abs"] = float(torch.logsumexp(out[:2048], dim=1).abs().max())
    ref = torch.log_softmax(Xt[:2048] @ torch.from_numpy(lda._coef).to(dev).T + torch.from_numpy(lda._intercept).to(dev), dim=1)
    res["check_max_err"] = float((out[:2048] - ref).abs().max())
    return res


def htk_decode():
    """HTK reader row: 20k files, 5M frames, decoded from device-resident bytes (algorithmic bytes = read the
    file bodies once + write the float32 output once)."""
    from plda_amd import MPlda as _M
    eng = _M(0)
    res = {}
    for dim, F in ((40, 0), (40, 5), (256, 0), (257, 1)):
        U = 20000
        g = torch.Generator(device="cpu").manual_seed(5)
        counts = torch.randint(100, 400, (U,), generator=g)
        off = torch.zeros(U + 1, dtype=torch.int64); off[1:] = torch.cumsum(counts, 0)
        T = int(off[-1])
        blob = torch.randint(0, 2 ** 31 - 1, (T * dim,), dtype=torch.int32, device=dev)
        file_off = (off[:-1] * dim).to(dev); doff = off.to(dev)
        out = torch.empty((T, (2 * F + 1) * dim), dtype=torch.float32, device=dev)
        run = lambda: eng._ck(eng._lib.plda_htk_frames_dev(eng._h, blob.data_ptr(), file_off.data_ptr(), doff.data_ptr(), U, T,
                                                           dim * 4, F, out.data_ptr()))
        s = timed(run, 5)
        byts = T * dim * 4 * (1 + (2 * F + 1))
        res["dim%d_F%d" % (dim, F)] = {"ms": 1e3 * s, "GBps": byts / s / 1e9, "frames": T}
        del blob, out
        torch.cuda.empty_cache()
    return res


def vendor_sgemm():
    """Yardstick only (not part of the product): the vendor fp32 GEMM (torch.mm -> hipBLASLt / rocBLAS) on the
    C2 trials shape, 100k x 200 @ 200 x 100k -> 100k x 100k fp32, without the bias terms the trials kernel fuses."""
    M = Nt = 100_000
    res = {}
    for D in (200, 150, 512):
        m = M if D != 512 else 10_000
        n = Nt if D != 512 else 1_000_000
        a = torch.randn((m, D), dtype=torch.float32, device=dev)
        b = torch.randn((n, D), dtype=torch.float32, device=dev)
        out = torch.empty((m, n), dtype=torch.float32, device=dev)
        s = timed(lambda: torch.mm(a, b.T, out=out), 3)
        res["D%d_ms" % D] = 1e3 * s
        res["D%d_tflops" % D] = 2.0 * D * m * n / s / 1e12
        del a, b, out
        torch.cuda.empty_cache()
    return res


for name, fn in (("C3", c3), ("lda", lda_c2), ("vendor_sgemm", vendor_sgemm), ("htk", htk_decode), ("C4_shard", c4_shard), ("C5", c5), ("frontend", frontend), ("C2_skew", c2_skew), ("pcie", pcie), ("eer", eer_full)):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    t0 = time.perf_counter()
    try:
        res[name] = fn()
    except Exception as ex:  # report, keep going
        res[name] = {"error": repr(ex)}
    res[name]["wall_s"] = time.perf_counter() - t0
    torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
