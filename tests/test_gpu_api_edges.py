"""GPU: edge cases of the C ABI and the shim -- empty / ragged inputs, strided outputs, the
host path's row slabs, numutts subsets, persistence, error codes (never an abort)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import make_data, score_tol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fitted(oracle):
    from plda_amd import MPlda
    x, y = make_data(41, 900, 20, 30, skew=True, scale_between=0.5)
    eng = MPlda(0)
    eng.fit(x, y, 4)
    return eng, oracle.fit(x, y, 4), x, y


def test_error_codes_do_not_abort(fitted):
    from plda_amd import MPlda, _native as N
    eng, ref, x, y = fitted
    lib = eng._lib
    fresh = MPlda(0)
    assert lib.plda_get_dims(fresh._h, None, None) == N.PLDA_E_NOT_FITTED
    assert "not fitted" in N.last_error(fresh._h)
    out = np.zeros((2, 2), np.float32)
    u = np.zeros((2, 20))
    # ld_out < Nt, NULL pointers, n_uniform <= 0 with no counts
    assert lib.plda_score_matrix(eng._h, u.ctypes.data, None, 1, 2, u.ctypes.data, 2, None, None, out.ctypes.data, 1) == N.PLDA_E_INVAL
    assert lib.plda_score_matrix(eng._h, None, None, 1, 2, u.ctypes.data, 2, None, None, out.ctypes.data, 2) == N.PLDA_E_INVAL
    assert lib.plda_score_matrix(eng._h, u.ctypes.data, None, 0, 2, u.ctypes.data, 2, None, None, out.ctypes.data, 2) == N.PLDA_E_INVAL
    # feature-dim mismatch in transform
    cap = C.c_int64(4)
    bad = np.zeros((4, 7))
    lab = np.arange(4, dtype=np.uint64)
    assert lib.plda_transform_groups(eng._h, bad.ctypes.data, 4, 7, lab.ctypes.data, lab.ctypes.data,
                                     np.zeros(4, np.int64).ctypes.data, np.zeros((4, 20)).ctypes.data, C.byref(cap)) == N.PLDA_E_INVAL
    # capacity too small is reported with the needed size
    cap = C.c_int64(1)
    xx = np.ascontiguousarray(x[:4])
    rc = lib.plda_transform_groups(eng._h, xx.ctypes.data, 4, 20, lab.ctypes.data, np.zeros(1, np.uint64).ctypes.data,
                                   np.zeros(1, np.int64).ctypes.data, np.zeros((1, 20)).ctypes.data, C.byref(cap))
    assert rc == N.PLDA_E_CAPACITY and cap.value == 4
    # non-dense labels at the ABI level (the shim compacts; the ABI refuses)
    xs = np.ascontiguousarray(x[:6]); gap = np.array([0, 0, 2, 2, 3, 3], np.uint64)
    assert lib.plda_fit(fresh._h, xs.ctypes.data, 6, 20, gap.ctypes.data, 2) == N.PLDA_E_LABELS
    assert lib.plda_fit(fresh._h, xs.ctypes.data, 6, 20, np.zeros(6, np.uint64).ctypes.data, 2) == N.PLDA_E_ONE_SPEAKER
    assert lib.plda_truncate(eng._h, 0) == N.PLDA_E_INVAL and lib.plda_smooth(eng._h, 1.5) == N.PLDA_E_INVAL
    # trial indexes outside the sets
    with pytest.raises(RuntimeError):
        eng.score_trials((np.ones(2, np.int32), u), (1, u), [0, 5], [0, 0])
    assert eng.dims() == (20, 20)   # handle still usable


def test_empty_and_single_inputs(fitted):
    eng, ref, x, y = fitted
    assert eng.transform(np.zeros((0, 20)), np.zeros(0, np.uint64)) == {}
    assert eng.score_matrix((1, np.zeros((0, 20))), (1, np.zeros((3, 20)))).shape == (0, 3)
    assert eng.score_matrix((1, np.zeros((3, 20))), (1, np.zeros((0, 20)))).shape == (3, 0)
    assert eng.score_trials((np.ones(1, np.int32), np.zeros((1, 20))), (1, np.zeros((1, 20))), [], []).shape == (0,)
    assert eng.norm(x[:10], {}) is None
    one = eng.transform(x[:1], np.array([123456789012], np.uint64))     # 64-bit label (quirk Q11)
    assert list(one) == [123456789012] and one[123456789012][0] == 1


def test_strided_output_and_host_slabs(fitted, oracle):
    eng, ref, x, y = fitted
    rng = np.random.default_rng(2)
    U = np.stack([oracle.transform_ivector(ref, r, 2) for r in rng.random((300, 20))])
    V = np.stack([oracle.transform_ivector(ref, r, 1) for r in rng.random((77, 20))])
    S_ref = oracle.score_block(ref["psi"], U, 2, V)
    big = np.full((300, 100), -7.0, np.float32)                # ld_out = 100 > Nt = 77
    rc = eng._lib.plda_score_matrix(eng._h, U.ctypes.data, None, 2, 300, V.ctypes.data, 77, None, None,
                                    big.ctypes.data, 100)
    assert rc == 0
    assert (np.abs(big[:, :77] - S_ref) <= score_tol(S_ref)).all() and (big[:, 77:] == -7.0).all()


def test_norm_numutts_subset_and_persistence(fitted, oracle, tmp_path):
    from plda_amd import MPlda
    eng0, ref, x, y = fitted
    eng = MPlda(0); eng.set_model(ref["mean"], ref["transform"], ref["psi"])
    enrol = eng.transform(x[:90], y[:90])
    ids = list(enrol)
    models = np.stack([enrol[k][1] for k in ids])
    bkg = x[300:420]
    eng.norm(bkg, enrol, 50)                                   # 50 of 120 rows, num_examples stays 120 (:224)
    sel = np.sort(np.random.default_rng(0).permutation(120)[:50])
    t = np.stack([oracle.transform_ivector(ref, r, 120) for r in bkg[sel]])
    S = np.array([[oracle.llr(ref["psi"], tr, 1, m) for m in models] for tr in t])
    zm, zs = eng.znorm_stats()
    np.testing.assert_allclose([zm[k] for k in ids], S.mean(0), rtol=1e-4, atol=1e-4 * np.abs(S.mean(0)).mean())
    np.testing.assert_allclose([zs[k] for k in ids], S.std(0), rtol=2e-4)
    path = os.path.join(tmp_path, "model.npz")
    eng.save(path)
    eng2 = MPlda(0); eng2.load(path)
    assert eng2.znorm_stats() == eng.znorm_stats()
    a = eng.score(ids[1], enrol[ids[1]], (1, models[3]))
    b = eng2.score(ids[1], enrol[ids[1]], (1, models[3]))
    assert a == b
    for k in ("mean", "transform", "psi", "offset"):
        np.testing.assert_array_equal(eng.get_model()[k], eng2.get_model()[k])


def test_float32_and_fortran_inputs_are_coerced(fitted, oracle):
    """superset of the reference, which silently mis-reads such arrays (quirk Q12)."""
    eng, ref, x, y = fitted
    a = eng.transform(np.asfortranarray(x[:40]), y[:40].astype(np.uint8))
    b = eng.transform(x[:40], y[:40])
    assert list(a) == list(b)
    for k in a:
        np.testing.assert_array_equal(a[k][1], b[k][1])
    c = eng.transform(x[:40].astype(np.float32), y[:40])
    for k in c:
        np.testing.assert_allclose(c[k][1], b[k][1], rtol=1e-5, atol=1e-6)


def test_fit_degenerate_settings(oracle):
    """iters = 0 (W = B = I, psi = 1), singleton speakers (n_k = 1), and more dims than samples."""
    from plda_amd import MPlda
    x, y = make_data(71, 300, 12, 30, skew=True)
    eng = MPlda(0)
    eng.fit(x, y, 0)
    m = eng.get_model()
    np.testing.assert_allclose(m["psi"], 1.0, atol=1e-13)
    np.testing.assert_allclose(m["transform"].T @ m["transform"], np.eye(12), atol=1e-12)
    # every speaker a singleton except one pair: scatter has rank 1, the EM prior keeps W positive definite
    ys = np.arange(40, dtype=np.uint64); ys[39] = 38
    xs = np.random.default_rng(1).random((40, 6))
    eng.fit(xs, ys, 4)
    ref = oracle.fit(xs, ys, 4)
    assert np.abs(eng.get_model()["psi"] - ref["psi"]).max() <= 1e-8 * ref["psi"].max()
    # D > N
    xw = np.random.default_rng(2).random((30, 64)); yw = (np.arange(30) % 5).astype(np.uint64)
    eng.fit(xw, yw, 3)
    refw = oracle.fit(xw, yw, 3)
    g = eng.get_model()
    assert np.abs(g["psi"] - refw["psi"]).max() <= 1e-8 * refw["psi"].max()
    assert np.abs(g["transform"].T @ g["transform"] - refw["transform"].T @ refw["transform"]).max() <= \
        1e-7 * np.abs(refw["transform"].T @ refw["transform"]).max()
    with pytest.raises(RuntimeError):
        eng.fit(xw, yw, -1)


def test_kaldi_file_interchange(tmp_path, oracle):
    """save_kaldi / load_kaldi: a model written in Kaldi's Plda layout scores identically after reloading
    into a fresh engine (binary and text)."""
    from plda_amd import MPlda
    x, y = make_data(77, 900, 24, 30, scale_between=0.5)
    a = MPlda(0)
    a.fit(x, y, 4)
    enrol = a.transform(x[:100], y[:100])
    test = a.transform(x[100:160], np.arange(60, dtype=np.uint64))
    want = a.score_matrix(enrol, test, znorm=False)
    for binary in (True, False):
        p = str(tmp_path / ("m%d.plda" % binary))
        a.save_kaldi(p, binary)
        b = MPlda(0).load_kaldi(p)
        np.testing.assert_array_equal(b.get_model()["psi"], a.get_model()["psi"])
        np.testing.assert_array_equal(b.score_matrix(enrol, test, znorm=False), want)


def test_one_handle_from_several_threads():
    """ctypes releases the GIL during a call, so Python threads do reach the library concurrently: calls on one
    handle are serialised by its mutex and every thread gets the right answers."""
    import threading
    from plda_amd import MPlda
    x, y = make_data(5, 800, 32, 20, scale_between=0.5)
    eng = MPlda(0)
    eng.fit(x, y, 3)
    enrol = eng.transform(x[:60], y[:60])
    test = eng.transform(x[60:100], np.arange(40, dtype=np.uint64))
    want_S = eng.score_matrix(enrol, test, znorm=False)
    ids = sorted(enrol)
    want = {(i, j): eng.score(i, enrol[i], test[j]) for i in ids[:5] for j in range(8)}
    errors = []

    def worker(seed):
        try:
            rng = np.random.default_rng(seed)
            for _ in range(40):
                i, j = ids[int(rng.integers(0, 5))], int(rng.integers(0, 8))
                if eng.score(i, enrol[i], test[j]) != want[(i, j)]:
                    errors.append("score mismatch")
                if seed % 2 and not np.array_equal(eng.score_matrix(enrol, test, znorm=False), want_S):
                    errors.append("matrix mismatch")
                if seed % 3 == 0:
                    t = eng.transform(x[:60], y[:60])
                    if not all(np.array_equal(t[k][1], enrol[k][1]) for k in enrol):
                        errors.append("transform mismatch")
        except Exception as ex:      # noqa: BLE001
            errors.append(repr(ex))
    threads = [threading.Thread(target=worker, args=(s,)) for s in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_absurd_size_returns_an_error_code():
    """Nothing aborts and nothing throws across the ABI: a request no machine can hold comes back as a status
    code with a message (the reference would die in a Kaldi assertion or std::bad_alloc, pldamodule.cpp has no
    try/catch).  Every entry point body runs inside api.hip's `guarded`."""
    import ctypes as C
    from plda_amd import MPlda, _native as N
    eng = MPlda(0)
    d = 8
    rng = np.random.default_rng(0)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    eng.set_model(rng.random(d), q, np.sort(rng.random(d))[::-1].copy())
    lib = N.load()
    x = np.zeros((4, d)); y = np.zeros(4, np.uint64)
    cap = C.c_int64(4)
    ol, oc, ov = np.zeros(4, np.uint64), np.zeros(4, np.int64), np.zeros((4, d))
    huge = 1 << 44
    rc = lib.plda_transform_groups(eng._h, x.ctypes.data, huge, d, y.ctypes.data, ol.ctypes.data, oc.ctypes.data,
                                   ov.ctypes.data, C.byref(cap))
    assert rc != N.PLDA_OK and N.last_error(eng._h)
    out = np.zeros(4, np.float32)
    rc = lib.plda_score_matrix(eng._h, x.ctypes.data, None, 1, huge, x.ctypes.data, huge, None, None, out.ctypes.data, huge)
    assert rc != N.PLDA_OK
    # the handle is still usable
    got = eng.transform(rng.random((6, d)), np.array([5, 5, 2**40 + 3, 7, 2**40 + 3, 5], np.uint64))
    assert list(got.keys()) == [5, 7, 2**40 + 3] and [got[k][0] for k in got] == [3, 1, 2]


def test_host_score_matrix_row_slabs_pack_the_test_side_once(oracle):
    """plda_score_matrix stages > 1 GiB of scores in row slabs; the test side is packed for the first slab
    only (reuse_packed_B) -- every slab must still see it."""
    from plda_amd import MPlda
    from oracle import plda_oracle_np as onp
    eng = MPlda(0)
    d = 24
    rng = np.random.default_rng(1)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    psi = np.sort(rng.random(d) * 3 + 0.1)[::-1].copy()
    eng.set_model(rng.random(d), q, psi)
    m, nt = 3000, 150000                     # 1.8 GB of fp32 scores -> slabs of 1664 rows
    U, V = rng.standard_normal((m, d)), rng.standard_normal((nt, d))
    S = eng.score_matrix((2, U), (1, V))
    rows = np.array([0, 1, 1663, 1664, 1665, 2999]); cols = rng.integers(0, nt, 3000)
    ref = onp.llr_matrix(psi, U[rows], 2, V[cols])
    got = S[np.ix_(rows, cols)].astype(np.float64)
    tol = 1e-4 * np.maximum(np.abs(ref), np.abs(ref).mean())
    assert (np.abs(got - ref) <= tol).all()


def test_singular_within_class_covariance_is_reported_by_the_deferred_check():
    """EM keeps W positive definite for finite data, so the failing input is a NaN feature: after one EM iteration
    W is NaN and GetOutput's Cholesky fails.  The flag is read back with the model copies at the end of the fit (no
    host round trip inside GetOutput), and the direct eigensolver, fed the NaNs of the failed factor, must neither
    fault nor hang on the way there."""
    from plda_amd import MPlda
    from plda_amd._native import PldaError
    rng = np.random.default_rng(5)
    X = rng.random((400, 24))
    X[7, 3] = np.nan
    y = np.repeat(np.arange(40, dtype=np.uint64), 10)
    eng = MPlda(0)
    with pytest.raises(PldaError, match="positive definite"):
        eng.fit(X, y, 1)
    # the handle stays usable
    X[7, 3] = 0.5
    eng.fit(X, y, 2)
    assert np.isfinite(eng.get_model()["psi"]).all()


def test_trace_spans_cover_the_fit_stages():
    from plda_amd import MPlda
    rng = np.random.default_rng(6)
    X = rng.random((600, 40))
    y = np.repeat(np.arange(60, dtype=np.uint64), 10)
    eng = MPlda(0)
    eng.trace_enable(True)
    eng.fit(X, y, 3)
    spans = {s["name"]: s for s in eng.trace_read()}
    for name in ("fit.label_sort (K1a)", "fit.centroids (K1)", "fit.scatter_syrk (K2)", "fit.em (all iterations)",
                 "getoutput.whiten (chol + inverse)", "getoutput.eig.tridiagonalise", "getoutput.eig.divide_and_conquer",
                 "getoutput.eig.back_transform", "getoutput.transform"):
        assert name in spans and spans[name]["calls"] == 1 and spans[name]["ms"] > 0.0, name
    # (the scatter SYRK carries the centroids' term -M^T M in the same launch: N + K rows)
    assert spans["fit.scatter_syrk (K2)"]["work"] == (600 + 60) * 40 * 41.0 and spans["fit.scatter_syrk (K2)"]["unit"] == "flop"
    assert eng.trace_read() == []          # reset by the first read
    eng.trace_enable(False)
    eng.fit(X, y, 1)
    assert eng.trace_read() == []


def test_transform_result_is_a_dict_that_remembers_its_arrays(oracle):
    """transform() returns the reference's dict {label: (n, vec)} (pldamodule.cpp:162-191) as a dict SUBCLASS that keeps the
    arrays it was built from, so score_matrix / score_trials / norm skip the per-entry re-stacking.  It must behave as the plain
    dict in every use: same scores as a plain copy; a change of the keys (delete, insert, pop, update) falls back to unpacking;
    writing into a vector is seen by both; copies and pickles are ordinary dicts again."""
    import copy
    import pickle
    from liblda import PLDA
    from plda_amd.libplda import Transformed
    x, y = make_data(12, 900, 24, 30)
    p = PLDA()
    p.fit(x, y, 3)
    enrol = p.transform(x[:600], y[:600])
    test = p.transform(x[600:], np.arange(300, dtype=np.uint64))
    assert isinstance(enrol, dict) and isinstance(enrol, Transformed) and enrol._packed is not None
    k0 = next(iter(enrol))
    assert isinstance(k0, int) and isinstance(enrol[k0], tuple) and isinstance(enrol[k0][0], int)
    S = p.score_matrix(enrol, test)
    assert np.array_equal(S, p.score_matrix(dict(enrol), dict(test)))             # plain copies: the generic path
    assert np.array_equal(S, p.score_matrix(copy.copy(enrol), pickle.loads(pickle.dumps(test))))
    # writing into a vector (a view of the remembered block) is seen by both paths
    enrol[k0][1][:] *= 0.5
    S2 = p.score_matrix(enrol, test)
    assert not np.array_equal(S2[0], S[0]) and np.array_equal(S2, p.score_matrix(dict(enrol), test))
    # z-norm statistics from the remembered block and from a plain dict
    q = PLDA(); q._instance.set_model(*[p._instance.get_model()[k] for k in ("mean", "transform", "psi")])
    p.norm(x[:200], enrol); q.norm(x[:200], dict(enrol))
    assert p._instance.znorm_stats() == q._instance.znorm_stats()
    # a change of the keys drops the memory
    removed = enrol.pop(k0)
    assert enrol._packed is None and p.score_matrix(enrol, test, znorm=False).shape == (len(enrol), len(test))
    assert np.array_equal(p.score_matrix(enrol, test, znorm=False), S2x := p.score_matrix(dict(enrol), test, znorm=False)) and S2x.shape[0] == S.shape[0] - 1
    test2 = p.transform(x[600:], np.arange(300, dtype=np.uint64))
    test2[10 ** 6] = removed
    assert test2._packed is None and p.score_matrix(enrol, test2, znorm=False).shape == (len(enrol), 301)
    del test2[10 ** 6]
    assert np.array_equal(p.score_matrix(enrol, test2, znorm=False), S2x)


def test_znorm_lookup_for_many_models_matches_the_per_key_lookup(oracle):
    """score_matrix with z-norm statistics for >= 256 models takes the statistics from one sorted copy (vectorised lookup)
    instead of two dict probes per model: same rows normalised, same values, models without statistics left as they are, a
    second norm() (insert-once: pldamodule.cpp:245,250) adds only the new labels and is seen by the next call."""
    from liblda import PLDA
    x, y = make_data(21, 1200, 16, 40, scale_between=0.3)
    p = PLDA()
    p.fit(x, y, 3)
    enrol = p.transform(x[:600], np.arange(600, dtype=np.uint64) * 7 + 3)       # 600 models, sparse labels
    test = p.transform(x[600:900], np.arange(300, dtype=np.uint64))
    first = {k: enrol[k] for k in list(enrol)[:350]}
    p.norm(x[900:1100], first)
    zm, zs = p._instance.znorm_stats()
    assert len(zm) == 350
    raw = p.score_matrix(enrol, test, znorm=False)
    got = p.score_matrix(enrol, test)
    keys = list(enrol)
    for i in (0, 1, 349, 350, 599):
        k = keys[i]
        want = (raw[i] - zm[k]) / zs[k] if k in zm else raw[i]
        np.testing.assert_allclose(got[i], want, rtol=2e-5, atol=2e-5)
    small = {k: enrol[k] for k in keys[340:360]}                                 # < 256 models: the per-key path
    np.testing.assert_array_equal(p.score_matrix(small, test), got[340:360])
    p.norm(x[1000:1200], enrol)                                                  # the other 250 labels; the first 350 keep theirs
    zm2, zs2 = p._instance.znorm_stats()
    assert len(zm2) == 600 and all(zm2[k] == zm[k] and zs2[k] == zs[k] for k in zm)
    got2 = p.score_matrix(enrol, test)
    np.testing.assert_array_equal(got2[:350], got[:350])
    np.testing.assert_allclose(got2[599], (raw[599] - zm2[keys[599]]) / zs2[keys[599]], rtol=2e-5, atol=2e-5)


DIAG_ONLY_VARIANTS = [1, 2, 3, 4, 9, 10, 11, 12, 31, 33, 34, 35, 36, 37, 41, 44, 45, 46, 47, 54, 58, 62, 63]


@pytest.mark.parametrize("variant", DIAG_ONLY_VARIANTS)
def test_measurement_arms_are_not_in_the_product_library(monkeypatch, variant):
    """PLDA_GEMM_VARIANT selects, among others, bounding arms of the trials GEMM that skip the operand DMA or the stores and
    return GARBAGE scores with rc = 0 (round-5 review).  They exist only in libplda_hip_diag.so (-DPLDA_DIAG=1); the library
    every caller gets refuses to create a handle under such a variant instead of silently corrupting scores."""
    from plda_amd import MPlda, _native
    from plda_amd._native import PldaError
    assert _native.load().plda_build_flags() == 0
    monkeypatch.setenv("PLDA_GEMM_VARIANT", str(variant))
    with pytest.raises(PldaError, match="measurement arm"):
        MPlda(0)
    monkeypatch.delenv("PLDA_GEMM_VARIANT")
    MPlda(0)                                            # (and nothing sticks)


def test_diagnostic_library_still_has_them(monkeypatch):
    """The diagnostic build accepts the same variants (bench.py's shader-clock reading uses 47: the product kernel + stamps,
    whose scores are the product's)."""
    import os
    from plda_amd import MPlda, _native
    if not os.path.exists(_native.SO_DIAG_PATH):
        pytest.skip("libplda_hip_diag.so not built (python -m plda_amd.build --diag)")
    assert _native.load(diag=True).plda_build_flags() == 1
    rng = np.random.default_rng(4)
    d, m, nt = 80, 2304, 49152                          # 9 x 192 tiles of 256 x 256: the one-wave-per-SIMD kernel's range
    psi = np.sort(rng.random(d) + 0.05)[::-1].copy()
    U, V = rng.standard_normal((m, d)), rng.standard_normal((nt, d))
    outs = []
    for diag in (False, True):
        if diag:
            monkeypatch.setenv("PLDA_GEMM_VARIANT", "47")
        eng = MPlda(0, diag=diag)
        eng.set_model(np.zeros(d), np.eye(d), psi)
        outs.append(eng.score_matrix((1, U), (1, V)))
    assert np.array_equal(outs[0], outs[1])
