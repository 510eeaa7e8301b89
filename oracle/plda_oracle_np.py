"""oracle/plda_oracle_np.py -- independent NumPy restatement of the PLDA hot path.

TEST INFRASTRUCTURE ONLY (see oracle/plda_oracle.h).  PARITY UNPINNED: Kaldi is
absent and the reference's tests pin no values; this file is a second,
independently written statement of SURVEY.md Appendix A (using np.linalg for
the factorisations, i.e. different code from both the C oracle and the HIP
kernels) used to cross-check the C oracle and to emit tests/golden/*.npz.

Reference call sites followed: /root/reference/src/pldamodule.cpp
  fit        :42-109   (AddSamples(1/n_k) :94-98, Sort :100, Estimate :102-106)
  transform  :111-194  (per-label mean :147-168, TransformIvector :171)
  norm       :196-256  (num_examples=Nb :224, roles swapped n=1 :235, pop. std :240-250)
  score      :258-277  (LogLikelihoodRatio :266, z-norm :269-273)
"""
import numpy as np

LOG_2PI = 1.8378770664093454835606594728112


def stats(X, labels):
    """PldaStats after the wrapper's AddSamples loop (w_k = 1/n_k)."""
    X = np.asarray(X, np.float64)
    labels = np.asarray(labels).astype(np.int64)
    K = int(labels.max()) + 1
    D = X.shape[1]
    counts = np.bincount(labels, minlength=K)
    assert (counts > 0).all(), "fit labels must be dense 0..K-1"
    sums = np.zeros((K, D))
    np.add.at(sums, labels, X)
    means = sums / counts[:, None]
    w = 1.0 / counts
    wrow = w[labels]
    scatter = (X * wrow[:, None]).T @ X - (means * (counts * w)[:, None]).T @ means
    scatter = 0.5 * (scatter + scatter.T)
    sum_ = (means * w[:, None]).sum(0)
    return dict(means=means, counts=counts, scatter=scatter, sum=sum_,
                class_weight=w.sum(), example_weight=float((w * counts).sum()))


def em_iter(st, W, B):
    """One EstimateOneIter, vectorised over classes of equal n."""
    D = W.shape[0]
    means, counts = st["means"], st["counts"]
    w = 1.0 / counts
    mu = st["sum"] / st["class_weight"]
    Winv = np.linalg.inv(W)
    Binv = np.linalg.inv(B)
    Wst = st["scatter"].copy()
    Bst = np.zeros((D, D))
    Wc = st["example_weight"] - st["class_weight"]
    Bc = 0.0
    for n in np.unique(counts):
        sel = counts == n
        mixed = np.linalg.inv(Binv + n * Winv)
        m = means[sel] - mu
        ww = (mixed @ (n * (Winv @ m.T))).T
        e = m - ww
        wk = w[sel]
        Bst += wk.sum() * mixed + (ww * wk[:, None]).T @ ww
        Wst += (wk * n).sum() * mixed + (e * (wk * n)[:, None]).T @ e
        Bc += wk.sum()
        Wc += wk.sum()
    Wn = Wst / Wc
    Bn = Bst / Bc
    return 0.5 * (Wn + Wn.T), 0.5 * (Bn + Bn.T)


def get_output(st, W, B):
    """GetOutput."""
    mean = st["sum"] / st["class_weight"]
    C = np.linalg.cholesky(W)
    T1 = np.linalg.inv(C)
    Bp = T1 @ B @ T1.T
    s, U = np.linalg.eigh(0.5 * (Bp + Bp.T))
    s = np.maximum(s, 0.0)
    order = np.argsort(-s, kind="stable")
    s, U = s[order], U[:, order]
    transform = U.T @ T1
    return dict(mean=mean, transform=transform, psi=s, offset=-transform @ mean)


def fit(X, labels, iters=10, return_wb=False):
    st = stats(X, labels)
    D = X.shape[1]
    W, B = np.eye(D), np.eye(D)
    for _ in range(iters):
        W, B = em_iter(st, W, B)
    model = get_output(st, W, B)
    if return_wb:
        model["W"], model["B"] = W, B
    return model


def _inv_longdouble(A):
    """Gauss-Jordan with partial pivoting in np.longdouble (numpy.linalg has no extended-precision path)."""
    m = A.shape[0]
    M = np.concatenate([A.astype(np.longdouble), np.eye(m, dtype=np.longdouble)], 1)
    for p in range(m):
        q = p + int(np.argmax(np.abs(M[p:, p])))
        if q != p:
            M[[p, q]] = M[[q, p]]
        M[p] /= M[p, p]
        col = M[:, p].copy()
        col[p] = 0
        M -= np.outer(col, M[p])
    return M[:, m:]


def fit_wb_longdouble(X, labels, iters):
    """stats + em_iter above, step for step, in x87 extended precision (64-bit mantissa, eps 1.1e-19): the
    yardstick for ill-conditioned fits (fewer samples than dimensions), where two fp64 implementations of the same
    EM legitimately differ by cond * eps and the question is which one is off.  Returns (W, B) as longdouble;
    O(iters * groups * D^3) in NumPy row operations -- D of a few hundred at most."""
    LD = np.longdouble
    assert np.finfo(LD).eps < 1e-18, "np.longdouble is not extended precision on this platform"
    X = np.asarray(X, np.float64).astype(LD)
    labels = np.asarray(labels).astype(np.int64)
    K = int(labels.max()) + 1
    D = X.shape[1]
    counts = np.bincount(labels, minlength=K)
    means = np.stack([X[labels == c].sum(0) / LD(counts[c]) for c in range(K)])
    w = LD(1) / counts.astype(LD)
    scatter = np.zeros((D, D), LD)
    for c in range(K):
        xc = X[labels == c] - means[c]
        scatter += (xc.T @ xc) * w[c]
    mu = (means * w[:, None]).sum(0) / w.sum()
    W, B = np.eye(D, dtype=LD), np.eye(D, dtype=LD)
    for _ in range(iters):
        Winv, Binv = _inv_longdouble(W), _inv_longdouble(B)
        Wst, Bst = scatter.copy(), np.zeros((D, D), LD)
        for n in np.unique(counts):
            sel = counts == n
            kg = LD(int(sel.sum()))
            mixed = _inv_longdouble(Binv + LD(int(n)) * Winv)
            m = means[sel] - mu
            ww = (mixed @ (LD(int(n)) * (Winv @ m.T))).T
            e = m - ww
            Bst += (kg / LD(int(n))) * mixed + (ww.T @ ww) / LD(int(n))
            Wst += kg * mixed + e.T @ e
        W, B = Wst / LD(K), Bst / w.sum()
        W, B = (W + W.T) / 2, (B + B.T) / 2
    return W, B


def transform_ivector(model, x, n, normalize_length=True, simple_length_norm=False):
    """TransformIvector (A.5); x may be [D] or [R,D] with n scalar or [R]."""
    x = np.atleast_2d(np.asarray(x, np.float64))
    n = np.broadcast_to(np.asarray(n, np.float64), (x.shape[0],))
    t = x @ model["transform"].T + model["offset"]
    D = t.shape[1]
    if simple_length_norm:
        f = np.sqrt(D) / np.linalg.norm(t, axis=1)
    else:
        f = np.sqrt(D / (t * t / (model["psi"][None, :] + 1.0 / n[:, None])).sum(1))
    return t * f[:, None] if normalize_length else t


def transform_groups(model, X, labels):
    """Mplda_transform: ascending label order, (label, n, vec)."""
    X = np.asarray(X, np.float64)
    labels = np.asarray(labels)
    uniq, inv, counts = np.unique(labels, return_inverse=True, return_counts=True)
    sums = np.zeros((len(uniq), X.shape[1]))
    np.add.at(sums, inv, X)
    means = sums * (1.0 / counts)[:, None]
    return uniq, counts, transform_ivector(model, means, counts)


def llr_pair(psi, u, n, v):
    """LogLikelihoodRatio(u, n, v) per-pair form (A.5)."""
    mean = n * psi / (n * psi + 1.0) * u
    var = 1.0 + psi / (n * psi + 1.0)
    given = -0.5 * (np.log(var).sum() + LOG_2PI * len(psi) + ((v - mean) ** 2 / var).sum())
    var0 = 1.0 + psi
    without = -0.5 * (np.log(var0).sum() + LOG_2PI * len(psi) + (v * v / var0).sum())
    return given - without


def llr_matrix(psi, U, n, V):
    """Batched GEMM form of the same LLR (A.5): S = A1 V^T + A2 (V*V)^T + r."""
    U = np.asarray(U, np.float64)
    V = np.asarray(V, np.float64)
    n = np.broadcast_to(np.asarray(n, np.float64), (U.shape[0],))[:, None]
    c = n * psi / (n * psi + 1.0)
    var = 1.0 + psi / (n * psi + 1.0)
    A1 = c * U / var
    A2 = -0.5 * (1.0 / var - 1.0 / (1.0 + psi))
    r = -0.5 * (np.log(var).sum(1) - np.log(1.0 + psi).sum() + (c * c * U * U / var).sum(1))
    return A1 @ V.T + A2 @ (V * V).T + r[:, None]


def norm(model, bkg, models):
    """MPlda_norm with numutts=0: returns per-model (mean, population std)."""
    bkg = np.asarray(bkg, np.float64)
    t = transform_ivector(model, bkg, bkg.shape[0])  # quirk Q6
    S = llr_matrix(model["psi"], t, 1, models)       # quirk Q7: cohort is the train side
    return S.mean(0), S.std(0)


def smooth(model, f):
    """SmoothWithinClassCovariance (A.6); returns a new model dict."""
    wc = 1.0 + f * model["psi"]
    out = dict(model)
    out["psi"] = model["psi"] / wc
    out["transform"] = model["transform"] * (wc ** -0.5)[:, None]
    out["offset"] = -out["transform"] @ model["mean"]
    return out


def dvector_pool(frames, offsets, method="mean", l2norm=True):
    """d-vector front-end (scoring/extractdvector.py:19-59): getnormalizedvector (:19-29)
    then np.mean / np.max / np.var over each utterance's frames (:32-47), in float64."""
    frames = np.asarray(frames, np.float64)
    out = []
    for a, b in zip(offsets[:-1], offsets[1:]):
        u = frames[a:b]
        if l2norm:
            u = u / np.linalg.norm(u, axis=1)[:, np.newaxis]
        out.append({"mean": np.mean, "max": np.max, "var": np.var}[method](u, axis=0))
    return np.array(out)


def eer(negatives, positives):
    """Equal error rate as scoring/eer.py:68-73 computes it through bob.measure (absent,
    un-pinned: published definition restated; PARITY UNPINNED for this row):
      farfrr(neg, pos, t): FAR = #{neg >= t}/Nn, FRR = #{pos < t}/Np;
      eer_threshold: candidates are the minimum score, then the midpoint after each distinct
      score of the union (last one: score + 1e-8); minimal |FAR - FRR| wins, later on ties.
    Returns (threshold, far, frr, eer) computed on float32 scores in float64."""
    neg = np.sort(np.asarray(negatives, np.float32).astype(np.float64))
    pos = np.sort(np.asarray(positives, np.float32).astype(np.float64))
    s = np.unique(np.concatenate([neg, pos]))
    mids = np.concatenate([[s[0]], s[:-1] + (s[1:] - s[:-1]) / 2.0, [s[-1] + 1e-8]])
    far = (len(neg) - np.searchsorted(neg, mids, side="left")) / len(neg)   # #{neg >= t} / Nn
    frr = np.searchsorted(pos, mids, side="left") / len(pos)           # pos < t
    pred = np.abs(far - frr)
    best = len(pred) - 1 - int(np.argmin(pred[::-1]))                  # later candidate on ties
    return float(mids[best]), float(far[best]), float(frr[best]), float(0.5 * (far[best] + frr[best]))


def det(negatives, positives, n_points):
    """The points of the DET curve scoring/eer.py:34-62 plots through bob.measure.plot.det(negatives, positives, 100)
    (absent, un-pinned: published definition restated; PARITY UNPINNED): n_points thresholds from the smallest to the
    largest score of the union, accumulated as t_0 = min, t_{i+1} = t_i + (max - min) / (n_points - 1) in float64;
    FAR_i = #{neg >= t_i} / Nn, FRR_i = #{pos < t_i} / Np (bob.measure.farfrr).  Returns (thresholds, far, frr); the plot's
    axes are ppndf (the normal deviate) of the two rates: `ppndf` below."""
    neg = np.sort(np.asarray(negatives, np.float32).astype(np.float64))
    pos = np.sort(np.asarray(positives, np.float32).astype(np.float64))
    lo = min(neg[0], pos[0]); hi = max(neg[-1], pos[-1])
    step = (hi - lo) / (n_points - 1.0)
    thr = np.empty(n_points)
    t = lo
    for i in range(n_points):
        thr[i] = t
        t += step
    far = (len(neg) - np.searchsorted(neg, thr, side="left")) / len(neg)
    frr = np.searchsorted(pos, thr, side="left") / len(pos)
    return thr, far, frr


def ppndf(p):
    """The normal deviate of a rate as bob.measure.ppndf defines it: the probit of p clipped to [eps, 1 - eps],
    eps = 2.2204e-16 (restated)."""
    from scipy.special import ndtri
    eps = 2.2204e-16
    return ndtri(np.clip(np.asarray(p, np.float64), eps, 1.0 - eps))
