"""Prototype (NumPy, CPU) of the symmetric eigensolver that csrc/eig_dc.hip implements on the device:
Householder tridiagonalisation -> divide and conquer on the tridiagonal (Cuppen; deflation as LAPACK dlaed2,
secular roots by the two-pole rational iteration with a bisection safeguard, eigenvectors through the
Gu-Eisenstat recomputed z) -> back-transformation.  Written in the shape of the kernels (what is sequential
there is a loop here, what is one-thread-per-root there is vectorised here) so that intermediate values can
be compared.  Development tool only: nothing in the product or the tests imports it.

    python scripts/proto_dc_eig.py            # self-test against numpy.linalg.eigh on hard matrices
"""
import numpy as np

EPS = np.finfo(np.float64).eps


def householder_tridiag(G):
    """A = Q T Q^T; returns d, e, V (reflector vectors in rows, v_j[j+1] = 1 implicit NOT used: full vectors), tau."""
    A = np.array(G, dtype=np.float64)
    n = A.shape[0]
    V = np.zeros((n, n))
    tau = np.zeros(n)
    for j in range(n - 2):
        x = A[j + 1:, j].copy()
        sigma = float(x[1:] @ x[1:])
        if sigma == 0.0:
            continue                                   # already tridiagonal in this column
        nx = np.sqrt(x[0] * x[0] + sigma)
        alpha = -nx if x[0] >= 0 else nx
        v = x.copy()
        v[0] -= alpha
        t = 2.0 / float(v @ v)
        V[j, j + 1:] = v
        tau[j] = t
        p = t * (A[j + 1:, j + 1:] @ v)
        w = p - (0.5 * t * float(p @ v)) * v
        A[j + 1:, j + 1:] -= np.outer(v, w) + np.outer(w, v)
        A[j + 1, j] = A[j, j + 1] = alpha
        A[j + 2:, j] = 0.0
        A[j, j + 2:] = 0.0
    d = np.diag(A).copy()
    e = np.diag(A, -1).copy()
    return d, e, V, tau


def back_transform(V, tau, Z):
    """columns of Z are eigenvectors of T -> eigenvectors of A: Z <- H_0 H_1 ... H_{n-3} Z."""
    n = Z.shape[0]
    Z = Z.copy()
    for j in range(n - 3, -1, -1):
        if tau[j] == 0.0:
            continue
        v = V[j]
        Z -= tau[j] * np.outer(v, v @ Z)
    return Z


def leaf_ql(d, e):
    """implicit QL with Wilkinson shift (tqli); returns eigenvalues (unsorted) and eigenvectors in columns."""
    n = len(d)
    d = d.copy()
    e = np.concatenate([e, [0.0]])
    Z = np.eye(n)
    for l in range(n):
        it = 0
        while True:
            m = l
            while m < n - 1:
                dd = abs(d[m]) + abs(d[m + 1])
                if abs(e[m]) <= EPS * dd:
                    break
                m += 1
            if m == l:
                break
            it += 1
            assert it < 60
            g = (d[l + 1] - d[l]) / (2.0 * e[l])
            r = np.hypot(g, 1.0)
            g = d[m] - d[l] + e[l] / (g + (r if g >= 0 else -r))
            s = c = 1.0
            p = 0.0
            i = m - 1
            underflow = False
            while i >= l:
                f = s * e[i]
                b = c * e[i]
                r = np.hypot(f, g)
                e[i + 1] = r
                if r == 0.0:
                    d[i + 1] -= p
                    e[m] = 0.0
                    underflow = True
                    break
                s = f / r
                c = g / r
                g = d[i + 1] - p
                r = (d[i] - g) * s + 2.0 * c * b
                p = s * r
                d[i + 1] = g + p
                g = c * r - b
                zi1 = Z[:, i + 1].copy()
                Z[:, i + 1] = s * Z[:, i] + c * zi1
                Z[:, i] = c * Z[:, i] - s * zi1
                i -= 1
            if underflow:
                continue
            d[l] -= p
            e[l] = g
            e[m] = 0.0
    return d, Z


def secular_roots(dk, zk, rho):
    """Roots of 1 + rho sum z_i^2 / (d_i - lam) for sorted distinct dk (k), rho > 0.
    Returns origin index o[j], mu[j] (lam_j = dk[o[j]] + mu[j]) and the k x k matrix delta[i, j] = dk[i] - lam_j
    computed as (dk[i] - dk[o[j]]) - mu[j]."""
    k = len(dk)
    z2 = zk * zk
    if k == 1:
        mu = np.array([rho * z2[0]])
        return np.array([0]), mu, np.array([[-mu[0]]])
    zz = float(z2.sum())
    org = np.zeros(k, np.int64)
    lo = np.zeros(k)
    hi = np.zeros(k)
    j = np.arange(k)
    jn = np.minimum(j + 1, k - 1)
    gap = dk[jn] - dk[j]
    gap[k - 1] = rho * zz
    # origin: sign of f at the interval midpoint (coordinates relative to d_j)
    Dj = dk[:, None] - dk[None, :]                    # [i, j] = d_i - d_j
    mid = 0.5 * gap
    with np.errstate(divide="ignore", invalid="ignore"):
        fmid = 1.0 + rho * (z2[:, None] / (Dj - mid[None, :])).sum(0)
    left = fmid >= 0.0
    left[k - 1] = True
    org = np.where(left, j, jn)
    Dl = dk[:, None] - dk[org][None, :]               # [i, j] = d_i - d_org(j)
    lo = np.where(left, 0.0, -mid)
    hi = np.where(left, mid, 0.0)
    hi[k - 1] = gap[k - 1]
    mu = np.where(left, 0.5 * hi, 0.5 * lo)           # start: quarter points
    mu[k - 1] = 0.5 * gap[k - 1]
    lower = (np.arange(k)[:, None] <= j[None, :])     # i <= j: the psi part
    done = np.zeros(k, bool)
    for it in range(80):
        den = Dl - mu[None, :]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = z2[:, None] / den
            psi = rho * np.where(lower, t, 0.0).sum(0)
            phi = rho * np.where(lower, 0.0, t).sum(0)
            dpsi = rho * np.where(lower, t / den, 0.0).sum(0)
            dphi = rho * np.where(lower, 0.0, t / den).sum(0)
        g = 1.0 + psi + phi
        bound = 8.0 * EPS * (1.0 + np.abs(psi) + np.abs(phi)) * 1.0
        newly = (np.abs(g) <= bound) | ~np.isfinite(g)
        # bracket update (g increasing in mu)
        pos = g > 0
        hi = np.where(~done & pos, mu, hi)
        lo = np.where(~done & ~pos, mu, lo)
        done |= newly
        width_done = (hi - lo) <= 2.0 * EPS * np.maximum(np.abs(lo), np.abs(hi))
        done |= width_done
        if done.all():
            break
        a = Dl[j, j] - mu                              # pole below (d_j - lam), < 0
        b = Dl[jn, j] - mu                             # pole above, > 0 (not for the last root)
        s_psi = dpsi * a * a
        r_psi = psi - dpsi * a
        s_phi = dphi * b * b
        r_phi = phi - dphi * b
        c = 1.0 + r_psi + r_phi
        A = c
        B = c * (a + b) + s_psi + s_phi
        C = a * b * g
        disc = np.maximum(B * B - 4.0 * A * C, 0.0)
        sq = np.sqrt(disc)
        with np.errstate(divide="ignore", invalid="ignore"):
            q = 0.5 * (B + np.where(B >= 0, sq, -sq))
            e1 = q / A
            e2 = C / q
            eta = np.where((e1 > a) & (e1 < b), e1, e2)
            eta = np.where((e1 > a) & (e1 < b) & (e2 > a) & (e2 < b) & (np.abs(e2) < np.abs(e1)), e2, eta)
            # last root: one pole
            c1 = 1.0 + r_psi
            eta_last = a + s_psi / c1
        eta[k - 1] = eta_last[k - 1]
        cand = mu + eta
        bad = ~np.isfinite(cand) | (cand <= lo) | (cand >= hi)
        cand = np.where(bad, 0.5 * (lo + hi), cand)
        mu = np.where(done, mu, cand)
    delta = Dl - mu[None, :]
    return org, mu, delta


def merge(d1, Q1, d2, Q2, e_mid, stats=None):
    """eigen-decomposition of diag(T1', T2') + |e| w w^T from those of the (already torn) halves."""
    n1, n2 = len(d1), len(d2)
    n = n1 + n2
    s = 1.0 if e_mid >= 0 else -1.0
    z = np.concatenate([Q1[-1, :], s * Q2[0, :]]) / np.sqrt(2.0)
    rho = 2.0 * abs(e_mid)
    d = np.concatenate([d1, d2])
    Q = np.zeros((n, n))
    Q[:n1, :n1] = Q1
    Q[n1:, n1:] = Q2
    perm = np.argsort(d, kind="stable")
    d = d[perm]
    z = z[perm]
    Q = Q[:, perm]
    tol = 8.0 * EPS * max(np.abs(d).max(), np.abs(z).max())
    keep = []            # indices (into the sorted arrays) of non-deflated entries, ascending d
    defl = []
    if rho * np.abs(z).max() <= tol:
        return d, Q       # nothing couples
    pj = -1
    for i in range(n):
        if rho * abs(z[i]) <= tol:
            defl.append(i)
            continue
        if pj < 0:
            pj = i
            continue
        # try to rotate z[pj] into z[i]
        tau_ = np.hypot(z[pj], z[i])
        c = z[i] / tau_
        sn = -z[pj] / tau_
        if abs((d[i] - d[pj]) * c * sn) <= tol:
            z[i] = tau_
            z[pj] = 0.0
            qp = Q[:, pj].copy()
            Q[:, pj] = c * qp + sn * Q[:, i]
            Q[:, i] = -sn * qp + c * Q[:, i]
            dp = d[pj] * c * c + d[i] * sn * sn
            di = d[pj] * sn * sn + d[i] * c * c
            d[pj], d[i] = dp, di
            defl.append(pj)
            pj = i
        else:
            keep.append(pj)
            pj = i
    keep.append(pj)
    keep = np.array(keep)
    k = len(keep)
    if stats is not None:
        stats.append((n, k))
    dk = d[keep]
    zk = z[keep]
    # the rotations can break the order by a rounding; the secular solver needs strictly increasing poles
    order = np.argsort(dk, kind="stable")
    dk, zk, keep = dk[order], zk[order], keep[order]
    org, mu, delta = secular_roots(dk, zk, rho)
    lam = dk[org] + mu
    # Gu-Eisenstat: zhat_i^2 = prod_j (lam_j - d_i) / prod_{j != i} (d_j - d_i)
    num = -delta                                        # lam_j - d_i, [i, j]
    dd = dk[None, :] - dk[:, None]                      # d_j - d_i, [i, j]
    np.fill_diagonal(dd, 1.0)
    # pair the factors so that the running product stays O(1): (lam_j - d_i) / (d_j - d_i) for j != i, times (lam_i - d_i)
    ratio = num / dd
    zhat2 = np.abs(np.prod(ratio, axis=1))
    zhat = np.sqrt(zhat2) * np.where(zk >= 0, 1.0, -1.0)
    U = zhat[:, None] / delta
    U /= np.linalg.norm(U, axis=0)[None, :]
    Qn = np.empty((n, n))
    dn = np.empty(n)
    Qn[:, :k] = Q[:, keep] @ U
    dn[:k] = lam
    defl = np.array(defl, np.int64)
    Qn[:, k:] = Q[:, defl]
    dn[k:] = d[defl]
    return dn, Qn


def dc_eig(d, e, leaf=16, stats=None):
    n = len(d)
    if n <= leaf:
        return leaf_ql(d, e)
    m = n // 2
    d = d.copy()
    em = e[m - 1]
    d[m - 1] -= abs(em)
    d[m] -= abs(em)
    d1, Q1 = dc_eig(d[:m], e[:m - 1], leaf, stats)
    d2, Q2 = dc_eig(d[m:], e[m:], leaf, stats)
    return merge(d1, Q1, d2, Q2, em, stats)


def sym_eig(G, leaf=16, stats=None):
    # scaled to max |g_ij| = O(1) by a power of two (exact): sums of squares neither overflow nor underflow
    amax = float(np.abs(G).max())
    if amax == 0.0:
        n = G.shape[0]
        return np.zeros(n), np.eye(n)
    scale = 2.0 ** -np.floor(np.log2(amax))
    d, e, V, tau = householder_tridiag(G * scale)
    lam, Z = dc_eig(d, e, leaf, stats)
    lam = lam / scale
    Z = back_transform(V, tau, Z)
    o = np.argsort(-lam, kind="stable")
    return lam[o], Z[:, o]


def _check(name, G):
    n = G.shape[0]
    st = []
    lam, Z = sym_eig(G, stats=st)
    ref = np.linalg.eigvalsh(G)[::-1]
    nrm = max(np.abs(ref).max(), 1e-300)
    e_val = np.abs(lam - ref).max() / nrm
    e_orth = np.abs(Z.T @ Z - np.eye(n)).max()
    e_res = np.abs(G @ Z - Z * lam[None, :]).max() / nrm
    print("%-28s n=%4d  eigval %.1e  orth %.1e  resid %.1e   merges (n,k): %s" % (name, n, e_val, e_orth, e_res, st[-3:]))
    assert e_val < 1e-13 and e_orth < 1e-12 and e_res < 1e-12, name


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n in (5, 17, 33, 64, 200, 256):
        A = rng.standard_normal((n, n))
        _check("gaussian symmetric", A + A.T)
        B = rng.standard_normal((n, max(n // 3, 1)))
        _check("rank-deficient PSD", B @ B.T)
        _check("identity", np.eye(n))
        _check("zero", np.zeros((n, n)))
        q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        lam = np.concatenate([np.ones(n // 2), np.full(n - n // 2, 2.0)])
        _check("two clusters", (q * lam) @ q.T)
        lam = 10.0 ** (-np.arange(n) * 16.0 / n)
        _check("graded 1..1e-16", (q * lam) @ q.T)
        _check("wilkinson-like tridiagonal", np.diag(np.abs(np.arange(n) - n // 2).astype(float)) + np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1))
        _check("diag + tiny coupling", np.diag(rng.random(n)) + 1e-12 * (A + A.T))
    print("ok")
