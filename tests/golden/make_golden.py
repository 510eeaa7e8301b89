"""tests/golden/make_golden.py -- regenerates tests/golden/*.npz.

The reference cannot be imported or built here (Kaldi / CPython-2),
so these vectors come from the independent NumPy restatement oracle/plda_oracle_np.py
(np.linalg factorisations) on seeded inputs; tests then hold BOTH the C oracle and the
HIP engine to them.  Fixtures are data only: inputs (or their seed) and expected outputs.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import plda_oracle_np as onp  # noqa: E402

CASES = {
    # name: (seed, N, D, K, skew, iters)
    "tiny_unequal": (3, 60, 6, 5, True, 10),          # unequal n_k
    "pldatest_shape": (2, 2000, 10, 10, False, 10),   # tests/pldatest.py:10-11
    "c1_readme": (1, 500, 200, 2, False, 10),         # BASELINE configs[0], README.md:54-55
}


def inputs(seed, n, d, k, skew):
    rng = np.random.default_rng(seed)
    if name_is_c1(seed, n, d, k):
        x = rng.random((n, d))
        y = rng.integers(0, 2, n).astype(np.uint64)
        y[:2] = [0, 1]
        return x, y
    if skew:
        y = np.concatenate([np.arange(k), rng.integers(0, k, n - k)]).astype(np.uint64)
    else:
        y = (np.arange(n) % k).astype(np.uint64)
    return rng.random((n, d)), y


def name_is_c1(seed, n, d, k):
    return (seed, n, d, k) == (1, 500, 200, 2)


def main():
    for name, (seed, n, d, k, skew, iters) in CASES.items():
        x, y = inputs(seed, n, d, k, skew)
        m = onp.fit(x, y, iters, return_wb=True)
        T, psi = m["transform"], m["psi"]
        rng = np.random.default_rng(seed + 100)
        ne, nt, nb = min(40, n // 3), min(30, n // 4), min(50, n // 3)
        ex, ey = x[:ne], y[:ne]
        tx = x[ne:ne + nt]
        bkg = rng.random((nb, d))
        el, ec, ev = onp.transform_groups(m, ex, ey)
        tv = onp.transform_ivector(m, tx, 1)
        S = onp.llr_matrix(psi, ev, ec, tv)
        zm, zs = onp.norm(m, bkg, ev)
        blk = slice(0, min(d, 24))   # large-D cases keep a leading block + traces (small fixtures)
        TtT, TtPsiT = T.T @ T, T.T @ np.diag(psi) @ T
        out = dict(seed=seed, N=n, D=d, K=k, iters=iters, y=y, psi=psi, mean=m["mean"],
                   TtT=TtT[blk, blk], TtPsiT=TtPsiT[blk, blk], W=m["W"][blk, blk], B=m["B"][blk, blk],
                   traces=np.array([np.trace(TtT), np.trace(TtPsiT), np.trace(m["W"]), np.trace(m["B"])]),
                   enrol_n=ne, test_n=nt, bkg=bkg, enrol_labels=el, enrol_counts=ec,
                   scores=S, znorm_mean=zm, znorm_std=zs)
        if d <= 16:
            out["X"] = x          # small cases carry the inputs; c1 is regenerated from its seed
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "psi[:3]", psi[:3], "scores", S.shape, float(S.min()), float(S.max()))


if __name__ == "__main__":
    main()
