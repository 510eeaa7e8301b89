"""tests/golden/make_lda_golden.py -- golden vectors for the LDA row, produced by the REFERENCE itself.

Run in the build container only (needs /root/reference):  python tests/golden/make_lda_golden.py
It imports /root/reference/python/liblda/lda.py unmodified and records inputs and outputs as data
(tests/golden/lda_*.npz).  The one accommodation: lda.py:5 does `from scipy.misc import logsumexp`,
which SciPy moved to scipy.special; the same function is aliased back before the import.  Nothing
of the reference's text is stored -- only arrays.
"""
import importlib.util
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/python/liblda/lda.py"


def load_reference():
    warnings.simplefilter("ignore", DeprecationWarning)
    import scipy.misc
    import scipy.special
    scipy.misc.logsumexp = scipy.special.logsumexp
    spec = importlib.util.spec_from_file_location("reference_lda", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def cases():
    rng = np.random.default_rng(20260928)
    out = {}
    y = np.arange(60) % 5
    out["small_k5_d6"] = (rng.random((60, 6)) + 0.6 * rng.standard_normal((5, 6))[y], y, rng.random((9, 6)), None)
    y = rng.integers(0, 12, 400)
    out["k12_d8"] = (rng.random((400, 8)) + 0.4 * rng.standard_normal((12, 8))[y], y, rng.random((16, 8)), None)
    # tests/ldacomp.py:13-14 shape: 10 x 10 data, three speakers, N - K < D (rank deficient)
    out["ldacomp_shape"] = (rng.random((10, 10)), np.array([1, 1, 1, 2, 2, 2, 3, 3, 3, 3]), rng.random((10, 10)), None)
    y = (rng.random(300) < 0.35).astype(np.int64) * 7 + 3          # labels {3, 10}
    out["binary_priors"] = (rng.standard_normal((300, 5)) + 0.8 * (y[:, None] == 10), y, rng.standard_normal((11, 5)),
                            np.array([2.0, 1.0]))
    return out


def main():
    ref = load_reference()
    for name, (X, y, Xt, priors) in cases().items():
        rec = dict(X=X, y=y, Xt=Xt)
        if priors is not None:
            rec["priors_in"] = priors
        for solver in ("svd", "eigen", "lsqr"):
            lda = ref.LDA(solver, None if priors is None else priors.copy())
            try:
                lda.fit(X, y)
            except np.linalg.LinAlgError as ex:      # singular within-class covariance (eigen solver)
                rec[solver + "_error"] = np.array(type(ex).__name__)
                continue
            rec[solver + "_priors"] = np.asarray(lda.priors)
            rec[solver + "_coef"] = lda._coef
            rec[solver + "_intercept"] = lda._intercept
            rec[solver + "_decision"] = lda.decision_function(Xt)
            rec[solver + "_log_proba"] = lda.predict_log_proba(Xt)
            rec[solver + "_proba"] = lda.predict_proba(Xt)
            if solver != "lsqr":
                rec[solver + "_scalings"] = lda._scalings
            if solver == "svd":
                rec["svd_xbar"] = lda._xbar
            if solver == "eigen":
                rec["eigen_transform"] = lda.transform(Xt)
                rec["eigen_transform2"] = lda.transform(Xt, 2)
                rec["eigen_evr"] = lda.explained_variance_ratio_
        path = os.path.join(HERE, "lda_%s.npz" % name)
        np.savez_compressed(path, **rec)
        print("wrote", path, {k: v.shape for k, v in rec.items() if k.endswith("_coef")})


if __name__ == "__main__":
    sys.exit(main())
