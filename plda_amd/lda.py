"""plda_amd/lda.py -- the reference's `LDA` class (python/liblda/lda.py:87-338) on the MI355X engine.

Same constructor, methods, argument meaning and error behaviour as the NumPy class; every
array operation runs in libplda_hip.so (csrc/lda.hip) through the C ABI of
include/plda_hip.h -- there is no NumPy fallback: without the library or a GPU the
constructor raises.  Attribute names follow the reference (`priors`, `_classes`, `_means`,
`_coef`, `_intercept`, `_scalings`, `_xbar`, `explained_variance_ratio_`).
"""
import ctypes as C

import numpy as np

from . import _native as N
from .libplda import MPlda, _ptr

_SOLVERS = {"svd": 0, "eigen": 1, "lsqr": 2}
_NOT_FITTED = "This LDA instance is not fitted yet"     # lda.py:259


class LDA(object):

    def __init__(self, solver="svd", priors=None, device=0, engine=None):
        """lda.py:89-100.  `engine`: share an existing plda_amd.MPlda handle (one GPU context)."""
        self.priors = priors
        self.solver = solver
        self._eng = engine if engine is not None else MPlda(device)
        self._lib = self._eng._lib
        self._h = self._eng._h
        self._fitted = False

    # ------------------------------------------------------------------ fit (lda.py:104-132)
    def fit(self, features, labels):
        X = np.ascontiguousarray(features, dtype=np.float64)
        if X.ndim != 2:
            raise ValueError("features must be (n_samples, feat_dim)")
        self._classes, inv = np.unique(np.asarray(labels), return_inverse=True)
        if inv.shape[0] != X.shape[0]:
            raise ValueError("features and labels differ in the number of samples")
        if self.solver not in _SOLVERS:
            return None                     # the reference falls through its if/elif chain (lda.py:127-132)
        pri = None
        if self.priors is not None:
            pri = np.ascontiguousarray(self.priors, dtype=np.float64)
            if pri.shape != (len(self._classes),):
                raise ValueError("priors must have one entry per class")
        dense = np.ascontiguousarray(inv.astype(np.uint64))
        rc = self._lib.plda_lda_fit(self._h, _ptr(X), X.shape[0], X.shape[1], _ptr(dense), _SOLVERS[self.solver],
                                    _ptr(pri) if pri is not None else None)
        if rc == N.PLDA_E_NUMERIC and self.solver == "eigen":
            # scipy.linalg.eigh(Sb, Sw) raises this in the reference (lda.py:157)
            raise np.linalg.LinAlgError(N.last_error(self._h))
        N.check(self._h, rc)
        self._pull()
        return None

    def _pull(self):
        k, d, r, s = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
        N.check(self._h, self._lib.plda_lda_dims(self._h, C.byref(k), C.byref(d), C.byref(r), C.byref(s)))
        k, d, r = k.value, d.value, r.value
        pri, means, coef, icpt = np.zeros(k), np.zeros((k, d)), np.zeros((k, d)), np.zeros(k)
        xbar = np.zeros(d) if self.solver == "svd" else None
        scal = np.zeros((d, r)) if self.solver != "lsqr" else None
        evr = np.zeros(d) if self.solver == "eigen" else None
        N.check(self._h, self._lib.plda_lda_get_model(
            self._h, _ptr(pri), _ptr(means), _ptr(xbar) if xbar is not None else None,
            _ptr(scal) if scal is not None else None, _ptr(coef), _ptr(icpt), _ptr(evr) if evr is not None else None))
        self.priors, self._means, self._coef, self._intercept = pri, means, coef, icpt
        if xbar is not None:
            self._xbar = xbar
        if scal is not None:
            self._scalings = scal
        if evr is not None:
            self.explained_variance_ratio_ = evr
        self._fitted = True

    # ------------------------------------------------------------------ predict
    def _predict(self, X, mode):
        if not self._fitted:
            raise ValueError(_NOT_FITTED)
        X = np.ascontiguousarray(X, dtype=np.float64)
        n_features = self._coef.shape[1]
        if X.shape[1] != n_features:
            raise ValueError("X has %d features per sample; expecting %d" % (X.shape[1], n_features))   # lda.py:264-266
        out = np.empty((X.shape[0], self._coef.shape[0]))
        N.check(self._h, self._lib.plda_lda_predict(self._h, _ptr(X), X.shape[0], X.shape[1], int(mode), _ptr(out)))
        return out

    def decision_function(self, X):
        """lda.py:242-270."""
        scores = self._predict(X, 0)
        return scores.ravel() if scores.shape[1] == 1 else scores

    def predict_proba(self, sample):
        """lda.py:272-294, including the two-class branch that stacks [1 - prob, prob] of the
        per-class logistic values (a [n, 4] array for two classes, as the reference returns)."""
        if self._fitted and len(self._classes) == 2:
            prob = self._predict(sample, 2)
            return np.column_stack([1 - prob, prob])
        return self._predict(sample, 3)

    def predict_log_proba(self, sample):
        """lda.py:296-314."""
        return self._predict(sample, 1)

    # ---- HBM-resident variants (raw device addresses, e.g. torch tensors' data_ptr()) ----
    def fit_dev(self, dX, n, d, dlabels, k, classes=None):
        """fit on device-resident X [n, d] fp64 and dense labels 0..k-1 (u64/non-negative i64)."""
        pri = None if self.priors is None else np.ascontiguousarray(self.priors, dtype=np.float64)
        rc = self._lib.plda_lda_fit_dev(self._h, C.c_void_p(int(dX)), int(n), int(d), C.c_void_p(int(dlabels)), int(k),
                                        _SOLVERS[self.solver], _ptr(pri) if pri is not None else None)
        if rc == N.PLDA_E_NUMERIC and self.solver == "eigen":
            raise np.linalg.LinAlgError(N.last_error(self._h))
        N.check(self._h, rc)
        self._classes = np.arange(k) if classes is None else np.asarray(classes)
        self._pull()

    def predict_dev(self, dX, n, mode, dout):
        """mode 0 decision, 1 log-proba, 2 logistic, 3 one-vs-rest proba; dout [n, K] fp64 on the device."""
        N.check(self._h, self._lib.plda_lda_predict_dev(self._h, C.c_void_p(int(dX)), int(n), int(mode),
                                                        C.c_void_p(int(dout))))

    # ------------------------------------------------------------------ transform (lda.py:317-338)
    def transform(self, X, n_components=None):
        if self.solver == "lsqr":
            raise NotImplementedError("transform not implemented for 'lsqr' "
                                      "solver (use 'svd' or 'eigen').")
        if not self._fitted:
            raise ValueError(_NOT_FITTED)
        X = np.ascontiguousarray(X, dtype=np.float64)
        ncols = self._scalings.shape[1]
        want = X.shape[1] if n_components is None else n_components
        ncomp = len(range(ncols)[:want])          # slice semantics of X_new[:, :n_components] (lda.py:338)
        out = np.empty((X.shape[0], ncomp))
        N.check(self._h, self._lib.plda_lda_transform(self._h, _ptr(X), X.shape[0], X.shape[1], ncomp, _ptr(out)))
        return out

    # ------------------------------------------------------------------ persistence (build extension)
    def save(self, path):
        if not self._fitted:
            raise ValueError(_NOT_FITTED)
        rec = dict(solver=np.array(self.solver), classes=self._classes, priors=self.priors, means=self._means,
                   coef=self._coef, intercept=self._intercept)
        for name in ("_xbar", "_scalings", "explained_variance_ratio_"):
            if hasattr(self, name):
                rec[name.strip("_")] = getattr(self, name)
        from .libplda import _npz_path
        np.savez(_npz_path(path), **rec)

    def load(self, path):
        from .libplda import _npz_path
        z = np.load(_npz_path(path, for_load=True), allow_pickle=False)
        self.solver = str(z["solver"])
        self._classes = z["classes"]
        coef = np.ascontiguousarray(z["coef"], np.float64)
        k, d = coef.shape
        scal = np.ascontiguousarray(z["scalings"], np.float64) if "scalings" in z else None
        xbar = np.ascontiguousarray(z["xbar"], np.float64) if "xbar" in z else None
        pri, means, icpt = (np.ascontiguousarray(z[n], np.float64) for n in ("priors", "means", "intercept"))
        N.check(self._h, self._lib.plda_lda_set_model(
            self._h, _SOLVERS[self.solver], k, d, 0 if scal is None else scal.shape[1], _ptr(pri), _ptr(means),
            _ptr(xbar) if xbar is not None else None, _ptr(scal) if scal is not None else None, _ptr(coef), _ptr(icpt)))
        self._pull()
        if "explained_variance_ratio" in z:
            self.explained_variance_ratio_ = z["explained_variance_ratio"]
        return self
