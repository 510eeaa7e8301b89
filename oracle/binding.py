"""oracle/binding.py -- ctypes access to oracle/libplda_oracle.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by plda_amd/ or liblda/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libplda_oracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)


def build(force=False):
    src = os.path.join(_HERE, "plda_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libplda_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.plda_oracle_transform_ivector.restype = C.c_double
        _lib.plda_oracle_llr.restype = C.c_double
        _lib.plda_oracle_objective.restype = C.c_double
        _lib.plda_oracle_smooth.restype = None
        _lib.plda_oracle_score_block.restype = None
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def fit(X, labels, iters=10):
    X = _f64(X)
    labels = np.ascontiguousarray(labels, dtype=np.uint64)
    N, D = X.shape
    mean, psi, offset = np.zeros(D), np.zeros(D), np.zeros(D)
    T, W, B = np.zeros((D, D)), np.zeros((D, D)), np.zeros((D, D))
    rc = lib().plda_oracle_fit(_d(X), C.c_int64(N), C.c_int(D), labels.ctypes.data_as(_u64p),
                               C.c_int(iters), _d(mean), _d(T), _d(psi), _d(offset), _d(W), _d(B))
    if rc == -2:
        raise ValueError("Number of speakers is 1. Aborting PLDA esimation, at least two speakers are required!")
    if rc != 0:
        raise RuntimeError("plda_oracle_fit failed rc=%d" % rc)
    return dict(mean=mean, transform=T, psi=psi, offset=offset, W=W, B=B)


def stats(X, labels):
    X = _f64(X)
    labels = np.ascontiguousarray(labels, dtype=np.uint64)
    N, D = X.shape
    K = int(labels.max()) + 1
    means, counts = np.zeros((K, D)), np.zeros(K, np.int64)
    scatter, sum_ = np.zeros((D, D)), np.zeros(D)
    cw, ew = C.c_double(), C.c_double()
    rc = lib().plda_oracle_stats(_d(X), C.c_int64(N), C.c_int(D), labels.ctypes.data_as(_u64p),
                                 C.c_int64(K), _d(means), counts.ctypes.data_as(_i64p), _d(scatter),
                                 _d(sum_), C.byref(cw), C.byref(ew))
    if rc != 0:
        raise RuntimeError("plda_oracle_stats rc=%d" % rc)
    return dict(means=means, counts=counts, scatter=scatter, sum=sum_,
                class_weight=cw.value, example_weight=ew.value)


def em_iter(st, W, B):
    order = np.argsort(st["counts"], kind="stable")
    means = _f64(st["means"][order])
    counts = np.ascontiguousarray(st["counts"][order], np.int64)
    W, B = _f64(W).copy(), _f64(B).copy()
    K, D = means.shape
    rc = lib().plda_oracle_em_iter(_d(means), counts.ctypes.data_as(_i64p), C.c_int64(K), C.c_int(D),
                                   _d(_f64(st["scatter"])), _d(_f64(st["sum"])),
                                   C.c_double(st["class_weight"]), C.c_double(st["example_weight"]),
                                   _d(W), _d(B))
    if rc != 0:
        raise RuntimeError("plda_oracle_em_iter rc=%d" % rc)
    return W, B


def objective(st, W, B):
    means = _f64(st["means"])
    counts = np.ascontiguousarray(st["counts"], np.int64)
    K, D = means.shape
    return lib().plda_oracle_objective(_d(means), counts.ctypes.data_as(_i64p), C.c_int64(K), C.c_int(D),
                                       _d(_f64(st["scatter"])), _d(_f64(st["sum"])),
                                       C.c_double(st["class_weight"]), C.c_double(st["example_weight"]),
                                       _d(_f64(W)), _d(_f64(B)))


def get_output(st, W, B):
    D = W.shape[0]
    mean, psi, offset, T = np.zeros(D), np.zeros(D), np.zeros(D), np.zeros((D, D))
    rc = lib().plda_oracle_get_output(_d(_f64(W)), _d(_f64(B)), _d(_f64(st["sum"])),
                                      C.c_double(st["class_weight"]), C.c_int(D),
                                      _d(mean), _d(T), _d(psi), _d(offset))
    if rc != 0:
        raise RuntimeError("plda_oracle_get_output rc=%d" % rc)
    return dict(mean=mean, transform=T, psi=psi, offset=offset)


def transform_ivector(model, x, n, normalize_length=True, simple_length_norm=False):
    x = _f64(x)
    if x.ndim != 1:
        raise ValueError("transform_ivector takes ONE vector [D]; use plda_oracle_np.transform_ivector for batches")
    D = x.shape[0]
    out = np.zeros(D)
    lib().plda_oracle_transform_ivector(_d(_f64(model["transform"])), _d(_f64(model["offset"])),
                                        _d(_f64(model["psi"])), C.c_int(D), _d(x), C.c_int(int(n)),
                                        C.c_int(int(normalize_length)), C.c_int(int(simple_length_norm)),
                                        _d(out))
    return out


def transform_groups(model, X, labels, smoothfactor=1.0):
    X = _f64(X)
    labels = np.ascontiguousarray(labels, dtype=np.uint64)
    N, D = X.shape
    T, off, psi = _f64(model["transform"]).copy(), _f64(model["offset"]).copy(), _f64(model["psi"]).copy()
    ol, oc, ov = np.zeros(N, np.uint64), np.zeros(N, np.int64), np.zeros((N, D))
    ku = C.c_int64(N)
    rc = lib().plda_oracle_transform_groups(_d(T), _d(off), _d(psi), _d(_f64(model["mean"])), C.c_int(D),
                                            _d(X), C.c_int64(N), labels.ctypes.data_as(_u64p),
                                            C.c_double(smoothfactor), ol.ctypes.data_as(_u64p),
                                            oc.ctypes.data_as(_i64p), _d(ov), C.byref(ku))
    if rc != 0:
        raise RuntimeError("plda_oracle_transform_groups rc=%d" % rc)
    k = ku.value
    return ol[:k].copy(), oc[:k].copy(), ov[:k].copy()


def llr(psi, u, n, v):
    psi, u, v = _f64(psi), _f64(u), _f64(v)
    return lib().plda_oracle_llr(_d(psi), C.c_int(len(psi)), _d(u), C.c_int(int(n)), _d(v))


def score_block(psi, U, n_enrol, V, zmean=None, zstd=None):
    psi, U, V = _f64(psi), _f64(U), _f64(V)
    M, D = U.shape
    Nt = V.shape[0]
    n_enrol = np.ascontiguousarray(np.broadcast_to(np.asarray(n_enrol), (M,)), np.int32)
    out = np.zeros((M, Nt))
    zm = _d(_f64(zmean)) if zmean is not None else None
    zs = _d(_f64(zstd)) if zstd is not None else None
    lib().plda_oracle_score_block(_d(psi), C.c_int(D), _d(U), n_enrol.ctypes.data_as(_i32p), C.c_int64(M),
                                  _d(V), C.c_int64(Nt), zm, zs, _d(out))
    return out


def norm(model, bkg, models):
    bkg, models = _f64(bkg), _f64(models)
    Nb, D = bkg.shape
    M = models.shape[0]
    om, os_ = np.zeros(M), np.zeros(M)
    rc = lib().plda_oracle_norm(_d(_f64(model["transform"])), _d(_f64(model["offset"])),
                                _d(_f64(model["psi"])), C.c_int(D), _d(bkg), C.c_int64(Nb),
                                _d(models), C.c_int64(M), _d(om), _d(os_))
    if rc != 0:
        raise RuntimeError("plda_oracle_norm rc=%d" % rc)
    return om, os_


def smooth(model, f):
    out = {k: _f64(v).copy() for k, v in model.items() if k in ("transform", "psi", "offset", "mean")}
    D = len(out["psi"])
    lib().plda_oracle_smooth(_d(out["transform"]), _d(out["psi"]), _d(out["offset"]), _d(out["mean"]),
                             C.c_int(D), C.c_double(f))
    return out


def sym_eig(A):
    A = _f64(A).copy()
    D = A.shape[0]
    s, U = np.zeros(D), np.zeros((D, D))
    rc = lib().plda_oracle_sym_eig(_d(A), C.c_int(D), _d(s), _d(U))
    if rc != 0:
        raise RuntimeError("plda_oracle_sym_eig rc=%d" % rc)
    return s, U


def cholesky(A):
    A = _f64(A).copy()
    rc = lib().plda_oracle_cholesky(_d(A), C.c_int(A.shape[0]))
    if rc != 0:
        raise RuntimeError("not positive definite")
    return A


def invert(A):
    A = _f64(A).copy()
    rc = lib().plda_oracle_invert(_d(A), C.c_int(A.shape[0]))
    if rc != 0:
        raise RuntimeError("singular")
    return A
