"""plda_amd/dvector.py -- d-vector front-end on the GPU: counterpart of
/root/reference/scoring/extractdvector.py:19-59.  The per-utterance functions keep the
reference's names and signatures (they are the values of scorePLDA.py's `methods` dict,
:76-80); `pool` is the batched form: many utterances in one launch."""
import ctypes as C

import numpy as np

from . import _native as N

_METHODS = {"mean": 0, "max": 1, "var": 2}
_engine = None


def _eng(device=0):
    global _engine
    if _engine is None:
        from .libplda import MPlda
        _engine = MPlda(device)
    return _engine


def pool(frames, offsets, method="mean", l2norm=True, engine=None):
    """frames [T, D] float32/float64, offsets [U+1] -> ndarray float64 [U, D]."""
    eng = engine or _eng()
    F = np.ascontiguousarray(frames)
    if F.dtype not in (np.float32, np.float64):
        F = F.astype(np.float64)
    if F.ndim != 2:
        raise ValueError("frames must be (n_frames, featdim)")
    off = np.ascontiguousarray(offsets, np.int64)
    U = off.shape[0] - 1
    out = np.zeros((max(U, 0), F.shape[1]), np.float64)
    if U <= 0:
        return out
    rc = eng._lib.plda_dvector_pool(eng._h, C.c_void_p(F.ctypes.data), 0 if F.dtype == np.float32 else 1,
                                    F.shape[0], F.shape[1], C.c_void_p(off.ctypes.data), U, _METHODS[method],
                                    1 if l2norm else 0, C.c_void_p(out.ctypes.data))
    N.check(eng._h, rc)
    return out


def pool_utterances(utts, method="mean", l2norm=True, engine=None):
    """list of [T_i, D] arrays -> [len(utts), D]."""
    lens = [len(u) for u in utts]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    frames = np.concatenate([np.asarray(u) for u in utts], axis=0) if utts else np.zeros((0, 1))
    return pool(frames, off, method, l2norm, engine)


def _one(utt, method, l2norm):
    u = np.asarray(utt)
    return pool(u, np.array([0, u.shape[0]], np.int64), method, l2norm)[0]


def extractdvectormean(utt):      # extractdvector.py:37-39
    return _one(utt, "mean", True)


def extractdvectormax(utt):       # :32-34
    return _one(utt, "max", True)


def extractdvectorvar(utt):       # :42-47
    return _one(utt, "var", True)
