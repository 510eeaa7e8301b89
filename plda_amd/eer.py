"""plda_amd/eer.py -- equal error rate on the GPU: counterpart of
/root/reference/scoring/eer.py:68-73 (bob.measure.eer_threshold + farfrr)."""
import ctypes as C

import numpy as np

from . import _native as N


def eer_from_lists(engine, truescores, impostscores):
    """(threshold, FAR, FRR, EER) from target / impostor score arrays (eer.py's two files)."""
    pos = np.ascontiguousarray(truescores, np.float32)
    neg = np.ascontiguousarray(impostscores, np.float32)
    out = np.zeros(6)
    N.check(engine._h, engine._lib.plda_eer_lists(engine._h, C.c_void_p(pos.ctypes.data), pos.shape[0],
                                                  C.c_void_p(neg.ctypes.data), neg.shape[0], C.c_void_p(out.ctypes.data)))
    return tuple(out[:4])


def eer_from_matrix_dev(engine, dscores, ld, m, nt, denrol_spk, dtest_spk):
    """Same on an HBM-resident fp32 trials matrix; trial (i, j) is a target iff
    enrol_spk[i] == test_spk[j] (int64 device arrays).  Returns the 6-vector
    (threshold, FAR, FRR, EER, #targets, #impostors)."""
    out = np.zeros(6)
    N.check(engine._h, engine._lib.plda_eer_matrix_dev(engine._h, C.c_void_p(int(dscores)), int(ld), int(m), int(nt),
                                                       C.c_void_p(int(denrol_spk)), C.c_void_p(int(dtest_spk)),
                                                       C.c_void_p(out.ctypes.data)))
    return out


def eer_from_operands_dev(engine, dU, dn, n_uniform, m, dV, nt, denrol_spk, dtest_spk, dzmean=None, dzstd=None):
    """The EER of the m x nt trials between HBM-resident transformed enrol / test vectors without the score matrix
    (`plda_score_eer_dev`): arguments as `MPlda.score_matrix_dev` plus the int64 speaker ids of both sides.  Returns the
    6-vector of `eer_from_matrix_dev`, identical to scoring the matrix first."""
    out = np.zeros(6)
    p = lambda x: C.c_void_p(int(x)) if x else None      # noqa: E731
    N.check(engine._h, engine._lib.plda_score_eer_dev(engine._h, p(dU), p(dn), int(n_uniform), int(m), p(dV), int(nt),
                                                      p(dzmean), p(dzstd), p(denrol_spk), p(dtest_spk),
                                                      C.c_void_p(out.ctypes.data)))
    return out


def det_from_lists(engine, truescores, impostscores, n_points=100):
    """(thresholds, FAR, FRR) at the n_points thresholds of the DET curve eer.py:34-62 plots (`bob.measure.plot.det(neg, pos,
    100)`): arrays of length n_points.  The plot's coordinates are `ppndf(FRR)`, `ppndf(FAR)`."""
    pos = np.ascontiguousarray(truescores, np.float32)
    neg = np.ascontiguousarray(impostscores, np.float32)
    far, frr, thr = np.zeros(n_points), np.zeros(n_points), np.zeros(n_points)
    N.check(engine._h, engine._lib.plda_det_lists(engine._h, C.c_void_p(pos.ctypes.data), pos.shape[0], C.c_void_p(neg.ctypes.data),
                                                  neg.shape[0], int(n_points), C.c_void_p(far.ctypes.data),
                                                  C.c_void_p(frr.ctypes.data), C.c_void_p(thr.ctypes.data)))
    return thr, far, frr


def det_from_matrix_dev(engine, dscores, ld, m, nt, denrol_spk, dtest_spk, n_points=100):
    """Same on an HBM-resident fp32 trials matrix with int64 speaker ids on the device."""
    far, frr, thr = np.zeros(n_points), np.zeros(n_points), np.zeros(n_points)
    N.check(engine._h, engine._lib.plda_det_matrix_dev(engine._h, C.c_void_p(int(dscores)), int(ld), int(m), int(nt),
                                                       C.c_void_p(int(denrol_spk)), C.c_void_p(int(dtest_spk)), int(n_points),
                                                       C.c_void_p(far.ctypes.data), C.c_void_p(frr.ctypes.data),
                                                       C.c_void_p(thr.ctypes.data)))
    return thr, far, frr


def ppndf(p):
    """The normal deviate of a rate, the scale of a DET plot's axes (bob.measure.ppndf: probit of p clipped to
    [2.2204e-16, 1 - 2.2204e-16]; restated)."""
    from scipy.special import ndtri
    eps = 2.2204e-16
    return ndtri(np.clip(np.asarray(p, np.float64), eps, 1.0 - eps))


def format_line(far, frr, threshold):
    """The line eer.py:72-73 writes."""
    return "EER = %.2f%%, FAR = %.2f, FRR=%.2f, Threshold = %e\n" % ((far + frr) / 2 * 100, far, frr, threshold)
