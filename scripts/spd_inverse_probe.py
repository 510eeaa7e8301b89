"""Residuals of the E-step's SPD inverse per kernel (PLDA_SWEEP_VARIANT: 0 block sweep on the matrix cores, 2 the
four-wave scalar sweep, 1 the 16-wave scalar sweep) against numpy.linalg.inv, over condition numbers."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


def spd(d, cond, seed):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    lam = np.exp(np.linspace(0.0, np.log(cond), d))
    a = (q * lam) @ q.T
    return 0.5 * (a + a.T)


def main():
    from plda_amd import MPlda
    engines = {}
    for v in ("0", "2", "1"):
        os.environ["PLDA_SWEEP_VARIANT"] = v
        engines[v] = MPlda(0)
    print("%5s %8s | %-32s | %-32s | %-32s | numpy" % ("D", "cond", "block sweep (res, err)", "scalar 4-wave", "scalar 16-wave"))
    for d in (65, 128, 200, 256):
        for cond in (1e2, 1e4, 1e6, 1e8, 1e10):
            a = spd(d, cond, d)
            want = np.linalg.inv(a)
            ld = np.linalg.inv(a.astype(np.longdouble).astype(np.float64))
            row = []
            for v in ("0", "2", "1"):
                x = engines[v].spd_inverse(a)
                row.append("%.2e %.2e" % (np.abs(x @ a - np.eye(d)).max(), np.abs(x - want).max() / np.abs(want).max()))
            print("%5d %8.0e | %-32s | %-32s | %-32s | %.2e" % (d, cond, row[0], row[1], row[2], np.abs(want @ a - np.eye(d)).max()))


if __name__ == "__main__":
    main()
