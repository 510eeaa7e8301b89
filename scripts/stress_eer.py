"""Randomised sweep of the EER / DET consumers (csrc/eer.hip) against the NumPy restatement of bob.measure's definition
(oracle/plda_oracle_np.py: eer, det -- scoring/eer.py:27-39 of the reference calls bob.measure.eer_threshold / farfrr / det).

Per case: random matrix shape (1 .. 3000 x 1 .. 6000, ragged leading dimension), random speaker layout (dense, few targets,
one target, one non-target), random score law (Gaussian, rounded to a grid -> ties, a handful of distinct values, separable
classes, inverted classes, magnitudes at both ends of the fp32 range, signed zeros), and every form of the consumer on the SAME matrix:
  three passes (PLDA_EER_VARIANT=1), the single pass forced (2), the product's own choice (0), the lists entry point,
  the DET points of matrix and lists.
All must agree bit for bit with each other and with the restatement (threshold to 1e-12 relative: the restatement works in
fp64 on the fp32 scores).

usage: python scripts/stress_eer.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from oracle import plda_oracle_np as onp
    from plda_amd import MPlda, eer
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    engines = {}
    for v in ("1", "2", "0"):
        os.environ["PLDA_EER_VARIANT"] = v
        engines[v] = MPlda(0)
    del os.environ["PLDA_EER_VARIANT"]
    fails = ran = 0
    seen = {}
    for c in range(cases):
        m = int(rng.choice([1, 2, 17, 255, 256, 257, 700, 1500, 3000]))
        nt = int(rng.choice([1, 2, 63, 64, 65, 1000, 2100, 4099, 6000]))
        ld = nt + int(rng.choice([0, 0, 1, 5, 61]))
        layout = rng.choice(["dense", "few_targets", "one_target", "one_nontarget"])
        law = rng.choice(["gauss", "grid", "few_values", "separable", "inverted", "huge", "signed_zeros"])
        k = int(rng.integers(1, 40))
        es, ts = rng.integers(0, k, m), rng.integers(0, k, nt)
        if layout == "few_targets":
            es = np.arange(m) + 1000; ts = np.arange(nt) + 50000
            es[: max(1, m // 50)] = 7; ts[: max(1, nt // 80)] = 7
        elif layout == "one_target":
            es = np.arange(m) + 1000; ts = np.arange(nt) + 50000
            es[rng.integers(0, m)] = 7; ts[rng.integers(0, nt)] = 7
        elif layout == "one_nontarget":
            es = np.zeros(m, dtype=np.int64); ts = np.zeros(nt, dtype=np.int64)
            if m * nt > 1:
                if nt > 1:
                    ts[rng.integers(0, nt)] = 3
                else:
                    es[rng.integers(0, m)] = 3
        tgt = es[:, None] == ts[None, :]
        if tgt.all() or not tgt.any():
            continue                       # (one class empty: refused by both sides -- tests/test_gpu_eer.py)
        ran += 1
        seen[str(law)] = seen.get(str(law), 0) + 1
        Sh = rng.standard_normal((m, ld)).astype(np.float32)
        shift = {"separable": 9.0, "inverted": -2.0}.get(str(law), float(rng.choice([0.0, 0.5, 1.5, 3.0])))
        Sh[:, :nt] += np.float32(shift) * tgt
        if law == "separable":
            Sh[:, :nt] = np.clip(Sh[:, :nt], -3, 3) + np.float32(9.0) * tgt
        if law == "grid":
            Sh = np.round(Sh, int(rng.integers(0, 3))).astype(np.float32)
        if law == "few_values":
            Sh = np.round(Sh).astype(np.float32)
        if law == "huge":                  # magnitudes at both ends of the fp32 range, denormals included
            Sh[rng.random(Sh.shape) < 0.01] *= np.float32(1e30)
            Sh[rng.random(Sh.shape) < 0.01] *= np.float32(1e-42)
        if law == "signed_zeros":          # -0.0 and +0.0 are ONE score (the keys must not order them)
            Sh = np.round(Sh).astype(np.float32)
            Sh[Sh == 0] = np.where(rng.random((Sh == 0).sum()) < 0.5, np.float32(0.0), np.float32(-0.0))
        S = torch.from_numpy(Sh).to(dev)
        des, dts = torch.from_numpy(es).to(dev), torch.from_numpy(ts).to(dev)
        torch.cuda.synchronize()
        sub = Sh[:, :nt]
        neg, pos = sub[~tgt], sub[tgt]
        ref = onp.eer(neg, pos)
        tag = "case %d: %dx%d ld=%d layout=%s law=%s k=%d" % (c, m, nt, ld, layout, law, k)
        outs = {v: eer.eer_from_matrix_dev(e, S.data_ptr(), ld, m, nt, des.data_ptr(), dts.data_ptr()) for v, e in engines.items()}
        outs["lists"] = eer.eer_from_lists(engines["0"], pos, neg)
        bad = []
        for v, a in outs.items():
            a = np.asarray(a, dtype=np.float64)
            same_thr = a[0] == ref[0] or (np.isfinite(ref[0]) and abs(a[0] - ref[0]) <= 1e-12 * abs(ref[0]))
            if not (same_thr and tuple(a[1:4]) == tuple(ref[1:])):
                bad.append((v, a[:4].tolist(), list(ref)))
        npts = int(rng.choice([2, 33, 100, 2047]))
        dref = onp.det(neg, pos, npts)
        dm = eer.det_from_matrix_dev(engines["0"], S.data_ptr(), ld, m, nt, des.data_ptr(), dts.data_ptr(), npts)
        dl = eer.det_from_lists(engines["0"], pos, neg, npts)
        for name, d in (("det_matrix", dm), ("det_lists", dl)):
            for what, got, want in zip(("thr", "far", "frr"), d, dref):
                scale = max(1.0, float(np.abs(want).max())) if what == "thr" else 1.0
                if what == "thr":
                    ok = np.allclose(got, want, rtol=0, atol=1e-12 * scale)
                else:
                    ok = np.array_equal(got, want)
                if not ok:
                    bad.append((name, what, float(np.abs(np.asarray(got) - want).max())))
        if bad:
            fails += 1
            print("FAIL", tag, bad[:3])
    print("stress_eer: %d cases drawn, %d run (both classes present), %d failures (seed %d); by score law: %s" % (cases, ran, fails, seed, seen))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
