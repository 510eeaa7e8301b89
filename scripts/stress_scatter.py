"""Statistics pass (K2, the block scatter kernel for 208 < D <= 512) run many times on fresh handles with the device's free memory
dirtied in between: every scatter must equal the first one bit for bit (a timing- or memory-state-dependent result is a bug).
usage: python scripts/stress_scatter.py [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plda_amd import MPlda

dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
side = torch.cuda.Stream(device=dev)
big = torch.rand((int(sys.argv[2]) if len(sys.argv) > 2 else 3000,) * 2, dtype=torch.float32, device=dev)
bad = 0
shapes = [(512, 6000, 100), (256, 5000, 70), (384, 4097, 64), (512, 700, 9), (300, 2500, 30)]
if os.environ.get("STRESS_SHAPES"):
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["STRESS_SHAPES"].split(",")]
for (d, n, k) in shapes:
    rng = np.random.default_rng(d + n)
    y = rng.integers(0, k, n); y[:k] = np.arange(k)
    x = rng.random((n, d)) + 0.5 * rng.standard_normal((k, d))[y]
    dX = torch.from_numpy(x).to(dev); dy = torch.from_numpy(y.astype(np.int64)).to(dev)
    first = None
    for r in range(reps):
        # dirty the allocators' free lists: NaN-filled torch blocks of varying size, released to the driver; and another handle
        # whose own buffers (hipMalloc) held a different problem, destroyed just before
        junk = [torch.full((int(rng.integers(1, 64)) << 18,), float("nan"), dtype=torch.float64, device=dev) for _ in range(3)]
        del junk
        torch.cuda.empty_cache()
        other = MPlda(0)
        n2, d2, k2 = int(rng.integers(500, 20000)), int(rng.choice([200, 256, 384, 512])), int(rng.integers(5, 300))
        y2 = rng.integers(0, k2, n2); y2[:k2] = np.arange(k2)
        x2 = torch.from_numpy(1e6 * rng.standard_normal((n2, d2))).to(dev); dy2 = torch.from_numpy(y2.astype(np.int64)).to(dev)
        other.fit_stats_dev(x2.data_ptr(), n2, d2, dy2.data_ptr(), k2)
        other.synchronize()
        del other, x2, dy2
        eng = MPlda(0)
        eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        if r % 2 == 1:      # every other run: the chip is busy with something else on another stream while the pass runs
            with torch.cuda.stream(side):
                for _ in range(int(rng.integers(1, 6))):
                    big2 = big @ big
        eng.fit_stats_dev(dX.data_ptr(), n, d, dy.data_ptr(), k)
        means = torch.empty((k, d), dtype=torch.float64, device=dev)
        counts = torch.empty((k,), dtype=torch.int64, device=dev)
        S = torch.empty((d, d), dtype=torch.float64, device=dev)
        eng.fit_get_stats_dev(means.data_ptr(), counts.data_ptr(), S.data_ptr())
        torch.cuda.synchronize()
        s = S.cpu().numpy()
        if first is None:
            first = s
        elif not np.array_equal(first, s, equal_nan=True):
            bad += 1
            print("D=%d N=%d rep %d: differs from the first run, max rel %.3g, nan %d" % (d, n, r, np.nanmax(np.abs(s - first)) / np.abs(first).max(), int(np.isnan(s).sum())), flush=True)
            e = np.abs(s - first); nbk = (d + 63) // 64
            print("   blocks (row, col, max, count): %s" % [(a, b, float("%.3g" % e[64 * a:64 * a + 64, 64 * b:64 * b + 64].max()), int((e[64 * a:64 * a + 64, 64 * b:64 * b + 64] > 0).sum()))
                                                     for a in range(nbk) for b in range(a + 1) if e[64 * a:64 * a + 64, 64 * b:64 * b + 64].max() > 0][:40], flush=True)
            # which rows?  diff block (a, b) = sum_r c_r x_r[a-block] x_r[b-block]^T over a few rows r of X (weight error c_r) or of the
            # centroids: greedy projection
            blk = [(a, b) for a in range(nbk) for b in range(a + 1) if e[64 * a:64 * a + 64, 64 * b:64 * b + 64].max() > 0][0]
            Dm = (s - first)[64 * blk[0]:64 * blk[0] + 64, 64 * blk[1]:64 * blk[1] + 64].copy()
            cnt = np.bincount(y, minlength=k).astype(np.float64)
            mu = np.stack([x[y == c].mean(0) for c in range(k)])
            rows = np.concatenate([x, mu]); tag = ["x%d(w=%.4g)" % (i, 1.0 / cnt[y[i]]) for i in range(n)] + ["mean%d(w=-1)" % c for c in range(k)]
            Aa, Bb = rows[:, 64 * blk[0]:64 * blk[0] + 64], rows[:, 64 * blk[1]:64 * blk[1] + 64]
            for it in range(4):
                num = np.einsum("ri,ij,rj->r", Aa, Dm, Bb); den = (Aa * Aa).sum(1) * (Bb * Bb).sum(1)
                c = num / den; gain = c * c * den
                rbest = int(np.argmax(gain))
                Dm -= c[rbest] * np.outer(Aa[rbest], Bb[rbest])
                print("   row %s coefficient %.6g, residual %.3g" % (tag[rbest], c[rbest], np.abs(Dm).max()), flush=True)
                if np.abs(Dm).max() < 1e-9: break
            # is the difference a rank-few update (a few rows counted wrongly)?
            sv = np.linalg.svd(s - first, compute_uv=False)
            print("   singular values of the difference: %s" % np.array2string(sv[:8], precision=3), flush=True)
        eng.set_stream(None)
        del eng
    print("D=%d N=%d K=%d: %d runs" % (d, n, k, reps), flush=True)
print("mismatches:", bad)
