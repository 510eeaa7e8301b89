"""Randomised parity sweep of the trials matrix at the shapes the small sweeps do not reach (scripts/stress_mixed_counts.py stops
at 1300 x 2100): sides on both sides of the 32 768-row switch of the packing kernels, count arrays on both sides of the
131 072-entry switch of the count-set kernel, tile counts across the 512 / 1 700-tile switches of the GEMM dispatch, uniform and
mixed enrol counts, z-norm on or off -- the product's own dispatch (no variant forced), device entry, checked against the
per-trial fp64 oracle on a random 160 x 240 sub-block (rows and columns drawn from the whole matrix, its last row and column
included) and for NaN / untouched elements over the whole output.
python scripts/stress_large_shapes.py [n_cases] [seed_offset]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import score_tol                      # noqa: E402
from oracle import binding as ob                    # noqa: E402
import torch                                        # noqa: E402
from plda_amd import MPlda                          # noqa: E402

ob.build()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
fails = 0
kernels = {}
for case in range(ncases):
    rng = np.random.default_rng(52000 + case + seed0)
    d = int(rng.choice([24, 64, 72, 200, 256]))
    m = int(rng.choice([300, 5000, 8192, 20000, 32768, 33000, 40000, 70000, 131072, 140000]))
    nt = int(rng.choice([257, 2000, 9000, 33000, 50000]))
    while m * nt > 3.0e9:
        nt //= 2
    if rng.random() < 0.3:
        m, nt = nt, m
    kind = str(rng.choice(["uniform", "few", "many", "wide"]))
    if kind == "uniform":
        counts = None; n_uniform = int(rng.integers(1, 9))
    else:
        vals = {"few": np.arange(1, 6), "many": rng.choice(np.arange(1, 60), size=14, replace=False),
                "wide": np.array([1, 7, 300, 4095])}[kind]
        counts = vals[rng.integers(0, len(vals), m)].astype(np.int32); n_uniform = 0
    znorm = bool(rng.integers(0, 2))
    tag = "case %d: D=%d M=%d Nt=%d counts=%s znorm=%s" % (case, d, m, nt, kind, znorm)
    eng = MPlda(0)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    psi = np.sort(rng.random(d) * 4.0 + 0.05)[::-1].copy()
    eng.set_model(rng.random(d), q * (1.0 + rng.random(d))[:, None], psi)
    st = torch.cuda.Stream(device=dev)
    eng.set_stream(st.cuda_stream)
    g = torch.Generator(device=dev); g.manual_seed(case + seed0)
    dU = torch.randn((m, d), dtype=torch.float64, device=dev, generator=g)
    dV = torch.randn((nt, d), dtype=torch.float64, device=dev, generator=g)
    dn = torch.from_numpy(counts).to(dev) if counts is not None else None
    zm = zs = dzm = dzs = None
    if znorm:
        zm, zs = rng.standard_normal(m) * 3.0 - 20.0, rng.random(m) * 2.0 + 0.5
        dzm, dzs = torch.from_numpy(zm).to(dev), torch.from_numpy(zs).to(dev)
    out = torch.full((m, nt), float("nan"), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for rep in range(2):                        # (twice: the second call runs on the coefficients / tables / counters the first left)
        eng.score_matrix_dev(dU.data_ptr(), dn.data_ptr() if dn is not None else None, n_uniform, m, dV.data_ptr(), nt,
                             out.data_ptr(), nt, dzm.data_ptr() if znorm else None, dzs.data_ptr() if znorm else None)
    eng.synchronize()
    kern = eng.score_last_kernel()
    kernels[kern] = kernels.get(kern, 0) + 1
    rows = np.unique(np.concatenate([rng.integers(0, m, 158), [0, m - 1]]))
    cols = np.unique(np.concatenate([rng.integers(0, nt, 238), [0, nt - 1]]))
    tr, tc = torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev)
    U, V = dU[tr].cpu().numpy(), dV[tc].cpu().numpy()
    cn = counts[rows] if counts is not None else n_uniform
    ref = ob.score_block(psi, U, cn, V, zm[rows], zs[rows]) if znorm else ob.score_block(psi, U, cn, V)
    got = out[tr][:, tc].cpu().numpy().astype(np.float64)
    bad_nan = int(torch.isnan(out).sum().item())
    err = np.abs(got - ref)
    if znorm:                                   # the raw score's tolerance, mapped like the score: / zstd_i
        raw = ob.score_block(psi, U, cn, V)
        tol = np.maximum(score_tol(ref), score_tol(raw) / zs[rows][:, None])
    else:
        tol = score_tol(ref)
    ok = bad_nan == 0 and bool((err <= tol).all())
    if not ok:
        fails += 1
        print("FAIL", tag, kern, "nan:", bad_nan, "max err %.3g (tol %.3g)" % (err.max(), float(np.min(tol))))
    del out, dU, dV
print("stress_large_shapes: %d cases, %d failures; kernels %s" % (ncases, fails, kernels))
sys.exit(1 if fails else 0)
