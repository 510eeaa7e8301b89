"""Randomised parity sweep of the trials matrix with MIXED enrol counts (bucketed form, depth-2D fallback, bf16x3 arm):
random dimensions, shapes, count sets (few / many / large / repeated values), z-norm on or off, every GEMM kernel, the host
entry, the device entry and the sharded entry -- against the per-trial fp64 oracle.  Hunting tool run by hand on the GPU box:
python scripts/stress_mixed_counts.py [n_cases] [seed_offset]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import score_tol                      # noqa: E402
from oracle import binding as ob                    # noqa: E402
import torch                                        # noqa: E402

ob.build()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
fails = []
forms = {}
for case in range(ncases):
    rng = np.random.default_rng(31000 + case + seed0)
    d = int(rng.choice([3, 8, 9, 16, 24, 40, 57, 64, 100, 130, 200, 256]))
    m = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 300, 700, 1300]))
    nt = int(rng.choice([1, 5, 127, 128, 129, 255, 256, 257, 513, 900, 2100]))
    kind = int(rng.integers(0, 6))
    if kind == 0:
        vals = rng.choice(np.arange(1, 8), size=int(rng.integers(1, 6)), replace=False)
    elif kind == 1:
        vals = rng.choice(np.arange(1, 200), size=int(rng.integers(2, 30)), replace=False)
    elif kind == 2:
        vals = rng.choice(np.arange(1, 5000), size=int(rng.integers(2, 9)), replace=False)       # may cross 4095 -> fallback
    elif kind == 3:
        vals = np.arange(1, int(rng.integers(60, 80)))                                           # many: > 64 or > D / 2 -> fallback
    elif kind == 4:
        vals = np.array([int(rng.integers(1, 50))])                                              # one value in an ARRAY -> uniform path
    else:
        vals = rng.choice(np.arange(1, 12), size=int(rng.integers(2, 10)), replace=False)
    counts = vals[rng.integers(0, len(vals), m)].astype(np.int32)
    variant = str(rng.choice(["", "20", "30", "40"]))
    dtype = str(rng.choice(["", "", "bf16x3"]))
    mixed_variant = str(rng.choice(["", "", "", "1"]))
    znorm = bool(rng.integers(0, 2))
    entry = str(rng.choice(["host", "dev", "sharded"]))
    tag = "case %d: D=%d M=%d Nt=%d kind=%d G=%d variant=%r dtype=%r mixed_variant=%r znorm=%s entry=%s" % (
        case, d, m, nt, kind, len(np.unique(counts)), variant, dtype, mixed_variant, znorm, entry)
    try:
        for k, v in (("PLDA_GEMM_VARIANT", variant), ("PLDA_SCORE_DTYPE", dtype), ("PLDA_MIXED_VARIANT", mixed_variant)):
            if v:
                os.environ[k] = v
            else:
                os.environ.pop(k, None)
        from plda_amd import MPlda
        eng = MPlda(0)
        q, _ = np.linalg.qr(rng.standard_normal((d, d)))
        psi = np.sort(rng.random(d) * 4.0 + 0.05)[::-1].copy()
        eng.set_model(rng.random(d), q * (1.0 + rng.random(d))[:, None], psi)
        U, V = rng.standard_normal((m, d)), rng.standard_normal((nt, d))
        zm = zs = None
        if znorm:
            raw = ob.score_block(psi, U, counts, V)
            zm, zs = raw.mean(1), raw.std(1) + 0.1
        ref = ob.score_block(psi, U, counts, V, zm, zs) if znorm else ob.score_block(psi, U, counts, V)
        if entry == "host":
            ids = np.arange(m, dtype=np.int64)
            if znorm:
                eng._meanz = {int(k): float(v) for k, v in zip(ids, zm)}
                eng._stdvz = {int(k): float(v) for k, v in zip(ids, zs)}
            got = eng.score_matrix((counts, U, ids), (1, V))
        else:
            st = torch.cuda.Stream(device=dev)
            eng.set_stream(st.cuda_stream)
            dU, dV, dn = torch.from_numpy(U).to(dev), torch.from_numpy(V).to(dev), torch.from_numpy(counts).to(dev)
            dzm = torch.from_numpy(zm).to(dev) if znorm else None
            dzs = torch.from_numpy(zs).to(dev) if znorm else None
            out = torch.full((m, nt), float("nan"), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            kw = dict(dzmean=dzm.data_ptr(), dzstd=dzs.data_ptr()) if znorm else {}
            if entry == "dev":
                eng.score_matrix_dev(dU.data_ptr(), dn.data_ptr(), 0, m, dV.data_ptr(), nt, out.data_ptr(), nt, **kw)
            else:
                R = int(rng.integers(1, 4))
                for r in range(R):
                    eng.comm_emulate(R, r)
                    eng.score_matrix_sharded_dev(dU.data_ptr(), dn.data_ptr(), 0, m, dV.data_ptr(), nt, out.data_ptr(), nt, block_rows=256, **kw)
                eng.comm_emulate(1, 0)
            eng.synchronize()
            got = out.cpu().numpy()
            eng.set_stream(None)
        depth = eng.score_last_shape()[2]
        forms[depth - d if depth != 2 * d else "2D"] = forms.get(depth - d if depth != 2 * d else "2D", 0) + 1
        err = np.abs(got - ref)
        tol = score_tol(ref)
        if znorm:      # a z-score is the raw score's error times 1 / zstd: with one test column the z-scores are all 0 and `score_tol` would be 0
            tol = np.maximum(tol, 1e-4 * np.abs(raw).mean() / zs[:, None])
        if not (np.isfinite(got).all() and (err <= tol).all()):
            fails.append((tag, float(err.max()), float((err / tol).max())))
            print("FAIL", tag, err.max(), flush=True)
    except Exception as e:      # noqa: BLE001
        fails.append((tag, repr(e)))
        print("ERROR", tag, repr(e), flush=True)
        traceback.print_exc()
print("stress_mixed_counts: %d cases, %d failures; extra depth histogram (depth - D, or 2D): %s" % (ncases, len(fails), forms))
for f in fails:
    print("  ", f)
sys.exit(1 if fails else 0)
