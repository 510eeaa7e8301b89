"""Randomised parity sweep, part 2: z-norm statistics, z-normalised trial matrices and lists, d-vector pooling,
EER of a labelled matrix, HTK decoding -- GPU vs the oracles over many seeded shapes (hunting tool)."""
import os
import sys
import tempfile
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_data, score_tol          # noqa: E402
from oracle import binding as ob, htk_oracle_np as ho, plda_oracle_np as onp   # noqa: E402
from plda_amd import MPlda, htk                     # noqa: E402
from plda_amd.dvector import pool                   # noqa: E402
from plda_amd import eer as geer                    # noqa: E402
import torch                                        # noqa: E402

ob.build()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # second argument: offset of every seed (fresh cases)
fails = []
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()

for case in range(ncases):
    rng = np.random.default_rng(7000 + case + seed0)
    tag = "case %d" % case
    try:
        # ---- z-norm ----
        d = int(rng.choice([2, 5, 16, 40, 64, 150, 200, 256]))
        k = int(rng.integers(3, 30))
        n = int(rng.integers(4 * k, 30 * k))
        x, y = make_data(9000 + case + seed0, n, d, k, skew=True, scale_between=float(rng.choice([0.1, 0.5, 1.0])))
        eng = MPlda(0)
        eng.fit(x, y, 3)
        m = eng.get_model()
        model = dict(mean=m["mean"], transform=m["transform"], psi=m["psi"], offset=m["offset"])
        nm = int(rng.integers(1, 60))
        nb = int(rng.integers(2, 300))
        nm = min(nm, n)
        models = eng.transform(x[:nm], np.arange(nm, dtype=np.uint64))
        bkg = x[rng.integers(0, n, nb)]
        eng.norm(bkg, models)
        mv = np.stack([models[i][1] for i in sorted(models)])
        rm, rs = ob.norm(model, bkg, mv)
        gm = np.array([eng._meanz[i] for i in sorted(models)])
        gs = np.array([eng._stdvz[i] for i in sorted(models)])
        if not (np.abs(gm - rm).max() <= 2e-4 * max(np.abs(rm).max(), 1e-30) and np.abs(gs - rs).max() <= 2e-4 * max(rs.max(), 1e-30)):
            fails.append(tag + " znorm D=%d nm=%d nb=%d: mean %.2e std %.2e (rel)" % (
                d, nm, nb, np.abs(gm - rm).max() / max(np.abs(rm).max(), 1e-30), np.abs(gs - rs).max() / max(rs.max(), 1e-30)))
        # z-normalised matrix and trial list vs the oracle using the GPU's own statistics
        nt = int(rng.integers(1, 90))
        nt = min(nt, n)
        tests = eng.transform(x[-nt:], np.arange(nt, dtype=np.uint64))
        tv = np.stack([tests[i][1] for i in sorted(tests)])
        S = eng.score_matrix(models, tests)
        Sr = ob.score_block(m["psi"], mv, np.ones(nm, np.int32), tv, gm, gs)
        if (np.abs(S - Sr) > score_tol(Sr)).any():
            fails.append(tag + " znormed matrix max err %.3e" % np.abs(S - Sr).max())
        P = int(rng.integers(1, 50))
        e_idx, t_idx = rng.integers(0, nm, P), rng.integers(0, nt, P)
        L = eng.score_trials(models, tests, e_idx, t_idx)
        if not np.allclose(L, Sr[e_idx, t_idx], rtol=1e-9, atol=1e-9):
            fails.append(tag + " trial list max err %.3e" % np.abs(L - Sr[e_idx, t_idx]).max())
        # ---- d-vector pooling ----
        dd = int(rng.choice([1, 3, 16, 32, 40, 64, 100, 128, 256, 400]))
        U = int(rng.integers(1, 40))
        lens = rng.integers(1, 80, U)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        fr = (rng.standard_normal((off[-1], dd)) * 3).astype(np.float32 if case % 2 else np.float64)
        for method in ("mean", "max", "var"):
            for l2 in (True, False):
                got = pool(fr, off, method, l2norm=l2, engine=eng)
                want = onp.dvector_pool(fr, off, method, l2norm=l2)
                if not np.allclose(got, want, rtol=1e-10, atol=1e-12):
                    fails.append(tag + " dvector D=%d %s l2=%s err %.2e" % (dd, method, l2, np.abs(got - want).max()))
        # ---- EER of a labelled matrix (quantised scores -> ties) ----
        Me, Ne = int(rng.integers(2, 70)), int(rng.integers(2, 300))
        es, ts = rng.integers(0, 6, Me), rng.integers(0, 6, Ne)
        sc = (rng.standard_normal((Me, Ne)) + 1.5 * (es[:, None] == ts[None, :])).astype(np.float32)
        if case % 3 == 0:
            sc = np.round(sc, 1).astype(np.float32)
        tgt = es[:, None] == ts[None, :]
        if tgt.any() and (~tgt).any():
            dS = torch.from_numpy(sc).to(dev)
            des, dts = torch.from_numpy(es).to(dev), torch.from_numpy(ts).to(dev)     # keep them alive over the call
            out = geer.eer_from_matrix_dev(eng, dS.data_ptr(), Ne, Me, Ne, des.data_ptr(), dts.data_ptr())
            ref = onp.eer(sc[~tgt], sc[tgt])
            if not (tuple(out[1:4]) == ref[1:] and abs(out[0] - ref[0]) <= 1e-12 * max(1.0, abs(ref[0]))):
                fails.append(tag + " eer %s vs %s" % (tuple(out[:4]), ref))
        # ---- HTK ----
        paths, raws = [], []
        hd = int(rng.choice([1, 4, 13, 40, 64]))
        for u in range(int(rng.integers(1, 12))):
            p = os.path.join(tmp, "c%d_%d.htk" % (case, u))
            ho.write_htk(p, rng.standard_normal((int(rng.integers(0, 50)), hd)).astype(np.float32))
            paths.append(p); raws.append(open(p, "rb").read())
        F = int(rng.integers(0, 4))
        frames, foff = htk.htk_load_batch(paths, F, engine=eng)
        want = np.concatenate([ho.htk_load(r, F) for r in raws])
        if not np.array_equal(frames.view(np.uint32), want):
            fails.append(tag + " htk dim=%d F=%d" % (hd, F))
    except Exception as ex:
        fails.append(tag + " -> EXC " + repr(ex)[:200] + " | " + " / ".join(l.strip() for l in traceback.format_exc().splitlines()[-4:-1])[:300])
print("%d cases, %d failures" % (ncases, len(fails)))
for f in fails:
    print(" ", f)
