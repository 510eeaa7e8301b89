"""Per-wave timeline of workgroup 0 of the trials GEMM (PLDA_GEMM_VARIANT=31, plda_profile_timeline):
shader-clock stamps at every stage barrier (arrive / leave), at the start of the stage's steps and
around every tile epilogue.  Diagnostic only.  usage: gemm_timeline.py [N] [D]"""
import ctypes as C
import os
os.environ.setdefault("PLDA_LIB_DIAG", "1")      # measurement arms: the diagnostic build (python -m plda_amd.build --diag)
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PLDA_GEMM_VARIANT"] = os.environ.get("TL_VARIANT", "31")
import torch
from plda_amd import MPlda

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
D = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
q, _ = np.linalg.qr(rng.standard_normal((D, D)))
eng = MPlda(0)
eng.set_model(rng.random(D), q * (1.0 + rng.random(D))[:, None], np.sort(rng.random(D) * 4.0)[::-1].copy())
eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
U = torch.from_numpy(rng.standard_normal((N, D))).to(dev)
out = torch.empty((N, N), dtype=torch.float32, device=dev)
for _ in range(2):
    eng.score_matrix_dev(U.data_ptr(), None, 1, N, U.data_ptr(), N, out.data_ptr(), N)
torch.cuda.synchronize()
tl = np.zeros((8, 16, 8, 8), np.uint64)
eng._ck(eng._lib.plda_profile_timeline(eng._h, C.c_void_p(tl.ctypes.data), tl.size))
tl = tl.astype(np.int64)
nst = int((tl[2, :, 0, 0] != 0).sum())
print("N=%d D=%d stages/tile=%d" % (N, D, nst))
np.set_printoptions(linewidth=200)
for t in range(3, 5):
    arr, lv = tl[t, :nst, :, 0], tl[t, :nst, :, 1]
    e0, e1 = tl[t, 15, :, 6], tl[t, 15, :, 7]
    nxt = tl[t + 1, 0, :, 0]
    t0 = lv[0].max()
    print("tile %d (times relative to the release of the tile's first barrier)" % t)
    print("  barrier wait (leave-arrive) per stage x wave:\n", (lv - arr))
    print("  stage length (arrive[s+1]-arrive[s]) per stage x wave:\n", np.diff(np.vstack([arr, nxt[None]]), axis=0))
    for w in (0, 4):
        ev = []
        for s in range(nst):
            ev.append(("s%d arrive" % s, arr[s, w])); ev.append(("s%d leave" % s, lv[s, w]))
            for k in range(3):
                if tl[t, s, w, 2 + k]: ev.append(("s%d step%d" % (s, k + 1), tl[t, s, w, 2 + k]))
        ev.append(("epi start", e0[w])); ev.append(("epi end", e1[w]))
        ev.sort(key=lambda x: x[1])
        print("  wave %d:" % w, "  ".join("%s@%d" % (n, v - t0) for n, v in ev))
    print("  epilogue cycles per wave:", e1 - e0)
    print("  tile length (wave 0):", tl[t + 1, 0, 0, 0] - tl[t, 0, 0, 0])
