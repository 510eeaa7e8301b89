"""Per-wave timeline of workgroup 0 of the trials GEMM (PLDA_GEMM_VARIANT=31, plda_profile_timeline):
shader-clock stamps at every stage barrier (arrive / leave) and around every tile epilogue.
Diagnostic only; prints, for tiles 2..5 (steady state), per wave: cycles per stage, cycles
waited at each barrier, epilogue cycles.  usage: gemm_timeline.py [N] [D]"""
import ctypes as C
import os
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
eng.set_stream(torch.cuda.current_stream(dev).cuda_stream or None)
U = torch.from_numpy(rng.standard_normal((N, D))).to(dev)
out = torch.empty((N, N), dtype=torch.float32, device=dev)
for _ in range(2):
    eng.score_matrix_dev(U.data_ptr(), None, 1, N, U.data_ptr(), N, out.data_ptr(), N)
torch.cuda.synchronize()
tl = np.zeros((8, 16, 8, 4), np.uint64)
eng._ck(eng._lib.plda_profile_timeline(eng._h, C.c_void_p(tl.ctypes.data), tl.size))
tl = tl.astype(np.int64)
nst = int((tl[2, :, 0, 0] != 0).sum())
print("N=%d D=%d stages/tile=%d" % (N, D, nst))
for t in range(2, 6):
    arr, lv = tl[t, :nst, :, 0], tl[t, :nst, :, 1]
    e0, e1 = tl[t, 15, :, 2], tl[t, 15, :, 3]
    nxt = tl[t + 1, 0, :, 0]
    print("tile %d" % t)
    print("  barrier wait (leave-arrive) per stage x wave:\n", (lv - arr))
    print("  stage length (arrive[s+1]-arrive[s]) per stage x wave:\n", np.diff(np.vstack([arr, nxt[None]]), axis=0))
    print("  arrive spread across waves per stage:", arr.max(1) - arr.min(1))
    print("  epilogue cycles per wave:", e1 - e0, " epilogue start spread:", e0.max() - e0.min())
    print("  tile length (wave 0):", tl[t + 1, 0, 0, 0] - tl[t, 0, 0, 0])
