"""Per-wave timeline of workgroup 0 of the one-wave-per-SIMD trials GEMM (PLDA_GEMM_VARIANT=41, score_bt4.inc):
shader-clock stamps at every stage barrier (arrive / leave), at the start of the stage's steps and around the
block-major part of a tile's last stage.  Diagnostic only.  usage: gemm_timeline4.py [N] [D]"""
import ctypes as C
import os
os.environ.setdefault("PLDA_LIB_DIAG", "1")      # measurement arms: the diagnostic build (python -m plda_amd.build --diag)
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PLDA_GEMM_VARIANT"] = "41"
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
raw = np.zeros(8 * 16 * 8 * 8, np.uint64)
eng._ck(eng._lib.plda_profile_timeline(eng._h, C.c_void_p(raw.ctypes.data), raw.size))
tl = raw[:8 * 16 * 4 * 8].reshape(8, 16, 4, 8).astype(np.int64)
nsteps = (max((D + 7) // 8 * 8, 16)) // 8
nst = (nsteps + 3) // 4
sbase, srem = nsteps // nst, nsteps % nst
sizes = [sbase + (1 if j >= nst - srem else 0) for j in range(nst)]
print("N=%d D=%d steps/tile=%d stages=%s ideal tile=%d cycles" % (N, D, nsteps, sizes, nsteps * 4096 + 1024))
np.set_printoptions(linewidth=220)
for t in range(3, 6):
    ns = min(nst, 16)
    arr, lv = tl[t, :ns, :, 0], tl[t, :ns, :, 1]
    b0, b1 = tl[t, 15, :, 6], tl[t, 15, :, 7]
    nxt = tl[t + 1, 0, :, 0]
    print("tile %d" % t)
    print("  barrier wait (leave - arrive), stage x wave:\n", (lv - arr))
    seg = np.diff(np.vstack([arr, nxt[None]]), axis=0)
    # between the barrier of stage s and of stage s+1 lie: last step of s (or, for the last stage, its block-major part +
    # the next tile's bias MFMAs) and all but the last step of s+1 (for the last stage: its first step only)
    ideal = []
    for s in range(ns):
        if s < nst - 2: ideal.append(4096 * (1 + sizes[s + 1] - 1))
        elif s == nst - 2: ideal.append(4096 * (1 + 1))
        else: ideal.append(4096 * (sizes[-1] - 1) + 1024 + 4096 * (sizes[0] - 1))
    print("  barrier-to-barrier cycles, stage x wave (ideal %s):\n" % ideal, seg)
    print("  excess over ideal, wave 0:", seg[:, 0] - np.array(ideal), " sum", int((seg[:, 0] - np.array(ideal)).sum()))
    print("  block-major part (ideal %d): " % (4096 * (sizes[-1] - 1)), b1 - b0, "  its start after the last barrier:", b0 - lv[ns - 1])
    top0, top1 = tl[t, 15, :, 2], tl[t, 15, :, 3]
    ntop0 = tl[t + 1, 15, :, 2]
    print("  tile top (entry -> behind the bias MFMAs + carried stores; ideal 1024):", top1 - top0,
          " | top -> first barrier:", arr[0] - top1, " (ideal %d)" % (4096 * (sizes[0] - 1)),
          " | block-major end -> next tile's top:", ntop0 - b1)
    print("  tile length per wave:", nxt - arr[0], " = %.1f %% MFMA-busy" % (100.0 * (nsteps * 4096 + 1024) / float((nxt - arr[0])[0])))
