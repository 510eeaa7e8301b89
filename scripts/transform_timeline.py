"""Phase stamps of workgroup 0 of transform_treg_kernel (PLDA_TRANSFORM_VARIANT=14): per row group and wave, shader-clock
stamps around MFMAs / epilogue parts / barrier (see the two role loops in csrc/transform.hip).  Diagnostic only."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["PLDA_TRANSFORM_VARIANT"] = "14"
import torch
from plda_amd import MPlda

dev = torch.device("cuda", 0)
D, R = 200, 400000
rng = np.random.default_rng(1)
q, _ = np.linalg.qr(rng.standard_normal((D, D)))
eng = MPlda(0)
eng.set_model(rng.random(D), q * (0.5 + rng.random(D))[:, None], np.sort(rng.random(D) * 3 + 0.01)[::-1].copy())
eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
X = torch.randn((R, D), dtype=torch.float64, device=dev)
O = torch.empty((R, D), dtype=torch.float64, device=dev)
for _ in range(3):
    eng.transform_rows_dev(X.data_ptr(), R, D, None, 2, O.data_ptr())
torch.cuda.synchronize()
raw = np.zeros(8 * 16 * 8 * 8, np.uint64)
eng._ck(eng._lib.plda_profile_timeline(eng._h, C.c_void_p(raw.ctypes.data), raw.size))
tl = raw[:16 * 8 * 8].reshape(16, 8, 8).astype(np.int64)
np.set_printoptions(linewidth=200)
t0 = tl[4, 0, 0]
print("two-tile waves 0-3: [top, MFMA end, epi1 end, vmcnt done, barrier left, epi2 end] relative to group 12's top of wave 0; then durations")
for m in (4, 5, 6):
    for w in range(4):
        s = tl[m, w, :6] - t0
        print("  group %2d wave %d:" % (m + 8, w), s, " MFMA %5d epi1 %5d wait %5d barrier %5d epi2 %5d | to next top %5d" % (
            s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], tl[m + 1, w, 0] - tl[m, w, 5]))
print("tile + share waves 4-7: [barrier left, MFMA part 1 end, epi2 end, MFMA rest end, epi1 end, -, at wait, vmcnt done]")
for m in (4, 5, 6):
    for w in range(4, 8):
        s = tl[m, w, :] - t0
        print("  group %2d wave %d:" % (m + 8, w), s[:5], s[6:], " MFMA1 %5d epi2 %5d MFMA2 %5d epi1 %5d wait %5d barrier %5d" % (
            s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], tl[m + 1, w, 7] - tl[m + 1, w, 6], tl[m + 1, w, 0] - tl[m + 1, w, 7]))
print("step length (wave 0 top to top):", [int(tl[m + 1, 0, 0] - tl[m, 0, 0]) for m in range(2, 12)])
