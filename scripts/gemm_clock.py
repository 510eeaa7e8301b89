"""Sustained clock and cycle efficiency of a trials-GEMM kernel: the product kernel with two stamp pairs of workgroup 0
(PLDA_GEMM_VARIANT 37 = the two-waves-per-SIMD kernel, 47 = the one-wave-per-SIMD kernel), launched back to back.
Per group of launches: ms per launch (HIP events), the shader clock (s_memtime against the 100 MHz s_memrealtime),
workgroup 0's cycles per tile against the tile's MFMA cycles -> MFMA-busy in CYCLES, beside the wall-clock fraction.
usage: gemm_clock.py variant N D launches"""
import ctypes as C
import os
os.environ.setdefault("PLDA_LIB_DIAG", "1")      # measurement arms: the diagnostic build (python -m plda_amd.build --diag)
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PLDA_GEMM_VARIANT"] = sys.argv[1] if len(sys.argv) > 1 else "47"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
D = int(sys.argv[3]) if len(sys.argv) > 3 else 200
L = int(sys.argv[4]) if len(sys.argv) > 4 else 60
import torch
from plda_amd import MPlda
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
q, _ = np.linalg.qr(rng.standard_normal((D, D)))
e = MPlda(0)
e.set_model(rng.random(D), q * (1.0 + rng.random(D))[:, None], np.sort(rng.random(D) * 4.0)[::-1].copy())
e.set_stream(torch.cuda.current_stream(dev).cuda_stream)
e.profile_enable(True)
U = torch.from_numpy(rng.standard_normal((N, D))).to(dev)
out = torch.empty((N, N), dtype=torch.float32, device=dev)
nsteps = max((D + 7) // 8 * 8, 16) // 8
ideal_tile = nsteps * 4096 + 1024          # MFMA cycles of a 256 x 256 tile per SIMD (k steps + the bias MFMAs)
raw = np.zeros(8 * 16 * 8 * 8, np.uint64)
group = 10
for g0 in range(0, L, group):
    for r in range(group):
        e.score_matrix_dev(U.data_ptr(), None, 1, N, U.data_ptr(), N, out.data_ptr(), N)
    t, n, fl = e.profile_read(reset=True)
    e._ck(e._lib.plda_profile_timeline(e._h, C.c_void_p(raw.ctypes.data), raw.size))
    cyc, real, tiles = int(raw[0]), int(raw[1]), int(raw[2])
    mhz = cyc / (real / 100.0)
    print("variant %s launches %3d..%3d: %.3f ms/launch = %.1f %% of 157.3 TF wall | shader clock %.0f MHz (%.3f of 2400) | workgroup 0: %d tiles, %.0f cycles/tile "
          "-> %.1f %% MFMA-busy in cycles | product %.1f %%" % (sys.argv[1], g0, g0 + group - 1, t / n, 100 * fl / (t * 1e-3) / 157.3e12, mhz, mhz / 2400.0,
                                                               tiles, cyc / max(tiles, 1), 100.0 * ideal_tile * tiles / max(cyc, 1),
                                                               100.0 * ideal_tile * tiles / max(cyc, 1) * mhz / 2400.0))
# the last launch, every workgroup: when it started and ended (100 MHz ticks), its cycles and tiles
wg = raw[8:8 + 4 * 256].reshape(256, 4).astype(np.int64)
cyc, r0, r1, tl = wg[:, 0], wg[:, 1], wg[:, 2], wg[:, 3]
span = (r1.max() - r0.min()) / 100.0
dur = (r1 - r0) / 100.0
print("last launch: first start -> last end %.1f us; workgroup durations min %.1f / median %.1f / max %.1f us; start spread %.1f us; end spread %.1f us"
      % (span, dur.min(), np.median(dur), dur.max(), (r0.max() - r0.min()) / 100.0, (r1.max() - r1.min()) / 100.0))
print("tiles per workgroup: min %d max %d mean %.2f;  cycles per tile min %.0f / median %.0f / max %.0f;  clock per workgroup min %.0f / max %.0f MHz"
      % (tl.min(), tl.max(), tl.mean(), (cyc / np.maximum(tl, 1)).min(), np.median(cyc / np.maximum(tl, 1)), (cyc / np.maximum(tl, 1)).max(),
         (cyc / dur).min(), (cyc / dur).max()))
for x in range(8):
    m = np.arange(256) % 8 == x
    print("  XCD %d: mean duration %.1f us, mean cycles/tile %.0f, mean clock %.0f MHz, tiles %d" % (x, dur[m].mean(), (cyc[m] / np.maximum(tl[m], 1)).mean(), (cyc[m] / dur[m]).mean(), tl[m].sum()))
os.system("rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\\|power' | head -4")
