"""Shader clock the part holds under the bf16x3 trials kernel (PLDA_SCORE_DTYPE=bf16x3, PLDA_GEMM_VARIANT=63: workgroup 0's
s_memtime against the 100 MHz s_memrealtime), and workgroup 0's cycles per tile pair against the MFMA cycles of the pair."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ["PLDA_SCORE_DTYPE"] = "bf16x3"; os.environ["PLDA_GEMM_VARIANT"] = "63"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 200
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
raw = np.zeros(8 * 16 * 8 * 8, np.uint64)
nsteps = ((D + 7) // 8 * 8 + 15) // 16
for g in range(3):
    for r in range(10):
        e.score_matrix_dev(U.data_ptr(), None, 1, N, U.data_ptr(), N, out.data_ptr(), N)
    t, n, fl = e.profile_read(reset=True)
    e._ck(e._lib.plda_profile_timeline(e._h, C.c_void_p(raw.ctypes.data), raw.size))
    cyc, real, tiles = int(raw[0]), int(raw[1]), int(raw[2])
    mhz = cyc / (real / 100.0)
    ideal = tiles * nsteps * 2 * 1536            # both groups' C phases, 48 MFMAs x 32 cycles each
    print("%.3f ms/launch | shader clock %.0f MHz | workgroup 0: %d tiles per group, %d cycles -> MFMA-busy in cycles %.3f" % (t / n, mhz, tiles, cyc, ideal / cyc))
