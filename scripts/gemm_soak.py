"""Sustained-clock check: one trials-GEMM variant back to back, per-launch ms (HIP events) printed in
groups, so that DVFS / thermal drift shows up.  usage: gemm_soak.py variant N D launches"""
import os, sys
os.environ.setdefault("PLDA_LIB_DIAG", "1")      # measurement arms: the diagnostic build (python -m plda_amd.build --diag)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PLDA_GEMM_VARIANT"] = sys.argv[1] if len(sys.argv) > 1 else "0"
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
ms = []
for r in range(L):
    e.score_matrix_dev(U.data_ptr(), None, 1, N, U.data_ptr(), N, out.data_ptr(), N)
for r in range(L):
    pass
t, n, fl = e.profile_read(reset=True)
print("variant %s N=%d D=%d: %d launches back to back, mean %.3f ms -> %.1f%% of 157.3 TF" % (sys.argv[1], N, D, n, t / n, 100 * fl / (t * 1e-3) / 157.3e12))
# per-launch: relaunch with a sync and a read each time
for r in range(L):
    e.score_matrix_dev(U.data_ptr(), None, 1, N, U.data_ptr(), N, out.data_ptr(), N)
    t, n, fl = e.profile_read(reset=True)
    ms.append(t)
ms = np.array(ms)
print("per-launch ms (synchronised after each):", " ".join("%.2f" % v for v in ms))
os.system("rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\\|power' | head -4")
