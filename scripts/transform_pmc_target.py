import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from plda_amd import MPlda
dev = torch.device("cuda", 0)
N, D = 1_200_000, 256
rng = np.random.default_rng(1)
q, _ = np.linalg.qr(rng.standard_normal((D, D)))
eng = MPlda(0)
eng.set_model(rng.random(D), q * (1.0 + rng.random(D))[:, None], np.sort(rng.random(D) * 4)[::-1].copy())
eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
X = torch.rand((N, D), dtype=torch.float64, device=dev)
U = torch.empty((N, D), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for _ in range(4):
    eng.transform_rows_dev(X.data_ptr(), N, D, None, 1, U.data_ptr())
torch.cuda.synchronize()
