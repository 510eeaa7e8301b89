import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from plda_amd import MPlda
dev = torch.device("cuda", 0)
N, D, K = 100_000, 200, 5000
rng = np.random.default_rng(2)
eng = MPlda(0)
X = torch.from_numpy(rng.random((N, D))).to(dev)
y = torch.from_numpy((np.arange(N) % K).astype(np.int64)).to(dev)
for _ in range(3):
    eng.fit_dev(X.data_ptr(), N, D, y.data_ptr(), K, 10)
print(eng.fit_timings())
