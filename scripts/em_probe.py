"""EM-dominated fit for a kernel trace: small N, D = 512 (or argv[1]), 10 iterations, a few fits.
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o t -- python scripts/em_probe.py 512"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from plda_amd import MPlda  # noqa: E402

dev = torch.device("cuda", 0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N, K = 40 * D, 4 * D
rng = np.random.default_rng(3)
eng = MPlda(0)
X = torch.from_numpy(rng.random((N, D))).to(dev)
y = torch.from_numpy((np.arange(N) % K).astype(np.int64)).to(dev)
torch.cuda.synchronize()
for _ in range(4):
    eng.fit_dev(X.data_ptr(), N, D, y.data_ptr(), K, 10)
print(eng.fit_timings())
