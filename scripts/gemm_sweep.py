"""Interleaved A/B of trials-GEMM variants (PLDA_GEMM_VARIANT) on the C2 shape.
Tuning tool only (synthetic model, no fit); numbers quoted anywhere come from bench.py."""
import os
os.environ.setdefault("PLDA_LIB_DIAG", "1")      # measurement arms: the diagnostic build (python -m plda_amd.build --diag)
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plda_amd import MPlda

variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4".split(","))]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
D = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
q, _ = np.linalg.qr(rng.standard_normal((D, D)))
T = q * (1.0 + rng.random(D))[:, None]
psi = np.sort(rng.random(D) * 4.0)[::-1].copy()
mean = rng.random(D)
engines = {}
for v in variants:
    os.environ["PLDA_GEMM_VARIANT"] = str(v)
    e = MPlda(0)
    e.set_model(mean, T, psi)
    e.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    e.profile_enable(True)
    engines[v] = e
X = torch.from_numpy(rng.random((N, D))).to(dev)
U = torch.empty((N, D), dtype=torch.float64, device=dev)
engines[variants[0]].transform_rows_dev(X.data_ptr(), N, D, None, 1, U.data_ptr())
out = torch.empty((N, N), dtype=torch.float32, device=dev)
res = {v: [] for v in variants}
for r in range(rounds + 1):
    for v in variants:
        e = engines[v]
        e.score_matrix_dev(U.data_ptr(), None, 1, N, U.data_ptr(), N, out.data_ptr(), N)
        ms, n, fl = e.profile_read(reset=True)
        if r > 0:
            res[v].append(ms)
# outputs of the arms that compute real scores, against the first one (the bounding arms 34-36 write garbage)
first = None
for v in variants:
    if 34 <= v <= 36 or 44 <= v <= 46:
        continue
    engines[v].score_matrix_dev(U.data_ptr(), None, 1, N, U.data_ptr(), N, out.data_ptr(), N)
    torch.cuda.synchronize()
    chk = (int(out.view(torch.int32)[::97].to(torch.int64).sum().item()), int(out.view(torch.int32)[-3000:].to(torch.int64).sum().item()))
    if first is None:
        first = chk
    print("variant %d: output checksum %s%s" % (v, chk, "" if chk == first else "  != variant %d" % variants[0]))
ref = None
for v in variants:
    a = np.array(res[v])
    tf = 2.0 * D * N * N / (np.median(a) * 1e-3) / 1e12
    print("variant %d: median %.3f ms  min %.3f ms  -> %.1f TFLOP/s (%.1f%% of 157.3)" % (v, np.median(a), a.min(), tf, 100 * tf / 157.3))
