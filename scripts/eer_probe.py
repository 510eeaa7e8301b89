"""EER of a trials-matrix-like fp32 array: ragged shapes against exact numpy counts, then time at full size.
(profiles/r02_eer_probe.json was written by the A/B version of this script, when the row layout still existed.)
Usage: python scripts/eer_probe.py [N]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from plda_amd import MPlda, eer  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000


eng = MPlda(0)
eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
res = {"N": N, "ragged": []}
g = torch.Generator(device=dev); g.manual_seed(3)
for (m, nt, ld, k) in [(1, 1, 1, 1), (3, 5, 7, 2), (257, 1023, 1023, 9), (300, 1025, 1028, 16), (1000, 4099, 4100, 50), (5000, 3001, 3001, 40)]:
    S = torch.randn((m, ld), generator=g, device=dev, dtype=torch.float32) * 8
    es = torch.randint(0, k, (m,), generator=g, device=dev, dtype=torch.int64)
    ts = torch.randint(0, k, (nt,), generator=g, device=dev, dtype=torch.int64)
    if m * nt > 1:
        es[0] = 0; ts[0] = 0; ts[-1] = 1 if k > 1 else 0
    S[:, :nt] += 12.0 * (es[:, None] == ts[None, :])
    S = torch.round(S * 4) / 4 if m < 1000 else S         # ties on the small shapes
    torch.cuda.synchronize()
    try:
        out = [float(x) for x in eer.eer_from_matrix_dev(eng, S.data_ptr(), ld, m, nt, es.data_ptr(), ts.data_ptr())]
        sub = S[:, :nt].cpu().numpy().astype(np.float64); tgt = (es[:, None] == ts[None, :]).cpu().numpy()
        ok = bool(float((sub[~tgt] >= out[0]).mean()) == out[1] and float((sub[tgt] < out[0]).mean()) == out[2])
    except Exception as ex:                                      # noqa: BLE001
        out, ok = repr(ex)[:80], None
    res["ragged"].append({"shape": (m, nt, ld), "farfrr_consistent": ok, "out": out})

K = max(N // 20, 1)
y = (torch.arange(N, device=dev) % K).to(torch.int64)
S = torch.empty((N, N), dtype=torch.float32, device=dev)
step = 4096
for r in range(0, N, step):
    blk = S[r:r + step]
    blk.normal_(generator=g).mul_(10.0)
    blk += 25.0 * (y[r:r + step, None] == y[None, :])
torch.cuda.synchronize()
out = eer.eer_from_matrix_dev(eng, S.data_ptr(), N, N, N, y.data_ptr(), y.data_ptr())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    out = eer.eer_from_matrix_dev(eng, S.data_ptr(), N, N, N, y.data_ptr(), y.data_ptr())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
eng.trace_enable(True); eng.trace_read(reset=True)
eer.eer_from_matrix_dev(eng, S.data_ptr(), N, N, N, y.data_ptr(), y.data_ptr())
spans = [{"name": sp["name"], "ms": round(sp["ms"], 3)} for sp in eng.trace_read(reset=True)]
eng.trace_enable(False)
res["full"] = {"ms": dt * 1e3, "GBps_of_one_read": N * N * 4 / dt / 1e9, "out": [float(x) for x in out], "spans": spans}
# the three-pass arm on the same matrix (PLDA_EER_VARIANT=1): must give the same six numbers
os.environ["PLDA_EER_VARIANT"] = "1"
e3 = MPlda(0)
del os.environ["PLDA_EER_VARIANT"]
e3.set_stream(torch.cuda.current_stream(dev).cuda_stream)
out3 = eer.eer_from_matrix_dev(e3, S.data_ptr(), N, N, N, y.data_ptr(), y.data_ptr())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    out3 = eer.eer_from_matrix_dev(e3, S.data_ptr(), N, N, N, y.data_ptr(), y.data_ptr())
torch.cuda.synchronize()
dt3 = (time.perf_counter() - t0) / 3
res["three_pass_arm"] = {"ms": dt3 * 1e3, "GBps_3pass": 3 * N * N * 4 / dt3 / 1e9, "identical": bool(np.array_equal(out, out3))}
print(json.dumps(res))
