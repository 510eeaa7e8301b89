"""Host enqueue time against GPU time of score_matrix_dev calls that alternate tile grids (tests/test_gpu_bigtile.py:
test_alternating_tile_grids_do_not_stall_the_host).  Usage: python scripts/probe/enqueue_probe.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from plda_amd import MPlda

dev = torch.device("cuda", 0)
d = 512
rng = np.random.default_rng(3)
q, _ = np.linalg.qr(rng.standard_normal((d, d)))
eng = MPlda(0)
eng.set_model(rng.random(d), q * (1.0 + rng.random(d))[:, None], np.sort(rng.random(d) * 4.0 + 0.05)[::-1].copy())
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
eng.set_stream(st.cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(11)
dU = torch.randn((8448, d), dtype=torch.float64, device=dev, generator=g)
dV = torch.randn((8448, d), dtype=torch.float64, device=dev, generator=g)
for shapes in ([(8192, 8192)] * 3, [(8192, 8192), (8192, 8448), (8448, 8192)]):
    outs = [torch.empty((m, n), dtype=torch.float32, device=dev) for m, n in shapes]
    for (m, n), o in zip(shapes, outs):
        eng.score_matrix_dev(dU.data_ptr(), None, 2, m, dV.data_ptr(), n, o.data_ptr(), n)
    torch.cuda.synchronize()
    per = []
    t0 = time.perf_counter()
    for it in range(40):
        (m, n), o = shapes[it % 3], outs[it % 3]
        t1 = time.perf_counter()
        eng.score_matrix_dev(dU.data_ptr(), None, 2, m, dV.data_ptr(), n, o.data_ptr(), n)
        per.append(time.perf_counter() - t1)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(shapes, "enqueue %.2f ms, all %.2f ms, per-call host us: first 6 %s, max %.0f" % (t_enq * 1e3, t_all * 1e3, [round(x * 1e6) for x in per[:6]], max(per) * 1e6))
eng.trace_enable(True)
for it in range(6):
    (m, n), o = shapes[it % 3], outs[it % 3]
    eng.score_matrix_dev(dU.data_ptr(), None, 2, m, dV.data_ptr(), n, o.data_ptr(), n)
torch.cuda.synchronize()
print(eng.trace_read())
