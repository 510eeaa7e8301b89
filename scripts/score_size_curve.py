"""trials/s of one score_matrix_dev call (operand packing + GEMM, HBM-resident fp64 inputs) over the sizes between a single
tile and the BASELINE shapes: where the dispatch changes kernels (128 x 128 -> 256 x 256 two waves per SIMD -> one wave
per SIMD) and what each size holds of the fp32 MFMA peak.  Tuning / evidence tool; headline numbers come from bench.py.

usage: python scripts/score_size_curve.py [D,D,...] [json path | -] [MxNt,MxNt,... | - (the built-in list)] [mixed: enrol counts 1..5]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plda_amd import MPlda

PEAK = 157.3e12
dims = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "64,200,512".split(","))]
dev = torch.device("cuda", 0)
rows = []
SHAPES = [(256, 256), (512, 512), (1024, 1024), (2048, 2048), (1000, 20000), (4096, 4096), (5000, 50000), (8192, 8192), (8000, 9000),
          (12000, 12000), (16384, 16384), (20000, 20000), (32768, 32768), (2048, 500000), (50000, 50000), (65536, 65536), (100000, 100000)]
MIXED = len(sys.argv) > 4 and sys.argv[4] == 'mixed'
if len(sys.argv) > 3 and sys.argv[3] != '-':
    SHAPES = [tuple(int(v) for v in t.split('x')) for t in sys.argv[3].split(',')]
for D in dims:
    rng = np.random.default_rng(D)
    q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    eng = MPlda(0)
    eng.set_model(rng.random(D), q * (1.0 + rng.random(D))[:, None], np.sort(rng.random(D) * 4.0)[::-1].copy())
    stream = torch.cuda.Stream(dev)
    eng.set_stream(stream.cuda_stream)
    nmax = max(max(s) for s in SHAPES)
    U = torch.randn((nmax, D), dtype=torch.float64, device=dev)
    V = torch.randn((nmax, D), dtype=torch.float64, device=dev)
    counts = torch.randint(1, 6, (nmax,), dtype=torch.int32, device=dev)
    dn = counts.data_ptr() if MIXED else None
    torch.cuda.synchronize()
    for m, nt in SHAPES:
        if m * nt * 4 > 60e9:
            continue
        out = torch.empty((m, nt), dtype=torch.float32, device=dev)
        reps = int(max(3, min(200, 2e10 / (m * nt))))
        with torch.cuda.stream(stream):
            for _ in range(2):
                eng.score_matrix_dev(U.data_ptr(), dn, 1, m, V.data_ptr(), nt, out.data_ptr(), nt)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                eng.score_matrix_dev(U.data_ptr(), dn, 1, m, V.data_ptr(), nt, out.data_ptr(), nt)
            e1.record(stream)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tps = m * nt / (ms * 1e-3)
        rows.append({"D": D, "M": m, "Nt": nt, "ms_per_call": ms, "trials_per_s": tps, "of_fp32_mfma_peak": tps * 2 * eng.score_last_shape()[2] / PEAK, "gemm_depth": eng.score_last_shape()[2],
                     "kernel": eng.score_last_kernel()})
        print("D=%4d %7d x %7d  %9.4f ms  %.3e trials/s  %.3f of peak  %s" % (D, m, nt, ms, tps, tps * 2 * eng.score_last_shape()[2] / PEAK, eng.score_last_kernel()))
        del out
if len(sys.argv) > 2 and sys.argv[2] != "-":
    json.dump(rows, open(sys.argv[2], "w"), indent=1)
