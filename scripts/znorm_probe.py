"""GPU: z-norm statistics (MPlda_norm, pldamodule.cpp:196-256) at the C5 shape -- by moments (default) against the
fused fp32 GEMM arm (PLDA_ZNORM_VARIANT=1): time, and error of mean / std against the fp64 GEMM-form oracle on
sampled models.   python scripts/znorm_probe.py [M Nb D]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from plda_amd import MPlda
from oracle import plda_oracle_np as onp

M, Nb, D = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (50000, 200000, 200)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(51)
q, _ = np.linalg.qr(rng.standard_normal((D, D)))
mean, T, psi = rng.random(D), q * (1.0 + rng.random(D))[:, None], np.sort(rng.random(D) * 4.0 + 0.05)[::-1].copy()
g = torch.Generator(device=dev); g.manual_seed(9)
bkg = torch.rand((Nb, D), dtype=torch.float64, device=dev, generator=g)
models = torch.randn((M, D), dtype=torch.float64, device=dev, generator=g)
sel = np.array([0, 1, 777 % M, M // 2, M - 1])
model = dict(mean=mean, transform=T, psi=psi, offset=-T @ mean)
rm, rs = onp.norm(model, bkg.cpu().numpy(), models[torch.from_numpy(sel).to(dev)].cpu().numpy())
for variant in ("0", "1"):
    os.environ["PLDA_ZNORM_VARIANT"] = variant
    eng = MPlda(0)
    eng.set_model(mean, T, psi)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        eng.set_stream(stream.cuda_stream)
        zm = torch.empty(M, dtype=torch.float64, device=dev); zs = torch.empty(M, dtype=torch.float64, device=dev)
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.znorm_stats_dev(bkg.data_ptr(), Nb, Nb, D, models.data_ptr(), M, zm.data_ptr(), zs.data_ptr())
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
    gm, gs = zm[torch.from_numpy(sel).to(dev)].cpu().numpy(), zs[torch.from_numpy(sel).to(dev)].cpu().numpy()
    print("variant %s (%s): %.3f ms for %d x %d (incl. cohort transform); max rel err mean %.2e, std %.2e" % (
        variant, "moments" if variant == "0" else "fused fp32 GEMM", best * 1e3, M, Nb,
        (np.abs(gm - rm) / np.maximum(np.abs(rm), np.abs(rm).mean())).max(), (np.abs(gs - rs) / rs).max()), flush=True)
    del eng
