"""GPU, BASELINE.json full size (C2: 100k x 100k trials, D = 200): size-independent
properties of the trials matrix, since the oracle cannot finish 1e10 trials.

  * random trials agree with the fp64 trial-list kernel (itself oracle-checked at small
    size in test_gpu_scoring.py) within the 1e-4 tolerance;
  * tiling invariance: any sub-block scored on its own is BIT-IDENTICAL to the same
    block of the full matrix (fixed k-order fp32 FMA chains, position independent);
  * with n = 1 the LLR is symmetric in (enrol, test): S[i, j] == S[j, i] to fp32 rounding.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_c2_full_size_properties():
    import torch
    from plda_amd import MPlda
    dev = torch.device("cuda", 0)
    D, N = 200, 100000
    rng = np.random.default_rng(2)
    # a realistic model without a 100k fit: random orthogonal-ish transform, decaying psi
    q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    T = q * (1.0 + rng.random(D))[:, None]
    psi = np.sort(rng.random(D) * 4.0)[::-1].copy()
    eng = MPlda(0)
    eng.set_model(rng.random(D), T, psi)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream or None)
    X = torch.from_numpy(rng.random((N, D))).to(dev)
    Ut = torch.empty((N, D), dtype=torch.float64, device=dev)
    eng.transform_rows_dev(X.data_ptr(), N, D, None, 1, Ut.data_ptr())
    out = torch.empty((N, N), dtype=torch.float32, device=dev)
    eng.score_matrix_dev(Ut.data_ptr(), None, 1, N, Ut.data_ptr(), N, out.data_ptr(), N)
    torch.cuda.synchronize()
    assert torch.isfinite(out[::997]).all()

    # (1) 4096 random trials vs the fp64 trial-list kernel
    P = 4096
    e = rng.integers(0, N, P); t = rng.integers(0, N, P)
    rows = np.unique(np.concatenate([e, t]))
    remap = {int(r): i for i, r in enumerate(rows)}
    Uh = Ut[torch.from_numpy(rows).to(dev)].cpu().numpy()
    ref = eng.score_trials((np.ones(len(rows), np.int32), Uh), (1, Uh),
                           np.array([remap[int(i)] for i in e]), np.array([remap[int(i)] for i in t]))
    got = out[torch.from_numpy(e).to(dev), torch.from_numpy(t).to(dev)].cpu().numpy().astype(np.float64)
    tol = 1e-4 * np.maximum(np.abs(ref), np.abs(ref).mean())
    assert (np.abs(got - ref) <= tol).all(), np.abs(got - ref).max()

    # (2) tiling invariance, bit-exact, at unaligned offsets
    for (r0, r1, c0, c1) in [(0, 300, 0, 500), (12345, 12345 + 777, 54321, 54321 + 1111), (N - 129, N, N - 257, N)]:
        blk = torch.empty((r1 - r0, c1 - c0), dtype=torch.float32, device=dev)
        eng.score_matrix_dev(Ut[r0:r1].data_ptr(), None, 1, r1 - r0, Ut[c0:c1].data_ptr(), c1 - c0,
                             blk.data_ptr(), c1 - c0)
        torch.cuda.synchronize()
        assert torch.equal(blk, out[r0:r1, c0:c1])

    # (3) symmetry for n = 1
    a = out[:4096, 50000:54096]
    b = out[50000:54096, :4096].T
    scale = float(a.abs().mean())
    assert float((a - b).abs().max()) <= 2e-5 * max(scale, 1.0)
