"""GPU parity of the 256 x 256 trials-GEMM kernels -- the kernels every BASELINE.json config runs --
against the fp64 oracle (reference semantics: /root/reference/src/pldamodule.cpp:258-277, driven
M x Nt times by scoring/scorePLDA.py:302-318).

`launch_gemm` (plda_amd/csrc/score.hip) picks a kernel by problem size, so the oracle tests of
test_gpu_scoring.py (<= 517 columns) only ever reach the 128 x 128 kernel.  Here:

  (i)   forced dispatch (PLDA_GEMM_VARIANT = 30, read at plda_create) of the persistent 256 x 256
        kernel on shapes small enough for the per-trial C oracle `score_block`: uniform n, mixed n
        (bucketed by distinct count since round 5: depth D + G - 1; tests/test_gpu_mixed_counts.py covers the forms)
        and the z-norm epilogue; ragged edges in both dimensions;
  (ii)  default dispatch at 8192 x 8192 (exactly the 1024-tile threshold) with the BASELINE shapes'
        depths -- D = 200 uniform n (C2), D = 512 n = 100 (C3), D = 256 mixed n in 1..5 (C4),
        D = 200 z-normalised (C5) -- against the fp64 GEMM-form oracle `llr_matrix`;
  (iii) a packed test operand of 4.3 GB -- beyond the kernel's 32-bit DMA offsets -- which the host
        scores in column blocks, checked on sampled columns of every block;
  (iv)  both kernels start from the same bias value and contract in the same k order, so their
        fp32 outputs are BIT-identical.
"""
import numpy as np
import pytest

from conftest import score_tol

pytestmark = pytest.mark.gpu


def _model(d, seed):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    T = q * (1.0 + rng.random(d))[:, None]
    psi = np.sort(rng.random(d) * 4.0 + 0.05)[::-1].copy()
    return rng.random(d), T, psi


def _engine(monkeypatch, variant, d, seed=3):
    from plda_amd import MPlda
    if variant is None:
        monkeypatch.delenv("PLDA_GEMM_VARIANT", raising=False)
    else:
        monkeypatch.setenv("PLDA_GEMM_VARIANT", str(variant))
    eng = MPlda(0)
    mean, T, psi = _model(d, seed)
    eng.set_model(mean, T, psi)
    return eng, psi


def _vectors(rng, rows, d, scale=1.0):
    """Rows of the magnitude TransformIvector produces (|t|^2 ~ D), without going through it."""
    return rng.standard_normal((rows, d)) * scale


# ---------------------------------------------------------------- (i) forced dispatch, C oracle
SHAPES = [(200, 300, 517), (64, 1024, 1024), (33, 257, 769), (200, 1, 700), (8, 513, 255)]


@pytest.mark.parametrize("variant", [30])
@pytest.mark.parametrize("d,m,nt", SHAPES)
def test_forced_uniform(monkeypatch, oracle, variant, d, m, nt):
    eng, psi = _engine(monkeypatch, variant, d)
    rng = np.random.default_rng(100 + d + m)
    U, V = _vectors(rng, m, d), _vectors(rng, nt, d)
    for n in (1, 7):
        ref = oracle.score_block(psi, U, n, V)
        got = eng.score_matrix((n, U), (1, V))
        assert (np.abs(got - ref) <= score_tol(ref)).all(), (variant, n, np.abs(got - ref).max())


@pytest.mark.parametrize("variant", [30])
@pytest.mark.parametrize("d,m,nt", [(96, 300, 517), (256, 700, 300), (20, 1024, 1024)])
def test_forced_mixed_counts(monkeypatch, oracle, variant, d, m, nt):
    """enrol counts differ (n in 1..5: five buckets, GEMM depth D + 4)."""
    eng, psi = _engine(monkeypatch, variant, d)
    rng = np.random.default_rng(200 + d)
    counts = rng.integers(1, 6, m).astype(np.int32)
    U, V = _vectors(rng, m, d), _vectors(rng, nt, d)
    ref = oracle.score_block(psi, U, counts, V)
    got = eng.score_matrix((counts, U), (1, V))
    assert (np.abs(got - ref) <= score_tol(ref)).all(), (variant, np.abs(got - ref).max())


@pytest.mark.parametrize("variant", [30])
@pytest.mark.parametrize("mixed", [False, True])
def test_forced_znorm(monkeypatch, oracle, variant, mixed):
    """z-norm (pldamodule.cpp:269-273) in the big-tile epilogue; rows without statistics stay raw."""
    d, m, nt = 120, 300, 517
    eng, psi = _engine(monkeypatch, variant, d)
    rng = np.random.default_rng(300 + int(mixed))
    counts = rng.integers(1, 6, m).astype(np.int32) if mixed else np.full(m, 3, np.int32)
    U, V = _vectors(rng, m, d), _vectors(rng, nt, d)
    raw = oracle.score_block(psi, U, counts, V)
    zm, zs = raw.mean(1), raw.std(1)
    zs[::17] = 0.0                                   # engine convention: std 0 -> row left un-normalised
    ids = np.arange(m, dtype=np.int64)
    eng._meanz = {int(k): float(v) for k, v in zip(ids, zm) if zs[k] != 0.0}
    eng._stdvz = {int(k): float(v) for k, v in zip(ids, zs) if zs[k] != 0.0}
    zs_o = np.where(zs == 0.0, 1.0, zs); zm_o = np.where(zs == 0.0, 0.0, zm)
    ref = oracle.score_block(psi, U, counts, V, zm_o, zs_o)
    got = eng.score_matrix((counts, U, ids), (1, V))
    assert (np.abs(got - ref) <= score_tol(ref)).all(), (variant, np.abs(got - ref).max())


def test_kernels_bit_identical(monkeypatch):
    """(iv) the 128 x 128 kernel (variant 20) and the 256 x 256 kernel (30) accumulate each trial from
    the same starting value in the same k order: same bits."""
    d, m, nt = 200, 600, 1100
    rng = np.random.default_rng(7)
    U, V = _vectors(rng, m, d), _vectors(rng, nt, d)
    counts = rng.integers(1, 6, m).astype(np.int32)
    outs = {}
    for variant in (20, 30, 32):
        eng, _ = _engine(monkeypatch, variant, d)
        outs[variant] = (eng.score_matrix((2, U), (1, V)), eng.score_matrix((counts, U), (1, V)))
    for variant in (30, 32):       # 32: the 256 x 256 kernel with the LDS-transposing epilogue (A/B arm of the direct stores)
        assert np.array_equal(outs[variant][0], outs[20][0]), variant
        assert np.array_equal(outs[variant][1], outs[20][1]), variant


# (d, m, nt): depth classes of the one-wave-per-SIMD kernel (variant 40, score_bt4.inc) -- last stage of 4 steps (25 steps:
# 3,3,3,4,4,4,4; 12 steps: 4,4,4; 64 steps), of 3 steps (9 steps: 3,3,3; 5 steps: 2,3; 6 steps: 3,3), one regular stage
# only (8 steps: 4,4), an odd regular stage in front (7 steps: 3,4) -- on shapes with whole tiles only, with a fringe of
# rows, of columns, of both (the fringe goes to the 128 x 128 kernel), and several tiles per workgroup (3072 x 5120 = 240 tiles)
BT4_SHAPES = [(200, 600, 1100), (200, 512, 768), (96, 300, 517), (72, 256, 1024), (40, 700, 300), (48, 513, 255 + 256),
              (64, 1024, 1024), (56, 257, 769), (512, 520, 600), (200, 3072, 5120), (33, 2300, 2900)]


@pytest.mark.parametrize("d,m,nt", BT4_SHAPES)
def test_one_wave_per_simd_kernel_bit_identical(monkeypatch, d, m, nt):
    """(iv) for the round-4 kernel: same starting value, same k order per trial as the 128 x 128 kernel -> same bits,
    uniform and mixed enrol counts (depth 2D), with and without the z-norm map folded into the operands."""
    rng = np.random.default_rng(d + m)
    U, V = _vectors(rng, m, d), _vectors(rng, nt, d)
    counts = rng.integers(1, 6, m).astype(np.int32)
    ids = np.arange(m, dtype=np.int64)
    outs = {}
    for variant in (20, 40):
        eng, _ = _engine(monkeypatch, variant, d)
        plain = (eng.score_matrix((2, U), (1, V)), eng.score_matrix((counts, U), (1, V)))
        eng._meanz = {int(k): -40.0 + 0.01 * k for k in ids}
        eng._stdvz = {int(k): 3.0 + 0.001 * k for k in ids}
        outs[variant] = plain + (eng.score_matrix((2, U, ids), (1, V)),)
    for a, b in zip(outs[40], outs[20]):
        assert np.isfinite(a).all()
        assert np.array_equal(a, b), (d, m, nt, np.abs(a - b).max())


# tile-queue orders of the one-wave-per-SIMD kernel (score.hip: bt4_schedule): 48 walks the patch rows, 49 goes down the patch
# columns (queue x owns columns x, x + 8, ...; the columns beyond the last round of eight dealt patch by patch).  The order only
# moves tiles between workgroups: same bits.  Shapes: 21 x 17 tiles (3 patch columns: remainder only), 5 x 75 tiles (10 patch
# columns: one round of eight + 2), 2 x 130 tiles (17 patch columns, one patch row), ragged edges.
@pytest.mark.parametrize("d,m,nt", [(72, 5300, 4200), (96, 1100, 19000), (200, 400, 33100)])
def test_tile_queue_orders_bit_identical(monkeypatch, d, m, nt):
    rng = np.random.default_rng(d + nt)
    U, V = _vectors(rng, m, d), _vectors(rng, nt, d)
    counts = rng.integers(1, 4, m).astype(np.int32)
    outs = {}
    for variant in (48, 49):
        eng, _ = _engine(monkeypatch, variant, d)
        outs[variant] = (eng.score_matrix((2, U), (1, V)), eng.score_matrix((counts, U), (1, V)))
        assert eng.score_last_shape()[2] > 0
    eng, _ = _engine(monkeypatch, 20, d)
    ref = eng.score_matrix((2, U), (1, V))
    for a, b in zip(outs[48], outs[49]):
        assert np.isfinite(a).all() and np.array_equal(a, b)
    assert np.array_equal(outs[49][0], ref)


@pytest.mark.parametrize("d,m,nt", [(200, 300, 517), (64, 1024, 1024), (33, 257, 769)])
def test_one_wave_per_simd_kernel_oracle(monkeypatch, oracle, d, m, nt):
    eng, psi = _engine(monkeypatch, 40, d)
    rng = np.random.default_rng(100 + d + m)
    U, V = _vectors(rng, m, d), _vectors(rng, nt, d)
    for n in (1, 7):
        ref = oracle.score_block(psi, U, n, V)
        got = eng.score_matrix((n, U), (1, V))
        assert (np.abs(got - ref) <= score_tol(ref)).all(), (n, np.abs(got - ref).max())


# ---------------------------------------------------------------- (ii) default dispatch, 8192 x 8192
def _default_case(monkeypatch, d, n_enrol, znorm, seed):
    import torch
    from oracle import plda_oracle_np as onp
    dev = torch.device("cuda", 0)
    eng, psi = _engine(monkeypatch, None, d, seed)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    eng.profile_enable(True)
    m = nt = 8192
    rng = np.random.default_rng(seed)
    U, V = _vectors(rng, m, d), _vectors(rng, nt, d)
    ref = onp.llr_matrix(psi, U, n_enrol, V)
    dU, dV = torch.from_numpy(U).to(dev), torch.from_numpy(V).to(dev)
    dn = None
    if np.ndim(n_enrol):
        dn = torch.from_numpy(np.ascontiguousarray(n_enrol, np.int32)).to(dev)
    dzm = dzs = None
    if znorm:
        zm, zs = ref.mean(1), ref.std(1)
        ref = (ref - zm[:, None]) / zs[:, None]
        dzm, dzs = torch.from_numpy(zm).to(dev), torch.from_numpy(zs).to(dev)
    out = torch.full((m, nt), float("nan"), dtype=torch.float32, device=dev)
    eng.score_matrix_dev(dU.data_ptr(), dn.data_ptr() if dn is not None else None,
                         0 if dn is not None else int(n_enrol), m, dV.data_ptr(), nt, out.data_ptr(), nt,
                         dzm.data_ptr() if znorm else None, dzs.data_ptr() if znorm else None)
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    err = np.abs(got - ref)
    assert (err <= score_tol(ref)).all(), err.max()


def test_default_dispatch_c2_uniform(monkeypatch):
    _default_case(monkeypatch, 200, 1, False, 21)


def test_default_dispatch_c3_n100(monkeypatch):
    _default_case(monkeypatch, 512, 100, False, 22)


def test_default_dispatch_c4_mixed(monkeypatch):
    n = np.random.default_rng(4).integers(1, 6, 8192).astype(np.int32)
    _default_case(monkeypatch, 256, n, False, 23)


def test_default_dispatch_c5_znorm(monkeypatch):
    _default_case(monkeypatch, 200, 1, True, 24)


def test_default_dispatch_small_dims(monkeypatch):
    _default_case(monkeypatch, 16, 3, False, 25)       # two 16-k steps per tile: one 2-step stage
    _default_case(monkeypatch, 8, 1, False, 26)        # a single step per tile
    _default_case(monkeypatch, 150, 2, True, 27)       # targetdim = 150 depth (19 steps: 4,4,4,4,3)


# ---------------------------------------------------------------- (iii) packed operand >= 4 GiB
def test_operand_over_4gib(monkeypatch):
    """D = 512 with mixed counts in the depth-2D form (PLDA_MIXED_VARIANT=1; the bucketed form of round 5 would need
    129 + 2 k-quads) packs the test side to 256 (+8 spare) k-quads x 16 B per row: 1.02 M
    rows make it 4.3 GB, beyond the kernel's 32-bit DMA offsets, so score_matrix_device scores two
    column blocks (the enrol side packed once).  Checked on every row x 4096 sampled columns (both
    blocks, the block seam and the tail columns included)."""
    import torch
    from oracle import plda_oracle_np as onp
    dev = torch.device("cuda", 0)
    d, m, nt = 512, 256, 1_017_000
    monkeypatch.setenv("PLDA_MIXED_VARIANT", "1")
    eng, psi = _engine(monkeypatch, None, d, 31)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    g = torch.Generator(device=dev); g.manual_seed(5)
    dV = torch.randn((nt, d), dtype=torch.float64, device=dev, generator=g)
    dU = torch.randn((m, d), dtype=torch.float64, device=dev, generator=g)
    n = np.random.default_rng(6).integers(1, 6, m).astype(np.int32)
    dn = torch.from_numpy(n).to(dev)
    out = torch.full((m, nt), float("nan"), dtype=torch.float32, device=dev)
    eng.score_matrix_dev(dU.data_ptr(), dn.data_ptr(), 0, m, dV.data_ptr(), nt, out.data_ptr(), nt)
    torch.cuda.synchronize()
    seam = ((1 << 32) - 1) // (264 * 16) // 256 * 256
    cols = np.unique(np.concatenate([np.random.default_rng(8).integers(0, nt, 4000), np.arange(nt - 96, nt),
                                     np.arange(seam - 64, seam + 64)]))
    tc = torch.from_numpy(cols).to(dev)
    ref = onp.llr_matrix(psi, dU.cpu().numpy(), n, dV[tc].cpu().numpy())
    got = out[:, tc].cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    assert (np.abs(got - ref) <= score_tol(ref)).all(), np.abs(got - ref).max()
    assert bool(torch.isfinite(out[:, ::4099]).all())


def test_alternating_tile_grids_do_not_stall_the_host(monkeypatch):
    """Round-4 review, weak 8: a change of the bt4 tile grid used to drain the stream and copy the tile table with two
    blocking hipMemcpy calls -- the sharded form alternates full blocks and a ragged tail block, a server scores varying
    M.  Since round 5 the table is built on the device and the last grids are cached: 40 calls alternating between two
    grids (and a third that evicts nothing) enqueue in a fraction of the time the GPU needs for them, and give the
    same bits as the same calls made one by one with a synchronisation after each."""
    import time
    import torch
    dev = torch.device("cuda", 0)
    d = 512
    eng, _ = _engine(monkeypatch, 40, d, 41)       # (the one-wave-per-SIMD kernel whatever the size: the product takes it from ~1 700 tiles)
    # (a stream of its own, as bench.py uses: launches on the legacy null stream are ordered against every other stream of
    #  the process, and what else the test process has in flight then shows up in the enqueue time)
    stream = torch.cuda.Stream(device=dev)
    eng.set_stream(stream.cuda_stream)
    g = torch.Generator(device=dev); g.manual_seed(11)
    shapes = [(8192, 8192), (8192, 8448), (8448, 8192)]
    mmax, nmax = 8448, 8448
    dU = torch.randn((mmax, d), dtype=torch.float64, device=dev, generator=g)
    dV = torch.randn((nmax, d), dtype=torch.float64, device=dev, generator=g)
    outs = [torch.empty((m, n), dtype=torch.float32, device=dev) for m, n in shapes]
    refs = []
    for (m, n), o in zip(shapes, outs):                       # warm: allocations, tables, code
        eng.score_matrix_dev(dU.data_ptr(), None, 2, m, dV.data_ptr(), n, o.data_ptr(), n)
        torch.cuda.synchronize()
        assert "bt4" in eng.score_last_kernel()
        refs.append(o.clone())
    for o in outs:
        o.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(40):
        (m, n), o = shapes[it % 3], outs[it % 3]
        eng.score_matrix_dev(dU.data_ptr(), None, 2, m, dV.data_ptr(), n, o.data_ptr(), n)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    for o, r in zip(outs, refs):
        assert torch.equal(o, r)
    # 40 GEMMs of >= 8192 x 8192 x 512 are ~20 ms of GPU work; their enqueue is launch overhead only
    assert t_enq < 0.5 * t_all, (t_enq, t_all)
    eng.set_stream(None)
