"""GPU: the library's own multi-GPU entry points (include/plda_hip.h, csrc/comm.hip) on the one GPU of the
test box, in ONE process.  plda_comm_emulate lets a handle play rank r of R without a communicator, so every
rank's shard can be produced in turn and checked to tile the single-call result exactly; a real RCCL
communicator is exercised at world size 1 (unique id -> comm_init -> sharded call with gather -> destroy).
The collectives themselves run between processes in tests/test_gpu_comm_procs.py (host transport; RCCL
refuses two ranks on one device, its cross-rank run is the driver's multi-GPU bench)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(d, seed=3):
    from plda_amd import MPlda
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    eng = MPlda(0)
    eng.set_model(rng.random(d), q * (1.0 + rng.random(d))[:, None], np.sort(rng.random(d) * 4.0 + 0.05)[::-1].copy())
    return eng


@pytest.mark.parametrize("world,block", [(2, 256), (3, 512), (8, 256)])
@pytest.mark.parametrize("mixed", [False, True])
def test_block_cyclic_shards_tile_the_matrix(world, block, mixed):
    import torch
    from plda_amd.sharding import block_cyclic_rows
    dev = torch.device("cuda", 0)
    d, m, nt = 64, 2900, 1500                      # ragged last super-block, ragged last block
    rng = np.random.default_rng(5)
    dU = torch.from_numpy(rng.standard_normal((m, d))).to(dev)
    dV = torch.from_numpy(rng.standard_normal((nt, d))).to(dev)
    n = torch.from_numpy(rng.integers(1, 6, m).astype(np.int32)).to(dev) if mixed else None
    eng = _engine(d)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    full = torch.empty((m, nt), dtype=torch.float32, device=dev)
    eng.score_matrix_dev(dU.data_ptr(), n.data_ptr() if mixed else None, 0 if mixed else 3, m, dV.data_ptr(), nt,
                         full.data_ptr(), nt)
    out = torch.full((m, nt), float("nan"), dtype=torch.float32, device=dev)
    covered = np.zeros(m, np.int32)
    for rank in range(world):
        eng.comm_emulate(world, rank)
        assert eng.comm_info() == (world, rank)
        mine = torch.full((m, nt), float("nan"), dtype=torch.float32, device=dev)
        eng.score_matrix_sharded_dev(dU.data_ptr(), n.data_ptr() if mixed else None, 0 if mixed else 3, m,
                                     dV.data_ptr(), nt, mine.data_ptr(), nt, block_rows=block, gather=True)
        torch.cuda.synchronize()
        rows = torch.isfinite(mine[:, 0]).cpu().numpy()
        want = np.zeros(m, bool)
        for a, b in block_cyclic_rows(m, world, rank, block):
            want[a:b] = True
        assert np.array_equal(rows, want)                       # exactly this rank's blocks, nothing else
        covered += rows
        out[torch.from_numpy(rows).to(dev)] = mine[torch.from_numpy(rows).to(dev)]
        # the same blocks back to back in a compact slab (plda_score_matrix_sharded_local_dev)
        nloc = int(rows.sum())
        slab = torch.full((nloc + 1, nt), float("nan"), dtype=torch.float32, device=dev)
        eng.score_matrix_sharded_local_dev(dU.data_ptr(), n.data_ptr() if mixed else None, 0 if mixed else 3, m,
                                           dV.data_ptr(), nt, slab.data_ptr(), nt, block_rows=block)
        torch.cuda.synchronize()
        assert torch.equal(slab[:nloc], full[torch.from_numpy(rows).to(dev)])
        assert torch.isnan(slab[nloc]).all()                      # nothing written past the slab
    eng.comm_emulate(1, 0)
    assert (covered == 1).all()                                  # every row scored by exactly one rank
    assert torch.equal(out, full)                                # and bit-identical to the single call


def test_real_communicator_world_1():
    import torch
    from plda_amd import MPlda
    dev = torch.device("cuda", 0)
    d, m, nt = 40, 700, 900
    rng = np.random.default_rng(6)
    eng = _engine(d)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    uid = MPlda.comm_unique_id()
    assert len(uid) == 128
    eng.comm_init(1, 0, uid)
    assert eng.comm_info() == (1, 0)
    desc = eng.comm_describe()
    assert desc["transport"] == "rccl" and desc["nranks"] == 1 and desc["rank"] == 0 and desc["rccl_version"] > 0
    dU = torch.from_numpy(rng.standard_normal((m, d))).to(dev)
    dV = torch.from_numpy(rng.standard_normal((nt, d))).to(dev)
    a = torch.empty((m, nt), dtype=torch.float32, device=dev)
    b = torch.empty((m, nt), dtype=torch.float32, device=dev)
    eng.score_matrix_dev(dU.data_ptr(), None, 2, m, dV.data_ptr(), nt, a.data_ptr(), nt)
    eng.score_matrix_sharded_dev(dU.data_ptr(), None, 2, m, dV.data_ptr(), nt, b.data_ptr(), nt, block_rows=256, gather=True)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    # sharded fit and z-norm statistics degenerate to the single-GPU calls
    x = rng.random((600, d)) + 0.4 * rng.standard_normal((30, d))[np.arange(600) % 30]
    y = (np.arange(600) % 30).astype(np.int64)
    dX = torch.from_numpy(x).to(dev); dy = torch.from_numpy(y).to(dev)
    e1, e2 = MPlda(0), eng
    e1.fit_dev(dX.data_ptr(), 600, d, dy.data_ptr(), 30, 3)
    e2.fit_sharded_dev(dX.data_ptr(), 600, d, dy.data_ptr(), 30, 3)
    m1, m2 = e1.get_model(), e2.get_model()
    assert np.array_equal(m1["psi"], m2["psi"]) and np.array_equal(m1["transform"], m2["transform"])
    models = torch.from_numpy(rng.standard_normal((50, d))).to(dev)
    bkg = torch.from_numpy(rng.random((300, d))).to(dev)
    za = torch.empty((2, 50), dtype=torch.float64, device=dev); zb = torch.empty((2, 50), dtype=torch.float64, device=dev)
    e2.znorm_stats_dev(bkg.data_ptr(), 300, 300, d, models.data_ptr(), 50, za[0].data_ptr(), za[1].data_ptr())
    e2.znorm_stats_sharded_dev(bkg.data_ptr(), 300, 300, d, models.data_ptr(), 50, zb[0].data_ptr(), zb[1].data_ptr())
    torch.cuda.synchronize()
    assert torch.allclose(za, zb, rtol=1e-12, atol=0)
    eng.comm_destroy()
    assert eng.comm_info() == (1, 0) and eng.comm_describe()["transport"] == "none"


@pytest.mark.parametrize("world", [2, 4])
def test_znorm_model_shards_tile(world):
    import torch
    dev = torch.device("cuda", 0)
    d, m = 32, 77
    rng = np.random.default_rng(8)
    eng = _engine(d)
    models = torch.from_numpy(rng.standard_normal((m, d))).to(dev)
    bkg = torch.from_numpy(rng.random((200, d))).to(dev)
    ref = torch.empty((2, m), dtype=torch.float64, device=dev)
    eng.znorm_stats_dev(bkg.data_ptr(), 200, 200, d, models.data_ptr(), m, ref[0].data_ptr(), ref[1].data_ptr())
    got = torch.full((2, m), float("nan"), dtype=torch.float64, device=dev)
    for rank in range(world):
        eng.comm_emulate(world, rank)
        eng.znorm_stats_sharded_dev(bkg.data_ptr(), 200, 200, d, models.data_ptr(), m, got[0].data_ptr(), got[1].data_ptr())
    eng.comm_emulate(1, 0)
    torch.cuda.synchronize()
    assert torch.allclose(got, ref, rtol=1e-12, atol=0)


def test_missing_rccl_reports_instead_of_crashing(tmp_path):
    """plda_comm_init on a machine without librccl (PLDA_RCCL_LIB points at nothing): PldaError with the loader's message
    per attempt (round-3 advisor: the second dlerror() call returned NULL and the message builder dereferenced it)."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import os
        os.environ["PLDA_RCCL_LIB"] = %r
        from plda_amd import MPlda
        from plda_amd._native import PldaError
        eng = MPlda(0)
        try:
            eng.comm_init(1, 0, bytes(128))
        except PldaError as e:
            print("PLDAERROR", e)
    """ % str(tmp_path / "no_such_librccl.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-400:])
    assert "PLDAERROR" in r.stdout and "librccl not found" in r.stdout and "no_such_librccl.so" in r.stdout, r.stdout
