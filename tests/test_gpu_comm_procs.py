"""GPU: the PRODUCT sharding path (csrc/comm.hip) with MORE THAN ONE RANK -- several processes on the one GPU of
the test box, one handle each, the handle's collective table backed by the host transport
(plda_comm_init_host: pinned staging inside the library, gloo between the processes).  RCCL cannot be the
transport here (it refuses two ranks on one device); everything else is the code the multi-GPU bench runs:
comm.hip's own loops, block offsets, side stream, events, ragged tail, all_gather / all_gather_v / all_reduce
calls.  Every result is compared with the single-rank call on the same inputs: trials matrices bit-identical
(in-place gather, compact slab, compact + gather; uniform and mixed counts, z-normalised, a super-block larger than
the transport's staging chunk), the EER of the row-sharded matrix identical, z-norm statistics and the
speaker-sharded fit equal to the single-rank ones and bit-identical between the ranks (replicas).

The reference has no counterpart (one process: SURVEY.md section 2c); north_star: "the trials matrix shards
row-wise across the 8 GPUs ... all-gather to assemble scores"."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _set_model(eng, d, seed=3):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    eng.set_model(rng.random(d), q * (1.0 + rng.random(d))[:, None], np.sort(rng.random(d) * 4.0 + 0.05)[::-1].copy())


def _rank_main(rank, world, port, q, transport="host"):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from conftest import make_data
        from plda_amd import MPlda
        from plda_amd.sharding import (block_cyclic_rows, eer_sharded, fit_sharded, init_comm, local_row_index,
                                       score_matrix_sharded, speaker_shard, znorm_stats_sharded)
        dev = torch.device("cuda", 0)                  # every rank shares the one GPU of the test box
        torch.cuda.set_device(0)
        ok = {}
        eng, one = MPlda(0), MPlda(0)                  # `one`: the single-rank reference, no communicator
        assert init_comm(eng, transport=transport) == (world, rank)
        desc = eng.comm_describe()
        ok["describe"] = desc["transport"] == transport and desc["nranks"] == world and desc["rank"] == rank
        st = torch.cuda.current_stream(dev).cuda_stream
        eng.set_stream(st); one.set_stream(st)

        # ---------------- trials matrix: every output mode against the single call ----------------
        def trials(tag, d, m, nt, block, mixed, znorm):
            _set_model(eng, d); _set_model(one, d)
            rng = np.random.default_rng(5)             # same data on every rank (replicated inputs)
            dU = torch.from_numpy(rng.standard_normal((m, d))).to(dev)
            dV = torch.from_numpy(rng.standard_normal((nt, d))).to(dev)
            n = torch.from_numpy(rng.integers(1, 6, m).astype(np.int32)).to(dev) if mixed else None
            zm = torch.from_numpy(rng.standard_normal(m)).to(dev) if znorm else None
            zs = torch.from_numpy(0.5 + rng.random(m)).to(dev) if znorm else None
            nu = 0 if mixed else 3
            kw = dict(dzmean=zm.data_ptr() if znorm else None, dzstd=zs.data_ptr() if znorm else None)
            ref = torch.empty((m, nt), dtype=torch.float32, device=dev)
            one.score_matrix_dev(dU.data_ptr(), n.data_ptr() if mixed else None, nu, m, dV.data_ptr(), nt, ref.data_ptr(), nt, **kw)
            rows = local_row_index(m, world, rank, block, device=dev)
            mine = torch.zeros(m, dtype=torch.bool, device=dev); mine[rows] = True
            # (a) rows written in place into the full matrix, scores left sharded: only MY rows are touched
            a = torch.full((m, nt), float("nan"), dtype=torch.float32, device=dev)
            eng.score_matrix_sharded_dev(dU.data_ptr(), n.data_ptr() if mixed else None, nu, m, dV.data_ptr(), nt,
                                         a.data_ptr(), nt, block_rows=block, gather=False, **kw)
            torch.cuda.synchronize()
            ok[tag + ".inplace_sharded"] = bool(torch.equal(a[mine], ref[mine]) and torch.isnan(a[~mine]).all())
            # (b) the same with the in-place all-gather of every super-block (+ the ragged tail)
            b = torch.full((m, nt), float("nan"), dtype=torch.float32, device=dev)
            eng.score_matrix_sharded_dev(dU.data_ptr(), n.data_ptr() if mixed else None, nu, m, dV.data_ptr(), nt,
                                         b.data_ptr(), nt, block_rows=block, gather=True, **kw)
            torch.cuda.synchronize()
            ok[tag + ".inplace_gather"] = bool(torch.equal(b, ref))
            # (c) compact slab, (d) compact slab + assembled matrix
            loc, r2, full = score_matrix_sharded(eng, dU, n, dV, n_uniform=nu, gather=False, block_rows=block, zmean=zm, zstd=zs)
            torch.cuda.synchronize()
            ok[tag + ".compact"] = bool(full is None and torch.equal(r2, rows) and torch.equal(loc, ref[rows]))
            loc, r2, full = score_matrix_sharded(eng, dU, n, dV, n_uniform=nu, gather=True, block_rows=block, zmean=zm, zstd=zs)
            torch.cuda.synchronize()
            ok[tag + ".compact_gather"] = bool(torch.equal(loc, ref[rows]) and torch.equal(full, ref))
            return ref, loc, rows

        trials("ragged", 64, 2900, 1500, 256, False, False)       # 5 full super-blocks of 256 x world + a ragged tail
        if transport in ("host", "peer"):
            trials("mixed_znorm", 48, 1300, 700, 512, True, True)     # depth-2D GEMM, z-norm folded into the operands
            trials("tiny", 16, 300, 130, 256, False, False)           # fewer rows than one super-block: a rank may own no row
            ref, loc, rows = trials("chunked", 32, 9000, 4096, 4096, False, False)   # a 64 MiB block: two staging chunks
        else:
            ref, loc, rows = trials("mixed_znorm", 48, 1300, 700, 512, True, True)

        # ---------------- EER of the row-sharded matrix (compact slabs, nothing gathered) ----------------
        from plda_amd import eer
        rng = np.random.default_rng(17)
        es = torch.from_numpy(rng.integers(0, 40, ref.shape[0])).to(dev)
        ts = torch.from_numpy(rng.integers(0, 40, ref.shape[1])).to(dev)
        got = eer_sharded(eng, loc, es[rows].contiguous(), ts)
        want = eer.eer_from_matrix_dev(one, ref.data_ptr(), ref.shape[1], ref.shape[0], ref.shape[1], es.data_ptr(), ts.data_ptr())
        ok["eer"] = bool(np.array_equal(got, want))

        # ---------------- z-norm statistics sharded by model ----------------
        d = 40
        _set_model(eng, d); _set_model(one, d)
        rng = np.random.default_rng(8)
        models = torch.from_numpy(rng.standard_normal((77, d))).to(dev)      # 77 models: uneven slabs
        bkg = torch.from_numpy(rng.random((300, d))).to(dev)
        zr = torch.empty((2, 77), dtype=torch.float64, device=dev)
        one.znorm_stats_dev(bkg.data_ptr(), 300, 300, d, models.data_ptr(), 77, zr[0].data_ptr(), zr[1].data_ptr())
        zmean, zstd = znorm_stats_sharded(eng, bkg, models, num_examples=300)
        torch.cuda.synchronize()
        zg = torch.stack([zmean, zstd])
        ok["znorm"] = bool(torch.allclose(zg, zr, rtol=1e-12, atol=0))
        zall = [None] * world
        dist.all_gather_object(zall, zg.cpu().numpy().tobytes())
        ok["znorm_replicas"] = all(z == zall[0] for z in zall)

        # ---------------- fit sharded by speaker ----------------
        x, y = make_data(31, 2600, 48, 37, skew=True, scale_between=0.4)     # 37 speakers, unequal counts, uneven split
        ty = torch.from_numpy(y.astype(np.int64))
        mask = speaker_shard(ty, world, rank)
        k_local = fit_sharded(eng, torch.from_numpy(x[mask.numpy()]).to(dev), ty[mask], iters=6)
        dx = torch.from_numpy(x).to(dev); dy = ty.to(dev)
        one.fit_dev(dx.data_ptr(), 2600, 48, dy.data_ptr(), 37, 6)
        torch.cuda.synchronize()
        gm, rm = eng.get_model(), one.get_model()
        rel = lambda a_, b_: float(np.abs(a_ - b_).max() / np.abs(b_).max())   # noqa: E731
        ok["fit"] = bool(k_local == len(np.unique(y[mask.numpy()]))
                         and rel(gm["psi"], rm["psi"]) < 1e-9 and rel(gm["mean"], rm["mean"]) < 1e-13
                         and rel(gm["transform"].T @ gm["transform"], rm["transform"].T @ rm["transform"]) < 1e-9)
        fall = [None] * world
        dist.all_gather_object(fall, gm["transform"].tobytes() + gm["psi"].tobytes())
        ok["fit_replicas"] = all(f == fall[0] for f in fall)                 # bit-identical model on every rank

        tr = eng._comm_transport
        if transport == "peer":
            # the bootstrap table carried handles and rendezvous tokens only: no all_reduce, and a few hundred bytes per
            # call where the device data of these cases is megabytes (the pieces went GPU to GPU through IPC mappings)
            ok["transport_used"] = tr.calls["all_gather_v"] >= 20 and tr.calls["all_reduce"] == 0 and tr.last_error is None and \
                tr.calls["bytes"] <= 200 * world * tr.calls["all_gather_v"]
        else:
            ok["transport_used"] = tr.calls["all_gather_v"] >= 5 and tr.calls["all_reduce"] >= 5 and tr.last_error is None and \
                (transport == "host" or tr.calls["all_gather"] >= 2)
        eng.comm_destroy()
        ok["destroyed"] = eng.comm_info() == (1, 0)
        q.put((rank, ok, dict(tr.calls)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001 -- report instead of hanging the peers' collectives
        import traceback
        q.put((rank, {"exception: %s" % traceback.format_exc(): False}, {}))
        raise e


@pytest.mark.parametrize("world,transport", [(2, "host"), (3, "host"), (2, "custom"), (2, "peer"), (3, "peer")])
def test_sharded_entry_points_between_processes(world, transport):
    """transport "host": plda_comm_init_host (the library stages, gloo moves host buffers); "custom": plda_comm_init_custom
    with a device-level table supplied by the caller (plda_amd.sharding.TorchDeviceTransport); "peer": plda_comm_init_peer,
    direct writes into the other ranks' buffers through HIP IPC mappings (round 4; between processes on one device here,
    over xGMI between GPUs) -- gloo carries the IPC handles and the rendezvous only."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 38500 + (os.getpid() % 2000) + world + {"host": 0, "custom": 7, "peer": 13}[transport]
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted((q.get(timeout=600) for _ in procs), key=lambda x: x[0])
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    for rank, ok, calls in res:
        bad = [k for k, v in ok.items() if not v]
        assert not bad, (rank, bad)
    assert [p.exitcode for p in procs] == [0] * world


def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` with NO launcher around it (round-4 review: it used to run one rank and print
    n_gpus: 1): the script re-executes itself under torch.distributed.run.  Two ranks share this box's one GPU through the
    direct-write peer provider (gloo carries the IPC handles), a small C2-shaped problem; the line must say n_gpus: 2,
    carry the gather-inclusive keys at top level and pass its own oracle check on every rank."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--transport", "peer",
                        "--rows", "20000", "--steps", "2", "--warmup", "1", "--no-cpu"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["comm_nranks"] == 2 and j["transport"] == "peer" and j["multi_gpu"]["transport"] == "peer"
    assert "rccl_nranks" not in j and "rccl_nranks" not in j["multi_gpu"]      # that key only ever comes from ncclCommCount
    assert "NO collective" in j["config"]["value_excludes"]
    assert j["oracle_check"]["within_1e-4_on_every_rank"] is True
    assert j["gather_inclusive"]["gathered_blocks_bit_identical_to_their_owners"] is True
    assert j["gather_trials_per_s"] == j["gather_inclusive"]["value"] > 0 and j["gather_transport"] == "peer"


def test_bench_gpus_2_c4_strong_scaling_over_the_peer_provider():
    """The code path of BASELINE's 8-GPU configuration (C4: enrol models with n in 1..5, found once per call; strong scaling:
    ONE matrix split block-cyclically; a ragged tail -- 20 003 rows are no multiple of the 256-row block) on two ranks that
    share this box's GPU, self-launched: `bench.py --gpus 2 --config C4 --scaling strong --rows 20003` is DESIGN.md
    section 5's recipe with a smaller matrix."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--transport", "peer",
                        "--config", "C4", "--scaling", "strong", "--rows", "20003", "--speakers", "700", "--steps", "2",
                        "--warmup", "1", "--no-cpu"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["comm_nranks"] == 2 and "rccl_nranks" not in j
    assert j["config"]["enrol_models"] == 20003 and j["config"]["test_vectors"] == 20003
    assert j["roofline"]["flop_per_trial"] == 2 * (256 + 5 - 1)             # mixed counts: depth D + G - 1
    assert j["oracle_check"]["within_1e-4_on_every_rank"] is True
    assert j["gather_inclusive"]["gathered_blocks_bit_identical_to_their_owners"] is True
