"""CPU: host-side logic of the shim (argument checks mirror pldamodule.cpp's
ValueErrors) and the row-sharding path on a 2-process gloo group with the oracle as
the per-slab scorer."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from plda_amd import libplda
from plda_amd.sharding import padded_shard, shard_rows


def test_feature_and_label_checks():
    x = np.random.default_rng(0).random((6, 3))
    assert libplda._features(x.astype(np.float32)).dtype == np.float64          # quirk Q12 superset
    assert libplda._features(np.asfortranarray(x)).flags["C_CONTIGUOUS"]
    with pytest.raises(ValueError, match="not floats"):
        libplda._features((x * 9).astype(np.int32))                               # pldamodule.cpp:59-62
    with pytest.raises(TypeError):
        libplda._features([[1.0, 2.0]])                                           # "O!" parse
    with pytest.raises(ValueError, match="not an unsigned"):
        libplda._labels(np.arange(6), 6)                                          # :55-58 / :133-136
    with pytest.raises(ValueError, match="not strings"):
        libplda._labels(np.array(["a"] * 6), 6, allow_strings_msg=True)           # :128-131
    with pytest.raises(ValueError, match="number of samples"):
        libplda._labels(np.arange(5, dtype=np.uint8), 6)
    y = libplda._labels(np.arange(6, dtype=np.uint16), 6)
    assert y.dtype == np.uint64 and y.flags["C_CONTIGUOUS"]


@pytest.mark.parametrize("m,world", [(10, 1), (10, 3), (7, 8), (100000, 8), (40000, 8), (5, 5)])
def test_shard_rows_partition(m, world):
    spans = [shard_rows(m, world, r) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == m
    for (a, b), (c, d) in zip(spans, spans[1:]):
        assert b == c and b >= a and d >= c
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1 and max(sizes) == padded_shard(m, world)


def _worker(rank, world, port, gather, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as ob
    from plda_amd.sharding import score_matrix_sharded, shard_rows as sr
    rng = np.random.default_rng(0)            # same data on every rank
    d, m, nt = 12, 37, 23
    psi = np.sort(rng.random(d) * 3)[::-1].copy()
    U, V = rng.standard_normal((m, d)), rng.standard_normal((nt, d))
    counts = rng.integers(1, 5, m).astype(np.int32)
    a, b = sr(m, world, rank)

    def block(Ub, nb, Vb):
        return torch.from_numpy(ob.score_block(psi, Ub.numpy(), nb.numpy(), Vb.numpy()).astype(np.float32))

    loc, full = score_matrix_sharded(block, torch.from_numpy(U[a:b]), torch.from_numpy(counts[a:b]),
                                     torch.from_numpy(V), m, gather=gather, slab_rows=8)
    ref = ob.score_block(psi, U, counts, V).astype(np.float32)
    ok = np.array_equal(loc.numpy(), ref[a:b])
    if gather:
        ok = ok and full is not None and np.array_equal(full.numpy(), ref)
    else:
        ok = ok and full is None
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def _znorm_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as ob
    from conftest import make_data
    from plda_amd.sharding import shard_rows as sr, znorm_stats_sharded
    x, y = make_data(61, 400, 10, 20, scale_between=0.5)
    model = ob.fit(x, y, 3)
    _, _, models = ob.transform_groups(model, x[:95], np.arange(95, dtype=np.uint64))   # 95 models: uneven slabs
    bkg = x[200:260]
    a, b = sr(95, world, rank)

    def block(mods):
        m, s = ob.norm(model, bkg, mods.numpy())
        return torch.from_numpy(m), torch.from_numpy(s)

    mean, std = znorm_stats_sharded(block, torch.from_numpy(models[a:b]), 95)
    rm, rs = ob.norm(model, bkg, models)
    q.put((rank, bool(np.array_equal(mean.numpy(), rm) and np.array_equal(std.numpy(), rs))))
    dist.barrier()
    dist.destroy_process_group()


def _fit_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import plda_oracle_np as onp
    from conftest import make_data
    from plda_amd.sharding import fit_sharded, speaker_shard
    x, y = make_data(71, 300, 8, 13, scale_between=0.7)      # 13 speakers, unequal counts, uneven split
    mask = speaker_shard(torch.from_numpy(y.astype(np.int64)), world, rank).numpy()
    got = {}

    def stats_block(X, dense, k):
        st = onp.stats(X.numpy(), dense.numpy())
        assert st["means"].shape[0] == k
        return (torch.from_numpy(st["means"]), torch.from_numpy(st["counts"].astype(np.int64)),
                torch.from_numpy(st["scatter"]))

    def em_block(means, counts, scatter, iters):
        means, counts = means.numpy(), counts.numpy()
        w = 1.0 / counts
        st = dict(means=means, counts=counts, scatter=scatter.numpy(), sum=(means * w[:, None]).sum(0),
                  class_weight=w.sum(), example_weight=float(len(counts)))
        d = means.shape[1]
        W, B = np.eye(d), np.eye(d)
        for _ in range(iters):
            W, B = onp.em_iter(st, W, B)
        got.update(onp.get_output(st, W, B))

    k = fit_sharded(stats_block, em_block, torch.from_numpy(x[mask]), torch.from_numpy(y[mask].astype(np.int64)), iters=4)
    ref = onp.fit(x, y, 4)
    T, R = got["transform"], ref["transform"]
    ok = (k == 13 and np.allclose(got["psi"], ref["psi"], rtol=1e-10, atol=1e-12)
          and np.allclose(T.T @ T, R.T @ R, rtol=1e-9, atol=1e-11) and np.allclose(got["mean"], ref["mean"], atol=1e-13))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_fit_gloo_world2(oracle):
    """fit statistics sharded by speaker: all-reduce of the scatter + all-gather of the centroids, then the
    replica EM, must reproduce the single-process fit on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_fit_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_sharded_znorm_gloo_world2(oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_znorm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


@pytest.mark.parametrize("gather", [False, True])
def test_sharded_trials_matrix_gloo_world2(oracle, gather):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (1 if gather else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, gather, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_block_cyclic_rows_tile_exactly():
    """The Python mirror of plda_score_matrix_sharded_dev's partition (csrc/comm.hip): every row belongs to exactly
    one rank, blocks are multiples of 256 rows, the remainder is dealt out in equal smaller blocks."""
    from plda_amd.sharding import block_cyclic_rows
    for m, world, block in [(1, 1, 256), (5, 4, 256), (2900, 2, 256), (2900, 3, 512), (2900, 8, 256),
                            (100000, 8, 4096), (100000, 2, 4096), (40000, 8, 4096), (40000, 4, 0), (4096 * 8, 8, 4096)]:
        cover = np.zeros(m, np.int32)
        sizes = []
        for r in range(world):
            rows = block_cyclic_rows(m, world, r, block)
            for a, b in rows:
                assert 0 <= a < b <= m and a % 256 == 0
                cover[a:b] += 1
            sizes.append(sum(b - a for a, b in rows))
        assert (cover == 1).all(), (m, world, block)
        blk = -(-(block if block > 0 else 4096) // 256) * 256
        assert max(sizes) - min(sizes) <= max(blk, 256), (m, world, block, sizes)   # balanced to within one block
