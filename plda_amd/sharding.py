"""plda_amd/sharding.py -- row-sharded trials matrix across ranks (one process per GPU).

The trials matrix partitions by enrol row: trial (i, j) needs only enrol row i, the
replicated test set and the replicated model, so every rank scores its contiguous row
slab with no data-path collective.  Assembling the full [M, Nt] score
matrix on every rank is ONE all-gather of the row slabs (RCCL over xGMI when the
process group is "nccl"); it is optional (`gather=False` keeps scores sharded, which is
what shard-local consumers -- thresholding, EER counting, z-norm -- want) because its
volume, not the GEMM, bounds scaling: each rank must receive (R-1)/R of M*Nt*4 bytes
over its xGMI links.  When requested, the gather is issued slab by slab on a side
stream so that slab c travels while slab c+1 is being scored.

`score_block` is any callable (U_rows, n_rows, V) -> scores tensor [rows, Nt]; on the
GPU it wraps MPlda.score_matrix_dev, in the gloo/CPU tests it wraps the oracle.
"""
import torch
import torch.distributed as dist


def shard_rows(m, world, rank):
    """Contiguous balanced partition of m rows: (start, stop) for `rank`."""
    base, extra = divmod(int(m), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def padded_shard(m, world):
    """Rows per rank when every rank's slab is padded to the same size (for all-gather)."""
    return (int(m) + world - 1) // world


def score_matrix_sharded(score_block, U_local, n_local, V, m_global, gather=False, slab_rows=8192,
                         group=None, out_local=None):
    """Score this rank's rows; optionally all-gather the full matrix.

    U_local [m_local, D], n_local [m_local] int32 (or None), V [Nt, D] -- tensors on this
    rank's device.  Returns (scores_local [m_local, Nt] float32, gathered or None); with
    gather=True `gathered` is [M, Nt] on every rank (rows in global order).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    start, stop = shard_rows(m_global, world, rank)
    m_local = stop - start
    assert U_local.shape[0] == m_local, "U_local must hold exactly this rank's rows"
    nt = V.shape[0]
    dev = U_local.device
    if out_local is None:
        out_local = torch.empty((m_local, nt), dtype=torch.float32, device=dev)
    if not gather or world == 1:
        for r0 in range(0, m_local, slab_rows):
            r1 = min(m_local, r0 + slab_rows)
            out_local[r0:r1] = score_block(U_local[r0:r1], None if n_local is None else n_local[r0:r1], V)
        return out_local, (out_local if gather else None)

    # ---- gather: equal padded slabs, all_gather_into_tensor slab by slab, overlapped ----
    m_pad = padded_shard(m_global, world)
    gathered = torch.empty((world, m_pad, nt), dtype=torch.float32, device=dev)
    use_streams = dev.type == "cuda"
    side = torch.cuda.Stream(device=dev) if use_streams else None
    for r0 in range(0, m_pad, slab_rows):
        r1 = min(m_pad, r0 + slab_rows)
        send = torch.zeros((r1 - r0, nt), dtype=torch.float32, device=dev)
        v1 = min(r1, m_local)
        if v1 > r0:
            blk = score_block(U_local[r0:v1], None if n_local is None else n_local[r0:v1], V)
            send[: v1 - r0] = blk
            out_local[r0:v1] = blk
        recv = torch.empty((world, r1 - r0, nt), dtype=torch.float32, device=dev)
        if use_streams:
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
                gathered[:, r0:r1] = recv
            send.record_stream(side)
            recv.record_stream(side)
        else:
            parts = [torch.empty_like(send) for _ in range(world)]
            dist.all_gather(parts, send, group=group)
            for w, p in enumerate(parts):
                gathered[w, r0:r1] = p
    if use_streams:
        torch.cuda.current_stream(dev).wait_stream(side)
    # drop the padding rows and restore global order
    rows = []
    for w in range(world):
        s, e = shard_rows(m_global, world, w)
        rows.append(gathered[w, : e - s])
    return out_local, torch.cat(rows, dim=0)


def znorm_stats_sharded(znorm_block, models_local, m_global, group=None):
    """z-norm statistics sharded by MODEL: every rank holds the whole cohort
    and computes (mean, std) for its contiguous slab of models with `znorm_block(models) ->
    (mean[m], std[m])`; the only collective is an all-gather of the padded [M/R, 2] results
    (a few hundred KB at C5).  Returns (mean[M], std[M]) on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    start, stop = shard_rows(m_global, world, rank)
    assert models_local.shape[0] == stop - start, "models_local must hold exactly this rank's models"
    mean, std = znorm_block(models_local)
    if world == 1:
        return mean, std
    m_pad = padded_shard(m_global, world)
    send = torch.zeros((m_pad, 2), dtype=torch.float64, device=mean.device)
    send[: stop - start, 0] = mean
    send[: stop - start, 1] = std
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(parts, send, group=group)
    rows = []
    for w in range(world):
        s0, e0 = shard_rows(m_global, world, w)
        rows.append(parts[w][: e0 - s0])
    full = torch.cat(rows, dim=0)
    return full[:, 0].contiguous(), full[:, 1].contiguous()


def gpu_znorm_block(engine, dbkg, nb, din):
    """znorm_block over MPlda.znorm_stats_dev: cohort `dbkg` [nb, din] fp64 tensor resident on this GPU."""
    def fn(models):
        m = models.shape[0]
        mean = torch.empty(m, dtype=torch.float64, device=models.device)
        std = torch.empty(m, dtype=torch.float64, device=models.device)
        engine.znorm_stats_dev(dbkg.data_ptr(), nb, 0, din, models.data_ptr(), m, mean.data_ptr(), std.data_ptr())
        return mean, std
    return fn


def gpu_score_block(engine, n_uniform=0):
    """score_block over MPlda.score_matrix_dev for HBM-resident fp64 tensors."""
    import torch as _t
    engine.set_stream(_t.cuda.current_stream().cuda_stream)   # 0 = HIP's default stream, honoured as such

    def fn(U, n, V):
        out = torch.empty((U.shape[0], V.shape[0]), dtype=torch.float32, device=U.device)
        engine.score_matrix_dev(U.data_ptr(), n.data_ptr() if n is not None else None, n_uniform, U.shape[0],
                                V.data_ptr(), V.shape[0], out.data_ptr(), V.shape[0])
        return out
    return fn
