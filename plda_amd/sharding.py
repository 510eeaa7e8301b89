"""plda_amd/sharding.py -- the path sharded across ranks (one process per GPU): trials by enrol row,
z-norm statistics by model, fit statistics by speaker.

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


def init_comm(engine, group=None, device=None):
    """Give `engine` (an MPlda) its RCCL communicator: rank 0 draws the unique id, torch.distributed
    (any backend -- it only carries 128 bytes) hands it round, every rank calls plda_comm_init.  After
    this the library's own sharded entry points (score_matrix_sharded_dev, fit_sharded_dev,
    znorm_stats_sharded_dev, plda_eer_matrix_comm_dev) run over RCCL without torch in the data path."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 1, 0
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    uid = [engine.comm_unique_id() if rank == 0 else None]
    if dist.get_backend(group) == "nccl":
        t = torch.tensor(list(uid[0]) if rank == 0 else [0] * 128, dtype=torch.uint8,
                         device=device if device is not None else torch.device("cuda", torch.cuda.current_device()))
        dist.broadcast(t, src=0, group=group)
        raw = bytes(t.cpu().tolist())
    else:
        dist.broadcast_object_list(uid, src=0, group=group)
        raw = uid[0]
    engine.comm_init(world, rank, raw)
    return world, rank


def block_cyclic_rows(m, world, rank, block_rows=4096):
    """Row ranges [(start, stop), ...] of `rank` under plda_score_matrix_sharded_dev's partition: block b of
    `block_rows` rows (rounded up to 256) belongs to rank b mod world; the rows left after the last full
    round of `world` blocks are dealt out once more in `world` equal smaller blocks."""
    block = -(-int(block_rows if block_rows > 0 else 4096) // 256) * 256
    sup = block * world
    nfull = m // sup
    out = [(s * sup + rank * block, s * sup + (rank + 1) * block) for s in range(nfull)]
    rem = m - nfull * sup
    if rem:
        tb = -(-(-(-rem // world)) // 256) * 256
        a = nfull * sup + rank * tb
        if a < m:
            out.append((a, min(m, a + tb)))
    return out


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


def speaker_shard(labels, world, rank):
    """Row mask of the speakers owned by `rank` (speaker id modulo world): a partition BY SPEAKER, so
    that every centroid is rank-local ("fit statistics")."""
    labels = torch.as_tensor(labels)
    return (labels.to(torch.int64) % int(world)) == int(rank)


def fit_sharded(stats_block, em_block, X_local, labels_local, iters=10, group=None):
    """PLDA fit with the statistics pass sharded by speaker.

    Every rank holds the rows of a disjoint set of speakers (`speaker_shard`).  Per rank:
    `stats_block(X_local, dense_labels, K_local) -> (means[K_local, D], counts[K_local] int64,
    scatter[D, D])` is the AddSamples pass (pldamodule.cpp:94-98) over its speakers.  Everything
    AddSamples accumulates is additive over speakers, so the one exchange is an all-reduce of the
    D x D offset scatter plus an all-gather of the centroids and counts (K*D*8 bytes: 8 MB at C2,
    41 MB at C3); the EM (pldamodule.cpp:102-106) is D x D work and runs as a replica on every rank
    from identical inputs: `em_block(means[K, D], counts[K], scatter, iters)`.
    Returns the global number of speakers.  A rank may own no speaker at all (K_local = 0).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    dev = X_local.device
    d = X_local.shape[1]
    labels_local = torch.as_tensor(labels_local)
    if labels_local.numel():
        _, dense = torch.unique(labels_local.to(torch.int64), sorted=True, return_inverse=True)
        k_local = int(dense.max().item()) + 1
        means, counts, scatter = stats_block(X_local, dense.to(dev), k_local)
    else:
        k_local = 0
        means = torch.zeros((0, d), dtype=torch.float64, device=dev)
        counts = torch.zeros((0,), dtype=torch.int64, device=dev)
        scatter = torch.zeros((d, d), dtype=torch.float64, device=dev)
    if world > 1:
        ks = torch.tensor([k_local], dtype=torch.int64, device=dev)
        all_k = [torch.empty_like(ks) for _ in range(world)]
        dist.all_gather(all_k, ks, group=group)
        all_k = [int(t.item()) for t in all_k]
        k_pad = max(max(all_k), 1)
        send_m = torch.zeros((k_pad, d), dtype=torch.float64, device=dev)
        send_c = torch.zeros((k_pad,), dtype=torch.int64, device=dev)
        send_m[:k_local] = means
        send_c[:k_local] = counts
        parts_m = [torch.empty_like(send_m) for _ in range(world)]
        parts_c = [torch.empty_like(send_c) for _ in range(world)]
        dist.all_gather(parts_m, send_m, group=group)
        dist.all_gather(parts_c, send_c, group=group)
        scatter = scatter.contiguous()
        dist.all_reduce(scatter, op=dist.ReduceOp.SUM, group=group)
        means = torch.cat([p[:k] for p, k in zip(parts_m, all_k)], dim=0).contiguous()
        counts = torch.cat([p[:k] for p, k in zip(parts_c, all_k)], dim=0).contiguous()
    k_global = int(means.shape[0])
    em_block(means, counts, scatter, int(iters))
    return k_global


def gpu_fit_blocks(engine):
    """(stats_block, em_block) over MPlda.fit_stats_dev / fit_em_dev for HBM-resident tensors."""
    if torch.cuda.is_available():
        engine.set_stream(torch.cuda.current_stream().cuda_stream)   # same stream as the torch ops around them

    def stats_block(X, dense, k):
        X = X.contiguous()
        lab = dense.to(torch.int64).contiguous()     # non-negative: same bits as the u64 the ABI reads
        n, d = X.shape
        means = torch.empty((k, d), dtype=torch.float64, device=X.device)
        counts = torch.empty((k,), dtype=torch.int64, device=X.device)
        scatter = torch.empty((d, d), dtype=torch.float64, device=X.device)
        engine.fit_stats_dev(X.data_ptr(), n, d, lab.data_ptr(), k)
        engine.fit_get_stats_dev(means.data_ptr(), counts.data_ptr(), scatter.data_ptr())
        return means, counts, scatter

    def em_block(means, counts, scatter, iters):
        engine.fit_em_dev(means.data_ptr(), counts.data_ptr(), means.shape[0], scatter.data_ptr(),
                          means.shape[1], iters)
    return stats_block, em_block


def eer_sharded(engine, scores_local, enrol_spk_local, test_spk, group=None):
    """Equal error rate of a ROW-SHARDED trials matrix without gathering it: every rank histograms its own
    slab (`scores_local` [m_local, Nt] float32 on its GPU, speaker ids of its enrol rows, all test speaker
    ids), and the counts are summed over the ranks between the three passes -- 3 x 32 KiB + 8 bytes of
    traffic for any number of trials.  Returns the 6-vector (threshold, FAR, FRR, EER, #targets,
    #impostors), identical on every rank and identical to the single-GPU result on the assembled matrix.
    A rank may own no row (m_local = 0)."""
    import ctypes as C
    import numpy as np
    from . import _native as N
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    on_gpu = world > 1 and dist.get_backend(group) == "nccl"
    dev = scores_local.device

    def allreduce(t, op):
        if world == 1:
            return t
        if on_gpu:
            g = t.to(dev)
            dist.all_reduce(g, op=op, group=group)
            return g.cpu()
        dist.all_reduce(t, op=op, group=group)
        return t

    CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint), C.POINTER(C.c_uint))

    def reduce(ctx, hist, below, above):
        try:
            if hist:
                a = np.ctypeslib.as_array(hist, shape=(4096,))
                t = allreduce(torch.from_numpy(a.view(np.int64).copy()), dist.ReduceOp.SUM)
                a[:] = t.numpy().view(np.uint64)
            else:
                below[0] = int(allreduce(torch.tensor([below[0]], dtype=torch.int64), dist.ReduceOp.MAX)[0])
                above[0] = int(allreduce(torch.tensor([above[0]], dtype=torch.int64), dist.ReduceOp.MIN)[0])
            return 0
        except Exception:       # never let an exception cross the C boundary
            return 1

    cb = CB(reduce)
    if scores_local.is_cuda:
        engine.set_stream(torch.cuda.current_stream(dev).cuda_stream)   # the slab was produced on torch's stream
    m, nt = scores_local.shape
    out = np.zeros(6)
    scores_local = scores_local.contiguous()
    N.check(engine._h, engine._lib.plda_eer_matrix_sharded_dev(
        engine._h, C.c_void_p(scores_local.data_ptr() if m else 0), nt, m, nt,
        C.c_void_p(enrol_spk_local.data_ptr() if m else 0), C.c_void_p(test_spk.data_ptr()), C.cast(cb, C.c_void_p),
        None, C.c_void_p(out.ctypes.data)))
    return out


def gpu_znorm_block(engine, dbkg, nb, din):
    """znorm_block over MPlda.znorm_stats_dev: cohort `dbkg` [nb, din] fp64 tensor resident on this GPU."""
    if dbkg.is_cuda:
        engine.set_stream(torch.cuda.current_stream(dbkg.device).cuda_stream)   # same stream as the torch ops around it

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
