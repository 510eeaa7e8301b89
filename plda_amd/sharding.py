"""plda_amd/sharding.py -- host-side glue of the multi-GPU path (one process per GPU).

The sharding itself -- trials by enrol row, z-norm statistics by model, fit statistics by speaker, EER
counters summed -- lives behind the C ABI (csrc/comm.hip: plda_score_matrix_sharded[_local]_dev,
plda_znorm_stats_sharded_dev, plda_fit_sharded_dev, plda_eer_matrix_comm_dev); this module only

  * gives an engine its communicator (`init_comm`): RCCL over xGMI (production), or any
    torch.distributed backend as a HOST transport (`TorchHostTransport`: the two host operations of
    `plda_host_collectives` over e.g. gloo -- what the multi-process tests on ONE GPU use, and what
    runs on CPU in the world-size-2 tests of the transport itself);
  * wraps the entry points for torch tensors (allocation of the compact slab, the row map).

The reference has no counterpart (one process, one thread: SURVEY.md section 2c).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _native as N


# ------------------------------------------------------------------------------------ transports
class TorchHostTransport(object):
    """`plda_host_collectives` over a torch.distributed process group (any backend that moves CPU tensors,
    e.g. gloo).  all_gather_v = one broadcast per non-empty piece, in place on the library's pinned
    buffer; all_reduce = dist.all_reduce on a view of it.  Exceptions never cross the C boundary: a
    callback that fails returns 1 and keeps the exception in `last_error`."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.last_error = None
        self.calls = {"all_gather_v": 0, "all_reduce": 0, "bytes": 0}
        self._agv = N.HOST_ALL_GATHER_V(self._all_gather_v)
        self._ar = N.HOST_ALL_REDUCE(self._all_reduce)
        self.table = N.HostCollectives(None, self._agv, self._ar, N.DESTROY_FN())

    def _src(self, q):
        return dist.get_global_rank(self.group, q) if self.group is not None else q

    def _all_gather_v(self, ctx, hbuf, offs, counts):
        try:
            world = self.world
            o = [int(offs[q]) for q in range(world)]
            c = [int(counts[q]) for q in range(world)]
            total = max(a + b for a, b in zip(o, c))
            if total <= 0:
                return 0
            buf = np.ctypeslib.as_array(C.cast(hbuf, C.POINTER(C.c_uint8)), shape=(total,))
            for q in range(world):
                if c[q] > 0:
                    dist.broadcast(torch.from_numpy(buf[o[q]:o[q] + c[q]]), src=self._src(q), group=self.group)
                    self.calls["bytes"] += c[q]
            self.calls["all_gather_v"] += 1
            return 0
        except Exception as e:       # noqa: BLE001 -- never let an exception cross the C boundary
            self.last_error = e
            return 1

    def _all_reduce(self, ctx, hbuf, count, dtype, op):
        try:
            count = int(count)
            if count <= 0:
                return 0
            rop = {N.PLDA_OP_SUM: dist.ReduceOp.SUM, N.PLDA_OP_MAX: dist.ReduceOp.MAX, N.PLDA_OP_MIN: dist.ReduceOp.MIN}[int(op)]
            if dtype == N.PLDA_DT_F64:
                a = np.ctypeslib.as_array(C.cast(hbuf, C.POINTER(C.c_double)), shape=(count,))
                dist.all_reduce(torch.from_numpy(a), op=rop, group=self.group)
            elif dtype == N.PLDA_DT_U64:
                # two's-complement sums wrap identically; counters stay far below 2^63, so max / min order too
                a = np.ctypeslib.as_array(C.cast(hbuf, C.POINTER(C.c_int64)), shape=(count,))
                dist.all_reduce(torch.from_numpy(a), op=rop, group=self.group)
            elif dtype == N.PLDA_DT_U32:
                a = np.ctypeslib.as_array(C.cast(hbuf, C.POINTER(C.c_uint32)), shape=(count,))
                t = torch.from_numpy(a.astype(np.int64))
                dist.all_reduce(t, op=rop, group=self.group)
                a[:] = t.numpy().astype(np.uint32)
            else:
                raise ValueError("unknown dtype %d" % dtype)
            self.calls["all_reduce"] += 1
            return 0
        except Exception as e:       # noqa: BLE001
            self.last_error = e
            return 1


class TorchDeviceTransport(object):
    """`plda_collectives` (the DEVICE-level table, plda_comm_init_custom) over a torch.distributed group: every
    operation synchronises the stream, moves the pieces through host memory with hipMemcpy and the group's CPU
    collectives, and returns with the data in place.  A reference implementation of the table for binders (a real
    one would use peer copies); the tests use it to cover plda_comm_init_custom between processes."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.last_error = None
        self.calls = {"all_gather": 0, "all_gather_v": 0, "all_reduce": 0}
        self._hip = C.CDLL("libamdhip64.so")
        self._hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self._hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self._ag = N.DEV_ALL_GATHER(self._all_gather)
        self._agv = N.DEV_ALL_GATHER_V(self._all_gather_v)
        self._ar = N.DEV_ALL_REDUCE(self._all_reduce)
        self.table = N.Collectives(None, self._ag, self._agv, self._ar, N.DESTROY_FN())

    def _src(self, q):
        return dist.get_global_rank(self.group, q) if self.group is not None else q

    def _d2h(self, dptr, nbytes):
        a = np.empty(nbytes, np.uint8)
        if nbytes and self._hip.hipMemcpy(a.ctypes.data, dptr, nbytes, 2) != 0:
            raise RuntimeError("hipMemcpy device -> host failed")
        return a

    def _h2d(self, dptr, a):
        if a.size and self._hip.hipMemcpy(dptr, a.ctypes.data, a.size, 1) != 0:
            raise RuntimeError("hipMemcpy host -> device failed")

    def _gather_v(self, dbuf, o, c, stream):
        if self._hip.hipStreamSynchronize(stream) != 0:
            raise RuntimeError("hipStreamSynchronize failed")
        for q in range(self.world):
            if c[q] <= 0:
                continue
            piece = self._d2h(dbuf + o[q], c[q]) if q == self.rank else np.empty(c[q], np.uint8)
            dist.broadcast(torch.from_numpy(piece), src=self._src(q), group=self.group)
            if q != self.rank:
                self._h2d(dbuf + o[q], piece)

    def _all_gather(self, ctx, dsend, drecv, nbytes, stream):
        try:
            nbytes = int(nbytes)
            if self._hip.hipStreamSynchronize(stream) != 0:
                raise RuntimeError("hipStreamSynchronize failed")
            mine = int(drecv) + self.rank * nbytes
            if int(dsend) != mine:
                self._h2d(mine, self._d2h(int(dsend), nbytes))
            self._gather_v(int(drecv), [q * nbytes for q in range(self.world)], [nbytes] * self.world, stream)
            self.calls["all_gather"] += 1
            return 0
        except Exception as e:       # noqa: BLE001
            self.last_error = e
            return 1

    def _all_gather_v(self, ctx, dbuf, offs, counts, stream):
        try:
            self._gather_v(int(dbuf), [int(offs[q]) for q in range(self.world)], [int(counts[q]) for q in range(self.world)], stream)
            self.calls["all_gather_v"] += 1
            return 0
        except Exception as e:       # noqa: BLE001
            self.last_error = e
            return 1

    def _all_reduce(self, ctx, dbuf, count, dtype, op, stream):
        try:
            if self._hip.hipStreamSynchronize(stream) != 0:
                raise RuntimeError("hipStreamSynchronize failed")
            np_t = {N.PLDA_DT_F64: np.float64, N.PLDA_DT_U64: np.int64, N.PLDA_DT_U32: np.uint32}[int(dtype)]
            rop = {N.PLDA_OP_SUM: dist.ReduceOp.SUM, N.PLDA_OP_MAX: dist.ReduceOp.MAX, N.PLDA_OP_MIN: dist.ReduceOp.MIN}[int(op)]
            a = self._d2h(int(dbuf), int(count) * np.dtype(np_t).itemsize).view(np_t)
            t = torch.from_numpy(a.astype(np.int64) if np_t is np.uint32 else a.copy())
            dist.all_reduce(t, op=rop, group=self.group)
            self._h2d(int(dbuf), np.ascontiguousarray(t.numpy().astype(np_t)).view(np.uint8))
            self.calls["all_reduce"] += 1
            return 0
        except Exception as e:       # noqa: BLE001
            self.last_error = e
            return 1


def init_comm(engine, group=None, device=None, transport="rccl"):
    """Give `engine` (an MPlda) its communicator; returns (world, rank).

    transport "rccl": rank 0 draws the unique id, torch.distributed (any backend -- it only carries 128 bytes)
    hands it round, every rank calls plda_comm_init; afterwards the library's sharded entry points run over
    RCCL without torch in the data path.  transport "host": the collectives travel through `group` itself
    (TorchHostTransport: the library stages, the group moves host buffers) -- no RCCL, works with several processes on
    one GPU.  transport "custom": the device-level table itself supplied from here (TorchDeviceTransport)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 1, 0
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if transport == "host":
        tr = TorchHostTransport(group)
        engine.comm_init_host(world, rank, tr.table)
        engine._comm_transport = tr          # callbacks live as long as the engine
        return world, rank
    if transport == "peer":
        if dist.get_backend(group) == "nccl":   # the bootstrap moves host bytes: a gloo group beside the nccl one
            group = dist.new_group(backend="gloo")
        tr = TorchHostTransport(group)          # IPC handles only; the data moves GPU to GPU
        engine.comm_init_peer(world, rank, tr.table)
        engine._comm_transport = tr
        return world, rank
    if transport == "custom":
        tr = TorchDeviceTransport(group)
        engine.comm_init_custom(world, rank, tr.table)
        engine._comm_transport = tr
        return world, rank
    if transport != "rccl":
        raise ValueError("transport must be 'rccl', 'host', 'peer' or 'custom'")
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


# ------------------------------------------------------------------------------------ the partitions
def block_cyclic_rows(m, world, rank, block_rows=4096):
    """Row ranges [(start, stop), ...] of `rank` under the library's partition of the trials matrix
    (plda_shard_plan, csrc/comm.hip): block b of `block_rows` rows (rounded up to 256) belongs to rank
    b mod world; the rows left after the last full round are dealt out once more in `world` equal smaller
    blocks.  Their order is the row order of the rank's compact slab."""
    from .libplda import MPlda
    return MPlda.shard_plan(m, world, rank, block_rows)


def local_row_index(m, world, rank, block_rows=4096, device=None):
    """int64 tensor of the global row numbers of this rank's compact slab, in slab order."""
    parts = [torch.arange(a, b, dtype=torch.int64, device=device) for a, b in block_cyclic_rows(m, world, rank, block_rows)]
    return torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=device)


def shard_rows(m, world, rank):
    """Contiguous balanced partition of m items (the z-norm models): (start, stop) for `rank`."""
    base, extra = divmod(int(m), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def speaker_shard(labels, world, rank):
    """Row mask of the speakers owned by `rank` (speaker id modulo world): a partition BY SPEAKER, so
    that every centroid is rank-local ("fit statistics")."""
    labels = torch.as_tensor(labels)
    return (labels.to(torch.int64) % int(world)) == int(rank)


# ------------------------------------------------------------------------------------ tensor wrappers
def _on_torch_stream(engine, dev):
    if dev.type == "cuda":
        engine.set_stream(torch.cuda.current_stream(dev).cuda_stream)   # same stream as the torch ops around the call


def score_matrix_sharded(engine, U, n, V, n_uniform=0, gather=False, block_rows=4096, zmean=None, zstd=None):
    """Row-sharded trials matrix on REPLICATED HBM-resident inputs (U [M, D], V [Nt, D] fp64 tensors; n int32 [M]
    or None with n_uniform): returns (local [m_local, Nt] float32 -- this rank's blocks back to back --, rows
    int64 [m_local] = their global row numbers, full [M, Nt] on every rank or None)."""
    dev = U.device
    _on_torch_stream(engine, dev)
    world, rank = engine.comm_info()
    m, nt = U.shape[0], V.shape[0]
    rows = local_row_index(m, world, rank, block_rows, device=dev)
    buf = torch.empty((max(rows.numel(), 1), nt), dtype=torch.float32, device=dev)   # (a rank may own no row)
    local = buf[:rows.numel()]
    full = torch.empty((m, nt), dtype=torch.float32, device=dev) if gather else None
    engine.score_matrix_sharded_local_dev(
        U.data_ptr(), n.data_ptr() if n is not None else None, n_uniform, m, V.data_ptr(), nt,
        buf.data_ptr(), nt, block_rows,
        dfull=full.data_ptr() if gather else None, ld_full=nt,
        dzmean=zmean.data_ptr() if zmean is not None else None, dzstd=zstd.data_ptr() if zstd is not None else None)
    return local, rows, full


def znorm_stats_sharded(engine, bkg, models, num_examples=0):
    """MPlda_norm sharded by MODEL: every rank scans the whole cohort `bkg` [Nb, Din] for its slab of the
    replicated `models` [M, Dout]; (mean[M], std[M]) on every rank."""
    dev = models.device
    _on_torch_stream(engine, dev)
    m = models.shape[0]
    mean = torch.empty(m, dtype=torch.float64, device=dev)
    std = torch.empty(m, dtype=torch.float64, device=dev)
    engine.znorm_stats_sharded_dev(bkg.data_ptr(), bkg.shape[0], num_examples, bkg.shape[1], models.data_ptr(), m,
                                   mean.data_ptr(), std.data_ptr())
    return mean, std


def fit_sharded(engine, X_local, labels_local, iters=10):
    """PLDA fit with the statistics pass over THIS rank's speakers (`speaker_shard`): labels are compacted to
    the local dense 0..K-1 on the device, plda_fit_sharded_dev does the rest (all-reduce of the scatter,
    all-gather of centroids and counts in rank order, replica EM).  Every rank must own at least one speaker."""
    dev = X_local.device
    _on_torch_stream(engine, dev)
    X_local = X_local.contiguous()
    _, dense = torch.unique(torch.as_tensor(labels_local).to(dev).to(torch.int64), sorted=True, return_inverse=True)
    dense = dense.contiguous()
    k_local = int(dense.max().item()) + 1
    engine.fit_sharded_dev(X_local.data_ptr(), X_local.shape[0], X_local.shape[1], dense.data_ptr(), k_local, iters)
    return k_local


def eer_sharded(engine, scores_local, enrol_spk_local, test_spk):
    """Equal error rate of a ROW-SHARDED trials matrix without gathering it: every rank histograms its own
    compact slab (`scores_local` [m_local, Nt] float32, int64 speaker ids of ITS rows, all test speaker ids);
    the counters are summed over the ranks by the handle's collectives between the three passes (3 x 32 KiB +
    8 bytes for any number of trials).  Identical 6-vector on every rank; a rank may own no row."""
    dev = test_spk.device
    _on_torch_stream(engine, dev)
    scores_local = scores_local.contiguous()
    m, nt = scores_local.shape
    return engine.eer_matrix_comm_dev(scores_local.data_ptr() if m else 0, nt, m, nt,
                                      enrol_spk_local.data_ptr() if m else 0, test_spk.data_ptr())
