// plda_amd/csrc/comm.hip -- the path sharded across the GPUs of one node, behind
// the C ABI: one process per GPU, one handle per process, RCCL (xGMI) inside the library.
//
// The reference has no parallelism of any kind (one process, one thread, GIL held: SURVEY.md
// section 2c); what shards is the build's own batched path:
//   * trials matrix  -- trial (i, j) needs only enrol row i, the replicated test set and the
//     replicated model: enrol rows are dealt out BLOCK-CYCLICALLY (blocks of `block_rows`), every
//     rank writes its blocks straight into their final place of the full [M, Nt] matrix (ld_out),
//     and -- only if the caller wants every rank to hold everything -- super-block s (R consecutive
//     blocks, one per rank) is assembled by ONE in-place all-gather on a side stream while
//     super-block s+1 is being scored.  No staging copies: the kernel's output buffer is the
//     collective's send AND receive buffer.
//   * z-norm statistics -- by model: every rank scans the whole cohort for its slab of models; one
//     exchange of [M] means and stds.
//   * fit statistics -- by speaker: AddSamples' accumulators are sums over speakers, so the D x D
//     offset scatter is all-reduced and the centroids / counts all-gathered; EM + GetOutput then run as
//     replicas ("replicas only", section 8e) from bit-identical inputs.
//   * EER of a sharded trials matrix -- the three histogram passes of eer.hip with the counters
//     summed over the ranks.
#include "common.hpp"

#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <vector>

namespace plda {

int score_matrix_device(plda_handle *h, const double *dU, const int32_t *dn, int n_uniform, int64_t M,
                        const double *dV, int64_t Nt, const double *dzmean, const double *dzstd,
                        float *dout, int64_t ld, bool reuse_packed_B);
int znorm_stats_device(plda_handle *h, const double *dbkg, int64_t Nb, int num_examples, int Din,
                       const double *dmodels, int64_t M, double *dmean, double *dstd);
int eer_matrix_device(plda_handle *h, const float *dscores, int64_t ld, int64_t M, int64_t Nt, const int64_t *despk,
                      const int64_t *dtspk, double *out,
                      int (*reduce)(void *, unsigned long long *, unsigned *, unsigned *), void *ctx);

static int nccl_fail(plda_handle *h, ncclResult_t r, const char *what, int line) {
  return fail(h, PLDA_E_HIP, "RCCL error %d (%s) at comm.hip:%d: %s", (int)r, ncclGetErrorString(r), line, what);
}
#define PLDA_NCCL(h, expr)                                              \
  do {                                                                  \
    ncclResult_t _r = (expr);                                           \
    if (_r != ncclSuccess) return nccl_fail((h), _r, #expr, __LINE__);  \
  } while (0)

static inline ncclComm_t comm_of(plda_handle *h) { return static_cast<ncclComm_t>(h->comm); }

// contiguous balanced partition of m items over `world` ranks
static inline void shard_range(int64_t m, int world, int rank, int64_t &b, int64_t &e) {
  const int64_t base = m / world, extra = m % world;
  b = rank * base + std::min<int64_t>(rank, extra);
  e = b + base + (rank < extra ? 1 : 0);
}

// all ranks contribute `counts[q]` elements at offset `offs[q]` of the same (replicated-layout) buffer:
// an all-gather with ragged pieces, as one group of broadcasts (in place)
static int allgatherv_inplace(plda_handle *h, void *buf, const std::vector<int64_t> &offs, const std::vector<int64_t> &counts,
                              size_t elem, hipStream_t st) {
  PLDA_NCCL(h, ncclGroupStart());
  for (int q = 0; q < h->comm_nranks; ++q) {
    if (counts[q] <= 0) continue;
    char *p = static_cast<char *>(buf) + (size_t)offs[q] * elem;
    const ncclResult_t r = ncclBroadcast(p, p, (size_t)counts[q] * elem, ncclChar, q, comm_of(h), st);
    if (r != ncclSuccess) { (void)ncclGroupEnd(); return nccl_fail(h, r, "ncclBroadcast", __LINE__); }
  }
  PLDA_NCCL(h, ncclGroupEnd());
  return PLDA_OK;
}

int comm_init(plda_handle *h, int nranks, int rank, const void *uid) {
  if (h->comm) return fail(h, PLDA_E_INVAL, "comm_init: this handle already has a communicator");
  if (nranks <= 0 || rank < 0 || rank >= nranks || !uid) return fail(h, PLDA_E_INVAL, "comm_init: bad argument");
  ncclUniqueId id;
  std::memcpy(&id, uid, sizeof(id));
  ncclComm_t c = nullptr;
  PLDA_NCCL(h, ncclCommInitRank(&c, nranks, id, rank));
  h->comm = c; h->comm_nranks = nranks; h->comm_rank = rank;
  PLDA_HIP(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
  for (auto &e : h->comm_ev) PLDA_HIP(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return PLDA_OK;
}

int comm_destroy(plda_handle *h) {
  if (!h->comm) return PLDA_OK;
  (void)hipStreamSynchronize(h->stream);
  (void)hipStreamSynchronize(h->comm_stream);
  (void)ncclCommDestroy(comm_of(h));
  for (auto &e : h->comm_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
  if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
  h->comm = nullptr; h->comm_stream = nullptr; h->comm_nranks = 1; h->comm_rank = 0;
  return PLDA_OK;
}

// ------------------------------------------------------------------------------------ trials matrix
int score_matrix_sharded_device(plda_handle *h, const double *dU, const int32_t *dn, int n_uniform, int64_t M,
                                const double *dV, int64_t Nt, const double *dzmean, const double *dzstd, float *dout,
                                int64_t ld, int64_t block_rows, int gather) {
  if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "score_matrix_sharded: model not fitted");
  if (M <= 0 || Nt <= 0) return PLDA_OK;
  if (!dU || !dV || !dout || ld < Nt) return fail(h, PLDA_E_INVAL, "score_matrix_sharded: bad argument");
  const int R = h->comm_nranks, me = h->comm_rank;   // (without a communicator: 1 / 0, or plda_comm_emulate's)
  if (block_rows <= 0) block_rows = 4096;
  block_rows = round_up(block_rows, 256);
  const int D = h->Dout;
  const bool zn = dzmean && dzstd;
  const int64_t super = block_rows * R;                       // rows of one super-block
  const int64_t nfull = M / super;                            // full super-blocks; the remainder is dealt out
  const int64_t rem = M - nfull * super;                      // again in (smaller) equal blocks, so that the
  const int64_t tail_block = rem ? round_up(ceil_div(rem, R), 256) : 0;   // last rows do not all land on rank 0
  const int64_t nsuper = nfull + (rem ? 1 : 0);
  const bool do_gather = gather && R > 1 && h->comm;
  bool packedB = false;
  for (int64_t s = 0; s < nsuper; ++s) {
    const int64_t s0 = s * super;
    const int64_t blk = s < nfull ? block_rows : tail_block;
    const int64_t r0 = s0 + (int64_t)me * blk;
    const int64_t cnt = std::max<int64_t>(0, std::min(blk, M - r0));
    if (cnt > 0) {
      PLDA_TRY(score_matrix_device(h, dU + r0 * D, dn ? dn + r0 : nullptr, n_uniform, cnt, dV, Nt, zn ? dzmean + r0 : nullptr,
                                   zn ? dzstd + r0 : nullptr, dout + r0 * ld, ld, packedB));
      packedB = true;                                          // the test side is packed once
    }
    if (!do_gather) continue;
    hipEvent_t ev = h->comm_ev[s & 3];
    PLDA_HIP(h, hipEventRecord(ev, h->stream));
    PLDA_HIP(h, hipStreamWaitEvent(h->comm_stream, ev, 0));
    if (s < nfull) {
      // full super-block: equal pieces, contiguous -> in-place all-gather
      PLDA_NCCL(h, ncclAllGather(dout + r0 * ld, dout + s0 * ld, (size_t)(block_rows * ld), ncclFloat, comm_of(h),
                                 h->comm_stream));
    } else {
      std::vector<int64_t> offs(R), counts(R);
      for (int q = 0; q < R; ++q) {
        const int64_t q0 = s0 + (int64_t)q * blk;
        offs[q] = q0 * ld;
        counts[q] = std::max<int64_t>(0, std::min(blk, M - q0)) * ld;
      }
      PLDA_TRY(allgatherv_inplace(h, dout, offs, counts, 4, h->comm_stream));
    }
  }
  if (do_gather) {
    // later work on the handle's stream sees the assembled matrix
    PLDA_HIP(h, hipEventRecord(h->comm_ev[4], h->comm_stream));
    PLDA_HIP(h, hipStreamWaitEvent(h->stream, h->comm_ev[4], 0));
  }
  h->last_M = M;
  return PLDA_OK;
}

// ------------------------------------------------------------------------------------ z-norm by model
int znorm_stats_sharded_device(plda_handle *h, const double *dbkg, int64_t Nb, int num_examples, int Din,
                               const double *dmodels, int64_t M, double *dmean, double *dstd) {
  const int R = h->comm_nranks, me = h->comm_rank;
  int64_t b, e;
  shard_range(M, R, me, b, e);
  if (e > b)
    PLDA_TRY(znorm_stats_device(h, dbkg, Nb, num_examples, Din, dmodels + b * h->Dout, e - b, dmean + b, dstd + b));
  if (R == 1 || !h->comm) return PLDA_OK;
  std::vector<int64_t> offs(R), counts(R);
  for (int q = 0; q < R; ++q) { int64_t qb, qe; shard_range(M, R, q, qb, qe); offs[q] = qb; counts[q] = qe - qb; }
  PLDA_TRY(allgatherv_inplace(h, dmean, offs, counts, 8, h->stream));
  PLDA_TRY(allgatherv_inplace(h, dstd, offs, counts, 8, h->stream));
  return PLDA_OK;
}

// ------------------------------------------------------------------------------------ fit by speaker
int fit_stats_device(plda_handle *h, const double *dX, int64_t N, int D, const uint64_t *dlabels, int64_t K);
int fit_em_device(plda_handle *h, int64_t K, int D, int iters);

int fit_sharded_device(plda_handle *h, const double *dX, int64_t N, int D, const uint64_t *dlabels, int64_t K, int iters) {
  const int R = h->comm_nranks, me = h->comm_rank;
  PLDA_TRY(fit_stats_device(h, dX, N, D, dlabels, K));        // means[K, D], counts[K], scatter[D, D] of MY speakers
  if (R == 1 || !h->comm) return fit_em_device(h, K, D, iters);
  const size_t DD = (size_t)D * D;
  // speaker counts of all ranks
  PLDA_HIP(h, h->w[6].reserve((size_t)R * 8));
  int64_t *dK = h->w[6].as<int64_t>();
  PLDA_HIP(h, hipMemcpyAsync(dK + me, &K, 8, hipMemcpyHostToDevice, h->stream));
  PLDA_NCCL(h, ncclAllGather(dK + me, dK, 1, ncclInt64, comm_of(h), h->stream));
  std::vector<int64_t> hK(R);
  PLDA_HIP(h, hipMemcpyAsync(hK.data(), dK, (size_t)R * 8, hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  std::vector<int64_t> off(R), cntM(R), offM(R);
  int64_t Kt = 0;
  for (int q = 0; q < R; ++q) { off[q] = Kt; Kt += hK[q]; }
  for (int q = 0; q < R; ++q) { offM[q] = off[q] * D; cntM[q] = hK[q] * D; }
  // merged statistics in rank order: means / counts gathered, scatter summed
  Tmp mm, mc;
  PLDA_HIP(h, mm.alloc((size_t)Kt * D * 8));
  PLDA_HIP(h, mc.alloc((size_t)Kt * 8));
  PLDA_HIP(h, hipMemcpyAsync(static_cast<double *>(mm.p) + offM[me], h->f_means.p, (size_t)K * D * 8, hipMemcpyDeviceToDevice, h->stream));
  PLDA_HIP(h, hipMemcpyAsync(static_cast<int64_t *>(mc.p) + off[me], h->f_counts.p, (size_t)K * 8, hipMemcpyDeviceToDevice, h->stream));
  PLDA_TRY(allgatherv_inplace(h, mm.p, offM, cntM, 8, h->stream));
  PLDA_TRY(allgatherv_inplace(h, mc.p, off, hK, 8, h->stream));
  PLDA_NCCL(h, ncclAllReduce(h->f_scatter.p, h->f_scatter.p, DD, ncclDouble, ncclSum, comm_of(h), h->stream));
  PLDA_HIP(h, h->f_means.reserve((size_t)Kt * D * 8));
  PLDA_HIP(h, h->f_counts.reserve((size_t)Kt * 8));
  PLDA_HIP(h, hipMemcpyAsync(h->f_means.p, mm.p, (size_t)Kt * D * 8, hipMemcpyDeviceToDevice, h->stream));
  PLDA_HIP(h, hipMemcpyAsync(h->f_counts.p, mc.p, (size_t)Kt * 8, hipMemcpyDeviceToDevice, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));               // the temporaries go out of scope
  h->fit_K = Kt;
  return fit_em_device(h, Kt, D, iters);
}

// ------------------------------------------------------------------------------------ EER, counters summed
struct EerCommCtx { plda_handle *h; };
static int eer_comm_reduce(void *vctx, unsigned long long *hist, unsigned *below, unsigned *above) {
  plda_handle *h = static_cast<EerCommCtx *>(vctx)->h;
  if (!h->comm || h->comm_nranks == 1) return 0;
  constexpr size_t NB = 2 * 2048;
  if (h->w[7].reserve(NB * 8 + 64) != hipSuccess) return 1;
  unsigned long long *d = h->w[7].as<unsigned long long>();
  if (hist) {
    if (hipMemcpyAsync(d, hist, NB * 8, hipMemcpyHostToDevice, h->stream) != hipSuccess) return 1;
    if (ncclAllReduce(d, d, NB, ncclUint64, ncclSum, comm_of(h), h->stream) != ncclSuccess) return 1;
    if (hipMemcpyAsync(hist, d, NB * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return 1;
    return hipStreamSynchronize(h->stream) == hipSuccess ? 0 : 1;
  }
  unsigned *du = reinterpret_cast<unsigned *>(d);
  if (hipMemcpyAsync(du, below, 4, hipMemcpyHostToDevice, h->stream) != hipSuccess) return 1;
  if (hipMemcpyAsync(du + 1, above, 4, hipMemcpyHostToDevice, h->stream) != hipSuccess) return 1;
  if (ncclAllReduce(du, du, 1, ncclUint32, ncclMax, comm_of(h), h->stream) != ncclSuccess) return 1;
  if (ncclAllReduce(du + 1, du + 1, 1, ncclUint32, ncclMin, comm_of(h), h->stream) != ncclSuccess) return 1;
  if (hipMemcpyAsync(below, du, 4, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return 1;
  if (hipMemcpyAsync(above, du + 1, 4, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return 1;
  return hipStreamSynchronize(h->stream) == hipSuccess ? 0 : 1;
}

int eer_matrix_comm_device(plda_handle *h, const float *dscores, int64_t ld, int64_t M, int64_t Nt, const int64_t *despk,
                           const int64_t *dtspk, double *out) {
  EerCommCtx ctx{h};
  return eer_matrix_device(h, dscores, ld, M, Nt, despk, dtspk, out, eer_comm_reduce, &ctx);
}

}  // namespace plda

using namespace plda;

extern "C" int plda_comm_unique_id(void *out, int64_t cap_bytes) {
  if (!out || cap_bytes < (int64_t)sizeof(ncclUniqueId)) return PLDA_E_CAPACITY;
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return PLDA_E_HIP;
  std::memcpy(out, &id, sizeof(id));
  return PLDA_OK;
}
