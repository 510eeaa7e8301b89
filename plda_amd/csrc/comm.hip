// plda_amd/csrc/comm.hip -- the path sharded across the GPUs of one node, behind
// the C ABI: one process per GPU, one handle per process.
//
// The reference has no parallelism of any kind (one process, one thread, GIL held: SURVEY.md
// section 2c); what shards is the build's own batched path:
//   * trials matrix  -- trial (i, j) needs only enrol row i, the replicated test set and the
//     replicated model: enrol rows are dealt out BLOCK-CYCLICALLY (blocks of `block_rows`); a rank
//     writes its blocks either back to back into a compact slab [M/R, Nt] or straight into their
//     final place of the full [M, Nt] matrix, and -- only if the caller wants every rank to hold
//     everything -- super-block s (R consecutive blocks, one per rank) is assembled by ONE all-gather
//     on a side stream while super-block s+1 is being scored.  No staging copies: the kernel's output
//     buffer is the collective's send buffer (and, in place, its receive buffer).
//   * z-norm statistics -- by model: every rank scans the whole cohort for its slab of models; one
//     exchange of [M] means and stds.
//   * fit statistics -- by speaker: AddSamples' accumulators are sums over speakers, so the D x D
//     offset scatter is all-reduced and the centroids / counts all-gathered; EM + GetOutput then run as
//     replicas ("replicas only", section 8e) from bit-identical inputs.
//   * EER of a sharded trials matrix -- the three histogram passes of eer.hip with the counters
//     summed over the ranks.
//
// Every collective goes through the handle's `plda_collectives` table (include/plda_hip.h).  Providers:
// RCCL over xGMI (production; librccl is dlopen'ed by plda_comm_init, so the .so does not depend on it),
// a host-staged adapter over two caller-supplied host operations (MPI / gloo / anything; also what lets
// several PROCESSES ON ONE GPU drive these very loops, events and offsets in the tests -- RCCL refuses two
// ranks on one device), or a caller-supplied device-level table.
#include "common.hpp"

#include <rccl/rccl.h>   // types and prototypes only; the functions are resolved with dlsym

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace plda {

int znorm_stats_device(plda_handle *h, const double *dbkg, int64_t Nb, int num_examples, int Din,
                       const double *dmodels, int64_t M, double *dmean, double *dstd);
int eer_matrix_device(plda_handle *h, const float *dscores, int64_t ld, int64_t M, int64_t Nt, const int64_t *despk,
                      const int64_t *dtspk, double *out,
                      int (*reduce)(void *, unsigned long long *, unsigned *, unsigned *), void *ctx);

// ------------------------------------------------------------------------------------ RCCL, resolved lazily
namespace {
struct Rccl {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  std::string err;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

// librccl of the process if one is already loaded (torch ships its own), else the ROCm installation's
const Rccl *rccl_api(std::string *why) {
  std::lock_guard<std::mutex> g(g_rccl_mu);
  if (g_rccl.lib) return &g_rccl;
  // PLDA_RCCL_LIB=<path> replaces the search list (a site-specific build; also how the not-found path is tested)
  std::vector<std::string> names;
  if (const char *one = std::getenv("PLDA_RCCL_LIB")) {
    names.push_back(one);
  } else {
    names = {"librccl.so.1", "librccl.so"};
    if (const char *rp = std::getenv("ROCM_PATH")) {
      names.push_back(std::string(rp) + "/lib/librccl.so.1");
      names.push_back(std::string(rp) + "/lib/librccl.so");
    }
    names.push_back("/opt/rocm/lib/librccl.so.1");
    names.push_back("/opt/rocm/lib/librccl.so");
  }
  void *lib = nullptr;
  std::string tried;
  for (const auto &n : names) {
    lib = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
    const char *e = dlerror();          // once per failed attempt: the call returns the message AND clears it
    tried += (tried.empty() ? "" : "; ") + n + ": " + (e ? e : "unknown dlopen error");
  }
  if (!lib) {
    if (why) *why = "librccl not found (" + tried + ")";
    return nullptr;
  }
  bool ok = true;
  auto sym = [&](auto &fp, const char *name) {
    fp = reinterpret_cast<std::remove_reference_t<decltype(fp)>>(dlsym(lib, name));
    if (!fp) { ok = false; if (why) *why = std::string("librccl lacks ") + name; }
  };
  sym(g_rccl.GetUniqueId, "ncclGetUniqueId");
  sym(g_rccl.CommInitRank, "ncclCommInitRank");
  sym(g_rccl.CommDestroy, "ncclCommDestroy");
  sym(g_rccl.AllGather, "ncclAllGather");
  sym(g_rccl.Broadcast, "ncclBroadcast");
  sym(g_rccl.AllReduce, "ncclAllReduce");
  sym(g_rccl.GroupStart, "ncclGroupStart");
  sym(g_rccl.GroupEnd, "ncclGroupEnd");
  sym(g_rccl.GetErrorString, "ncclGetErrorString");
  sym(g_rccl.CommCount, "ncclCommCount");
  sym(g_rccl.CommUserRank, "ncclCommUserRank");
  sym(g_rccl.CommCuDevice, "ncclCommCuDevice");
  sym(g_rccl.GetVersion, "ncclGetVersion");
  if (!ok) { dlclose(lib); return nullptr; }
  g_rccl.lib = lib;
  return &g_rccl;
}

struct RcclCtx {
  const Rccl *api;
  ncclComm_t comm;
  int nranks;
  plda_handle *h;
};

int rccl_err(RcclCtx *c, ncclResult_t r, const char *what) {
  fail(c->h, PLDA_E_HIP, "RCCL error %d (%s): %s", (int)r, c->api->GetErrorString(r), what);
  return 1;
}

int rccl_all_gather(void *vc, const void *dsend, void *drecv, int64_t bytes, void *st) {
  auto *c = static_cast<RcclCtx *>(vc);
  const ncclResult_t r = c->api->AllGather(dsend, drecv, (size_t)bytes, ncclChar, c->comm, static_cast<hipStream_t>(st));
  return r == ncclSuccess ? 0 : rccl_err(c, r, "ncclAllGather");
}

// ragged all-gather in place, as one group of broadcasts
int rccl_all_gather_v(void *vc, void *dbuf, const int64_t *offs, const int64_t *counts, void *st) {
  auto *c = static_cast<RcclCtx *>(vc);
  ncclResult_t r = c->api->GroupStart();
  if (r != ncclSuccess) return rccl_err(c, r, "ncclGroupStart");
  for (int q = 0; q < c->nranks; ++q) {
    if (counts[q] <= 0) continue;
    char *p = static_cast<char *>(dbuf) + offs[q];
    r = c->api->Broadcast(p, p, (size_t)counts[q], ncclChar, q, c->comm, static_cast<hipStream_t>(st));
    if (r != ncclSuccess) { (void)c->api->GroupEnd(); return rccl_err(c, r, "ncclBroadcast"); }
  }
  r = c->api->GroupEnd();
  return r == ncclSuccess ? 0 : rccl_err(c, r, "ncclGroupEnd");
}

int rccl_all_reduce(void *vc, void *dbuf, int64_t count, int32_t dtype, int32_t op, void *st) {
  auto *c = static_cast<RcclCtx *>(vc);
  const ncclDataType_t dt = dtype == PLDA_DT_F64 ? ncclDouble : dtype == PLDA_DT_U64 ? ncclUint64 : ncclUint32;
  const ncclRedOp_t ro = op == PLDA_OP_SUM ? ncclSum : op == PLDA_OP_MAX ? ncclMax : ncclMin;
  const ncclResult_t r = c->api->AllReduce(dbuf, dbuf, (size_t)count, dt, ro, c->comm, static_cast<hipStream_t>(st));
  return r == ncclSuccess ? 0 : rccl_err(c, r, "ncclAllReduce");
}

void rccl_destroy(void *vc) {
  auto *c = static_cast<RcclCtx *>(vc);
  if (c->comm) (void)c->api->CommDestroy(c->comm);
  delete c;
}

// ------------------------------------------------------------------------------------ host-staged adapter
// device-level table over two host operations: device data crosses a pinned bounce buffer in chunks of at most
// HS_CHUNK bytes per rank, so the staging memory is bounded whatever the size of a super-block
constexpr int64_t HS_CHUNK = (int64_t)32 << 20;

struct HostStage {
  plda_handle *h;
  plda_host_collectives t;
  int nranks, rank;
  void *pinned = nullptr;
  size_t cap = 0;
  std::vector<int64_t> hoff, hcnt;
  int reserve(size_t bytes) {
    if (bytes <= cap) return 0;
    if (pinned) (void)hipHostFree(pinned);
    pinned = nullptr; cap = 0;
    if (hipHostMalloc(&pinned, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return 1; }
    cap = bytes;
    return 0;
  }
};

int hs_fail(HostStage *c, const char *what) {
  fail(c->h, PLDA_E_HIP, "host-staged collective failed: %s", what);
  return 1;
}

int hs_all_gather_v(void *vc, void *dbuf, const int64_t *offs, const int64_t *counts, void *vst) {
  auto *c = static_cast<HostStage *>(vc);
  hipStream_t st = static_cast<hipStream_t>(vst);
  const int R = c->nranks, me = c->rank;
  int64_t maxc = 0;
  for (int q = 0; q < R; ++q) maxc = std::max(maxc, counts[q]);
  if (maxc <= 0) return 0;
  if (c->reserve((size_t)std::min(maxc, HS_CHUNK) * R)) return hs_fail(c, "pinned staging buffer");
  char *hb = static_cast<char *>(c->pinned);
  char *db = static_cast<char *>(dbuf);
  c->hoff.resize(R); c->hcnt.resize(R);
  for (int64_t c0 = 0; c0 < maxc; c0 += HS_CHUNK) {
    int64_t run = 0;
    for (int q = 0; q < R; ++q) {
      c->hcnt[q] = std::max<int64_t>(0, std::min(HS_CHUNK, counts[q] - c0));
      c->hoff[q] = run;
      run += c->hcnt[q];
    }
    if (c->hcnt[me] > 0 &&
        hipMemcpyAsync(hb + c->hoff[me], db + offs[me] + c0, (size_t)c->hcnt[me], hipMemcpyDeviceToHost, st) != hipSuccess)
      return hs_fail(c, "device -> host copy");
    if (hipStreamSynchronize(st) != hipSuccess) return hs_fail(c, "stream synchronisation");
    if (c->t.all_gather_v(c->t.ctx, hb, c->hoff.data(), c->hcnt.data()) != 0) return hs_fail(c, "all_gather_v callback");
    for (int q = 0; q < R; ++q) {
      if (q == me || c->hcnt[q] <= 0) continue;
      if (hipMemcpyAsync(db + offs[q] + c0, hb + c->hoff[q], (size_t)c->hcnt[q], hipMemcpyHostToDevice, st) != hipSuccess)
        return hs_fail(c, "host -> device copy");
    }
    if (hipStreamSynchronize(st) != hipSuccess) return hs_fail(c, "stream synchronisation");   // the buffer is reused
  }
  return 0;
}

int hs_all_gather(void *vc, const void *dsend, void *drecv, int64_t bytes, void *vst) {
  auto *c = static_cast<HostStage *>(vc);
  char *mine = static_cast<char *>(drecv) + (int64_t)c->rank * bytes;
  if (dsend != mine &&
      hipMemcpyAsync(mine, dsend, (size_t)bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(vst)) != hipSuccess)
    return hs_fail(c, "device -> device copy");
  std::vector<int64_t> offs(c->nranks), counts(c->nranks, bytes);
  for (int q = 0; q < c->nranks; ++q) offs[q] = (int64_t)q * bytes;
  return hs_all_gather_v(vc, drecv, offs.data(), counts.data(), vst);
}

int hs_all_reduce(void *vc, void *dbuf, int64_t count, int32_t dtype, int32_t op, void *vst) {
  auto *c = static_cast<HostStage *>(vc);
  hipStream_t st = static_cast<hipStream_t>(vst);
  const int64_t es = dtype == PLDA_DT_U32 ? 4 : 8;
  const int64_t per = HS_CHUNK / es;
  if (count <= 0) return 0;
  if (c->reserve((size_t)std::min(count, per) * es)) return hs_fail(c, "pinned staging buffer");
  char *db = static_cast<char *>(dbuf);
  for (int64_t e0 = 0; e0 < count; e0 += per) {
    const int64_t n = std::min(per, count - e0);
    if (hipMemcpyAsync(c->pinned, db + e0 * es, (size_t)(n * es), hipMemcpyDeviceToHost, st) != hipSuccess)
      return hs_fail(c, "device -> host copy");
    if (hipStreamSynchronize(st) != hipSuccess) return hs_fail(c, "stream synchronisation");
    if (c->t.all_reduce(c->t.ctx, c->pinned, n, dtype, op) != 0) return hs_fail(c, "all_reduce callback");
    if (hipMemcpyAsync(db + e0 * es, c->pinned, (size_t)(n * es), hipMemcpyHostToDevice, st) != hipSuccess)
      return hs_fail(c, "host -> device copy");
    if (hipStreamSynchronize(st) != hipSuccess) return hs_fail(c, "stream synchronisation");
  }
  return 0;
}

void hs_destroy(void *vc) {
  auto *c = static_cast<HostStage *>(vc);
  if (c->t.destroy) c->t.destroy(c->t.ctx);
  if (c->pinned) (void)hipHostFree(c->pinned);
  delete c;
}

// ------------------------------------------------------------------------------------ peer provider (HIP IPC, direct writes)
// The xGMI-native gather (SURVEY.md section 5 / 8e, round-3 review missing 4): xGMI is point to point, 7 links per GPU, so
// the fastest all-gather is not a ring (one link's ~153 GB/s) but every rank WRITING its piece straight into the other
// ranks' buffers, one copy stream per peer, all links busy at once (~1.07 TB/s out of a GPU).  Every rank opens the other
// ranks' buffers through HIP IPC memory handles (exchanged through the bootstrap HOST collectives the caller supplies --
// the same table plda_comm_init_host takes; 72 bytes per rank and call, nothing else crosses the host) and pushes with plain
// device-to-device copies.  What an RCCL rendezvous gives implicitly -- nobody writes into a buffer before its owner's
// stream has reached the collective, nobody continues before every piece has landed -- is done with sequence numbers in a
// small UNCACHED flag page per rank, itself IPC-mapped by every peer:
//   ready[q -> me]  written by q (a one-thread kernel on q's stream, through q's mapping of MY page) when q's stream has
//                   reached collective number n: q's piece is final and q's buffer may be written;
//   done[q -> me]   written by q behind its push into my buffer: q's piece has landed here.
// A copy stream spins (one thread, s_sleep) on ready[q] >= n before it pushes to q; the collective's stream spins on
// done[q] >= n for every q before it hands back to the caller.  Nothing blocks the host: the next super-block's scoring
// is enqueued while this one's pushes run.  A spin gives up after ~20 s of s_memrealtime and raises an error word that the
// next call reports (a peer that died must not hang the device).
// (Built first: HIP inter-process events.  ROCm's are a ring of 32 signals per event -- measured here: the test's ~30th
//  collective fails in hipStreamWaitEvent with `invalid argument`, with one event per peer and direction and every record
//  matched by exactly one wait too.  Sequence numbers in memory have no such horizon.)
// Works between processes on ONE device as well (how tests/test_gpu_comm_procs.py drives it).
constexpr int PEER_MAXR = 16;
struct PeerFlags {                              // one page per rank, written by the peers, read by its owner
  unsigned long long ready[PEER_MAXR], done[PEER_MAXR];
};

__global__ void peer_flag_set_kernel(unsigned long long *slot, unsigned long long v) {
  __threadfence_system();
  __hip_atomic_store(slot, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// `done`: only when no wait of this rank has given up -- a push that ran although its peer never signalled `ready` (the
// copy behind a timed-out wait cannot be cancelled) must not be announced as landed: the peer's own wait then gives up too,
// and both sides report the collective as failed (round-4 advisor)
__global__ void peer_flag_set_unless_kernel(unsigned long long *slot, unsigned long long v, const unsigned long long *err) {
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
  __threadfence_system();
  __hip_atomic_store(slot, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// waits until *slot >= v for every slot of the list (stride: PeerFlags::ready / ::done of peers q in mask)
__global__ void peer_flag_wait_kernel(unsigned long long *base, unsigned mask, unsigned long long v, unsigned long long *err) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
  for (int q = 0; q < PEER_MAXR; ++q) {
    if (!((mask >> q) & 1u)) continue;
    while (__hip_atomic_load(base + q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
      __builtin_amdgcn_s_sleep(32);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000000ull) {   // 20 s
        __hip_atomic_store(err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
      }
    }
  }
  __threadfence_system();
}

struct PeerCtx {
  plda_handle *h;
  plda_host_collectives t;
  int R, me;
  std::vector<hipStream_t> pstream;           // one copy stream per peer
  hipStream_t xs = nullptr;                   // the collective's own stream (the caller's may be the null stream)
  hipEvent_t ev_in = nullptr, ev_out = nullptr;   // local: caller's stream -> xs -> caller's stream
  hipEvent_t ev_mine = nullptr;               // local: xs has reached the collective
  std::vector<hipEvent_t> pcopy;              // local: my push to q has completed
  PeerFlags *flags = nullptr;                 // my page (uncached device memory)
  unsigned long long *err = nullptr;          // pinned host word a spin that gave up raises (read without any HIP call)
  std::vector<PeerFlags *> peer_flags;        // peer q's page, through its IPC mapping
  unsigned long long seq = 0;                 // collectives so far (the same on every rank: they are collective)
  // per peer: opened allocations.  An entry is the peer's allocation as (IPC handle, size, buffer id); the first entry
  // is the peer's flag page (pinned for the communicator's lifetime), the others are kept while they are among the
  // PEER_MAXMAPS most recently used and closed otherwise -- a service gathering into fresh buffers, or a peer whose
  // temporaries were freed, must not grow this list (nor keep the peer's freed memory pinned) without bound
  struct Map { hipIpcMemHandle_t handle; char *base; unsigned long long size, id, last; };
  std::vector<std::vector<Map>> maps;
  DevBuf scratch;                             // all_reduce: every rank's operand, rank-major
  std::vector<char> hbuf;
  std::vector<int64_t> hoff, hcnt;
};

int peer_fail(PeerCtx *c, const char *what, hipError_t e = hipSuccess) {
  if (e != hipSuccess) fail(c->h, PLDA_E_HIP, "peer collective failed: %s: %s", what, hipGetErrorString(e));
  else fail(c->h, PLDA_E_HIP, "peer collective failed: %s", what);
  (void)hipGetLastError();
  return 1;
}

// host all-gather of `bytes` per rank into c->hbuf (rank-major)
int peer_exchange(PeerCtx *c, const void *mine, int64_t bytes) {
  c->hbuf.assign((size_t)bytes * c->R, 0);
  std::memcpy(c->hbuf.data() + (size_t)c->me * bytes, mine, (size_t)bytes);
  c->hoff.resize(c->R); c->hcnt.resize(c->R);
  for (int q = 0; q < c->R; ++q) { c->hoff[q] = (int64_t)q * bytes; c->hcnt[q] = bytes; }
  return c->t.all_gather_v(c->t.ctx, c->hbuf.data(), c->hoff.data(), c->hcnt.data());
}

constexpr size_t PEER_MAXMAPS = 8;            // cached mappings per peer besides its flag page
int peer_map(PeerCtx *c, int q, const hipIpcMemHandle_t &hd, unsigned long long size, unsigned long long id, char **base) {
  auto &mp = c->maps[q];
  for (size_t k = 1; k < mp.size(); ++k) {
    if (std::memcmp(&mp[k].handle, &hd, sizeof(hd)) != 0) continue;
    if (mp[k].size == size && mp[k].id == id) { mp[k].last = c->seq; *base = mp[k].base; return 0; }
    // the same handle bytes for ANOTHER allocation (the owner freed the buffer and the runtime recycled the handle): the
    // cached mapping points at the old memory -- drop it (my earlier pushes through it first)
    (void)hipStreamSynchronize(c->pstream[q]);
    (void)hipIpcCloseMemHandle(mp[k].base);
    mp.erase(mp.begin() + (long)k);
    break;
  }
  while (mp.size() > PEER_MAXMAPS) {          // least recently used out
    size_t lru = 1;
    for (size_t k = 2; k < mp.size(); ++k) if (mp[k].last < mp[lru].last) lru = k;
    (void)hipStreamSynchronize(c->pstream[q]);
    (void)hipIpcCloseMemHandle(mp[lru].base);
    mp.erase(mp.begin() + (long)lru);
  }
  void *p = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return peer_fail(c, "hipIpcOpenMemHandle", e);
  mp.push_back({hd, static_cast<char *>(p), size, id, c->seq});
  *base = static_cast<char *>(p);
  return 0;
}

int peer_all_gather_v(void *vc, void *dbuf, const int64_t *offs, const int64_t *counts, void *vst) {
  auto *c = static_cast<PeerCtx *>(vc);
  hipStream_t caller = static_cast<hipStream_t>(vst);
  hipStream_t st = c->xs;
  const int R = c->R, me = c->me;
  // an earlier collective's spin that gave up?
  if (*static_cast<volatile unsigned long long *>(c->err) != 0) return peer_fail(c, "an earlier collective timed out waiting for a peer (20 s)");
  const unsigned long long n = ++c->seq;
  // onto the collective's own stream, behind everything the caller has enqueued
  if (hipEventRecord(c->ev_in, caller) != hipSuccess || hipStreamWaitEvent(st, c->ev_in, 0) != hipSuccess) return peer_fail(c, "stream hand-over");
  // this rank's allocation behind dbuf, as a handle the others can open, and dbuf's offset inside it
  struct Msg { hipIpcMemHandle_t handle; int64_t off; unsigned long long size, id; } mine;
  void *base = nullptr; size_t size = 0;
  hipError_t e = hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t *>(&base), &size, dbuf);
  if (e != hipSuccess) return peer_fail(c, "hipMemGetAddressRange", e);
  e = hipIpcGetMemHandle(&mine.handle, base);
  if (e != hipSuccess) return peer_fail(c, "hipIpcGetMemHandle", e);
  mine.off = static_cast<char *>(dbuf) - static_cast<char *>(base);
  mine.size = size;
  // the allocation's identity, so that an importer can tell a recycled handle from the buffer it has mapped (the runtime's
  // per-allocation id where it reports one; 0 otherwise: handle + size then are all there is)
  unsigned long long bid = 0;
  if (hipPointerGetAttribute(&bid, HIP_POINTER_ATTRIBUTE_BUFFER_ID, reinterpret_cast<hipDeviceptr_t>(base)) != hipSuccess) { (void)hipGetLastError(); bid = 0; }
  mine.id = bid;
  // ready: my piece is final and my buffer may be written, once my stream gets here -> every peer's page
  for (int q = 0; q < R; ++q)
    if (q != me) peer_flag_set_kernel<<<1, 1, 0, st>>>(&c->peer_flags[q]->ready[me], n);
  if ((e = hipEventRecord(c->ev_mine, st)) != hipSuccess) return peer_fail(c, "hipEventRecord(ready)", e);
  if (peer_exchange(c, &mine, sizeof(Msg)) != 0) return peer_fail(c, "bootstrap all_gather_v (handles)");
  std::vector<Msg> all(R);
  std::memcpy(all.data(), c->hbuf.data(), sizeof(Msg) * R);
  unsigned others = 0;
  for (int q = 0; q < R; ++q) {
    if (q == me) continue;
    others |= 1u << q;
    hipStream_t ps = c->pstream[q];
    if ((e = hipStreamWaitEvent(ps, c->ev_mine, 0)) != hipSuccess) return peer_fail(c, "hipStreamWaitEvent(own ready)", e);
    if (counts[me] > 0) {
      char *pb = nullptr;
      if (peer_map(c, q, all[q].handle, all[q].size, all[q].id, &pb) != 0) return 1;
      peer_flag_wait_kernel<<<1, 1, 0, ps>>>(c->flags->ready, 1u << q, n, c->err);     // q's buffer may be written
      if ((e = hipMemcpyAsync(pb + all[q].off + offs[me], static_cast<char *>(dbuf) + offs[me], (size_t)counts[me], hipMemcpyDeviceToDevice, ps)) != hipSuccess)
        return peer_fail(c, "push", e);
    }
    peer_flag_set_unless_kernel<<<1, 1, 0, ps>>>(&c->peer_flags[q]->done[me], n, c->err);        // my piece has landed at q
    if ((e = hipEventRecord(c->pcopy[q], ps)) != hipSuccess) return peer_fail(c, "hipEventRecord(push)", e);
    if ((e = hipStreamWaitEvent(st, c->pcopy[q], 0)) != hipSuccess) return peer_fail(c, "hipStreamWaitEvent(push)", e);
  }
  // every peer's piece has landed here
  peer_flag_wait_kernel<<<1, 1, 0, st>>>(c->flags->done, others, n, c->err);
  if (hipGetLastError() != hipSuccess) return peer_fail(c, "flag kernels");
  // and back: the caller's stream continues when the collective has completed
  if (hipEventRecord(c->ev_out, st) != hipSuccess || hipStreamWaitEvent(caller, c->ev_out, 0) != hipSuccess) return peer_fail(c, "stream hand-back");
  return 0;
}

int peer_all_gather(void *vc, const void *dsend, void *drecv, int64_t bytes, void *vst) {
  auto *c = static_cast<PeerCtx *>(vc);
  char *mine = static_cast<char *>(drecv) + (int64_t)c->me * bytes;
  if (dsend != mine) {
    const hipError_t e = hipMemcpyAsync(mine, dsend, (size_t)bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(vst));
    if (e != hipSuccess) return peer_fail(c, "device -> device copy", e);
  }
  std::vector<int64_t> offs(c->R), counts(c->R, bytes);
  for (int q = 0; q < c->R; ++q) offs[q] = (int64_t)q * bytes;
  return peer_all_gather_v(vc, drecv, offs.data(), counts.data(), vst);
}

template <typename T>
__global__ void peer_reduce_kernel(const T *__restrict__ all, int R, int64_t count, int op, T *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= count) return;
  T v = all[i];
  for (int q = 1; q < R; ++q) {               // rank order: the same bits on every rank
    const T w = all[(int64_t)q * count + i];
    v = op == PLDA_OP_SUM ? v + w : op == PLDA_OP_MAX ? (w > v ? w : v) : (w < v ? w : v);
  }
  out[i] = v;
}

// all-reduce = all-gather of the operands into a rank-major scratch + a local reduction in rank order (deterministic, and
// identical on every rank: the replicas of a sharded fit stay bit-identical)
int peer_all_reduce(void *vc, void *dbuf, int64_t count, int32_t dtype, int32_t op, void *vst) {
  auto *c = static_cast<PeerCtx *>(vc);
  hipStream_t st = static_cast<hipStream_t>(vst);
  if (count <= 0) return 0;
  const int64_t es = dtype == PLDA_DT_U32 ? 4 : 8, bytes = count * es;
  if (c->scratch.reserve((size_t)bytes * c->R) != hipSuccess) return peer_fail(c, "scratch allocation");
  if (peer_all_gather(vc, dbuf, c->scratch.p, bytes, vst) != 0) return 1;
  const unsigned grid = (unsigned)ceil_div(count, 256);
  if (dtype == PLDA_DT_F64) peer_reduce_kernel<double><<<grid, 256, 0, st>>>(c->scratch.as<double>(), c->R, count, op, static_cast<double *>(dbuf));
  else if (dtype == PLDA_DT_U64) peer_reduce_kernel<unsigned long long><<<grid, 256, 0, st>>>(c->scratch.as<unsigned long long>(), c->R, count, op, static_cast<unsigned long long *>(dbuf));
  else peer_reduce_kernel<unsigned><<<grid, 256, 0, st>>>(c->scratch.as<unsigned>(), c->R, count, op, static_cast<unsigned *>(dbuf));
  if (hipGetLastError() != hipSuccess) return peer_fail(c, "reduction kernel");
  // (the scratch is written by the peers of the NEXT call, which wait for this rank's `ready` of that call -- recorded on
  //  this stream behind the reduction above)
  return 0;
}

void peer_destroy(void *vc) {
  auto *c = static_cast<PeerCtx *>(vc);
  for (auto s : c->pstream) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
  if (c->xs) { (void)hipStreamSynchronize(c->xs); (void)hipStreamDestroy(c->xs); }
  for (auto &per : c->maps) for (auto &m : per) (void)hipIpcCloseMemHandle(m.base);
  for (auto e : c->pcopy) if (e) (void)hipEventDestroy(e);
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_out) (void)hipEventDestroy(c->ev_out);
  if (c->ev_mine) (void)hipEventDestroy(c->ev_mine);
  if (c->flags) (void)hipFree(c->flags);
  if (c->err) (void)hipHostFree(c->err);
  c->scratch.release();
  if (c->t.destroy) c->t.destroy(c->t.ctx);
  delete c;
}
}  // namespace

// ------------------------------------------------------------------------------------ installing a table
static int install(plda_handle *h, int nranks, int rank, const plda_collectives &t, int kind) {
  h->coll = t; h->comm_kind = kind; h->comm = true; h->comm_nranks = nranks; h->comm_rank = rank;
  PLDA_HIP(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
  for (auto &e : h->comm_ev) PLDA_HIP(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return PLDA_OK;
}

static int check_new(plda_handle *h, int nranks, int rank, const void *p, const char *fn) {
  if (h->comm) return fail(h, PLDA_E_INVAL, "%s: this handle already has a communicator", fn);
  if (nranks <= 0 || rank < 0 || rank >= nranks || !p) return fail(h, PLDA_E_INVAL, "%s: bad argument", fn);
  return PLDA_OK;
}

int comm_init(plda_handle *h, int nranks, int rank, const void *uid) {
  PLDA_TRY(check_new(h, nranks, rank, uid, "comm_init"));
  std::string why;
  const Rccl *api = rccl_api(&why);
  if (!api) return fail(h, PLDA_E_HIP, "comm_init: %s", why.c_str());
  ncclUniqueId id;
  std::memcpy(&id, uid, sizeof(id));
  ncclComm_t c = nullptr;
  const ncclResult_t r = api->CommInitRank(&c, nranks, id, rank);
  if (r != ncclSuccess) return fail(h, PLDA_E_HIP, "comm_init: ncclCommInitRank: %s", api->GetErrorString(r));
  auto *ctx = new RcclCtx{api, c, nranks, h};
  const plda_collectives t = {ctx, rccl_all_gather, rccl_all_gather_v, rccl_all_reduce, rccl_destroy};
  return install(h, nranks, rank, t, 1);
}

int comm_init_custom(plda_handle *h, int nranks, int rank, const plda_collectives *t) {
  PLDA_TRY(check_new(h, nranks, rank, t, "comm_init_custom"));
  if (!t->all_gather || !t->all_gather_v || !t->all_reduce) return fail(h, PLDA_E_INVAL, "comm_init_custom: the table lacks an operation");
  return install(h, nranks, rank, *t, 3);
}

int comm_init_host(plda_handle *h, int nranks, int rank, const plda_host_collectives *t) {
  PLDA_TRY(check_new(h, nranks, rank, t, "comm_init_host"));
  if (!t->all_gather_v || !t->all_reduce) return fail(h, PLDA_E_INVAL, "comm_init_host: the table lacks an operation");
  auto *ctx = new HostStage{h, *t, nranks, rank};
  const plda_collectives dt = {ctx, hs_all_gather, hs_all_gather_v, hs_all_reduce, hs_destroy};
  return install(h, nranks, rank, dt, 2);
}

int comm_init_peer(plda_handle *h, int nranks, int rank, const plda_host_collectives *t) {
  PLDA_TRY(check_new(h, nranks, rank, t, "comm_init_peer"));
  if (!t->all_gather_v) return fail(h, PLDA_E_INVAL, "comm_init_peer: the bootstrap table lacks all_gather_v");
  if (nranks > PEER_MAXR) return fail(h, PLDA_E_INVAL, "comm_init_peer: at most %d ranks", PEER_MAXR);
  auto *c = new PeerCtx();
  c->h = h; c->t = *t; c->R = nranks; c->me = rank;
  c->pstream.assign(nranks, nullptr); c->pcopy.assign(nranks, nullptr);
  c->peer_flags.assign(nranks, nullptr);
  c->maps.resize(nranks);
  auto bail = [&](const char *what, hipError_t e) {
    fail(h, PLDA_E_HIP, "comm_init_peer: %s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    c->t.destroy = nullptr;      // the caller keeps ownership of the bootstrap table on failure
    peer_destroy(c);
    return PLDA_E_HIP;
  };
  hipError_t e;
  if ((e = hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreateWithFlags", e);
  if ((e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreateWithFlags", e);
  if ((e = hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreateWithFlags", e);
  if ((e = hipEventCreateWithFlags(&c->ev_mine, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreateWithFlags", e);
  // the flag page: uncached, so that a peer's write is what the owner's next read sees
  void *fp = nullptr;
  if ((e = hipExtMallocWithFlags(&fp, 4096, hipDeviceMallocUncached)) != hipSuccess) return bail("hipExtMallocWithFlags(uncached)", e);
  c->flags = static_cast<PeerFlags *>(fp);
  if ((e = hipMemset(fp, 0, 4096)) != hipSuccess) return bail("hipMemset", e);
  void *ep = nullptr;
  if ((e = hipHostMalloc(&ep, 64, hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc", e);
  c->err = static_cast<unsigned long long *>(ep);
  *c->err = 0;
  hipIpcMemHandle_t mine;
  if ((e = hipIpcGetMemHandle(&mine, fp)) != hipSuccess) return bail("hipIpcGetMemHandle(flag page)", e);
  if (peer_exchange(c, &mine, sizeof(mine)) != 0) return bail("bootstrap all_gather_v (flag pages)", hipErrorUnknown);
  std::vector<hipIpcMemHandle_t> all(nranks);
  std::memcpy(all.data(), c->hbuf.data(), sizeof(hipIpcMemHandle_t) * nranks);
  for (int q = 0; q < nranks; ++q) {
    if (q == rank) continue;
    void *pp = nullptr;
    if ((e = hipIpcOpenMemHandle(&pp, all[q], hipIpcMemLazyEnablePeerAccess)) != hipSuccess) return bail("hipIpcOpenMemHandle(flag page)", e);
    c->peer_flags[q] = static_cast<PeerFlags *>(pp);
    c->maps[q].push_back({all[q], static_cast<char *>(pp), 4096, 0, 0});      // entry 0: pinned; closed by peer_destroy with the other mappings
    if ((e = hipStreamCreateWithFlags(&c->pstream[q], hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreateWithFlags", e);
    if ((e = hipEventCreateWithFlags(&c->pcopy[q], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreateWithFlags", e);
  }
  const plda_collectives dt = {c, peer_all_gather, peer_all_gather_v, peer_all_reduce, peer_destroy};
  return install(h, nranks, rank, dt, 4);
}

// A wait of the peer provider that gave up (a peer that died, or never reached the collective) raises a word in pinned host
// memory; the streams cannot be stopped from there, so whatever synchronises -- plda_synchronize, the end of a sharded fit,
// the communicator's destruction -- reports it: no sharded call's result may be trusted past a PLDA_E_HIP from here.
int comm_check(plda_handle *h) {
  if (!h->comm || h->comm_kind != 4) return PLDA_OK;
  auto *c = static_cast<PeerCtx *>(h->coll.ctx);
  if (c->err && *static_cast<volatile unsigned long long *>(c->err) != 0)
    return fail(h, PLDA_E_HIP, "peer collective failed: a wait for a peer gave up after 20 s; the data of the collectives since the last successful check is incomplete");
  return PLDA_OK;
}

int comm_destroy(plda_handle *h) {
  if (!h->comm) { h->comm_nranks = 1; h->comm_rank = 0; return PLDA_OK; }
  (void)hipStreamSynchronize(h->stream);
  if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
  const int late = comm_check(h);             // (reported after the teardown below)
  if (h->coll.destroy) h->coll.destroy(h->coll.ctx);
  h->coll = plda_collectives{nullptr, nullptr, nullptr, nullptr, nullptr};
  for (auto &e : h->comm_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
  if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
  h->comm = false; h->comm_kind = 0; h->comm_stream = nullptr; h->comm_nranks = 1; h->comm_rank = 0;
  return late;
}

int comm_describe(plda_handle *h, std::string &js) {
  int nranks = h->comm_nranks, rank = h->comm_rank, device = h->device, version = 0;
  const char *transport = !h->comm ? (h->comm_nranks > 1 ? "emulated" : "none")
                                   : h->comm_kind == 1 ? "rccl" : h->comm_kind == 2 ? "host" : h->comm_kind == 4 ? "peer" : "custom";
  if (h->comm && h->comm_kind == 1) {
    auto *c = static_cast<RcclCtx *>(h->coll.ctx);
    if (c->api->CommCount(c->comm, &nranks) != ncclSuccess || c->api->CommUserRank(c->comm, &rank) != ncclSuccess ||
        c->api->CommCuDevice(c->comm, &device) != ncclSuccess)
      return fail(h, PLDA_E_HIP, "comm_describe: the communicator does not answer");
    (void)c->api->GetVersion(&version);
  }
  char bus[64] = "";
  if (hipDeviceGetPCIBusId(bus, sizeof bus, h->device) != hipSuccess) { (void)hipGetLastError(); bus[0] = 0; }
  char buf[384];
  std::snprintf(buf, sizeof buf,
                "{\"transport\": \"%s\", \"nranks\": %d, \"rank\": %d, \"device\": %d, \"pci_bus_id\": \"%s\", \"rccl_version\": %d}",
                transport, nranks, rank, device, bus, version);
  js = buf;
  return PLDA_OK;
}

// a failed collective: the built-in providers have written their own message, a caller-supplied table has not
static int coll_failed(plda_handle *h, const char *what) {
  if (h->err.empty()) return fail(h, PLDA_E_HIP, "collective %s failed (the transport's callback returned non-zero)", what);
  return PLDA_E_HIP;
}
#define PLDA_COLL(h, expr, what)                      \
  do {                                                \
    (h)->err.clear();                                 \
    if ((expr) != 0) return coll_failed((h), what);   \
  } while (0)

// contiguous balanced partition of m items over `world` ranks
static inline void shard_range(int64_t m, int world, int rank, int64_t &b, int64_t &e) {
  const int64_t base = m / world, extra = m % world;
  b = rank * base + std::min<int64_t>(rank, extra);
  e = b + base + (rank < extra ? 1 : 0);
}

// ------------------------------------------------------------------------------------ trials matrix
// The partition (also exported as plda_shard_plan): super-block s = R consecutive blocks, block q of it -> rank q.
struct ShardPlan {
  int64_t block, super, nfull, rem, tail_block, nsuper;
  ShardPlan(int64_t M, int R, int64_t block_rows) {
    if (block_rows <= 0) block_rows = 4096;
    block = round_up(block_rows, 256);
    super = block * R;
    nfull = M / super;                                          // full super-blocks; the remainder is dealt out
    rem = M - nfull * super;                                    // again in (smaller) equal blocks, so that the
    tail_block = rem ? round_up(ceil_div(rem, R), 256) : 0;     // last rows do not all land on rank 0
    nsuper = nfull + (rem ? 1 : 0);
  }
  int64_t blk(int64_t s) const { return s < nfull ? block : tail_block; }
  // rows [r0, r0 + cnt) of rank q in super-block s (cnt may be 0 in the tail)
  void rows(int64_t s, int q, int64_t M, int64_t &r0, int64_t &cnt) const {
    r0 = s * super + (int64_t)q * blk(s);
    cnt = std::max<int64_t>(0, std::min(blk(s), M - r0));
  }
};

int shard_plan(int64_t M, int R, int rank, int64_t block_rows, int64_t *row_start, int64_t *row_count, int64_t cap,
               int64_t *nblocks, int64_t *local_rows) {
  if (M < 0 || R <= 0 || rank < 0 || rank >= R) return PLDA_E_INVAL;
  const ShardPlan p(M, R, block_rows);
  int64_t nb = 0, rows = 0;
  for (int64_t s = 0; s < p.nsuper; ++s) {
    int64_t r0, cnt;
    p.rows(s, rank, M, r0, cnt);
    if (cnt <= 0) continue;
    if (nb < cap) {
      if (row_start) row_start[nb] = r0;
      if (row_count) row_count[nb] = cnt;
    }
    ++nb; rows += cnt;
  }
  if (nblocks) *nblocks = nb;
  if (local_rows) *local_rows = rows;
  return (row_start || row_count) && nb > cap ? PLDA_E_CAPACITY : PLDA_OK;
}

// dlocal != nullptr: compact slab (this rank's blocks back to back); else this rank's rows of dfull.
// gather: assemble dfull on every rank (needs a communicator; ignored without one).
int score_matrix_sharded_device(plda_handle *h, const double *dU, const int32_t *dn, int n_uniform, int64_t M,
                                const double *dV, int64_t Nt, const double *dzmean, const double *dzstd, float *dlocal,
                                int64_t ld_local, float *dfull, int64_t ld_full, int64_t block_rows, int gather) {
  if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "score_matrix_sharded: model not fitted");
  if (M <= 0 || Nt <= 0) return PLDA_OK;
  if (!dU || !dV || (!dlocal && !dfull)) return fail(h, PLDA_E_INVAL, "score_matrix_sharded: bad argument");
  if ((dlocal && ld_local < Nt) || (dfull && ld_full < Nt)) return fail(h, PLDA_E_INVAL, "score_matrix_sharded: ld < Nt");
  const int R = h->comm_nranks, me = h->comm_rank;   // (without a communicator: 1 / 0, or plda_comm_emulate's)
  const ShardPlan p(M, R, block_rows);
  const int D = h->Dout;
  const bool zn = dzmean && dzstd;
  const bool do_gather = gather && dfull && R > 1 && h->comm;
  if (do_gather && dlocal && ld_local != ld_full)
    return fail(h, PLDA_E_INVAL, "score_matrix_sharded: gathering a compact slab needs ld_local == ld_full");
  bool packedB = false;
  int64_t lo = 0;                                               // rows of the compact slab written so far
  // mixed counts: the distinct counts of ALL M rows (replicated on every rank: every rank finds the same set), found once
  // per call -- one small device pass and one wait for the stream -- not per super-block
  CountSet cs;
  if (dn) PLDA_TRY(score_count_set_device(h, dn, M, &cs));
  for (int64_t s = 0; s < p.nsuper; ++s) {
    int64_t r0, cnt;
    p.rows(s, me, M, r0, cnt);
    float *mine = dlocal ? dlocal + lo * ld_local : dfull + r0 * ld_full;
    const int64_t ldm = dlocal ? ld_local : ld_full;
    if (cnt > 0) {
      PLDA_TRY(score_matrix_device(h, dU + r0 * D, dn ? dn + r0 : nullptr, n_uniform, cnt, dV, Nt, zn ? dzmean + r0 : nullptr,
                                   zn ? dzstd + r0 : nullptr, mine, ldm, packedB, dn ? &cs : nullptr));
      packedB = true;                                          // the test side is packed once
      lo += cnt;
    }
    if (!do_gather) continue;
    hipEvent_t ev = h->comm_ev[s & 3];
    PLDA_HIP(h, hipEventRecord(ev, h->stream));
    PLDA_HIP(h, hipStreamWaitEvent(h->comm_stream, ev, 0));
    const int64_t s0 = s * p.super;
    if (s < p.nfull) {
      // full super-block: R equal pieces, contiguous in dfull -> one all-gather (in place when `mine` lives there)
      PLDA_COLL(h, h->coll.all_gather(h->coll.ctx, mine, dfull + s0 * ld_full, p.block * ld_full * 4, h->comm_stream), "all_gather");
    } else {
      std::vector<int64_t> offs(R), counts(R);
      for (int q = 0; q < R; ++q) {
        int64_t q0, qc;
        p.rows(s, q, M, q0, qc);
        offs[q] = q0 * ld_full * 4;
        counts[q] = qc * ld_full * 4;
      }
      if (dlocal && cnt > 0)
        PLDA_HIP(h, hipMemcpyAsync(dfull + r0 * ld_full, mine, (size_t)(cnt * ld_full * 4), hipMemcpyDeviceToDevice, h->comm_stream));
      PLDA_COLL(h, h->coll.all_gather_v(h->coll.ctx, dfull, offs.data(), counts.data(), h->comm_stream), "all_gather_v");
    }
  }
  if (do_gather) {
    // later work on the handle's stream sees the assembled matrix
    PLDA_HIP(h, hipEventRecord(h->comm_ev[4], h->comm_stream));
    PLDA_HIP(h, hipStreamWaitEvent(h->stream, h->comm_ev[4], 0));
  } else if (gather && dfull && dlocal && R == 1) {
    PLDA_HIP(h, hipMemcpy2DAsync(dfull, (size_t)ld_full * 4, dlocal, (size_t)ld_local * 4, (size_t)Nt * 4, (size_t)M,
                                 hipMemcpyDeviceToDevice, h->stream));
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
  for (int q = 0; q < R; ++q) { int64_t qb, qe; shard_range(M, R, q, qb, qe); offs[q] = qb * 8; counts[q] = (qe - qb) * 8; }
  PLDA_COLL(h, h->coll.all_gather_v(h->coll.ctx, dmean, offs.data(), counts.data(), h->stream), "all_gather_v");
  PLDA_COLL(h, h->coll.all_gather_v(h->coll.ctx, dstd, offs.data(), counts.data(), h->stream), "all_gather_v");
  return PLDA_OK;
}

// ------------------------------------------------------------------------------------ fit by speaker
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
  PLDA_COLL(h, h->coll.all_gather(h->coll.ctx, dK + me, dK, 8, h->stream), "all_gather");
  std::vector<int64_t> hK(R);
  PLDA_HIP(h, hipMemcpyAsync(hK.data(), dK, (size_t)R * 8, hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  std::vector<int64_t> offC(R), cntC(R), offM(R), cntM(R);
  int64_t Kt = 0;
  for (int q = 0; q < R; ++q) {
    if (hK[q] <= 0) return fail(h, PLDA_E_INVAL, "fit_sharded: rank %d reports %lld speakers", q, (long long)hK[q]);
    offC[q] = Kt * 8; cntC[q] = hK[q] * 8;
    offM[q] = Kt * D * 8; cntM[q] = hK[q] * D * 8;
    Kt += hK[q];
  }
  // merged statistics in rank order: means / counts gathered, scatter summed
  // (persistent buffers: under the peer provider the other ranks map what a collective gathers into, and a temporary
  //  freed at scope exit would leave a mapping per fit behind on every peer)
  DevBuf &mm = h->comm_mm, &mc = h->comm_mc;
  PLDA_HIP(h, mm.reserve((size_t)Kt * D * 8));
  PLDA_HIP(h, mc.reserve((size_t)Kt * 8));
  PLDA_HIP(h, hipMemcpyAsync(static_cast<char *>(mm.p) + offM[me], h->f_means.p, (size_t)K * D * 8, hipMemcpyDeviceToDevice, h->stream));
  PLDA_HIP(h, hipMemcpyAsync(static_cast<char *>(mc.p) + offC[me], h->f_counts.p, (size_t)K * 8, hipMemcpyDeviceToDevice, h->stream));
  PLDA_COLL(h, h->coll.all_gather_v(h->coll.ctx, mm.p, offM.data(), cntM.data(), h->stream), "all_gather_v");
  PLDA_COLL(h, h->coll.all_gather_v(h->coll.ctx, mc.p, offC.data(), cntC.data(), h->stream), "all_gather_v");
  PLDA_COLL(h, h->coll.all_reduce(h->coll.ctx, h->f_scatter.p, (int64_t)DD, PLDA_DT_F64, PLDA_OP_SUM, h->stream), "all_reduce");
  PLDA_HIP(h, h->f_means.reserve((size_t)Kt * D * 8));
  PLDA_HIP(h, h->f_counts.reserve((size_t)Kt * 8));
  PLDA_HIP(h, hipMemcpyAsync(h->f_means.p, mm.p, (size_t)Kt * D * 8, hipMemcpyDeviceToDevice, h->stream));
  PLDA_HIP(h, hipMemcpyAsync(h->f_counts.p, mc.p, (size_t)Kt * 8, hipMemcpyDeviceToDevice, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  PLDA_TRY(comm_check(h));                                    // a peer wait that gave up: the merged statistics are incomplete
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
    if (h->coll.all_reduce(h->coll.ctx, d, (int64_t)NB, PLDA_DT_U64, PLDA_OP_SUM, h->stream) != 0) return 1;
    if (hipMemcpyAsync(hist, d, NB * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return 1;
    return hipStreamSynchronize(h->stream) == hipSuccess ? 0 : 1;
  }
  unsigned *du = reinterpret_cast<unsigned *>(d);
  if (hipMemcpyAsync(du, below, 4, hipMemcpyHostToDevice, h->stream) != hipSuccess) return 1;
  if (hipMemcpyAsync(du + 1, above, 4, hipMemcpyHostToDevice, h->stream) != hipSuccess) return 1;
  if (h->coll.all_reduce(h->coll.ctx, du, 1, PLDA_DT_U32, PLDA_OP_MAX, h->stream) != 0) return 1;
  if (h->coll.all_reduce(h->coll.ctx, du + 1, 1, PLDA_DT_U32, PLDA_OP_MIN, h->stream) != 0) return 1;
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
  const Rccl *api = rccl_api(nullptr);
  if (!api) return PLDA_E_HIP;
  ncclUniqueId id;
  if (api->GetUniqueId(&id) != ncclSuccess) return PLDA_E_HIP;
  std::memcpy(out, &id, sizeof(id));
  return PLDA_OK;
}

extern "C" int plda_shard_plan(int64_t M, int32_t nranks, int32_t rank, int64_t block_rows, int64_t *row_start,
                               int64_t *row_count, int64_t cap, int64_t *nblocks, int64_t *local_rows) {
  return shard_plan(M, nranks, rank, block_rows, row_start, row_count, cap, nblocks, local_rows);
}
