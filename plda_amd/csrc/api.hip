// plda_amd/csrc/api.hip -- the C ABI of libplda_hip.so (include/plda_hip.h).
// Host-pointer entry points stage through device buffers owned by the handle and
// call the *_device implementations; nothing here computes on the CPU.
#include "common.hpp"
#include "hostio.hpp"

#include <mutex>

#include <algorithm>
#include <cstdarg>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <stdexcept>

namespace plda {

static thread_local std::string g_create_err;

int fail(plda_handle *h, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) h->err = buf; else g_create_err = buf;
  return code;
}

int hip_fail(plda_handle *h, hipError_t e, const char *what, const char *file, int line) {
  (void)hipGetLastError();   // a failed call leaves its code behind: the next launch check would report it again
  return fail(h, PLDA_E_HIP, "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
}

int model_to_device(plda_handle *h) {
  const size_t Dout = h->Dout, Din = h->Din;
  PLDA_HIP(h, h->d_mean.reserve(Din * 8));
  PLDA_HIP(h, h->d_transform.reserve(Dout * Din * 8));
  PLDA_HIP(h, h->d_psi.reserve(Dout * 8));
  PLDA_HIP(h, h->d_offset.reserve(Dout * 8));
  PLDA_HIP(h, hipMemcpyAsync(h->d_mean.p, h->h_mean.data(), Din * 8, hipMemcpyHostToDevice, h->stream));
  PLDA_HIP(h, hipMemcpyAsync(h->d_transform.p, h->h_transform.data(), Dout * Din * 8, hipMemcpyHostToDevice, h->stream));
  PLDA_HIP(h, hipMemcpyAsync(h->d_psi.p, h->h_psi.data(), Dout * 8, hipMemcpyHostToDevice, h->stream));
  PLDA_HIP(h, hipMemcpyAsync(h->d_offset.p, h->h_offset.data(), Dout * 8, hipMemcpyHostToDevice, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  return PLDA_OK;
}

// implemented in score.hip / fit.hip
int score_pairs_device(plda_handle *h, const double *dU, const int32_t *dn, const double *dV,
                       const int64_t *de, const int64_t *dt, int64_t P, const double *dzmean,
                       const double *dzstd, double *dout, int64_t M = 0, const plda::CountSet *cs = nullptr);
int znorm_stats_device(plda_handle *h, const double *dbkg, int64_t Nb, int num_examples, int Din,
                       const double *dmodels, int64_t M, double *dmean, double *dstd);
int group_means_device(plda_handle *h, const double *dX, int64_t N, int D, const uint64_t *ddense,
                       int64_t Ku, double *dmeans, int32_t *dcounts32);
int compute_offset_device(plda_handle *h);
int group_by_label_device(plda_handle *h, const uint64_t *dlabels, int64_t N, uint32_t **perm_out, int **offsets_out,
                          uint64_t **uniq_out, int64_t *G_out);
int group_centroids_device(plda_handle *h, const double *dX, int64_t N, int D, const uint32_t *perm, const int *offsets,
                           int64_t G, double *dmeans, int32_t *dcounts32);
int eer_matrix_device(plda_handle *h, const float *dscores, int64_t ld, int64_t M, int64_t Nt, const int64_t *despk,
                      const int64_t *dtspk, double *out,
                      int (*reduce)(void *, unsigned long long *, unsigned *, unsigned *) = nullptr, void *ctx = nullptr);
int eer_lists_device(plda_handle *h, const float *dpos, int64_t np, const float *dneg, int64_t nn, double *out);
int det_matrix_device(plda_handle *h, const float *dscores, int64_t ld, int64_t M, int64_t Nt, const int64_t *despk, const int64_t *dtspk,
                      int npoints, double *far, double *frr, double *thresholds);
int det_lists_device(plda_handle *h, const float *dpos, int64_t np, const float *dneg, int64_t nn, int npoints, double *far, double *frr,
                     double *thresholds);
int score_eer_device(plda_handle *h, const double *dU, const int32_t *dn, int n_uniform, int64_t M, const double *dV, int64_t Nt,
                     const double *dzmean, const double *dzstd, const int64_t *despk, const int64_t *dtspk, double *out);
// comm.hip
int comm_init(plda_handle *h, int nranks, int rank, const void *uid);
int comm_init_custom(plda_handle *h, int nranks, int rank, const plda_collectives *t);
int comm_init_host(plda_handle *h, int nranks, int rank, const plda_host_collectives *t);
int comm_init_peer(plda_handle *h, int nranks, int rank, const plda_host_collectives *t);
int comm_destroy(plda_handle *h);
int comm_check(plda_handle *h);
int comm_describe(plda_handle *h, std::string &js);
int score_matrix_sharded_device(plda_handle *h, const double *dU, const int32_t *dn, int n_uniform, int64_t M,
                                const double *dV, int64_t Nt, const double *dzmean, const double *dzstd, float *dlocal,
                                int64_t ld_local, float *dfull, int64_t ld_full, int64_t block_rows, int gather);
int znorm_stats_sharded_device(plda_handle *h, const double *dbkg, int64_t Nb, int num_examples, int Din,
                               const double *dmodels, int64_t M, double *dmean, double *dstd);
int fit_sharded_device(plda_handle *h, const double *dX, int64_t N, int D, const uint64_t *dlabels, int64_t K, int iters);
int eer_matrix_comm_device(plda_handle *h, const float *dscores, int64_t ld, int64_t M, int64_t Nt, const int64_t *despk,
                           const int64_t *dtspk, double *out);

// the pinned ring + copy threads of the host-pointer entry points (hostio.hip), created on first use
static int host_pipe(plda_handle *h, HostPipe **out) {
  if (!h->hostpipe) {
    h->hostpipe = new HostPipe(default_host_threads());
  }
  PLDA_HIP(h, h->hostpipe->init());
  *out = h->hostpipe;
  return PLDA_OK;
}

// caller (pageable) memory -> a fresh device temporary; large arrays travel through the pinned ring
static int upload(plda_handle *h, Tmp &t, const void *src, size_t bytes) {
  PLDA_HIP(h, t.alloc(bytes));
  if (!bytes) return PLDA_OK;
  if (bytes >= ((size_t)4 << 20) && h->host_variant == 0) {
    HostPipe *hp = nullptr;
    PLDA_TRY(host_pipe(h, &hp));
    PLDA_HIP(h, hp->upload(h->stream, t.p, src, bytes));
    return PLDA_OK;
  }
  PLDA_HIP(h, hipMemcpyAsync(t.p, src, bytes, hipMemcpyHostToDevice, h->stream));
  return PLDA_OK;
}

// device -> caller (pageable) memory, contiguous; synchronises the stream
static int download(plda_handle *h, void *dst, const void *dsrc, size_t bytes) {
  if (bytes >= ((size_t)4 << 20) && h->host_variant == 0) {
    HostPipe *hp = nullptr;
    PLDA_TRY(host_pipe(h, &hp));
    advise_huge(dst, bytes);
    size_t i = 0;
    for (size_t off = 0; off < bytes; off += HostPipe::SLOT_BYTES, ++i) {
      const size_t n = std::min(HostPipe::SLOT_BYTES, bytes - off);
      const hipError_t e = hp->ship_slab(h->stream, i, static_cast<const char *>(dsrc) + off, n, static_cast<char *>(dst) + off, n, n, 1);
      if (e != hipSuccess) { (void)hp->finish(); return hip_fail(h, e, "ship_slab", __FILE__, __LINE__); }
    }
    PLDA_HIP(h, hp->finish());
    return PLDA_OK;
  }
  PLDA_HIP(h, hipMemcpyAsync(dst, dsrc, bytes, hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  return PLDA_OK;
}

static int set_device(plda_handle *h) {
  PLDA_HIP(h, hipSetDevice(h->device));
  return PLDA_OK;
}

// Plda::SmoothWithinClassCovariance: psi /= wc, transform rows *= wc^-1/2
__global__ void smooth_kernel(double *__restrict__ T, double *__restrict__ psi, int Dout, int Din, double f) {
  const int o = blockIdx.x;
  const double wc = 1.0 + f * psi[o];
  const double sc = 1.0 / sqrt(wc);
  for (int d = threadIdx.x; d < Din; d += blockDim.x) T[(size_t)o * Din + d] *= sc;
  __syncthreads();
  if (threadIdx.x == 0) psi[o] = psi[o] / wc;
}

}  // namespace plda

using namespace plda;

// one handle = one GPU + one stream: calls on the same handle from several threads are serialised
#define PLDA_LOCK(h) std::lock_guard<std::recursive_mutex> plda_lock_guard_((h)->mu)

// Nothing throws across the ABI: every entry point runs its body inside `guarded`, which turns a C++
// exception (std::bad_alloc / std::length_error from a host-side container, std::system_error from the
// mutex) into a status code + plda_last_error text.  (The reference lets Kaldi assertions abort the
// interpreter: pldamodule.cpp has no try/catch.)
namespace {
int fail_quiet(plda_handle *h, int code, const char *fn, const char *what) noexcept {
  try { return fail(h, code, "%s: %s", fn, what); } catch (...) { return code; }
}
template <typename F> int guarded(plda_handle *h, const char *fn, F &&body) noexcept {
  try { return body(); }
  catch (const std::bad_alloc &) { return fail_quiet(h, PLDA_E_HIP, fn, "out of host memory"); }
  catch (const std::exception &e) { return fail_quiet(h, PLDA_E_HIP, fn, e.what()); }
  catch (...) { return fail_quiet(h, PLDA_E_HIP, fn, "unknown C++ exception"); }
}
}  // namespace

extern "C" {

int plda_abi_version(void) { return 2; }
int plda_build_flags(void) { return PLDA_DIAG ? 1 : 0; }

int plda_create(int device, plda_handle **out) {
  return guarded(nullptr, "plda_create", [&]() -> int {
    if (!out) return fail(nullptr, PLDA_E_INVAL, "plda_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
      return fail(nullptr, PLDA_E_HIP, "plda_create: no HIP device available (%s); this engine has no CPU fallback",
                  e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= count) return fail(nullptr, PLDA_E_INVAL, "plda_create: device %d out of range [0,%d)", device, count);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(nullptr, PLDA_E_HIP, "plda_create: hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(nullptr, PLDA_E_HIP, "plda_create: device %d is %s; libplda_hip is built for gfx950 only", device, prop.gcnArchName);
    plda_handle *h = new (std::nothrow) plda_handle();
    if (!h) return fail(nullptr, PLDA_E_HIP, "plda_create: out of host memory");
    h->device = device;
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) {
      int n = 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n > 0) h->num_cus = n;
    }
    if (e != hipSuccess) { delete h; return fail(nullptr, PLDA_E_HIP, "plda_create: %s", hipGetErrorString(e)); }
    h->stream = h->own_stream;
    if (const char *v = std::getenv("PLDA_GEMM_VARIANT")) h->gemm_variant = std::atoi(v);
#if !PLDA_DIAG
    {   // measurement arms (garbage scores or clock stamps): not in this build
      static const int diag_only[] = {1, 2, 3, 4, 9, 10, 11, 12, 31, 33, 34, 35, 36, 37, 41, 44, 45, 46, 47, 54, 58, 62, 63};
      for (int d : diag_only)
        if (h->gemm_variant == d) {
          delete h;
          return fail(nullptr, PLDA_E_INVAL, "plda_create: PLDA_GEMM_VARIANT=%d is a measurement arm of the diagnostic build (PLDA_DIAG=1, "
                                             "libplda_hip_diag.so); this library does not contain it", d);
        }
    }
#endif
    if (const char *v = std::getenv("PLDA_PREP_VARIANT")) h->prep_variant = std::atoi(v);
    if (const char *v = std::getenv("PLDA_MIXED_VARIANT")) h->mixed_variant = std::atoi(v);
    if (const char *v = std::getenv("PLDA_SCORE_DTYPE")) {
      if (std::strcmp(v, "bf16x3") == 0) h->score_dtype = 1;
      else if (std::strcmp(v, "f32") != 0 && *v) { delete h; return fail(nullptr, PLDA_E_INVAL, "plda_create: PLDA_SCORE_DTYPE=%s (f32 or bf16x3)", v); }
    }
    if (const char *v = std::getenv("PLDA_EM_VARIANT")) h->em_variant = std::atoi(v);
    if (const char *v = std::getenv("PLDA_JACOBI_VARIANT")) h->jacobi_variant = std::atoi(v);
    if (const char *v = std::getenv("PLDA_GEMM64_VARIANT")) h->gemm64_variant = std::atoi(v);
    if (const char *v = std::getenv("PLDA_EIG_VARIANT")) h->eig_variant = std::atoi(v);
    if (const char *v = std::getenv("PLDA_EIG_DEBUG")) h->eig_debug = std::atoi(v);
    if (const char *v = std::getenv("PLDA_TRANSFORM_VARIANT")) h->transform_variant = std::atoi(v);
    if (const char *v = std::getenv("PLDA_SORT_VARIANT")) h->sort_variant = std::atoi(v);
    if (const char *v = std::getenv("PLDA_ZNORM_VARIANT")) h->znorm_variant = std::atoi(v);
    if (const char *v = std::getenv("PLDA_EER_VARIANT")) h->eer_variant = std::atoi(v);
    if (const char *v = std::getenv("PLDA_EER_SLAB_ROWS")) h->eer_slab_rows = std::atoll(v);
    if (const char *v = std::getenv("PLDA_HIP_TRACE")) h->trace_on = h->trace_print = std::atoi(v) != 0;
    if (const char *v = std::getenv("PLDA_HOST_VARIANT")) h->host_variant = std::atoi(v);
    if (const char *v = std::getenv("PLDA_SWEEP_VARIANT")) h->sweep_variant = std::atoi(v);
    *out = h;
    return PLDA_OK;
  });
}

static int trace_summary(plda_handle *h, std::string &out, bool reset);

int plda_destroy(plda_handle *h) {
  return guarded(h, "plda_destroy", [&]() -> int {
    if (!h) return PLDA_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->trace_print && h->trace_used) {   // PLDA_HIP_TRACE=1: the summary goes to stderr when the handle dies
      std::string js;
      if (trace_summary(h, js, true) == PLDA_OK) std::fprintf(stderr, "[plda_hip trace] %s\n", js.c_str());
    }
    for (auto &sp : h->trace_spans) { (void)hipEventDestroy(sp.e0); (void)hipEventDestroy(sp.e1); }
    (void)comm_destroy(h);
    delete h->hostpipe;
    h->hostpipe = nullptr;
    DevBuf *bufs[] = {&h->d_mean, &h->d_transform, &h->d_psi, &h->d_offset, &h->f_means, &h->f_counts,
                      &h->f_scatter, &h->f_sum, &h->f_W, &h->f_B, &h->fit_flag, &h->s_Apk, &h->s_Bpk, &h->s_rbias,
                      &h->s_rscale, &h->s_cbias, &h->s_rpair, &h->s_cpair, &h->tf_pad, &h->l_means, &h->l_priors, &h->l_xbar, &h->l_scalings,
                      &h->l_coef, &h->l_intercept, &h->l_evr, &h->timeline, &h->eigdc, &h->zn_rows, &h->zn_y, &h->zn_small, &h->hio_O[0], &h->hio_O[1],
                      &h->bt4_cnt, &h->bt4_fringe, &h->cs_work, &h->comm_mm, &h->comm_mc, &h->eer_list[0], &h->eer_list[1], &h->s_A16, &h->s_B16, &h->eer_slab, &h->eer_smp};
    for (DevBuf *b : bufs) b->release();
    for (auto &t : h->bt4_tabs) t.tab.release();
    if (h->cs_pin) (void)hipHostFree(h->cs_pin);
    for (auto &b : h->w) b.release();
    if (h->one_host) (void)hipHostFree(h->one_host);
    if (h->pin_model) (void)hipHostFree(h->pin_model);
    for (hipEvent_t e : h->fit_ev) if (e) (void)hipEventDestroy(e);
    if (h->jac_exec) (void)hipGraphExecDestroy(h->jac_exec);
    for (auto &ev : h->prof_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return PLDA_OK;
  });
}

const char *plda_last_error(const plda_handle *h) {
  // a per-thread copy taken under the handle's lock: another thread failing at the same moment
  // reassigns h->err, so the handle's own buffer must not be handed out
  static thread_local std::string tl_err;
  try {
    if (!h) return g_create_err.c_str();
    plda_handle *hm = const_cast<plda_handle *>(h);
    std::lock_guard<std::recursive_mutex> g(hm->mu);
    tl_err = h->err;
    return tl_err.c_str();
  } catch (...) { return "plda_last_error: out of host memory"; }
}

int plda_set_stream(plda_handle *h, void *hip_stream) {
  return guarded(h, "plda_set_stream", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    h->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return PLDA_OK;
  });
}

int plda_reset_stream(plda_handle *h) {
  return guarded(h, "plda_reset_stream", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    h->stream = h->own_stream;
    return PLDA_OK;
  });
}

int plda_synchronize(plda_handle *h) {
  return guarded(h, "plda_synchronize", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    return comm_check(h);      // a peer-provider wait that gave up is reported where the caller synchronises
  });
}

// ---------------------------------------------------------------- fit
int plda_fit_dev(plda_handle *h, const double *dX, int64_t N, int32_t D, const uint64_t *dlabels, int64_t K,
                 int32_t iters) {
  return guarded(h, "plda_fit_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return fit_device(h, dX, N, D, dlabels, K, iters);
  });
}

int plda_fit(plda_handle *h, const double *X, int64_t N, int32_t D, const uint64_t *labels, int32_t iters) {
  return guarded(h, "plda_fit", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!X || !labels || N <= 0 || D <= 0) return fail(h, PLDA_E_INVAL, "fit: bad argument");
    PLDA_TRY(set_device(h));
    uint64_t mx = 0;
    for (int64_t r = 0; r < N; ++r) mx = std::max(mx, labels[r]);
    if (mx >= (uint64_t)N) return fail(h, PLDA_E_LABELS, "fit: labels must be dense 0..K-1");
    const int64_t K = (int64_t)mx + 1;
    Tmp dX, dL;
    PLDA_TRY(upload(h, dX, X, (size_t)N * D * 8));
    PLDA_TRY(upload(h, dL, labels, (size_t)N * 8));
    return fit_device(h, dX.as<double>(), N, D, dL.as<uint64_t>(), K, iters);
  });
}

int plda_fit_stats_dev(plda_handle *h, const double *dX, int64_t N, int32_t D, const uint64_t *dlabels, int64_t K) {
  return guarded(h, "plda_fit_stats_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return fit_stats_device(h, dX, N, D, dlabels, K);
  });
}

int plda_fit_get_stats_dev(plda_handle *h, double *dmeans, int64_t *dcounts, double *dscatter) {
  return guarded(h, "plda_fit_get_stats_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (h->fit_K <= 0) return fail(h, PLDA_E_NOT_FITTED, "fit_get_stats_dev: no statistics pass has run on this handle");
    PLDA_TRY(set_device(h));
    const size_t K = (size_t)h->fit_K, D = (size_t)h->fit_D;
    if (dmeans) PLDA_HIP(h, hipMemcpyAsync(dmeans, h->f_means.p, K * D * 8, hipMemcpyDeviceToDevice, h->stream));
    if (dcounts) PLDA_HIP(h, hipMemcpyAsync(dcounts, h->f_counts.p, K * 8, hipMemcpyDeviceToDevice, h->stream));
    if (dscatter) PLDA_HIP(h, hipMemcpyAsync(dscatter, h->f_scatter.p, D * D * 8, hipMemcpyDeviceToDevice, h->stream));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    return PLDA_OK;
  });
}

int plda_fit_em_dev(plda_handle *h, const double *dmeans, const int64_t *dcounts, int64_t K, const double *dscatter,
                    int32_t D, int32_t iters) {
  return guarded(h, "plda_fit_em_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!dmeans || !dcounts || !dscatter || K <= 0 || D <= 0) return fail(h, PLDA_E_INVAL, "fit_em: bad argument");
    if (D > 2048) return fail(h, PLDA_E_INVAL, "fit: featdim %d > 2048 unsupported", D);
    PLDA_TRY(set_device(h));
    PLDA_HIP(h, h->f_means.reserve((size_t)K * D * 8));
    PLDA_HIP(h, h->f_counts.reserve((size_t)K * 8));
    PLDA_HIP(h, h->f_scatter.reserve((size_t)D * D * 8));
    if (dmeans != h->f_means.p)
      PLDA_HIP(h, hipMemcpyAsync(h->f_means.p, dmeans, (size_t)K * D * 8, hipMemcpyDeviceToDevice, h->stream));
    if ((const void *)dcounts != h->f_counts.p)
      PLDA_HIP(h, hipMemcpyAsync(h->f_counts.p, dcounts, (size_t)K * 8, hipMemcpyDeviceToDevice, h->stream));
    if (dscatter != h->f_scatter.p)
      PLDA_HIP(h, hipMemcpyAsync(h->f_scatter.p, dscatter, (size_t)D * D * 8, hipMemcpyDeviceToDevice, h->stream));
    h->fit_ms[0] = 0.0;
    return fit_em_device(h, K, D, iters);
  });
}

int plda_fit_timings(plda_handle *h, double out_ms[4]) {
  return guarded(h, "plda_fit_timings", [&]() -> int {
    if (!h || !out_ms) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    for (int i = 0; i < 4; ++i) out_ms[i] = h->fit_ms[i];
    return PLDA_OK;
  });
}

int plda_fit_plan(plda_handle *h, int32_t out[2]) {
  return guarded(h, "plda_fit_plan", [&]() -> int {
    if (!h || !out) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    out[0] = h->em_groups;
    out[1] = h->em_form;
    return PLDA_OK;
  });
}

int plda_fit_num_classes(plda_handle *h, int64_t *K) {
  return guarded(h, "plda_fit_num_classes", [&]() -> int {
    if (!h || !K) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    *K = h->fit_K;
    return PLDA_OK;
  });
}

int plda_fit_get_stats(plda_handle *h, double *means, int64_t *counts, double *scatter, double *sum,
                       double *W, double *B) {
  return guarded(h, "plda_fit_get_stats", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (h->fit_K <= 0) return fail(h, PLDA_E_NOT_FITTED, "fit_get_stats: no fit has run on this handle");
    PLDA_TRY(set_device(h));
    const size_t K = (size_t)h->fit_K, D = (size_t)h->fit_D;
    if (means) PLDA_HIP(h, hipMemcpyAsync(means, h->f_means.p, K * D * 8, hipMemcpyDeviceToHost, h->stream));
    if (counts) PLDA_HIP(h, hipMemcpyAsync(counts, h->f_counts.p, K * 8, hipMemcpyDeviceToHost, h->stream));
    if (scatter) PLDA_HIP(h, hipMemcpyAsync(scatter, h->f_scatter.p, D * D * 8, hipMemcpyDeviceToHost, h->stream));
    if (sum) PLDA_HIP(h, hipMemcpyAsync(sum, h->f_sum.p, D * 8, hipMemcpyDeviceToHost, h->stream));
    if (W) PLDA_HIP(h, hipMemcpyAsync(W, h->f_W.p, D * D * 8, hipMemcpyDeviceToHost, h->stream));
    if (B) PLDA_HIP(h, hipMemcpyAsync(B, h->f_B.p, D * D * 8, hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    return PLDA_OK;
  });
}

// ---------------------------------------------------------------- model
int plda_get_dims(plda_handle *h, int32_t *Dout, int32_t *Din) {
  return guarded(h, "plda_get_dims", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "model not fitted");
    if (Dout) *Dout = h->Dout;
    if (Din) *Din = h->Din;
    return PLDA_OK;
  });
}

int plda_get_model(plda_handle *h, double *mean, double *transform, double *psi, double *offset) {
  return guarded(h, "plda_get_model", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "model not fitted");
    if (mean) std::memcpy(mean, h->h_mean.data(), h->h_mean.size() * 8);
    if (transform) std::memcpy(transform, h->h_transform.data(), h->h_transform.size() * 8);
    if (psi) std::memcpy(psi, h->h_psi.data(), h->h_psi.size() * 8);
    if (offset) std::memcpy(offset, h->h_offset.data(), h->h_offset.size() * 8);
    return PLDA_OK;
  });
}

static int refresh_offset(plda_handle *h) {
  // offset = -transform . mean on the device (Plda::ComputeDerivedVars), mirrored back to the host
  PLDA_TRY(compute_offset_device(h));
  h->h_offset.resize(h->Dout);
  PLDA_HIP(h, hipMemcpyAsync(h->h_offset.data(), h->d_offset.p, (size_t)h->Dout * 8, hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  return PLDA_OK;
}

int plda_set_model(plda_handle *h, int32_t Dout, int32_t Din, const double *mean, const double *transform,
                   const double *psi) {
  return guarded(h, "plda_set_model", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (Dout <= 0 || Din <= 0 || Dout > Din || !mean || !transform || !psi)
      return fail(h, PLDA_E_INVAL, "set_model: bad argument");
    PLDA_TRY(set_device(h));
    h->Dout = Dout; h->Din = Din;
    h->h_mean.assign(mean, mean + Din);
    h->h_transform.assign(transform, transform + (size_t)Dout * Din);
    h->h_psi.assign(psi, psi + Dout);
    h->h_offset.assign(Dout, 0.0);
    PLDA_TRY(model_to_device(h));
    h->fitted = true;
    ++h->model_epoch;
    return refresh_offset(h);
  });
}

int plda_truncate(plda_handle *h, int32_t targetdim) {
  return guarded(h, "plda_truncate", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "model not fitted");
    if (targetdim <= 0 || targetdim > h->Dout) return fail(h, PLDA_E_INVAL, "truncate: targetdim %d not in [1,%d]", targetdim, h->Dout);
    PLDA_TRY(set_device(h));
    h->Dout = targetdim;
    h->h_transform.resize((size_t)targetdim * h->Din);
    h->h_psi.resize(targetdim);
    h->h_offset.resize(targetdim);
    ++h->model_epoch;
    return PLDA_OK;  // device arrays are row-major prefixes: nothing to move
  });
}

int plda_smooth(plda_handle *h, double factor) {
  return guarded(h, "plda_smooth", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "model not fitted");
    if (!(factor >= 0.0 && factor <= 1.0)) return fail(h, PLDA_E_INVAL, "smooth: factor must be in [0,1]");
    PLDA_TRY(set_device(h));
    smooth_kernel<<<h->Dout, 256, 0, h->stream>>>(h->d_transform.as<double>(), h->d_psi.as<double>(), h->Dout, h->Din, factor);
    PLDA_LAUNCH_CHECK(h);
    PLDA_HIP(h, hipMemcpyAsync(h->h_transform.data(), h->d_transform.p, (size_t)h->Dout * h->Din * 8, hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipMemcpyAsync(h->h_psi.data(), h->d_psi.p, (size_t)h->Dout * 8, hipMemcpyDeviceToHost, h->stream));
    ++h->model_epoch;
    return refresh_offset(h);
  });
}

// ---------------------------------------------------------------- transform
int plda_transform_rows_dev(plda_handle *h, const double *dXbar, int64_t R, int32_t Din, const int32_t *dn,
                            int32_t n_uniform, double *dout) {
  return guarded(h, "plda_transform_rows_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!dn && n_uniform <= 0) return fail(h, PLDA_E_INVAL, "transform_rows: need num_examples or n_uniform > 0");
    PLDA_TRY(set_device(h));
    return transform_rows_device(h, dXbar, R, Din, dn, n_uniform, dout);
  });
}

int plda_transform_rows(plda_handle *h, const double *Xbar, int64_t R, int32_t Din, const int32_t *num_examples,
                        int32_t n_uniform, double *out) {
  return guarded(h, "plda_transform_rows", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "transform: model not fitted");
    if (R <= 0) return PLDA_OK;
    if (!Xbar || !out) return fail(h, PLDA_E_INVAL, "transform_rows: bad argument");
    if (!num_examples && n_uniform <= 0) return fail(h, PLDA_E_INVAL, "transform_rows: need num_examples or n_uniform > 0");
    PLDA_TRY(set_device(h));
    Tmp dX, dN, dO;
    PLDA_TRY(upload(h, dX, Xbar, (size_t)R * Din * 8));
    if (num_examples) PLDA_TRY(upload(h, dN, num_examples, (size_t)R * 4));
    PLDA_HIP(h, dO.alloc((size_t)R * h->Dout * 8));
    PLDA_TRY(transform_rows_device(h, dX.as<double>(), R, Din, num_examples ? dN.as<int32_t>() : nullptr, n_uniform, dO.as<double>()));
    return download(h, out, dO.p, (size_t)R * h->Dout * 8);
  });
}

// grouping, per-label means and TransformIvector on device-resident rows; results stay on the device
// (out arrays sized by the caller's capacity *Ku; PLDA_E_CAPACITY reports the needed size in *Ku)
static int transform_groups_device(plda_handle *h, const double *dX, int64_t N, int32_t Din, const uint64_t *dlabels,
                                   uint64_t *dout_labels, int32_t *dcounts32, double *dout_vecs, int64_t *Ku,
                                   Tmp &dM) {
  uint32_t *perm = nullptr;
  int *offsets = nullptr;
  uint64_t *uniq = nullptr;
  int64_t G = 0;
  // label compaction on the device (the reference does it with std::map, pldamodule.cpp:118,147-156)
  PLDA_TRY(group_by_label_device(h, dlabels, N, &perm, &offsets, &uniq, &G));
  if (G > *Ku) { *Ku = G; return fail(h, PLDA_E_CAPACITY, "transform_groups: %lld groups > capacity", (long long)G); }
  PLDA_HIP(h, dM.alloc((size_t)G * Din * 8));
  PLDA_TRY(group_centroids_device(h, dX, N, Din, perm, offsets, G, dM.as<double>(), dcounts32));
  PLDA_TRY(transform_rows_device(h, dM.as<double>(), G, Din, dcounts32, 0, dout_vecs));
  PLDA_HIP(h, hipMemcpyAsync(dout_labels, uniq, (size_t)G * 8, hipMemcpyDeviceToDevice, h->stream));
  *Ku = G;
  return PLDA_OK;
}

int plda_transform_groups_dev(plda_handle *h, const double *dX, int64_t N, int32_t Din, const uint64_t *dlabels,
                              uint64_t *dout_labels, int32_t *dout_counts, double *dout_vecs, int64_t *Ku) {
  return guarded(h, "plda_transform_groups_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "transform: model not fitted");
    if (!Ku) return fail(h, PLDA_E_INVAL, "transform_groups: Ku is NULL");
    if (N <= 0) { *Ku = 0; return PLDA_OK; }
    if (!dX || !dlabels || !dout_labels || !dout_counts || !dout_vecs) return fail(h, PLDA_E_INVAL, "transform_groups: bad argument");
    if (Din != h->Din) return fail(h, PLDA_E_INVAL, "transform: feature dim %d != model dim %d", Din, h->Din);
    PLDA_TRY(set_device(h));
    Tmp dM;
    PLDA_TRY(transform_groups_device(h, dX, N, Din, dlabels, dout_labels, dout_counts, dout_vecs, Ku, dM));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));   // dM is released on return
    return PLDA_OK;
  });
}

int plda_transform_groups(plda_handle *h, const double *X, int64_t N, int32_t Din, const uint64_t *labels,
                          uint64_t *out_labels, int64_t *out_counts, double *out_vecs, int64_t *Ku) {
  return guarded(h, "plda_transform_groups", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "transform: model not fitted");
    if (!Ku) return fail(h, PLDA_E_INVAL, "transform_groups: Ku is NULL");
    if (N <= 0) { *Ku = 0; return PLDA_OK; }
    if (!X || !labels || !out_labels || !out_counts || !out_vecs) return fail(h, PLDA_E_INVAL, "transform_groups: bad argument");
    if (Din != h->Din) return fail(h, PLDA_E_INVAL, "transform: feature dim %d != model dim %d", Din, h->Din);
    PLDA_TRY(set_device(h));
    const int64_t cap = *Ku;
    if (cap <= 0) return fail(h, PLDA_E_CAPACITY, "transform_groups: capacity must be > 0");
    Tmp dX, dL, dM, dC, dO, dU;
    PLDA_TRY(upload(h, dX, X, (size_t)N * Din * 8));
    PLDA_TRY(upload(h, dL, labels, (size_t)N * 8));
    const int64_t gmax = std::min(cap, N);
    PLDA_HIP(h, dC.alloc((size_t)gmax * 4));
    PLDA_HIP(h, dO.alloc((size_t)gmax * h->Dout * 8));
    PLDA_HIP(h, dU.alloc((size_t)gmax * 8));
    int64_t G = gmax;
    const int rc = transform_groups_device(h, dX.as<double>(), N, Din, dL.as<uint64_t>(), dU.as<uint64_t>(),
                                           dC.as<int32_t>(), dO.as<double>(), &G, dM);
    if (rc != PLDA_OK) { if (rc == PLDA_E_CAPACITY) *Ku = G; return rc; }
    std::vector<int32_t> c32((size_t)G);
    PLDA_HIP(h, hipMemcpyAsync(c32.data(), dC.p, (size_t)G * 4, hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipMemcpyAsync(out_labels, dU.p, (size_t)G * 8, hipMemcpyDeviceToHost, h->stream));
    PLDA_TRY(download(h, out_vecs, dO.p, (size_t)G * h->Dout * 8));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    for (int64_t g = 0; g < G; ++g) out_counts[g] = c32[g];
    *Ku = G;
    return PLDA_OK;
  });
}

// ---------------------------------------------------------------- score
int plda_score_matrix_dev(plda_handle *h, const double *dU, const int32_t *dn_enrol, int32_t n_uniform, int64_t M,
                          const double *dV, int64_t Nt, const double *dzmean, const double *dzstd, float *dout,
                          int64_t ld_out) {
  return guarded(h, "plda_score_matrix_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return score_matrix_device(h, dU, dn_enrol, n_uniform, M, dV, Nt, dzmean, dzstd, dout, ld_out);
  });
}

int plda_score_prepare_dev(plda_handle *h, const double *dV, int64_t Nt, int32_t mixed_counts, int32_t n_uniform) {
  return guarded(h, "plda_score_prepare_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return score_prepare_device(h, dV, Nt, mixed_counts != 0 ? 1 : 0, n_uniform, nullptr);
  });
}

int plda_score_prepare_counts_dev(plda_handle *h, const double *dV, int64_t Nt, const int32_t *counts, int32_t num_counts) {
  return guarded(h, "plda_score_prepare_counts_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    if (!counts || num_counts <= 0) return fail(h, PLDA_E_INVAL, "score_prepare_counts: empty count list");
    CountSet cs;
    score_count_set_host(counts, num_counts, &cs);     // (sorted, duplicates dropped; G = 0: outside what the bucketed form takes)
    if (cs.G == 0) return score_prepare_device(h, dV, Nt, 1, 0, nullptr);
    return score_prepare_device(h, dV, Nt, 2, 0, &cs);
  });
}

int plda_score_unprepare(plda_handle *h) {
  return guarded(h, "plda_score_unprepare", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    h->prep_valid = false;
    return PLDA_OK;
  });
}

// the round-1/2 form of the host-pointer trials matrix: one slab buffer, the runtime's pageable copy, a
// synchronisation per slab.  Kept as the A/B arm (PLDA_HOST_VARIANT=1) and for test sets so wide that a
// single row of scores does not fit a slot of the pinned ring.
static int score_matrix_host_serial(plda_handle *h, const double *U, const int32_t *n_enrol, int32_t n_uniform, int64_t M,
                                    const double *V, int64_t Nt, const double *zmean, const double *zstd, float *out,
                                    int64_t ld_out) {
  const int D = h->Dout;
  Tmp dV, dU, dN, dZm, dZs, dO;
  CountSet cs;                                           // the distinct counts of ALL rows: every slab sees the same set
  if (n_enrol) score_count_set_host(n_enrol, M, &cs);
  PLDA_TRY(upload(h, dV, V, (size_t)Nt * D * 8));
  // row slabs so that the device score block stays <= 1 GiB
  int64_t slab = std::max<int64_t>(128, ((1ll << 30) / 4 / Nt) / 128 * 128);
  slab = std::min(slab, round_up(M, 128));
  PLDA_HIP(h, dU.alloc((size_t)slab * D * 8));
  PLDA_HIP(h, dO.alloc((size_t)slab * Nt * 4));
  if (n_enrol) PLDA_HIP(h, dN.alloc((size_t)slab * 4));
  if (zmean && zstd) { PLDA_HIP(h, dZm.alloc((size_t)slab * 8)); PLDA_HIP(h, dZs.alloc((size_t)slab * 8)); }
  for (int64_t r0 = 0; r0 < M; r0 += slab) {
    const int64_t m = std::min(slab, M - r0);
    PLDA_HIP(h, hipMemcpyAsync(dU.p, U + r0 * D, (size_t)m * D * 8, hipMemcpyHostToDevice, h->stream));
    if (n_enrol) PLDA_HIP(h, hipMemcpyAsync(dN.p, n_enrol + r0, (size_t)m * 4, hipMemcpyHostToDevice, h->stream));
    if (zmean && zstd) {
      PLDA_HIP(h, hipMemcpyAsync(dZm.p, zmean + r0, (size_t)m * 8, hipMemcpyHostToDevice, h->stream));
      PLDA_HIP(h, hipMemcpyAsync(dZs.p, zstd + r0, (size_t)m * 8, hipMemcpyHostToDevice, h->stream));
    }
    PLDA_TRY(score_matrix_device(h, dU.as<double>(), n_enrol ? dN.as<int32_t>() : nullptr, n_uniform, m,
                                 dV.as<double>(), Nt, (zmean && zstd) ? dZm.as<double>() : nullptr,
                                 (zmean && zstd) ? dZs.as<double>() : nullptr, dO.as<float>(), Nt,
                                 /*reuse_packed_B=*/r0 > 0, n_enrol ? &cs : nullptr));   // the test side is packed once, not per slab
    if (ld_out == Nt)
      PLDA_HIP(h, hipMemcpyAsync(out + r0 * ld_out, dO.p, (size_t)m * Nt * 4, hipMemcpyDeviceToHost, h->stream));
    else
      PLDA_HIP(h, hipMemcpy2DAsync(out + r0 * ld_out, (size_t)ld_out * 4, dO.p, (size_t)Nt * 4, (size_t)Nt * 4,
                                   (size_t)m, hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
  }
  return PLDA_OK;
}

// Host-pointer trials matrix.  What bounds it is PCIe, not the GEMM (1.4 TB/s of scores against <= 64 GB/s of
// link): slabs of <= 64 MiB of scores alternate between two device buffers; while the GEMM of slab i + 1 runs,
// slab i crosses the link into a pinned slot on the copy stream and slab i - 1 is copied from its slot into the
// caller's array by the host threads, which also take the page faults of a freshly allocated output in
// parallel (hostio.hip).  Enrol vectors, counts and z-norm maps are uploaded once, not per slab.
int plda_score_matrix(plda_handle *h, const double *U, const int32_t *n_enrol, int32_t n_uniform, int64_t M,
                      const double *V, int64_t Nt, const double *zmean, const double *zstd, float *out,
                      int64_t ld_out) {
  return guarded(h, "plda_score_matrix", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "score_matrix: model not fitted");
    if (M <= 0 || Nt <= 0) return PLDA_OK;
    if (!U || !V || !out || ld_out < Nt) return fail(h, PLDA_E_INVAL, "score_matrix: bad argument");
    if (!n_enrol && n_uniform <= 0) return fail(h, PLDA_E_INVAL, "score_matrix: n_uniform must be > 0 when n_enrol is NULL");
    PLDA_TRY(set_device(h));
    const int D = h->Dout;
    const bool zn = zmean && zstd;
    h->last_M = M;
    const int64_t fit_rows = (int64_t)(HostPipe::SLOT_BYTES / 4) / Nt;      // rows of scores per pinned slot
    if (h->host_variant == 1 || fit_rows < 1 || (size_t)M * D * 8 > ((size_t)2 << 30)) {
      const int rc = score_matrix_host_serial(h, U, n_enrol, n_uniform, M, V, Nt, zmean, zstd, out, ld_out);
      h->last_M = M;
      return rc;
    }
    HostPipe *hp = nullptr;
    PLDA_TRY(host_pipe(h, &hp));
    const int64_t slab = fit_rows >= 256 ? fit_rows / 128 * 128 : fit_rows;
    Tmp dV, dU, dN, dZm, dZs;
    CountSet cs;                                         // the distinct counts of ALL rows, from the host array: no device pass
    if (n_enrol) score_count_set_host(n_enrol, M, &cs);
    PLDA_TRY(upload(h, dV, V, (size_t)Nt * D * 8));
    PLDA_TRY(upload(h, dU, U, (size_t)M * D * 8));
    if (n_enrol) PLDA_TRY(upload(h, dN, n_enrol, (size_t)M * 4));
    if (zn) { PLDA_TRY(upload(h, dZm, zmean, (size_t)M * 8)); PLDA_TRY(upload(h, dZs, zstd, (size_t)M * 8)); }
    for (auto &b : h->hio_O) PLDA_HIP(h, b.reserve((size_t)std::min(slab, M) * Nt * 4));
    advise_huge(out, (size_t)((M - 1) * ld_out + Nt) * 4);
    size_t i = 0;
    int rc = PLDA_OK;
    for (int64_t r0 = 0; r0 < M && rc == PLDA_OK; r0 += slab, ++i) {
      const int64_t m = std::min(slab, M - r0);
      float *dO = h->hio_O[i & 1].as<float>();
      hipError_t e = hp->begin_slab(h->stream, i);
      if (e != hipSuccess) { rc = hip_fail(h, e, "begin_slab", __FILE__, __LINE__); break; }
      rc = score_matrix_device(h, dU.as<double>() + r0 * D, n_enrol ? dN.as<int32_t>() + r0 : nullptr, n_uniform, m,
                               dV.as<double>(), Nt, zn ? dZm.as<double>() + r0 : nullptr, zn ? dZs.as<double>() + r0 : nullptr,
                               dO, Nt, /*reuse_packed_B=*/r0 > 0, n_enrol ? &cs : nullptr);
      if (rc != PLDA_OK) break;
      e = hp->ship_slab(h->stream, i, dO, (size_t)m * Nt * 4, reinterpret_cast<char *>(out + r0 * ld_out), (size_t)ld_out * 4,
                        (size_t)Nt * 4, (size_t)m);
      if (e != hipSuccess) rc = hip_fail(h, e, "ship_slab", __FILE__, __LINE__);
    }
    const hipError_t ef = hp->finish();                       // also on failure: nothing of this call stays in flight
    if (rc == PLDA_OK && ef != hipSuccess) rc = hip_fail(h, ef, "HostPipe::finish", __FILE__, __LINE__);
    PLDA_HIP(h, hipStreamSynchronize(h->stream));              // the temporaries go out of scope
    h->last_M = M;
    return rc;
  });
}

// spans aggregated by name, in order of first appearance: [{"name": .., "calls": n, "ms": total, "work": w, "unit": "flop"|"bytes"|""}, ..]
static int trace_summary(plda_handle *h, std::string &out, bool reset) {
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  struct Agg { const char *name; long calls; double ms, work; int unit; };
  std::vector<Agg> agg;
  for (size_t i = 0; i < h->trace_used; ++i) {
    auto &sp = h->trace_spans[i];
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, sp.e0, sp.e1) != hipSuccess) { (void)hipGetLastError(); continue; }
    Agg *a = nullptr;
    for (auto &x : agg)
      if (std::strcmp(x.name, sp.name) == 0) { a = &x; break; }
    if (!a) { agg.push_back(Agg{sp.name, 0, 0.0, 0.0, sp.unit}); a = &agg.back(); }
    a->calls++; a->ms += ms; a->work += sp.work;
  }
  out = "[";
  char buf[256];
  for (size_t i = 0; i < agg.size(); ++i) {
    std::snprintf(buf, sizeof buf, "%s{\"name\": \"%s\", \"calls\": %ld, \"ms\": %.6f, \"work\": %.6g, \"unit\": \"%s\"}", i ? ", " : "",
                  agg[i].name, agg[i].calls, agg[i].ms, agg[i].work, agg[i].unit == 1 ? "flop" : agg[i].unit == 2 ? "bytes" : "");
    out += buf;
  }
  out += "]";
  if (reset) h->trace_used = 0;
  return PLDA_OK;
}

int plda_trace_enable(plda_handle *h, int32_t on) {
  return guarded(h, "plda_trace_enable", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    h->trace_on = on != 0;
    return PLDA_OK;
  });
}

int plda_trace_read(plda_handle *h, char *json, int64_t cap, int32_t reset) {
  return guarded(h, "plda_trace_read", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!json || cap <= 0) return fail(h, PLDA_E_INVAL, "trace_read: bad argument");
    PLDA_TRY(set_device(h));
    std::string js;
    PLDA_TRY(trace_summary(h, js, false));
    if ((int64_t)js.size() + 1 > cap) return fail(h, PLDA_E_CAPACITY, "trace_read: need %zu bytes", js.size() + 1);
    std::memcpy(json, js.c_str(), js.size() + 1);
    if (reset) h->trace_used = 0;
    return PLDA_OK;
  });
}

int plda_profile_enable(plda_handle *h, int32_t on) {
  return guarded(h, "plda_profile_enable", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    h->prof_on = on != 0;
    return PLDA_OK;
  });
}

int plda_profile_read(plda_handle *h, double *gemm_ms, int64_t *launches, double *gemm_flop, int32_t reset) {
  return guarded(h, "plda_profile_read", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    double total = 0.0;
    for (size_t i = 0; i < h->prof_used; ++i) {
      float ms = 0.f;
      PLDA_HIP(h, hipEventElapsedTime(&ms, h->prof_events[i].first, h->prof_events[i].second));
      total += ms;
    }
    if (gemm_ms) *gemm_ms = total;
    if (launches) *launches = (int64_t)h->prof_used;
    if (gemm_flop) *gemm_flop = h->prof_flop;
    if (reset) { h->prof_used = 0; h->prof_flop = 0.0; }
    return PLDA_OK;
  });
}

int plda_profile_timeline(plda_handle *h, uint64_t *out, int64_t cap_words) {
  return guarded(h, "plda_profile_timeline", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!out || cap_words < (int64_t)TIMELINE_WORDS) return fail(h, PLDA_E_CAPACITY, "profile_timeline: need %zu words", TIMELINE_WORDS);
    if (!h->timeline_valid) return fail(h, PLDA_E_INVAL, "profile_timeline: no PLDA_GEMM_VARIANT=31 launch has run on this handle");
    PLDA_TRY(set_device(h));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    PLDA_HIP(h, hipMemcpy(out, h->timeline.p, TIMELINE_WORDS * 8, hipMemcpyDeviceToHost));
    return PLDA_OK;
  });
}

int plda_gemm_f64(plda_handle *h, int64_t M, int64_t N, int64_t K, double alpha, const double *A, int32_t transA,
                  const double *B, int32_t transB, const double *kw, double beta, double *C, int32_t batch) {
  return guarded(h, "plda_gemm_f64", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || (kw && batch != 1))
      return fail(h, PLDA_E_INVAL, "gemm_f64: bad argument");
    PLDA_TRY(set_device(h));
    const size_t nA = (size_t)M * K, nB = (size_t)K * N, nC = (size_t)M * N;
    Tmp dA, dB, dC, dW;
    // the same host array on both sides (X^T diag(w) X: the scatter's product) stays ONE device array, so that the
    // dispatch sees what fit's statistics pass hands it and takes the symmetric kernels
    const bool same = A == B && nA == nB && batch == 1;
    PLDA_HIP(h, dA.alloc(nA * batch * 8));
    if (!same) PLDA_HIP(h, dB.alloc(nB * batch * 8));
    PLDA_HIP(h, dC.alloc(nC * batch * 8));
    PLDA_HIP(h, hipMemcpyAsync(dA.p, A, nA * batch * 8, hipMemcpyHostToDevice, h->stream));
    if (!same) PLDA_HIP(h, hipMemcpyAsync(dB.p, B, nB * batch * 8, hipMemcpyHostToDevice, h->stream));
    PLDA_HIP(h, hipMemcpyAsync(dC.p, C, nC * batch * 8, hipMemcpyHostToDevice, h->stream));
    if (kw) {
      PLDA_HIP(h, dW.alloc((size_t)K * 8));
      PLDA_HIP(h, hipMemcpyAsync(dW.p, kw, (size_t)K * 8, hipMemcpyHostToDevice, h->stream));
    }
    // element (m, k) of op(A): m * sam + k * sak; element (k, n) of op(B): k * sbk + n * sbn
    const int64_t sam = transA ? 1 : K, sak = transA ? M : 1, sbk = transB ? 1 : N, sbn = transB ? K : 1;
    PLDA_TRY(gemm_f64_batched(h, M, N, K, alpha, dA.as<double>(), sam, sak, (int64_t)nA, same ? dA.as<double>() : dB.as<double>(), sbk, sbn, (int64_t)nB,
                              kw ? dW.as<double>() : nullptr, beta, dC.as<double>(), N, (int64_t)nC, batch));
    PLDA_HIP(h, hipMemcpyAsync(C, dC.p, nC * batch * 8, hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    return PLDA_OK;
  });
}

int plda_spd_inverse(plda_handle *h, const double *A, int32_t D, double *inverse) {
  return guarded(h, "plda_spd_inverse", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!A || !inverse || D <= 0 || D > 2048) return fail(h, PLDA_E_INVAL, "spd_inverse: bad argument");
    PLDA_TRY(set_device(h));
    const size_t DD = (size_t)D * D;
    Tmp dA, dO, dS, dF;
    PLDA_HIP(h, dA.alloc(DD * 8));
    PLDA_HIP(h, dO.alloc(DD * 8));
    PLDA_HIP(h, dS.alloc(3 * DD * 8));
    PLDA_HIP(h, dF.alloc(sizeof(int)));
    PLDA_HIP(h, hipMemcpyAsync(dA.p, A, DD * 8, hipMemcpyHostToDevice, h->stream));
    PLDA_HIP(h, hipMemsetAsync(dF.p, 0, sizeof(int), h->stream));
    PLDA_TRY(spd_inverse_blocked(h, dA.as<double>(), D, D, (int64_t)DD, dO.as<double>(), D, (int64_t)DD, dS.as<double>(),
                                 (int64_t)(3 * DD), dF.as<int>(), 1));
    int bad = 0;
    PLDA_HIP(h, hipMemcpyAsync(inverse, dO.p, DD * 8, hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipMemcpyAsync(&bad, dF.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    if (bad) return fail(h, PLDA_E_NUMERIC, "spd_inverse: the matrix is not positive definite");
    return PLDA_OK;
  });
}

int plda_sym_eig(plda_handle *h, const double *G, int32_t D, int32_t method, double *eigenvalues, double *eigenvectors,
                 int32_t *method_used) {
  return guarded(h, "plda_sym_eig", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!G || !eigenvalues || !eigenvectors || D <= 0 || D > 2048) return fail(h, PLDA_E_INVAL, "sym_eig: bad argument");
    if (method < 0 || method > 2) return fail(h, PLDA_E_INVAL, "sym_eig: method must be 0 (default), 1 (Jacobi) or 2 (direct)");
    if (method == 1 && D > 1024) return fail(h, PLDA_E_INVAL, "sym_eig: the block Jacobi solver (method 1) stops at D = 1024; D = %d needs method 0 or 2", D);
    PLDA_TRY(set_device(h));
    const size_t DD = (size_t)D * D;
    Tmp dG, dS, dV;
    PLDA_HIP(h, dG.alloc(DD * 8));
    PLDA_HIP(h, dS.alloc((size_t)D * 8));
    PLDA_HIP(h, dV.alloc(DD * 8));
    PLDA_HIP(h, hipMemcpyAsync(dG.p, G, DD * 8, hipMemcpyHostToDevice, h->stream));
    const bool keep = h->eig_keep_sign;
    const int variant = h->eig_variant;
    h->eig_keep_sign = true;                       // signed eigenvalues: this entry point floors nothing
    int rc = PLDA_OK, status = 1;
    if (method == 2 || (method == 0 && variant != 1)) {
      rc = sym_eig_dc_f64(h, dG.as<double>(), D, dS.as<double>(), dV.as<double>(), &status);
      if (rc == PLDA_OK && status != 0 && method == 2) {
        h->eig_keep_sign = keep;
        return fail(h, PLDA_E_NUMERIC, "sym_eig: the direct method does not handle this input (status %d)", status);
      }
    }
    if (rc == PLDA_OK && status != 0) {
      rc = sym_eig_f64(h, dG.as<double>(), D, dS.as<double>(), dV.as<double>(), nullptr, nullptr);
      if (method_used) *method_used = 1;
    } else if (method_used) {
      *method_used = 2;
    }
    h->eig_keep_sign = keep;
    PLDA_TRY(rc);
    PLDA_HIP(h, hipMemcpyAsync(eigenvalues, dS.p, (size_t)D * 8, hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipMemcpyAsync(eigenvectors, dV.p, DD * 8, hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    return PLDA_OK;
  });
}

int plda_score_last_shape(plda_handle *h, int64_t *M, int64_t *Nt, int32_t *gemm_k) {
  return guarded(h, "plda_score_last_shape", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (M) *M = h->last_M;
    if (Nt) *Nt = h->last_Nt;
    if (gemm_k) *gemm_k = h->last_k;
    return PLDA_OK;
  });
}

int plda_score_last_kernel(plda_handle *h, char *name, int64_t cap) {
  return guarded(h, "plda_score_last_kernel", [&]() -> int {
    if (!h || !name || cap <= 0) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    const char *k = h->last_kernel ? h->last_kernel : "";
    if ((int64_t)std::strlen(k) + 1 > cap) return fail(h, PLDA_E_CAPACITY, "score_last_kernel: need %zu bytes", std::strlen(k) + 1);
    std::memcpy(name, k, std::strlen(k) + 1);
    return PLDA_OK;
  });
}

int plda_score_pairs(plda_handle *h, const double *U, const int32_t *n_enrol, int64_t M, const double *V,
                     int64_t Nt, const int64_t *e_idx, const int64_t *t_idx, int64_t P, const double *zmean,
                     const double *zstd, double *out) {
  return guarded(h, "plda_score_pairs", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "score: model not fitted");
    if (P <= 0) return PLDA_OK;
    if (!U || !n_enrol || !V || !e_idx || !t_idx || !out || M <= 0 || Nt <= 0) return fail(h, PLDA_E_INVAL, "score_pairs: bad argument");
    const bool long_list = P >= 16384;      // (long lists: the indices are checked on the device, below -- 10^7 pairs are 10 ms of this loop)
    if (!long_list)
      for (int64_t p = 0; p < P; ++p)
        if (e_idx[p] < 0 || e_idx[p] >= M || t_idx[p] < 0 || t_idx[p] >= Nt)
          return fail(h, PLDA_E_INVAL, "score_pairs: trial %lld indexes outside the enrol/test sets", (long long)p);
    for (int64_t i = 0; i < M; ++i)
      if (n_enrol[i] <= 0) return fail(h, PLDA_E_INVAL, "score_pairs: num_examples must be > 0");
    PLDA_TRY(set_device(h));
    const int D = h->Dout;
    if (P == 1 && M == 1 && Nt == 1) {
      // One trial -- the reference's MPlda_score call (pldamodule.cpp:258-277).  No allocation and no copy
      // engine: the operands go into a host buffer that is mapped into the GPU's address space, the kernel
      // reads them and writes the score back through that mapping, and one stream synchronisation ends the call.
      const size_t need = ((size_t)2 * D + 8) * 8;
      if (h->one_cap < need) {
        if (h->one_host) (void)hipHostFree(h->one_host);
        h->one_host = nullptr; h->one_cap = 0;
        PLDA_HIP(h, hipHostMalloc(&h->one_host, need, hipHostMallocMapped));
        PLDA_HIP(h, hipHostGetDevicePointer(&h->one_dev, h->one_host, 0));
        h->one_cap = need;
      }
      double *hb = static_cast<double *>(h->one_host);
      double *db = static_cast<double *>(h->one_dev);
      // layout: u[D] v[D] zmean zstd out | e_idx t_idx (int64) | n (int32)
      std::memcpy(hb, U, (size_t)D * 8);
      std::memcpy(hb + D, V, (size_t)D * 8);
      const bool zn1 = zmean && zstd;
      hb[2 * D] = zn1 ? zmean[0] : 0.0;
      hb[2 * D + 1] = zn1 ? zstd[0] : 0.0;
      int64_t *hi = reinterpret_cast<int64_t *>(hb + 2 * D + 3);
      hi[0] = 0; hi[1] = 0;
      *reinterpret_cast<int32_t *>(hb + 2 * D + 5) = n_enrol[0];
      PLDA_TRY(score_pairs_device(h, db, reinterpret_cast<const int32_t *>(db + 2 * D + 5), db + D,
                                  reinterpret_cast<const int64_t *>(db + 2 * D + 3),
                                  reinterpret_cast<const int64_t *>(db + 2 * D + 4), 1, zn1 ? db + 2 * D : nullptr,
                                  zn1 ? db + 2 * D + 1 : nullptr, db + 2 * D + 2));
      PLDA_HIP(h, hipStreamSynchronize(h->stream));
      out[0] = hb[2 * D + 2];
      return PLDA_OK;
    }
    Tmp dU, dN, dV, dE, dT, dZm, dZs, dO;
    PLDA_TRY(upload(h, dU, U, (size_t)M * D * 8));
    PLDA_TRY(upload(h, dN, n_enrol, (size_t)M * 4));
    PLDA_TRY(upload(h, dV, V, (size_t)Nt * D * 8));
    PLDA_TRY(upload(h, dE, e_idx, (size_t)P * 8));
    PLDA_TRY(upload(h, dT, t_idx, (size_t)P * 8));
    const bool zn = zmean && zstd;
    if (zn) { PLDA_TRY(upload(h, dZm, zmean, (size_t)M * 8)); PLDA_TRY(upload(h, dZs, zstd, (size_t)M * 8)); }
    PLDA_HIP(h, dO.alloc((size_t)P * 8));
    if (long_list) {
      long long bad = -1;
      PLDA_TRY(plda::pairs_validate_device(h, dE.as<int64_t>(), dT.as<int64_t>(), P, M, Nt, &bad));
      if (bad >= 0) return fail(h, PLDA_E_INVAL, "score_pairs: trial %lld indexes outside the enrol/test sets", bad);
    }
    plda::CountSet cs;                                   // the distinct enrol counts: long lists run on per-count tables (score.hip)
    if (P >= 16384) plda::score_count_set_host(n_enrol, M, &cs);
    PLDA_TRY(score_pairs_device(h, dU.as<double>(), dN.as<int32_t>(), dV.as<double>(), dE.as<int64_t>(),
                                dT.as<int64_t>(), P, zn ? dZm.as<double>() : nullptr,
                                zn ? dZs.as<double>() : nullptr, dO.as<double>(), M, &cs));
    PLDA_TRY(download(h, out, dO.p, (size_t)P * 8));
    return PLDA_OK;
  });
}

// One trial on the HOST -- the reference's MPlda_score call (pldamodule.cpp:258-277: Plda::LogLikelihoodRatio at :266,
// z-norm at :269-273) for callers that score one pair per Python call (scoring/scorePLDA.py:302-318,
// tests/pldatest.py:29-33).  A single trial is 2 D numbers and ~5 D flop: any trip to the GPU (>= 30 us of launch and
// synchronisation latency) costs ten times what the arithmetic does, so this entry point evaluates
//   -1/2 [ sum_d log var_d - log(1 + psi_d) + (v_d - c_d u_d)^2 / var_d - v_d^2 / (1 + psi_d) ],
//   c_d = n psi_d / (n psi_d + 1),  var_d = 1 + psi_d / (n psi_d + 1)
// on the handle's host mirror of psi, with the per-count terms (c, 1/var, 1/(1 + psi), the log sum) cached for the last
// n.  fp64; own code of the library (the oracle is not involved); callers with more than a handful of trials belong
// on plda_score_pairs / plda_score_matrix.
int plda_score_one(plda_handle *h, const double *u, int32_t n, const double *v, int32_t has_z, double zmean, double zstd,
                   double *out) {
  return guarded(h, "plda_score_one", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "score: model not fitted");
    if (!u || !v || !out) return fail(h, PLDA_E_INVAL, "score_one: bad argument");
    if (n <= 0) return fail(h, PLDA_E_INVAL, "score_one: num_examples must be > 0");
    const int D = h->Dout;
    auto &o = h->one;
    if (o.epoch != h->model_epoch || o.n != n || o.D != D) {
      o.c.resize(D); o.ivar.resize(D); o.ipsi1.resize(D);
      double lt = 0.0;
      for (int d = 0; d < D; ++d) {
        const double ps = h->h_psi[d], den = (double)n * ps + 1.0;
        const double var = 1.0 + ps / den;
        o.c[d] = (double)n * ps / den;
        o.ivar[d] = 1.0 / var;
        o.ipsi1[d] = 1.0 / (1.0 + ps);
        lt += std::log(var) - std::log(1.0 + ps);
      }
      o.logterm = lt; o.epoch = h->model_epoch; o.n = n; o.D = D;
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;     // four partial sums: the loop vectorises
    const double *c = o.c.data(), *iv = o.ivar.data(), *ip = o.ipsi1.data();
    int d = 0;
    for (; d + 4 <= D; d += 4) {
      const double e0 = v[d] - c[d] * u[d], e1 = v[d + 1] - c[d + 1] * u[d + 1];
      const double e2 = v[d + 2] - c[d + 2] * u[d + 2], e3 = v[d + 3] - c[d + 3] * u[d + 3];
      a0 += e0 * e0 * iv[d] - v[d] * v[d] * ip[d];
      a1 += e1 * e1 * iv[d + 1] - v[d + 1] * v[d + 1] * ip[d + 1];
      a2 += e2 * e2 * iv[d + 2] - v[d + 2] * v[d + 2] * ip[d + 2];
      a3 += e3 * e3 * iv[d + 3] - v[d + 3] * v[d + 3] * ip[d + 3];
    }
    for (; d < D; ++d) { const double e = v[d] - c[d] * u[d]; a0 += e * e * iv[d] - v[d] * v[d] * ip[d]; }
    double sc = -0.5 * (o.logterm + ((a0 + a1) + (a2 + a3)));
    if (has_z && zstd != 0.0) sc = (sc - zmean) / zstd;   // zstd == 0: left un-normalised, as the trial-list kernel does
    *out = sc;
    return PLDA_OK;
  });
}

// ---------------------------------------------------------------- z-norm
int plda_znorm_stats_dev(plda_handle *h, const double *dbkg, int64_t Nb, int32_t num_examples, int32_t Din,
                         const double *dmodels, int64_t M, double *dout_mean, double *dout_std) {
  return guarded(h, "plda_znorm_stats_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return znorm_stats_device(h, dbkg, Nb, num_examples, Din, dmodels, M, dout_mean, dout_std);
  });
}

int plda_znorm_stats(plda_handle *h, const double *bkg, int64_t Nb, int32_t num_examples, int32_t Din,
                     const double *models, int64_t M, double *out_mean, double *out_std) {
  return guarded(h, "plda_znorm_stats", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "norm: model not fitted");
    if (!bkg || !models || !out_mean || !out_std || Nb <= 0 || M <= 0) return fail(h, PLDA_E_INVAL, "norm: bad argument");
    if (Din != h->Din) return fail(h, PLDA_E_INVAL, "norm: feature dim %d != model dim %d", Din, h->Din);
    PLDA_TRY(set_device(h));
    Tmp dB, dM, dMean, dStd;
    PLDA_TRY(upload(h, dB, bkg, (size_t)Nb * Din * 8));
    PLDA_TRY(upload(h, dM, models, (size_t)M * h->Dout * 8));
    PLDA_HIP(h, dMean.alloc((size_t)M * 8));
    PLDA_HIP(h, dStd.alloc((size_t)M * 8));
    PLDA_TRY(znorm_stats_device(h, dB.as<double>(), Nb, num_examples, Din, dM.as<double>(), M, dMean.as<double>(),
                                dStd.as<double>()));
    PLDA_HIP(h, hipMemcpyAsync(out_mean, dMean.p, (size_t)M * 8, hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipMemcpyAsync(out_std, dStd.p, (size_t)M * 8, hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    return PLDA_OK;
  });
}

// ---------------------------------------------------------------- d-vector front-end
int plda_dvector_pool_dev(plda_handle *h, const void *dframes, int32_t dtype, int64_t T, int32_t D,
                          const int64_t *doffsets, int64_t U, int32_t method, int32_t l2norm, double *dout) {
  return guarded(h, "plda_dvector_pool_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return dvector_pool_device(h, dframes, dtype, T, D, doffsets, U, method, l2norm, dout);
  });
}

int plda_dvector_pool(plda_handle *h, const void *frames, int32_t dtype, int64_t T, int32_t D, const int64_t *offsets,
                      int64_t U, int32_t method, int32_t l2norm, double *out) {
  return guarded(h, "plda_dvector_pool", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (U <= 0) return PLDA_OK;
    if (!frames || !offsets || !out || T < 0 || D <= 0 || (dtype != 0 && dtype != 1))
      return fail(h, PLDA_E_INVAL, "dvector_pool: bad argument");
    for (int64_t u = 0; u < U; ++u)
      if (offsets[u] < 0 || offsets[u + 1] < offsets[u] || offsets[u + 1] > T)
        return fail(h, PLDA_E_INVAL, "dvector_pool: offsets must be non-decreasing within [0, T]");
    PLDA_TRY(set_device(h));
    Tmp dF, dO, dOut;
    PLDA_TRY(upload(h, dF, frames, (size_t)T * D * (dtype == 0 ? 4 : 8)));
    PLDA_TRY(upload(h, dO, offsets, (size_t)(U + 1) * 8));
    PLDA_HIP(h, dOut.alloc((size_t)U * D * 8));
    PLDA_TRY(dvector_pool_device(h, dF.p, dtype, T, D, dO.as<int64_t>(), U, method, l2norm, dOut.as<double>()));
    PLDA_HIP(h, hipMemcpyAsync(out, dOut.p, (size_t)U * D * 8, hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    return PLDA_OK;
  });
}

// ---------------------------------------------------------------- LDA (python/liblda/lda.py)
int plda_lda_fit_dev(plda_handle *h, const double *dX, int64_t N, int32_t D, const uint64_t *dlabels, int64_t K,
                     int32_t solver, const double *priors) {
  return guarded(h, "plda_lda_fit_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return lda_fit_device(h, dX, N, D, dlabels, K, solver, priors);
  });
}

int plda_lda_fit(plda_handle *h, const double *X, int64_t N, int32_t D, const uint64_t *labels, int32_t solver,
                 const double *priors) {
  return guarded(h, "plda_lda_fit", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!X || !labels || N <= 0 || D <= 0) return fail(h, PLDA_E_INVAL, "lda_fit: bad argument");
    uint64_t mx = 0;
    for (int64_t r = 0; r < N; ++r) mx = std::max(mx, labels[r]);
    if (mx >= (uint64_t)N) return fail(h, PLDA_E_LABELS, "lda_fit: labels must be dense 0..K-1");
    PLDA_TRY(set_device(h));
    Tmp dX, dL;
    PLDA_TRY(upload(h, dX, X, (size_t)N * D * 8));
    PLDA_TRY(upload(h, dL, labels, (size_t)N * 8));
    return lda_fit_device(h, dX.as<double>(), N, D, dL.as<uint64_t>(), (int64_t)mx + 1, solver, priors);
  });
}

int plda_lda_dims(plda_handle *h, int64_t *K, int32_t *D, int32_t *rank, int32_t *solver) {
  return guarded(h, "plda_lda_dims", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!h->lda_fitted) return fail(h, PLDA_E_NOT_FITTED, "This LDA instance is not fitted yet");
    if (K) *K = h->lda_K;
    if (D) *D = h->lda_D;
    if (rank) *rank = h->lda_rank;
    if (solver) *solver = h->lda_solver;
    return PLDA_OK;
  });
}

int plda_lda_get_model(plda_handle *h, double *priors, double *means, double *xbar, double *scalings, double *coef,
                       double *intercept, double *evr) {
  return guarded(h, "plda_lda_get_model", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!h->lda_fitted) return fail(h, PLDA_E_NOT_FITTED, "This LDA instance is not fitted yet");
    PLDA_TRY(set_device(h));
    const size_t K = (size_t)h->lda_K, D = (size_t)h->lda_D, R = (size_t)h->lda_rank;
    auto get = [&](double *dst, const DevBuf &src, size_t n) -> hipError_t {
      return (dst && n) ? hipMemcpyAsync(dst, src.p, n * 8, hipMemcpyDeviceToHost, h->stream) : hipSuccess;
    };
    PLDA_HIP(h, get(priors, h->l_priors, K));
    PLDA_HIP(h, get(means, h->l_means, K * D));
    PLDA_HIP(h, get(coef, h->l_coef, K * D));
    PLDA_HIP(h, get(intercept, h->l_intercept, K));
    if (h->lda_solver == 0) PLDA_HIP(h, get(xbar, h->l_xbar, D));
    if (h->lda_solver != 2) PLDA_HIP(h, get(scalings, h->l_scalings, D * R));
    if (h->lda_solver == 1) PLDA_HIP(h, get(evr, h->l_evr, D));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    return PLDA_OK;
  });
}

int plda_lda_set_model(plda_handle *h, int32_t solver, int64_t K, int32_t D, int32_t rank, const double *priors,
                       const double *means, const double *xbar, const double *scalings, const double *coef,
                       const double *intercept) {
  return guarded(h, "plda_lda_set_model", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (solver < 0 || solver > 2 || K <= 0 || D <= 0 || rank < 0 || rank > D || !coef || !intercept)
      return fail(h, PLDA_E_INVAL, "lda_set_model: bad argument");
    if (solver != 2 && (!scalings || rank == 0)) return fail(h, PLDA_E_INVAL, "lda_set_model: scalings required");
    if (solver == 0 && !xbar) return fail(h, PLDA_E_INVAL, "lda_set_model: xbar required for the svd solver");
    PLDA_TRY(set_device(h));
    auto put = [&](DevBuf &dst, const double *src, size_t n) -> hipError_t {
      hipError_t e = dst.reserve((n ? n : 1) * 8);
      if (e != hipSuccess || !src || !n) return e;
      return hipMemcpyAsync(dst.p, src, n * 8, hipMemcpyHostToDevice, h->stream);
    };
    const size_t k = (size_t)K, d = (size_t)D;
    PLDA_HIP(h, put(h->l_priors, priors, k));
    PLDA_HIP(h, put(h->l_means, means, k * d));
    PLDA_HIP(h, put(h->l_xbar, xbar, d));
    PLDA_HIP(h, put(h->l_scalings, scalings, d * (size_t)rank));
    PLDA_HIP(h, put(h->l_coef, coef, k * d));
    PLDA_HIP(h, put(h->l_intercept, intercept, k));
    PLDA_HIP(h, h->l_evr.reserve(d * 8));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    h->lda_fitted = true; h->lda_solver = solver; h->lda_K = K; h->lda_D = D; h->lda_rank = rank;
    return PLDA_OK;
  });
}

int plda_lda_predict_dev(plda_handle *h, const double *dX, int64_t N, int32_t mode, double *dout) {
  return guarded(h, "plda_lda_predict_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return lda_predict_device(h, dX, N, mode, dout);
  });
}

int plda_lda_predict(plda_handle *h, const double *X, int64_t N, int32_t D, int32_t mode, double *out) {
  return guarded(h, "plda_lda_predict", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!h->lda_fitted) return fail(h, PLDA_E_NOT_FITTED, "This LDA instance is not fitted yet");
    if (D != h->lda_D)
      return fail(h, PLDA_E_INVAL, "X has %d features per sample; expecting %d", D, h->lda_D);   // lda.py:264-266
    if (N <= 0) return PLDA_OK;
    if (!X || !out) return fail(h, PLDA_E_INVAL, "lda_predict: bad argument");
    PLDA_TRY(set_device(h));
    // row slabs bound the device footprint of the [N, K] result
    const int64_t slab = std::max<int64_t>(1, std::min<int64_t>(N, ((int64_t)1 << 28) / std::max<int64_t>(h->lda_K, D)));
    Tmp dX, dO;
    PLDA_HIP(h, dX.alloc((size_t)slab * D * 8));
    PLDA_HIP(h, dO.alloc((size_t)slab * h->lda_K * 8));
    for (int64_t r0 = 0; r0 < N; r0 += slab) {
      const int64_t r = std::min(slab, N - r0);
      PLDA_HIP(h, hipMemcpyAsync(dX.p, X + r0 * D, (size_t)r * D * 8, hipMemcpyHostToDevice, h->stream));
      PLDA_TRY(lda_predict_device(h, dX.as<double>(), r, mode, dO.as<double>()));
      PLDA_HIP(h, hipMemcpyAsync(out + r0 * h->lda_K, dO.p, (size_t)r * h->lda_K * 8, hipMemcpyDeviceToHost, h->stream));
      PLDA_HIP(h, hipStreamSynchronize(h->stream));
    }
    return PLDA_OK;
  });
}

int plda_lda_transform_dev(plda_handle *h, const double *dX, int64_t N, int32_t ncomp, double *dout) {
  return guarded(h, "plda_lda_transform_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return lda_transform_device(h, dX, N, ncomp, dout);
  });
}

int plda_lda_transform(plda_handle *h, const double *X, int64_t N, int32_t D, int32_t ncomp, double *out) {
  return guarded(h, "plda_lda_transform", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!h->lda_fitted) return fail(h, PLDA_E_NOT_FITTED, "This LDA instance is not fitted yet");
    if (D != h->lda_D) return fail(h, PLDA_E_INVAL, "X has %d features per sample; expecting %d", D, h->lda_D);
    if (N <= 0 || ncomp <= 0) return PLDA_OK;
    if (!X || !out) return fail(h, PLDA_E_INVAL, "lda_transform: bad argument");
    PLDA_TRY(set_device(h));
    Tmp dX, dO;
    PLDA_TRY(upload(h, dX, X, (size_t)N * D * 8));
    PLDA_HIP(h, dO.alloc((size_t)N * ncomp * 8));
    PLDA_TRY(lda_transform_device(h, dX.as<double>(), N, ncomp, dO.as<double>()));
    return download(h, out, dO.p, (size_t)N * ncomp * 8);
  });
}

// ---------------------------------------------------------------- HTK feature files
int plda_htk_frames_dev(plda_handle *h, const void *dblob, const int64_t *dfile_off, const int64_t *dframe_off,
                        int64_t U, int64_t T, int32_t samplesize, int32_t frm_ext, float *dout) {
  return guarded(h, "plda_htk_frames_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return htk_frames_device(h, dblob, dfile_off, dframe_off, U, T, samplesize, frm_ext, dout);
  });
}

int plda_htk_frames(plda_handle *h, const void *blob, int64_t blob_bytes, const int64_t *file_off,
                    const int64_t *frame_off, int64_t U, int32_t samplesize, int32_t frm_ext, float *out) {
  return guarded(h, "plda_htk_frames", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (U <= 0) return PLDA_OK;
    if (!blob || !file_off || !frame_off || !out || blob_bytes < 0 || samplesize <= 0 || frm_ext < 0)
      return fail(h, PLDA_E_INVAL, "htk_frames: bad argument");
    if (samplesize % 4) return fail(h, PLDA_E_INVAL, "htk_frames: samplesize %d is not a multiple of 4", samplesize);
    const int64_t W = samplesize / 4;
    for (int64_t u = 0; u < U; ++u) {
      const int64_t n = frame_off[u + 1] - frame_off[u];
      if (n < 0 || file_off[u] < 0 || (file_off[u] + n * W) * 4 > blob_bytes)
        return fail(h, PLDA_E_INVAL, "htk_frames: file %lld does not fit the blob (pad short files with zeros)", (long long)u);
    }
    const int64_t T = frame_off[U] - frame_off[0];
    if (frame_off[0] != 0) return fail(h, PLDA_E_INVAL, "htk_frames: frame_off[0] must be 0");
    if (T <= 0) return PLDA_OK;
    PLDA_TRY(set_device(h));
    Tmp dB, dF, dO, dOut;
    PLDA_TRY(upload(h, dB, blob, (size_t)blob_bytes));
    PLDA_TRY(upload(h, dF, file_off, (size_t)U * 8));
    PLDA_TRY(upload(h, dO, frame_off, (size_t)(U + 1) * 8));
    const size_t obytes = (size_t)T * (2 * frm_ext + 1) * samplesize;
    PLDA_HIP(h, dOut.alloc(obytes));
    PLDA_TRY(htk_frames_device(h, dB.p, dF.as<int64_t>(), dO.as<int64_t>(), U, T, samplesize, frm_ext, dOut.as<float>()));
    return download(h, out, dOut.p, obytes);
  });
}

// ---------------------------------------------------------------- multi-GPU (comm.hip)
int plda_comm_init(plda_handle *h, int32_t nranks, int32_t rank, const void *unique_id) {
  return guarded(h, "plda_comm_init", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return comm_init(h, nranks, rank, unique_id);
  });
}

int plda_comm_init_custom(plda_handle *h, int32_t nranks, int32_t rank, const plda_collectives *table) {
  return guarded(h, "plda_comm_init_custom", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return comm_init_custom(h, nranks, rank, table);
  });
}

int plda_comm_init_peer(plda_handle *h, int32_t nranks, int32_t rank, const plda_host_collectives *bootstrap) {
  return guarded(h, "plda_comm_init_peer", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return comm_init_peer(h, nranks, rank, bootstrap);
  });
}

int plda_comm_init_host(plda_handle *h, int32_t nranks, int32_t rank, const plda_host_collectives *table) {
  return guarded(h, "plda_comm_init_host", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return comm_init_host(h, nranks, rank, table);
  });
}

int plda_comm_destroy(plda_handle *h) {
  return guarded(h, "plda_comm_destroy", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return comm_destroy(h);
  });
}

int plda_comm_describe(plda_handle *h, char *json, int64_t cap) {
  return guarded(h, "plda_comm_describe", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!json || cap <= 0) return fail(h, PLDA_E_INVAL, "comm_describe: bad argument");
    PLDA_TRY(set_device(h));
    std::string js;
    PLDA_TRY(comm_describe(h, js));
    if ((int64_t)js.size() + 1 > cap) return fail(h, PLDA_E_CAPACITY, "comm_describe: need %zu bytes", js.size() + 1);
    std::memcpy(json, js.c_str(), js.size() + 1);
    return PLDA_OK;
  });
}

int plda_comm_emulate(plda_handle *h, int32_t nranks, int32_t rank) {
  return guarded(h, "plda_comm_emulate", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (h->comm) return fail(h, PLDA_E_INVAL, "comm_emulate: this handle has a real communicator");
    if (nranks <= 0 || rank < 0 || rank >= nranks) return fail(h, PLDA_E_INVAL, "comm_emulate: bad argument");
    h->comm_nranks = nranks; h->comm_rank = rank;
    return PLDA_OK;
  });
}

int plda_comm_info(plda_handle *h, int32_t *nranks, int32_t *rank) {
  return guarded(h, "plda_comm_info", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (nranks) *nranks = h->comm_nranks;
    if (rank) *rank = h->comm_rank;
    return PLDA_OK;
  });
}

int plda_score_matrix_sharded_dev(plda_handle *h, const double *dU, const int32_t *dn_enrol, int32_t n_uniform, int64_t M,
                                  const double *dV, int64_t Nt, const double *dzmean, const double *dzstd, float *dout,
                                  int64_t ld_out, int64_t block_rows, int32_t gather) {
  return guarded(h, "plda_score_matrix_sharded_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!dn_enrol && n_uniform <= 0) return fail(h, PLDA_E_INVAL, "score_matrix_sharded: n_uniform must be > 0 when n_enrol is NULL");
    PLDA_TRY(set_device(h));
    return score_matrix_sharded_device(h, dU, dn_enrol, n_uniform, M, dV, Nt, dzmean, dzstd, nullptr, 0, dout, ld_out,
                                       block_rows, gather);
  });
}

int plda_score_matrix_sharded_local_dev(plda_handle *h, const double *dU, const int32_t *dn_enrol, int32_t n_uniform,
                                        int64_t M, const double *dV, int64_t Nt, const double *dzmean, const double *dzstd,
                                        float *dlocal, int64_t ld_local, int64_t block_rows, float *dfull, int64_t ld_full) {
  return guarded(h, "plda_score_matrix_sharded_local_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!dn_enrol && n_uniform <= 0) return fail(h, PLDA_E_INVAL, "score_matrix_sharded: n_uniform must be > 0 when n_enrol is NULL");
    if (!dlocal) return fail(h, PLDA_E_INVAL, "score_matrix_sharded_local: dlocal is NULL");
    PLDA_TRY(set_device(h));
    return score_matrix_sharded_device(h, dU, dn_enrol, n_uniform, M, dV, Nt, dzmean, dzstd, dlocal, ld_local, dfull, ld_full,
                                       block_rows, dfull ? 1 : 0);
  });
}

int plda_znorm_stats_sharded_dev(plda_handle *h, const double *dbkg, int64_t Nb, int32_t num_examples, int32_t Din,
                                 const double *dmodels, int64_t M, double *dout_mean, double *dout_std) {
  return guarded(h, "plda_znorm_stats_sharded_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "norm: model not fitted");
    if (!dbkg || !dmodels || !dout_mean || !dout_std || Nb <= 0 || M <= 0) return fail(h, PLDA_E_INVAL, "norm: bad argument");
    PLDA_TRY(set_device(h));
    return znorm_stats_sharded_device(h, dbkg, Nb, num_examples, Din, dmodels, M, dout_mean, dout_std);
  });
}

int plda_fit_sharded_dev(plda_handle *h, const double *dX, int64_t N, int32_t D, const uint64_t *dlabels, int64_t K,
                         int32_t iters) {
  return guarded(h, "plda_fit_sharded_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return fit_sharded_device(h, dX, N, D, dlabels, K, iters);
  });
}

int plda_eer_matrix_comm_dev(plda_handle *h, const float *dscores, int64_t ld, int64_t M, int64_t Nt,
                             const int64_t *denrol_spk, const int64_t *dtest_spk, double *out) {
  return guarded(h, "plda_eer_matrix_comm_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return eer_matrix_comm_device(h, dscores, ld, M, Nt, denrol_spk, dtest_spk, out);
  });
}

// ---------------------------------------------------------------- EER
int plda_eer_matrix_dev(plda_handle *h, const float *dscores, int64_t ld, int64_t M, int64_t Nt,
                        const int64_t *denrol_spk, const int64_t *dtest_spk, double *out) {
  return guarded(h, "plda_eer_matrix_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return eer_matrix_device(h, dscores, ld, M, Nt, denrol_spk, dtest_spk, out);
  });
}

int plda_score_eer_dev(plda_handle *h, const double *dU, const int32_t *dn_enrol, int32_t n_uniform, int64_t M, const double *dV,
                       int64_t Nt, const double *dzmean, const double *dzstd, const int64_t *denrol_spk, const int64_t *dtest_spk,
                       double *out) {
  return guarded(h, "plda_score_eer_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return score_eer_device(h, dU, dn_enrol, n_uniform, M, dV, Nt, dzmean, dzstd, denrol_spk, dtest_spk, out);
  });
}

int plda_eer_matrix_sharded_dev(plda_handle *h, const float *dscores, int64_t ld, int64_t M, int64_t Nt,
                                const int64_t *denrol_spk, const int64_t *dtest_spk, plda_eer_reduce_fn reduce, void *ctx,
                                double *out) {
  return guarded(h, "plda_eer_matrix_sharded_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!reduce) return fail(h, PLDA_E_INVAL, "eer_matrix_sharded: a reduction callback is required");
    PLDA_TRY(set_device(h));
    return eer_matrix_device(h, dscores, ld, M, Nt, denrol_spk, dtest_spk, out, reduce, ctx);
  });
}

int plda_det_matrix_dev(plda_handle *h, const float *dscores, int64_t ld, int64_t M, int64_t Nt, const int64_t *denrol_spk,
                        const int64_t *dtest_spk, int32_t n_points, double *far, double *frr, double *thresholds) {
  return guarded(h, "plda_det_matrix_dev", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    PLDA_TRY(set_device(h));
    return det_matrix_device(h, dscores, ld, M, Nt, denrol_spk, dtest_spk, n_points, far, frr, thresholds);
  });
}

int plda_det_lists(plda_handle *h, const float *pos, int64_t np, const float *neg, int64_t nn, int32_t n_points, double *far,
                   double *frr, double *thresholds) {
  return guarded(h, "plda_det_lists", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!pos || !neg || !far || !frr || np <= 0 || nn <= 0)
      return fail(h, PLDA_E_INVAL, "det: need at least one target and one impostor score");
    PLDA_TRY(set_device(h));
    Tmp dP, dN;
    PLDA_TRY(upload(h, dP, pos, (size_t)np * 4));
    PLDA_TRY(upload(h, dN, neg, (size_t)nn * 4));
    const int rc = det_lists_device(h, dP.as<float>(), np, dN.as<float>(), nn, n_points, far, frr, thresholds);
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    return rc;
  });
}

int plda_eer_lists(plda_handle *h, const float *pos, int64_t np, const float *neg, int64_t nn, double *out) {
  return guarded(h, "plda_eer_lists", [&]() -> int {
    if (!h) return PLDA_E_INVAL;
    PLDA_LOCK(h);
    if (!pos || !neg || !out || np <= 0 || nn <= 0)
      return fail(h, PLDA_E_INVAL, "eer: need at least one target and one impostor score");
    PLDA_TRY(set_device(h));
    Tmp dP, dN;
    PLDA_TRY(upload(h, dP, pos, (size_t)np * 4));
    PLDA_TRY(upload(h, dN, neg, (size_t)nn * 4));
    return eer_lists_device(h, dP.as<float>(), np, dN.as<float>(), nn, out);
  });
}

}  // extern "C"
