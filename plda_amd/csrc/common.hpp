// plda_amd/csrc/common.hpp -- handle, error plumbing and device buffers shared by
// the translation units of libplda_hip.so (gfx950 only; no CPU fallback).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <atomic>
#include <vector>

#include "../../include/plda_hip.h"

// PLDA_DIAG=1 (plda_amd/build.py --diag -> libplda_hip_diag.so): the measurement arms of the trials GEMM -- bounding arms that
// skip the operand DMA or the stores and return GARBAGE scores, per-wave clock stamps, stage-depth sweeps -- are compiled in and
// selectable through PLDA_GEMM_VARIANT.  The product library is built without it: those kernels are not instantiated and
// plda_create refuses their variant numbers (round-5 review: a stray environment variable must not be able to corrupt scores).
#ifndef PLDA_DIAG
#define PLDA_DIAG 0
#endif

namespace plda {

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// RAII device temporary of the host-pointer entry points
struct Tmp {
  void *p = nullptr;
  ~Tmp() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
  template <typename T> T *as() { return reinterpret_cast<T *>(p); }
};

}  // namespace plda

namespace plda { struct HostPipe; }

namespace plda {
// The distinct enrol counts of a mixed-count trials call (score.hip, "bucketed" path): bucket g <-> vals[g], ascending.
// Passed BY VALUE to the kernels that need it (260 bytes of kernel arguments; no device-side table to keep coherent).
constexpr int CS_MAX = 64;        // more distinct counts than this: the depth-2D form
constexpr int CS_NMAX = 4095;     // a count above this: the depth-2D form
struct CountSet { int G = 0; int32_t vals[CS_MAX] = {}; };
// One (btM, btN) tile grid of trials_gemm_bt4_kernel: per-XCD queues over a table of tiles (score.hip: bt4_schedule)
struct Bt4Table { int btM = -1, btN = -1, colwalk = 0; DevBuf tab; int qbase[8] = {}, qlen[8] = {}; uint64_t used = 0; };
}  // namespace plda

// a chunk of rows of ONE source and its coefficients in the EM's two rank-k sums (linalg.hip: em_rank_sums_mstep_f64)
struct SyrkChunk { const double *rows; int nrows; int pad; double c1, c2; };

struct plda_handle {
  std::recursive_mutex mu;   // taken by every C-ABI entry point (api.hip)
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;

  // ---- model (Kaldi `Plda`, pldamodule.cpp:29): host mirror + device copies ----
  bool fitted = false;
  int Dout = 0, Din = 0;
  std::vector<double> h_mean, h_transform, h_psi, h_offset;
  plda::DevBuf d_mean, d_transform, d_psi, d_offset;
  void *pin_model = nullptr;     // pinned landing area of the model copies that end a fit (mean | transform | psi | offset)
  size_t pin_model_cap = 0;

  // ---- fit state kept for plda_fit_get_stats ----
  int64_t fit_K = 0;
  int fit_D = 0;
  plda::DevBuf f_means, f_counts, f_scatter, f_sum, f_W, f_B;
  plda::DevBuf fit_flag;         // the EM's factorisation flag (read by export_model_kernel at the end of a fit)
  double fit_ms[4] = {0, 0, 0, 0};
  hipEvent_t fit_ev[3] = {nullptr, nullptr, nullptr};   // EM start / end, statistics start on the stream (fit.hip)
  int *fit_dbad = nullptr;       // label-check flags of a statistics pass whose read-back fit_em_device takes over (plda_fit)
  double fit_t0 = 0.0;           // host clock at the start of that pass

  // ---- LDA model (lda.hip; /root/reference/python/liblda/lda.py) ----
  bool lda_fitted = false;
  int lda_solver = 0, lda_D = 0, lda_rank = 0;
  int64_t lda_K = 0;
  plda::DevBuf l_means, l_priors, l_xbar, l_scalings, l_coef, l_intercept, l_evr;

  // ---- host-pointer entry points: pinned ring + copy threads (hostio.hip), two alternating score slabs ----
  plda::HostPipe *hostpipe = nullptr;
  plda::DevBuf hio_O[2];
  int host_variant = 0;          // PLDA_HOST_VARIANT=1: the serial pageable-copy arm of rounds 1-2 (A/B)

  // ---- one-trial score() on the host mirror of psi (plda_score_one): per-n coefficient cache ----
  uint64_t model_epoch = 0;      // bumped whenever the model changes (fit, set_model, truncate, smooth)
  struct OneTrial { uint64_t epoch = ~0ull; int n = 0, D = 0; double logterm = 0.0; std::vector<double> c, ivar, ipsi1; } one;

  // ---- one-trial score through the trial-list kernel: host buffer mapped into the device address space ----
  void *one_host = nullptr, *one_dev = nullptr;
  size_t one_cap = 0;

  // ---- scoring workspace ----
  plda::DevBuf s_Apk, s_Bpk, s_rbias, s_rscale, s_cbias, s_rpair, s_cpair;
  plda::DevBuf s_A16, s_B16;   // the packed operands as three bf16 planes per k-oct (score_bf16x3.inc: opt-in arm)
  int score_dtype = 0;         // PLDA_SCORE_DTYPE=bf16x3 -> 1: the trials GEMM's contraction as three bf16 terms (opt-in; default fp32 MFMA)
  plda::DevBuf tf_pad;   // zero-padded copy of the transform for the one-pass K4 kernel, cached per model
  uint64_t tf_pad_epoch = ~0ull;
  int tf_pad_rows = 0, tf_pad_dinp = 0;
  int64_t last_M = 0, last_Nt = 0;
  const char *last_kernel = nullptr;   // the trials-GEMM kernel of the last score_matrix launch (static string)
  int last_k = 0;
  // a test side packed ahead of time (plda_score_prepare_dev / _counts_dev): reused while pointer, size, model and count
  // kind match.  prep_kind: 0 uniform count (prep_nuniform), 1 mixed counts in the depth-2D form, 2 mixed counts in the
  // bucketed form for the count set prep_counts
  bool prep_valid = false;
  int prep_kind = 0;
  plda::CountSet prep_counts;
  const double *prep_dV = nullptr;
  int64_t prep_Nt = 0;
  uint64_t prep_epoch = 0;
  unsigned long long prep_fp = 0;   // content fingerprint of the prepared rows (score.hip: fingerprint_kernel)
  int prep_nuniform = 0;

  // ---- Jacobi sweep graph + warm-start state (linalg.hip) ----
  hipGraphExec_t jac_exec = nullptr;
  double *jac_G = nullptr, *jac_V = nullptr;
  int jac_D = 0;
  long jac_total_sweeps = 0;
  int simdiag_D = 0;
  bool simdiag_has_vr = false;
  bool eig_keep_sign = false;   // LDA: keep negative eigenvalues (PLDA floors them, Kaldi ApplyFloor)

  bool panel_attr_set[16] = {};
  int num_cus = 256;           // hipDeviceAttributeMultiprocessorCount (persistent grids)
  int sort_variant = 0;        // PLDA_SORT_VARIANT=1: fit groups the rows by the radix sort always (0: by counting where the tables fit)
  int transform_variant = 0;   // PLDA_TRANSFORM_VARIANT=1: general GEMM + separate length-norm pass; 2: no tail launch (A/B arms)
  int gemm_variant = 0;
  int mixed_variant = 0;   // PLDA_MIXED_VARIANT=1: mixed enrol counts always in the depth-2D form [A1 | A2] x [V | V*V] (A/B arm of the bucketed form)
  const double *gcoef_ptr = nullptr; uint64_t gcoef_epoch = 0; int gcoef_D = 0; plda::CountSet gcoef_set;   // bucket tables in w[11], same rule
  const double *ucoef_ptr = nullptr; uint64_t ucoef_epoch = 0; int ucoef_n = 0, ucoef_D = 0;   // uniform-count coefficients in w[11] (score.hip: prepare_operands)
  int prep_variant = 0;    // PLDA_PREP_VARIANT=1: scoring prep as separate bias / pack / pair kernels (A/B arm of prep_side_kernel)
  int gemm64_variant = 0;  // PLDA_GEMM64_VARIANT=1: fp64 GEMM always on 64 x 64 tiles (A/B arm)
  int jacobi_variant = 0;  // 0: Gram-form block Jacobi round; 1: rotation-by-rotation inner tournament
  int sweep_variant = 0;  // PLDA_SWEEP_VARIANT=1: the 16-wave register kernels of round 2 (SPD inverse, tridiagonalisation); 2: the four-wave scalar sweep at every size (no matrix-core block sweep)
  bool sweep_mfma_attr[17] = {};   // dynamic-LDS attribute of spd_inverse_mfma_kernel<NT> set
  int em_variant = 0;     // 0: grouped closed-form EM (moment or row form by shape); 3 / 4: the moment / the row form always; 1: EM in the simultaneously-diagonalised basis
  int em_groups = 0;      // groups (distinct class counts) of the last grouped EM, 0 if the other path ran
  int em_form = 0;        // EM of the last fit: 0 simultaneously-diagonalised basis, 1 grouped moment form, 2 grouped row form
  bool bt2_attr_set = false;
  bool bt4_attr_set = false;
  // tile schedule of the one-wave-per-SIMD trials GEMM (score.hip: bt4_schedule): per XCD a queue of tiles, consumed
  // through one device-scope counter per queue; the table is rebuilt when the tile grid changes
  plda::DevBuf bt4_cnt, bt4_fringe;   // (bt4_fringe: 256 x 256 scratch slots of the tiles that cross the matrix edge)
  plda::Bt4Table bt4_tabs[6];         // the last few tile grids (sharded scoring alternates full blocks and a ragged tail)
  uint64_t bt4_clock = 0;
  // distinct enrol counts of a mixed-count call: presence bytes + result on the device, pinned landing area
  plda::DevBuf cs_work;
  void *cs_pin = nullptr; int cs_seq = 0;   // pinned words of the count-set result + the sequence number its kernel stores last (score.hip)
  bool timeline_valid = false;   // `timeline` holds the stamps of a PLDA_GEMM_VARIANT=31 launch
  plda::DevBuf timeline;

  // ---- profiling (plda_profile_*): event pairs around each trials-GEMM launch ----
  bool prof_on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
  double prof_flop = 0.0;
  // ---- PLDA_HIP_TRACE / plda_trace_*: named spans (HIP event pairs on the stream) around the stages of fit,
  // transform and scoring; unit 1 = flop, 2 = bytes of algorithmic work ----
  struct TraceSpan { const char *name; hipEvent_t e0, e1; double work; int unit; };
  bool trace_on = false, trace_print = false;
  std::vector<TraceSpan> trace_spans;
  size_t trace_used = 0;

  // ---- multi-GPU (comm.hip): the collective table (RCCL, host-staged or caller-supplied), side stream for the
  // gather, ordering events.  comm_kind: 0 none / emulated, 1 RCCL, 2 host-staged, 3 custom ----
  plda_collectives coll = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int comm_kind = 0;
  bool comm = false;             // a communicator is installed (collectives run)
  int comm_nranks = 1, comm_rank = 0;
  hipStream_t comm_stream = nullptr;
  hipEvent_t comm_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  plda::DevBuf comm_mm, comm_mc;   // merged centroids / counts of a sharded fit (persistent: peers map them)

  // ---- general scratch (fit / transform / znorm) ----
  plda::DevBuf w[16];
  plda::DevBuf em_chunks;                       // the row-form EM's chunk table (device) ...
  std::vector<SyrkChunk> em_chunks_host;        // ... and its host copy, alive while the upload is in flight
  plda::DevBuf eigdc;            // eig_dc.hip workspace
  plda::DevBuf zn_rows, zn_y, zn_small, zn_cpad;   // z-norm statistics by moments (score.hip; zn_cpad: the padded covariance of the model pass, transform.hip)
  int znorm_variant = 0;         // PLDA_ZNORM_VARIANT=1: every LLR on the fused fp32 GEMM (A/B arm); 2: moments in five passes
  plda::DevBuf eer_slab, eer_smp; // plda_score_eer_dev: the row slab of scores in flight; the pilot's gathered enrol rows
  plda::DevBuf eer_list[2];      // eer.hip, single-pass form: the impostor / target scores inside the pilot's key window
  int eer_variant = 0;           // PLDA_EER_VARIANT=1: always the three passes; 2: the single-pass form at every size (tests)
  int64_t eer_slab_rows = 0;     // PLDA_EER_SLAB_ROWS: rows per slab of plda_score_eer_dev (0: <= 4 GiB of scores)
  int eer_last_passes = 0;       // full passes over the matrix the last plda_eer_matrix_dev made (1 or 3)
  const int *eigdc_flag = nullptr;   // device flag of the last direct decomposition (sym_eig_dc_status)
  int eig_variant = 0;           // PLDA_EIG_VARIANT: 0 = direct method where supported, 1 = block Jacobi always
  int eig_debug = 0;             // PLDA_EIG_DEBUG (timing experiments only: results are wrong when set)
  int eig_last_method = 0;       // 1 = block Jacobi, 2 = tridiagonalisation + divide and conquer
};

namespace plda {

int fail(plda_handle *h, int code, const char *fmt, ...);
int hip_fail(plda_handle *h, hipError_t e, const char *what, const char *file, int line);

#define PLDA_HIP(h, expr)                                                          \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) return plda::hip_fail((h), _e, #expr, __FILE__, __LINE__); \
  } while (0)

#define PLDA_TRY(...)             \
  do {                            \
    int _rc = (__VA_ARGS__);      \
    if (_rc != PLDA_OK) return _rc; \
  } while (0)

#define PLDA_LAUNCH_CHECK(h) PLDA_HIP(h, hipGetLastError())

// RAII span of the trace: records an event on h->stream at construction and destruction (nothing when tracing is off;
// never inside a stream capture).  `name` must be a string literal.
struct TraceScope {
  plda_handle *h;
  long idx = -1;
  TraceScope(plda_handle *hh, const char *name, double work = 0.0, int unit = 0) : h(hh) { open(name, work, unit); }
  // closes the running span and opens the next stage's
  void next(const char *name, double work = 0.0, int unit = 0) {
    close();
    open(name, work, unit);
  }
  void close() {
    if (idx >= 0) (void)hipEventRecord(h->trace_spans[idx].e1, h->stream);
    idx = -1;
  }
  void open(const char *name, double work, int unit) {
    if (!h->trace_on) return;
    // bounded: a long-running process that never reads the trace keeps its first 8192 spans (16 384 events), not an
    // ever-growing vector; plda_trace_read(reset) or plda_destroy start over
    if (h->trace_used >= 8192) return;
    if (h->trace_used == h->trace_spans.size()) {
      plda_handle::TraceSpan sp{name, nullptr, nullptr, 0.0, 0};
      if (hipEventCreate(&sp.e0) != hipSuccess || hipEventCreate(&sp.e1) != hipSuccess) return;
      h->trace_spans.push_back(sp);
    }
    idx = (long)h->trace_used++;
    auto &sp = h->trace_spans[idx];
    sp.name = name; sp.work = work; sp.unit = unit;
    (void)hipEventRecord(sp.e0, h->stream);
  }
  ~TraceScope() { close(); }
};

// Function attributes (the dynamic-LDS opt-in) are per DEVICE, and handles on different GPUs or threads share a launch
// template's statics: one bit per device, set only after the attribute call succeeded (two racing threads both set it).
struct DeviceOnce {
  std::atomic<unsigned long long> mask{0};
  bool needed(int device) const { return !(mask.load(std::memory_order_acquire) & (1ull << (device & 63))); }
  void done(int device) { mask.fetch_or(1ull << (device & 63), std::memory_order_release); }
};

constexpr size_t TIMELINE_WORDS = 8 * 16 * 8 * 8;   // [tile < 8][stage < 16][wave < 8][8] shader-clock stamps

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
inline int64_t ceil_div(int64_t x, int64_t m) { return (x + m - 1) / m; }

// upload model arrays held in h->h_* to the device copies
int model_to_device(plda_handle *h);

// ---- linalg.hip (fp64 building blocks; all enqueue on h->stream) ----
// C[M,N] (ldc) = alpha * sum_k A(m,k) * kw[k]? * B(k,n) + beta * C
// A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]; exactly one of (sam,sak)
// and one of (sbk,sbn) must be 1.  kw nullable.  Uses split-K with a
// deterministic second-stage reduction when K is long and M*N small.
int gemm_f64_batched(plda_handle *h, int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t sam,
                     int64_t sak, int64_t strideA, const double *B, int64_t sbk, int64_t sbn, int64_t strideB,
                     const double *kw, double beta, double *C, int64_t ldc, int64_t strideC, int batch);
// C = X^T diag(kw) X + w2 X2^T X2 (one launch for D <= 208)
int syrk_pair_f64(plda_handle *h, int D, int64_t K1, const double *X, int64_t ldx, const double *kw, int64_t K2,
                  const double *X2, int64_t ldx2, double w2, double *C, int64_t ldc);
// the row-form EM's two rank-k sums + M-step in two launches (D <= 208; *used = false otherwise): a table of row chunks with
// their coefficients in the two sums (built once per fit), Bout != Bin
int em_rank_chunk_bound(int G, int D, int64_t K, int cus);
int em_rank_chunks(int G, int D, int64_t K, int cus, const double *X, const double *gn, const double *gk, const double *Z,
                   const double *Wn, SyrkChunk *out);
int em_rank_sums_mstep_f64(plda_handle *h, int D, const SyrkChunk *chunks, int nchunks, const double *S, double sumK, double cw,
                           double cntW, double cntB, double *W, const double *Bin, double *Bout, bool *used);
int syrk_znorm_f64(plda_handle *h, int D0, int64_t K, const double *X, const double *zc, const double *zs, double *C, bool *used);
int gemm_f64(plda_handle *h, int64_t M, int64_t N, int64_t K, double alpha, const double *A,
             int64_t sam, int64_t sak, const double *B, int64_t sbk, int64_t sbn,
             const double *kw, double beta, double *C, int64_t ldc);
// in-place lower Cholesky (upper triangle zeroed); *dflag (device int) set to 1 on failure
// X = L^{-1} for lower-triangular L (row-major); X written fully (upper = 0)
// up to three independent batched products C = A B of one shape in one launch (M, N, K <= 256), else sequentially
struct GemmSet {
  const double *A; int64_t sam, sak, strideA;
  const double *B; int64_t sbk, sbn, strideB;
  double *C; int64_t ldc, strideC;
};
int gemm_f64_multi(plda_handle *h, int64_t M, int64_t N, int64_t K, const GemmSet *sets, int nsets, int batch);
int spd_inverse_f64(plda_handle *h, const double *W, const double *B, const double *gn, int D, double *out,
                    int *dflag, int batch);
// T_g = chol(W + gn[g] B)^-1 (lower triangular) for g < batch; scr: 3 D^2 doubles per group
int whiten_groups_f64(plda_handle *h, const double *W, const double *B, const double *gn, int D, double *T, double *scr,
                      int *dflag, int batch);
// [batch] SPD inverses of any size (A^-1 = T^T T, T = blocked whitening); out may be A itself; scr: 3 n^2 doubles each
int spd_inverse_blocked(plda_handle *h, const double *A, int n, int lda, int64_t sa, double *out, int ldo,
                        int64_t so, double *scr, int64_t sscr, int *dflag, int batch);
// symmetric eigendecomposition of G (row-major, destroyed): eigenvalues sorted
// descending in s[D] (floored at 0), eigenvectors in the ROWS of Vrows.
int sym_eig_f64(plda_handle *h, double *G, int D, double *s, double *Vrows, int *sweeps_out,
                const double *warm);
// direct method (eig_dc.hip): Householder tridiagonalisation + divide and conquer + back-transformation.
// *status != 0: not supported / gave up -> use sym_eig_f64.  G is not modified.
int sym_eig_dc_f64(plda_handle *h, const double *G, int D, double *s, double *Vrows, int *status);   // status == nullptr: deferred
int sym_eig_dc_status(plda_handle *h, int *status);
int sym_eig_auto_f64(plda_handle *h, double *G, int D, double *s, double *Vrows);   // direct, else block Jacobi
int simdiag_enqueue(plda_handle *h, const double *W, const double *B, int D, double *T, double *Tinv, double *psi,
                    bool *pending);
int simdiag_finish(plda_handle *h, const double *W, const double *B, int D, double *T, double *Tinv, double *psi,
                   bool *redo);
void simdiag_flags(plda_handle *h, int D, const int **chol_flag, const int **eig_flag);
int simdiag_finish_with(plda_handle *h, const double *W, const double *B, int D, double *T, double *Tinv, double *psi,
                        int chol_flag, int eig_status, bool *redo);
// simultaneous diagonalisation of (W,B): T W T^T = I, T B T^T = diag(psi);
// T [D,D], Tinv = T^{-1} (nullable), psi[D].  W,B are not modified.
int simdiag_f64(plda_handle *h, const double *W, const double *B, int D, double *T,
                double *Tinv, double *psi, bool warm_start);

// ---- frontend.hip ----
int htk_frames_device(plda_handle *h, const void *dblob, const int64_t *dfile_off, const int64_t *dframe_off,
                      int64_t U, int64_t T, int samplesize, int frm_ext, float *dout);
int dvector_pool_device(plda_handle *h, const void *dframes, int dtype, int64_t T, int D,
                        const int64_t *doffsets, int64_t U, int method, int l2norm, double *dout);

// ---- fit.hip ----
int group_means_device(plda_handle *h, const double *dX, int64_t N, int D, const uint64_t *ddense,
                       int64_t Ku, double *dmeans, int32_t *dcounts32);
int lda_fit_device(plda_handle *h, const double *dX, int64_t N, int D, const uint64_t *dlabels, int64_t K,
                   int solver, const double *priors_host);
int lda_predict_device(plda_handle *h, const double *dX, int64_t N, int mode, double *dout);
int lda_transform_device(plda_handle *h, const double *dX, int64_t N, int ncomp, double *dout);
int fit_stats_device(plda_handle *h, const double *dX, int64_t N, int D, const uint64_t *dlabels, int64_t K, bool defer_check = false);
int fit_em_device(plda_handle *h, int64_t K, int D, int iters);
int fit_device(plda_handle *h, const double *dX, int64_t N, int D, const uint64_t *dlabels,
               int64_t K, int iters);

// ---- score.hip ----
// cs: the distinct enrol counts of the caller's whole call (mixed counts; nullptr: found from dn on the device)
int score_matrix_device(plda_handle *h, const double *dU, const int32_t *dn, int n_uniform, int64_t M,
                        const double *dV, int64_t Nt, const double *dzmean, const double *dzstd,
                        float *dout, int64_t ld, bool reuse_packed_B = false, const CountSet *cs = nullptr);
void score_count_set_host(const int32_t *n, int64_t M, CountSet *cs);
int pairs_validate_device(plda_handle *h, const int64_t *de, const int64_t *dt, int64_t P, int64_t M, int64_t Nt, long long *bad);
int score_count_set_device(plda_handle *h, const int32_t *dn, int64_t M, CountSet *cs);
int score_prepare_device(plda_handle *h, const double *dV, int64_t Nt, int kind, int n_uniform, const CountSet *cs);
// norm()'s model pass: per row x, out_mean = sum_c x_c (m_c - q_c x_c / 2) + *mD, out_std = sqrt(max(x^T C x + lin . x + *crr, 0)); D <= 208
int quadform_rows_device(plda_handle *h, const double *dX, int64_t R, int D, const double *C, int ldc, const double *lin,
                         const double *m, const double *q, const double *mD, const double *crr, double *out_mean,
                         double *out_std, bool *used);
int transform_rows_device(plda_handle *h, const double *dX, int64_t R, int Din,
                          const int32_t *dn, int n_uniform, double *dout);

#ifdef __HIPCC__
// fp64 wave-wide sum through DPP: quad butterflies, then half-row and row mirrors (every
// lane of a 16-lane row then holds the row sum), then four readlanes.  ~6x shorter
// dependency chain than __shfl_xor, which lowers to ds_bpermute for 64-bit values.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double x, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l),
                          __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ double wave_sum_f64(double x) {
  x += dpp_f64<0xB1>(x);    // quad_perm [1,0,3,2]
  x += dpp_f64<0x4E>(x);    // quad_perm [2,3,0,1]
  x += dpp_f64<0x141>(x);   // row_half_mirror
  x += dpp_f64<0x140>(x);   // row_mirror
  return (readlane_f64(x, 0) + readlane_f64(x, 16)) + (readlane_f64(x, 32) + readlane_f64(x, 48));
}
#endif

}  // namespace plda
