/*
 * include/plda_hip.h -- C ABI of libplda_hip.so, the MI355X (gfx950) PLDA engine.
 *
 * This is the drop-in boundary for the reference's native object
 * `libplda.MPlda` (RicherMans/PLDA src/pldamodule.cpp): every entry point names
 * the reference interface it replaces.  Plain pointers and sizes only; no
 * CPython, NumPy or torch types cross this boundary.  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - every function returns an int status: PLDA_OK (0) or a negative PLDA_E_*;
 *    the text of the last failure on a handle is plda_last_error(h)
 *    (plda_last_error(NULL) = last failure of plda_create on this thread).
 *    Nothing aborts and nothing throws across the ABI (the reference lets Kaldi
 *    assertions abort the process, pldamodule.cpp has no try/catch).
 *  - matrices are row-major, C-contiguous, fp64 (the reference reads every
 *    array as f64: pldamodule.cpp:72, kaldi-utils.hpp:99-111); labels are
 *    uint64 (npy_long read as 8 bytes, pldamodule.cpp:74).
 *  - "host" entry points borrow caller memory for the duration of the call and
 *    write only caller-allocated outputs (ownership: the reference copies its
 *    inputs immediately, kaldi-utils.hpp:99-122).  "_dev" entry points take
 *    pointers into this GPU's HBM and enqueue on the handle's stream without
 *    synchronising.
 *  - one handle = one GPU + one HIP stream.  Every entry point takes the handle's mutex, so calls on
 *    one handle from several threads are safe and serialised (the reference holds the GIL for the
 *    whole call, pldamodule.cpp has no threads); use one handle per thread for concurrency.
 *    plda_destroy must not race with other calls on the same handle.
 *  - there is NO CPU fallback: without a usable gfx950 device plda_create fails.
 *  - feature dimension: 1 ... 2048 (plda_fit*, plda_lda_fit*, plda_sym_eig return PLDA_E_INVAL above).  The
 *    reference has no cap (its own tests stop at 1024, tests/pldatest.py:55); the engine's is the direct eigensolver's
 *    (one workgroup per 8 rows, all 256 CUs at 2048).  Above 1024 ONLY that solver exists: it needs ceil(D/8) co-resident
 *    workgroups, so on a device that exposes fewer CUs (CU masking, a partitioned GPU) a fit / GetOutput / plda_sym_eig
 *    with D in (1024, 2048] returns PLDA_E_INVAL with a message that says so, and plda_sym_eig(method = 1, the block
 *    Jacobi solver) is rejected above 1024 at the boundary.  INTEGRATION.md, "Limits".
 */
#ifndef PLDA_HIP_H_
#define PLDA_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLDA_OK 0
#define PLDA_E_INVAL (-1)     /* bad argument (NULL, non-positive size, dim mismatch) */
#define PLDA_E_ONE_SPEAKER (-2) /* fit with a single speaker (pldamodule.cpp:83-86) */
#define PLDA_E_NUMERIC (-3)   /* not positive definite / eigensolver did not converge */
#define PLDA_E_NOT_FITTED (-4)
#define PLDA_E_HIP (-5)       /* HIP runtime failure (message has the hipError string) */
#define PLDA_E_LABELS (-6)    /* fit labels not dense 0..K-1 (pldamodule.cpp:88-92 indexes by value) */
#define PLDA_E_CAPACITY (-7)  /* caller output too small */

typedef struct plda_handle plda_handle;

/* ---- lifecycle: replaces Plda_new / MPLDA_dealloc (pldamodule.cpp:297-316) and
 *      the module init initlibplda (pldamodule.cpp:371-384) ---- */
int plda_create(int device, plda_handle **out);
int plda_destroy(plda_handle *h);
const char *plda_last_error(const plda_handle *h);
int plda_abi_version(void);
/* bit 0: built with -DPLDA_DIAG=1 (libplda_hip_diag.so: + the measurement arms of the trials GEMM, some of which return
 * garbage scores; selected by PLDA_GEMM_VARIANT).  The product library returns 0 and plda_create refuses those variants. */
int plda_build_flags(void);
/* enqueue on exactly this hipStream_t (e.g. torch's current stream; NULL is HIP's default
 * stream, with its implicit-synchronisation rules); plda_reset_stream goes back to the
 * handle's own non-blocking stream */
int plda_set_stream(plda_handle *h, void *hip_stream);
int plda_reset_stream(plda_handle *h);
int plda_synchronize(plda_handle *h);

/* ---- fit: replaces MPlda_fit (pldamodule.cpp:42-109) ----
 * labels must be dense 0..K-1 (the Python shim compacts arbitrary unsigned
 * labels first).  Runs: label counting-sort, per-speaker centroids,
 * AddSamples(1/n_k) scatter (pldamodule.cpp:94-98), `iters` EM iterations
 * (Kaldi PldaEstimator::Estimate, :102-106) and GetOutput, all on the GPU in fp64. */
int plda_fit(plda_handle *h, const double *X, int64_t N, int32_t D /* <= 2048 */,
             const uint64_t *labels, int32_t iters);
int plda_fit_dev(plda_handle *h, const double *dX, int64_t N, int32_t D,
                 const uint64_t *dlabels, int64_t K, int32_t iters);
/* The same fit in its two halves, for fitting on several GPUs: every
 * quantity AddSamples accumulates (pldamodule.cpp:94-98) is a sum over speakers, so each rank
 * runs the statistics pass on the rows of ITS speakers (local dense labels 0..K-1),
 *   plda_fit_stats_dev      -> means[K,D], counts[K], offset scatter[D,D] of those speakers,
 *   plda_fit_get_stats_dev  copies them to caller-owned DEVICE buffers (any may be NULL),
 * the ranks all-reduce the scatter and all-gather means/counts, and every rank then runs
 *   plda_fit_em_dev         EM + GetOutput (pldamodule.cpp:102-106) on the merged statistics.
 * plda_fit_dev == plda_fit_stats_dev followed by plda_fit_em_dev on the handle's own buffers. */
int plda_fit_stats_dev(plda_handle *h, const double *dX, int64_t N, int32_t D,
                       const uint64_t *dlabels, int64_t K);
int plda_fit_get_stats_dev(plda_handle *h, double *dmeans, int64_t *dcounts, double *dscatter);
int plda_fit_em_dev(plda_handle *h, const double *dmeans, const int64_t *dcounts, int64_t K,
                    const double *dscatter, int32_t D, int32_t iters);
/* timings of the last fit, milliseconds: [0] statistics pass (sort+centroid+scatter): host wall clock of
 * plda_fit_stats*, its span on the stream inside plda_fit (which does not synchronise between the two halves);
 * [1] EM loop (all iterations): its span on the stream between two events (planning of the count groups
 * on the host included); [2] GetOutput: the rest of the call's wall clock behind the EM (its kernels, the
 * export of the model and status words to the host mirror, the one synchronisation); [3] = iterations run.
 * [1] + [2] = wall clock of plda_fit_em_dev; [0] + [1] + [2] = wall clock of plda_fit. */
int plda_fit_timings(plda_handle *h, double out_ms[4]);
/* how the EM of the last fit ran (pldamodule.cpp:102-105 is one loop over the classes; here the classes are grouped by
 * their utterance count, the reason the reference sorts them, pldamodule.cpp:94-100): out[0] = number of distinct counts G
 * (0: not grouped), out[1] = 0 EM in the simultaneously-diagonalised basis, 1 grouped on per-group second moments
 * (few groups of many classes), 2 grouped on the class means (many groups). */
int plda_fit_plan(plda_handle *h, int32_t out[2]);
/* staged access to the fit internals (parity tests of SURVEY.md rows a3-a7):
 * any pointer may be NULL.  means[K*D] in label order, counts[K], scatter[D*D],
 * sum[D], W[D*D], B[D*D] (final within/between covariances before GetOutput). */
int plda_fit_get_stats(plda_handle *h, double *means, int64_t *counts, double *scatter,
                       double *sum, double *W, double *B);
int plda_fit_num_classes(plda_handle *h, int64_t *K);

/* ---- model state: the Kaldi `Plda` held at pldamodule.cpp:29 ----
 * transform is [Dout, Din] row-major; after fit Dout == Din == D. */
int plda_get_dims(plda_handle *h, int32_t *Dout, int32_t *Din);
int plda_get_model(plda_handle *h, double *mean /*[Din]*/, double *transform /*[Dout*Din]*/,
                   double *psi /*[Dout]*/, double *offset /*[Dout]*/);
int plda_set_model(plda_handle *h, int32_t Dout, int32_t Din, const double *mean,
                   const double *transform, const double *psi);
/* build extension ("targetdim"): keep the first
 * `targetdim` rows of transform / psi (largest between-class variance). */
int plda_truncate(plda_handle *h, int32_t targetdim);
/* replaces Plda::SmoothWithinClassCovariance reached at pldamodule.cpp:158-160 */
int plda_smooth(plda_handle *h, double factor);

/* ---- transform: replaces Mplda_transform (pldamodule.cpp:111-194) ----
 * groups rows by label (any u64 values), averages, applies
 * Plda::TransformIvector(mean, n) (:171) incl. length normalisation.  Outputs
 * ascending by label (std::map order, :164).  *Ku: in = capacity (rows of the
 * out arrays), out = number of groups. */
int plda_transform_groups(plda_handle *h, const double *X, int64_t N, int32_t Din,
                          const uint64_t *labels, uint64_t *out_labels,
                          int64_t *out_counts, double *out_vecs /*[Ku*Dout]*/,
                          int64_t *Ku);
/* the same on HBM-resident rows and labels; outputs (device): out_labels[Ku] u64, out_counts[Ku] int32,
 * out_vecs[Ku, Dout].  Grouping is a device radix sort over as many 8-bit digits as the largest label has. */
int plda_transform_groups_dev(plda_handle *h, const double *dX, int64_t N, int32_t Din,
                              const uint64_t *dlabels, uint64_t *dout_labels, int32_t *dout_counts,
                              double *dout_vecs, int64_t *Ku);
/* batched Plda::TransformIvector on R already-averaged rows; num_examples[R]
 * (int32) or, if NULL, `n_uniform` for every row. */
int plda_transform_rows(plda_handle *h, const double *Xbar, int64_t R, int32_t Din,
                        const int32_t *num_examples, int32_t n_uniform, double *out);
int plda_transform_rows_dev(plda_handle *h, const double *dXbar, int64_t R, int32_t Din,
                            const int32_t *dnum_examples, int32_t n_uniform, double *dout);

/* ---- score: replaces MPlda_score (pldamodule.cpp:258-277) ----
 * Trial list: P pairs (enrol row e_idx[p] of U, test row t_idx[p] of V), each
 * Plda::LogLikelihoodRatio(U[e], n[e], V[t]) (:266) in fp64, then the optional
 * z-norm (s - zmean[e]) / zstd[e] (:269-273) where zmean/zstd are non-NULL and
 * zstd[e] != 0.  plda.score() is the P == 1 case.
 * Lists of >= 16 384 trials (a trials file: scoring/scorePLDA.py:299-321) run on per-count tables -- the terms of the LLR that
 * depend on (n[e], dimension) only are tabulated per distinct count -- in fp64, equal to the per-element form to 1e-11; their
 * indices are checked on the device.  An index outside [0, M) x [0, Nt) is PLDA_E_INVAL naming the first such trial. */
int plda_score_pairs(plda_handle *h, const double *U, const int32_t *n_enrol, int64_t M,
                     const double *V, int64_t Nt, const int64_t *e_idx,
                     const int64_t *t_idx, int64_t P, const double *zmean,
                     const double *zstd, double *out);
/* ONE trial, evaluated on the host: plda.score() (pldamodule.cpp:258-277) is one LogLikelihoodRatio (:266) per
 * Python call, i.e. 2 D numbers and ~5 D flop -- a GPU launch + synchronisation costs ten times that, so the scalar
 * call is served from the handle's host mirror of psi (fp64, per-count terms cached; the library's own code, no
 * oracle).  u, v: already-transformed vectors [Dout]; has_z != 0 applies (s - zmean) / zstd where zstd != 0 (:269-273).
 * Batches belong on plda_score_pairs / plda_score_matrix. */
int plda_score_one(plda_handle *h, const double *u, int32_t n_enrol, const double *v, int32_t has_z,
                   double zmean, double zstd, double *out);
/* Dense trials matrix out[i*ld_out + j] = LLR(U[i], n[i], V[j]) for the M x Nt
 * block (the nested Python loop of scoring/scorePLDA.py:302-318 and
 * tests/pldatest.py:29-33 as one launch): fp64 bias terms + fp32 MFMA GEMM,
 * fp32 scores.  n_enrol NULL => every row uses n_uniform (GEMM depth Dout).
 * Otherwise the rows are bucketed by their G DISTINCT counts and the depth is
 * Dout + G - 1 (one column-bias vector per distinct count, carried as extra
 * contraction columns; C4's n in 1..5 at D = 256: 264 instead of 512); counts
 * above 4095, more than 64 distinct ones or G - 1 > Dout / 2 take the depth-2*Dout
 * form [A1 | A2] x [V | V*V].  plda_score_matrix finds the distinct counts on the
 * host; the _dev entry points find them with one small device pass and ONE wait
 * for the handle's stream per call (the host must know G to size the operands);
 * uniform calls and calls that reuse a prepared test side of the depth-2*Dout
 * form enqueue only.  zmean/zstd as above (nullable). */
int plda_score_matrix(plda_handle *h, const double *U, const int32_t *n_enrol,
                      int32_t n_uniform, int64_t M, const double *V, int64_t Nt,
                      const double *zmean, const double *zstd, float *out,
                      int64_t ld_out);
int plda_score_matrix_dev(plda_handle *h, const double *dU, const int32_t *dn_enrol,
                          int32_t n_uniform, int64_t M, const double *dV, int64_t Nt,
                          const double *dzmean, const double *dzstd, float *dout,
                          int64_t ld_out);
/* One test set against many enrol sets (the reference's callers score every enrol model against the same test
 * utterances, scoring/scorePLDA.py:302-318): plda_score_prepare_dev packs the test side once -- fp64 -> k-quad packed
 * fp32 (+ V*V when enrol counts differ: mixed_counts != 0; else the column biases for n_uniform) -- and later
 * plda_score_matrix_dev / _sharded_dev calls with the SAME dV, Nt, model and kind of enrol counts skip that work (C3:
 * 2.1 of 72 ms per call).  The cache is keyed on (dV, Nt, model epoch, kind of counts) AND guarded by a content
 * fingerprint of 64 rows spread over the set (first and last included), taken at prepare time: a reusing call recomputes
 * it (one small kernel + a stream synchronisation, ~30 us; once per call, not per block of a sharded call) and, when the
 * rows behind the pointer have changed -- an in-place update, or an allocator that handed the address to another tensor
 * -- treats the cache as MISSED: the rows are packed again and the call succeeds (round 5; rounds 3-4 returned
 * PLDA_E_INVAL here, which failed legitimate calls whose new tensor sat at a recycled address).  Rows outside the sample
 * are still the caller's promise.  A different test side, a model change (fit, set_model, truncate, smooth) or
 * plda_score_unprepare end the reuse silently.  Test sides whose packed form would reach 4 GiB are refused (such calls
 * are scored in column blocks, each packed per call).
 * mixed_counts != 0 prepares the depth-2*Dout form, which later mixed-count calls then USE (no count pass, no wait);
 * plda_score_prepare_counts_dev prepares the faster bucketed form for the distinct enrol counts the later calls will
 * bring (host array `counts`, any order, duplicates allowed): calls whose counts are a subset reuse it, others repack. */
int plda_score_prepare_dev(plda_handle *h, const double *dV, int64_t Nt, int32_t mixed_counts, int32_t n_uniform);
int plda_score_prepare_counts_dev(plda_handle *h, const double *dV, int64_t Nt, const int32_t *counts, int32_t num_counts);
int plda_score_unprepare(plda_handle *h);
/* Kernel timing for roofline accounting: when enabled, every trials-GEMM launch is
 * bracketed by HIP events recorded on the stream it is launched on; plda_profile_read
 * synchronises and returns the accumulated GEMM milliseconds, launches, and the
 * algorithmic flop (2 * gemm_k per trial) since the last reset. */
int plda_profile_enable(plda_handle *h, int32_t on);
int plda_profile_read(plda_handle *h, double *gemm_ms, int64_t *launches, double *gemm_flop,
                      int32_t reset);
/* Diagnostic (PLDA_HIP tracing knob): with PLDA_GEMM_VARIANT=31 in the environment at
 * plda_create, the trials GEMM runs an instrumented instantiation whose workgroup 0 stamps the
 * shader clock per wave at every stage barrier (arrive, leave) and around every tile epilogue.
 * out[tile < 8][stage < 16][wave < 8][8] = {barrier arrive, leave, start of steps 1..3 of the stage (0 if
 * absent), -, epilogue start, epilogue end (the last two in stage slot 15)}. */
int plda_profile_timeline(plda_handle *h, uint64_t *out, int64_t cap_words);

/* Per-stage timing (PLDA_HIP_TRACE): named spans bracketed by HIP events on the stream around
 * the stages of fit (label sort, centroids K1, scatter SYRK K2, EM, GetOutput: whitening / tridiagonalisation /
 * divide and conquer / back-transformation), transform and scoring.  plda_trace_read synchronises the stream and
 * writes a JSON array [{"name", "calls", "ms", "work", "unit"}] aggregated by name ("work" = algorithmic flop or
 * bytes of the stage where one is defined) into json[cap]; PLDA_E_CAPACITY if it does not fit.  With the
 * environment variable PLDA_HIP_TRACE=1 tracing is on from plda_create and the summary is printed to stderr by
 * plda_destroy.  The reference has no counterpart (its stages are Kaldi calls inside pldamodule.cpp:76-106). */
int plda_trace_enable(plda_handle *h, int32_t on);
int plda_trace_read(plda_handle *h, char *json, int64_t cap, int32_t reset);

/* The symmetric eigensolver of GetOutput on its own (diagnostics and tests; the reference reaches it only through
 * Plda estimation, pldamodule.cpp:102-106 -> Kaldi SpMatrix::Eig).  G [D,D] row-major symmetric, host pointers.
 * eigenvalues[D] descending (signed), eigenvectors [D,D] with eigenvector i in ROW i.  method: 0 = what fit uses
 * (tridiagonalisation + divide and conquer where supported, else block Jacobi), 1 = block Jacobi, 2 = direct
 * method or PLDA_E_NUMERIC.  *method_used (nullable) reports 1 or 2. */
int plda_sym_eig(plda_handle *h, const double *G, int32_t D, int32_t method, double *eigenvalues,
                 double *eigenvectors, int32_t *method_used);
/* The fp64 GEMM behind fit and GetOutput on its own (diagnostics and tests; the reference reaches its counterpart,
 * ATLAS dgemm, only through Kaldi inside pldamodule.cpp:76-106).  `batch` products C_b = alpha op(A_b) op(B_b) + beta C_b,
 * host pointers, row-major: A_b is [M,K] (transA = 0) or [K,M] (transA = 1), B_b is [K,N] or [N,K], C_b [M,N]; operands
 * are stored back to back.  kw (nullable, batch = 1 only): K weights folded into the contraction, sum_k w_k a_mk b_kn.
 * The dispatch is the product's: one 16 x 16 tile per workgroup for M, N, K <= 256, the panel kernel for deeper
 * small products, 64 x 64 / 128 x 128 tiles with split-K above. */
int plda_gemm_f64(plda_handle *h, int64_t M, int64_t N, int64_t K, double alpha, const double *A, int32_t transA,
                  const double *B, int32_t transB, const double *kw, double beta, double *C, int32_t batch);

/* The SPD inverse of the EM's E-step on its own (diagnostics and tests; the reference reaches it only through
 * PldaEstimator::GetStatsFromClassMeans, Kaldi ivector/plda.cc:436-447 -> SpMatrix::Invert).  A [D,D] row-major
 * symmetric positive definite, host pointers; inverse [D,D].  D <= 64: scalar sweep operator in registers,
 * D <= 256: block sweeps on the fp64 matrix cores, above: blocked whitening.  PLDA_E_NUMERIC when a pivot is not
 * positive. */
int plda_spd_inverse(plda_handle *h, const double *A, int32_t D, double *inverse);
/* algorithmic work of the last score_matrix call: flop of the trials GEMM and
 * its depth, for roofline accounting */
int plda_score_last_shape(plda_handle *h, int64_t *M, int64_t *Nt, int32_t *gemm_k);
/* name of the trials-GEMM kernel the last plda_score_matrix* call launched (roofline accounting: bench.py names the
 * kernel its `roofline` block prices); "" before the first call.  PLDA_E_CAPACITY when it does not fit name[cap]. */
int plda_score_last_kernel(plda_handle *h, char *name, int64_t cap);

/* ---- z-norm: replaces MPlda_norm (pldamodule.cpp:196-256) ----
 * Every cohort row is transformed with num_examples = Nb (:224) and scored as
 * the TRAIN side with n = 1 against every model vector (:235); per model the
 * mean and population std over the cohort (:240-250).  Fused: the Nb x M score
 * matrix is never materialised.  models are already-transformed vectors. */
/* num_examples: the count the reference hands to TransformIvector at :224, i.e. the
 * row count of the FULL background matrix; differs from Nb only when the caller
 * passes a row subset (numutts > 0, :204-216).  0 means Nb. */
int plda_znorm_stats(plda_handle *h, const double *bkg, int64_t Nb, int32_t num_examples,
                     int32_t Din, const double *models, int64_t M, double *out_mean,
                     double *out_std);
int plda_znorm_stats_dev(plda_handle *h, const double *dbkg, int64_t Nb, int32_t num_examples,
                         int32_t Din, const double *dmodels, int64_t M, double *dout_mean,
                         double *dout_std);

/* ---- d-vector front-end (the step before the path): replaces
 * scoring/extractdvector.py:19-58 -- per-frame L2 normalisation (getnormalizedvector,
 * :19-29; skipped when l2norm == 0, the *_nol2 variants :50-59) and pooling over each
 * utterance's frames: method 0 = mean (:37-39), 1 = max (:32-34), 2 = population
 * variance (:42-47).  frames [T, D] row-major, dtype 0 = float32 / 1 = float64;
 * offsets[U+1] frame boundaries of the U utterances; out [U, D] fp64. ---- */
int plda_dvector_pool(plda_handle *h, const void *frames, int32_t dtype, int64_t T, int32_t D,
                      const int64_t *offsets, int64_t U, int32_t method, int32_t l2norm,
                      double *out);
int plda_dvector_pool_dev(plda_handle *h, const void *dframes, int32_t dtype, int64_t T, int32_t D,
                          const int64_t *doffsets, int64_t U, int32_t method, int32_t l2norm,
                          double *dout);

/* ---- HTK feature files (the data format in front of the path): replaces the reference's reader
 * chtk::htk_load (chtk/chtk.cpp:38-88, called at src/kaldi-utils.hpp:22; header layout
 * chtk/chtk.h:52-57).  A call decodes a BATCH of U files whose DATA sections (everything after
 * the 12-byte header) sit in one blob: file u starts at 32-bit word file_off[u] of the blob and
 * has frame_off[u+1] - frame_off[u] frames of `samplesize` bytes (a multiple of 4); a file
 * shorter than its header claims must be zero-padded to that size, which is what the
 * reference's zero-initialised read buffer yields (chtk.cpp:56-57).  out [T, (2 frm_ext + 1) *
 * samplesize / 4] float32, T = frame_off[U]: every word byte-swapped (big-endian floats), frame
 * i = frames clamp(i - frm_ext .. i + frm_ext, 0, n - 1) of its file concatenated (:71-86).
 * Bit-exact with the reference; the output feeds plda_dvector_pool_dev with the same
 * frame_off. ---- */
int plda_htk_frames(plda_handle *h, const void *blob, int64_t blob_bytes, const int64_t *file_off,
                    const int64_t *frame_off, int64_t U, int32_t samplesize, int32_t frm_ext,
                    float *out);
int plda_htk_frames_dev(plda_handle *h, const void *dblob, const int64_t *dfile_off,
                        const int64_t *dframe_off, int64_t U, int64_t T, int32_t samplesize,
                        int32_t frm_ext, float *dout);

/* ---- equal error rate (the step after the path): replaces scoring/eer.py:68-73
 * (bob.measure.eer_threshold + farfrr; bob is absent and un-pinned, its published
 * definition is restated: FAR = #{impostor >= t}/Nn, FRR = #{target < t}/Np, t = the
 * midpoint after the score where |FAR - FRR| is minimal, later candidate on ties).
 * out[6] = threshold, FAR, FRR, EER = (FAR + FRR)/2, #targets, #impostors.
 * _matrix: trial (i, j) of the fp32 score matrix is a target iff enrol_spk[i] == test_spk[j];
 * _lists: separate target / impostor score arrays (the two files eer.py reads). ---- */
int plda_eer_matrix_dev(plda_handle *h, const float *dscores, int64_t ld, int64_t M, int64_t Nt,
                        const int64_t *denrol_spk, const int64_t *dtest_spk, double *out);
int plda_eer_lists(plda_handle *h, const float *pos, int64_t np, const float *neg, int64_t nn,
                   double *out);
/* The points of the DET curve scoring/eer.py:34-62 plots (bob.measure.plot.det(negatives, positives, 100); definition
 * restated, bob absent): n_points (2 .. 2047) thresholds spread evenly from the smallest to the largest score, accumulated in
 * float64; far[i] = #{impostor >= t_i} / Nn, frr[i] = #{target < t_i} / Np (HOST arrays; thresholds nullable).  The plot's axes
 * are the normal deviates of the two rates (plda_amd.eer.ppndf); drawing it stays with the caller. */
int plda_det_matrix_dev(plda_handle *h, const float *dscores, int64_t ld, int64_t M, int64_t Nt,
                        const int64_t *denrol_spk, const int64_t *dtest_spk, int32_t n_points, double *far, double *frr,
                        double *thresholds);
int plda_det_lists(plda_handle *h, const float *pos, int64_t np, const float *neg, int64_t nn, int32_t n_points,
                   double *far, double *frr, double *thresholds);
/* The EER of the trials between enrol models and test vectors WITHOUT the matrix (round 5): what the reference's caller
 * wants from its M x Nt calls of MPlda_score (scoring/scorePLDA.py:302-318 -> scoring/eer.py:68-76) is these six numbers,
 * not the scores -- BASELINE C4's matrix is 192 GB.  Arguments as plda_score_matrix_dev (transformed vectors, per-model
 * counts or one count, optional z-norm statistics) plus the speaker ids of plda_eer_matrix_dev; the scores exist one row
 * slab of <= 4 GiB at a time (scored by the trials GEMM, consumed by the one-pass EER, dropped).  The result is identical
 * to plda_score_matrix_dev + plda_eer_matrix_dev.  out is a HOST array; the call synchronises the handle's stream. */
int plda_score_eer_dev(plda_handle *h, const double *dU, const int32_t *dn_enrol, int32_t n_uniform, int64_t M,
                       const double *dV, int64_t Nt, const double *dzmean, const double *dzstd,
                       const int64_t *denrol_spk, const int64_t *dtest_spk, double *out);
/* Row-sharded trials matrix (one process per GPU, each holding a slab of enrol rows and ALL test
 * labels): the same three histogram passes over the local slab; after each pass the library calls
 * `reduce(ctx, hist, NULL, NULL)` with hist[2 * 2048] host counters to be SUMMED over the ranks in
 * place, and once at the end `reduce(ctx, NULL, below, above)` with two host words to be replaced
 * by their MAX resp. MIN over the ranks.  The callback returns 0 on success.  Every rank gets the
 * global result; scores never leave their GPU (the exchange is 3 x 32 KiB + 8 bytes).
 * plda_amd/sharding.py:eer_sharded supplies a torch.distributed callback. */
typedef int (*plda_eer_reduce_fn)(void *ctx, unsigned long long *hist, unsigned *below, unsigned *above);
int plda_eer_matrix_sharded_dev(plda_handle *h, const float *dscores, int64_t ld, int64_t M, int64_t Nt,
                                const int64_t *denrol_spk, const int64_t *dtest_spk,
                                plda_eer_reduce_fn reduce, void *ctx, double *out);

/* ---- several GPUs of one node: one process per GPU, one handle per process.  The reference
 * has no counterpart (one process, one thread; its native object libplda.MPlda, pldamodule.cpp:280-295, is the only
 * thing callers bind, so the sharded path lives behind the same object).
 *
 * Every collective the library issues goes through ONE table of three operations on DEVICE pointers
 * (plda_collectives); three providers fill it:
 *   plda_comm_init         RCCL over xGMI (the default and the production transport).  librccl is opened lazily by
 *                          this call -- a single-GPU user never needs it.  Rank 0 calls plda_comm_unique_id and
 *                          distributes the 128 bytes by any means (MPI, a file, torch.distributed's store); every rank
 *                          then calls plda_comm_init (collective).
 *   plda_comm_init_host    any HOST transport (MPI, gloo, shared memory): the caller supplies the two operations on
 *                          host buffers, the library stages device data through a pinned bounce buffer in bounded
 *                          chunks.  This is also how the sharded entry points are tested with several processes on
 *                          ONE GPU (RCCL refuses two ranks on one device): tests/test_gpu_comm_procs.py.
 *   plda_comm_init_peer    direct writes over xGMI (round 4): every rank opens the others' buffers through HIP IPC and
 *                          PUSHES its piece into them with device-to-device copies, one copy stream per peer -- all 7
 *                          links of a GPU busy at once (~1.07 TB/s) where a ring moves one link's ~153 GB/s (SURVEY.md
 *                          section 5).  `bootstrap` = the host operations of plda_comm_init_host; only its
 *                          all_gather_v is used: it carries the IPC handles (72 bytes per rank and call) and is the
 *                          rendezvous the inter-process events need -- no device data crosses the host.  Works between
 *                          processes on one device too.  "transport": "peer" in plda_comm_describe.
 *   plda_comm_init_custom  the device-level table itself.
 * All callbacks return 0 on success.  Buffers are byte-addressed; `hip_stream` is the hipStream_t the operation must
 * be ordered on (enqueue, or synchronise it and work on the host).
 *
 *   plda_shard_plan                 the row partition of the trials matrix, as a pure function (no handle, no GPU):
 *       block b of `block_rows` rows (rounded up to a multiple of 256; <= 0: 4096) belongs to rank b mod R; the rows
 *       left over after the last full round of R blocks are dealt out once more in R equal smaller blocks, so the
 *       tail does not land on rank 0.  Writes this rank's blocks (first row, row count) in ascending order; a rank's
 *       COMPACT slab holds them back to back, *local_rows rows in all.
 *   plda_score_matrix_sharded_dev   every rank passes the SAME replicated inputs (all M enrol rows, all Nt tests,
 *       a replicated model) and scores the blocks plda_shard_plan gives it, straight into their final rows of the
 *       full matrix dout[M, ld_out] (which must hold M * ld_out floats).  gather == 0: that is all (scores stay
 *       sharded: what thresholding, counting, EER want; no collective).  gather != 0: every R consecutive blocks are
 *       assembled on every rank by one IN-PLACE all-gather on a side stream while the next blocks are being
 *       scored (no staging copy; ragged tail: all_gather_v); whole ld_out-wide rows travel, so the padding columns
 *       [Nt, ld_out) of dout are overwritten with unspecified values; the handle's stream is ordered behind the
 *       last one.
 *   plda_score_matrix_sharded_local_dev   the same partition with COMPACT output: this rank's blocks back to back in
 *       dlocal[local_rows, ld_local] -- a rank holds M/R rows of scores, not M (C4 on 8 GPUs: 24 GB instead of
 *       192 GB).  dfull == NULL: scores stay sharded.  dfull != NULL (needs ld_full == ld_local): the full
 *       [M, ld_full] matrix is assembled there on every rank as well, super-block by super-block, overlapped as above.
 *   plda_znorm_stats_sharded_dev    MPlda_norm (pldamodule.cpp:196-256) with the M models split contiguously
 *       over the ranks (every rank scans the whole cohort); full mean / std arrays on every rank.
 *   plda_fit_sharded_dev            MPlda_fit with the statistics pass (pldamodule.cpp:76-100) over THIS rank's
 *       speakers (local dense labels 0..K-1; a speaker's rows must all be on one rank); the D x D offset
 *       scatter is all-reduced, centroids and counts are all-gathered in rank order, and EM + GetOutput
 *       (:102-106) run as replicas on every rank from identical inputs.
 *   plda_eer_matrix_comm_dev        plda_eer_matrix_sharded_dev with the handle's collectives; the compact slab of
 *       plda_score_matrix_sharded_local_dev (with the speaker ids of ITS rows) is what it takes.
 * Without a communicator all of them run as a single rank. ---- */
#define PLDA_DT_F64 0
#define PLDA_DT_U64 1
#define PLDA_DT_U32 2
#define PLDA_OP_SUM 0
#define PLDA_OP_MAX 1
#define PLDA_OP_MIN 2
typedef struct plda_collectives {
  void *ctx;
  /* every rank contributes `bytes` at dsend; drecv receives nranks * bytes in rank order.  dsend may be
   * drecv + rank * bytes (in place). */
  int (*all_gather)(void *ctx, const void *dsend, void *drecv, int64_t bytes, void *hip_stream);
  /* ragged, in place: rank q owns bytes [offs[q], offs[q] + counts[q]) of the same dbuf on every rank (counts may be
   * 0); afterwards every rank holds every piece.  offs / counts: nranks host int64 each, identical on all ranks. */
  int (*all_gather_v)(void *ctx, void *dbuf, const int64_t *offs, const int64_t *counts, void *hip_stream);
  /* element-wise, in place: dtype PLDA_DT_*, op PLDA_OP_* */
  int (*all_reduce)(void *ctx, void *dbuf, int64_t count, int32_t dtype, int32_t op, void *hip_stream);
  /* called once by plda_comm_destroy / plda_destroy; may be NULL */
  void (*destroy)(void *ctx);
} plda_collectives;
typedef struct plda_host_collectives {
  void *ctx;
  /* the same two operations on HOST memory (the library's pinned bounce buffer), blocking */
  int (*all_gather_v)(void *ctx, void *hbuf, const int64_t *offs, const int64_t *counts);
  int (*all_reduce)(void *ctx, void *hbuf, int64_t count, int32_t dtype, int32_t op);
  void (*destroy)(void *ctx);
} plda_host_collectives;
int plda_comm_unique_id(void *out, int64_t cap_bytes /* >= 128 */);
int plda_comm_init(plda_handle *h, int32_t nranks, int32_t rank, const void *unique_id);
int plda_comm_init_custom(plda_handle *h, int32_t nranks, int32_t rank, const plda_collectives *table);
int plda_comm_init_host(plda_handle *h, int32_t nranks, int32_t rank, const plda_host_collectives *table);
int plda_comm_init_peer(plda_handle *h, int32_t nranks, int32_t rank, const plda_host_collectives *bootstrap);
int plda_comm_destroy(plda_handle *h);
int plda_comm_info(plda_handle *h, int32_t *nranks, int32_t *rank);
/* who takes part, as the TRANSPORT reports it, written as a JSON object into json[cap]: {"transport": "rccl" | "host"
 * | "peer" | "custom" | "none" | "emulated", "nranks", "rank", "device", "pci_bus_id"}; for RCCL nranks / rank / device come
 * from ncclCommCount / ncclCommUserRank / ncclCommCuDevice (plus "rccl_version"), not from what the caller passed in */
int plda_comm_describe(plda_handle *h, char *json, int64_t cap);
/* test hook: act as rank `rank` of `nranks` WITHOUT a communicator (no collective runs, gather is ignored):
 * lets one GPU play every rank in turn and check that the shards tile the whole problem */
int plda_comm_emulate(plda_handle *h, int32_t nranks, int32_t rank);
int plda_shard_plan(int64_t M, int32_t nranks, int32_t rank, int64_t block_rows, int64_t *row_start,
                    int64_t *row_count, int64_t cap, int64_t *nblocks, int64_t *local_rows);
int plda_score_matrix_sharded_dev(plda_handle *h, const double *dU, const int32_t *dn_enrol, int32_t n_uniform,
                                  int64_t M, const double *dV, int64_t Nt, const double *dzmean,
                                  const double *dzstd, float *dout, int64_t ld_out, int64_t block_rows,
                                  int32_t gather);
int plda_score_matrix_sharded_local_dev(plda_handle *h, const double *dU, const int32_t *dn_enrol, int32_t n_uniform,
                                        int64_t M, const double *dV, int64_t Nt, const double *dzmean,
                                        const double *dzstd, float *dlocal, int64_t ld_local, int64_t block_rows,
                                        float *dfull, int64_t ld_full);
int plda_znorm_stats_sharded_dev(plda_handle *h, const double *dbkg, int64_t Nb, int32_t num_examples,
                                 int32_t Din, const double *dmodels, int64_t M, double *dout_mean,
                                 double *dout_std);
int plda_fit_sharded_dev(plda_handle *h, const double *dX, int64_t N, int32_t D, const uint64_t *dlabels,
                         int64_t K, int32_t iters);
int plda_eer_matrix_comm_dev(plda_handle *h, const float *dscores, int64_t ld, int64_t M, int64_t Nt,
                             const int64_t *denrol_spk, const int64_t *dtest_spk, double *out);

/* ---- LDA: replaces the reference's second model, the pure-Python
 * class LDA of python/liblda/lda.py (used by scoring/scoreLDA.py:175,224,241), on the same
 * handle.  All fp64.  solver: 0 = 'svd' (lda.py:171-209), 1 = 'eigen' (:134-169),
 * 2 = 'lsqr' (:211-240).  labels dense 0..K-1 (the shim compacts np.unique order, :112-116);
 * priors: K host doubles or NULL = class frequencies (:113-119), renormalised when their sum
 * is not exactly 1 (:121-122).  A singular within-class covariance with the eigen solver
 * returns PLDA_E_NUMERIC (the reference raises LinAlgError from scipy.linalg.eigh, :157).
 *   plda_lda_dims        K, D, rank (columns of scalings: svd = retained rank, eigen = D,
 *                        lsqr = 0), solver
 *   plda_lda_get_model   priors[K], means[K,D], xbar[D] (svd), scalings[D,rank] (svd, eigen),
 *                        coef[K,D], intercept[K], explained_variance_ratio[D] (eigen); any NULL
 *   plda_lda_set_model   restores a saved model (the reference cannot persist one)
 *   plda_lda_predict     out[N,K]; mode 0 decision_function (:242-270), 1 predict_log_proba
 *                        (:296-314), 2 the logistic of the decision values (first half of
 *                        predict_proba, :283-287), 3 the same one-vs-rest normalised (:292)
 *   plda_lda_transform   out[N,ncomp] = X scalings[:, :ncomp] (eigen, :333-334) or
 *                        (X - xbar) scalings[:, :ncomp] (svd, :331-332); lsqr -> PLDA_E_INVAL ---- */
int plda_lda_fit(plda_handle *h, const double *X, int64_t N, int32_t D, const uint64_t *labels,
                 int32_t solver, const double *priors);
int plda_lda_fit_dev(plda_handle *h, const double *dX, int64_t N, int32_t D, const uint64_t *dlabels,
                     int64_t K, int32_t solver, const double *priors);
int plda_lda_dims(plda_handle *h, int64_t *K, int32_t *D, int32_t *rank, int32_t *solver);
int plda_lda_get_model(plda_handle *h, double *priors, double *means, double *xbar, double *scalings,
                       double *coef, double *intercept, double *evr);
int plda_lda_set_model(plda_handle *h, int32_t solver, int64_t K, int32_t D, int32_t rank,
                       const double *priors, const double *means, const double *xbar,
                       const double *scalings, const double *coef, const double *intercept);
int plda_lda_predict(plda_handle *h, const double *X, int64_t N, int32_t D, int32_t mode, double *out);
int plda_lda_predict_dev(plda_handle *h, const double *dX, int64_t N, int32_t mode, double *dout);
int plda_lda_transform(plda_handle *h, const double *X, int64_t N, int32_t D, int32_t ncomp, double *out);
int plda_lda_transform_dev(plda_handle *h, const double *dX, int64_t N, int32_t ncomp, double *dout);

#ifdef __cplusplus
}
#endif
#endif
