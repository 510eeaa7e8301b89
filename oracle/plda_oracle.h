/*
 * oracle/plda_oracle.h -- CPU restatement of the reference's PLDA hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under plda_amd/ or liblda/ may include,
 * link, import or execute this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / timed baseline.
 *
 * PARITY UNPINNED: the reference (RicherMans/PLDA) holds no PLDA arithmetic of
 * its own -- every numeric step is a call into Kaldi (kaldi-asr/kaldi,
 * src/ivector/plda.{h,cc}; un-vendored and un-pinned: CMakeLists.txt:41-48,
 * cmake/FindKaldi.cmake:14-17), which is absent from /root/reference and from
 * this image, and the reference's own tests (tests/pldatest.py:14,23-24,33)
 * pin no values.  This file restates Kaldi's published algorithm (SURVEY.md
 * Appendix A) as driven by the reference's call sites in src/pldamodule.cpp.
 * It is guarded by algebraic invariants, closed-form known-answer tests and an
 * independent NumPy restatement (oracle/plda_oracle_np.py); see tests/.
 *
 * All arithmetic is fp64, single-threaded, structured like Kaldi's per-class /
 * per-trial loops so that it doubles as the faithful timed CPU baseline.
 */
#ifndef PLDA_ORACLE_H_
#define PLDA_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- dense helpers (Kaldi matrix-library calls reached from plda.cc) ---- */

/* In-place Cholesky A = C C^T of a row-major symmetric D x D matrix; on return
 * the lower triangle (incl. diagonal) holds C and the strict upper triangle is
 * zero.  Returns 0, or -1 if A is not positive definite.
 * (Kaldi TpMatrix::Cholesky, reached from ComputeNormalizingTransform.) */
int plda_oracle_cholesky(double *A, int D);

/* In-place inverse of a lower-triangular row-major matrix (TpMatrix::Invert). */
int plda_oracle_tri_invert(double *L, int D);

/* In-place inverse of a general row-major D x D matrix by LU with partial
 * pivoting (the path SpMatrix::Invert takes under HAVE_ATLAS). 0 / -1. */
int plda_oracle_invert(double *A, int D);

/* Symmetric eigendecomposition A = U diag(s) U^T by Householder
 * tridiagonalisation + implicit QL (the algorithm family of Kaldi's
 * SpMatrix::Eig).  A is row-major, destroyed.  U is row-major with eigenvectors
 * in COLUMNS.  Eigenvalues unsorted-ascending as produced.  0 / -1. */
int plda_oracle_sym_eig(double *A, int D, double *s, double *U);

/* ---- Kaldi Plda model ops ---- */

/* Plda::TransformIvector (called at pldamodule.cpp:171,224).  Returns the
 * normalisation factor.  normalize_length / simple_length_norm are the
 * PldaConfig fields (defaults true / false; pldamodule.cpp:31 never changes
 * them). */
double plda_oracle_transform_ivector(const double *transform, const double *offset,
                                     const double *psi, int D, const double *x,
                                     int num_examples, int normalize_length,
                                     int simple_length_norm, double *out);

/* Plda::LogLikelihoodRatio(train, n, test) (called at pldamodule.cpp:235,266). */
double plda_oracle_llr(const double *psi, int D, const double *train, int n,
                       const double *test);

/* Plda::SmoothWithinClassCovariance (called at pldamodule.cpp:159).
 * Mutates transform / psi / offset in place. */
void plda_oracle_smooth(double *transform, double *psi, double *offset,
                        const double *mean, int D, double factor);

/* ---- wrapper-level entry points (src/pldamodule.cpp) ---- */

/* MPlda_fit (pldamodule.cpp:42-109): labels must be dense 0..K-1 (the
 * reference indexes a VLA by label value, :88-92).  Per speaker
 * AddSamples(1/n_k, rows) (:94-98), Sort (:100), Estimate(iters) (:102-106).
 * Outputs (caller-allocated): mean[D], transform[D*D] row-major, psi[D],
 * offset[D]; W_out/B_out (nullable, [D*D]) receive the final within/between
 * covariances before GetOutput.  Returns 0; -1 bad args; -2 single speaker
 * (:83-86); -3 numerical failure. */
int plda_oracle_fit(const double *X, int64_t N, int D, const uint64_t *labels,
                    int iters, double *mean, double *transform, double *psi,
                    double *offset, double *W_out, double *B_out);

/* The statistics half of fit only (PldaStats after the AddSamples loop):
 * means[K*D] in label order, counts[K], scatter[D*D], sum[D]. */
int plda_oracle_stats(const double *X, int64_t N, int D, const uint64_t *labels,
                      int64_t K, double *means, int64_t *counts, double *scatter,
                      double *sum, double *class_weight, double *example_weight);

/* One EstimateOneIter() on explicit inputs, Kaldi's per-class
 * loop with explicit inversions.  means must be sorted by counts ascending for
 * the "n changed" logic to match Kaldi, but any order gives the same maths.
 * W,B are updated in place. */
int plda_oracle_em_iter(const double *means, const int64_t *counts, int64_t K,
                        int D, const double *scatter, const double *sum,
                        double class_weight, double example_weight,
                        double *W, double *B);

/* PldaEstimator::GetOutput. */
int plda_oracle_get_output(const double *W, const double *B, const double *sum,
                           double class_weight, int D, double *mean,
                           double *transform, double *psi, double *offset);

/* Mplda_transform (pldamodule.cpp:111-194): per-label sum/count, mean,
 * TransformIvector(mean, n).  Labels are arbitrary u64; outputs ascending by
 * label.  *Ku in = capacity of the out arrays, out = number of groups.
 * smoothfactor == 1.0 means "skip" (:158-160); otherwise the model arrays are
 * mutated exactly as the reference does. */
int plda_oracle_transform_groups(double *transform, double *offset, double *psi,
                                 const double *mean, int D, const double *X,
                                 int64_t N, const uint64_t *labels,
                                 double smoothfactor, uint64_t *out_labels,
                                 int64_t *out_counts, double *out_vecs,
                                 int64_t *Ku);

/* MPlda_norm (pldamodule.cpp:196-256) with numutts == 0 (all rows, so the
 * unseeded shuffle only permutes the summation order): every bkg row is
 * transformed with num_examples = Nb (:224, quirk Q6), scored as the TRAIN
 * side with n = 1 against every model vector (:235, quirk Q7); per model the
 * mean and population std (two-pass, :240-250). */
int plda_oracle_norm(const double *transform, const double *offset,
                     const double *psi, int D, const double *bkg, int64_t Nb,
                     const double *models, int64_t M, double *out_mean,
                     double *out_std);

/* MPlda_score (pldamodule.cpp:258-277) driven over a dense M x Nt block the
 * way callers drive it (scoring/scorePLDA.py:302-318): one LLR per pair,
 * including the wrapper's per-call copies of both vectors (:264-265).
 * n_enrol[M]; zmean/zstd nullable (z-norm :269-273).  out[M*Nt] fp64. */
void plda_oracle_score_block(const double *psi, int D, const double *U,
                             const int32_t *n_enrol, int64_t M, const double *V,
                             int64_t Nt, const double *zmean, const double *zstd,
                             double *out);

/* EM objective -- test invariant only (non-decreasing). */
double plda_oracle_objective(const double *means, const int64_t *counts,
                             int64_t K, int D, const double *scatter,
                             const double *sum, double class_weight,
                             double example_weight, const double *W,
                             const double *B);

#ifdef __cplusplus
}
#endif
#endif
