/*
 * oracle/plda_oracle.c -- see plda_oracle.h.  TEST INFRASTRUCTURE ONLY;
 * PARITY UNPINNED (Kaldi absent; reference tests pin no values).
 *
 * Each function cites the reference call site (src/pldamodule.cpp:line) it
 * serves and the Kaldi routine (upstream src/ivector/plda.cc, restated in
 * SURVEY.md Appendix A) whose published algorithm it follows.
 */
#include "plda_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IDX(i, j, D) ((size_t)(i) * (size_t)(D) + (size_t)(j))

static double *dalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }

/* ------------------------------------------------------------------ */
/* dense helpers                                                       */
/* ------------------------------------------------------------------ */

int plda_oracle_cholesky(double *A, int D) {
  for (int j = 0; j < D; j++) {
    double d = A[IDX(j, j, D)];
    for (int k = 0; k < j; k++) d -= A[IDX(j, k, D)] * A[IDX(j, k, D)];
    if (!(d > 0.0)) return -1;
    d = sqrt(d);
    A[IDX(j, j, D)] = d;
    for (int i = j + 1; i < D; i++) {
      double s = A[IDX(i, j, D)];
      for (int k = 0; k < j; k++) s -= A[IDX(i, k, D)] * A[IDX(j, k, D)];
      A[IDX(i, j, D)] = s / d;
    }
  }
  for (int i = 0; i < D; i++)
    for (int j = i + 1; j < D; j++) A[IDX(i, j, D)] = 0.0;
  return 0;
}

int plda_oracle_tri_invert(double *L, int D) {
  /* column-by-column forward substitution: X = L^{-1}, lower triangular */
  for (int j = 0; j < D; j++) {
    if (L[IDX(j, j, D)] == 0.0) return -1;
  }
  double *X = dalloc((size_t)D * D);
  if (!X) return -1;
  for (int j = 0; j < D; j++) {
    X[IDX(j, j, D)] = 1.0 / L[IDX(j, j, D)];
    for (int i = j + 1; i < D; i++) {
      double s = 0.0;
      for (int k = j; k < i; k++) s += L[IDX(i, k, D)] * X[IDX(k, j, D)];
      X[IDX(i, j, D)] = -s / L[IDX(i, i, D)];
    }
  }
  memcpy(L, X, sizeof(double) * (size_t)D * D);
  free(X);
  return 0;
}

int plda_oracle_invert(double *A, int D) {
  /* Gauss-Jordan with partial pivoting on [A | I]. */
  double *inv = dalloc((size_t)D * D);
  if (!inv) return -1;
  for (int i = 0; i < D; i++) inv[IDX(i, i, D)] = 1.0;
  for (int c = 0; c < D; c++) {
    int piv = c;
    double best = fabs(A[IDX(c, c, D)]);
    for (int r = c + 1; r < D; r++) {
      double v = fabs(A[IDX(r, c, D)]);
      if (v > best) { best = v; piv = r; }
    }
    if (best == 0.0) { free(inv); return -1; }
    if (piv != c) {
      for (int j = 0; j < D; j++) {
        double t = A[IDX(c, j, D)]; A[IDX(c, j, D)] = A[IDX(piv, j, D)]; A[IDX(piv, j, D)] = t;
        t = inv[IDX(c, j, D)]; inv[IDX(c, j, D)] = inv[IDX(piv, j, D)]; inv[IDX(piv, j, D)] = t;
      }
    }
    double p = 1.0 / A[IDX(c, c, D)];
    for (int j = 0; j < D; j++) { A[IDX(c, j, D)] *= p; inv[IDX(c, j, D)] *= p; }
    for (int r = 0; r < D; r++) {
      if (r == c) continue;
      double f = A[IDX(r, c, D)];
      if (f == 0.0) continue;
      for (int j = 0; j < D; j++) {
        A[IDX(r, j, D)] -= f * A[IDX(c, j, D)];
        inv[IDX(r, j, D)] -= f * inv[IDX(c, j, D)];
      }
    }
  }
  memcpy(A, inv, sizeof(double) * (size_t)D * D);
  free(inv);
  return 0;
}

/* symmetric inverse, then symmetrise (SpMatrix::Invert keeps one triangle) */
static int sym_invert(double *A, int D) {
  if (plda_oracle_invert(A, D) != 0) return -1;
  for (int i = 0; i < D; i++)
    for (int j = 0; j < i; j++) {
      double v = 0.5 * (A[IDX(i, j, D)] + A[IDX(j, i, D)]);
      A[IDX(i, j, D)] = v; A[IDX(j, i, D)] = v;
    }
  return 0;
}

/* Householder tridiagonalisation (EISPACK tred2 family). V in/out row-major. */
static void tridiagonalise(double *V, int n, double *d, double *e) {
  for (int j = 0; j < n; j++) d[j] = V[IDX(n - 1, j, n)];
  for (int i = n - 1; i > 0; i--) {
    double scale = 0.0, h = 0.0;
    for (int k = 0; k < i; k++) scale += fabs(d[k]);
    if (scale == 0.0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; j++) {
        d[j] = V[IDX(i - 1, j, n)];
        V[IDX(i, j, n)] = 0.0; V[IDX(j, i, n)] = 0.0;
      }
    } else {
      for (int k = 0; k < i; k++) { d[k] /= scale; h += d[k] * d[k]; }
      double f = d[i - 1];
      double g = sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h -= f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; j++) e[j] = 0.0;
      for (int j = 0; j < i; j++) {
        f = d[j];
        V[IDX(j, i, n)] = f;
        g = e[j] + V[IDX(j, j, n)] * f;
        for (int k = j + 1; k <= i - 1; k++) {
          g += V[IDX(k, j, n)] * d[k];
          e[k] += V[IDX(k, j, n)] * f;
        }
        e[j] = g;
      }
      f = 0.0;
      for (int j = 0; j < i; j++) { e[j] /= h; f += e[j] * d[j]; }
      double hh = f / (h + h);
      for (int j = 0; j < i; j++) e[j] -= hh * d[j];
      for (int j = 0; j < i; j++) {
        f = d[j]; g = e[j];
        for (int k = j; k <= i - 1; k++) V[IDX(k, j, n)] -= (f * e[k] + g * d[k]);
        d[j] = V[IDX(i - 1, j, n)];
        V[IDX(i, j, n)] = 0.0;
      }
    }
    d[i] = h;
  }
  for (int i = 0; i < n - 1; i++) {
    V[IDX(n - 1, i, n)] = V[IDX(i, i, n)];
    V[IDX(i, i, n)] = 1.0;
    double h = d[i + 1];
    if (h != 0.0) {
      for (int k = 0; k <= i; k++) d[k] = V[IDX(k, i + 1, n)] / h;
      for (int j = 0; j <= i; j++) {
        double g = 0.0;
        for (int k = 0; k <= i; k++) g += V[IDX(k, i + 1, n)] * V[IDX(k, j, n)];
        for (int k = 0; k <= i; k++) V[IDX(k, j, n)] -= g * d[k];
      }
    }
    for (int k = 0; k <= i; k++) V[IDX(k, i + 1, n)] = 0.0;
  }
  for (int j = 0; j < n; j++) { d[j] = V[IDX(n - 1, j, n)]; V[IDX(n - 1, j, n)] = 0.0; }
  V[IDX(n - 1, n - 1, n)] = 1.0;
  e[0] = 0.0;
}

/* implicit-shift QL on the tridiagonal (d,e), accumulating into V. */
static int tridiag_ql(double *V, int n, double *d, double *e) {
  for (int i = 1; i < n; i++) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = 2.220446049250313e-16;
  for (int l = 0; l < n; l++) {
    double t = fabs(d[l]) + fabs(e[l]);
    if (t > tst1) tst1 = t;
    int m = l;
    while (m < n) { if (fabs(e[m]) <= eps * tst1) break; m++; }
    if (m > l) {
      int iter = 0;
      do {
        if (++iter > 200) return -1;
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; i++) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c, el1 = e[l + 1], s = 0.0, s2 = 0.0;
        for (int i = m - 1; i >= l; i--) {
          c3 = c2; c2 = c; s2 = s;
          g = c * e[i];
          h = c * p;
          r = hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          for (int k = 0; k < n; k++) {
            h = V[IDX(k, i + 1, n)];
            V[IDX(k, i + 1, n)] = s * V[IDX(k, i, n)] + c * h;
            V[IDX(k, i, n)] = c * V[IDX(k, i, n)] - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (fabs(e[l]) > eps * tst1);
    }
    d[l] += f;
    e[l] = 0.0;
  }
  return 0;
}

int plda_oracle_sym_eig(double *A, int D, double *s, double *U) {
  double *e = dalloc((size_t)D);
  if (!e) return -1;
  memcpy(U, A, sizeof(double) * (size_t)D * D);
  if (D == 1) { s[0] = A[0]; U[0] = 1.0; free(e); return 0; }
  tridiagonalise(U, D, s, e);
  int rc = tridiag_ql(U, D, s, e);
  free(e);
  return rc;
}

/* ------------------------------------------------------------------ */
/* Plda model ops                                                      */
/* ------------------------------------------------------------------ */

/* Plda::TransformIvector + GetNormalizationFactor;
 * called at pldamodule.cpp:171 (transform) and :224 (norm). */
double plda_oracle_transform_ivector(const double *transform, const double *offset,
                                     const double *psi, int D, const double *x,
                                     int num_examples, int normalize_length,
                                     int simple_length_norm, double *out) {
  for (int i = 0; i < D; i++) {
    double t = offset[i];
    const double *row = transform + IDX(i, 0, D);
    for (int j = 0; j < D; j++) t += row[j] * x[j];
    out[i] = t;
  }
  double factor;
  if (simple_length_norm) {
    double nrm = 0.0;
    for (int i = 0; i < D; i++) nrm += out[i] * out[i];
    factor = sqrt((double)D) / sqrt(nrm);
  } else {
    double dot = 0.0;
    for (int i = 0; i < D; i++) dot += out[i] * out[i] / (psi[i] + 1.0 / (double)num_examples);
    factor = sqrt((double)D / dot);
  }
  if (normalize_length)
    for (int i = 0; i < D; i++) out[i] *= factor;
  return factor;
}

/* Plda::LogLikelihoodRatio; called at pldamodule.cpp:235,266. */
double plda_oracle_llr(const double *psi, int D, const double *train, int n,
                       const double *test) {
  static const double LOG_2PI = 1.8378770664093454835606594728112;
  double logdet_given = 0.0, quad_given = 0.0;
  for (int i = 0; i < D; i++) {
    double mean = (double)n * psi[i] / ((double)n * psi[i] + 1.0) * train[i];
    double var = 1.0 + psi[i] / ((double)n * psi[i] + 1.0);
    double diff = test[i] - mean;
    logdet_given += log(var);
    quad_given += diff * diff / var;
  }
  double given = -0.5 * (logdet_given + LOG_2PI * D + quad_given);
  double logdet_wo = 0.0, quad_wo = 0.0;
  for (int i = 0; i < D; i++) {
    double var = psi[i] + 1.0;
    logdet_wo += log(var);
    quad_wo += test[i] * test[i] / var;
  }
  double without = -0.5 * (logdet_wo + LOG_2PI * D + quad_wo);
  return given - without;
}

/* Plda::SmoothWithinClassCovariance + ComputeDerivedVars;
 * called at pldamodule.cpp:159. */
void plda_oracle_smooth(double *transform, double *psi, double *offset,
                        const double *mean, int D, double factor) {
  for (int i = 0; i < D; i++) {
    double wc = 1.0 + factor * psi[i];
    psi[i] /= wc;
    double sc = pow(wc, -0.5);
    for (int j = 0; j < D; j++) transform[IDX(i, j, D)] *= sc;
  }
  for (int i = 0; i < D; i++) {
    double t = 0.0;
    for (int j = 0; j < D; j++) t += transform[IDX(i, j, D)] * mean[j];
    offset[i] = -t;
  }
}

/* ------------------------------------------------------------------ */
/* PldaStats / PldaEstimator                                           */
/* ------------------------------------------------------------------ */

/* PldaStats::AddSamples(w, group) looped as pldamodule.cpp:94-98 does
 * (w = 1/n_k, quirk Q1).  SURVEY.md A.1. */
int plda_oracle_stats(const double *X, int64_t N, int D, const uint64_t *labels,
                      int64_t K, double *means, int64_t *counts, double *scatter,
                      double *sum, double *class_weight, double *example_weight) {
  memset(means, 0, sizeof(double) * (size_t)K * D);
  memset(counts, 0, sizeof(int64_t) * (size_t)K);
  memset(scatter, 0, sizeof(double) * (size_t)D * D);
  memset(sum, 0, sizeof(double) * (size_t)D);
  /* bucket row ids per label (pldamodule.cpp:88-92) */
  for (int64_t r = 0; r < N; r++) {
    if (labels[r] >= (uint64_t)K) return -1;
    counts[labels[r]]++;
  }
  int64_t *start = (int64_t *)calloc((size_t)K + 1, sizeof(int64_t));
  int64_t *fill = (int64_t *)calloc((size_t)K + 1, sizeof(int64_t));
  int64_t *rows = (int64_t *)calloc((size_t)(N ? N : 1), sizeof(int64_t));
  if (!start || !fill || !rows) { free(start); free(fill); free(rows); return -1; }
  for (int64_t k = 0; k < K; k++) start[k + 1] = start[k] + counts[k];
  for (int64_t r = 0; r < N; r++) { int64_t k = (int64_t)labels[r]; rows[start[k] + fill[k]++] = r; }
  double cw = 0.0, ew = 0.0;
  for (int64_t k = 0; k < K; k++) {
    int64_t n = counts[k];
    if (n == 0) { free(start); free(fill); free(rows); return -1; } /* not dense */
    double w = 1.0 / (double)n;
    double *m = means + IDX(k, 0, D);
    /* mean->AddRowSumMat(1/n, group) */
    for (int64_t t = 0; t < n; t++) {
      const double *x = X + IDX(rows[start[k] + t], 0, D);
      for (int j = 0; j < D; j++) m[j] += x[j];
    }
    for (int j = 0; j < D; j++) m[j] /= (double)n;
    /* offset_scatter.AddMat2(w, group, kTrans, 1.0) */
    for (int64_t t = 0; t < n; t++) {
      const double *x = X + IDX(rows[start[k] + t], 0, D);
      for (int i = 0; i < D; i++) {
        double wx = w * x[i];
        double *srow = scatter + IDX(i, 0, D);
        for (int j = 0; j <= i; j++) srow[j] += wx * x[j];
      }
    }
    /* offset_scatter.AddVec2(-n*w, mean) */
    for (int i = 0; i < D; i++) {
      double a = -(double)n * w * m[i];
      double *srow = scatter + IDX(i, 0, D);
      for (int j = 0; j <= i; j++) srow[j] += a * m[j];
    }
    cw += w; ew += w * (double)n;
    for (int j = 0; j < D; j++) sum[j] += w * m[j];
  }
  for (int i = 0; i < D; i++)
    for (int j = 0; j < i; j++) scatter[IDX(j, i, D)] = scatter[IDX(i, j, D)];
  *class_weight = cw; *example_weight = ew;
  free(start); free(fill); free(rows);
  return 0;
}

/* PldaEstimator::EstimateOneIter = ResetPerIterStats + GetStatsFromIntraClass
 * + GetStatsFromClassMeans + EstimateFromStats; reached via
 * estimator.Estimate at pldamodule.cpp:106. */
int plda_oracle_em_iter(const double *means, const int64_t *counts, int64_t K,
                        int D, const double *scatter, const double *sum,
                        double class_weight, double example_weight,
                        double *W, double *B) {
  size_t DD = (size_t)D * D;
  double *Wst = dalloc(DD), *Bst = dalloc(DD), *Binv = dalloc(DD), *Winv = dalloc(DD);
  double *mixed = dalloc(DD), *m = dalloc(D), *tmp = dalloc(D), *w = dalloc(D), *mw = dalloc(D);
  int rc = -3;
  if (!Wst || !Bst || !Binv || !Winv || !mixed || !m || !tmp || !w || !mw) goto done;
  /* GetStatsFromIntraClass */
  memcpy(Wst, scatter, sizeof(double) * DD);
  double Wcount = example_weight - class_weight, Bcount = 0.0;
  /* GetStatsFromClassMeans */
  memcpy(Binv, B, sizeof(double) * DD);
  memcpy(Winv, W, sizeof(double) * DD);
  if (sym_invert(Binv, D) != 0 || sym_invert(Winv, D) != 0) goto done;
  int64_t ncur = -1;
  for (int64_t k = 0; k < K; k++) {
    int64_t n = counts[k];
    double weight = 1.0 / (double)n; /* pldamodule.cpp:97 */
    if (n != ncur) {
      ncur = n;
      for (size_t t = 0; t < DD; t++) mixed[t] = Binv[t] + (double)n * Winv[t];
      if (sym_invert(mixed, D) != 0) goto done;
    }
    const double *mk = means + IDX(k, 0, D);
    for (int j = 0; j < D; j++) m[j] = mk[j] - sum[j] / class_weight;
    for (int i = 0; i < D; i++) {
      double t = 0.0;
      for (int j = 0; j < D; j++) t += Winv[IDX(i, j, D)] * m[j];
      tmp[i] = (double)n * t;
    }
    for (int i = 0; i < D; i++) {
      double t = 0.0;
      for (int j = 0; j < D; j++) t += mixed[IDX(i, j, D)] * tmp[j];
      w[i] = t; mw[i] = m[i] - t;
    }
    double wn = weight * (double)n;
    for (int i = 0; i < D; i++)
      for (int j = 0; j <= i; j++) {
        double mx = mixed[IDX(i, j, D)];
        Bst[IDX(i, j, D)] += weight * (mx + w[i] * w[j]);
        Wst[IDX(i, j, D)] += wn * (mx + mw[i] * mw[j]);
      }
    Bcount += weight; Wcount += weight;
  }
  /* EstimateFromStats */
  for (int i = 0; i < D; i++)
    for (int j = 0; j <= i; j++) {
      double wv = Wst[IDX(i, j, D)] / Wcount, bv = Bst[IDX(i, j, D)] / Bcount;
      W[IDX(i, j, D)] = wv; W[IDX(j, i, D)] = wv;
      B[IDX(i, j, D)] = bv; B[IDX(j, i, D)] = bv;
    }
  rc = 0;
done:
  free(Wst); free(Bst); free(Binv); free(Winv); free(mixed); free(m); free(tmp); free(w); free(mw);
  return rc;
}

/* PldaEstimator::GetOutput + ComputeNormalizingTransform + SortSvd +
 * Plda::ComputeDerivedVars. */
int plda_oracle_get_output(const double *W, const double *B, const double *sum,
                           double class_weight, int D, double *mean,
                           double *transform, double *psi, double *offset) {
  size_t DD = (size_t)D * D;
  double *T1 = dalloc(DD), *tmp = dalloc(DD), *Bp = dalloc(DD), *U = dalloc(DD), *s = dalloc(D);
  int *ord = (int *)calloc((size_t)D, sizeof(int));
  int rc = -3;
  if (!T1 || !tmp || !Bp || !U || !s || !ord) goto done;
  for (int j = 0; j < D; j++) mean[j] = sum[j] / class_weight;
  memcpy(T1, W, sizeof(double) * DD);
  if (plda_oracle_cholesky(T1, D) != 0) goto done;
  if (plda_oracle_tri_invert(T1, D) != 0) goto done;
  /* between_var_proj = T1 B T1^T */
  for (int i = 0; i < D; i++)
    for (int j = 0; j < D; j++) {
      double t = 0.0;
      for (int k = 0; k <= i; k++) t += T1[IDX(i, k, D)] * B[IDX(k, j, D)];
      tmp[IDX(i, j, D)] = t;
    }
  for (int i = 0; i < D; i++)
    for (int j = 0; j <= i; j++) {
      double t = 0.0;
      for (int k = 0; k <= j; k++) t += tmp[IDX(i, k, D)] * T1[IDX(j, k, D)];
      Bp[IDX(i, j, D)] = t; Bp[IDX(j, i, D)] = t;
    }
  if (plda_oracle_sym_eig(Bp, D, s, U) != 0) goto done;
  for (int i = 0; i < D; i++) if (s[i] < 0.0) s[i] = 0.0; /* ApplyFloor(0.0) */
  /* SortSvd: descending */
  for (int i = 0; i < D; i++) ord[i] = i;
  for (int i = 1; i < D; i++) {
    int o = ord[i]; int j = i - 1;
    while (j >= 0 && s[ord[j]] < s[o]) { ord[j + 1] = ord[j]; j--; }
    ord[j + 1] = o;
  }
  /* transform = U^T T1 (rows = sorted eigenvectors) */
  for (int i = 0; i < D; i++) {
    int c = ord[i];
    psi[i] = s[c];
    for (int j = 0; j < D; j++) {
      double t = 0.0;
      for (int k = j; k < D; k++) t += U[IDX(k, c, D)] * T1[IDX(k, j, D)];
      transform[IDX(i, j, D)] = t;
    }
  }
  for (int i = 0; i < D; i++) {
    double t = 0.0;
    for (int j = 0; j < D; j++) t += transform[IDX(i, j, D)] * mean[j];
    offset[i] = -t;
  }
  rc = 0;
done:
  free(T1); free(tmp); free(Bp); free(U); free(s); free(ord);
  return rc;
}

static int cmp_class(const void *a, const void *b) {
  const int64_t *x = (const int64_t *)a, *y = (const int64_t *)b;
  if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
  return x[1] < y[1] ? -1 : (x[1] > y[1]);
}

/* MPlda_fit, pldamodule.cpp:42-109. */
int plda_oracle_fit(const double *X, int64_t N, int D, const uint64_t *labels,
                    int iters, double *mean, double *transform, double *psi,
                    double *offset, double *W_out, double *B_out) {
  if (!X || !labels || N <= 0 || D <= 0) return -1;
  int64_t K = 0;
  for (int64_t r = 0; r < N; r++) if ((int64_t)labels[r] + 1 > K) K = (int64_t)labels[r] + 1;
  if (K == 1) return -2; /* pldamodule.cpp:83-86 */
  size_t DD = (size_t)D * D;
  double *means = dalloc((size_t)K * D), *scatter = dalloc(DD), *sum = dalloc(D);
  double *smeans = dalloc((size_t)K * D), *W = dalloc(DD), *B = dalloc(DD);
  int64_t *counts = (int64_t *)calloc((size_t)K, sizeof(int64_t));
  int64_t *scounts = (int64_t *)calloc((size_t)K, sizeof(int64_t));
  int64_t *keys = (int64_t *)calloc((size_t)K * 2, sizeof(int64_t));
  int rc = -3;
  double cw, ew;
  if (!means || !scatter || !sum || !smeans || !W || !B || !counts || !scounts || !keys) goto done;
  rc = plda_oracle_stats(X, N, D, labels, K, means, counts, scatter, sum, &cw, &ew);
  if (rc != 0) { rc = -1; goto done; }
  /* stats.Sort(): by num_examples (pldamodule.cpp:100) */
  for (int64_t k = 0; k < K; k++) { keys[2 * k] = counts[k]; keys[2 * k + 1] = k; }
  qsort(keys, (size_t)K, 2 * sizeof(int64_t), cmp_class);
  for (int64_t k = 0; k < K; k++) {
    scounts[k] = keys[2 * k];
    memcpy(smeans + IDX(k, 0, D), means + IDX(keys[2 * k + 1], 0, D), sizeof(double) * (size_t)D);
  }
  /* InitParameters: W = B = I */
  for (int i = 0; i < D; i++) { W[IDX(i, i, D)] = 1.0; B[IDX(i, i, D)] = 1.0; }
  rc = 0;
  for (int it = 0; it < iters && rc == 0; it++)
    rc = plda_oracle_em_iter(smeans, scounts, K, D, scatter, sum, cw, ew, W, B);
  if (rc != 0) goto done;
  if (W_out) memcpy(W_out, W, sizeof(double) * DD);
  if (B_out) memcpy(B_out, B, sizeof(double) * DD);
  rc = plda_oracle_get_output(W, B, sum, cw, D, mean, transform, psi, offset);
done:
  free(means); free(scatter); free(sum); free(smeans); free(W); free(B);
  free(counts); free(scounts); free(keys);
  return rc;
}

/* ------------------------------------------------------------------ */
/* transform / norm / score (wrapper level)                            */
/* ------------------------------------------------------------------ */

typedef struct { uint64_t label; int64_t row; } lab_row;
static int cmp_lab(const void *a, const void *b) {
  const lab_row *x = (const lab_row *)a, *y = (const lab_row *)b;
  if (x->label != y->label) return x->label < y->label ? -1 : 1;
  return x->row < y->row ? -1 : (x->row > y->row);
}

/* Mplda_transform, pldamodule.cpp:111-194 (std::map iteration = ascending
 * label, :164; per row AddVec in row order, :147-156). */
int plda_oracle_transform_groups(double *transform, double *offset, double *psi,
                                 const double *mean, int D, const double *X,
                                 int64_t N, const uint64_t *labels,
                                 double smoothfactor, uint64_t *out_labels,
                                 int64_t *out_counts, double *out_vecs,
                                 int64_t *Ku) {
  if (N <= 0) { *Ku = 0; return 0; }
  lab_row *lr = (lab_row *)malloc(sizeof(lab_row) * (size_t)N);
  double *acc = dalloc(D);
  if (!lr || !acc) { free(lr); free(acc); return -1; }
  for (int64_t r = 0; r < N; r++) { lr[r].label = labels[r]; lr[r].row = r; }
  qsort(lr, (size_t)N, sizeof(lab_row), cmp_lab);
  if (smoothfactor != 1.0) plda_oracle_smooth(transform, psi, offset, mean, D, smoothfactor);
  int64_t g = 0, cap = *Ku;
  int64_t r = 0;
  int rc = 0;
  while (r < N) {
    int64_t r1 = r;
    while (r1 < N && lr[r1].label == lr[r].label) r1++;
    if (g >= cap) { rc = -1; break; }
    int64_t n = r1 - r;
    memset(acc, 0, sizeof(double) * (size_t)D);
    for (int64_t t = r; t < r1; t++) {
      const double *x = X + IDX(lr[t].row, 0, D);
      for (int j = 0; j < D; j++) acc[j] += x[j];
    }
    for (int j = 0; j < D; j++) acc[j] *= 1.0 / (double)n; /* Scale(1/n), :168 */
    plda_oracle_transform_ivector(transform, offset, psi, D, acc, (int)n, 1, 0,
                                  out_vecs + IDX(g, 0, D));
    out_labels[g] = lr[r].label; out_counts[g] = n;
    g++; r = r1;
  }
  *Ku = g;
  free(lr); free(acc);
  return rc;
}

/* MPlda_norm, pldamodule.cpp:196-256 (numutts = 0 => all rows). */
int plda_oracle_norm(const double *transform, const double *offset,
                     const double *psi, int D, const double *bkg, int64_t Nb,
                     const double *models, int64_t M, double *out_mean,
                     double *out_std) {
  double *t = dalloc((size_t)Nb * D);
  double *sc = dalloc((size_t)Nb);
  if (!t || !sc) { free(t); free(sc); return -1; }
  for (int64_t i = 0; i < Nb; i++) /* :224 num_examples = bkg.NumRows() */
    plda_oracle_transform_ivector(transform, offset, psi, D, bkg + IDX(i, 0, D),
                                  (int)Nb, 1, 0, t + IDX(i, 0, D));
  for (int64_t k = 0; k < M; k++) {
    const double *repr = models + IDX(k, 0, D);
    double sum = 0.0;
    for (int64_t i = 0; i < Nb; i++) { /* :235 LLR(transformed, 1, repr) */
      sc[i] = plda_oracle_llr(psi, D, t + IDX(i, 0, D), 1, repr);
      sum += sc[i];
    }
    double mean = sum / (double)Nb, sq = 0.0;
    for (int64_t i = 0; i < Nb; i++) sq += (sc[i] - mean) * (sc[i] - mean);
    out_mean[k] = mean;
    out_std[k] = sqrt(sq / (double)Nb);
  }
  free(t); free(sc);
  return 0;
}

/* MPlda_score, pldamodule.cpp:258-277, driven as scorePLDA.py:302-318 does. */
void plda_oracle_score_block(const double *psi, int D, const double *U,
                             const int32_t *n_enrol, int64_t M, const double *V,
                             int64_t Nt, const double *zmean, const double *zstd,
                             double *out) {
  for (int64_t i = 0; i < M; i++)
    for (int64_t j = 0; j < Nt; j++) {
      /* pyarraytovector copies, :264-265 */
      double *e = (double *)malloc(sizeof(double) * (size_t)D);
      double *t = (double *)malloc(sizeof(double) * (size_t)D);
      memcpy(e, U + IDX(i, 0, D), sizeof(double) * (size_t)D);
      memcpy(t, V + IDX(j, 0, D), sizeof(double) * (size_t)D);
      double s = plda_oracle_llr(psi, D, e, n_enrol[i], t);
      if (zmean && zstd) s = (s - zmean[i]) / zstd[i]; /* :269-273 */
      out[(size_t)i * (size_t)Nt + (size_t)j] = s;
      free(e); free(t);
    }
}

/* PldaEstimator::ComputeObjf; test invariant only. */
double plda_oracle_objective(const double *means, const int64_t *counts,
                             int64_t K, int D, const double *scatter,
                             const double *sum, double class_weight,
                             double example_weight, const double *W,
                             const double *B) {
  static const double LOG_2PI = 1.8378770664093454835606594728112;
  size_t DD = (size_t)D * D;
  double *C = dalloc(DD), *Winv = dalloc(DD), *comb = dalloc(DD), *cinv = dalloc(DD), *m = dalloc(D);
  double obj = NAN;
  if (!C || !Winv || !comb || !cinv || !m) goto done;
  memcpy(C, W, sizeof(double) * DD);
  if (plda_oracle_cholesky(C, D) != 0) goto done;
  double logdetW = 0.0;
  for (int i = 0; i < D; i++) logdetW += 2.0 * log(C[IDX(i, i, D)]);
  memcpy(Winv, W, sizeof(double) * DD);
  if (sym_invert(Winv, D) != 0) goto done;
  double tr = 0.0;
  for (size_t t = 0; t < DD; t++) tr += Winv[t] * scatter[t];
  double within = -0.5 * ((example_weight - class_weight) * (logdetW + LOG_2PI * D) + tr);
  double between = 0.0;
  for (int64_t k = 0; k < K; k++) {
    double n = (double)counts[k], wk = 1.0 / n;
    for (size_t t = 0; t < DD; t++) comb[t] = B[t] + W[t] / n;
    memcpy(cinv, comb, sizeof(double) * DD);
    if (plda_oracle_cholesky(comb, D) != 0) goto done;
    double ld = 0.0;
    for (int i = 0; i < D; i++) ld += 2.0 * log(comb[IDX(i, i, D)]);
    if (sym_invert(cinv, D) != 0) goto done;
    const double *mk = means + IDX(k, 0, D);
    for (int j = 0; j < D; j++) m[j] = mk[j] - sum[j] / class_weight;
    double q = 0.0;
    for (int i = 0; i < D; i++)
      for (int j = 0; j < D; j++) q += m[i] * cinv[IDX(i, j, D)] * m[j];
    between += wk * -0.5 * (ld + LOG_2PI * D + q);
  }
  obj = (within + between) / example_weight;
done:
  free(C); free(Winv); free(comb); free(cinv); free(m);
  return obj;
}
