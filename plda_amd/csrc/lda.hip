// plda_amd/csrc/lda.hip -- LDA on gfx950: the reference's second model,
// /root/reference/python/liblda/lda.py, rebuilt on the PLDA kernels.
//
//   fit   'svd'   lda.py:171-209  within-class whitening + SVD of the scaled centroids.  Both SVDs are taken
//                                 through the Gram matrix (D x D weighted SYRK over the N rows, then the
//                                 symmetric eigensolver): X is read once, nothing N-sized is ever factorised.
//         'eigen' lda.py:134-169  Sb v = lambda Sw v by simultaneous diagonalisation (the PLDA GetOutput kernels)
//         'lsqr'  lda.py:211-240  coef = Sw^+ means^T through the eigendecomposition of Sw
//   predict       lda.py:242-314  decision = X coef^T + intercept (fp64 MFMA GEMM), row kernels for the
//                                 log-softmax / one-vs-rest logistic
//   transform     lda.py:317-338
// Everything is fp64, like the NumPy reference.
#include "common.hpp"

#include <cmath>
#include <vector>

namespace plda {

enum { LDA_SVD = 0, LDA_EIGEN = 1, LDA_LSQR = 2 };

__global__ void lda_row_weight_kernel(const uint64_t *__restrict__ labels, const double *__restrict__ cw, int64_t N,
                                      double *__restrict__ rw) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < N) rw[i] = cw[labels[i]];
}

// out[d] = sum_k wk[k] means[k][d]
__global__ void lda_weighted_colsum_kernel(const double *__restrict__ means, const double *__restrict__ wk, int64_t K,
                                           int D, double *__restrict__ out) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  double s = 0.0;
  for (int64_t k = 0; k < K; ++k) s = fma(wk[k], means[k * D + d], s);
  out[d] = s;
}

// out[k][d] = (means[k][d] - xbar[d]) * (rowscale ? rowscale[k] : 1)
__global__ void lda_center_kernel(const double *__restrict__ means, const double *__restrict__ xbar,
                                  const double *__restrict__ rowscale, int64_t K, int D, double *__restrict__ out) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= K * D) return;
  const int64_t k = idx / D;
  const int d = (int)(idx % D);
  out[idx] = (means[idx] - xbar[d]) * (rowscale ? rowscale[k] : 1.0);
}

__global__ void lda_std_kernel(const double *__restrict__ S, int64_t N, int D, double *__restrict__ std) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  const double v = S[(size_t)d * D + d] / (double)N;
  const double s = v > 0.0 ? sqrt(v) : 0.0;
  std[d] = s == 0.0 ? 1.0 : s;     // lda.py:187
}

// G = fac * S / (std std^T), symmetrised
__global__ void lda_whiten_gram_kernel(const double *__restrict__ S, const double *__restrict__ std, double fac, int D,
                                       double *__restrict__ G) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D * D) return;
  const int i = idx / D, j = idx % D;
  const double a = 0.5 * (S[idx] + S[(size_t)j * D + i]);
  G[idx] = fac * a / (std[i] * std[j]);
}

// scal1[d][r] = Vrows[r][d] / std[d] / sqrt(lam[r]),  r < r1
__global__ void lda_scal1_kernel(const double *__restrict__ Vrows, const double *__restrict__ lam,
                                 const double *__restrict__ std, int D, int r1, double *__restrict__ scal1) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D * r1) return;
  const int d = idx / r1, r = idx % r1;
  scal1[idx] = Vrows[(size_t)r * D + d] / std[d] / sqrt(lam[r]);
}

// A -= alpha * u u^T
__global__ void lda_rank1_sub_kernel(double *__restrict__ A, const double *__restrict__ u, int D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D * D) return;
  A[idx] -= u[idx / D] * u[idx % D];
}

// Sb = sym(St - Sw); Sw = sym(Sw)
__global__ void lda_between_kernel(double *__restrict__ Sw, const double *__restrict__ St, int D,
                                   double *__restrict__ Sb) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D * D) return;
  const int i = idx / D, j = idx % D;
  if (j > i) return;
  const size_t ij = (size_t)i * D + j, ji = (size_t)j * D + i;
  const double w = 0.5 * (Sw[ij] + Sw[ji]), t = 0.5 * (St[ij] + St[ji]);
  Sw[ij] = w; Sw[ji] = w;
  Sb[ij] = t - w; Sb[ji] = t - w;
}

__global__ void lda_symmetrize_kernel(double *A, int D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D * D) return;
  const int i = idx / D, j = idx % D;
  if (j >= i) return;
  const double v = 0.5 * (A[(size_t)i * D + j] + A[(size_t)j * D + i]);
  A[(size_t)i * D + j] = v;
  A[(size_t)j * D + i] = v;
}

// scalings[d][q] = T[q][d] / |T[q]|  (one wave per row q of T)
__global__ __launch_bounds__(64) void lda_unit_columns_kernel(const double *__restrict__ T, int D,
                                                              double *__restrict__ scalings) {
  const int q = blockIdx.x, lane = threadIdx.x;
  double s = 0.0;
  for (int d = lane; d < D; d += 64) { const double v = T[(size_t)q * D + d]; s = fma(v, v, s); }
  s = wave_sum_f64(s);
  const double inv = 1.0 / sqrt(s);
  for (int d = lane; d < D; d += 64) scalings[(size_t)d * D + q] = T[(size_t)q * D + d] * inv;
}

// columns r of tmp[K][D] scaled by 1/lam[r] (lam[r] > cut) or 0: the pseudo-inverse spectrum
__global__ void lda_pinv_scale_kernel(double *__restrict__ tmp, const double *__restrict__ lam, double cut, int64_t K,
                                      int D) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= K * D) return;
  const double l = lam[idx % D];
  tmp[idx] = l > cut ? tmp[idx] / l : 0.0;
}

// mode 0 (svd): intercept_k = -0.5 sum_q proj[k][q]^2 + log p_k - xbar . coef_k
// mode 1 (eigen, lsqr): intercept_k = -0.5 means_k . coef_k + log p_k
__global__ __launch_bounds__(64) void lda_intercept_kernel(int mode, const double *__restrict__ proj, int r2,
                                                           const double *__restrict__ vec /*xbar or means*/,
                                                           const double *__restrict__ coef,
                                                           const double *__restrict__ priors, int D,
                                                           double *__restrict__ intercept) {
  const int64_t k = blockIdx.x;
  const int lane = threadIdx.x;
  double a = 0.0, b = 0.0;
  if (mode == 0) {
    for (int q = lane; q < r2; q += 64) { const double v = proj[k * r2 + q]; a = fma(v, v, a); }
    for (int d = lane; d < D; d += 64) b = fma(vec[d], coef[k * D + d], b);
    a = wave_sum_f64(a);
    b = wave_sum_f64(b);
    if (lane == 0) intercept[k] = -0.5 * a + log(priors[k]) - b;
  } else {
    for (int d = lane; d < D; d += 64) a = fma(vec[k * D + d], coef[k * D + d], a);
    a = wave_sum_f64(a);
    if (lane == 0) intercept[k] = -0.5 * a + log(priors[k]);
  }
}

// One workgroup per sample row of the decision matrix [N][K], in place.
//   mode 0: v + intercept                       (decision_function, lda.py:268)
//   mode 1: log-softmax of (v + intercept)      (predict_log_proba, lda.py:311-314)
//   mode 2: logistic 1 / (1 + exp(-(v + b)))    (first half of predict_proba, lda.py:283-287)
//   mode 3: mode 2 divided by its row sum       (one-vs-rest normalisation, lda.py:292)
__global__ __launch_bounds__(256) void lda_row_kernel(double *__restrict__ out, const double *__restrict__ intercept,
                                                      int64_t K, int mode) {
  __shared__ double red[4];
  double *row = out + (size_t)blockIdx.x * K;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (mode == 0) {
    for (int64_t k = t; k < K; k += 256) row[k] += intercept[k];
    return;
  }
  if (mode == 1) {
    double m = -INFINITY;
    for (int64_t k = t; k < K; k += 256) {
      const double v = row[k] + intercept[k];
      row[k] = v;
      m = fmax(m, v);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    double s = 0.0;
    for (int64_t k = t; k < K; k += 256) s += exp(row[k] - m);
    s = wave_sum_f64(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const double lse = log(red[0] + red[1] + red[2] + red[3]);
    for (int64_t k = t; k < K; k += 256) row[k] = (row[k] - m) - lse;
    return;
  }
  double s = 0.0;
  for (int64_t k = t; k < K; k += 256) {
    const double p = 1.0 / (1.0 + exp(-(row[k] + intercept[k])));
    row[k] = p;
    s += p;
  }
  if (mode == 3) {
    s = wave_sum_f64(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const double tot = red[0] + red[1] + red[2] + red[3];
    for (int64_t k = t; k < K; k += 256) row[k] /= tot;
  }
}

__global__ void lda_sub_rowvec_kernel(double *__restrict__ out, const double *__restrict__ v, int64_t N, int C) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx < N * C) out[idx] -= v[idx % C];
}

static inline unsigned blocks(int64_t n) { return (unsigned)ceil_div(n, 256); }

int lda_fit_device(plda_handle *h, const double *dX, int64_t N, int D, const uint64_t *dlabels, int64_t K,
                   int solver, const double *priors_host) {
  if (!dX || !dlabels || N <= 0 || D <= 0 || K <= 0 || K > N) return fail(h, PLDA_E_INVAL, "lda_fit: bad argument");
  if (solver < LDA_SVD || solver > LDA_LSQR) return fail(h, PLDA_E_INVAL, "lda_fit: unknown solver %d", solver);
  if (D > 2048) return fail(h, PLDA_E_INVAL, "lda_fit: featdim %d > 2048 unsupported", D);
  if (solver == LDA_SVD && N <= K) return fail(h, PLDA_E_INVAL, "lda_fit: the svd solver needs more samples than classes");
  const size_t DD = (size_t)D * D;
  h->lda_fitted = false;
  PLDA_HIP(h, h->l_means.reserve((size_t)K * D * 8));
  PLDA_HIP(h, h->l_priors.reserve((size_t)K * 8));
  PLDA_HIP(h, h->l_xbar.reserve((size_t)D * 8));
  PLDA_HIP(h, h->l_scalings.reserve(DD * 8));
  PLDA_HIP(h, h->l_coef.reserve((size_t)K * D * 8));
  PLDA_HIP(h, h->l_intercept.reserve((size_t)K * 8));
  PLDA_HIP(h, h->l_evr.reserve((size_t)D * 8));
  double *means = h->l_means.as<double>(), *dpri = h->l_priors.as<double>(), *xbar = h->l_xbar.as<double>(),
         *scalings = h->l_scalings.as<double>(), *coef = h->l_coef.as<double>(),
         *intercept = h->l_intercept.as<double>();

  // ---- class means and counts (the PLDA K1a/K1 kernels) ----
  // scratch: [counts32 K][cw K][nk K][rw N][S DD][G DD][Vr DD][lam D][std D][mu D][Mc K*D][tmp K*D][small DD*2]
  const size_t bytes = (size_t)K * 4 + 64 + (size_t)K * 16 + (size_t)N * 8 + DD * 8 * 5 + (size_t)D * 8 * 3 +
                       (size_t)K * D * 8 * 2 + 256;
  PLDA_HIP(h, h->w[12].reserve(bytes));
  char *base = h->w[12].as<char>();
  int32_t *dcounts = reinterpret_cast<int32_t *>(base);
  double *cw = reinterpret_cast<double *>(base + round_up((int64_t)K * 4, 64));
  double *nk = cw + K, *rw = nk + K, *S = rw + N, *G = S + DD, *Vr = G + DD, *lam = Vr + DD, *std = lam + D,
         *mu = std + D, *Mc = mu + D, *tmp = Mc + (size_t)K * D, *sm1 = tmp + (size_t)K * D, *sm2 = sm1 + DD;
  PLDA_TRY(group_means_device(h, dX, N, D, dlabels, K, means, dcounts));
  std::vector<int32_t> hc((size_t)K);
  PLDA_HIP(h, hipMemcpyAsync(hc.data(), dcounts, (size_t)K * 4, hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  std::vector<double> p((size_t)K), hcw((size_t)K), hnk((size_t)K);
  double psum = 0.0;
  for (int64_t k = 0; k < K; ++k) {
    if (hc[k] <= 0) return fail(h, PLDA_E_LABELS, "lda_fit: labels must be dense 0..K-1 (label %lld unused)", (long long)k);
    p[k] = priors_host ? priors_host[k] : (double)hc[k] / (double)N;      // lda.py:113-119
    psum += p[k];
  }
  if (psum != 1.0)                                                        // lda.py:121-122
    for (int64_t k = 0; k < K; ++k) p[k] /= psum;
  for (int64_t k = 0; k < K; ++k) {
    if (!(p[k] > 0.0)) return fail(h, PLDA_E_INVAL, "lda_fit: priors must be positive");
    hcw[k] = p[k] / (double)hc[k];
    hnk[k] = (double)hc[k];
  }
  PLDA_HIP(h, hipMemcpyAsync(dpri, p.data(), (size_t)K * 8, hipMemcpyHostToDevice, h->stream));
  PLDA_HIP(h, hipMemcpyAsync(cw, hcw.data(), (size_t)K * 8, hipMemcpyHostToDevice, h->stream));
  PLDA_HIP(h, hipMemcpyAsync(nk, hnk.data(), (size_t)K * 8, hipMemcpyHostToDevice, h->stream));
  int rank = D;

  if (solver == LDA_SVD) {
    lda_weighted_colsum_kernel<<<blocks(D), 256, 0, h->stream>>>(means, dpri, K, D, xbar);   // xbar = priors . means
    // within scatter of the class-centred data: X^T X - sum_k n_k m_k m_k^T
    PLDA_TRY(gemm_f64(h, D, D, N, 1.0, dX, 1, D, dX, D, 1, nullptr, 0.0, S, D));
    PLDA_TRY(gemm_f64(h, D, D, K, -1.0, means, 1, D, means, D, 1, nk, 1.0, S, D));
    const double fac = 1.0 / (double)(N - K);
    lda_std_kernel<<<blocks(D), 256, 0, h->stream>>>(S, N, D, std);
    lda_whiten_gram_kernel<<<blocks((int64_t)DD), 256, 0, h->stream>>>(S, std, fac, D, G);
    PLDA_LAUNCH_CHECK(h);
    PLDA_TRY(sym_eig_auto_f64(h, G, D, lam, Vr));           // singular values^2, right vectors in rows
    std::vector<double> hl((size_t)D);
    PLDA_HIP(h, hipMemcpyAsync(hl.data(), lam, (size_t)D * 8, hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    int r1 = 0;
    while (r1 < D && sqrt(hl[r1]) > 1e-4) ++r1;                           // lda.py:193 (tol = 1e-4, absolute)
    if (r1 == 0) return fail(h, PLDA_E_NUMERIC, "lda_fit: the class-centred data has no variance");
    double *scal1 = sm1;                                                  // [D][r1]
    lda_scal1_kernel<<<blocks((int64_t)D * r1), 256, 0, h->stream>>>(Vr, lam, std, D, r1, scal1);
    // scaled centroids: sqrt(N p_k fac) (m_k - xbar), projected on the whitening directions
    std::vector<double> hs((size_t)K);
    for (int64_t k = 0; k < K; ++k) hs[k] = sqrt((double)N * p[k] * fac);
    PLDA_HIP(h, hipMemcpyAsync(cw, hs.data(), (size_t)K * 8, hipMemcpyHostToDevice, h->stream));
    lda_center_kernel<<<blocks(K * (int64_t)D), 256, 0, h->stream>>>(means, xbar, cw, K, D, Mc);
    PLDA_LAUNCH_CHECK(h);
    double *cen = tmp;                                                    // [K][r1]
    PLDA_TRY(gemm_f64(h, K, r1, D, 1.0, Mc, D, 1, scal1, r1, 1, nullptr, 0.0, cen, r1));
    double *G2 = G, *V2 = Vr;                                             // [r1][r1]
    PLDA_TRY(gemm_f64(h, r1, r1, K, 1.0, cen, 1, r1, cen, r1, 1, nullptr, 0.0, G2, r1));
    PLDA_TRY(sym_eig_auto_f64(h, G2, r1, lam, V2));
    PLDA_HIP(h, hipMemcpyAsync(hl.data(), lam, (size_t)r1 * 8, hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    int r2 = 0;
    while (r2 < r1 && sqrt(hl[r2]) > 1e-4 * sqrt(hl[0])) ++r2;            // lda.py:202
    if (r2 == 0) return fail(h, PLDA_E_NUMERIC, "lda_fit: the class centroids coincide");
    // scalings[D][r2] = scal1 V2^T[:, :r2]
    PLDA_TRY(gemm_f64(h, D, r2, r1, 1.0, scal1, r1, 1, V2, 1, r1, nullptr, 0.0, scalings, r2));
    lda_center_kernel<<<blocks(K * (int64_t)D), 256, 0, h->stream>>>(means, xbar, nullptr, K, D, Mc);
    PLDA_LAUNCH_CHECK(h);
    double *proj = tmp;                                                   // [K][r2]
    PLDA_TRY(gemm_f64(h, K, r2, D, 1.0, Mc, D, 1, scalings, r2, 1, nullptr, 0.0, proj, r2));
    PLDA_TRY(gemm_f64(h, K, D, r2, 1.0, proj, r2, 1, scalings, 1, r2, nullptr, 0.0, coef, D));
    lda_intercept_kernel<<<(unsigned)K, 64, 0, h->stream>>>(0, proj, r2, xbar, coef, dpri, D, intercept);
    PLDA_LAUNCH_CHECK(h);
    rank = r2;
  } else {
    // Sw = sum_k p_k cov_k = X^T diag(p_label / n_label) X - sum_k p_k m_k m_k^T   (lda.py:10-16,153)
    lda_row_weight_kernel<<<blocks(N), 256, 0, h->stream>>>(dlabels, cw, N, rw);
    PLDA_LAUNCH_CHECK(h);
    double *Sw = S;
    PLDA_TRY(gemm_f64(h, D, D, N, 1.0, dX, 1, D, dX, D, 1, rw, 0.0, Sw, D));
    PLDA_TRY(gemm_f64(h, D, D, K, -1.0, means, 1, D, means, D, 1, dpri, 1.0, Sw, D));
    if (solver == LDA_EIGEN) {
      // St = X^T X / N - mu mu^T (lda.py:156), Sb = St - Sw
      double *St = G, *Sb = sm1, *T = sm2;
      PLDA_TRY(gemm_f64(h, D, D, N, 1.0 / (double)N, dX, 1, D, dX, D, 1, nullptr, 0.0, St, D));
      std::vector<double> hw((size_t)K);
      for (int64_t k = 0; k < K; ++k) hw[k] = (double)hc[k] / (double)N;
      PLDA_HIP(h, hipMemcpyAsync(cw, hw.data(), (size_t)K * 8, hipMemcpyHostToDevice, h->stream));
      lda_weighted_colsum_kernel<<<blocks(D), 256, 0, h->stream>>>(means, cw, K, D, mu);
      lda_rank1_sub_kernel<<<blocks((int64_t)DD), 256, 0, h->stream>>>(St, mu, D);
      lda_between_kernel<<<blocks((int64_t)DD), 256, 0, h->stream>>>(Sw, St, D, Sb);
      PLDA_LAUNCH_CHECK(h);
      h->simdiag_has_vr = false;
      h->eig_keep_sign = true;            // St - Sw is indefinite when the priors are not the class frequencies
      const int rc = simdiag_f64(h, Sw, Sb, D, T, nullptr, lam, false);   // T Sw T^T = I, T Sb T^T = diag(lam) desc
      h->eig_keep_sign = false;
      if (rc != PLDA_OK) return rc;
      lda_unit_columns_kernel<<<D, 64, 0, h->stream>>>(T, D, scalings);   // lda.py:162
      PLDA_LAUNCH_CHECK(h);
      // coef = means E E^T
      PLDA_TRY(gemm_f64(h, K, D, D, 1.0, means, D, 1, scalings, D, 1, nullptr, 0.0, tmp, D));
      PLDA_TRY(gemm_f64(h, K, D, D, 1.0, tmp, D, 1, scalings, 1, D, nullptr, 0.0, coef, D));
      std::vector<double> hl((size_t)D);
      PLDA_HIP(h, hipMemcpyAsync(hl.data(), lam, (size_t)D * 8, hipMemcpyDeviceToHost, h->stream));
      PLDA_HIP(h, hipStreamSynchronize(h->stream));
      double tot = 0.0;
      for (int d = 0; d < D; ++d) tot += hl[d];
      for (int d = 0; d < D; ++d) hl[d] /= tot;
      PLDA_HIP(h, hipMemcpyAsync(h->l_evr.p, hl.data(), (size_t)D * 8, hipMemcpyHostToDevice, h->stream));
      PLDA_HIP(h, hipStreamSynchronize(h->stream));
    } else {
      // coef = (Sw^+ means^T)^T, Sw^+ from the eigendecomposition with numpy.linalg.lstsq's default cut-off
      lda_symmetrize_kernel<<<blocks((int64_t)DD), 256, 0, h->stream>>>(Sw, D);
      PLDA_LAUNCH_CHECK(h);
      PLDA_TRY(sym_eig_auto_f64(h, Sw, D, lam, Vr));
      double lmax = 0.0;
      PLDA_HIP(h, hipMemcpyAsync(&lmax, lam, 8, hipMemcpyDeviceToHost, h->stream));
      PLDA_HIP(h, hipStreamSynchronize(h->stream));
      const double cut = 2.220446049250313e-16 * (double)D * lmax;
      PLDA_TRY(gemm_f64(h, K, D, D, 1.0, means, D, 1, Vr, 1, D, nullptr, 0.0, tmp, D));   // tmp[k][r] = m_k . v_r
      lda_pinv_scale_kernel<<<blocks(K * (int64_t)D), 256, 0, h->stream>>>(tmp, lam, cut, K, D);
      PLDA_LAUNCH_CHECK(h);
      PLDA_TRY(gemm_f64(h, K, D, D, 1.0, tmp, D, 1, Vr, D, 1, nullptr, 0.0, coef, D));
      rank = 0;
    }
    lda_intercept_kernel<<<(unsigned)K, 64, 0, h->stream>>>(1, nullptr, 0, means, coef, dpri, D, intercept);
    PLDA_LAUNCH_CHECK(h);
  }
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  h->lda_fitted = true;
  h->lda_solver = solver;
  h->lda_K = K;
  h->lda_D = D;
  h->lda_rank = rank;
  return PLDA_OK;
}

// out [N][K] fp64, mode as in lda_row_kernel
int lda_predict_device(plda_handle *h, const double *dX, int64_t N, int mode, double *dout) {
  if (!h->lda_fitted) return fail(h, PLDA_E_NOT_FITTED, "This LDA instance is not fitted yet");
  if (N <= 0) return PLDA_OK;
  if (!dX || !dout || mode < 0 || mode > 3) return fail(h, PLDA_E_INVAL, "lda_predict: bad argument");
  if (N > 0x7fffffff) return fail(h, PLDA_E_INVAL, "lda_predict: too many rows in one call");
  const int64_t K = h->lda_K;
  const int D = h->lda_D;
  PLDA_TRY(gemm_f64(h, N, K, D, 1.0, dX, D, 1, h->l_coef.as<double>(), 1, D, nullptr, 0.0, dout, K));
  lda_row_kernel<<<(unsigned)N, 256, 0, h->stream>>>(dout, h->l_intercept.as<double>(), K, mode);
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

// out [N][ncomp]: eigen  X scalings ; svd  (X - xbar) scalings
int lda_transform_device(plda_handle *h, const double *dX, int64_t N, int ncomp, double *dout) {
  if (!h->lda_fitted) return fail(h, PLDA_E_NOT_FITTED, "This LDA instance is not fitted yet");
  if (h->lda_solver == LDA_LSQR) return fail(h, PLDA_E_INVAL, "transform not implemented for 'lsqr' solver (use 'svd' or 'eigen').");
  if (N <= 0 || ncomp <= 0) return PLDA_OK;
  const int D = h->lda_D, R = h->lda_rank;
  if (!dX || !dout || ncomp > R) return fail(h, PLDA_E_INVAL, "lda_transform: bad argument (n_components %d, rank %d)", ncomp, R);
  const double *sc = h->l_scalings.as<double>();
  PLDA_TRY(gemm_f64(h, N, ncomp, D, 1.0, dX, D, 1, sc, R, 1, nullptr, 0.0, dout, ncomp));
  if (h->lda_solver == LDA_SVD) {
    PLDA_HIP(h, h->w[12].reserve((size_t)ncomp * 8));
    double *off = h->w[12].as<double>();
    PLDA_TRY(gemm_f64(h, 1, ncomp, D, 1.0, h->l_xbar.as<double>(), D, 1, sc, R, 1, nullptr, 0.0, off, ncomp));
    lda_sub_rowvec_kernel<<<blocks(N * (int64_t)ncomp), 256, 0, h->stream>>>(dout, off, N, ncomp);
    PLDA_LAUNCH_CHECK(h);
  }
  return PLDA_OK;
}

}  // namespace plda
