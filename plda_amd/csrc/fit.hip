// plda_amd/csrc/fit.hip -- PLDA estimation on gfx950, fp64.
//
// Replaces MPlda_fit (/root/reference/src/pldamodule.cpp:42-109):
//   :76-92   label grouping            -> K1a  stable LSD radix sort of (label,row)
//   :94-98   AddSamples(1/n_k, rows_k) -> K1   segmented centroid accumulation
//                                         K2   weighted SYRK  X^T diag(1/n_label) X  -  M^T M
//   :100     stats.Sort()              -> not needed (no per-distinct-n inversions below)
//   :102-106 Estimate(iters)           -> K3   EM in the simultaneously-diagonalised basis
//                                              (identical maths to A.2)
//            GetOutput                 -> K6/K7 Cholesky + triangular inverse + Jacobi eig
#include "common.hpp"
#include <cstring>

#include <algorithm>
#include <chrono>

namespace plda {

// ------------------------------------------------------------------------------------
// K1a: stable LSD radix sort (8-bit digits) of row ids by label
// ------------------------------------------------------------------------------------
constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 16;                       // items per thread per block
constexpr int RS_CHUNK = RS_THREADS * RS_ITEMS;    // 4096 keys per block

__global__ void labels_check_kernel(const uint64_t *__restrict__ labels, int64_t N, int64_t K,
                                    uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                    int *__restrict__ counts, int *__restrict__ bad) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  const uint64_t l = labels[r];
  if (l >= (uint64_t)K) { *bad = 1; keys[r] = 0; vals[r] = (uint32_t)r; return; }
  keys[r] = (uint32_t)l;
  vals[r] = (uint32_t)r;
  atomicAdd(counts + l, 1);
}

__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const uint32_t *__restrict__ keys, int64_t N,
                                                             int shift, int nblocks,
                                                             int *__restrict__ hist /*[256][nblocks]*/) {
  __shared__ int lh[256];
  lh[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_CHUNK;
  for (int it = 0; it < RS_ITEMS; ++it) {
    const int64_t idx = base + it * RS_THREADS + threadIdx.x;
    if (idx < N) atomicAdd(&lh[(keys[idx] >> shift) & 255], 1);
  }
  __syncthreads();
  hist[threadIdx.x * nblocks + blockIdx.x] = lh[threadIdx.x];
}

// exclusive scan of a flat int array by one workgroup (n up to a few million)
__global__ __launch_bounds__(1024) void scan_kernel(int *__restrict__ data, int64_t n) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t idx = base + t;
    const int v = idx < n ? data[idx] : 0;
    int x = v;
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int carry = carry_s;
    if (idx < n) data[idx] = carry + woff + x - v;
    __syncthreads();
    if (t == 1023) carry_s = carry + woff + x;
    __syncthreads();
  }
}

__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(
    const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin, uint32_t *__restrict__ kout,
    uint32_t *__restrict__ vout, int64_t N, int shift, int nblocks, const int *__restrict__ hist) {
  __shared__ int base_s[256];      // running global position of each digit for this block
  __shared__ int wcount[4][256];   // per-wave digit counts of the current sub-tile
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  base_s[t] = hist[t * nblocks + blockIdx.x];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_CHUNK;
  for (int it = 0; it < RS_ITEMS; ++it) {
    const int64_t idx = base + it * RS_THREADS + t;
    const bool valid = idx < N;
    const uint32_t key = valid ? kin[idx] : 0u;
    const int digit = valid ? (int)((key >> shift) & 255) : 256;  // 256 = inactive
    for (int d = lane; d < 256; d += 64) wcount[wave][d] = 0;
    __syncthreads();
    // peers = lanes of this wave with the same digit
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const unsigned long long m = __ballot((digit >> bit) & 1);
      peers &= ((digit >> bit) & 1) ? m : ~m;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int rank_in_wave = __popcll(peers & lt);
    if (valid && rank_in_wave == 0) wcount[wave][digit] = __popcll(peers);
    __syncthreads();
    int pos = 0;
    if (valid) {
      int woff = 0;
      for (int w = 0; w < wave; ++w) woff += wcount[w][digit];
      pos = base_s[digit] + woff + rank_in_wave;
    }
    __syncthreads();
    {
      const int tot = wcount[0][t] + wcount[1][t] + wcount[2][t] + wcount[3][t];
      base_s[t] += tot;
    }
    if (valid) { kout[pos] = key; vout[pos] = vin[idx]; }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------
// K1: centroids.  One workgroup per speaker; threads span the feature dimension so
// every row read is one contiguous D*8-byte burst; rows are summed in ascending row
// order (deterministic).  Also emits the per-row weight 1/n_label used by K2.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void centroid_kernel(const double *__restrict__ X, int D,
                                                       const uint32_t *__restrict__ perm,
                                                       const int *__restrict__ offsets,
                                                       double *__restrict__ means,
                                                       double *__restrict__ roww) {
  const int k = blockIdx.x;
  const int beg = offsets[k], end = offsets[k + 1];
  const int n = end - beg;
  const double inv = 1.0 / (double)n;
  for (int d0 = threadIdx.x; d0 < D; d0 += blockDim.x) {
    double acc = 0.0;
    int r = beg;
    for (; r + 4 <= end; r += 4) {
      const double a = X[(int64_t)perm[r] * D + d0], b = X[(int64_t)perm[r + 1] * D + d0];
      const double c = X[(int64_t)perm[r + 2] * D + d0], e = X[(int64_t)perm[r + 3] * D + d0];
      acc += a; acc += b; acc += c; acc += e;
    }
    for (; r < end; ++r) acc += X[(int64_t)perm[r] * D + d0];
    means[(int64_t)k * D + d0] = acc * inv;
  }
  for (int r = beg + threadIdx.x; r < end; r += blockDim.x) roww[perm[r]] = inv;
}

// sum_d = sum_k w_k m_kd (w_k = 1/n_k) and class_weight = sum_k w_k, deterministic two-stage:
// (1) class_sum_partial_kernel: grid (D / 64, CS_SPLIT); a block sums its K-slice for 64
//     columns (4 k-sub-slices in parallel, combined in LDS in fixed order) -> partial[s][d];
//     block (0, s) also sums 1/n_k over its slice -> wpart[s];
// (2) class_sum_final_kernel: fixed-order sum over the slices, mu = sum / class_weight.
constexpr int CS_SPLIT = 64;

__global__ __launch_bounds__(256) void class_sum_partial_kernel(const double *__restrict__ means,
                                                                const int *__restrict__ offsets, int64_t K, int D,
                                                                double *__restrict__ partial /*[CS_SPLIT][D]*/,
                                                                double *__restrict__ wpart /*[CS_SPLIT]*/) {
  __shared__ double red[4][64];
  __shared__ double wred[256];
  const int t = threadIdx.x, c = t & 63, sub = t >> 6;
  const int d = blockIdx.x * 64 + c;
  const int64_t per = (K + CS_SPLIT - 1) / CS_SPLIT;
  const int64_t k0 = (int64_t)blockIdx.y * per, k1 = min(K, k0 + per);
  double acc = 0.0;
  if (d < D)
    for (int64_t k = k0 + sub; k < k1; k += 4) acc += means[k * D + d] / (double)(offsets[k + 1] - offsets[k]);
  red[sub][c] = acc;
  if (blockIdx.x == 0) {
    double w = 0.0;
    for (int64_t k = k0 + t; k < k1; k += 256) w += 1.0 / (double)(offsets[k + 1] - offsets[k]);
    wred[t] = w;
  }
  __syncthreads();
  if (sub == 0 && d < D) partial[(size_t)blockIdx.y * D + d] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
  if (blockIdx.x == 0) {
    for (int o = 128; o > 0; o >>= 1) {
      if (t < o) wred[t] += wred[t + o];
      __syncthreads();
    }
    if (t == 0) wpart[blockIdx.y] = wred[0];
  }
}

__global__ void class_sum_final_kernel(const double *__restrict__ partial, const double *__restrict__ wpart, int D,
                                       double *__restrict__ sum, double *__restrict__ mu,
                                       double *__restrict__ scalars /*[0]=class_weight*/) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  double cw = 0.0;
  for (int s2 = 0; s2 < CS_SPLIT; ++s2) cw += wpart[s2];
  if (d == 0) scalars[0] = cw;
  if (d >= D) return;
  double acc = 0.0;
  for (int s2 = 0; s2 < CS_SPLIT; ++s2) acc += partial[(size_t)s2 * D + d];
  sum[d] = acc;
  mu[d] = acc / cw;
}

// ------------------------------------------------------------------------------------
// K3 helpers
// ------------------------------------------------------------------------------------
// Mc = means - mu
__global__ void center_kernel(const double *__restrict__ means, const double *__restrict__ mu, int64_t K,
                              int D, double *__restrict__ Mc) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < K * D) Mc[idx] = means[idx] - mu[idx % D];
}

// Y1 = sqrt(w_k) c_kd p_kd ; Y2 = sqrt(w_k n_k) (1 - c_kd) p_kd   (in place: P -> Y1, Y2)
__global__ void em_scale_kernel(const double *P, const int *__restrict__ offsets,
                                const double *__restrict__ psi, int64_t K, int D, double *Y1, double *Y2) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= K * D) return;
  const int64_t k = idx / D;
  const int d = (int)(idx % D);
  const double n = (double)(offsets[k + 1] - offsets[k]);
  const double w = 1.0 / n;
  const double ps = psi[d];
  const double c = n * ps / (1.0 + n * ps);
  const double p = P[idx];
  Y1[idx] = sqrt(w) * c * p;
  Y2[idx] = sqrt(w * n) * (1.0 - c) * p;
}

// db_d = sum_k w_k psi_d/(1+n_k psi_d) ; dw_d = sum_k w_k n_k psi_d/(1+n_k psi_d); added to diagonals
__global__ __launch_bounds__(256) void em_diag_kernel(const int *__restrict__ offsets,
                                                      const double *__restrict__ psi, int64_t K, int D,
                                                      double *__restrict__ Bt, double *__restrict__ Wt) {
  const int d = blockIdx.x;
  __shared__ double rb[256], rw[256];
  const double ps = psi[d];
  double ab = 0.0, aw = 0.0;
  for (int64_t k = threadIdx.x; k < K; k += blockDim.x) {
    const double n = (double)(offsets[k + 1] - offsets[k]);
    const double mx = ps / (1.0 + n * ps);
    ab += mx / n;
    aw += mx;
  }
  rb[threadIdx.x] = ab; rw[threadIdx.x] = aw;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { rb[threadIdx.x] += rb[threadIdx.x + o]; rw[threadIdx.x] += rw[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    Bt[(size_t)d * D + d] += rb[0];
    Wt[(size_t)d * D + d] += rw[0];
  }
}

// W = (S + Wu) / cntW ; B = Bu / cntB, symmetrised
__global__ void em_mstep_kernel(const double *__restrict__ S, const double *__restrict__ Wu,
                                const double *__restrict__ Bu, int D, double cntW, double cntB,
                                double *__restrict__ W, double *__restrict__ B) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D * D) return;
  const int i = idx / D, j = idx % D;
  const size_t a = (size_t)i * D + j, b = (size_t)j * D + i;
  W[a] = (0.5 * (S[a] + S[b]) + 0.5 * (Wu[a] + Wu[b])) / cntW;
  B[a] = 0.5 * (Bu[a] + Bu[b]) / cntB;
}

// ---- grouped closed-form EM (see fit_em_device) ----
__global__ void gather_center_kernel(const double *__restrict__ means, const double *__restrict__ mu,
                                     const int *__restrict__ cls, int64_t K, int D, double *__restrict__ out) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= K * D) return;
  const int64_t r = idx / D;
  const int d = (int)(idx % D);
  out[idx] = means[(int64_t)cls[r] * D + d] - mu[d];
}

// Grouped EM, MOMENT form (few groups of many classes; fit_em_device chooses).  Inside a group every class shares A_g = W + n_g B;
// with T_g its whitening factor (T_g A_g T_g^T = I, A_g^-1 = T_g^T T_g), X_g = T_g B and the group's second moments
// C_g = sum_k m_k m_k^T (computed once):
//   Q_g = B A_g^-1 = X_g^T T_g,   Q_g C_g = X_g^T (T_g C_g),   Q_g C_g Q_g^T = X_g^T (T_g C_g T_g^T) X_g,   Mx_g = B - n_g X_g^T X_g
//   W_stats = S + sum_g [ K_g Mx_g + C_g - n_g (QC_g + QC_g^T) + n_g^2 QCQ_g ],     B_stats = sum_g [ (K_g / n_g) Mx_g + n_g QCQ_g ]
// Seven D^3 products per group in FOUR dependent launches (X | T C  ->  X^T(T C) | (T C) T^T | X^T X  ->  (T C T^T) X  ->  X^T (..)),
// nothing K-sized.  Round 6: every product is a T-form -- Q_g is never formed, nothing is multiplied by an explicit A_g^-1 -- so
// the error is sqrt(cond(A_g)) eps, not cond(A_g) eps, and the refinement step of rounds 3-5 (Q += (B - Q A) A^-1: two more
// dependent launches, and a third for T^T T) is gone: 8 launches -> 6 per iteration, 106 -> ~95 us at D = 200, G = 1.  Against
// the x87 EM (N = 149, D = 200, cond(W) = 3e8) W 1.4e-14, B 2.5e-13 (with the refinement step: 3e-14, 2e-13; the reference's
// own formulation 4e-12, 6e-9).
// i <= j computes both (i, j) and (j, i), then symmetrises like Kaldi's CopyToSp.
__global__ void em_moment_mstep_kernel(const double *__restrict__ S, const double *__restrict__ Csum, const double *__restrict__ XtX,
                                       const double *__restrict__ QC, const double *__restrict__ QCQ, const double *__restrict__ gn,
                                       const double *__restrict__ gk, int G, int D, double sumK, double cw, double cntW, double cntB,
                                       double *__restrict__ W, double *__restrict__ B) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D * D) return;
  const int i = idx / D, j = idx % D;
  if (i > j) return;
  const size_t ij = (size_t)i * D + j, ji = (size_t)j * D + i, DD = (size_t)D * D;
  const double b_ij = B[ij], b_ji = B[ji];
  double wij = S[ij] + fma(sumK, b_ij, Csum[ij]), wji = S[ji] + fma(sumK, b_ji, Csum[ji]);
  double bij = cw * b_ij, bji = cw * b_ji;
  for (int g = 0; g < G; ++g) {
    const double n = gn[g], k = gk[g];
    const double *xx = XtX + g * DD, *qc = QC + g * DD, *qcq = QCQ + g * DD;
    const double cross = n * (qc[ij] + qc[ji]);
    wij += -k * n * xx[ij] - cross + n * n * qcq[ij];
    wji += -k * n * xx[ji] - cross + n * n * qcq[ji];
    bij += -k * xx[ij] + n * qcq[ij];
    bji += -k * xx[ji] + n * qcq[ji];
  }
  const double w = 0.5 * (wij / cntW + wji / cntW), b = 0.5 * (bij / cntB + bji / cntB);
  W[ij] = w; W[ji] = w;
  B[ij] = b; B[ji] = b;
}

// ------------------------------------------------------------------------------------
// Grouped EM in the ROW form (round 6; the default).  With T_g the whitening factor of A_g = W + n_g B (lower triangular,
// T_g A_g T_g^T = I, so A_g^-1 = T_g^T T_g) and X_g = T_g B:
//   posterior mean of class k of group g:  w_k = n_g y_k,  y_k = B A_g^-1 m_k = X_g^T (T_g m_k)          (rows: Y = (M T_g^T) X_g)
//   posterior covariance of the group:     Mx_g = (B^-1 + n_g W^-1)^-1 = B - n_g X_g^T X_g
//   W_stats = S + sum_k (m_k - w_k)(m_k - w_k)^T + sum_g K_g Mx_g   = S + Z^T Z   + K B       - sum_g K_g n_g X_g^T X_g
//   B_stats =     sum_k (1/n_k) w_k w_k^T + sum_g (K_g / n_g) Mx_g  =     Wn^T Wn + (sum_k 1/n_k) B - sum_g K_g X_g^T X_g
// with Z = M - n Y and Wn = sqrt(n) Y.  Per group this is ONE D^3 product (X_g) beside the factorisation; everything else is
// work on the K class means (two products per row tile) and two symmetric rank-k sums over the stacked rows [X_1; ..; X_G | Z]
// and [X_1; ..; X_G | Wn] -- where the moment form of rounds 2-5 (per-group second moments C_g; still the faster one for FEW groups of
// MANY classes, see fit_em_device) runs seven batched D^3 products per iteration.  Numerically the T-forms lose sqrt(cond(A_g)) where an explicit
// A_g^-1 loses cond(A_g): against the x87 EM (N = 149, D = 200, six iterations, cond(W) = 3e8) W 1.3e-14, B 2.6e-13 without any
// refinement step (rounds 2-5 with one: 3e-14, 2e-13; the reference's own formulation: 4e-12, 6e-9).
// ------------------------------------------------------------------------------------
typedef double f64x4_fit __attribute__((ext_vector_type(4)));

// T_g of the FIRST iteration: W = B = I there, so chol(W + n B)^-1 = I / sqrt(1 + n).  grid (ceil(D^2 / 256), groups)
__global__ void em_first_T_kernel(const double *__restrict__ gn, int D, int64_t stride, double *__restrict__ T) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D * D) return;
  T[(int64_t)blockIdx.y * stride + idx] = (idx / D == idx % D) ? 1.0 / sqrt(1.0 + gn[blockIdx.y]) : 0.0;
}

// row weights of the stacked X_g in the two rank-k sums: kw1 = -K_g n_g, kw2 = -K_g (row i of group g at g D + i)
__global__ void em_row_weights_kernel(const double *__restrict__ gn, const double *__restrict__ gk, int D, int64_t GD,
                                      double *__restrict__ kw1, double *__restrict__ kw2) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= GD) return;
  const int g = (int)(idx / D);
  kw1[idx] = -gk[g] * gn[g];
  kw2[idx] = -gk[g];
}

// RB stacked 16 x 16 blocks (16 RB rows of the left operand, one column block) of a product whose left operand (k-contiguous,
// leading dimension ld) sits in LDS and whose right operand comes from global memory through `bload(k)` (this lane's element for k
// index k + fk).  `a` = the lane's position (row fi, k offset fk) in the LDS tile; nch chunks of 16 k.  Per trip the 16 fragments of
// four chunks are requested, then multiplied; the other waves of the SIMD (eight with two workgroups per CU) cover the round trip.
//  * What bounds these kernels is the CU's vector-memory path, not latency: a fragment load is four 128-byte rows, ~20 cycles of
//    the L1, and with ONE MFMA per fragment the four SIMDs ask for one every 16 cycles.  RB = 2 halves that (two MFMAs per fragment).
//  * Software pipelining across trips did not survive the compiler (round 6; rotating fragment sets by moves, by renamed slots, by
//    fixed slots between scheduling barriers, and a compile-time recursion over the rounds): the register allocator copies the sets
//    at the loop's back edge and a copy waits for the load it copies, or the recursion spills.  All four measured slower than this.
//  * More fragments per trip do not help either (7 chunks = 28 loads in flight, two trips per block at D = 200 instead of four:
//    em_xtb_kernel 22.2 us and em_rows_kernel 33.3 us, unchanged): the trips' round trips are already covered.
//  * the loads must be BRANCH-FREE: a guard per load puts every load under its own exec branch with a vmcnt(0) behind it.
typedef unsigned u32x2_fit __attribute__((ext_vector_type(2)));
// (right operand: `rs` = a buffer resource over ONE D x D matrix, `voff` = this lane's byte offset (k offset fk, its column; a lane
//  whose column does not exist carries an offset past the matrix), rowbytes = 8 D.  The k of a fragment rides on the SCALAR offset:
//  no vector arithmetic per load -- as pointer arithmetic (64-bit multiply, clamp, add per load) the two kernels executed 6.7
//  vector instructions per MFMA (PMC, round 6) -- and what lies past the matrix reads as zero.)
template <int RB>
__device__ __forceinline__ void em_block_product(const double *a, int ld, int nch, const __amdgpu_buffer_rsrc_t rs, unsigned voff,
                                                 int rowbytes, f64x4_fit (&out)[RB]) {
  f64x4_fit acc[RB][2];
#pragma unroll
  for (int b = 0; b < RB; ++b) acc[b][0] = acc[b][1] = f64x4_fit{0.0, 0.0, 0.0, 0.0};
  const int last = 16 * (nch - 1);
  for (int c = 0; c < nch; c += 4) {
    double q[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int k0 = min(16 * (c + p), last);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        q[p][u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (k0 + 4 * u) * rowbytes, 0));
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (c + p < nch) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int b = 0; b < RB; ++b)
            acc[b][u & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[b * 16 * ld + 16 * (c + p) + 4 * u], q[p][u], acc[b][u & 1], 0, 0, 0);
      }
  }
#pragma unroll
  for (int b = 0; b < RB; ++b) out[b] = acc[b][0] + acc[b][1];
}

// X_g = T_g B and the transposed copy TT_g = T_g^T: workgroup (x, g) takes the 16 RB rows of T_g that END with block
// i = NT - 1 - RB x (the long rows first) -- T_g is lower triangular, so they end at k < 16 (i + 1) -- into LDS with
// row-contiguous loads, writes them out transposed, and multiplies: sixteen waves, a column block each, B's fragments straight
// from L2 (B is shared by every group).  The batched 16 x 16-tile kernel of linalg.hip read T_g's fragments as sixteen 32-byte
// pieces per load and the whole k extent: 34 us for 36 groups at D = 200.  D <= 512.
template <int RB>
__global__ __launch_bounds__(1024) void em_xtb_kernel(const double *__restrict__ T, const double *__restrict__ B, int D,
                                                      double *__restrict__ X, double *__restrict__ TT, int split) {
  extern __shared__ __attribute__((aligned(16))) double em_xtb_lds[];
  const int NT = (D + 15) >> 4, ld = 16 * NT + 2;
  // split == 2: blockIdx.x = 2 tile + half, the column blocks of a row tile are shared by TWO workgroups (the longest tile's 13
  // blocks on one workgroup's four SIMDs are 4 + 3 + 3 + 3: its 11 us set the kernel's time when few groups leave CUs idle)
  // (split = 2 when the whole launch is at most ~1.5 workgroups per CU -- 12 groups at D = 200: 20 -> 14 us; with more groups
  //  the chip is full anyway and the second staging of the tile costs more than the balance gains: 36 groups 22 -> 24 us)
  const int half = split == 2 ? (int)(blockIdx.x & 1) : 0, tile = split == 2 ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
  const int jlo = half ? (NT + 1) / 2 : 0, jhi = split == 2 && !half ? (NT + 1) / 2 : NT;
  const int i = NT - 1 - RB * tile, g = blockIdx.y, kext = 16 * (i + 1);
  const int m0 = 16 * (i - (RB - 1));              // (may be negative for the last workgroup of a group: those rows are skipped)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fi = lane & 15, fk = lane >> 4;
  const double *__restrict__ Tg = T + (int64_t)g * D * D;
  double *__restrict__ Xg = X + (int64_t)g * D * D, *__restrict__ TTg = TT + (int64_t)g * D * D;
  double *Ts = em_xtb_lds;
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(B), 0, D * D * 8, 0x00020000);
  for (int idx = t; idx < 16 * RB * kext; idx += 1024) {
    const int r = idx / kext, c = idx - r * kext, row = m0 + r;
    Ts[r * ld + c] = (row >= 0 && row < D && c < D) ? Tg[(int64_t)row * D + c] : 0.0;
  }
  __syncthreads();
  if (half == 0)
    for (int idx = t; idx < 16 * RB * kext; idx += 1024) {
      const int r = idx & (16 * RB - 1), c = idx / (16 * RB), row = m0 + r;
      if (row >= 0 && row < D && c < D) TTg[(int64_t)c * D + row] = Ts[r * ld + c];
    }
  for (int j = jlo + wave; j < jhi; j += 16) {
    const int col = 16 * j + fi;
    const bool okc = col < D;
    f64x4_fit x[RB];
    em_block_product<RB>(Ts + fi * ld + fk, ld, i + 1, rsB, okc ? (unsigned)(fk * D + col) * 8u : 0x7fffff00u, D * 8, x);
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 16 * b + fk + 4 * r;
        if (row >= 0 && row < D && okc) Xg[(int64_t)row * D + col] = x[b][r];
      }
  }
}

// The class means' share of an iteration: a workgroup takes 16 RB centred means of ONE group (tiles: {first row, rows, group}),
//   V = M T_g^T  (T_g lower triangular: column block j needs k < 16 (j + 1); read from the TRANSPOSED copy em_xtb_kernel leaves,
//                 so that a fragment is four 128-byte rows like X_g's and not sixteen 32-byte pieces)          ->  LDS
//   Y = V X_g,   Z = M - n_g Y,   Wn = sqrt(n_g) Y                                                          ->  global
// Sixteen waves, a column block each in both phases.  LDS: two [16 RB][16 NT + 2] tiles (pitch = 2 mod 4 doubles: a fragment read -- 16 rows x
// 2 k per 32-lane group of a ds_read_b64 -- then covers all 64 banks once; with + 4 the rows 8 apart shared banks and 72 % of the
// kernel's LDS cycles were conflicts, PMC round 6): RB = 2 for D <= 256, RB = 1 up to D = 512.
template <int RB>
__global__ __launch_bounds__(1024) void em_rows_kernel(const double *__restrict__ Mg, const int4 *__restrict__ tiles,
                                                       const double *__restrict__ T /*transposed: [k][j]*/, const double *__restrict__ X,
                                                       const double *__restrict__ gn, int D, double *__restrict__ Z,
                                                       double *__restrict__ Wn) {
  extern __shared__ __attribute__((aligned(16))) double em_rows_lds[];
  const int NT = (D + 15) >> 4, ld = 16 * NT + 2;
  double *Ms = em_rows_lds, *Vs = Ms + 16 * RB * ld;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fi = lane & 15, fk = lane >> 4;
  const int4 tile = tiles[blockIdx.x];
  const int row0 = tile.x, nrows = tile.y, g = tile.z;
  const double n = gn[g];
  const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(T + (int64_t)g * D * D), 0, D * D * 8, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(X + (int64_t)g * D * D), 0, D * D * 8, 0x00020000);
  for (int idx = t; idx < 16 * RB * 16 * NT; idx += 1024) {
    const int r = idx / (16 * NT), c = idx - r * 16 * NT;
    Ms[r * ld + c] = (r < nrows && c < D) ? Mg[(int64_t)(row0 + r) * D + c] : 0.0;
  }
  __syncthreads();
  for (int j = NT - 1 - wave; j >= 0; j -= 16) {       // (the long blocks on the low waves)
    const int col = 16 * j + fi;
    const bool okc = col < D;
    f64x4_fit v[RB];
    em_block_product<RB>(Ms + fi * ld + fk, ld, j + 1, rsT, okc ? (unsigned)(fk * D + col) * 8u : 0x7fffff00u, D * 8, v);
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) Vs[(16 * b + fk + 4 * r) * ld + col] = okc ? v[b][r] : 0.0;   // (columns past D: zeros for phase 2's k extent)
  }
  __syncthreads();
  const double sn = sqrt(n);
  for (int j = wave; j < NT; j += 16) {
    const int col = 16 * j + fi;
    const bool okc = col < D;
    f64x4_fit y[RB];
    em_block_product<RB>(Vs + fi * ld + fk, ld, NT, rsX, okc ? (unsigned)(fk * D + col) * 8u : 0x7fffff00u, D * 8, y);
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * b + fk + 4 * r;
        if (row < nrows && okc) {
          const int64_t o = (int64_t)(row0 + row) * D + col;
          Z[o] = fma(-n, y[b][r], Ms[row * ld + col]);
          Wn[o] = sn * y[b][r];
        }
      }
  }
}

// D > 512 (the row tiles above do not fit in LDS): Y from two plain products per group, then Z = M - n Y and Wn = sqrt(n) Y
// here; n of sorted row r = counts[cls[r]].  Y arrives in Z.
__global__ void em_rows_finish_kernel(const double *__restrict__ Mg, const int *__restrict__ cls, const int64_t *__restrict__ counts,
                                      int64_t K, int D, double *__restrict__ Z, double *__restrict__ Wn) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= K * D) return;
  const double n = (double)counts[cls[idx / D]], y = Z[idx];
  Z[idx] = fma(-n, y, Mg[idx]);
  Wn[idx] = sqrt(n) * y;
}

// W = (S + K B + P1) / cntW,  B = (cw B + P2) / cntB  (cw = sum_k 1/n_k), symmetrised like Kaldi's CopyToSp
__global__ void em_rows_mstep_kernel(const double *__restrict__ S, const double *__restrict__ P1, const double *__restrict__ P2,
                                     int D, double sumK, double cw, double cntW, double cntB, double *__restrict__ W,
                                     double *__restrict__ B) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D * D) return;
  const int i = idx / D, j = idx % D;
  if (i > j) return;
  const size_t ij = (size_t)i * D + j, ji = (size_t)j * D + i;
  const double bij = B[ij], bji = B[ji];
  const double wij = S[ij] + fma(sumK, bij, P1[ij]), wji = S[ji] + fma(sumK, bji, P1[ji]);
  const double vij = fma(cw, bij, P2[ij]), vji = fma(cw, bji, P2[ji]);
  const double w = 0.5 * (wij / cntW + wji / cntW), b = 0.5 * (vij / cntB + vji / cntB);
  W[ij] = w; W[ji] = w;
  B[ij] = b; B[ji] = b;
}

__global__ void set_identity2_kernel(double *W, double *B, int D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < D * D) { const double v = (idx / D == idx % D) ? 1.0 : 0.0; W[idx] = v; B[idx] = v; }
}

// offset = -T mean (Plda::ComputeDerivedVars), one wave per output row
__global__ void offset_kernel(const double *__restrict__ T, const double *__restrict__ mean, int Dout, int Din,
                              double *__restrict__ offset) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (o >= Dout) return;
  double acc = 0.0;
  for (int d = lane; d < Din; d += 64) acc += T[(size_t)o * Din + d] * mean[d];
  for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
  if (lane == 0) offset[o] = -acc;
}

__global__ void counts_to_i64_kernel(const int *__restrict__ offsets, int64_t K, int64_t *__restrict__ counts) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K) counts[k] = offsets[k + 1] - offsets[k];
}

__global__ void counts_to_i32_kernel(const int *__restrict__ offsets, int64_t K, int32_t *__restrict__ counts) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K) counts[k] = offsets[k + 1] - offsets[k];
}

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// The host mirror of a fitted model in ONE launch: mean | transform | psi | offset and the EM's factorisation flag,
// written by the kernel itself into mapped pinned host memory (round 4).  Four hipMemcpyAsync to the host were four
// blit launches with 25-45 us of host round trip between them -- 120 us behind GetOutput's last kernel at D = 200.
__global__ void export_model_kernel(const double *__restrict__ mean, const double *__restrict__ T, const double *__restrict__ psi,
                                    const double *__restrict__ off, int D, const int *__restrict__ em_flag,
                                    const int *__restrict__ chol_flag, const int *__restrict__ eig_flag, double *__restrict__ out) {
  const size_t DD = (size_t)D * D, total = 3 * (size_t)D + DD;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  double v;
  if (e < (size_t)D) v = mean[e];
  else if (e < D + DD) v = T[e - D];
  else if (e < 2 * (size_t)D + DD) v = psi[e - D - DD];
  else v = off[e - 2 * (size_t)D - DD];
  out[e] = v;
  if (e == 0) {      // flags: the EM's factorisations, GetOutput's Cholesky, the eigensolver's status (8: not taken)
    int *f = reinterpret_cast<int *>(out + total);
    f[0] = *em_flag;
    f[1] = chol_flag ? *chol_flag : 0;
    f[2] = eig_flag ? *eig_flag : 8;
  }
}

int compute_offset_device(plda_handle *h) {
  offset_kernel<<<(unsigned)ceil_div(h->Dout, 4), 256, 0, h->stream>>>(
      h->d_transform.as<double>(), h->d_mean.as<double>(), h->Dout, h->Din, h->d_offset.as<double>());
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

// Sort rows by label: outputs perm (row ids grouped by label, ascending row within label)
// and offsets[K+1] in h->w[0], h->w[1].
// defer_bad != nullptr: the range check is NOT read back here; *defer_bad receives the device flag (bit 0: a label
// >= K; dense_check_kernel adds bit 1: an unused label, first one in [1]) for the caller to read with its own
// synchronisation.  Everything enqueued after a failed check works on valid indices (garbage values only).
__global__ void dense_check_kernel(const int *__restrict__ offsets, int64_t K, int *__restrict__ bad) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K && offsets[k + 1] == offsets[k]) {
    atomicOr(bad, 2);
    atomicMin(bad + 1, (int)k);
  }
}

// ------------------------------------------------------------------------------------
// The same grouping without a sort, for dense labels with K <= 32768 (round 3): the position of row r is
//   offsets[l] + (rows of label l in the 1024-row chunks before r's) + (rows of label l before r inside its chunk),
// all three from counting.  group_count_kernel: counts[l] and cnt[chunk][l] by atomics (sums: order does not matter);
// scan: offsets; group_base_kernel: one thread per label walks down the chunks (cnt -> exclusive prefix + offsets[l])
// and does the dense check; group_place_kernel: one WAVE per chunk ranks its rows in row order -- 64 at a time, the
// lanes with the same label found by ballots over the label's bits, their common running count kept in the wave's own
// LDS table (LDS operations of a wave execute in order) -- and writes perm.  Five launches and two memsets instead of
// the two radix passes' eleven: 88 -> ~40 us at C2.  The result is the radix sort's, bit for bit.
// ------------------------------------------------------------------------------------
constexpr int GR_CHUNK = 1024;     // rows per chunk (one wave)
constexpr int GR_KMAX = 32768;     // labels per LDS table: 128 KiB hold 4 / 2 / 1 waves' tables at K <= 8192 / 16384 / 32768

__global__ __launch_bounds__(256) void group_count_kernel(const uint64_t *__restrict__ labels, int64_t N, int64_t K,
                                                          int *__restrict__ counts, int *__restrict__ cnt,
                                                          int *__restrict__ bad) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  uint64_t l = labels[r];
  if (l >= (uint64_t)K) { *bad = 1; l = 0; }          // (counted under label 0: every index downstream stays valid)
  atomicAdd(counts + l, 1);
  atomicAdd(cnt + (r / GR_CHUNK) * K + (int64_t)l, 1);
}

__global__ __launch_bounds__(256) void group_base_kernel(const int *__restrict__ offsets, int64_t K, int64_t nchunks,
                                                         int *__restrict__ cnt, int *__restrict__ bad) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= K) return;
  int run = offsets[l];
  if (offsets[l + 1] == run) {
    atomicOr(bad, 2);
    atomicMin(bad + 1, (int)l);
  }
  int64_t c = 0;
  for (; c + 8 <= nchunks; c += 8) {
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = cnt[(c + u) * K + l];
#pragma unroll
    for (int u = 0; u < 8; ++u) { cnt[(c + u) * K + l] = run; run += v[u]; }
  }
  for (; c < nchunks; ++c) { const int v = cnt[c * K + l]; cnt[c * K + l] = run; run += v; }
}

__global__ __launch_bounds__(256) void group_place_kernel(const uint64_t *__restrict__ labels, int64_t N, int64_t K,
                                                          int nbits, const int *__restrict__ base, int64_t nchunks,
                                                          uint32_t *__restrict__ perm) {
  extern __shared__ int gr_table[];                   // [waves of the workgroup][K]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int *tab = gr_table + (size_t)wave * K;
  for (int64_t q = lane; q < K; q += 64) tab[q] = 0;
  const int64_t chunk = (int64_t)blockIdx.x * nw + wave;
  if (chunk >= nchunks) return;
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll 4
  for (int it = 0; it < GR_CHUNK / 64; ++it) {
    const int64_t r = chunk * GR_CHUNK + it * 64 + lane;
    const bool valid = r < N;
    uint64_t l64 = valid ? labels[r] : 0;
    if (l64 >= (uint64_t)K) l64 = 0;
    const int l = (int)l64;
    unsigned long long peers = __ballot(valid);
    for (int bit = 0; bit < nbits; ++bit) {
      const bool one = (l >> bit) & 1;
      const unsigned long long m = __ballot(one);
      peers &= one ? m : ~m;
    }
    const int rank = __popcll(peers & lt);
    int before = 0;
    if (valid && rank == 0) before = atomicAdd(tab + l, __popcll(peers));     // (LDS: in order inside the wave)
    before = __shfl(before, valid ? (int)__builtin_ctzll(peers) : 0);
    if (valid) perm[base[chunk * K + l] + before + rank] = (uint32_t)r;
  }
}

static int group_by_counting(plda_handle *h, const uint64_t *dlabels, int64_t N, int64_t K, uint32_t **perm_out,
                             int **offsets_out, int **defer_bad) {
  const int64_t nchunks = ceil_div(N, (int64_t)GR_CHUNK);
  PLDA_HIP(h, h->w[0].reserve((size_t)N * 4));                              // perm
  PLDA_HIP(h, h->w[1].reserve((size_t)(K + 3) * 4 + 64));                   // counts -> offsets (+ bad flag, first unused label)
  PLDA_HIP(h, h->w[2].reserve((size_t)nchunks * K * 4));                    // per-chunk counts -> bases
  uint32_t *perm = h->w[0].as<uint32_t>();
  int *offsets = h->w[1].as<int>(), *bad = offsets + K + 1, *cnt = h->w[2].as<int>();
  PLDA_HIP(h, hipMemsetAsync(offsets, 0, (size_t)(K + 2) * 4, h->stream));
  PLDA_HIP(h, hipMemsetAsync(bad + 1, 0x7f, 4, h->stream));
  PLDA_HIP(h, hipMemsetAsync(cnt, 0, (size_t)nchunks * K * 4, h->stream));
  group_count_kernel<<<(unsigned)ceil_div(N, 256), 256, 0, h->stream>>>(dlabels, N, K, offsets, cnt, bad);
  scan_kernel<<<1, 1024, 0, h->stream>>>(offsets, K + 1);
  group_base_kernel<<<(unsigned)ceil_div(K, 256), 256, 0, h->stream>>>(offsets, K, nchunks, cnt, bad);
  int nbits = 1;
  while ((1ll << nbits) < K) nbits++;
  const int nw = K <= 8192 ? 4 : K <= 16384 ? 2 : 1;          // waves (chunks) per workgroup: their tables share 128 KiB
  const size_t lds = (size_t)nw * K * sizeof(int);
  static DeviceOnce attr;          // (per device: handles on different GPUs share this function)
  if (attr.needed(h->device)) {
    PLDA_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&group_place_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, GR_KMAX * (int)sizeof(int)));
    attr.done(h->device);
  }
  group_place_kernel<<<(unsigned)ceil_div(nchunks, (int64_t)nw), 64 * nw, lds, h->stream>>>(dlabels, N, K, nbits, cnt, nchunks, perm);
  PLDA_LAUNCH_CHECK(h);
  *defer_bad = bad;
  *perm_out = perm;
  *offsets_out = offsets;
  return PLDA_OK;
}

static int sort_by_label(plda_handle *h, const uint64_t *dlabels, int64_t N, int64_t K, uint32_t **perm_out,
                         int **offsets_out, int **defer_bad = nullptr) {
  if (N >= (1ll << 31)) return fail(h, PLDA_E_INVAL, "fit: N too large");
  // counting instead of sorting where its tables fit (PLDA_SORT_VARIANT=1: the radix sort always)
  if (defer_bad && K <= GR_KMAX && ceil_div(N, (int64_t)GR_CHUNK) * K <= (64ll << 20) && h->sort_variant == 0)
    return group_by_counting(h, dlabels, N, K, perm_out, offsets_out, defer_bad);
  const int nblocks = (int)ceil_div(N, RS_CHUNK);
  PLDA_HIP(h, h->w[0].reserve((size_t)N * 4 * 4));                  // keys a/b, vals a/b
  PLDA_HIP(h, h->w[1].reserve((size_t)(K + 3) * 4 + 64));          // counts -> offsets (+ bad flag, first unused label)
  PLDA_HIP(h, h->w[2].reserve((size_t)256 * nblocks * 4));          // digit histograms
  uint32_t *ka = h->w[0].as<uint32_t>(), *va = ka + N, *kb = va + N, *vb = kb + N;
  int *offsets = h->w[1].as<int>();
  int *bad = offsets + K + 1;
  int *hist = h->w[2].as<int>();
  PLDA_HIP(h, hipMemsetAsync(offsets, 0, (size_t)(K + 2) * 4, h->stream));
  PLDA_HIP(h, hipMemsetAsync(bad + 1, 0x7f, 4, h->stream));
  labels_check_kernel<<<(unsigned)ceil_div(N, 256), 256, 0, h->stream>>>(dlabels, N, K, ka, va, offsets, bad);
  PLDA_LAUNCH_CHECK(h);
  int bits = 1;
  while ((1ll << bits) < K) bits++;
  for (int shift = 0; shift < bits; shift += 8) {
    rs_hist_kernel<<<nblocks, RS_THREADS, 0, h->stream>>>(ka, N, shift, nblocks, hist);
    scan_kernel<<<1, 1024, 0, h->stream>>>(hist, (int64_t)256 * nblocks);
    rs_scatter_kernel<<<nblocks, RS_THREADS, 0, h->stream>>>(ka, va, kb, vb, N, shift, nblocks, hist);
    PLDA_LAUNCH_CHECK(h);
    std::swap(ka, kb);
    std::swap(va, vb);
  }
  // counts -> exclusive offsets (K+1 entries: the (K+1)-th input is 0 so offsets[K] = N)
  scan_kernel<<<1, 1024, 0, h->stream>>>(offsets, K + 1);
  PLDA_LAUNCH_CHECK(h);
  if (defer_bad) {
    dense_check_kernel<<<(unsigned)ceil_div(K, 256), 256, 0, h->stream>>>(offsets, K, bad);
    PLDA_LAUNCH_CHECK(h);
    *defer_bad = bad;
  } else {
    int hbad = 0;
    PLDA_HIP(h, hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    if (hbad) return fail(h, PLDA_E_LABELS, "fit: labels must be dense 0..K-1");
  }
  *perm_out = va;
  *offsets_out = offsets;
  return PLDA_OK;
}

// Statistics pass (pldamodule.cpp:76-100): leaves means[K,D], counts[K] and the offset scatter of THESE
// classes in the handle.  The scatter, the weighted class sum and the class weight are all additive over
// disjoint sets of speakers, which is what lets the pass shard by speaker.
// defer_check (plda_fit: the EM follows at once): the label checks are NOT read back here -- fit_em_device reads them with
// the one host round trip its planning needs anyway, and takes the pass's time from events.
int fit_stats_device(plda_handle *h, const double *dX, int64_t N, int D, const uint64_t *dlabels, int64_t K, bool defer_check) {
  if (!dX || !dlabels || N <= 0 || D <= 0) return fail(h, PLDA_E_INVAL, "fit: bad argument");
  if (K <= 0 || K > N) return fail(h, PLDA_E_LABELS, "fit: labels must be dense 0..K-1");
  if (D > 2048) return fail(h, PLDA_E_INVAL, "fit: featdim %d > 2048 unsupported", D);
  const size_t DD = (size_t)D * D;
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  const double t0 = now_ms();
  h->fit_dbad = nullptr;
  if (defer_check) {
    for (hipEvent_t &e : h->fit_ev)
      if (!e) PLDA_HIP(h, hipEventCreate(&e));
    PLDA_HIP(h, hipEventRecord(h->fit_ev[2], h->stream));
  }

  // ---------------- statistics (K1a, K1, K2) ----------------
  uint32_t *perm = nullptr;
  int *offsets = nullptr, *dbad = nullptr;
  {
    TraceScope ts(h, "fit.label_sort (K1a)", (double)N * 8.0, 2);
    PLDA_TRY(sort_by_label(h, dlabels, N, K, &perm, &offsets, &dbad));
  }
  PLDA_HIP(h, h->f_means.reserve((size_t)K * D * 8));
  PLDA_HIP(h, h->f_counts.reserve((size_t)K * 8));
  PLDA_HIP(h, h->f_scatter.reserve(DD * 8));
  PLDA_HIP(h, h->w[3].reserve((size_t)N * 8));               // row weights
  double *means = h->f_means.as<double>();
  double *S = h->f_scatter.as<double>();
  double *roww = h->w[3].as<double>();
  counts_to_i64_kernel<<<(unsigned)ceil_div(K, 256), 256, 0, h->stream>>>(offsets, K, h->f_counts.as<int64_t>());
  {
    TraceScope ts(h, "fit.centroids (K1)", (double)N * D * 8.0, 2);
    centroid_kernel<<<(unsigned)K, 256, 0, h->stream>>>(dX, D, perm, offsets, means, roww);
    PLDA_LAUNCH_CHECK(h);
  }
  // offset_scatter = X^T diag(1/n_label) X - sum_k (n_k w_k) m_k m_k^T,  n_k w_k = 1
  {
    // X^T diag(1 / n_label) X and - M^T M in one pass (D <= 208: one launch + one reduction; the centroids' term was a
    // launch pair of its own, 48 us at 0.10 of the fp64 peak at C2)
    // work = the flop of the lower TRIANGLE, (N + K) D (D + 1) ("N D (D+1) if only the triangle"):
    // what the kernels execute.  (Rounds 1-3 credited the full 2 N D^2 here, which let the reported fraction pass 1.)
    TraceScope ts(h, "fit.scatter_syrk (K2)", (double)(N + K) * D * (D + 1.0), 1);
    PLDA_TRY(syrk_pair_f64(h, D, N, dX, D, roww, K, means, D, -1.0, S, D));
  }
  // the label checks are read back only now, with the synchronisation the pass ends on anyway: a failed check
  // leaves garbage values (never an invalid index) in what was enqueued after it
  h->fit_K = K; h->fit_D = D;
  h->fit_ms[0] = h->fit_ms[1] = h->fit_ms[2] = h->fit_ms[3] = 0.0;
  if (defer_check) {       // no host round trip: the EM's planning copy brings the flags back (35 us of idle GPU less)
    h->fit_dbad = dbad;
    h->fit_t0 = t0;
    return PLDA_OK;
  }
  int hbad[2] = {0, 0};
  PLDA_HIP(h, hipMemcpyAsync(hbad, dbad, 8, hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  if (hbad[0] & 1) return fail(h, PLDA_E_LABELS, "fit: labels must be dense 0..K-1");
  if (hbad[0] & 2) return fail(h, PLDA_E_LABELS, "fit: labels must be dense 0..K-1 (label %d unused)", hbad[1]);
  h->fit_ms[0] = now_ms() - t0;
  return PLDA_OK;
}

__global__ void counts_to_offsets_kernel(const int64_t *__restrict__ counts, int64_t K, int *__restrict__ offsets,
                                         int *__restrict__ bad) {
  const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (k > K) return;
  int v = 0;
  if (k < K) {
    const int64_t c = counts[k];
    if (c <= 0 || c > 0x7fffffff) atomicOr(bad, 1);
    v = (int)c;
  }
  offsets[k] = v;
}

// EM + GetOutput (pldamodule.cpp:102-106) from the statistics held by the handle: means[K,D], counts[K],
// offset scatter.  Everything else the estimator needs (sum_, class_weight, the global mean) follows from
// the means and counts, so this is also the replica step after a sharded statistics pass.
int fit_em_device(plda_handle *h, int64_t K, int D, int iters) {
  // (taken over -- and cleared -- before ANY return: a deferred label check left behind would send a later, unrelated
  //  plda_fit_em_dev down the no-synchronisation path with a stale pointer and start time; round-4 advisor)
  int *const stats_bad = h->fit_dbad;      // a statistics pass of this same plda_fit is still on the stream: no synchronisation,
  h->fit_dbad = nullptr;                   // its label checks come back with the planning copies below
  if (K <= 0 || D <= 0 || iters < 0) return fail(h, PLDA_E_INVAL, "fit: bad argument");
  if (K == 1)
    return fail(h, PLDA_E_ONE_SPEAKER,
                "Number of speakers is 1. Aborting PLDA esimation, at least two speakers are required!");
  const size_t DD = (size_t)D * D;
  h->simdiag_has_vr = false;   // a new fit starts cold
  h->jac_total_sweeps = 0;
  if (!stats_bad) PLDA_HIP(h, hipStreamSynchronize(h->stream));
  const double t1 = stats_bad ? h->fit_t0 : now_ms();
  // EM | GetOutput boundary of plda_fit_timings: a pair of events instead of a host synchronisation, so that GetOutput is
  // enqueued while the EM still runs (the synchronisation left the GPU idle for the host's launch latency, 30-50 us)
  for (hipEvent_t &e : h->fit_ev)
    if (!e) PLDA_HIP(h, hipEventCreate(&e));
  PLDA_HIP(h, hipEventRecord(h->fit_ev[0], h->stream));
  // pinned landing area of everything this call reads back: the model (mean | transform | psi | offset) and the EM's
  // factorisation flag -- copies into it are queued without blocking the host, one synchronisation ends the fit
  // (+ the EM's planning traffic: class counts, count check and class weight coming back, the classes' order by count and
  //  the groups' counts going out -- pageable, each of those six copies was a blocking 20-40 us)
  const size_t pin_model_bytes = (3 * (size_t)D + DD) * 8 + 64;
  const size_t pin_need = pin_model_bytes + (size_t)K * (8 + 4 + 8 + 8 + 16) + 96 + 16;
  if (h->pin_model_cap < pin_need) {
    if (h->pin_model) (void)hipHostFree(h->pin_model);
    h->pin_model = nullptr; h->pin_model_cap = 0;
    PLDA_HIP(h, hipHostMalloc(&h->pin_model, pin_need, hipHostMallocMapped));
    h->pin_model_cap = pin_need;
  }
  double *const pm = static_cast<double *>(h->pin_model);
  int *const em_flag_host = reinterpret_cast<int *>(pm + 3 * (size_t)D + DD);
  *em_flag_host = 0;
  PLDA_HIP(h, h->fit_flag.reserve(64));
  PLDA_HIP(h, hipMemsetAsync(h->fit_flag.p, 0, 4, h->stream));
  TraceScope ts_em(h, "fit.em (all iterations)");
  PLDA_HIP(h, h->f_sum.reserve((size_t)D * 8));
  PLDA_HIP(h, h->f_W.reserve(DD * 8));
  PLDA_HIP(h, h->f_B.reserve(DD * 8));
  PLDA_HIP(h, h->w[1].reserve((size_t)(K + 2) * 4 + 64));
  PLDA_HIP(h, h->w[4].reserve((size_t)D * 8 * 2 + 64));       // mu, scalars
  double *means = h->f_means.as<double>();
  double *S = h->f_scatter.as<double>();
  double *sum = h->f_sum.as<double>();
  double *W = h->f_W.as<double>(), *B = h->f_B.as<double>();
  double *mu = h->w[4].as<double>();
  double *scalars = mu + D;
  int *offsets = h->w[1].as<int>();
  char *const pin_plan = static_cast<char *>(h->pin_model) + pin_model_bytes;
  int64_t *const hcounts = reinterpret_cast<int64_t *>(pin_plan);                      // [K]
  double *const pin_gn = reinterpret_cast<double *>(pin_plan + (size_t)K * 8);         // [<= K]
  double *const pin_gk = pin_gn + K;                                                   // [<= K]
  double *const pin_cw = pin_gk + K;                                                   // class weight, then the count check,
  int *const pin_bad = reinterpret_cast<int *>(pin_cw + 1);                            // then the statistics pass's two label words
  int *const pin_lab = reinterpret_cast<int *>(pin_cw + 2);
  int *const pin_cls = reinterpret_cast<int *>(pin_cw + 3);                            // [K]
  int4 *const pin_tiles = reinterpret_cast<int4 *>((reinterpret_cast<uintptr_t>(pin_cls + K) + 15) & ~(uintptr_t)15);   // [<= K] row tiles of the EM
  pin_lab[0] = pin_lab[1] = 0;
  // one host round trip for everything the group planning needs: the count check, the class weight and the counts
  int *bad = offsets + K + 1;
  int hbad = 0;
  // (the statistics pass kept its label words in this same buffer: out before the count check's word is cleared)
  if (stats_bad) PLDA_HIP(h, hipMemcpyAsync(pin_lab, stats_bad, 8, hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipMemsetAsync(bad, 0, 4, h->stream));
  counts_to_offsets_kernel<<<(unsigned)ceil_div(K + 1, 256), 256, 0, h->stream>>>(h->f_counts.as<int64_t>(), K,
                                                                                offsets, bad);
  scan_kernel<<<1, 1024, 0, h->stream>>>(offsets, K + 1);
  PLDA_LAUNCH_CHECK(h);
  PLDA_HIP(h, h->w[7].reserve((size_t)CS_SPLIT * (D + 1) * 8));
  {
    double *partial = h->w[7].as<double>(), *wpart = partial + (size_t)CS_SPLIT * D;
    class_sum_partial_kernel<<<dim3((unsigned)ceil_div(D, 64), CS_SPLIT), 256, 0, h->stream>>>(means, offsets, K, D,
                                                                                              partial, wpart);
    class_sum_final_kernel<<<(unsigned)ceil_div(D, 256), 256, 0, h->stream>>>(partial, wpart, D, sum, mu, scalars);
  }
  PLDA_LAUNCH_CHECK(h);
  PLDA_HIP(h, hipMemcpyAsync(pin_bad, bad, 4, hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipMemcpyAsync(pin_cw, scalars, 8, hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipMemcpyAsync(hcounts, h->f_counts.p, (size_t)K * 8, hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  if (pin_lab[0] & 1) return fail(h, PLDA_E_LABELS, "fit: labels must be dense 0..K-1");
  if (pin_lab[0] & 2) return fail(h, PLDA_E_LABELS, "fit: labels must be dense 0..K-1 (label %d unused)", pin_lab[1]);
  hbad = *pin_bad;
  const double class_weight = *pin_cw;
  if (hbad) return fail(h, PLDA_E_INVAL, "fit: class counts must be positive");
  const double example_weight = (double)K;  // sum_k w_k n_k with w_k = 1/n_k

  // ---------------- EM (K3) ----------------
  // Group the classes by their count n.  Inside one group every class shares
  //   A = W + nB,  mixed = (B^-1 + n W^-1)^-1 = W A^-1 B,  w_k = mixed n W^-1 m_k = n Q m_k,  Q = B A^-1,
  // so the sums over the classes of the group collapse onto C_g = sum_k m_k m_k^T, which never changes:
  //   sum_k w_k w_k^T = n^2 Q C_g Q^T,   sum_k (m_k - w_k)(m_k - w_k)^T = C_g - n(Q C_g + (Q C_g)^T) + n^2 Q C_g Q^T.
  // One iteration is then D x D work only -- per group one Cholesky, one triangular inverse and five
  // GEMMs, all batched over the groups -- with no inverse of W or B (B may be singular) and no
  // eigendecomposition.  Same estimator as SURVEY.md A.2, different association of the sums.
  int *const cls = pin_cls;
  {
    // classes in ascending order of their count, ties in class order: by counting when the counts span a small range (a
    // comparison sort of 5 000 classes through an index array was 100+ us of host time with the GPU idle), else by sorting
    int64_t cmin = hcounts[0], cmax = hcounts[0];
    for (int64_t k = 1; k < K; ++k) { cmin = std::min(cmin, hcounts[k]); cmax = std::max(cmax, hcounts[k]); }
    const int64_t span = cmax - cmin + 1;
    if (span <= 65536) {
      std::vector<int> start((size_t)span + 1, 0);
      for (int64_t k = 0; k < K; ++k) ++start[(size_t)(hcounts[k] - cmin) + 1];
      for (int64_t v = 0; v < span; ++v) start[(size_t)v + 1] += start[(size_t)v];
      for (int64_t k = 0; k < K; ++k) cls[start[(size_t)(hcounts[k] - cmin)]++] = (int)k;
    } else {
      for (int64_t k = 0; k < K; ++k) cls[k] = (int)k;
      std::stable_sort(cls, cls + K, [&](int a, int b) { return hcounts[a] < hcounts[b]; });
    }
  }
  std::vector<int64_t> goff;   // group g = sorted positions goff[g] .. goff[g+1]
  std::vector<double> gn, gk;
  for (int64_t r = 0; r < K; ++r)
    if (r == 0 || hcounts[cls[r]] != hcounts[cls[r - 1]]) { goff.push_back(r); gn.push_back((double)hcounts[cls[r]]); }
  goff.push_back(K);
  const int G = (int)gn.size();
  for (int g = 0; g < G; ++g) gk.push_back((double)(goff[g + 1] - goff[g]));
  const double cntW = (example_weight - class_weight) + class_weight;  // = K
  const double cntB = class_weight;
  const unsigned gDD = (unsigned)ceil_div((int64_t)DD, 256);
  const unsigned gKD = (unsigned)ceil_div(K * (int64_t)D, 256);
  set_identity2_kernel<<<gDD, 256, 0, h->stream>>>(W, B, D);
  PLDA_LAUNCH_CHECK(h);
  const size_t group_bytes = (size_t)G * DD * 8 * 9;      // (the moment form's nine matrices per group; the row form takes five)
  const bool grouped = h->em_variant != 1 && group_bytes <= ((size_t)24 << 30) && G <= 16384;
  h->em_groups = grouped ? G : 0;
  // Two closed forms of the grouped EM.  The moment form (rounds 2-5) runs seven D^3 products per group and iteration on per-group
  // second moments and never touches the K means again; the row form (round 6) runs one D^3 product per group and works on the K
  // means every iteration.  At D = 200: moment ~ 98 + 8 G us, row ~ 112 + 0.0096 K + 0.64 G us per iteration -- the row form when
  // there are enough groups for the means they stand for (PLDA_EM_VARIANT=3 / 4 force the moment / the row form).
  const bool row_form = h->em_variant == 4 || (h->em_variant == 0 && G >= 4 && 4 * (int64_t)G * D >= K);
  h->em_form = !grouped ? 0 : row_form ? 2 : 1;
  if (grouped && row_form) {
    // ---- row form (the kernels' header above) ----
    const int64_t sDD = (int64_t)DD, GD = (int64_t)G * D;
    const int RB = D <= 256 ? 2 : 1;    // row blocks per tile of em_rows_kernel / em_xtb_kernel (LDS)
    size_t ntiles = 0;                  // (at most K: every tile holds a row)
    for (int g = 0; g < G; ++g)
      for (int64_t r = goff[g]; r < goff[g + 1]; r += 16 * RB)
        pin_tiles[ntiles++] = make_int4((int)r, (int)std::min<int64_t>(16 * RB, goff[g + 1] - r), g, 0);
    PLDA_HIP(h, h->w[5].reserve((size_t)K * D * 8 * 3 + (size_t)K * 4 + ntiles * 16 + 64));
    PLDA_HIP(h, h->w[6].reserve((size_t)G * DD * 8 * 5 + DD * 8 * 3 + (size_t)G * 16 + (size_t)GD * 16 + 64));
    double *Mg = h->w[5].as<double>(), *Zr = Mg + (size_t)K * D, *Wn = Zr + (size_t)K * D;
    int4 *dtiles = reinterpret_cast<int4 *>(Wn + (size_t)K * D);
    int *dcls = reinterpret_cast<int *>(dtiles + ntiles);
    double *Tg = h->w[6].as<double>(), *Xg = Tg + (size_t)G * DD, *scr = Xg + (size_t)G * DD, *P1 = scr + 3 * (size_t)G * DD,
           *P2 = P1 + DD, *Balt = P2 + DD, *dgn = Balt + DD, *dgk = dgn + G, *kw1 = dgk + G, *kw2 = kw1 + GD;
    int *dflag = h->fit_flag.as<int>();          // (its own buffer: the export kernel that ends the fit reads it)
    std::copy(gn.begin(), gn.end(), pin_gn);
    std::copy(gk.begin(), gk.end(), pin_gk);
    PLDA_HIP(h, hipMemcpyAsync(dcls, cls, (size_t)K * 4, hipMemcpyHostToDevice, h->stream));
    PLDA_HIP(h, hipMemcpyAsync(dgn, pin_gn, (size_t)G * 8, hipMemcpyHostToDevice, h->stream));
    PLDA_HIP(h, hipMemcpyAsync(dgk, pin_gk, (size_t)G * 8, hipMemcpyHostToDevice, h->stream));
    PLDA_HIP(h, hipMemcpyAsync(dtiles, pin_tiles, ntiles * 16, hipMemcpyHostToDevice, h->stream));
    gather_center_kernel<<<gKD, 256, 0, h->stream>>>(means, mu, dcls, K, D, Mg);
    em_row_weights_kernel<<<(unsigned)ceil_div(GD, 256), 256, 0, h->stream>>>(dgn, dgk, D, GD, kw1, kw2);
    PLDA_LAUNCH_CHECK(h);
    const int NT = (int)ceil_div(D, 16);
    const size_t rows_lds = (size_t)2 * 16 * RB * (16 * NT + 2) * 8, xtb_lds = (size_t)16 * RB * (16 * NT + 2) * 8;
    double *TTg = scr;                    // T_g^T (D <= 512; the blocked whitening's scratch is dead by then)
    if (D <= 512) {
      static DeviceOnce attr;
      if (attr.needed(h->device)) {
        PLDA_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&em_rows_kernel<1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 16 * (16 * 32 + 4) * 8));
        PLDA_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&em_xtb_kernel<1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 16 * (16 * 32 + 4) * 8));
        PLDA_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&em_rows_kernel<2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32 * (16 * 16 + 4) * 8));
        PLDA_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&em_xtb_kernel<2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 32 * (16 * 16 + 4) * 8));
        attr.done(h->device);
      }
    }
    // the chunk table of the two rank-k sums (D <= 208): fixed pointers, built once
    const SyrkChunk *dchunks = nullptr;
    int nchunks = 0;
    if (D <= 208) {
      std::vector<SyrkChunk> &hc = h->em_chunks_host;      // (kept in the handle: the upload below may still be reading it)
      // (about one chunk per CU: with 1.5 or 2 per CU -- smaller chunks, two workgroups resident -- the iteration is 8 % slower)
      hc.resize((size_t)em_rank_chunk_bound(G, D, K, h->num_cus));
      nchunks = em_rank_chunks(G, D, K, h->num_cus, Xg, gn.data(), gk.data(), Zr, Wn, hc.data());
      PLDA_HIP(h, h->em_chunks.reserve((size_t)nchunks * sizeof(SyrkChunk)));
      PLDA_HIP(h, hipMemcpyAsync(h->em_chunks.p, hc.data(), (size_t)nchunks * sizeof(SyrkChunk), hipMemcpyHostToDevice, h->stream));
      dchunks = h->em_chunks.as<SyrkChunk>();
    }
    double *Bcur = B, *Bnext = Balt;
    for (int it = 0; it < iters; ++it) {
      if (it == 0) em_first_T_kernel<<<dim3(gDD, G), 256, 0, h->stream>>>(dgn, D, sDD, Tg);
      else PLDA_TRY(whiten_groups_f64(h, W, Bcur, dgn, D, Tg, scr, dflag, G));
      PLDA_LAUNCH_CHECK(h);
      if (D <= 512) {
        const int xsplit = 2 * (int)ceil_div(NT, RB) * G <= h->num_cus * 3 / 2 ? 2 : 1;
        if (RB == 2) {
          em_xtb_kernel<2><<<dim3(xsplit * (unsigned)ceil_div(NT, 2), (unsigned)G), 1024, xtb_lds, h->stream>>>(Tg, Bcur, D, Xg, TTg, xsplit);
          em_rows_kernel<2><<<(unsigned)ntiles, 1024, rows_lds, h->stream>>>(Mg, dtiles, TTg, Xg, dgn, D, Zr, Wn);
        } else {
          em_xtb_kernel<1><<<dim3(xsplit * (unsigned)NT, (unsigned)G), 1024, xtb_lds, h->stream>>>(Tg, Bcur, D, Xg, TTg, xsplit);
          em_rows_kernel<1><<<(unsigned)ntiles, 1024, rows_lds, h->stream>>>(Mg, dtiles, TTg, Xg, dgn, D, Zr, Wn);
        }
        PLDA_LAUNCH_CHECK(h);
      } else {
        PLDA_TRY(gemm_f64_batched(h, D, D, D, 1.0, Tg, D, 1, sDD, Bcur, D, 1, 0, nullptr, 0.0, Xg, D, sDD, G));
        for (int g = 0; g < G; ++g) {
          const int64_t kg = goff[g + 1] - goff[g];
          const double *rows = Mg + (size_t)goff[g] * D;
          double *V = Wn + (size_t)goff[g] * D, *Y = Zr + (size_t)goff[g] * D;
          PLDA_TRY(gemm_f64(h, kg, D, D, 1.0, rows, D, 1, Tg + (size_t)g * DD, 1, D, nullptr, 0.0, V, D));
          PLDA_TRY(gemm_f64(h, kg, D, D, 1.0, V, D, 1, Xg + (size_t)g * DD, D, 1, nullptr, 0.0, Y, D));
        }
        em_rows_finish_kernel<<<gKD, 256, 0, h->stream>>>(Mg, dcls, h->f_counts.as<int64_t>(), K, D, Zr, Wn);
        PLDA_LAUNCH_CHECK(h);
      }
      bool fused = false;
      PLDA_TRY(em_rank_sums_mstep_f64(h, D, dchunks, nchunks, S, (double)K, class_weight, cntW, cntB, W, Bcur, Bnext, &fused));
      if (fused) {
        std::swap(Bcur, Bnext);
      } else {
        PLDA_TRY(syrk_pair_f64(h, D, GD, Xg, D, kw1, K, Zr, D, 1.0, P1, D));
        PLDA_TRY(syrk_pair_f64(h, D, GD, Xg, D, kw2, K, Wn, D, 1.0, P2, D));
        em_rows_mstep_kernel<<<gDD, 256, 0, h->stream>>>(S, P1, P2, D, (double)K, class_weight, cntW, cntB, W, Bcur);
        PLDA_LAUNCH_CHECK(h);
      }
    }
    if (Bcur != B) PLDA_HIP(h, hipMemcpyAsync(B, Bcur, DD * 8, hipMemcpyDeviceToDevice, h->stream));   // (the M-step alternates two buffers)
  } else if (grouped) {
    // ---- moment form (header of em_moment_mstep_kernel) ----
    PLDA_HIP(h, h->w[5].reserve((size_t)K * D * 8 + (size_t)K * 4 + 64));
    PLDA_HIP(h, h->w[6].reserve(group_bytes + DD * 8 + (size_t)G * 16 + 64));
    double *Mg = h->w[5].as<double>();
    int *dcls = reinterpret_cast<int *>(Mg + (size_t)K * D);
    const size_t GDD = (size_t)G * DD;
    double *Cg = h->w[6].as<double>(), *Tg = Cg + GDD, *Xg = Tg + GDD, *P1 = Xg + GDD, *P2 = P1 + GDD, *QC = P2 + GDD, *XtX = QC + GDD,
           *Rg = XtX + GDD, *QCQ = Rg + GDD, *Csum = QCQ + GDD, *dgn = Csum + DD, *dgk = dgn + G;
    int *dflag = h->fit_flag.as<int>();          // (its own buffer: the export kernel that ends the fit reads it)
    std::copy(gn.begin(), gn.end(), pin_gn);
    std::copy(gk.begin(), gk.end(), pin_gk);
    PLDA_HIP(h, hipMemcpyAsync(dcls, cls, (size_t)K * 4, hipMemcpyHostToDevice, h->stream));
    PLDA_HIP(h, hipMemcpyAsync(dgn, pin_gn, (size_t)G * 8, hipMemcpyHostToDevice, h->stream));
    PLDA_HIP(h, hipMemcpyAsync(dgk, pin_gk, (size_t)G * 8, hipMemcpyHostToDevice, h->stream));
    gather_center_kernel<<<gKD, 256, 0, h->stream>>>(means, mu, dcls, K, D, Mg);
    PLDA_LAUNCH_CHECK(h);
    for (int g = 0; g < G; ++g) {
      const double *rows = Mg + (size_t)goff[g] * D;
      PLDA_TRY(gemm_f64(h, D, D, goff[g + 1] - goff[g], 1.0, rows, 1, D, rows, D, 1, nullptr, 0.0, Cg + (size_t)g * DD, D));
    }
    if (G == 1) PLDA_HIP(h, hipMemcpyAsync(Csum, Cg, DD * 8, hipMemcpyDeviceToDevice, h->stream));
    else PLDA_TRY(gemm_f64(h, D, D, K, 1.0, Mg, 1, D, Mg, D, 1, nullptr, 0.0, Csum, D));
    const int64_t sDD = (int64_t)DD;
    for (int it = 0; it < iters; ++it) {
      if (it == 0) em_first_T_kernel<<<dim3(gDD, G), 256, 0, h->stream>>>(dgn, D, sDD, Tg);
      else PLDA_TRY(whiten_groups_f64(h, W, B, dgn, D, Tg, P2 /* 3 G D^2 of scratch: P2 | QC | XtX, dead between iterations */, dflag, G));
      PLDA_LAUNCH_CHECK(h);
      // (A(m, k) = A[m sam + k sak], B(k, n) = B[k sbk + n sbn])
      const GemmSet s1[2] = {{Tg, D, 1, sDD, B, D, 1, 0, Xg, D, sDD},            // X  = T B
                             {Tg, D, 1, sDD, Cg, D, 1, sDD, P1, D, sDD}};        // P1 = T C_g
      PLDA_TRY(gemm_f64_multi(h, D, D, D, s1, 2, G));
      const GemmSet s2[3] = {{Xg, 1, D, sDD, P1, D, 1, sDD, QC, D, sDD},         // QC  = X^T P1     (= Q C_g, Q = B A^-1 = X^T T)
                             {P1, D, 1, sDD, Tg, 1, D, sDD, P2, D, sDD},         // P2  = P1 T^T     (= T C_g T^T, the whitened moments)
                             {Xg, 1, D, sDD, Xg, D, 1, sDD, XtX, D, sDD}};       // XtX = X^T X      (Mx = B - n XtX)
      PLDA_TRY(gemm_f64_multi(h, D, D, D, s2, 3, G));
      const GemmSet s3[1] = {{P2, D, 1, sDD, Xg, D, 1, sDD, Rg, D, sDD}};        // R   = P2 X
      PLDA_TRY(gemm_f64_multi(h, D, D, D, s3, 1, G));
      const GemmSet s4[1] = {{Xg, 1, D, sDD, Rg, D, 1, sDD, QCQ, D, sDD}};       // QCQ = X^T R      (= Q C_g Q^T)
      PLDA_TRY(gemm_f64_multi(h, D, D, D, s4, 1, G));
      em_moment_mstep_kernel<<<gDD, 256, 0, h->stream>>>(S, Csum, XtX, QC, QCQ, dgn, dgk, G, D, (double)K, class_weight, cntW, cntB, W, B);
      PLDA_LAUNCH_CHECK(h);
    }
    // the flag of the EM's factorisations travels with the model export that ends the fit (round 4: read here, the
    // synchronisation left the GPU idle for ~45 us before GetOutput's first kernel).  A failed factorisation hands
    // GetOutput non-finite matrices, which its kernels refuse at once; the error reported is this one.
  } else {
  PLDA_HIP(h, h->w[5].reserve((size_t)K * D * 8 * 3));   // Mc / P, Y1, Y2
  PLDA_HIP(h, h->w[6].reserve(DD * 8 * 7 + (size_t)D * 8));
  double *Mc = h->w[5].as<double>(), *Y1 = Mc + (size_t)K * D, *Y2 = Y1 + (size_t)K * D;
  double *T = h->w[6].as<double>(), *Tinv = T + DD, *Bt = Tinv + DD, *Wt = Bt + DD, *tmp = Wt + DD,
         *Bu = tmp + DD, *Wu = Bu + DD, *psi = Wu + DD;
  center_kernel<<<gKD, 256, 0, h->stream>>>(means, mu, K, D, Mc);
  PLDA_LAUNCH_CHECK(h);
  for (int it = 0; it < iters; ++it) {
    PLDA_TRY(simdiag_f64(h, W, B, D, T, Tinv, psi, it > 0));
    // P = Mc T^T  (reuse Y1 as P, then scale into Y1/Y2)
    PLDA_TRY(gemm_f64(h, K, D, D, 1.0, Mc, D, 1, T, 1, D, nullptr, 0.0, Y2, D));
    em_scale_kernel<<<gKD, 256, 0, h->stream>>>(Y2, offsets, psi, K, D, Y1, Y2);
    PLDA_LAUNCH_CHECK(h);
    PLDA_TRY(gemm_f64(h, D, D, K, 1.0, Y1, 1, D, Y1, D, 1, nullptr, 0.0, Bt, D));
    PLDA_TRY(gemm_f64(h, D, D, K, 1.0, Y2, 1, D, Y2, D, 1, nullptr, 0.0, Wt, D));
    em_diag_kernel<<<D, 256, 0, h->stream>>>(offsets, psi, K, D, Bt, Wt);
    PLDA_LAUNCH_CHECK(h);
    // un-project: Bu = Tinv Bt Tinv^T ; Wu = Tinv Wt Tinv^T
    PLDA_TRY(gemm_f64(h, D, D, D, 1.0, Tinv, D, 1, Bt, D, 1, nullptr, 0.0, tmp, D));
    PLDA_TRY(gemm_f64(h, D, D, D, 1.0, tmp, D, 1, Tinv, 1, D, nullptr, 0.0, Bu, D));
    PLDA_TRY(gemm_f64(h, D, D, D, 1.0, Tinv, D, 1, Wt, D, 1, nullptr, 0.0, tmp, D));
    PLDA_TRY(gemm_f64(h, D, D, D, 1.0, tmp, D, 1, Tinv, 1, D, nullptr, 0.0, Wu, D));
    em_mstep_kernel<<<gDD, 256, 0, h->stream>>>(S, Wu, Bu, D, cntW, cntB, W, B);
    PLDA_LAUNCH_CHECK(h);
  }
  }
  ts_em.close();
  PLDA_HIP(h, hipEventRecord(h->fit_ev[1], h->stream));

  // ---------------- GetOutput ----------------
  PLDA_HIP(h, h->d_mean.reserve((size_t)D * 8));
  PLDA_HIP(h, h->d_transform.reserve(DD * 8));
  PLDA_HIP(h, h->d_psi.reserve((size_t)D * 8));
  PLDA_HIP(h, h->d_offset.reserve((size_t)D * 8));
  // from here on the model buffers are overwritten: the handle counts as fitted again only once the
  // factorisation flags have been read back (a failed GetOutput must not leave a NaN model behind a true flag)
  h->fitted = false;
  // enqueue only: with the direct eigensolver GetOutput reads nothing back before the model copies below
  bool pending = false;
  if (iters > 0 && h->simdiag_has_vr)   // the per-iteration EM arm ran: warm start from its last eigenvectors
    PLDA_TRY(simdiag_f64(h, W, B, D, h->d_transform.as<double>(), nullptr, h->d_psi.as<double>(), true));
  else
    PLDA_TRY(simdiag_enqueue(h, W, B, D, h->d_transform.as<double>(), nullptr, h->d_psi.as<double>(), &pending));
  PLDA_HIP(h, hipMemcpyAsync(h->d_mean.p, mu, (size_t)D * 8, hipMemcpyDeviceToDevice, h->stream));
  h->Dout = D; h->Din = D;
  h->h_mean.resize(D); h->h_transform.resize(DD); h->h_psi.resize(D); h->h_offset.resize(D);
  // the host mirror of the model: the four copies land in one pinned area (queued back to back, no staging through the
  // runtime's bounce buffer: to the pageable vectors they were ~35 us each, a fifth of GetOutput at D = 200)
  for (int attempt = 0; attempt < 2; ++attempt) {
    PLDA_TRY(compute_offset_device(h));
    {
      double *pm_dev = nullptr;
      PLDA_HIP(h, hipHostGetDevicePointer(reinterpret_cast<void **>(&pm_dev), h->pin_model, 0));
      const int *chol_dev = nullptr, *eig_dev = nullptr;
      if (pending) simdiag_flags(h, D, &chol_dev, &eig_dev);
      export_model_kernel<<<(unsigned)ceil_div((int64_t)(3 * (size_t)D + DD), 256), 256, 0, h->stream>>>(
          h->d_mean.as<double>(), h->d_transform.as<double>(), h->d_psi.as<double>(), h->d_offset.as<double>(), D,
          h->fit_flag.as<int>(), chol_dev, eig_dev, pm_dev);
      PLDA_LAUNCH_CHECK(h);
    }
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    if (*em_flag_host) return fail(h, PLDA_E_NUMERIC, "fit: W + nB is not positive definite");
    std::memcpy(h->h_mean.data(), pm, (size_t)D * 8);
    std::memcpy(h->h_transform.data(), pm + D, DD * 8);
    std::memcpy(h->h_psi.data(), pm + D + DD, (size_t)D * 8);
    std::memcpy(h->h_offset.data(), pm + 2 * (size_t)D + DD, (size_t)D * 8);
    if (!pending) break;
    pending = false;
    bool redo = false;
    PLDA_TRY(simdiag_finish_with(h, W, B, D, h->d_transform.as<double>(), nullptr, h->d_psi.as<double>(), em_flag_host[1],
                                 em_flag_host[2], &redo));
    if (!redo) break;
  }
  const double t3 = now_ms();
  h->fitted = true;
  ++h->model_epoch;
  h->fit_K = K; h->fit_D = D;
  float em_span = 0.f, stats_span = 0.f;
  PLDA_HIP(h, hipEventElapsedTime(&em_span, h->fit_ev[0], h->fit_ev[1]));
  if (stats_bad) {       // the statistics pass of the same call: its span on the stream (t1 = the host clock at ITS start)
    PLDA_HIP(h, hipEventElapsedTime(&stats_span, h->fit_ev[2], h->fit_ev[0]));
    h->fit_ms[0] = (double)stats_span;
  }
  // em_ms: the EM's span on the stream; output_ms: the rest of the wall clock of this call (GetOutput, model copies)
  h->fit_ms[1] = (double)em_span; h->fit_ms[2] = (t3 - t1) - (double)em_span - (double)stats_span; h->fit_ms[3] = (double)iters;
  return PLDA_OK;
}

int fit_device(plda_handle *h, const double *dX, int64_t N, int D, const uint64_t *dlabels, int64_t K,
               int iters) {
  if (!dX || !dlabels || N <= 0 || D <= 0 || iters < 0) return fail(h, PLDA_E_INVAL, "fit: bad argument");
  if (K == 1)
    return fail(h, PLDA_E_ONE_SPEAKER,
                "Number of speakers is 1. Aborting PLDA esimation, at least two speakers are required!");
  PLDA_TRY(fit_stats_device(h, dX, N, D, dlabels, K, true));
  return fit_em_device(h, K, D, iters);
}

// ------------------------------------------------------------------------------------
// Mplda_transform grouping (pldamodule.cpp:139-168): labels are arbitrary u64 here, so
// the host compacts them (sorted unique) and the device reuses K1a/K1.
// ------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------
// Grouping by ARBITRARY uint64 labels (Mplda_transform, pldamodule.cpp:139-156: a std::map keyed by
// label value): the same LSD radix sort over as many 8-bit digits as the largest label has, then
// boundaries by adjacent difference.  Leaves, on the device: perm (row ids grouped by label,
// ascending row within a label -- the order the reference accumulates in), offsets[G+1], the
// distinct labels in ascending order (std::map iteration order, :164) and returns G.
// ------------------------------------------------------------------------------------
__global__ void labels_split_kernel(const uint64_t *__restrict__ labels, int64_t N, uint32_t *__restrict__ keys,
                                    uint32_t *__restrict__ vals, unsigned long long *__restrict__ maxlab) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long l = 0;
  if (r < N) { l = labels[r]; keys[r] = (uint32_t)l; vals[r] = (uint32_t)r; }
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long y = __shfl_xor(l, o); l = y > l ? y : l; }
  if ((threadIdx.x & 63) == 0 && l) atomicMax(maxlab, l);
}

__global__ void labels_hi_kernel(const uint64_t *__restrict__ labels, const uint32_t *__restrict__ vals, int64_t N,
                                 uint32_t *__restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) keys[i] = (uint32_t)(labels[vals[i]] >> 32);
}

__global__ void group_flag_kernel(const uint64_t *__restrict__ labels, const uint32_t *__restrict__ perm, int64_t N,
                                  int *__restrict__ flag /*[N+1]*/) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) flag[i] = (i == 0 || labels[perm[i]] != labels[perm[i - 1]]) ? 1 : 0;
  if (i == N) flag[N] = 0;
}

// pos = exclusive scan of the boundary flags: group id of sorted position i = pos[i + 1] - 1
__global__ void group_emit_kernel(const uint64_t *__restrict__ labels, const uint32_t *__restrict__ perm,
                                  const int *__restrict__ pos, int64_t N, uint64_t *__restrict__ uniq,
                                  int *__restrict__ offsets) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N && pos[i + 1] != pos[i]) { uniq[pos[i]] = labels[perm[i]]; offsets[pos[i]] = (int)i; }
  if (i == N) offsets[pos[N]] = (int)N;
}

int group_by_label_device(plda_handle *h, const uint64_t *dlabels, int64_t N, uint32_t **perm_out, int **offsets_out,
                          uint64_t **uniq_out, int64_t *G_out) {
  if (N >= (1ll << 31)) return fail(h, PLDA_E_INVAL, "transform: N too large");
  const int nblocks = (int)ceil_div(N, RS_CHUNK);
  PLDA_HIP(h, h->w[0].reserve((size_t)N * 4 * 4));                  // keys a/b, vals a/b
  PLDA_HIP(h, h->w[2].reserve((size_t)256 * nblocks * 4 + 16));     // digit histograms (+ max label)
  PLDA_HIP(h, h->w[4].reserve((size_t)(N + 2) * 4));                // boundary flags -> positions
  uint32_t *ka = h->w[0].as<uint32_t>(), *va = ka + N, *kb = va + N, *vb = kb + N;
  int *hist = h->w[2].as<int>();
  unsigned long long *dmax = reinterpret_cast<unsigned long long *>(hist + (size_t)256 * nblocks + (((size_t)256 * nblocks) & 1));
  int *pos = h->w[4].as<int>();
  PLDA_HIP(h, hipMemsetAsync(dmax, 0, 8, h->stream));
  labels_split_kernel<<<(unsigned)ceil_div(N, 256), 256, 0, h->stream>>>(dlabels, N, ka, va, dmax);
  PLDA_LAUNCH_CHECK(h);
  unsigned long long hmax = 0;
  PLDA_HIP(h, hipMemcpyAsync(&hmax, dmax, 8, hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  int bits = 1;
  while (bits < 64 && (hmax >> bits)) bits++;
  auto passes = [&](int nbits) -> int {
    for (int shift = 0; shift < nbits; shift += 8) {
      rs_hist_kernel<<<nblocks, RS_THREADS, 0, h->stream>>>(ka, N, shift, nblocks, hist);
      scan_kernel<<<1, 1024, 0, h->stream>>>(hist, (int64_t)256 * nblocks);
      rs_scatter_kernel<<<nblocks, RS_THREADS, 0, h->stream>>>(ka, va, kb, vb, N, shift, nblocks, hist);
      PLDA_LAUNCH_CHECK(h);
      std::swap(ka, kb);
      std::swap(va, vb);
    }
    return PLDA_OK;
  };
  PLDA_TRY(passes(std::min(bits, 32)));
  if (bits > 32) {
    labels_hi_kernel<<<(unsigned)ceil_div(N, 256), 256, 0, h->stream>>>(dlabels, va, N, ka);
    PLDA_LAUNCH_CHECK(h);
    PLDA_TRY(passes(bits - 32));
  }
  group_flag_kernel<<<(unsigned)ceil_div(N + 1, 256), 256, 0, h->stream>>>(dlabels, va, N, pos);
  scan_kernel<<<1, 1024, 0, h->stream>>>(pos, N + 1);
  PLDA_LAUNCH_CHECK(h);
  int hG = 0;
  PLDA_HIP(h, hipMemcpyAsync(&hG, pos + N, 4, hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  const int64_t G = hG;
  PLDA_HIP(h, h->w[1].reserve((size_t)(G + 2) * 4 + 64));
  PLDA_HIP(h, h->w[5].reserve((size_t)G * 8));
  int *offsets = h->w[1].as<int>();
  uint64_t *uniq = h->w[5].as<uint64_t>();
  group_emit_kernel<<<(unsigned)ceil_div(N + 1, 256), 256, 0, h->stream>>>(dlabels, va, pos, N, uniq, offsets);
  PLDA_LAUNCH_CHECK(h);
  *perm_out = va; *offsets_out = offsets; *uniq_out = uniq; *G_out = G;
  return PLDA_OK;
}

// per-label means of already grouped rows (pldamodule.cpp:147-168)
int group_centroids_device(plda_handle *h, const double *dX, int64_t N, int D, const uint32_t *perm, const int *offsets,
                           int64_t G, double *dmeans, int32_t *dcounts32) {
  PLDA_HIP(h, h->w[3].reserve((size_t)N * 8));
  centroid_kernel<<<(unsigned)G, 256, 0, h->stream>>>(dX, D, perm, offsets, dmeans, h->w[3].as<double>());
  PLDA_LAUNCH_CHECK(h);
  counts_to_i32_kernel<<<(unsigned)ceil_div(G, 256), 256, 0, h->stream>>>(offsets, G, dcounts32);
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

int group_means_device(plda_handle *h, const double *dX, int64_t N, int D, const uint64_t *ddense,
                       int64_t Ku, double *dmeans, int32_t *dcounts32) {
  uint32_t *perm = nullptr;
  int *offsets = nullptr;
  PLDA_TRY(sort_by_label(h, ddense, N, Ku, &perm, &offsets));
  PLDA_HIP(h, h->w[3].reserve((size_t)N * 8));
  centroid_kernel<<<(unsigned)Ku, 256, 0, h->stream>>>(dX, D, perm, offsets, dmeans, h->w[3].as<double>());
  PLDA_LAUNCH_CHECK(h);
  counts_to_i32_kernel<<<(unsigned)ceil_div(Ku, 256), 256, 0, h->stream>>>(offsets, Ku, dcounts32);
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

}  // namespace plda
