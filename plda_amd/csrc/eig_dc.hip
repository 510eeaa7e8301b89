// plda_amd/csrc/eig_dc.hip -- symmetric eigensolver of GetOutput (K7), direct method:
//
//   Householder tridiagonalisation  ->  divide and conquer on the tridiagonal  ->  back-transformation
//
// The reference's GetOutput (Kaldi PldaEstimator::GetOutput as driven by pldamodule.cpp:102-106; restated in
// oracle/plda_oracle.c:432-485) diagonalises the whitened between-class covariance with SpMatrix::Eig, which is
// itself tridiagonalisation + QL.  The one-sided block Jacobi of linalg.hip needs ~11 sweeps of D/4 launches
// (5.3 ms at D = 200); everything here is O(log D) launches after one D-step tridiagonalisation.
//
//   tridiag_reg_kernel    D <= 256: the matrix lives in the registers of ONE workgroup (lower-triangle 32 x 32
//                         block ownership as chol_small_kernel), D - 2 steps of (publish column -> v, tau ->
//                         p = A v -> w -> rank-2 update), three barriers per step, no global synchronisation
//   tridiag_rows_kernel   D <= 2048: rows dealt cyclically to D/8 workgroups (register-resident), ONE all-gather
//                         per step through self-validating words (p_i and the next column travel together)
//   dc_leaf_kernel        implicit QL (Wilkinson shift) on leaves of <= 16, one wave per leaf, d / e in lanes
//   dc_merge_roots_kernel per merge: rank sort, deflation (dlaed2's two criteria), secular roots by the two-pole
//                         rational iteration with a bisection safeguard, 8 lanes per root
//   dc_merge_vectors_kernel  Gu-Eisenstat z, normalised eigenvectors of D + rho z z^T -> the level's update matrix
//   (one dense fp64 MFMA GEMM per level applies it to the eigenvector rows)
//   householder_T / _rows_kernel  back-transformation in compact-WY blocks of 8 reflectors, two eigenvectors per wave
//
// scripts/proto_dc_eig.py is the NumPy prototype of the same algorithm (same deflation rule, same iteration).
#include <algorithm>
#include <cmath>

#include "common.hpp"

#include <utility>

namespace plda {

namespace {

constexpr double DC_EPS = 2.220446049250313e-16;
constexpr int DC_LEAF = 16;       // leaves have ceil(n / 2^depth) <= 16 rows
constexpr int DC_NMAX = 2048;
constexpr int DC_G = 16;          // lanes per secular root (one DPP row)
constexpr int DC_RS = 16;         // roots per workgroup of dc_merge_roots_kernel (256 threads)

__device__ __forceinline__ double dc_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double dc_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}

// segment `idx` of the 2^depth-way recursive halving of [0, n): the left part gets floor(len / 2)
__host__ __device__ inline void dc_segment(int n, int depth, int idx, int &off, int &len) {
  off = 0;
  len = n;
  for (int bit = depth - 1; bit >= 0; --bit) {
    const int h1 = len / 2;
    if ((idx >> bit) & 1) { off += h1; len -= h1; }
    else len = h1;
  }
}

// ------------------------------------------------------------------------------------
// scale = 2^-floor(log2 max|g_ij|): sums of squares then neither overflow nor underflow
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void eig_absmax_kernel(const double *__restrict__ G, int n,
                                                         unsigned long long *__restrict__ amax_bits, int *__restrict__ flag) {
  double m = 0.0;
  bool bad = false;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n * n; i += gridDim.x * 256) {
    const double x = fabs(G[i]);
    if (!(x <= 1.7976931348623157e308)) bad = true;   // NaN or inf
    m = fmax(m, x);
  }
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(amax_bits, (unsigned long long)__double_as_longlong(m));   // non-negative doubles order as integers
  if (bad) atomicOr(flag, 4);
}
__global__ void eig_scale_kernel(const unsigned long long *__restrict__ amax_bits, double *__restrict__ scale,
                                 const int *__restrict__ flag) {
  const double m = __longlong_as_double((long long)*amax_bits);
  double s = 1.0;
  if (m > 0.0 && !*flag) {
    int ex;
    (void)frexp(m, &ex);          // m = f 2^ex, f in [0.5, 1)
    s = ldexp(1.0, 1 - ex);       // m s in [1, 2)
  }
  scale[0] = s;
  scale[1] = 1.0 / s;
}

// ------------------------------------------------------------------------------------
// tridiagonalisation, matrix in the registers of one workgroup (n <= 32 NB <= 256)
// thread (ty, tx) of the 32 x 32 grid owns A[32a + ty][32b + tx] for the blocks a >= b (diagonal blocks whole)
// ------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(1024) void tridiag_reg_kernel(const double *__restrict__ G, int n,
                                                           const double *__restrict__ scale,
                                                           double *__restrict__ dd, double *__restrict__ ee,
                                                           double *__restrict__ Vh, double *__restrict__ tau) {
  constexpr int NE = NB * (NB + 1) / 2;
  constexpr int NR = NB * 32;
  constexpr int NW = (NR + 63) / 64;          // waves that hold the NR p-values
  __shared__ double xs[2][NR];
  __shared__ double ps[NR];
  __shared__ double prow[NR];
  __shared__ double part[16][NR];
  __shared__ double red[4];
  const int t = threadIdx.x, tx = t & 31, ty = t >> 5, lane = t & 63, wave = t >> 6;
  const double sc = scale[0];
  double r[NE];
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const int i = a * 32 + ty, j = b * 32 + tx;
      r[a * (a + 1) / 2 + b] = (i < n && j < n) ? G[(size_t)i * n + j] * sc : 0.0;
    }
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    for (int kl = 0; kl < 32; ++kl) {
      const int j = kb * 32 + kl;
      if (j >= n - 2) break;
      double *x = xs[j & 1];
      // ---- publish column j below the diagonal (zeros above it) ----
      if (tx == kl) {
#pragma unroll
        for (int a = 0; a < NB; ++a) {
          const int i = a * 32 + ty;
          double v = 0.0;
          if (a >= kb) v = (i > j) ? r[a * (a + 1) / 2 + kb] : 0.0;
          x[i] = v;
        }
        if (ty == kl) dd[j] = r[kb * (kb + 1) / 2 + kb];
      }
      __syncthreads();
      // ---- v, tau (every wave for itself) ----
      double s2 = 0.0;
#pragma unroll
      for (int q = 0; q < (NR + 63) / 64; ++q) {
        const int i = lane + 64 * q;
        const double xv = i < NR ? x[i] : 0.0;
        s2 += (i > j + 1) ? xv * xv : 0.0;
      }
      s2 = wave_sum_f64(s2);
      const double x0 = x[j + 1];
      if (s2 == 0.0) {   // column already tridiagonal (uniform over the workgroup)
        if (t == 0) { ee[j] = x0; tau[j] = 0.0; }
        continue;
      }
      const double nx2 = fma(x0, x0, s2);
      const double nx = nx2 * dc_rsqrt(nx2);
      const double alpha = x0 >= 0.0 ? -nx : nx;
      const double v0 = x0 - alpha;
      const double tt = 2.0 * dc_rcp(fma(v0, v0, s2));
      double vc[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int c = b * 32 + tx;
        vc[b] = b >= kb ? (c == j + 1 ? v0 : x[c]) : 0.0;
      }
      // ---- p = A v: row sums over the stored blocks + column sums of the strictly-lower blocks ----
      double pc[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) pc[b] = 0.0;
#pragma unroll
      for (int a = kb; a < NB; ++a) {
        const int i = a * 32 + ty;
        const double vra = i == j + 1 ? v0 : x[i];
        double pr = 0.0;
#pragma unroll
        for (int b = kb; b <= a; ++b) pr = fma(r[a * (a + 1) / 2 + b], vc[b], pr);
#pragma unroll
        for (int b = kb; b < a; ++b) pc[b] = fma(r[a * (a + 1) / 2 + b], vra, pc[b]);
        // sum over tx: the 16-lane rows by DPP, the two rows of each half through readlane
        pr += dpp_f64<0xB1>(pr);
        pr += dpp_f64<0x4E>(pr);
        pr += dpp_f64<0x141>(pr);
        pr += dpp_f64<0x140>(pr);
        const double h0 = readlane_f64(pr, 0) + readlane_f64(pr, 16);
        const double h1 = readlane_f64(pr, 32) + readlane_f64(pr, 48);
        if (lane == 0) {
          prow[a * 32 + 2 * wave] = h0;
          prow[a * 32 + 2 * wave + 1] = h1;
        }
      }
#pragma unroll
      for (int b = kb; b < NB - 1; ++b) {
        double c = pc[b];
        c += __shfl_xor(c, 32);
        if (lane < 32) part[wave][b * 32 + tx] = c;
      }
      __syncthreads();
      double vp = 0.0;
      if (t < NR) {
        const int bb = t >> 5;
        double p = 0.0;
        if (bb >= kb && t > j) {
          p = prow[t];
          if (bb < NB - 1) {
#pragma unroll
            for (int w = 0; w < 16; ++w) p += part[w][t];
          }
        }
        const double vi = t == j + 1 ? v0 : x[t];
        ps[t] = p;
        vp = vi * p;
        if (t < n) Vh[(size_t)j * n + t] = vi;
      }
      if (wave < NW) {
        vp = wave_sum_f64(vp);
        if (lane == 0) red[wave] = vp;
      }
      if (t == 0) { ee[j] = alpha; tau[j] = tt; }
      __syncthreads();
      double vtp = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) vtp += red[w];
      const double beta = 0.5 * tt * tt * vtp;
      // ---- w = tt p - beta v; A -= v w^T + w v^T on the trailing blocks ----
      double wc[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) wc[b] = b >= kb ? fma(tt, ps[b * 32 + tx], -beta * vc[b]) : 0.0;
#pragma unroll
      for (int a = kb; a < NB; ++a) {
        const int i = a * 32 + ty;
        const double vra = i == j + 1 ? v0 : x[i];
        const double wra = fma(tt, ps[i], -beta * vra);
#pragma unroll
        for (int b = kb; b <= a; ++b)
          r[a * (a + 1) / 2 + b] = fma(-vra, wc[b], fma(-wra, vc[b], r[a * (a + 1) / 2 + b]));
      }
    }
  }
  // ---- the last 2 x 2 block ----
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const int i = a * 32 + ty, j = b * 32 + tx;
      const double v = r[a * (a + 1) / 2 + b];
      if (i == j && i < n && i >= n - 2) dd[i] = v;
      if (n >= 2 && i == n - 1 && j == n - 2) ee[n - 2] = v;
    }
  if (t == 0) {
    if (n >= 2) tau[n - 2] = 0.0;
    tau[n - 1] = 0.0;
    ee[n - 1] = 0.0;
  }
}

// ------------------------------------------------------------------------------------
// The same on FOUR waves (round 3): thread (ty, tx) of a 16 x 16 grid owns A[16a + ty][16b + tx] for the blocks a >= b
// (diagonal blocks whole) -- 91 doubles at n = 200, in the 512 registers a wave has when it is alone on its SIMD, so the
// register-resident form reaches n = 256 without spilling and GetOutput at D = 200 no longer pays a fabric round trip
// (2.6 us) per Householder step.  Vectors live in LDS in a permuted order (entry i at (i mod 16) PAD + i / 16) so that a
// thread's entries of its rows (16a + ty) and of its columns (16b + tx) are contiguous 16-byte reads.  Per step:
// publish column j | barrier | v, tau (every wave for itself); p = A v: row sums over tx by DPP inside the 16-lane rows,
// column sums over ty by two lane exchanges + four per-wave partials in LDS | barrier | p, v.p | barrier | w, rank-2 update.
// ------------------------------------------------------------------------------------
template <typename F, int... KB>
__device__ __forceinline__ void tr16_blocks(F &f, std::integer_sequence<int, KB...>) {
  (f(std::integral_constant<int, KB>{}), ...);
}

template <int NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void tridiag_reg16_kernel(const double *__restrict__ G, int n,
                                                            const double *__restrict__ scale,
                                                            double *__restrict__ dd, double *__restrict__ ee,
                                                            double *__restrict__ Vh, double *__restrict__ tau) {
  constexpr int NE = NB * (NB + 1) / 2;
  constexpr int PAD = (NB + 1) & ~1;
  constexpr int NV = 16 * PAD;                // positions of a permuted vector (<= 256)
  __shared__ __attribute__((aligned(16))) double xs[2][NV];
  __shared__ __attribute__((aligned(16))) double ps[NV];
  __shared__ __attribute__((aligned(16))) double prow[NV];
  __shared__ __attribute__((aligned(16))) double part[4][NV];
  __shared__ double red[4];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4, lane = t & 63, wave = t >> 6;
  const double sc = scale[0];
  // the vector position this thread serves in the per-entry phases, and its index
  const int pos_i = (t < NV) ? (t % PAD) * 16 + t / PAD : -1;
  const bool pos_ok = t < NV && (t % PAD) < NB;
  for (int q = t; q < NV; q += 256) { xs[0][q] = 0.0; xs[1][q] = 0.0; ps[q] = 0.0; prow[q] = 0.0; part[0][q] = part[1][q] = part[2][q] = part[3][q] = 0.0; }
  double r[NE];
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const int i = a * 16 + ty, j = b * 16 + tx;
      r[a * (a + 1) / 2 + b] = (i < n && j < n) ? G[(size_t)i * n + j] * sc : 0.0;
    }
  __syncthreads();
  // (the block index of the pivot column must be a compile-time constant -- the owners of column j are named registers --
  //  and a plain `#pragma unroll` gives up on a body of this size from NB = 13 on, which puts r[] into scratch memory:
  //  the blocks are instantiated one by one)
  auto block = [&](auto kbc) {
    constexpr int kb = decltype(kbc)::value;
    for (int kl = 0; kl < 16; ++kl) {
      const int j = kb * 16 + kl;
      if (j >= n - 2) break;
      double *x = xs[j & 1];
      // ---- publish column j below the diagonal (zeros above it) ----
      if (tx == kl) {
#pragma unroll
        for (int a = 0; a < NB; ++a) {
          const int i = a * 16 + ty;
          double v = 0.0;
          if (a >= kb) v = (i > j) ? r[a * (a + 1) / 2 + kb] : 0.0;
          x[ty * PAD + a] = v;
        }
        if (ty == kl) dd[j] = r[kb * (kb + 1) / 2 + kb];
      }
      __syncthreads();
      // ---- v, tau (every wave for itself) ----
      double s2 = 0.0;
#pragma unroll
      for (int q = 0; q < (NV + 63) / 64; ++q) {
        const int pos = lane + 64 * q;
        if (pos < NV) {
          const int i = (pos % PAD) * 16 + pos / PAD;
          const double xv = x[pos];
          s2 += (i > j + 1) ? xv * xv : 0.0;
        }
      }
      s2 = wave_sum_f64(s2);
      const double x0 = x[((j + 1) & 15) * PAD + ((j + 1) >> 4)];
      if (s2 == 0.0) {   // column already tridiagonal (uniform over the workgroup)
        if (t == 0) { ee[j] = x0; tau[j] = 0.0; }
        continue;
      }
      const double nx2 = fma(x0, x0, s2);
      const double nx = nx2 * dc_rsqrt(nx2);
      const double alpha = x0 >= 0.0 ? -nx : nx;
      const double v0 = x0 - alpha;
      const double tt = 2.0 * dc_rcp(fma(v0, v0, s2));
      double vr[PAD], vc[PAD];
      {
        const double2 *pr2 = reinterpret_cast<const double2 *>(x + ty * PAD), *pc2 = reinterpret_cast<const double2 *>(x + tx * PAD);
#pragma unroll
        for (int a = 0; a < PAD / 2; ++a) {
          const double2 u = pr2[a], w = pc2[a];
          vr[2 * a] = u.x; vr[2 * a + 1] = u.y; vc[2 * a] = w.x; vc[2 * a + 1] = w.y;
        }
#pragma unroll
        for (int a = 0; a < NB; ++a) {
          if (a * 16 + ty == j + 1) vr[a] = v0;
          if (a * 16 + tx == j + 1) vc[a] = v0;
        }
      }
      // ---- p = A v: row sums over the stored blocks + column sums of the strictly-lower blocks ----
      double pc[PAD];
#pragma unroll
      for (int b = 0; b < PAD; ++b) pc[b] = 0.0;
      double myrow = 0.0;                     // the row sum this lane publishes (block a == tx)
#pragma unroll
      for (int a = kb; a < NB; ++a) {
        double pr = 0.0;
#pragma unroll
        for (int b = kb; b <= a; ++b) pr = fma(r[a * (a + 1) / 2 + b], vc[b], pr);
#pragma unroll
        for (int b = kb; b < a; ++b) pc[b] = fma(r[a * (a + 1) / 2 + b], vr[a], pc[b]);
        // sum over tx: the 16-lane row by DPP (every lane of the row ends up with the sum)
        pr += dpp_f64<0xB1>(pr);
        pr += dpp_f64<0x4E>(pr);
        pr += dpp_f64<0x141>(pr);
        pr += dpp_f64<0x140>(pr);
        myrow = tx == a ? pr : myrow;
      }
      if (tx >= kb && tx < NB) prow[ty * PAD + tx] = myrow;
      // column sums: over the four rows of the wave by lane exchanges, then one partial per wave
#pragma unroll
      for (int b = kb; b < NB - 1; ++b) {
        double c = pc[b];
        c += __shfl_xor(c, 16);
        c += __shfl_xor(c, 32);
        pc[b] = c;
      }
      if (lane < 16) {
#pragma unroll
        for (int b = 0; b < PAD / 2; ++b)
          if (2 * b + 1 >= kb) *reinterpret_cast<double2 *>(&part[wave][tx * PAD + 2 * b]) = double2{pc[2 * b], pc[2 * b + 1]};
      }
      __syncthreads();
      double vp = 0.0;
      if (pos_ok) {
        const int i = pos_i;
        double p = 0.0;
        if (i > j) {
          p = prow[t];
          if ((i >> 4) < NB - 1) p += (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
        }
        const double vi = i == j + 1 ? v0 : x[t];
        ps[t] = p;
        vp = vi * p;
        if (i < n) Vh[(size_t)j * n + i] = vi;
      }
      vp = wave_sum_f64(vp);
      if (lane == 0) red[wave] = vp;
      if (t == 0) { ee[j] = alpha; tau[j] = tt; }
      __syncthreads();
      const double vtp = (red[0] + red[1]) + (red[2] + red[3]);
      const double beta = 0.5 * tt * tt * vtp;
      // ---- w = tt p - beta v; A -= v w^T + w v^T on the trailing blocks ----
      double wr[PAD], wc[PAD];
      {
        const double2 *pr2 = reinterpret_cast<const double2 *>(ps + ty * PAD), *pc2 = reinterpret_cast<const double2 *>(ps + tx * PAD);
#pragma unroll
        for (int a = 0; a < PAD / 2; ++a) {
          const double2 u = pr2[a], w = pc2[a];
          wr[2 * a] = fma(tt, u.x, -beta * vr[2 * a]); wr[2 * a + 1] = fma(tt, u.y, -beta * vr[2 * a + 1]);
          wc[2 * a] = fma(tt, w.x, -beta * vc[2 * a]); wc[2 * a + 1] = fma(tt, w.y, -beta * vc[2 * a + 1]);
        }
      }
#pragma unroll
      for (int a = kb; a < NB; ++a)
#pragma unroll
        for (int b = kb; b <= a; ++b)
          r[a * (a + 1) / 2 + b] = fma(-vr[a], wc[b], fma(-wr[a], vc[b], r[a * (a + 1) / 2 + b]));
    }
  };
  tr16_blocks(block, std::make_integer_sequence<int, NB>{});
  // ---- the last 2 x 2 block ----
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const int i = a * 16 + ty, j = b * 16 + tx;
      const double v = r[a * (a + 1) / 2 + b];
      if (i == j && i < n && i >= n - 2) dd[i] = v;
      if (n >= 2 && i == n - 1 && j == n - 2) ee[n - 2] = v;
    }
  if (t == 0) {
    if (n >= 2) tau[n - 2] = 0.0;
    tau[n - 1] = 0.0;
    ee[n - 1] = 0.0;
  }
}

// ------------------------------------------------------------------------------------
// Eight waves, the FULL matrix in registers (n <= 208: thread (ty, tx) of a 32 x 16 grid owns A[32a + ty][16b + tx], 91
// doubles), ONE workgroup barrier per Householder step.  With both triangles stored p = A v is row sums only -- no
// column sums across ty, no per-wave partials, no assembly phase -- and, as in the multi-workgroup kernel below, the
// next column is not read out of the updated matrix but follows from what everyone can see: its owners publish
// A[:, j + 1] as it stands beside p, and behind the step's barrier every wave forms for itself
//     v . p,  w = tau p - beta v,  x' = A[:, j + 1] - v w_{j+1} - w v_{j+1}  (column j + 1 of the updated matrix)
// into LDS (the eight waves write the same values), then applies the rank-2 update to its registers.  Costs the full
// update (3 n^2 instead of 2 n^2 FMAs per step) and saves two of the three barriers, the cross-ty lane exchanges, the
// partial-sum traffic and the p-assembly phase of the symmetric-storage kernel above.  (The same on FOUR waves needs
// 169 doubles per thread: half of them live in AGPRs and every use pays two v_accvgpr moves -- 0.54 against 0.53 ms.)
// The two triangles are updated by differently ordered FMAs and drift apart by an ulp per step; p is formed from rows.
// ------------------------------------------------------------------------------------
#ifndef TRF_CLOCK
#define TRF_CLOCK(i)     // (phase clocks of scripts/probe/tridiag_full_probe.hip)
#endif
template <int NB, int NA>
__global__ __launch_bounds__(512) void tridiag_full_kernel(const double *__restrict__ G, int n,
                                                           double *__restrict__ scale, double *__restrict__ dd,
                                                           double *__restrict__ ee, double *__restrict__ Vh,
                                                           double *__restrict__ tau, int *__restrict__ flag) {
  constexpr int NV = 16 * NB;                 // vector length, padded
  constexpr int NVP = NV + 16;                // (32 NA may exceed it by one 16-block)
  constexpr int NQ = (NV + 63) / 64;
  static_assert(32 * NA <= NVP, "row blocks beyond the padded vector");
  __shared__ double PS[2][NVP];               // p = A v
  __shared__ double XN[2][NVP];               // column j + 1 before the step's update
  __shared__ double VW[8][3][NVP];            // every wave's own v (two, alternating) and w: the fragment reads need no fix-ups
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4, lane = t & 63, wave = t >> 6;
  __shared__ double amax_s[8];
  __shared__ int bad_s;
  for (int q = t; q < NVP; q += 512) { PS[0][q] = PS[1][q] = 0.0; XN[0][q] = XN[1][q] = 0.0; }
  for (int q = t; q < 24 * NVP; q += 512) (&VW[0][0][0])[q] = 0.0;
  double *const ww = VW[wave][2];
  if (t == 0) bad_s = 0;
  double r[NA][NB];
  // the scaling of the input to [1, 2) (eig_absmax_kernel + eig_scale_kernel of the other paths) happens here, where
  // the matrix is loaded anyway: two launches and a memset fewer in front of the kernel
  double amax = 0.0;
  bool nonfinite = false;
  for (int q = t; q < n * n; q += 512) {      // (a pass of its own: with the maximum taken from the register copy the
    const double x = fabs(G[q]);              //  loads and the scaling pull apart and the matrix spills, 612 bytes a lane)
    if (!(x <= 1.7976931348623157e308)) nonfinite = true;   // NaN or inf
    amax = fmax(amax, x);
  }
  for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_xor(amax, o));
  __syncthreads();
  if (lane == 0) amax_s[wave] = amax;
  if (nonfinite) bad_s = 1;
  __syncthreads();
  double sc = 1.0;
  {
    double m = amax_s[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) m = fmax(m, amax_s[w]);
    const bool bad = bad_s != 0;
    if (m > 0.0 && !bad) {
      int ex;
      (void)frexp(m, &ex);          // m = f 2^ex, f in [0.5, 1)
      sc = ldexp(1.0, 1 - ex);      // m sc in [1, 2)
    }
    if (t == 0) {
      *flag = bad_s ? 4 : 0;        // (this kernel opens the decomposition: it sets the status word, no memset by the host)
      scale[0] = sc;
      scale[1] = 1.0 / sc;
    }
  }
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int i = a * 32 + ty, j = b * 16 + tx;
      r[a][b] = (i < n && j < n) ? G[(size_t)i * n + j] * sc : 0.0;
    }
  __syncthreads();
  if (tx == 0) {
#pragma unroll
    for (int a = 0; a < NA; ++a) XN[1][a * 32 + ty] = r[a][0];     // column 0 (XN[1]: the slot "step -1" would have used)
  }
  __syncthreads();
  // The reflector of a step is prepared where its column is made -- at the end of the step before, ahead of that step's
  // rank-2 update, whose FMAs cover the chain wave sum -> rsqrt -> reciprocal -- and handed over in registers:
  // alpha, v0, tau, and v itself in this wave's LDS copy VW[wave][j & 1].
  double alpha, v0, tt;
  auto reflector = [&](int j, const double (&xi)[NQ], double s2part) {   // xi: column j below the diagonal, this lane's entries
    const double s2 = wave_sum_f64(s2part);
    double x0 = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) x0 = (lane + 64 * q == j + 1) ? xi[q] : x0;
    x0 = readlane_f64(x0, (j + 1) & 63);
    const bool skip = s2 == 0.0;            // column already tridiagonal (the same in every wave): H = I
    const double nx2 = fma(x0, x0, s2);
    const double nx = nx2 * dc_rsqrt(skip ? 1.0 : nx2);
    alpha = skip ? x0 : (x0 >= 0.0 ? -nx : nx);
    v0 = skip ? 0.0 : x0 - alpha;
    tt = skip ? 0.0 : 2.0 * dc_rcp(fma(v0, v0, s2));
    double *vnext = VW[wave][j & 1];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = lane + 64 * q;
      if (i < NV) vnext[i] = i == j + 1 ? v0 : xi[q];
    }
  };
  {
    double xi[NQ], s2p = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = lane + 64 * q;
      xi[q] = (i < NV && i > 0) ? XN[1][i] : 0.0;
      s2p = i > 1 ? fma(xi[q], xi[q], s2p) : s2p;
    }
    reflector(0, xi, s2p);
  }
#ifdef TRF_PROBE_LOCALS
  TRF_PROBE_LOCALS
#endif
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto block = [&](auto kbc) {
    constexpr int kb = decltype(kbc)::value;      // 16-column block of the pivot
    constexpr int ka = kb / 2;                    // its 32-row block
    for (int kl = 0; kl < 16; ++kl) {
      const int j = kb * 16 + kl;
      if (j >= n - 2) break;
      const double *vw = VW[wave][j & 1];
      double *ps = PS[j & 1], *xn = XN[j & 1];
      TRF_CLOCK(0);
      if (tx == kl && ty == (j & 31)) dd[j] = r[ka][kb];
      wave_sync();                          // (v of this step: written by this wave at the end of the last one)
      TRF_CLOCK(1);
      // ---- p = A v: row sums over tx by DPP inside the 16-lane rows ----
      {
        double vc[NB];
#pragma unroll
        for (int b = kb; b < NB; ++b) vc[b] = vw[b * 16 + tx];
        double myrow = 0.0;
#pragma unroll
        for (int a = ka; a < NA; ++a) {
          double pr = 0.0;
#pragma unroll
          for (int b = kb; b < NB; ++b) pr = fma(r[a][b], vc[b], pr);
          pr += dpp_f64<0xB1>(pr);
          pr += dpp_f64<0x4E>(pr);
          pr += dpp_f64<0x141>(pr);
          pr += dpp_f64<0x140>(pr);
          myrow = tx == a ? pr : myrow;
        }
        if (tx >= ka && tx < NA) ps[tx * 32 + ty] = myrow;
      }
      // ---- column j + 1 as it stands ----
      if (tx == ((j + 1) & 15)) {
        if (kl < 15) {
#pragma unroll
          for (int a = 0; a < NA; ++a) xn[a * 32 + ty] = r[a][kb];
        } else if constexpr (kb + 1 < NB) {
#pragma unroll
          for (int a = 0; a < NA; ++a) xn[a * 32 + ty] = r[a][kb + 1];
        }
      }
      TRF_CLOCK(2);
      __syncthreads();
      TRF_CLOCK(3);
      // ---- every wave: v . p, w, the next column, its reflector ----
      double vq[NQ], pq[NQ], vp = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int i = lane + 64 * q;
        vq[q] = i < NV ? vw[i] : 0.0;
        pq[q] = (i < NV && i > j) ? ps[i] : 0.0;
        vp = fma(vq[q], pq[q], vp);
      }
      const double vtp = wave_sum_f64(vp);
      const double beta = 0.5 * tt * tt * vtp;
      const double vj1 = v0, ttj = tt, alphaj = alpha;
      const double wj1 = fma(ttj, ps[j + 1], -beta * vj1);
      double xi[NQ], s2p = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int i = lane + 64 * q;
        xi[q] = 0.0;
        if (i < NV) {
          const double wi = fma(ttj, pq[q], -beta * vq[q]);
          ww[i] = wi;
          xi[q] = i > j + 1 ? fma(-wi, vj1, fma(-vq[q], wj1, xn[i])) : 0.0;
          s2p = i > j + 2 ? fma(xi[q], xi[q], s2p) : s2p;
          if (wave == 0 && i < n) Vh[(size_t)j * n + i] = vq[q];
        }
      }
      if (t == 0) { ee[j] = alphaj; tau[j] = ttj; }
      wave_sync();
      TRF_CLOCK(4);
      reflector(j + 1, xi, s2p);           // (in the same scheduling region as the update below)
      // ---- A -= v w^T + w v^T (rows and columns up to j carry zeros in both vectors) ----
      double vr[NA], wr[NA];
#pragma unroll
      for (int a = ka; a < NA; ++a) {
        vr[a] = vw[a * 32 + ty];
        wr[a] = ww[a * 32 + ty];
      }
#pragma unroll
      for (int b = kb; b < NB; ++b) {
        const double vcb = vw[b * 16 + tx], wcb = ww[b * 16 + tx];
#pragma unroll
        for (int a = ka; a < NA; ++a) r[a][b] = fma(-vr[a], wcb, fma(-wr[a], vcb, r[a][b]));
      }
      TRF_CLOCK(5);
    }
  };
  tr16_blocks(block, std::make_integer_sequence<int, NB>{});
#ifdef TRF_PROBE_END
  TRF_PROBE_END
#endif
  // ---- the last 2 x 2 block ----
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int i = a * 32 + ty, j = b * 16 + tx;
      const double v = r[a][b];
      if (i == j && i < n && i >= n - 2) dd[i] = v;
      if (n >= 2 && i == n - 1 && j == n - 2) ee[n - 2] = v;
    }
  if (t == 0) {
    if (n >= 2) tau[n - 2] = 0.0;
    tau[n - 1] = 0.0;
    ee[n - 1] = 0.0;
  }
}

// ------------------------------------------------------------------------------------
// tridiagonalisation, rows dealt cyclically to W = ceil(n / 8) workgroups, n <= 2048 (above 1024 the row registers spill: correct, slow).  A row lives in the registers
// of 32 lanes (element k in lane k % 32).  Step j, every workgroup: v, tau from column j (all hold it) ->
// p_i = A_i . v for its rows -> publish p_i together with A[i][j+1] -> ONE all-gather -> w; column j+1 of the
// UPDATED matrix follows locally as A[i][j+1] - v_i w_{j+1} - w_i v_{j+1} -> rank-2 update of the own rows.
// The all-gather has no flags and no fences: every published 64-bit word carries 32 bits of payload and the step
// number, is stored and loaded with relaxed agent-scope atomics, and a reader simply re-loads a word until its
// tag is the step's (two buffers by step parity: a workgroup can run at most one step ahead of a reader).
// Launched cooperatively (all workgroups resident: they wait for each other).
// ------------------------------------------------------------------------------------
constexpr int TR_ROWS = 8;   // GetOutput with 4 / 8 / 16 rows per workgroup: 1.34 / 1.32 / 1.65 ms at D = 200, 4.18 / 4.02 / 4.36 at D = 512

constexpr int TR_FIRST_POLL = 10;              // x 128 cycles
constexpr int TR_TIMEOUT = 1 << 19;            // polls of ~1 us: a word that never arrives ends the kernel with flag 16

// sum over the 32 lanes of a half wave: DPP within the 16-lane rows, row_bcast15 into the upper row; the total is
// in lanes 16..31 (48..63) of the half
__device__ __forceinline__ double tr_sum32_upper(double x) {
  x += dpp_f64<0xB1>(x);
  x += dpp_f64<0x4E>(x);
  x += dpp_f64<0x141>(x);
  x += dpp_f64<0x140>(x);
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xa, 0xf, false);
  return x + __hiloint2double(hi, lo);
}

template <int E, int ROWS>   // E = ceil(n / 32) rounded up to the instantiations below; ROWS rows per workgroup of 32 ROWS threads
__global__ __launch_bounds__(ROWS * 32) void tridiag_rows_kernel(const double *__restrict__ G, int n, int W,
                                                           const double *__restrict__ scale, double *__restrict__ dd,
                                                           double *__restrict__ ee, double *__restrict__ Vh,
                                                           double *__restrict__ tau, unsigned long long *words,
                                                           int *flag, int dbg, long long *tl) {
  constexpr int NP = E * 32;                   // padded length
  constexpr int NT = ROWS * 32;
  constexpr int KQ = (NP + NT - 1) / NT;       // gathered elements per thread
  __shared__ double pp[2][NP], cc[2][NP];      // gathered p and next-column values, by step parity
  __shared__ int sh_to;
  const int t = threadIdx.x, slot = t >> 5, l = t & 31, half = (t >> 5) & 1;
  if (t == 0) sh_to = 0;
  const int w = blockIdx.x;
  const double sc = scale[0];
  const int myrow = w + slot * W;
  const bool rowok = myrow < n;
  const int mylane = (myrow & 31) + 32 * half, myq = myrow >> 5;   // where element `myrow` of a vector lives
  const int m0 = __builtin_amdgcn_readlane(mylane, 0), m1 = __builtin_amdgcn_readlane(mylane, 32);
  // a[q] = A[myrow][l + 32 q]; x[q] = column j of the current matrix in the same layout (every half wave holds it whole)
  double a[E], x[E];
#pragma unroll
  for (int q = 0; q < E; ++q) {
    const int k = l + 32 * q;
    a[q] = (rowok && k < n) ? G[(size_t)myrow * n + k] * sc : 0.0;
    x[q] = k < n ? G[k] * sc : 0.0;            // column 0 = row 0 (symmetric)
  }
  auto pick = [&](const double (&r)[E], int qq) {   // r[qq] for a runtime qq
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < E; ++q) v = qq == q ? r[q] : v;
    return v;
  };
  __syncthreads();
  int jw = 0, jq = 0;                          // j % W, j / W
  for (int j = 0; j < n - 2; ++j) {
    const int par = j & 1;
    const bool stamp = (dbg & 2) && blockIdx.x == 1 && t == 0 && j >= 16 && j < 32;
    if (stamp) tl[(j - 16) * 8 + 0] = __builtin_readcyclecounter();
    const int j1lane = (j + 1) & 31, j1q = (j + 1) >> 5;
    // ---- v, tau ----
    double s2 = 0.0;
#pragma unroll
    for (int q = 0; q < E; ++q) s2 += (l + 32 * q > j + 1) ? x[q] * x[q] : 0.0;
    s2 = readlane_f64(tr_sum32_upper(s2), 31);
    const double x0 = readlane_f64(pick(x, j1q), j1lane);
    const bool skip = s2 == 0.0;
    double alpha = x0, v0 = 0.0, tt = 0.0;
    if (!skip) {
      const double nx2 = fma(x0, x0, s2);
      const double nx = nx2 * dc_rsqrt(nx2);
      alpha = x0 >= 0.0 ? -nx : nx;
      v0 = x0 - alpha;
      tt = 2.0 * dc_rcp(fma(v0, v0, s2));
    }
    // v_k = 0 (k <= j), v0 (k == j + 1), x_k (k > j + 1); all zero when the column is already tridiagonal
    double v[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const int k = l + 32 * q;
      double vk = k > j + 1 ? x[q] : 0.0;
      vk = k == j + 1 ? v0 : vk;
      v[q] = skip ? 0.0 : vk;
    }
    if (w == 0) {
      if (slot == 0) {
#pragma unroll
        for (int q = 0; q < E; ++q) {
          const int k = l + 32 * q;
          if (k < n) Vh[(size_t)j * n + k] = v[q];
        }
      }
      if (t == 0) { ee[j] = alpha; tau[j] = tt; }
    }
    if (w == jw && slot == jq && l == (j & 31)) dd[j] = pick(a, j >> 5);
    if (stamp) tl[(j - 16) * 8 + 1] = __builtin_readcyclecounter();
    // ---- p_i for the own row, published with A[i][j+1] by the last lane of the half wave ----
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < E; ++q) acc = fma(a[q], v[q], acc);
    acc = tr_sum32_upper(myrow > j ? acc : 0.0);
    const double cj1 = pick(a, j1q);
    const double cv = half ? readlane_f64(cj1, 32 + j1lane) : readlane_f64(cj1, j1lane);
    const unsigned long long tag = (unsigned long long)(j + 1) << 32;
    unsigned long long *wp = words + (size_t)par * 4 * NP;
    if (rowok && l == 31) {
      const unsigned long long pb = (unsigned long long)__double_as_longlong(acc), cb = (unsigned long long)__double_as_longlong(cv);
      __hip_atomic_store(wp + myrow, tag | (pb & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(wp + NP + myrow, tag | (pb >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(wp + 2 * NP + myrow, tag | (cb & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(wp + 3 * NP + myrow, tag | (cb >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (stamp) tl[(j - 16) * 8 + 2] = __builtin_readcyclecounter();
    // ---- all-gather: re-load until every word of this thread carries the step's tag ----
    {
      unsigned long long wv[KQ][4];
      bool ready = false;
      int polls = 0;
      // a word needs ~1400 cycles to become visible to the other workgroups and a poll is one ~1400-cycle round trip:
      // polling at once always costs a second poll, so the first one is sent ~1300 cycles late (measured: GetOutput
      // 1.56 -> 1.44 ms at D = 200, 4.90 -> 4.49 ms at D = 512; PLDA_EIG_DEBUG bits 8.. override the delay)
      for (int z = (dbg >> 8) ? ((dbg >> 8) & 63) - 1 : TR_FIRST_POLL; z > 0; --z) __builtin_amdgcn_s_sleep(2);
      while (!ready) {
        ready = true;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          const int k = t + NT * q;
#pragma unroll
          for (int c = 0; c < 4; ++c)
            wv[q][c] = k < n ? __hip_atomic_load(wp + c * NP + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
        }
#pragma unroll
        for (int q = 0; q < KQ; ++q)
#pragma unroll
          for (int c = 0; c < 4; ++c) ready = ready && (wv[q][c] >> 32) == (tag >> 32);
        if (dbg & 1) ready = true;
        if (!ready) {
          if (++polls > TR_TIMEOUT || *(volatile int *)&sh_to) { sh_to = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        const int k = t + NT * q;
        if (k < NP) {
          pp[par][k] = __longlong_as_double((long long)((wv[q][0] & 0xffffffffull) | (wv[q][1] << 32)));
          cc[par][k] = __longlong_as_double((long long)((wv[q][2] & 0xffffffffull) | (wv[q][3] << 32)));
        }
      }
    }
    if (stamp) tl[(j - 16) * 8 + 3] = __builtin_readcyclecounter();
    __syncthreads();   // the only barrier of a step (pp / cc alternate, so the next step's writes cannot overtake readers)
    if (stamp) tl[(j - 16) * 8 + 4] = __builtin_readcyclecounter();
    if (*(volatile int *)&sh_to) {   // a word never arrived (uniform after the barrier): give up, the host falls back
      if (t == 0) atomicOr(flag, 16);
      return;
    }
    // ---- w = tt p - beta v, next column, rank-2 update of the own row ----
    double pr[E], cr[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
      pr[q] = pp[par][l + 32 * q];
      cr[q] = cc[par][l + 32 * q];
    }
    double vp = 0.0;
#pragma unroll
    for (int q = 0; q < E; ++q) vp = fma(v[q], pr[q], vp);
    const double vtp = readlane_f64(tr_sum32_upper(vp), 31);
    const double beta = 0.5 * tt * tt * vtp;
    const double vj1 = skip ? 0.0 : v0;
    const double wj1 = fma(tt, readlane_f64(pick(pr, j1q), j1lane), -beta * vj1);
    if (stamp) tl[(j - 16) * 8 + 5] = __builtin_readcyclecounter();
    double wk[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const int k = l + 32 * q;
      wk[q] = k > j ? fma(tt, pr[q], -beta * v[q]) : 0.0;
      x[q] = cr[q] - v[q] * wj1 - wk[q] * vj1;           // column j + 1 of the updated matrix
    }
    {
      const double vsel = pick(v, myq), wsel = pick(wk, myq);
      const double vm = half ? readlane_f64(vsel, m1) : readlane_f64(vsel, m0);   // v and w at index myrow
      const double wm = half ? readlane_f64(wsel, m1) : readlane_f64(wsel, m0);
      if (myrow > j) {
#pragma unroll
        for (int q = 0; q < E; ++q) a[q] = fma(-vm, wk[q], fma(-wm, v[q], a[q]));
      }
    }
    if (stamp) tl[(j - 16) * 8 + 6] = __builtin_readcyclecounter();
    if (stamp) tl[(j - 16) * 8 + 7] = __builtin_readcyclecounter();
    if (++jw == W) { jw = 0; ++jq; }
  }
  if (n >= 2 && w == (n - 2) % W && slot == (n - 2) / W && l == ((n - 2) & 31)) dd[n - 2] = pick(a, (n - 2) >> 5);
  if (w == (n - 1) % W && slot == (n - 1) / W) {
    if (l == ((n - 1) & 31)) { dd[n - 1] = pick(a, (n - 1) >> 5); ee[n - 1] = 0.0; tau[n - 1] = 0.0; }
    if (n >= 2 && l == ((n - 2) & 31)) { ee[n - 2] = pick(a, (n - 2) >> 5); tau[n - 2] = 0.0; }
  }
}

// ------------------------------------------------------------------------------------
// leaves: implicit QL with Wilkinson shift, one wave per leaf.  Lane i holds d_i and e_i (uniform
// indices -> readlane), lane k holds row k of the eigenvector matrix in LDS.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void dc_leaf_kernel(const double *__restrict__ dd, const double *__restrict__ ee, int n,
                                                     int depth, double *__restrict__ lam, double *__restrict__ Qt,
                                                     int *__restrict__ flag) {
  __shared__ double Z[DC_LEAF][DC_LEAF + 1];
  const int lane = threadIdx.x;
  int off, m;
  dc_segment(n, depth, blockIdx.x, off, m);
  double d = 0.0, e = 0.0;
  if (lane < m) {
    d = dd[off + lane];
    if (lane == 0 && off > 0) d -= fabs(ee[off - 1]);
    if (lane == m - 1 && off + m < n) d -= fabs(ee[off + m - 1]);
    if (lane < m - 1) e = ee[off + lane];
  }
  if (lane < DC_LEAF)
    for (int c = 0; c < DC_LEAF; ++c) Z[lane][c] = lane == c ? 1.0 : 0.0;
  // a non-finite tridiagonal (non-finite input) must not reach the merges: their index lists come from comparisons
  const bool finite = fabs(d) <= 1.7976931348623157e308 && fabs(e) <= 1.7976931348623157e308;
  if (__ballot(!finite) != 0ull) {
    if (lane == 0) atomicOr(flag, 4);
    if (lane < m) lam[off + lane] = 0.0;
    return;
  }
  auto getd = [&](int i) { return readlane_f64(d, __builtin_amdgcn_readfirstlane(i)); };
  auto gete = [&](int i) { return readlane_f64(e, __builtin_amdgcn_readfirstlane(i)); };
  bool failed = false;
  for (int l = 0; l < m && !failed; ++l) {
    int iter = 0;
    while (true) {
      // smallest mm >= l with a negligible e[mm] (mm = m - 1 if none)
      const double dn = __shfl_down(d, 1);   // d of lane i + 1
      const bool small = fabs(e) <= DC_EPS * (fabs(d) + fabs(dn));
      const unsigned long long mask = __ballot(small && lane >= l && lane < m - 1);
      const int mm = mask ? __builtin_ctzll(mask) : m - 1;
      if (mm == l) break;
      if (++iter > 80) { failed = true; break; }
      const double dl = getd(l), el = gete(l);
      double g = (getd(l + 1) - dl) * dc_rcp(2.0 * el);
      const double rr = sqrt(fma(g, g, 1.0));
      g = getd(mm) - dl + el * dc_rcp(g + (g >= 0.0 ? rr : -rr));
      double s = 1.0, c = 1.0, p = 0.0;
      bool underflow = false;
      // eigenvector entries: column i + 1 of a rotation is the column i the previous one produced -- it stays in a
      // register (z1; zc = the column it stands for) and column i is requested one rotation ahead, so that no LDS
      // round trip sits in the rotation chain (round 4: the read-modify-write per rotation cost as much as the chain)
      double z1 = 0.0, z0n = 0.0;
      int zc = mm;
      if (lane < m) { z1 = Z[lane][mm]; z0n = Z[lane][mm - 1]; }
      for (int i = mm - 1; i >= l; --i) {
        const double z0 = z0n;
        if (i - 1 >= l && lane < m) z0n = Z[lane][i - 1];
        const double ei = gete(i), di = getd(i), di1 = getd(i + 1);
        const double f = s * ei, b = c * ei;
        const double h2 = fma(f, f, g * g);
        if (h2 == 0.0) {
          if (lane == i + 1) { d -= p; e = 0.0; }
          if (lane == mm) e = 0.0;
          underflow = true;
          break;
        }
        const double hinv = dc_rsqrt(h2);
        if (lane == i + 1) e = h2 * hinv;
        s = f * hinv;
        c = g * hinv;
        g = di1 - p;
        const double r2 = fma(di - g, s, 2.0 * c * b);
        p = s * r2;
        if (lane == i + 1) d = g + p;
        g = fma(c, r2, -b);
        if (lane < m) Z[lane][i + 1] = fma(s, z0, c * z1);
        z1 = fma(c, z0, -s * z1);
        zc = i;
      }
      if (lane < m) Z[lane][zc] = z1;
      if (underflow) continue;
      if (lane == l) { d -= p; e = g; }
      if (lane == mm) e = 0.0;
    }
  }
  if (failed && lane == 0) atomicOr(flag, 1);
  if (lane < m) lam[off + lane] = d;
  // eigenvector i of the leaf = column i of Z -> row off + i of Qt (the rest of the row is zero: memset)
  // (whole rows: zero outside the leaf's diagonal block -- no memset of Qt by the host)
  for (int i = 0; i < m; ++i)
    for (int c = lane; c < n; c += 64) Qt[(size_t)(off + i) * n + c] = (c >= off && c < off + m) ? Z[c - off][i] : 0.0;
}

// ------------------------------------------------------------------------------------
// merge, part 1: sort, deflate, secular roots.  grid (merges of the level, slices of DC_RS roots); every
// slice repeats the (deterministic) sort + deflation and solves its own roots; slice 0 also writes the lists
// dc_merge_vectors_kernel needs.
//   lam_in / Qt_in: eigenvalues / eigenvector rows of the two children (rows off .. off + nn)
//   meta (ints per merge, stride 4): k, number of rotations
//   keep[off + i]   child row of kept slot i (ascending d);      defl[off + t] child row of deflated slot t
//   dk / zk[off+i]  poles and weights of the secular problem;    lam_out: roots, then the deflated values
//   deltaT[(off + j) n + off + i] = d_i - lam_j
// ------------------------------------------------------------------------------------
struct DcRot { int p, q; double c, s; };

__global__ __launch_bounds__(256) void dc_merge_roots_kernel(const double *__restrict__ lam_in,
                                                             const double *__restrict__ Qt_in,
                                                             const double *__restrict__ ee, int n, int depth,
                                                             double *__restrict__ lam_out, double *__restrict__ deltaT,
                                                             int *__restrict__ meta, int *__restrict__ keep,
                                                             int *__restrict__ defl, double *__restrict__ dkg,
                                                             double *__restrict__ zkg, DcRot *__restrict__ rots,
                                                             int *__restrict__ flag) {
  __shared__ double ds[DC_NMAX], zs[DC_NMAX];
  __shared__ int orig[DC_NMAX];
  __shared__ int kidx[DC_NMAX], didx[DC_NMAX];
  __shared__ double redd[8];
  __shared__ int sh_k, sh_nd, sh_nrot, sh_need;
  __shared__ int wcnt[4][2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int mg = blockIdx.x, slice = blockIdx.y;
  if (*(volatile int *)flag) return;      // an earlier stage gave up: the lists below would be built from garbage
  int off, nn;
  dc_segment(n, depth, mg, off, nn);
  const int n1 = nn / 2;
  const double emid = ee[off + n1 - 1];
  const double rho = 2.0 * fabs(emid);
  const double sgn = emid >= 0.0 ? 1.0 : -1.0;
  // ---- z and d of the children, rank sort ascending (ties by child row) ----
  double dv[DC_NMAX / 256], zv[DC_NMAX / 256];
#pragma unroll
  for (int q = 0; q < DC_NMAX / 256; ++q) {
    const int c = t + 256 * q;
    dv[q] = 0.0; zv[q] = 0.0;
    if (c < nn) {
      dv[q] = lam_in[off + c];
      const double zr = c < n1 ? Qt_in[(size_t)(off + c) * n + off + n1 - 1] : sgn * Qt_in[(size_t)(off + c) * n + off + n1];
      zv[q] = zr * 0.70710678118654752440;
      ds[c] = dv[q];          // unsorted copy for the ranking
    }
  }
  {
    bool bad = false;
#pragma unroll
    for (int q = 0; q < DC_NMAX / 256; ++q)
      bad = bad || !(fabs(dv[q]) <= 1.7976931348623157e308) || !(fabs(zv[q]) <= 1.7976931348623157e308);
    if (__syncthreads_or(bad ? 1 : 0)) {   // ranks of NaNs collide: no sort, no lists
      if (t == 0) atomicOr(flag, 4);
      return;
    }
  }
  int rk[DC_NMAX / 256];
#pragma unroll
  for (int q = 0; q < DC_NMAX / 256; ++q) {
    const int c = t + 256 * q;
    int rank = 0;
    if (c < nn) {
      const double mine = dv[q];
      for (int o = 0; o < nn; ++o) {
        const double other = ds[o];
        rank += (other < mine || (other == mine && o < c)) ? 1 : 0;
      }
    }
    rk[q] = rank;
  }
  __syncthreads();
  double dmax = 0.0, zmax = 0.0;
#pragma unroll
  for (int q = 0; q < DC_NMAX / 256; ++q) {
    const int c = t + 256 * q;
    if (c < nn) {
      ds[rk[q]] = dv[q];
      zs[rk[q]] = zv[q];
      orig[rk[q]] = c;
      dmax = fmax(dmax, fabs(dv[q]));
      zmax = fmax(zmax, fabs(zv[q]));
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    dmax = fmax(dmax, __shfl_xor(dmax, o));
    zmax = fmax(zmax, __shfl_xor(zmax, o));
  }
  if (lane == 0) { redd[wave] = dmax; redd[4 + wave] = zmax; }
  if (t == 0) { sh_need = 0; sh_nrot = 0; }
  __syncthreads();
  dmax = fmax(fmax(redd[0], redd[1]), fmax(redd[2], redd[3]));
  zmax = fmax(fmax(redd[4], redd[5]), fmax(redd[6], redd[7]));
  const double tol = 8.0 * DC_EPS * fmax(dmax, zmax);
  int k = 0, nd = 0;
  if (rho * zmax <= tol) {
    // nothing couples: every child pair is an eigenpair of the merged problem
    for (int i = t; i < nn; i += 256) didx[i] = i;
    k = 0; nd = nn;
    __syncthreads();
  } else {
    // ---- can any neighbouring pair of coupled entries be rotated?  (if not, deflation is the type-1 test alone) ----
    for (int i = t; i < nn; i += 256) {
      if (rho * fabs(zs[i]) > tol) {
        int p = i - 1;
        while (p >= 0 && rho * fabs(zs[p]) <= tol) --p;
        if (p >= 0) {
          const double zi = zs[i], zp = zs[p];
          const double cs = zi * zp / (zi * zi + zp * zp);
          if (fabs((ds[i] - ds[p]) * cs) <= tol) sh_need = 1;
        }
      }
    }
    __syncthreads();
    if (!sh_need) {
      // compaction by ballots: kept entries ascending, deflated entries ascending
      int basek = 0, based = 0;
      for (int i0 = 0; i0 < nn; i0 += 256) {
        const int i = i0 + t;
        const bool valid = i < nn;
        const bool kp = valid && rho * fabs(zs[i]) > tol;
        const bool df = valid && !kp;
        const unsigned long long mk = __ballot(kp), md = __ballot(df);
        if (lane == 0) { wcnt[wave][0] = __popcll(mk); wcnt[wave][1] = __popcll(md); }
        __syncthreads();
        int pk = basek, pd = based;
        for (int w = 0; w < wave; ++w) { pk += wcnt[w][0]; pd += wcnt[w][1]; }
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        if (kp) kidx[pk + __popcll(mk & below)] = i;
        if (df) didx[pd + __popcll(md & below)] = i;
        basek += wcnt[0][0] + wcnt[1][0] + wcnt[2][0] + wcnt[3][0];
        based += wcnt[0][1] + wcnt[1][1] + wcnt[2][1] + wcnt[3][1];
        __syncthreads();
      }
      k = basek; nd = based;
    } else {
      // ---- sequential scan (dlaed2): rotate z_pj into z_i when that perturbs the matrix by <= tol ----
      if (t == 0) {
        int kk = 0, dn = 0, nr = 0, pj = -1;
        for (int i = 0; i < nn; ++i) {
          if (rho * fabs(zs[i]) <= tol) { didx[dn++] = i; continue; }
          if (pj < 0) { pj = i; continue; }
          const double zp = zs[pj], zi = zs[i];
          const double tau = sqrt(zp * zp + zi * zi);
          const double c = zi / tau, s = -zp / tau;
          if (fabs((ds[i] - ds[pj]) * c * s) <= tol) {
            zs[i] = tau;
            zs[pj] = 0.0;
            const double dp = ds[pj] * c * c + ds[i] * s * s;
            const double di = ds[pj] * s * s + ds[i] * c * c;
            ds[pj] = dp;
            ds[i] = di;
            if (slice == 0) {
              DcRot rt;
              rt.p = orig[pj]; rt.q = orig[i]; rt.c = c; rt.s = s;
              rots[off + nr] = rt;
            }
            nr++;
            didx[dn++] = pj;
            pj = i;
          } else {
            kidx[kk++] = pj;
            pj = i;
          }
        }
        kidx[kk++] = pj;
        sh_k = kk; sh_nd = dn; sh_nrot = nr;
      }
      __syncthreads();
      k = sh_k; nd = sh_nd;
    }
  }
  const int nrot = sh_nrot;
  // ---- lists for the second kernel ----
  if (slice == 0) {
    if (t == 0) { meta[4 * mg] = k; meta[4 * mg + 1] = nrot; }
    for (int i = t; i < k; i += 256) {
      keep[off + i] = orig[kidx[i]];
      dkg[off + i] = ds[kidx[i]];
      zkg[off + i] = zs[kidx[i]];
    }
    for (int i = t; i < nd; i += 256) {
      defl[off + i] = orig[didx[i]];
      lam_out[off + k + i] = ds[didx[i]];
    }
  }
  if (slice * DC_RS >= k) return;
  __syncthreads();
  // compact poles / weights (reuse the rank-sort scratch: dv is dead)
  double *dk = ds, *z2 = zs;
  {
    double tmpd[DC_NMAX / 256], tmpz[DC_NMAX / 256];
#pragma unroll
    for (int q = 0; q < DC_NMAX / 256; ++q) {
      const int i = t + 256 * q;
      tmpd[q] = 0.0; tmpz[q] = 0.0;
      if (i < k) { tmpd[q] = ds[kidx[i]]; tmpz[q] = zs[kidx[i]]; }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < DC_NMAX / 256; ++q) {
      const int i = t + 256 * q;
      if (i < k) { dk[i] = tmpd[q]; z2[i] = tmpz[q] * tmpz[q]; }
    }
    __syncthreads();
  }
  // ---- secular roots: DC_G lanes per root ----
  const int g = t & (DC_G - 1);
  const int j = slice * DC_RS + (t / DC_G);
  const bool active = j < k;
  const int jj = active ? j : k - 1;          // inactive groups shadow the last root (results discarded)
  const int jn = jj + 1 < k ? jj + 1 : k - 1;
  const bool last = jj == k - 1;
  auto gsum = [&](double x) {   // sum over the 16 lanes of a root (a DPP row); every lane gets it
    x += dpp_f64<0xB1>(x);
    x += dpp_f64<0x4E>(x);
    x += dpp_f64<0x141>(x);
    x += dpp_f64<0x140>(x);
    return x;
  };
  double zz = 0.0;
  for (int i = g; i < k; i += DC_G) zz += z2[i];
  zz = gsum(zz);
  const double dj = dk[jj];
  const double gap = last ? rho * zz : dk[jn] - dj;
  const double mid = 0.5 * gap;
  int org = jj;
  double lo = 0.0, hi = mid, mu;
  if (k == 1) {
    mu = rho * z2[0];
    lo = hi = mu;
  } else {
    if (!last) {
      double f = 0.0;
      for (int i = g; i < k; i += DC_G) f += z2[i] * dc_rcp((dk[i] - dj) - mid);
      f = fma(rho, gsum(f), 1.0);
      if (!(f >= 0.0)) { org = jn; lo = -mid; hi = 0.0; }
    } else {
      hi = gap;
    }
    mu = last ? 0.5 * gap : (org == jj ? 0.5 * hi : 0.5 * lo);
  }
  const double dorg = dk[org];
  const double polea = dj - dorg, poleb = dk[jn] - dorg;
  bool done = k == 1;
  int it = 0;
  for (; it < 100; ++it) {
    if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
    double psi = 0.0, phi = 0.0, dpsi = 0.0, dphi = 0.0;
    for (int i = g; i < k; i += DC_G) {
      const double inv = dc_rcp((dk[i] - dorg) - mu);
      const double tq = z2[i] * inv, tq2 = tq * inv;
      if (i <= jj) { psi += tq; dpsi += tq2; }
      else { phi += tq; dphi += tq2; }
    }
    psi = rho * gsum(psi); phi = rho * gsum(phi);
    dpsi = rho * gsum(dpsi); dphi = rho * gsum(dphi);
    const double gv = 1.0 + psi + phi;
    if (!done) {
      const bool fin = fabs(gv) <= 1.7976931348623157e308;
      if (fin) { if (gv > 0.0) hi = mu; else lo = mu; }
      if (!fin || fabs(gv) <= 8.0 * DC_EPS * (1.0 + fabs(psi) + fabs(phi))) done = true;
      if (hi - lo <= 2.0 * DC_EPS * fmax(fabs(lo), fabs(hi))) done = true;
    }
    if (!done) {
      const double a = polea - mu, b = poleb - mu;
      const double s_psi = dpsi * a * a, r_psi = psi - dpsi * a;
      double eta;
      if (last) {
        eta = a + s_psi * dc_rcp(1.0 + r_psi);
      } else {
        const double s_phi = dphi * b * b, r_phi = phi - dphi * b;
        const double c = 1.0 + r_psi + r_phi;
        const double B = c * (a + b) + s_psi + s_phi;
        const double C = a * b * gv;
        const double disc = fmax(B * B - 4.0 * c * C, 0.0);
        const double sq = sqrt(disc);
        const double q = 0.5 * (B + (B >= 0.0 ? sq : -sq));
        const double e1 = q / c, e2 = C / q;
        const bool in1 = e1 > a && e1 < b, in2 = e2 > a && e2 < b;
        eta = in1 ? e1 : e2;
        if (in1 && in2 && fabs(e2) < fabs(e1)) eta = e2;
      }
      double cand = mu + eta;
      if (!(cand > lo && cand < hi)) cand = 0.5 * (lo + hi);
      mu = cand;
    }
  }
  if (!done) atomicOr(flag, 2);
  if (active) {
    for (int i = g; i < k; i += DC_G) deltaT[(size_t)(off + j) * n + off + i] = (dk[i] - dorg) - mu;
    if (g == 0) lam_out[off + j] = dorg + mu;
  }
}

// ------------------------------------------------------------------------------------
// merge, part 2: grid (merges of the level, slices of DC_VS columns).  Every workgroup recomputes the Gu-Eisenstat
// z (products of k ratios per entry, split over the threads in chunks of j and combined through LDS), then
// normalises its own columns of the rank-one update's eigenvector matrix, one wave per column -> UmatT[out row]
// [child row] of the level (the dense GEMM Qt_out = UmatT Qt_in forms the merged eigenvectors; every slice zeroes its own
// rows of the block first).  Slice 0 also applies the deflation rotations to the children's eigenvector rows and places the
// deflated columns.
// ------------------------------------------------------------------------------------
constexpr int DC_VS = 64;

__global__ __launch_bounds__(1024) void dc_merge_vectors_kernel(double *__restrict__ Qt_in, int n, int depth,
                                                                const double *__restrict__ deltaT,
                                                                const int *__restrict__ meta,
                                                                const int *__restrict__ keep,
                                                                const int *__restrict__ defl,
                                                                const double *__restrict__ dkg,
                                                                const double *__restrict__ zkg,
                                                                const DcRot *__restrict__ rots,
                                                                double *__restrict__ UmatT, const int *flag) {
  __shared__ double dk[DC_NMAX], zh[DC_NMAX];
  __shared__ int kp[DC_NMAX];
  if (*(volatile const int *)flag) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int mg = blockIdx.x, slice = blockIdx.y;
  int off, nn;
  dc_segment(n, depth, mg, off, nn);
  const int k = meta[4 * mg], nrot = meta[4 * mg + 1];
  if (slice == 0) {
    for (int rix = 0; rix < nrot; ++rix) {
      const DcRot rt = rots[off + rix];
      double *rp = Qt_in + (size_t)(off + rt.p) * n + off, *rq = Qt_in + (size_t)(off + rt.q) * n + off;
      for (int c = t; c < nn; c += 1024) {
        const double qp = rp[c], qq = rq[c];
        rp[c] = fma(rt.c, qp, rt.s * qq);
        rq[c] = fma(-rt.s, qp, rt.c * qq);
      }
    }
  }
  // this slice's rows of UmatT start from zero over their whole length (round 4: the host's memset of the matrix per level
  // is gone; a row belongs to exactly one merge of a level, so the rows' owners cover the matrix); deflated rows get
  // their single 1.  The __syncthreads below orders the zeros in front of the column entries other waves write into
  // the same rows.
  {
    const int r0 = slice * DC_VS, r1 = min(nn, r0 + DC_VS);
    for (int e = t; e < (r1 - r0) * n; e += 1024) {
      const int r = r0 + e / n, c = e % n - off;
      UmatT[(size_t)(off + r) * n + off + c] = (r >= k && c == defl[off + r - k]) ? 1.0 : 0.0;
    }
  }
  if (slice * DC_VS >= k) return;
  for (int i = t; i < k; i += 1024) { dk[i] = dkg[off + i]; kp[i] = keep[off + i]; zh[i] = 1.0; }
  __syncthreads();
  // zhat_i^2 = |prod_j (lam_j - d_i) / (d_j - d_i)| (j != i in the denominator): thread (i, chunk) multiplies the
  // ratios of j = chunk, chunk + C, ...; ratio by ratio the running product stays O(1) (interlacing)
  {
    const int kpad = (k + 63) & ~63;
    const int C = 1024 / kpad > 0 ? 1024 / kpad : 1;
    for (int i0 = 0; i0 < kpad; i0 += 1024) {          // one pass unless k > 1024 / C
      const int i = i0 + t % kpad, c = t / kpad;
      double prod = 1.0;
      if (i < k && c < C) {
        const double di = dk[i];
        for (int j = c; j < k; j += C) {
          const double num = -deltaT[(size_t)(off + j) * n + off + i];          // lam_j - d_i
          const double den = j == i ? 1.0 : dk[j] - di;
          prod *= num * dc_rcp(den);
        }
      }
      // combine the chunks: zh[i] *= prod, one chunk at a time
      for (int cc = 0; cc < C; ++cc) {
        if (i < k && c == cc) zh[i] *= prod;
        __syncthreads();
      }
    }
  }
  for (int i = t; i < k; i += 1024) {
    const double z = sqrt(fabs(zh[i]));
    zh[i] = zkg[off + i] >= 0.0 ? z : -z;
  }
  __syncthreads();
  for (int jj = wave; jj < DC_VS; jj += 16) {
    const int j = slice * DC_VS + jj;
    if (j >= k) break;
    const double *drow = deltaT + (size_t)(off + j) * n + off;
    double u[DC_NMAX / 64];
    double ss = 0.0;
#pragma unroll
    for (int q = 0; q < DC_NMAX / 64; ++q) {
      const int i = lane + 64 * q;
      u[q] = 0.0;
      if (i < k) {
        u[q] = zh[i] * dc_rcp(drow[i]);
        ss = fma(u[q], u[q], ss);
      }
    }
    ss = wave_sum_f64(ss);
    const double inv = dc_rsqrt(ss);
    double *urow = UmatT + (size_t)(off + j) * n + off;
#pragma unroll
    for (int q = 0; q < DC_NMAX / 64; ++q) {
      const int i = lane + 64 * q;
      if (i < k) urow[kp[i]] = u[q] * inv;
    }
  }
}

// ------------------------------------------------------------------------------------
// back-transformation: eigenvector rows y <- H_0 H_1 ... H_{n-3} y, H_j = I - tau_j v_j v_j^T (v_j = row j of Vh).
// Reflector by reflector this is a chain of n - 2 dependent (dot -> wave reduction -> update) steps per row.  In
// blocks of HB = 8 (compact WY, LAPACK dlarft): H_j0 ... H_j0+7 = I - V T V^T with an 8 x 8 upper-triangular T, so
// a block costs 8 INDEPENDENT dots, one tiny T z and one update: 95 -> 21 us at n = 200, 318 -> 52 us at n = 512.
//   householder_T_kernel     T of every block (Gram matrix of its 8 reflectors -> dlarft recurrence)
//   householder_rows_kernel  two rows per wave; the V block is staged through LDS (next block prefetched into
//                            registers while the current one is applied); eigenvalues are unscaled on the way
// ------------------------------------------------------------------------------------
constexpr int HB = 8;

__global__ __launch_bounds__(256) void householder_T_kernel(const double *__restrict__ Vh, const double *__restrict__ tau,
                                                            int n, double *__restrict__ Tg) {
  __shared__ double S[HB][HB];
  __shared__ double part[4][HB * (HB + 1) / 2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int j0 = blockIdx.x * HB;
  double acc[HB * (HB + 1) / 2];
#pragma unroll
  for (int e = 0; e < HB * (HB + 1) / 2; ++e) acc[e] = 0.0;
  for (int k = t; k < n; k += 256) {
    double v[HB];
#pragma unroll
    for (int i = 0; i < HB; ++i) {
      const int j = j0 + i;
      v[i] = (j <= n - 3 && k > j) ? Vh[(size_t)j * n + k] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < HB; ++i)
#pragma unroll
      for (int c = 0; c <= i; ++c) acc[i * (i + 1) / 2 + c] = fma(v[i], v[c], acc[i * (i + 1) / 2 + c]);
  }
#pragma unroll
  for (int e = 0; e < HB * (HB + 1) / 2; ++e) {
    const double sum = wave_sum_f64(acc[e]);
    if (lane == 0) part[wave][e] = sum;
  }
  __syncthreads();
  if (t < HB * (HB + 1) / 2) {
    const double sum = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    int i = 0;
    while ((i + 1) * (i + 2) / 2 <= t) ++i;
    const int c = t - i * (i + 1) / 2;
    S[i][c] = sum;
    S[c][i] = sum;
  }
  __syncthreads();
  if (t == 0) {
    double T[HB][HB];
    for (int i = 0; i < HB; ++i)
      for (int c = 0; c < HB; ++c) T[i][c] = 0.0;
    for (int i = 0; i < HB; ++i) {
      const int j = j0 + i;
      const double ti = j <= n - 3 ? tau[j] : 0.0;
      // T[0:i, i] = -tau_i T[0:i, 0:i] (V[:, 0:i]^T v_i)
      for (int r = 0; r < i; ++r) {
        double sum = 0.0;
        for (int c = r; c < i; ++c) sum = fma(T[r][c], S[c][i], sum);
        T[r][i] = -ti * sum;
      }
      T[i][i] = ti;
    }
    for (int i = 0; i < HB; ++i)
      for (int c = 0; c < HB; ++c) Tg[(size_t)blockIdx.x * HB * HB + i * HB + c] = T[i][c];
  }
}

template <int E>
__global__ __launch_bounds__(256) void householder_rows_kernel(const double *__restrict__ Qt, int n,
                                                               const double *__restrict__ Vh,
                                                               const double *__restrict__ Tg,
                                                               const double *__restrict__ lam_in,
                                                               const double *__restrict__ scale,
                                                               double *__restrict__ Vout, double *__restrict__ lam_out) {
  constexpr int R = 2;                       // rows per wave
  constexpr int NP = E * 64;                 // padded row length
  constexpr int PRE = HB * NP / 256;         // V-block elements staged by one thread
  extern __shared__ __attribute__((aligned(16))) double hh_sm[];
  double *vt = hh_sm;                        // [HB][NP]
  double *Tt = hh_sm + HB * NP;              // [HB][HB]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int row0 = (blockIdx.x * 4 + wave) * R;
  double y[R][E];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const int c = lane + 64 * q;
      y[r][q] = (row0 + r < n && c < n) ? Qt[(size_t)(row0 + r) * n + c] : 0.0;
    }
  const int ntiles = n >= 3 ? (n - 2 + HB - 1) / HB : 0;
  // The V blocks come from the tridiagonalisation's workgroup on another XCD: a load is ~2 us away while a block's
  // arithmetic is ~0.4 us, so the blocks are requested DEPTH ahead (a ring of register sets, the loop unrolled over
  // it so that every set has static register names).  One block ahead, the loop ran at the load latency: 25 blocks x
  // 2.3 us = 59 us at n = 200.
  constexpr int DEPTH = PRE <= 8 ? 6 : (PRE <= 16 ? 3 : 1);
  double pre[DEPTH][PRE];
  double tpre[DEPTH];
#pragma unroll
  for (int s = 0; s < DEPTH; ++s) tpre[s] = 0.0;
  auto prefetch = [&](int b, double (&pr)[PRE], double &tp) {
#pragma unroll
    for (int e = 0; e < PRE; ++e) {
      const int idx = t + 256 * e, i = idx / NP, k = idx % NP, j = b * HB + i;
      pr[e] = (j <= n - 3 && k > j && k < n) ? Vh[(size_t)j * n + k] : 0.0;
    }
    if (t < HB * HB) tp = Tg[(size_t)b * HB * HB + t];
  };
#pragma unroll
  for (int s = 0; s < DEPTH; ++s)
    if (ntiles - 1 - s >= 0) prefetch(ntiles - 1 - s, pre[s], tpre[s]);
  for (int b0 = ntiles - 1; b0 >= 0; b0 -= DEPTH) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      const int b = b0 - s;
      if (b < 0) break;
      __syncthreads();                         // the previous block has been applied
#pragma unroll
      for (int e = 0; e < PRE; ++e) vt[t + 256 * e] = pre[s][e];
      if (t < HB * HB) Tt[t] = tpre[s];
      __syncthreads();
      if (b - DEPTH >= 0) prefetch(b - DEPTH, pre[s], tpre[s]);
      double z[R][HB];
#pragma unroll
      for (int i = 0; i < HB; ++i) {
        double vv[E];
#pragma unroll
        for (int q = 0; q < E; ++q) vv[q] = vt[i * NP + lane + 64 * q];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          double d = 0.0;
#pragma unroll
          for (int q = 0; q < E; ++q) d = fma(vv[q], y[r][q], d);
          z[r][i] = d;
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < HB; ++i) z[r][i] = wave_sum_f64(z[r][i]);
      double u[R][HB];
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < HB; ++i) {
          double sum = 0.0;
#pragma unroll
          for (int c = i; c < HB; ++c) sum = fma(Tt[i * HB + c], z[r][c], sum);
          u[r][i] = sum;
        }
#pragma unroll
      for (int i = 0; i < HB; ++i) {
#pragma unroll
        for (int q = 0; q < E; ++q) {
          const double vq = vt[i * NP + lane + 64 * q];
#pragma unroll
          for (int r = 0; r < R; ++r) y[r][q] = fma(-u[r][i], vq, y[r][q]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (row0 + r >= n) continue;
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const int c = lane + 64 * q;
      if (c < n) Vout[(size_t)(row0 + r) * n + c] = y[r][q];
    }
    if (lane == 0) lam_out[row0 + r] = lam_in[row0 + r] * scale[1];
  }
}

// Eight wave-wide sums at once (the eight dots of a compact-WY block).  Eight separate wave_sum_f64 are 8 x 4 DPP steps
// plus 8 x 4 readlanes; here each exchange step also halves the number of live values: lanes whose bit b is 0 keep the
// even member of a pair and send the odd one, and the other way round (quad_perm xor 1, xor 2, row_ror:4 -- a rotation
// pairs every lane with one of the other bit-2 class, which is all the sum needs), so that after three steps a lane
// holds ONE value, the partial sum of z[lane & 7]; row_ror:8 and the two row-swap instructions of gfx950
// (v_permlane16_swap / v_permlane32_swap: 16- and 32-lane exchanges in the register file) finish it.  7 + 3 exchange
// steps instead of 32; every lane ends with the total of z[lane & 7].
__device__ __forceinline__ double wave_sum8_by_class(const double (&z)[8], int lane) {
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0;
  double w[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const double keep = b0 ? z[2 * p + 1] : z[2 * p], send = b0 ? z[2 * p] : z[2 * p + 1];
    w[p] = keep + dpp_f64<0xB1>(send);          // quad_perm [1,0,3,2]
  }
  double x[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const double keep = b1 ? w[2 * p + 1] : w[2 * p], send = b1 ? w[2 * p] : w[2 * p + 1];
    x[p] = keep + dpp_f64<0x4E>(send);          // quad_perm [2,3,0,1]
  }
  double y;
  {
    const double keep = b2 ? x[1] : x[0], send = b2 ? x[0] : x[1];
    y = keep + dpp_f64<0x124>(send);            // row_ror:4
  }
  y += dpp_f64<0x128>(y);                       // row_ror:8
  {
    const unsigned lo = (unsigned)__double2loint(y), hi = (unsigned)__double2hiint(y);
    const auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    y = __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
  }
  {
    const unsigned lo = (unsigned)__double2loint(y), hi = (unsigned)__double2hiint(y);
    const auto l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    y = __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
  }
  return y;
}

// householder_rows_kernel for n <= 512 (round 4): ONE row per wave.  The back-transformation is a chain of ceil((n - 2) / 8)
// blocks, and with 256 CUs for n / 2 waves what counts is the length of a block in one wave's instruction stream, not the
// number of waves: a row per wave halves the dots and updates, the eight dots end in wave_sum8_by_class (about 75
// instructions instead of 370), and the block's V entries stay in registers between the dots and the update.
// 58 -> about 30 us at n = 200.
template <int E>
__global__ __launch_bounds__(256) void householder_row1_kernel(const double *__restrict__ Qt, int n,
                                                               const double *__restrict__ Vh,
                                                               const double *__restrict__ Tg,
                                                               const double *__restrict__ lam_in,
                                                               const double *__restrict__ scale,
                                                               double *__restrict__ Vout, double *__restrict__ lam_out) {
  constexpr int NP = E * 64;                 // padded row length
  constexpr int PRE = HB * NP / 256;         // V-block elements staged by one thread
  extern __shared__ __attribute__((aligned(16))) double hh_sm[];
  double *vt = hh_sm;                        // [HB][NP]
  double *Tt = hh_sm + HB * NP;              // [HB][HB]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int row = blockIdx.x * 4 + wave;
  double y[E];
#pragma unroll
  for (int q = 0; q < E; ++q) {
    const int c = lane + 64 * q;
    y[q] = (row < n && c < n) ? Qt[(size_t)row * n + c] : 0.0;
  }
  const int ntiles = n >= 3 ? (n - 2 + HB - 1) / HB : 0;
  constexpr int DEPTH = PRE <= 8 ? 6 : 3;
  double pre[DEPTH][PRE];
  double tpre[DEPTH];
#pragma unroll
  for (int s = 0; s < DEPTH; ++s) tpre[s] = 0.0;
  auto prefetch = [&](int b, double (&pr)[PRE], double &tp) {
#pragma unroll
    for (int e = 0; e < PRE; ++e) {
      const int idx = t + 256 * e, i = idx / NP, k = idx % NP, j = b * HB + i;
      pr[e] = (j <= n - 3 && k > j && k < n) ? Vh[(size_t)j * n + k] : 0.0;
    }
    if (t < HB * HB) tp = Tg[(size_t)b * HB * HB + t];
  };
#pragma unroll
  for (int s = 0; s < DEPTH; ++s)
    if (ntiles - 1 - s >= 0) prefetch(ntiles - 1 - s, pre[s], tpre[s]);
  for (int b0 = ntiles - 1; b0 >= 0; b0 -= DEPTH) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      const int b = b0 - s;
      if (b < 0) break;
      __syncthreads();                         // the previous block has been applied
#pragma unroll
      for (int e = 0; e < PRE; ++e) vt[t + 256 * e] = pre[s][e];
      if (t < HB * HB) Tt[t] = tpre[s];
      __syncthreads();
      if (b - DEPTH >= 0) prefetch(b - DEPTH, pre[s], tpre[s]);
      double vv[HB][E], z[HB];
#pragma unroll
      for (int i = 0; i < HB; ++i) {
#pragma unroll
        for (int q = 0; q < E; ++q) vv[i][q] = vt[i * NP + lane + 64 * q];
        double d = 0.0;
#pragma unroll
        for (int q = 0; q < E; ++q) d = fma(vv[i][q], y[q], d);
        z[i] = d;
      }
      const double zc = wave_sum8_by_class(z, lane);      // lane l: the total of z[l & 7]
      double zt[HB];
#pragma unroll
      for (int i = 0; i < HB; ++i) zt[i] = readlane_f64(zc, i);
#pragma unroll
      for (int i = 0; i < HB; ++i) {
        double u = 0.0;
#pragma unroll
        for (int c = i; c < HB; ++c) u = fma(Tt[i * HB + c], zt[c], u);
#pragma unroll
        for (int q = 0; q < E; ++q) y[q] = fma(-u, vv[i][q], y[q]);
      }
    }
  }
  if (row < n) {
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const int c = lane + 64 * q;
      if (c < n) Vout[(size_t)row * n + c] = y[q];
    }
    if (lane == 0) lam_out[row] = lam_in[row] * scale[1];
  }
}

}  // namespace

// eig_sort_kernel of linalg.hip (rank sort descending, optional floor at zero, row permutation)
int eig_sort_rows(plda_handle *h, const double *lam, const double *V, int D, double *s, double *Vsorted);

int sym_eig_dc_status(plda_handle *h, int *status) {
  *status = 8;
  if (!h->eigdc_flag) return PLDA_OK;
  int hflag = 0;
  PLDA_HIP(h, hipMemcpyAsync(&hflag, h->eigdc_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  *status = hflag;
  return PLDA_OK;
}

// Returns PLDA_OK with *status = 0 when the decomposition is in s / Vrows, *status != 0 when the direct method
// gave up (non-finite input, an iteration cap): the caller then falls back to the Jacobi solver.  G is not modified.
int sym_eig_dc_f64(plda_handle *h, const double *G, int D, double *s, double *Vrows, int *status) {
  if (status) *status = 0;
  h->eigdc_flag = nullptr;
  if (D > DC_NMAX) {
    if (status) *status = 8;
    return status ? PLDA_OK : fail(h, PLDA_E_INVAL, "sym_eig_dc: D=%d > %d", D, DC_NMAX);
  }
  const int n = D;
  const size_t DD = (size_t)n * n;
  // workspace: Vh, QtA, QtB, UmatT, deltaT (n^2 each), then vectors and lists
  const size_t vec = (size_t)round_up(n, 32);
  const size_t need = DD * 8 * 5 + vec * 8 * 10 + 8 * (size_t)DC_NMAX * 8 + 64 + vec * 4 * 4 + vec * sizeof(DcRot) + (8 * vec + 64) * 8 + 4096;
  PLDA_HIP(h, h->eigdc.reserve(need));
  double *Vh = h->eigdc.as<double>();
  double *QtA = Vh + DD, *QtB = QtA + DD, *UmatT = QtB + DD, *deltaT = UmatT + DD;
  double *dd = deltaT + DD, *ee = dd + vec, *tau = ee + vec, *lamA = tau + vec, *lamB = lamA + vec;
  double *dkg = lamB + vec, *zkg = dkg + vec, *lamU = zkg + vec, *scale = lamU + vec;   // scale: 2 doubles
  unsigned long long *words = reinterpret_cast<unsigned long long *>(scale + vec);   // all-gather: [parity][4][<= DC_NMAX]
  int *keep = reinterpret_cast<int *>(words + 8 * DC_NMAX);
  int *defl = keep + vec, *meta = defl + vec, *flag = meta + vec;                      // meta: 4 ints per merge (<= 64 merges)
  DcRot *rots = reinterpret_cast<DcRot *>(flag + vec);
  double *Tg = reinterpret_cast<double *>(rots + vec);   // compact-WY T blocks: ceil((n - 2) / 8) x 64 doubles
  TraceScope ts(h, "getoutput.eig.tridiagonalise", 4.0 / 3.0 * (double)n * n * n, 1);
  const bool ev0 = h->eig_variant == 0 || h->eig_variant == 4;   // (4: the default kernels with the two-row back-transformation)
  const bool full_kernel = h->sweep_variant == 0 && ev0 && n > 32 && n <= 208;   // (scales its input and clears the status word itself)
  if (!full_kernel) {
    PLDA_HIP(h, hipMemsetAsync(flag, 0, sizeof(int), h->stream));
    PLDA_HIP(h, hipMemsetAsync(scale + 2, 0, 8, h->stream));   // running max |g_ij| (as bits)
    eig_absmax_kernel<<<(unsigned)std::min<int64_t>(ceil_div((int64_t)DD, 1024), 128), 256, 0, h->stream>>>(
        G, n, reinterpret_cast<unsigned long long *>(scale + 2), flag);
    eig_scale_kernel<<<1, 1, 0, h->stream>>>(reinterpret_cast<unsigned long long *>(scale + 2), scale, flag);
  }
  // tridiagonalisation: one workgroup with the matrix in registers while that fits without spilling (n <= 160),
  // else rows over ceil(n / 8) cooperating workgroups (PLDA_EIG_VARIANT=2 / 3 force one or the other)
  const bool reg_kernel = h->eig_variant == 2 ? n <= 256 : (h->eig_variant == 3 ? false : n <= 160);
  // round 3: the four-wave register kernel up to n = 224 (PLDA_SWEEP_VARIANT=1 or PLDA_EIG_VARIANT=2 / 3: the round-2 choice)
  if (full_kernel) {   // full storage, one barrier per step (PLDA_SWEEP_VARIANT=2: the symmetric-storage kernel)
    const int nb = (int)ceil_div(n, 16);
#define TRF(NBB) tridiag_full_kernel<NBB, (NBB + 1) / 2><<<1, 512, 0, h->stream>>>(G, n, scale, dd, ee, Vh, tau, flag)
    switch (nb) {
      case 3: TRF(3); break;
      case 4: TRF(4); break;
      case 5: TRF(5); break;
      case 6: TRF(6); break;
      case 7: TRF(7); break;
      case 8: TRF(8); break;
      case 9: TRF(9); break;
      case 10: TRF(10); break;
      case 11: TRF(11); break;
      case 12: TRF(12); break;
      default: TRF(13); break;
    }
#undef TRF
  } else if (h->sweep_variant != 1 && ev0 && n <= 224) {      // (NB = 15, 16 spill 380 / 680 bytes per lane)
    const int nb = (int)ceil_div(n, 16);
#define TR16(NBB) tridiag_reg16_kernel<NBB><<<1, 256, 0, h->stream>>>(G, n, scale, dd, ee, Vh, tau)
    switch (nb) {
      case 1: TR16(1); break;
      case 2: TR16(2); break;
      case 3: TR16(3); break;
      case 4: TR16(4); break;
      case 5: TR16(5); break;
      case 6: TR16(6); break;
      case 7: TR16(7); break;
      case 8: TR16(8); break;
      case 9: TR16(9); break;
      case 10: TR16(10); break;
      case 11: TR16(11); break;
      case 12: TR16(12); break;
      case 13: TR16(13); break;
      case 14: TR16(14); break;
      case 15: TR16(15); break;
      default: TR16(16); break;
    }
#undef TR16
  } else if (reg_kernel) {
    const int nb = (int)ceil_div(n, 32);
#define TR(NBB) tridiag_reg_kernel<NBB><<<1, 1024, 0, h->stream>>>(G, n, scale, dd, ee, Vh, tau)
    switch (nb) {
      case 1: TR(1); break;
      case 2: TR(2); break;
      case 3: TR(3); break;
      case 4: TR(4); break;
      case 5: TR(5); break;
      case 6: TR(6); break;
      case 7: TR(7); break;
      default: TR(8); break;
    }
#undef TR
  } else {
    const int E = (int)ceil_div(n, 32);
    const int NP = (E <= 7 ? 7 : E <= 8 ? 8 : E <= 16 ? 16 : E <= 32 ? 32 : 64) * 32;
    PLDA_HIP(h, hipMemsetAsync(words, 0, (size_t)8 * NP * sizeof(unsigned long long), h->stream));
    int nn = n;
    const double *Gp = G;
    const double *scp = scale;
    int dbg = n >= 64 ? h->eig_debug : (h->eig_debug & ~2);   // the stamps go to the (then free) deltaT: 1 KiB
    int W = (int)ceil_div(n, TR_ROWS);
    long long *tl = reinterpret_cast<long long *>(deltaT);   // debug stamps (deltaT is free until the merges)
    void *args[] = {&Gp, &nn, &W, &scp, &dd, &ee, &Vh, &tau, &words, &flag, &dbg, &tl};
    const void *fn = E <= 7    ? reinterpret_cast<const void *>(&tridiag_rows_kernel<7, TR_ROWS>)
                     : E <= 8  ? reinterpret_cast<const void *>(&tridiag_rows_kernel<8, TR_ROWS>)
                     : E <= 16 ? reinterpret_cast<const void *>(&tridiag_rows_kernel<16, TR_ROWS>)
                     : E <= 32 ? reinterpret_cast<const void *>(&tridiag_rows_kernel<32, TR_ROWS>)
                               : reinterpret_cast<const void *>(&tridiag_rows_kernel<64, TR_ROWS>);
    {
      // a device that cannot hold all W workgroups at once (CU masking, a partitioned GPU) refuses the launch:
      // that is not an error of the fit -- the caller falls back to the block Jacobi solver
      const hipError_t ce = hipLaunchCooperativeKernel(fn, dim3(W), dim3(TR_ROWS * 32), args, 0, h->stream);
      if (ce != hipSuccess) {
        (void)hipGetLastError();
        h->eigdc_flag = nullptr;         // sym_eig_dc_status then reports 8 ("not handled")
        if (status) *status = 8;
        return PLDA_OK;
      }
    }
    if (dbg & 2) {
      long long st[16 * 8];
      PLDA_HIP(h, hipStreamSynchronize(h->stream));
      PLDA_HIP(h, hipMemcpy(st, tl, sizeof(st), hipMemcpyDeviceToHost));
      for (int r = 0; r < 16; ++r) {
        std::fprintf(stderr, "tridiag step %2d:", 16 + r);
        for (int c = 1; c < 8; ++c) std::fprintf(stderr, " %6lld", st[r * 8 + c] - st[r * 8 + c - 1]);
        if (r < 15) std::fprintf(stderr, "  | next %6lld", st[(r + 1) * 8] - st[r * 8 + 7]);
        std::fprintf(stderr, "\n");
      }
    }
  }
  PLDA_LAUNCH_CHECK(h);
  ts.next("getoutput.eig.divide_and_conquer");
  int depth = 0;
  while ((int)ceil_div(n, 1 << depth) > DC_LEAF) depth++;
  dc_leaf_kernel<<<1 << depth, 64, 0, h->stream>>>(dd, ee, n, depth, lamA, QtA, flag);
  PLDA_LAUNCH_CHECK(h);
  double *qin = QtA, *qout = QtB, *lin = lamA, *lout = lamB;
  for (int dl = depth - 1; dl >= 0; --dl) {
    const int merges = 1 << dl;
    const int maxn = (int)ceil_div(n, merges);
    const int slices = (int)ceil_div(maxn, DC_RS);
    dc_merge_roots_kernel<<<dim3(merges, slices), 256, 0, h->stream>>>(lin, qin, ee, n, dl, lout, deltaT, meta, keep,
                                                                        defl, dkg, zkg, rots, flag);
    dc_merge_vectors_kernel<<<dim3(merges, (unsigned)ceil_div(maxn, DC_VS)), 1024, 0, h->stream>>>(qin, n, dl, deltaT, meta, keep, defl,
                                                                                             dkg, zkg, rots, UmatT, flag);
    PLDA_LAUNCH_CHECK(h);
    PLDA_TRY(gemm_f64(h, n, n, n, 1.0, UmatT, n, 1, qin, n, 1, nullptr, 0.0, qout, n));
    std::swap(qin, qout);
    std::swap(lin, lout);
  }
  ts.next("getoutput.eig.back_transform", 2.0 * (double)n * n * n, 1);
  {
    const int E = (int)ceil_div(n, 64);
    const int EE = E <= 1 ? 1 : E <= 2 ? 2 : E <= 4 ? 4 : E <= 8 ? 8 : E <= 16 ? 16 : 32;
    const int ntiles = n >= 3 ? (n - 2 + HB - 1) / HB : 0;
    if (ntiles) householder_T_kernel<<<ntiles, 256, 0, h->stream>>>(Vh, tau, n, Tg);
    const unsigned grid = (unsigned)ceil_div(n, 8);
    const size_t lds = ((size_t)HB * EE * 64 + HB * HB) * sizeof(double);
#define HR(E2)                                                                                                         \
  do {                                                                                                                 \
    PLDA_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&householder_rows_kernel<E2>),                       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                            \
    householder_rows_kernel<E2><<<grid, 256, lds, h->stream>>>(qin, n, Vh, Tg, lin, scale, qout, lamU);               \
  } while (0)
    // n <= 512: one row per wave, the eight dots of a block reduced together (PLDA_EIG_VARIANT=4: the two-row kernel)
#define HR1(E2)                                                                                                        \
  do {                                                                                                                 \
    PLDA_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&householder_row1_kernel<E2>),                       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                            \
    householder_row1_kernel<E2><<<(unsigned)ceil_div(n, 4), 256, lds, h->stream>>>(qin, n, Vh, Tg, lin, scale, qout, lamU); \
  } while (0)
    const bool row1 = h->eig_variant != 4;
    if (EE == 1 && row1) HR1(1);
    else if (EE == 2 && row1) HR1(2);
    else if (EE == 4 && row1) HR1(4);
    else if (EE == 8 && row1) HR1(8);
    else if (EE == 1) HR(1);
    else if (EE == 2) HR(2);
    else if (EE == 4) HR(4);
    else if (EE == 8) HR(8);
    else if (EE == 16) HR(16);
    else HR(32);
#undef HR
#undef HR1
  }
  PLDA_LAUNCH_CHECK(h);
  PLDA_TRY(eig_sort_rows(h, lamU, qout, n, s, Vrows));
  ts.close();
  h->eigdc_flag = flag;
  if (status) PLDA_TRY(sym_eig_dc_status(h, status));   // status == nullptr: the caller reads it later (sym_eig_dc_status)
  return PLDA_OK;
}

}  // namespace plda
