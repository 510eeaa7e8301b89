// plda_amd/csrc/score.hip -- batched PLDA log-likelihood-ratio scoring on gfx950.
//
// Replaces the per-trial path MPlda_score -> Plda::LogLikelihoodRatio
// (/root/reference/src/pldamodule.cpp:258-277, driven M x Nt times by the nested
// Python loop of scoring/scorePLDA.py:302-318) and the z-norm statistics of
// MPlda_norm (:196-256) with the algebraically identical GEMM form
// (SURVEY.md Appendix A.5):
//
//   S_ij = sum_d A1_id v_jd  [+ sum_d A2_id v_jd^2]  + r_i  [+ q_j]
//   A1 = c u / var,  A2 = -1/2 (1/var - 1/(1+psi)),  c = n psi/(n psi + 1),
//   var = 1 + psi/(n psi + 1),
//   r_i = -1/2 sum_d [log var - log(1+psi) + c^2 u^2 / var]
//   uniform n:  q_j = -1/2 sum_d (1/var_d - 1/(1+psi_d)) v_jd^2  (GEMM depth D)
//   mixed   n:  second half of the contraction carries A2 x V*V   (GEMM depth 2D)
//
// Bias terms and operand values are computed in fp64 and rounded once to fp32;
// the contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains).
//
// HBM layout of the GEMM operands ("k-quad packed"): P[kq][row][4] fp32, i.e.
// for each group of 4 consecutive k the rows are contiguous 16-byte items.  With
// that layout (1) a wave's global->LDS DMA (global_load_lds_dwordx4) of 64 rows is
// one contiguous 1 KiB burst and lands lane-linear in LDS, and (2) the MFMA
// fragment read is one conflict-free ds_read_b128 per lane that feeds 4 MFMAs.
#include "common.hpp"
#include <atomic>
#include <chrono>
#include <cstring>

#include <algorithm>

namespace plda {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// ------------------------------------------------------------------------------------
// per-row coefficient helpers (fp64)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void llr_coef(double n, double psi, double &c, double &var) {
  const double den = n * psi + 1.0;
  c = n * psi / den;
  var = 1.0 + psi / den;
}

// r_i: one wave per enrol row.  Also (optionally) folds the z-norm affine map
// s -> (s - zmean)/zstd into (rscale, rbias): out = rscale * (acc + q) + rbias.
__global__ void enrol_bias_kernel(const double *__restrict__ U, const int32_t *__restrict__ n_arr,
                                  int n_uniform, const double *__restrict__ psi, int D, int64_t M,
                                  const double *__restrict__ zmean, const double *__restrict__ zstd,
                                  float *__restrict__ rbias, float *__restrict__ rscale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= M) return;
  const double n = n_arr ? (double)n_arr[row] : (double)n_uniform;
  const double *u = U + row * (int64_t)D;
  double acc = 0.0;
  for (int d = lane; d < D; d += 64) {
    double c, var;
    const double p = psi[d];
    llr_coef(n, p, c, var);
    const double cu = c * u[d];
    acc += log(var) - log(1.0 + p) + cu * cu / var;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) {
    double r = -0.5 * acc, sc = 1.0;
    if (zmean && zstd) {
      const double sd = zstd[row];
      if (sd != 0.0) { sc = 1.0 / sd; r = (r - zmean[row]) * sc; }
    }
    rbias[row] = (float)r;
    if (rscale) rscale[row] = (float)sc;
  }
}

// Uniform enrol count: the per-dimension coefficients do not depend on the row, so they are
// computed once per call (one workgroup): w_d = c^2/var, g_d = 1/var - 1/(1+psi) and
// L = sum_d [log var - log(1+psi)]; the bias kernels then are plain weighted sums of squares.
// Mixed counts in the bucketed form: one workgroup per DISTINCT count, tables back to back.
__device__ __forceinline__ void count_coef(const double *__restrict__ psi, int D, int n, double *__restrict__ coef /*[2*D + 1]: w, g, L*/) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int d = threadIdx.x; d < D; d += 256) {
    double c, var;
    const double p = psi[d];
    llr_coef((double)n, p, c, var);
    coef[d] = c * c / var;
    coef[D + d] = 1.0 / var - 1.0 / (1.0 + p);
    acc += log(var) - log(1.0 + p);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) coef[2 * D] = red[0];
}
__global__ __launch_bounds__(256) void uniform_coef_kernel(const double *__restrict__ psi, int D, int n_uniform, double *__restrict__ coef) {
  count_coef(psi, D, n_uniform, coef);
}
__global__ __launch_bounds__(256) void bucket_coef_kernel(const double *__restrict__ psi, int D, const CountSet cs, double *__restrict__ coef /*[G][2*D + 1]*/) {
  count_coef(psi, D, cs.vals[blockIdx.x], coef + (size_t)blockIdx.x * (2 * D + 1));
}

// out[row] = scale_row * (-1/2) * (L + sum_d w_d x_d^2) (+ z-norm folding), one wave per row
__global__ void weighted_sq_bias_kernel(const double *__restrict__ X, const double *__restrict__ w, double Lconst_on,
                                        const double *__restrict__ Lptr, int D, int64_t R,
                                        const double *__restrict__ zmean, const double *__restrict__ zstd,
                                        float *__restrict__ bias, float *__restrict__ rscale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= R) return;
  const double *x = X + row * (int64_t)D;
  double acc = 0.0;
  for (int d = lane; d < D; d += 64) acc += w[d] * x[d] * x[d];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) {
    double r = -0.5 * (acc + (Lconst_on != 0.0 ? *Lptr : 0.0)), sc = 1.0;
    if (zmean && zstd) {
      const double sd = zstd[row];
      if (sd != 0.0) { sc = 1.0 / sd; r = (r - zmean[row]) * sc; }
    }
    bias[row] = (float)r;
    if (rscale) rscale[row] = (float)sc;
  }
}

// The 256 x 256 kernel forms its starting value r'_i + s_i q_j with one rank-2 MFMA per accumulator
// (k = 0: r'_i x 1, k = 1: s_i x q_j), so it wants the biases as k-pairs:
//   rpair[row] = (r'_i, s_i)   cpair[col] = (1, q_j)
// where r'_i = r_i, s_i = 1 without z-norm and r'_i = (r_i - zmean_i) / zstd_i, s_i = 1 / zstd_i with it.
__global__ void bias_pairs_kernel(const float *__restrict__ rbias, const float *__restrict__ rscale, int64_t M,
                                  const float *__restrict__ cbias, int64_t Nt, float2 *__restrict__ rpair,
                                  float2 *__restrict__ cpair) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < M) rpair[t] = make_float2(rbias[t], rscale[t]);
  if (t < Nt) cpair[t] = make_float2(1.f, cbias[t]);
}

// ------------------------------------------------------------------------------------
// pack: fp64 [R, D] row-major  ->  fp32 k-quad packed P[kq][Rpad][4], through an LDS
// transpose so that both the fp64 reads (256 B per half-wave) and the packed
// writes (1 KiB per wave) are coalesced.
//   MODE 0: enrol A1 (= c u / var), k in [0, Dp)
//   MODE 1: enrol [A1 | A2], k in [0, 2 Dp)       (mixed n)
//   MODE 2: test  V
//   MODE 3: test  [V | V*V]                        (mixed n)
//   rs (nullable, MODE 0/1): per-row scale folded into the A operand (z-norm)
// ------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void pack_kernel(const double *__restrict__ X,
                                                   const int32_t *__restrict__ n_arr,
                                                   int n_uniform, const double *__restrict__ psi,
                                                   const float *__restrict__ rs, int D, int Dp,
                                                   int64_t R, int64_t Rpad, int KQ,
                                                   float *__restrict__ P) {
  __shared__ float tile[32][65];
  const int64_t row0 = (int64_t)blockIdx.x * 64;
  const int k0 = blockIdx.y * 32;
  const int t = threadIdx.x;
  {
    const int c = t & 31;
    const int k = k0 + c;
    const bool second = (MODE == 1 || MODE == 3) && k >= Dp;
    const int d = second ? k - Dp : k;
    const bool dvalid = d < D && k < ((MODE == 1 || MODE == 3) ? 2 * Dp : Dp);
    const double p = dvalid ? psi[d] : 0.0;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int r = (t >> 5) + pass * 8;
      const int64_t row = row0 + r;
      float val = 0.f;
      if (dvalid && row < R) {
        const double x = X[row * (int64_t)D + d];
        if (MODE == 0 || MODE == 1) {
          const double n = n_arr ? (double)n_arr[row] : (double)n_uniform;
          double cc, var;
          llr_coef(n, p, cc, var);
          double v = second ? -0.5 * (1.0 / var - 1.0 / (1.0 + p)) : cc * x / var;
          if (rs) v *= (double)rs[row];
          val = (float)v;
        } else {
          val = (float)(second ? x * x : x);
        }
      }
      tile[c][r] = val;
    }
  }
  __syncthreads();
  {
    const int r = t & 63;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int q = (t >> 6) + pass * 4;  // local kq 0..7
      const int kq = (k0 >> 2) + q;
      if (kq < KQ) {
        f32x4 v;
        v.x = tile[4 * q + 0][r];
        v.y = tile[4 * q + 1][r];
        v.z = tile[4 * q + 2][r];
        v.w = tile[4 * q + 3][r];
        reinterpret_cast<f32x4 *>(P)[(int64_t)kq * Rpad + row0 + r] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// prep_side_kernel (round 4): weighted_sq_bias + pack + bias_pairs of ONE side in ONE pass over the
// fp64 rows, for the uniform-count path (every BASELINE configuration but C4).  The three kernels above
// read a side's rows twice (2 x R D 8 B) -- at C2 0.21 ms of a 28.5 ms step, more than the trials GEMM
// has left to give.  A workgroup owns 64 rows; it walks the row in chunks of 64 dimensions: wave w
// loads its 16 rows (one 512-byte piece per row and chunk, the next chunk's loads in flight under this
// one's arithmetic), adds the lane's weighted square to the row's running sum -- lane l holds the
// partial over d = l, l + 64, ... in increasing d, exactly the order of weighted_sq_bias_kernel, and the
// same xor butterfly ends it: the biases are BIT-identical to the two-pass path (tests) -- and drops
// the fp32 operand value into a 64 x 64 LDS tile, from which the k-quad planes leave as 1 KiB pieces.
//   SIDE 0: enrol  A1 = c u / var (* s_i), r'_i, s_i, rpair        SIDE 1: test  V, q_j, cpair
// HBM: R D 8 B read + R Kg 4 B written (Kg = D rounded up to 8).
// RW = rows per wave: 16 (above) for the long sides, where 64-row workgroups fill the chip several times over; 4 for sides of
// up to 32 768 rows, where they do not -- 8 192 rows are 128 workgroups of lone waves walking their chunks one memory round
// trip after the other: 17.4 us whatever the row count (256 rows x 512: 25 us), a seventh of an 8192 x 8192 x 200 call.  With
// 16-row workgroups (256-byte pieces of the planes) the same arithmetic in the same order (bit-identical, tests) takes
// PLDA_PREP_VARIANT=2 / 3: force 16 / 4.
// ------------------------------------------------------------------------------------
template <int SIDE, int RW>
__device__ __forceinline__ void prep_side_body(const int block, const double *__restrict__ X, const double *__restrict__ w, const double *__restrict__ Lptr,
                                               int n_uniform, const double *__restrict__ psi, int D, int64_t R, int64_t Rpad, int KQ,
                                               const double *__restrict__ zmean, const double *__restrict__ zstd,
                                               float *__restrict__ P, float *__restrict__ bias, float *__restrict__ rscale,
                                               float2 *__restrict__ pair) {
  constexpr int RPB = 4 * RW;                 // rows of a workgroup
  __shared__ float tile[2][64][RPB + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = (int64_t)block * RPB;
  const int64_t wrow0 = row0 + wave * RW;
  const int nchunk = (KQ * 4 + 63) >> 6;
  const bool zn = SIDE == 0 && zmean && zstd;
  // per-row scale of the packed A operand (the z-norm map's 1 / zstd_i, rounded to fp32 as pack_kernel reads it)
  float rsf[RW];
#pragma unroll
  for (int j = 0; j < RW; ++j) {
    rsf[j] = 1.f;
    if (zn && wrow0 + j < R) { const double sd = zstd[wrow0 + j]; rsf[j] = (float)(sd != 0.0 ? 1.0 / sd : 1.0); }
  }
  // PF chunks are fetched together, one group ahead of the arithmetic: 1 for the 16-row waves (the next chunk's 16 loads under
  // this chunk's arithmetic), 4 for the 4-row waves of short sides (a row of up to 256 dimensions is ONE memory round trip)
  constexpr int PF = RW == 16 ? 1 : 4;
  double acc[RW], xc[PF][RW], xn[PF][RW];
#pragma unroll
  for (int j = 0; j < RW; ++j) acc[j] = 0.0;
  auto fetch = [&](int c, double (&x)[RW]) {
    const int d = c * 64 + lane;
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      const int64_t row = wrow0 + j;
      x[j] = (c < nchunk && d < D && row < R) ? X[row * (int64_t)D + d] : 0.0;
    }
  };
  auto chunk = [&](int c, const double (&x)[RW]) {
    const int d = c * 64 + lane;
    const bool dv = d < D;
    const double wd = dv ? w[d] : 0.0;
    double cc = 0.0, var = 1.0;
    if (SIDE == 0 && dv) llr_coef((double)n_uniform, psi[d], cc, var);
    float(*const tl)[RPB + 1] = tile[c & 1];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      const double xv = x[j];
      float val = 0.f;
      if (dv && wrow0 + j < R) {
        acc[j] += wd * xv * xv;
        if (SIDE == 0) {
          double v = cc * xv / var;
          if (zn) v *= (double)rsf[j];
          val = (float)v;
        } else {
          val = (float)xv;
        }
      }
      tl[lane][wave * RW + j] = val;
    }
    __syncthreads();
    {
      const int r = threadIdx.x % RPB;
#pragma unroll
      for (int pass = 0; pass < RPB / 16; ++pass) {
        const int q = threadIdx.x / RPB + pass * (256 / RPB);   // k-quad of the chunk, 0..15
        const int kq = c * 16 + q;
        if (kq < KQ) {
          f32x4 v;
          v.x = tl[4 * q + 0][r];
          v.y = tl[4 * q + 1][r];
          v.z = tl[4 * q + 2][r];
          v.w = tl[4 * q + 3][r];
          reinterpret_cast<f32x4 *>(P)[(int64_t)kq * Rpad + row0 + r] = v;
        }
      }
    }
  };
#pragma unroll
  for (int k = 0; k < PF; ++k) fetch(k, xn[k]);
  for (int c0 = 0; c0 < nchunk; c0 += PF) {
#pragma unroll
    for (int k = 0; k < PF; ++k)
#pragma unroll
      for (int j = 0; j < RW; ++j) xc[k][j] = xn[k][j];
    if (c0 + PF < nchunk) {
#pragma unroll
      for (int k = 0; k < PF; ++k) fetch(c0 + PF + k, xn[k]);
    }
#pragma unroll
    for (int k = 0; k < PF; ++k)
      if (c0 + k < nchunk) chunk(c0 + k, xc[k]);
  }
  const double Lc = SIDE == 0 ? *Lptr : 0.0;
#pragma unroll
  for (int j = 0; j < RW; ++j) {
    double a = acc[j];
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    const int64_t row = wrow0 + j;
    if (lane == 0 && row < R) {
      double r = -0.5 * (a + Lc), sc = 1.0;
      if (zn) {
        const double sd = zstd[row];
        if (sd != 0.0) { sc = 1.0 / sd; r = (r - zmean[row]) * sc; }
      }
      bias[row] = (float)r;
      if (SIDE == 0) { rscale[row] = (float)sc; pair[row] = make_float2((float)r, (float)sc); }
      else pair[row] = make_float2(1.f, (float)r);
    }
  }
}

template <int SIDE, int RW>
__global__ __launch_bounds__(256) void prep_side_kernel(const double *__restrict__ X, const double *__restrict__ w, const double *__restrict__ Lptr,
                                                        int n_uniform, const double *__restrict__ psi, int D, int64_t R, int64_t Rpad, int KQ,
                                                        const double *__restrict__ zmean, const double *__restrict__ zstd,
                                                        float *__restrict__ P, float *__restrict__ bias, float *__restrict__ rscale,
                                                        float2 *__restrict__ pair) {
  prep_side_body<SIDE, RW>((int)blockIdx.x, X, w, Lptr, n_uniform, psi, D, R, Rpad, KQ, zmean, zstd, P, bias, rscale, pair);
}
// both short sides in one launch (a dependent launch costs ~4.5 us whatever it does): blocks [0, blocksA) the enrol side, the rest the test side
struct PrepSideArgs { const double *X; const double *w; int64_t R, Rpad; float *P; float *bias; float2 *pair; };
__global__ __launch_bounds__(256) void prep_both_kernel(const PrepSideArgs a, const PrepSideArgs b, int blocksA, const double *__restrict__ Lptr, int n_uniform,
                                                        const double *__restrict__ psi, int D, int KQ, const double *__restrict__ zmean,
                                                        const double *__restrict__ zstd, float *__restrict__ rscale) {
  if ((int)blockIdx.x < blocksA)
    prep_side_body<0, 4>((int)blockIdx.x, a.X, a.w, Lptr, n_uniform, psi, D, a.R, a.Rpad, KQ, zmean, zstd, a.P, a.bias, rscale, a.pair);
  else
    prep_side_body<1, 4>((int)blockIdx.x - blocksA, b.X, b.w, Lptr, 0, psi, D, b.R, b.Rpad, KQ, nullptr, nullptr, b.P, b.bias, nullptr, b.pair);
}

// ------------------------------------------------------------------------------------
// Mixed enrol counts at GEMM depth D + G - 1 (round 5; SURVEY.md Appendix A.5 collapses the second operand half only
// when ALL counts are equal, and rounds 1-4 followed that: any count array -> depth 2 D).  A2_i = -1/2 (1/var - 1/(1+psi))
// depends on the row only through n_i, so with the G distinct counts n_(0) < n_(1) < ... ("buckets")
//     S_ij = A1_i . v_j + r_i + q_(b_i)[j],      q_g[j] = -1/2 sum_d (1/var_d(n_(g)) - 1/(1+psi_d)) v_jd^2
// -- one column-bias vector per DISTINCT count.  Bucket 0's rides where the uniform path keeps its q_j (the rank-2 bias
// MFMA: r'_i x 1 + s_i x q_0[j]); the others enter as DIFFERENCES dq_g = q_g - q_0 behind the D columns of the
// contraction: the enrol side carries s_i x onehot(b_i - 1), the test side dq_1 .. dq_(G-1), padded to whole 8-k steps.
// No row permutation, no new epilogue, the GEMM kernels are untouched; G = 1 IS the uniform path.  BASELINE C4
// (n in 1..5, D = 256): depth 264 instead of 512.  Counts above CS_NMAX, more than CS_MAX distinct ones, or G - 1 > D / 2
// fall back to the depth-2D form (pack_kernel<1> / <3>), which stays as the A/B arm (PLDA_MIXED_VARIANT=1).
// Reference: Plda::LogLikelihoodRatio through MPlda_score, /root/reference/src/pldamodule.cpp:258-277.
// ------------------------------------------------------------------------------------
struct CountSetDev { int G; int overflow; int32_t vals[CS_MAX]; };

__global__ void count_presence_kernel(const int32_t *__restrict__ n, int64_t M, unsigned char *__restrict__ present /*[CS_NMAX + 1]*/,
                                      int *__restrict__ flags) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
    const int v = n[i];
    if (v >= 1 && v <= CS_NMAX) present[v] = 1;     // (racing stores of the same byte value)
    else flags[0] = 1;
  }
}

// presence bytes -> ascending list of the distinct counts (one workgroup; lane l of wave 0 owns counts [64 l, 64 l + 64))
__global__ __launch_bounds__(256) void count_compact_kernel(const unsigned char *__restrict__ present, const int *__restrict__ flags,
                                                            CountSetDev *__restrict__ out) {
  __shared__ unsigned char pr[CS_NMAX + 1];
  for (int i = threadIdx.x; i <= CS_NMAX; i += 256) pr[i] = present[i];
  __syncthreads();
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  int cnt = 0;
  for (int k = 0; k < 64; ++k) cnt += pr[lane * 64 + k] ? 1 : 0;
  int incl = cnt;
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
  const int total = __shfl(incl, 63);
  int pos = incl - cnt;
  for (int k = 0; k < 64; ++k)
    if (pr[lane * 64 + k]) { if (pos < CS_MAX) out->vals[pos] = lane * 64 + k; ++pos; }
  if (lane == 0) { out->G = total; out->overflow = (flags[0] != 0 || total > CS_MAX) ? 1 : 0; }
}

// Both steps in ONE workgroup for up to CS_ONE_WG counts, the result written straight into the caller's pinned host words
// behind a system-scope fence, `seq` last: the host needs the set before it can size the operands, and memset + two
// kernels + a copy + a stream synchronisation were 40 us of a 154 us call at 4096 x 4096 x 200 with counts 1..5.
constexpr int64_t CS_ONE_WG = 131072;
struct CountSetHost { CountSetDev set; int seq; };
__global__ __launch_bounds__(1024) void count_set_one_wg_kernel(const int32_t *__restrict__ n, int M, int seq, CountSetHost *__restrict__ out /*pinned host*/) {
  __shared__ unsigned char pr[CS_NMAX + 1];
  __shared__ int bad;
  for (int i = threadIdx.x; i <= CS_NMAX; i += 1024) pr[i] = 0;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  for (int i0 = 0; i0 < M; i0 += 4096) {
    int v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = i0 + u * 1024 + (int)threadIdx.x; v[u] = i < M ? n[i] : -1; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (v[u] >= 1 && v[u] <= CS_NMAX) pr[v[u]] = 1;            // (racing stores of the same byte value)
      else if (i0 + u * 1024 + (int)threadIdx.x < M) bad = 1;
    }
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  int cnt = 0;
  for (int k = 0; k < 64; ++k) cnt += pr[lane * 64 + k] ? 1 : 0;
  int incl = cnt;
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
  const int total = __shfl(incl, 63);
  int pos = incl - cnt;
  for (int k = 0; k < 64; ++k)
    if (pr[lane * 64 + k]) { if (pos < CS_MAX) out->set.vals[pos] = lane * 64 + k; ++pos; }
  if (lane == 0) { out->set.G = total; out->set.overflow = (bad != 0 || total > CS_MAX) ? 1 : 0; }
  __threadfence_system();
  if (lane == 0) __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// enrol side of the bucketed form, one pass over the fp64 rows (the structure of prep_side_kernel<0>; the coefficients
// are the row's own): A1 = c u / var (* s_i), r'_i, s_i, rpair, and the KQx extra k-quad planes s_i x onehot(b_i - 1)
template <int RW>
__device__ __forceinline__ void prep_enrol_buckets_body(const int block, const double *__restrict__ X, const int32_t *__restrict__ n_arr, const CountSet &cs,
                                                        const double *__restrict__ coefG /*[G][2 D + 1]*/, const double *__restrict__ psi, int D,
                                                        int64_t R, int64_t Rpad, int KQm, int KQx, const double *__restrict__ zmean,
                                                        const double *__restrict__ zstd, float *__restrict__ P, float *__restrict__ bias,
                                                        float *__restrict__ rscale, float2 *__restrict__ pair) {
  constexpr int RPB = 4 * RW;                 // rows of a workgroup (RW rows per wave: 16, or 4 for short sides -- see prep_side_kernel)
  __shared__ float tile[2][64][RPB + 1];
  __shared__ int sb[RPB];
  __shared__ float ss[RPB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = (int64_t)block * RPB;
  const int64_t wrow0 = row0 + wave * RW;
  const int nchunk = (KQm * 4 + 63) >> 6;
  const bool zn = zmean && zstd;
  const int S = 2 * D + 1;
  float rsf[RW];
  double nn[RW], Lr[RW];
#pragma unroll
  for (int j = 0; j < RW; ++j) {
    const int64_t row = wrow0 + j;
    rsf[j] = 1.f;
    if (zn && row < R) { const double sd = zstd[row]; rsf[j] = (float)(sd != 0.0 ? 1.0 / sd : 1.0); }
    const int n = row < R ? n_arr[row] : cs.vals[0];
    int b = 0;
    for (int g = 1; g < cs.G; ++g) b = cs.vals[g] == n ? g : b;
    nn[j] = (double)n;
    // a count that is NOT in the set handed over (the host builds the set from these very counts; a caller-supplied set has to
    // cover every slab): the row's bias becomes NaN -- a row of NaN scores, not quietly the scores of another count
    Lr[j] = (b == 0 && cs.vals[0] != n) ? __longlong_as_double(0x7ff8000000000000ll) : coefG[(size_t)b * S + 2 * D];
    if (lane == 0) { sb[wave * RW + j] = row < R ? b : 0; ss[wave * RW + j] = rsf[j]; }
  }
  double acc[RW], xc[RW], xn[RW];
#pragma unroll
  for (int j = 0; j < RW; ++j) { acc[j] = 0.0; xn[j] = 0.0; }
  auto fetch = [&](int c, double (&x)[RW]) {
    const int d = c * 64 + lane;
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      const int64_t row = wrow0 + j;
      x[j] = (d < D && row < R) ? X[row * (int64_t)D + d] : 0.0;
    }
  };
  fetch(0, xn);
  for (int c = 0; c < nchunk; ++c) {
#pragma unroll
    for (int j = 0; j < RW; ++j) xc[j] = xn[j];
    if (c + 1 < nchunk) fetch(c + 1, xn);
    const int d = c * 64 + lane;
    const bool dv = d < D;
    const double p = dv ? psi[d] : 0.0;
    float(*const tl)[RPB + 1] = tile[c & 1];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      const double x = xc[j];
      float val = 0.f;
      if (dv && wrow0 + j < R) {
        double cc, var;
        llr_coef(nn[j], p, cc, var);
        acc[j] += (cc * cc / var) * x * x;
        double v = cc * x / var;
        if (zn) v *= (double)rsf[j];
        val = (float)v;
      }
      tl[lane][wave * RW + j] = val;
    }
    __syncthreads();
    {
      const int r = threadIdx.x % RPB;
#pragma unroll
      for (int pass = 0; pass < RPB / 16; ++pass) {
        const int q = threadIdx.x / RPB + pass * (256 / RPB);   // k-quad of the chunk, 0..15
        const int kq = c * 16 + q;
        if (kq < KQm) {
          f32x4 v;
          v.x = tl[4 * q + 0][r];
          v.y = tl[4 * q + 1][r];
          v.z = tl[4 * q + 2][r];
          v.w = tl[4 * q + 3][r];
          reinterpret_cast<f32x4 *>(P)[(int64_t)kq * Rpad + row0 + r] = v;
        }
      }
    }
  }
  // (sb / ss were written before the loop's first barrier; nchunk >= 1)
  {
    const int r = threadIdx.x % RPB;
    const int pcol = sb[r] - 1;
    for (int e = threadIdx.x / RPB; e < KQx; e += 256 / RPB) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pcol >= 0 && (pcol >> 2) == e) v[pcol & 3] = ss[r];
      reinterpret_cast<f32x4 *>(P)[(int64_t)(KQm + e) * Rpad + row0 + r] = v;
    }
  }
#pragma unroll
  for (int j = 0; j < RW; ++j) {
    double a = acc[j];
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    const int64_t row = wrow0 + j;
    if (lane == 0 && row < R) {
      double r = -0.5 * (a + Lr[j]), sc = 1.0;
      if (zn) {
        const double sd = zstd[row];
        if (sd != 0.0) { sc = 1.0 / sd; r = (r - zmean[row]) * sc; }
      }
      bias[row] = (float)r;
      rscale[row] = (float)sc;
      pair[row] = make_float2((float)r, (float)sc);
    }
  }
}

template <int RW>
__global__ __launch_bounds__(256) void prep_enrol_buckets_kernel(const double *__restrict__ X, const int32_t *__restrict__ n_arr, const CountSet cs,
                                                                 const double *__restrict__ coefG, const double *__restrict__ psi, int D,
                                                                 int64_t R, int64_t Rpad, int KQm, int KQx, const double *__restrict__ zmean,
                                                                 const double *__restrict__ zstd, float *__restrict__ P, float *__restrict__ bias,
                                                                 float *__restrict__ rscale, float2 *__restrict__ pair) {
  prep_enrol_buckets_body<RW>((int)blockIdx.x, X, n_arr, cs, coefG, psi, D, R, Rpad, KQm, KQx, zmean, zstd, P, bias, rscale, pair);
}

// test side of the bucketed form, the extra planes: dq_g[j] = -1/2 sum_d (g_gd - g_0d) v_jd^2 for g = 1 .. G - 1 (fp64, rounded
// once), zero beyond.  The D main planes, q_0 and cpair come from prep_side_kernel<1> with bucket 0's coefficients.  A
// second read of the rows (L2 mostly): Nt D 8 B against the GEMM's Nt M (D + G) flop.
// (round 5, second version: the coefficient differences of a group of four buckets live in REGISTERS for the 16 rows a wave
//  handles -- per lane eight chunks of 64 dimensions, D <= 512 -- so that the inner loop is one 8-byte load and four FMAs per
//  element; the first version re-read five coefficients from L1 per element and took 1.26 ms for C4's 2.4 GB of test rows.)
template <int RW>
__device__ __forceinline__ void prep_test_buckets_body(const int block, const double *__restrict__ V, const double *__restrict__ coefG, int G, int D,
                                                       int64_t R, int64_t Rpad, int KQm, int KQx, float *__restrict__ P) {
  constexpr int RPB = 4 * RW;
  __shared__ float q[CS_MAX][RPB + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = (int64_t)block * RPB;
  for (int i = threadIdx.x; i < CS_MAX * (RPB + 1); i += 256) (&q[0][0])[i] = 0.f;
  __syncthreads();
  const int S = 2 * D + 1;
  const int64_t wrow0 = row0 + wave * RW;
  const bool cached = D <= 512;
  for (int g0 = 1; g0 < G && wrow0 < R; g0 += 4) {
    const bool h1 = g0 + 1 < G, h2 = g0 + 2 < G, h3 = g0 + 3 < G;
    const double *c0 = coefG + (size_t)g0 * S + D;
    double dc[4][8];
    if (cached) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int d = lane + 64 * c;
        const bool ok = d < D;
        const double gb = ok ? coefG[D + d] : 0.0;
        dc[0][c] = ok ? c0[d] - gb : 0.0;
        dc[1][c] = ok && h1 ? c0[S + d] - gb : 0.0;
        dc[2][c] = ok && h2 ? c0[2 * S + d] - gb : 0.0;
        dc[3][c] = ok && h3 ? c0[3 * S + d] - gb : 0.0;
      }
    }
    for (int j = 0; j < RW; ++j) {
      const int64_t row = wrow0 + j;
      if (row >= R) break;
      const double *v = V + row * (int64_t)D;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      if (cached) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int d = lane + 64 * c;
          const double x = d < D ? v[d] : 0.0, x2 = x * x;
          a0 = fma(dc[0][c], x2, a0); a1 = fma(dc[1][c], x2, a1); a2 = fma(dc[2][c], x2, a2); a3 = fma(dc[3][c], x2, a3);
        }
      } else {
        for (int d = lane; d < D; d += 64) {
          const double x = v[d], x2 = x * x, gb = coefG[D + d];
          a0 += (c0[d] - gb) * x2;
          if (h1) a1 += (c0[S + d] - gb) * x2;
          if (h2) a2 += (c0[2 * S + d] - gb) * x2;
          if (h3) a3 += (c0[3 * S + d] - gb) * x2;
        }
      }
      a0 = wave_sum_f64(a0); a1 = wave_sum_f64(a1); a2 = wave_sum_f64(a2); a3 = wave_sum_f64(a3);
      if (lane == 0) {
        q[g0 - 1][wave * RW + j] = (float)(-0.5 * a0);
        if (h1) q[g0][wave * RW + j] = (float)(-0.5 * a1);
        if (h2) q[g0 + 1][wave * RW + j] = (float)(-0.5 * a2);
        if (h3) q[g0 + 2][wave * RW + j] = (float)(-0.5 * a3);
      }
    }
  }
  __syncthreads();
  const int r = threadIdx.x % RPB;
  for (int e = threadIdx.x / RPB; e < KQx; e += 256 / RPB) {
    f32x4 v;
    v.x = q[4 * e + 0][r];
    v.y = q[4 * e + 1][r];
    v.z = q[4 * e + 2][r];
    v.w = q[4 * e + 3][r];
    reinterpret_cast<f32x4 *>(P)[(int64_t)(KQm + e) * Rpad + row0 + r] = v;
  }
}

template <int RW>
__global__ __launch_bounds__(256) void prep_test_buckets_kernel(const double *__restrict__ V, const double *__restrict__ coefG, int G, int D,
                                                                int64_t R, int64_t Rpad, int KQm, int KQx, float *__restrict__ P) {
  prep_test_buckets_body<RW>((int)blockIdx.x, V, coefG, G, D, R, Rpad, KQm, KQx, P);
}
// the bucketed form's three packing kernels of two SHORT sides in one launch (as prep_both_kernel for the uniform path): blocks
// [0, blocksA) the enrol side; the rest the test side -- its D main planes with bucket 0's coefficients, then its dq planes
struct PrepBucketArgs { const double *U; const int32_t *n; int64_t M, Mpad; float *Apk; float *rbias; float *rscale; float2 *rpair;
                        const double *V; int64_t Nt, Npad; float *Bpk; float *cbias; float2 *cpair; };
__global__ __launch_bounds__(256) void prep_buckets_both_kernel(const PrepBucketArgs a, int blocksA, const CountSet cs, const double *__restrict__ coefG,
                                                                const double *__restrict__ psi, int D, int KQm, int KQx,
                                                                const double *__restrict__ zmean, const double *__restrict__ zstd) {
  if ((int)blockIdx.x < blocksA) {
    prep_enrol_buckets_body<4>((int)blockIdx.x, a.U, a.n, cs, coefG, psi, D, a.M, a.Mpad, KQm, KQx, zmean, zstd, a.Apk, a.rbias, a.rscale, a.rpair);
  } else {
    const int b = (int)blockIdx.x - blocksA;
    prep_side_body<1, 4>(b, a.V, coefG + D, coefG + 2 * D, 0, psi, D, a.Nt, a.Npad, KQm, nullptr, nullptr, a.Bpk, a.cbias, nullptr, a.cpair);
    prep_test_buckets_body<4>(b, a.V, coefG, cs.G, D, a.Nt, a.Npad, KQm, KQx, a.Bpk);
  }
}

// ------------------------------------------------------------------------------------
// K5 / K8: the trials GEMM, 128 x 128 form (small / medium problems and the fused z-norm
// epilogue; large EPI-0 problems take the persistent 256 x 256 form further down).
//   block  = 256 threads = 4 waves (2 x 2), block tile 128 x 128, wave tile 64 x 64
//            = 2 x 2 MFMA tiles of 32 x 32 (64 fp32 accumulators per lane)
//   stage  = NKQ k-quads (4 NKQ values of k) of both operands, global->LDS by DMA,
//            double buffered; LDS = 2 * NKQ * 4 KiB
//   step   = 8 k (two k-quads): lane (i = lane & 31, h = lane >> 5) reads the float4
//            of row i in k-quad 2p + h and issues 4 MFMAs per (tm, tn): component t
//            contracts the k pair {8p + t, 8p + 4 + t}.  (Any fixed pairing of k's
//            is a valid contraction order; A and B use the same one.)
//   grid   = 1-D, XCD-aware: block b -> XCD b % 8 (observed dispatch order), and each
//            XCD walks its own sequence of PM x PN tile patches so that the panels
//            its resident blocks share stay in that XCD's 4 MiB L2.
//   EPI 0  : out[i][j] = acc (started from the bias, z-norm map folded in); transposed through LDS,
//            16-byte non-temporal stores
//   EPI 1  : fused z-norm statistics -- per column j accumulate sum / sum of squares
//            of (score - shift_j) over rows i < M into fp64 (no score matrix)
// ------------------------------------------------------------------------------------
constexpr int PATCH_M = 8;
constexpr int PATCH_N = 16;

template <int NKQ>
__device__ __forceinline__ void stage_tiles(const f32x4 *__restrict__ Apk,
                                            const f32x4 *__restrict__ Bpk, int64_t Mpad,
                                            int64_t Npad, int kq0, int KQ, int64_t r0,
                                            int64_t c0, f32x4 *buf, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < NKQ; ++j) {
    const int c = wave + 4 * j;  // chunk id in [0, 4 NKQ)
    const bool isB = c >= 2 * NKQ;
    const int cc = isB ? c - 2 * NKQ : c;
    const int kql = cc >> 1, half = cc & 1;
    const int kq = kq0 + kql;
    if (kq < KQ) {
      const f32x4 *g = isB ? Bpk + ((int64_t)kq * Npad + c0 + half * 64 + lane)
                           : Apk + ((int64_t)kq * Mpad + r0 + half * 64 + lane);
      f32x4 *l = buf + (isB ? NKQ * 128 : 0) + kql * 128 + half * 64;
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)g, (LDS_AS void *)l, 16, 0, 0);
    }
  }
}

// ------------------------------------------------------------------------------------
// Two details measured on MI355X (scripts/gemm_sweep.py):
//  (1) fragment registers ping-pong between two sets and a sched_barrier pins the LDS
//      reads of step p+1 ABOVE the 16 MFMAs of step p (hipcc otherwise coalesces the
//      prefetch registers with the live ones, which forces the reads below the MFMAs
//      and exposes the LDS latency every 8 k);
//  (2) the epilogue transposes each wave's 64x64 tile through its own LDS region
//      (the staging buffers are free after the last barrier) and writes 16-byte
//      non-temporal stores, 4 rows x 256 B per wave instruction: 16 store instructions
//      per wave instead of 64 (the store path is issue-bound).
// ------------------------------------------------------------------------------------
#define MFMA16(A0, A1, B0, B1)                                                                \
  _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                             \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[t], B0[t], acc[0][0], 0, 0, 0);       \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[t], B1[t], acc[0][1], 0, 0, 0);       \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[t], B0[t], acc[1][0], 0, 0, 0);       \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[t], B1[t], acc[1][1], 0, 0, 0);       \
  }

#define MFMA4(T, A0, A1, B0, B1)                                                              \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[T], B0[T], acc[0][0], 0, 0, 0);         \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[T], B1[T], acc[0][1], 0, 0, 0);         \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[T], B0[T], acc[1][0], 0, 0, 0);         \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[T], B1[T], acc[1][1], 0, 0, 0);

template <int NKQ, int EPI, int MINW, int ABL = 0>
__global__ __launch_bounds__(256, MINW) void trials_gemm_kernel(
    const f32x4 *__restrict__ Apk, const f32x4 *__restrict__ Bpk, int64_t Mpad, int64_t Npad,
    int KQ, const float *__restrict__ rbias, const float *__restrict__ rscale,
    const float *__restrict__ cbias, float *__restrict__ out, int64_t ld, int64_t M, int64_t Nt,
    int tilesM, int tilesN, int patchesN, int numPatches, const float *__restrict__ shift,
    double *__restrict__ colsum, double *__restrict__ colsq) {
  static_assert(NKQ % 2 == 0 && NKQ >= 4, "stage must hold whole 8-k steps; epilogue needs 32 KiB");
  __shared__ f32x4 smem[2 * NKQ * 256];

  const unsigned b = blockIdx.x;
  const unsigned xcd = b & 7u, seq = b >> 3;
  const unsigned per = PATCH_M * PATCH_N;
  const unsigned patch = (seq / per) * 8u + xcd;
  if (patch >= (unsigned)numPatches) return;
  const unsigned w = seq % per;
  const int tile_m = (int)(patch / patchesN) * PATCH_M + (int)(w / PATCH_N);
  const int tile_n = (int)(patch % patchesN) * PATCH_N + (int)(w % PATCH_N);
  if (tile_m >= tilesM || tile_n >= tilesN) return;
  const int64_t r0 = (int64_t)tile_m * 128, c0 = (int64_t)tile_n * 128;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 31, hh = lane >> 5;
  const int64_t wrow0 = r0 + wm * 64, wcol0 = c0 + wn * 64;

  stage_tiles<NKQ>(Apk, Bpk, Mpad, Npad, 0, KQ, r0, c0, smem, wave, lane);

  f32x4 rb[2][4];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      rb[tm][q] = *reinterpret_cast<const f32x4 *>(rbias + wrow0 + tm * 32 + 8 * q + 4 * hh);
  float cb[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) cb[tn] = cbias[wcol0 + tn * 32 + i];

  // accumulators start from the bias fma(s_i, q_j, r'_i) (s_i = 1, r'_i = r_i without z-norm; with it
  // the map (x - zmean_i) / zstd_i is folded into r', s and the A operand): the same initial value
  // and k order as the 256 x 256 kernel, hence the same bits
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 rs = *reinterpret_cast<const f32x4 *>(rscale + wrow0 + a * 32 + 8 * q + 4 * hh);
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[a][c][4 * q + e] = __builtin_fmaf(rs[e], cb[c], rb[a][q][e]);
    }

  const int nst = (KQ + NKQ - 1) / NKQ;
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    const f32x4 *cur = smem + (st & 1) * (NKQ * 256);
    if (st + 1 < nst && !(ABL & 2))
      stage_tiles<NKQ>(Apk, Bpk, Mpad, Npad, (st + 1) * NKQ, KQ, r0, c0,
                       smem + ((st + 1) & 1) * (NKQ * 256), wave, lane);
    const int np = min(NKQ, KQ - st * NKQ) >> 1;
    const f32x4 *Al = cur + hh * 128 + wm * 64 + i;
    const f32x4 *Bl = cur + NKQ * 128 + hh * 128 + wn * 64 + i;
    f32x4 xa0 = Al[0], xa1 = Al[32], xb0 = Bl[0], xb1 = Bl[32];
    f32x4 ya0, ya1, yb0, yb1;
    int p = 0;
    // The next step's LDS reads are issued after the first 4 MFMAs of a block, so the
    // (conservative, full) wait the compiler places before a block's first MFMA only ever
    // covers reads issued >= 12 MFMAs (768 cycles) earlier.
#pragma unroll 1
    for (; p + 1 < np; p += 2) {
      MFMA4(0, xa0, xa1, xb0, xb1)
      __builtin_amdgcn_sched_barrier(0);
      const int o1 = (p + 1) * 256;
      if (ABL & 1) { ya0 = xa1; ya1 = xa0; yb0 = xb1; yb1 = xb0; }
      else { ya0 = Al[o1]; ya1 = Al[o1 + 32]; yb0 = Bl[o1]; yb1 = Bl[o1 + 32]; }
      __builtin_amdgcn_sched_barrier(0);
      MFMA4(1, xa0, xa1, xb0, xb1) MFMA4(2, xa0, xa1, xb0, xb1) MFMA4(3, xa0, xa1, xb0, xb1)
      MFMA4(0, ya0, ya1, yb0, yb1)
      __builtin_amdgcn_sched_barrier(0);
      const int o2 = min(p + 2, np - 1) * 256;
      if (ABL & 1) { xa0 = ya1; xa1 = ya0; xb0 = yb1; xb1 = yb0; }
      else { xa0 = Al[o2]; xa1 = Al[o2 + 32]; xb0 = Bl[o2]; xb1 = Bl[o2 + 32]; }
      __builtin_amdgcn_sched_barrier(0);
      MFMA4(1, ya0, ya1, yb0, yb1) MFMA4(2, ya0, ya1, yb0, yb1) MFMA4(3, ya0, ya1, yb0, yb1)
    }
    if (p < np) { MFMA16(xa0, xa1, xb0, xb1) }
    if (!(ABL & 4)) __syncthreads();
  }

  if (EPI == 2) {   // ablation only
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[tm][tn][r]));
  } else if (EPI == 0) {
    // wave-private 32 x 64 fp32 staging tile (8 KiB); rows of 256 B, conflict-free both ways
    float *tw = reinterpret_cast<float *>(smem) + wave * 2048;
    const int rrow = lane >> 4, rcol = (lane & 15) * 4;     // transposed read: 4 rows x 256 B
    const bool interior = (r0 + 128 <= M) && (c0 + 128 <= Nt) && ((ld & 3) == 0) &&
                          ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          tw[((r & 3) + 8 * (r >> 2) + 4 * hh) * 64 + tn * 32 + i] = acc[tm][tn][r];
        }
      // (same wave wrote and reads: LDS operations of one wave execute in order)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int lr = rrow + 4 * k;                       // row within this 32-row half
        const f32x4 v = *reinterpret_cast<const f32x4 *>(tw + lr * 64 + rcol);
        const int64_t row = wrow0 + tm * 32 + lr;
        float *dst = out + row * ld + wcol0 + rcol;
        if (interior) {
          __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(dst));
        } else if (row < M) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (wcol0 + rcol + e < Nt) __builtin_nontemporal_store(v[e], dst + e);
        }
      }
    }
  } else {
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int64_t col = wcol0 + tn * 32 + i;
      const float sh = shift[col];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = wrow0 + 4 * hh + tm * 32 + (r & 3) + 8 * (r >> 2);
          const float d = (row < M) ? acc[tm][tn][r] - sh : 0.f;
          s1 += d;
          s2 += d * d;
        }
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (hh == 0 && col < Nt) {
        atomicAdd(colsum + col, (double)s1);
        atomicAdd(colsq + col, (double)s2);
      }
    }
  }
}

constexpr int BPR = 4, BPC = 8;   // patch of 256x256 tiles per XCD iteration (32 CUs)

// ------------------------------------------------------------------------------------
// K5, large problems (the product path of every BASELINE.json configuration): persistent 256 x 256
// form.  A workgroup of 8 waves (2 x 4) owns a 256 x 256 tile, each wave 128 x 64 (4 x 2 MFMA
// tiles, 128 accumulator registers); one workgroup per CU, persistent over tiles; tile order: at
// iteration `it` XCD x works on patch (8 it + x) = BPR x BPC tiles, its 32 workgroups one tile
// each, so an XCD's resident tiles share 4 + 8 operand panels in its L2.  Same operand layout,
// fragment scheme and k order as the 128 x 128 kernel above, so the scores are bit-identical.
//
// Pipeline (round 2; the round-1 kernel of this shape ran at 82 % MFMA-busy and lost the rest at
// the stage boundary: after `vmcnt(0) + s_barrier` all 8 waves executed the stage's DMA issue --
// ~200 scalar instructions of 64-bit addressing with SGPR spills per wave -- and then waited for
// their first fragment reads, with the matrix pipe idle):
//   * DMA addressing is two scalar adds per 1 KiB piece: per-wave byte offsets into a buffer
//     descriptor (soffset) and into LDS (M0) live in SGPRs and advance per stage;
//   * the stage barrier sits EARLY, in front of the LAST 8-k step of a stage, when that step's
//     fragments are already in registers.  Behind it every wave knows (a) stage g+1 has landed
//     (each wave drained its own DMAs first) and (b) nobody reads stage g's buffer any more -- so
//     the step's 32 MFMAs issue at once, the fragments of stage g+1's first step are fetched under
//     them, and the DMA pieces of stage g+2 go out one per two MFMAs into the buffer just freed.
//     No wave ever stands at a barrier with nothing queued behind it;
//   * the pipeline is uniform across tile boundaries: the epilogue owns a separate 16 KiB staging
//     area (2 KiB per wave, 8 x 64 outputs per round trip, software-pipelined write / read-back /
//     store), so the DMA stream and the fragment prefetch of the next tile run through it, and
//     there is no barrier at a tile boundary at all;
//   * no vector-ALU work at the tile boundary: while its partner on the SIMD is MFMA-dense, a wave's
//     VALU instructions get one issue slot per partner MFMA (~64 cycles each).  The accumulators
//     start from the bias r'_i + s_i q_j, formed by one rank-2 MFMA per accumulator on k-pairs
//     (r'_i, s_i) x (1, q_j) staged per tile by DMA; the z-norm map is folded into r', s and the A
//     operand; the epilogue only moves data (LDS transpose, scalar-addressed 16-byte stores);
//   * K is cut into balanced stages of 2..4 steps of 8 k (25 steps at D = 200 -> 4,4,4,4,3,3,3).
// LDS: 2 x 64 KiB stage buffers + 16 KiB staging + 3 x 4 KiB bias slots = 156 KiB.
// Packed operands must be < 4 GiB each (32-bit soffset) and ld < 2^22 (32-bit store offsets inside a
// tile); the host splits larger problems into column / row blocks.
// ------------------------------------------------------------------------------------
constexpr int BT2_STG = 2 * 65536;                 // byte offset of the epilogue staging area
constexpr int BT2_BIAS = BT2_STG + 16384;          // 3 slots x (256 row pairs + 256 column pairs)
constexpr int BT2_LDS = BT2_BIAS + 3 * 4096;       // 159744 B

#define BT2_SB __builtin_amdgcn_sched_barrier(0)
#define BT2_MFMA2(T, S, TM)                                                                         \
  acc[TM][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(S##a[TM][T], S##b[0][T], acc[TM][0], 0, 0, 0);  \
  acc[TM][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(S##a[TM][T], S##b[1][T], acc[TM][1], 0, 0, 0);
#define BT2_MFMA8(T, S) BT2_MFMA2(T, S, 0) BT2_MFMA2(T, S, 1) BT2_MFMA2(T, S, 2) BT2_MFMA2(T, S, 3)
#define BT2_LOAD(S, AP, BP)                                                                         \
  S##a[0] = (AP)[0]; S##a[1] = (AP)[32]; S##a[2] = (AP)[64]; S##a[3] = (AP)[96];                    \
  S##b[0] = (BP)[0]; S##b[1] = (BP)[32];
// Steps run in pairs on two alternating fragment sets: the first step of a pair computes on x and
// fetches the following step's fragments into y, the second computes on y and fetches into x.  Every
// non-MFMA instruction in this stream costs the wave ~4 cycles of MFMA issue (measured: 68 instead of
// 64 cycles per MFMA with 24 register copies + 12 other fillers per step), so nothing is copied
// except once per stage with an odd number of steps: its first step runs alone (x, fetch into y,
// y -> x).  Few bodies with fixed register roles (separately unrolled 1/2/3/4-step stage bodies with
// swapping roles made the register allocator spill the accumulators): a loop over inner pairs, then the
// stage's last pair, whose second step carries the end-of-stage work -- early barrier, next stage's
// first fragments, the DMA pieces of the stage after that.
#define BT2_XYC(T)                                                                                  \
  xa[0][T] = ya[0][T]; xa[1][T] = ya[1][T]; xa[2][T] = ya[2][T]; xa[3][T] = ya[3][T];               \
  xb[0][T] = yb[0][T]; xb[1][T] = yb[1][T];
#define BT2_STEP_ODD(AP, BP)                                                                        \
  BT2_MFMA8(0, x) BT2_SB; BT2_LOAD(y, AP, BP) BT2_SB;                                               \
  BT2_MFMA8(1, x) BT2_SB; BT2_XYC(0) BT2_SB; BT2_MFMA8(2, x) BT2_SB; BT2_XYC(1) BT2_SB;             \
  BT2_MFMA8(3, x) BT2_SB; BT2_XYC(2) BT2_XYC(3) BT2_SB;
#define BT2_STEP_FIRST(AP, BP)                                                                      \
  BT2_MFMA8(0, x) BT2_SB; BT2_LOAD(y, AP, BP) BT2_SB;                                               \
  BT2_MFMA8(1, x) BT2_MFMA8(2, x) BT2_MFMA8(3, x) BT2_SB;
// second step of a pair inside a stage, and the stage's last step (early barrier, next stage's first
// fragments, DMA pieces of the stage after that): two bodies -- eight skipped DMA slots cost 16 scalar
// instructions per pair, i.e. ~1 cycle per MFMA
#define BT2_STEP_SECOND(AP, BP)                                                                     \
  BT2_MFMA8(0, y) BT2_SB; BT2_LOAD(x, AP, BP) BT2_SB;                                               \
  BT2_MFMA8(1, y) BT2_MFMA8(2, y) BT2_MFMA8(3, y) BT2_SB;
#define BT2_STEP_LAST()                                                                             \
  early_barrier();                                                                                  \
  BT2_SB; BT2_MFMA8(0, y) BT2_SB; BT2_LOAD(x, An, Bn) BT2_SB;                                       \
  BT2_MFMA2(1, y, 0) BT2_SB; dma_piece(0); BT2_SB; BT2_MFMA2(1, y, 1) BT2_SB; dma_piece(1); BT2_SB; \
  BT2_MFMA2(1, y, 2) BT2_SB; dma_piece(2); BT2_SB; BT2_MFMA2(1, y, 3) BT2_SB; dma_piece(3); BT2_SB; \
  BT2_MFMA2(2, y, 0) BT2_SB; dma_piece(4); BT2_SB; BT2_MFMA2(2, y, 1) BT2_SB; dma_piece(5); BT2_SB; \
  BT2_MFMA2(2, y, 2) BT2_SB; dma_piece(6); BT2_SB; BT2_MFMA2(2, y, 3) BT2_SB; dma_piece(7); BT2_SB; \
  BT2_MFMA2(3, y, 0) BT2_MFMA2(3, y, 1) BT2_SB; dma_advance();                                      \
  BT2_SB; BT2_MFMA2(3, y, 2) BT2_MFMA2(3, y, 3) BT2_SB;

template <int MODE>
__global__ __launch_bounds__(512, 2) void trials_gemm_bt2_kernel(
    const f32x4 *__restrict__ Apk, const f32x4 *__restrict__ Bpk, unsigned Mpad, unsigned Npad, int KQ,
    const float2 *__restrict__ rpair, const float2 *__restrict__ cpair, float *__restrict__ out, int64_t ld, int64_t M, int64_t Nt, int tilesM, int tilesN, int patchesN,
    int numPatches, unsigned long long *__restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
  constexpr bool TL = (MODE & 1) != 0;                            // timeline instrumentation (diagnostic)
  constexpr bool DIRECT = (MODE & 2) == 0;                        // epilogue stores straight from the accumulator layout (MODE bit 1: the LDS-transpose epilogue, A/B arm)
  // bounding arms (timing only, the scores are garbage): MODE bit 2 = no operand DMA (the MFMA stream, fragment reads,
  // barriers and the epilogue as they are, operands "resident"), bit 3 = no output stores (the epilogue's cost)
  constexpr bool NODMA = (MODE & 4) != 0, NOSTORE = (MODE & 8) != 0;
  // MODE bit 4: the product kernel + two stamp pairs of workgroup 0 (shader clock s_memtime, 100 MHz s_memrealtime) at
  // its start and end and its tile count: dbg[0..4] -> the clock the chip SUSTAINS under this kernel and the kernel's
  // cycles per tile (scripts/gemm_clock.py)
  constexpr bool CLK = (MODE & 16) != 0;
  unsigned long long clk_c0 = 0, clk_r0 = 0;
  if (CLK && threadIdx.x == 0) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;                        // 2 x 4 waves
  const int i = lane & 31, hh = lane >> 5;
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
  const int lbm = lb / BPC, lbn = lb % BPC;                       // this workgroup's tile inside a patch
  const int nsteps = KQ >> 1;                                     // 8-k steps per tile (>= 2: the host pads K to 16)
  const int nst = (nsteps + 3) >> 2;                              // stages per tile
  const int sbase = nsteps / nst, srem = nsteps - sbase * nst;    // stage j has sbase + (j < srem) steps
  const int lane16 = lane * 16;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4 *>(Apk), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4 *>(Bpk), 0, -1, 0x00020000);
  LDS_AS char *const lds = (LDS_AS char *)smem;
  (void)numPatches;

  // ---- tile walk (same order as the kernel above: XCD x takes patches x, x + 8, ...; its 32
  //      workgroups one tile each), stepped without divisions.  Only the DMA cursor walks; the
  //      compute cursor reads the tiles back from a 3-entry queue (the DMA runs <= 2 tiles ahead).
  int t_pm = 0, t_pn = xcd;
  auto tile_valid = [&]() { return t_pm * BPR + lbm < tilesM && t_pn * BPC + lbn < tilesN; };
  auto tile_norm = [&]() { while (t_pn >= patchesN) { t_pn -= patchesN; ++t_pm; } };
  auto tile_next = [&]() -> bool {      // false: no tile left (t_pm only grows)
    for (;;) {
      t_pn += 8;
      tile_norm();
      if (t_pm * BPR >= tilesM) return false;
      if (tile_valid()) return true;
    }
  };
  tile_norm();
  bool d_ok = t_pm * BPR < tilesM;
  if (d_ok && !tile_valid()) d_ok = tile_next();
  if (!d_ok) return;

  // ---- DMA cursor: the stage fetched next.  This wave moves rows [quarter*64, +64) of k-quads
  //      kqw, kqw+2, kqw+4, kqw+6 of a stage, A side then B side: 8 pieces of 1 KiB ----
  const unsigned quarter = wave & 3, kqw = wave >> 2;
  const unsigned strideA2 = Mpad * 32u, strideB2 = Npad * 32u;    // bytes per 2 k-quad planes
  const unsigned ldsw = kqw * 4096u + quarter * 1024u;            // this wave's first piece inside a stage buffer
  int d_st = 0, d_slot = 0;
  int d_r0 = (t_pm * BPR + lbm) * 256, d_c0 = (t_pn * BPC + lbn) * 256;
  int q_r0[3] = {d_r0, 0, 0}, q_c0[3] = {d_c0, 0, 0};
  int q_ok[3] = {1, 0, 0};   // (ints, not bools: the tile hand-over then stays on the scalar unit)
  unsigned d_offA = ((unsigned)kqw * Mpad + (unsigned)d_r0 + quarter * 64u) * 16u;
  unsigned d_offB = ((unsigned)kqw * Npad + (unsigned)d_c0 + quarter * 64u) * 16u;
  unsigned d_lds = 0;                                             // byte offset of the buffer the DMA cursor fills
  float *const bias_lds = reinterpret_cast<float *>(reinterpret_cast<char *>(smem) + BT2_BIAS);

  auto dma_piece = [&](int jj) {
    if (NODMA) return;
    if (jj < 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void *)(lds + d_lds + ldsw + jj * 8192u), 16, lane16,
                                               (int)(d_offA + (unsigned)jj * strideA2), 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void *)(lds + d_lds + 32768u + ldsw + (jj - 4) * 8192u), 16,
                                               lane16, (int)(d_offB + (unsigned)(jj - 4) * strideB2), 0, 0);
  };
  // after the 8 pieces of the cursor's stage have been issued: step the cursor to the following
  // stage.  The common case is a handful of scalar operations (it sits between the MFMAs of the
  // stage's last step); entering a tile -- its biases (row biases, column biases, row scales ->
  // slot d_slot) ride with its first stage -- and leaving one are the rare branches.  Past the last
  // tile the pieces re-fetch the last position; nothing consumes them.
  auto dma_advance = [&]() {
    if (d_st == 0 && d_ok && !NODMA) {
      // the tile's bias pairs, 2 KiB per side, one piece each from waves 0..3.  Buffer (MUBUF) DMA, not
      // global_load_lds: a pending FLAT-encoded LDS load makes the compiler's waitcnt pass turn every
      // LDS wait of the epilogue into lgkmcnt(0)
      float *dst = bias_lds + d_slot * 1024 + wave * 256;
      if (wave < 2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(rpair), 0, -1, 0x00020000),
                                                 (LDS_AS void *)dst, 16, lane16, d_r0 * 8 + wave * 1024, 0, 0);
      else if (wave < 4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(cpair), 0, -1, 0x00020000),
                                                 (LDS_AS void *)dst, 16, lane16, d_c0 * 8 + (wave - 2) * 1024, 0, 0);
    }
    const unsigned npd = (unsigned)(sbase + (d_st < srem ? 1 : 0));
    d_offA += npd * strideA2;
    d_offB += npd * strideB2;
    d_lds ^= 65536u;
    if (++d_st == nst) {
      d_st = 0;
      d_slot = d_slot == 2 ? 0 : d_slot + 1;
      if (d_ok) d_ok = tile_next();
      if (d_ok) { d_r0 = (t_pm * BPR + lbm) * 256; d_c0 = (t_pn * BPC + lbn) * 256; }
      if (d_slot == 0) { q_r0[0] = d_r0; q_c0[0] = d_c0; q_ok[0] = d_ok ? 1 : 0; }
      else if (d_slot == 1) { q_r0[1] = d_r0; q_c0[1] = d_c0; q_ok[1] = d_ok ? 1 : 0; }
      else { q_r0[2] = d_r0; q_c0[2] = d_c0; q_ok[2] = d_ok ? 1 : 0; }
      d_offA = ((unsigned)kqw * Mpad + (unsigned)d_r0 + quarter * 64u) * 16u;
      d_offB = ((unsigned)kqw * Npad + (unsigned)d_c0 + quarter * 64u) * 16u;
    }
  };

  unsigned long long t_arr = 0, t_lv = 0;
  int tseq = 0, st = 0;
  // (the asm memory clobbers around every raw s_barrier of this file: llvm.amdgcn.s.barrier is IntrNoMem, so the optimiser may move
  //  plain LDS loads across it -- it did in syrk_blk.inc, a one-in-a-thousand wrong block, round 6.  Here the compiler had kept the
  //  last step's fragment reads in front of the barrier; with the clobbers it has to, and the code is the same but for three
  //  scalar instructions.)
  auto early_barrier = [&]() {
    if (TL) t_arr = __builtin_amdgcn_s_memtime();
    // (the builtin, not inline asm: the compiler's waitcnt pass must see this drain, or it keeps the
    //  bias DMA -- a FLAT-encoded global_load_lds -- "pending" forever and turns every later LDS wait
    //  of the epilogue into lgkmcnt(0))
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
    if (TL) {
      t_lv = __builtin_amdgcn_s_memtime();
      if (blockIdx.x == 0 && lane == 0 && tseq < 8 && st < 16) {
        unsigned long long *p = dbg + (((size_t)tseq * 16 + st) * 8 + wave) * 8;
        p[0] = t_arr; p[1] = t_lv;
      }
    }
  };

  auto step_stamp = [&](int sn) {
    if (TL) {
      const unsigned long long ts = __builtin_amdgcn_s_memtime();
      if (blockIdx.x == 0 && lane == 0 && tseq < 8 && st < 16) dbg[(((size_t)tseq * 16 + st) * 8 + wave) * 8 + 1 + sn] = ts;
    }
  };
  // ---- prologue: stage 0 of the first tile, barrier, then stage 1 in a burst ----
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) dma_piece(jj);
  dma_advance();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0070);
  __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) dma_piece(jj);
  dma_advance();

  int cur = 0, c_slot = 0;                                         // buffer / queue slot of the compute cursor
  int r0 = q_r0[0], c0 = q_c0[0];
  bool have = true;
  const f32x4 *const Abase = smem + hh * 256 + wm * 128 + i;
  const f32x4 *const Bbase = smem + 8 * 256 + hh * 256 + wn * 64 + i;
  f32x4 xa[4], xb[2], ya[4], yb[2];
  BT2_LOAD(x, Abase, Bbase)

  while (have) {
    // accumulators start from the bias r'_i + s_i q_j, formed by ONE rank-2 MFMA per accumulator on the
    // tile's bias pairs (srcC = 0): no vector-ALU work at the tile boundary.  (A wave's VALU
    // instructions get one issue slot per MFMA of its partner while that one is MFMA-dense: the 128
    // register writes of a conventional setup took the trailing wave ~6500 cycles per tile.)
    const float *const bl = bias_lds + c_slot * 1024;
    f32x16 acc[4][2];
    {
      float ap[4], bp[2];
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) ap[tm] = bl[(wm * 128 + tm * 32 + i) * 2 + hh];
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) bp[tn] = bl[512 + (wn * 64 + tn * 32 + i) * 2 + hh];
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[tm], bp[tn], zero16, 0, 0, 0);
    }

    for (st = 0; st < nst; ++st) {
      const int np = sbase + (st < srem ? 1 : 0);
      const f32x4 *Ac = Abase + cur * 4096, *Bc = Bbase + cur * 4096;
      const f32x4 *An = Abase + (cur ^ 1) * 4096, *Bn = Bbase + (cur ^ 1) * 4096;
      int sn = 0;                // step whose fragments are in x
      if (np & 1) {
        step_stamp(1);
        BT2_STEP_ODD(Ac + 512, Bc + 512)
        sn = 1;
      }
#pragma unroll 1
      for (; sn + 2 < np; sn += 2) {
        step_stamp(sn + 1);
        BT2_STEP_FIRST(Ac + (sn + 1) * 512, Bc + (sn + 1) * 512)
        BT2_STEP_SECOND(Ac + (sn + 2) * 512, Bc + (sn + 2) * 512)
      }
      step_stamp(sn + 1);
      BT2_STEP_FIRST(Ac + (sn + 1) * 512, Bc + (sn + 1) * 512)
      BT2_STEP_LAST()
      cur ^= 1;
    }

    // ---- epilogue: 16 round trips of 8 rows x 64 columns through this wave's 2 KiB of staging ----
    unsigned long long t_e0 = 0;
    if (TL) t_e0 = __builtin_amdgcn_s_memtime();
    {
      float *stg = reinterpret_cast<float *>(reinterpret_cast<char *>(smem) + BT2_STG) + wave * 512;
      const int rrow = lane >> 4, rcol = (lane & 15) * 4;
      const bool interior = ((int64_t)r0 + 256 <= M) && ((int64_t)c0 + 256 <= Nt) && ((ld & 3) == 0) &&
                            ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
      // chunk c = 8 rows x 64 columns: W(c) 8 values per lane into the staging tile, R(c) two 16-byte
      // row reads back, S(c) two 16-byte non-temporal stores (4 rows x 256 B per instruction)
      auto chunk_w = [&](int c) {
        const int tm = c >> 2, q = c & 3;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int e = 0; e < 4; ++e) stg[(4 * hh + e) * 64 + tn * 32 + i] = acc[tm][tn][4 * q + e];
      };
      if (NOSTORE && r0 >= 0) {
        // (bounding arm: r0 is never negative, but the compiler cannot know; the accumulators stay live)
      } else if (DIRECT && interior) {
        // no LDS round trip: a register of the 32 x 32 accumulator holds rows (8 q + e) and (8 q + e + 4) (lane >> 5)
        // of 32 consecutive columns, i.e. one store instruction = two full 128-byte row pieces.  The row walks on a
        // scalar base; the lane offset (half-row, column) is constant.  128 dword stores per tile and wave against
        // 128 ds_write + 32 ds_read + 32 16-byte stores of the transposing epilogue below: what an epilogue costs
        // beside an MFMA-dense partner wave is the NUMBER of vector-memory instructions it issues (~31-47 cycles
        // each), not their bytes -- 87.7 -> 88.8 % at D = 200, 94.2 -> 94.7 % at D = 512 in one interleaved sweep.
        const char *tb = reinterpret_cast<const char *>(out + (((int64_t)r0 + wm * 128) * ld + c0 + wn * 64));
        const int64_t ldb = 4 * ld;                             // bytes per row
        unsigned voff = (unsigned)hh * (4u * (unsigned)ld * 4u) + (unsigned)i * 4u;
        asm volatile("" : "+v"(voff));
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              __builtin_nontemporal_store(acc[tm][0][4 * q + e], reinterpret_cast<float *>(const_cast<char *>(tb) + voff));
              __builtin_nontemporal_store(acc[tm][1][4 * q + e], reinterpret_cast<float *>(const_cast<char *>(tb) + 128 + voff));
              tb += ldb;
            }
            tb += 4 * ldb;
          }
      } else if (interior) {
        // stores address the tile through a scalar base that walks down the wave's 128 rows, 4 rows per
        // store, plus a constant 32-bit lane offset (the host guarantees ld < 2^22).
        // Software pipeline W(c) R(c) S(c-1): a wave's LDS operations execute in order, so W(c) may be
        // issued behind R(c-1) on the same 2 KiB, and the stores of chunk c-1 wait only for R(c-1).
        // (the walk is done on the SCALAR base -- two SALU adds per store -- and the lane offset stays
        //  constant: vector ALU work in the epilogue waits behind the partner wave's MFMAs)
        const char *tbase = reinterpret_cast<const char *>(out + (((int64_t)r0 + wm * 128) * ld + c0 + wn * 64));
        const int64_t ldb4 = 16 * ld;                           // bytes per 4 rows
        unsigned voff = (unsigned)rrow * ((unsigned)ld * 4u) + (unsigned)rcol * 4u;
        asm volatile("" : "+v"(voff));   // opaque: one live register, not 32 hoisted addresses
        f32x4 pv[2][2];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          chunk_w(c);
          __builtin_amdgcn_sched_barrier(0);
          pv[c & 1][0] = *reinterpret_cast<const f32x4 *>(stg + lane * 4);
          pv[c & 1][1] = *reinterpret_cast<const f32x4 *>(stg + 256 + lane * 4);
          __builtin_amdgcn_sched_barrier(0);
          if (c > 0) {
            __builtin_nontemporal_store(pv[(c - 1) & 1][0], reinterpret_cast<f32x4 *>(const_cast<char *>(tbase) + voff));
            __builtin_nontemporal_store(pv[(c - 1) & 1][1], reinterpret_cast<f32x4 *>(const_cast<char *>(tbase + ldb4) + voff));
            tbase += 2 * ldb4;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_nontemporal_store(pv[1][0], reinterpret_cast<f32x4 *>(const_cast<char *>(tbase) + voff));
        __builtin_nontemporal_store(pv[1][1], reinterpret_cast<f32x4 *>(const_cast<char *>(tbase + ldb4) + voff));
      } else {
        const int64_t wrow0 = (int64_t)r0 + wm * 128, wcol0 = (int64_t)c0 + wn * 64;
        float *dst = out + (wrow0 + rrow) * ld + wcol0 + rcol;
        int64_t row = wrow0 + rrow;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          chunk_w(c);
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(stg + k * 256 + lane * 4);
            if (row < M) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (wcol0 + rcol + e < Nt) __builtin_nontemporal_store(v[e], dst + e);
            }
            dst += 4 * ld;
            row += 4;
          }
        }
      }
    }
    if (TL) {
      const unsigned long long t_e1 = __builtin_amdgcn_s_memtime();
      if (blockIdx.x == 0 && lane == 0 && tseq < 8) {
        unsigned long long *p = dbg + (((size_t)tseq * 16 + 15) * 8 + wave) * 8;
        p[6] = t_e0; p[7] = t_e1;
      }
    }
    ++tseq;
    c_slot = c_slot == 2 ? 0 : c_slot + 1;
    r0 = c_slot == 0 ? q_r0[0] : (c_slot == 1 ? q_r0[1] : q_r0[2]);
    c0 = c_slot == 0 ? q_c0[0] : (c_slot == 1 ? q_c0[1] : q_c0[2]);
    have = (c_slot == 0 ? q_ok[0] : (c_slot == 1 ? q_ok[1] : q_ok[2])) != 0;
  }
  __builtin_amdgcn_s_waitcnt(0x0070);   // no DMA may land in LDS after the workgroup has gone
  if (CLK && threadIdx.x == 0) {     // every workgroup: dbg[8 + 4 b ..] = start / end in shader cycles and in 100 MHz ticks
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long *p = dbg + 8 + 4 * (size_t)blockIdx.x;
    p[0] = c1 - clk_c0; p[1] = clk_r0; p[2] = r1; p[3] = (unsigned long long)tseq;
    if (blockIdx.x == 0) { dbg[0] = c1 - clk_c0; dbg[1] = r1 - clk_r0; dbg[2] = (unsigned long long)tseq; }
  }
}
#undef BT2_STEP_LAST
#undef BT2_STEP_SECOND
#undef BT2_STEP_FIRST
#undef BT2_STEP_ODD
#undef BT2_XYC
#undef BT2_LOAD
#undef BT2_MFMA8
#undef BT2_MFMA2
#undef BT2_SB

// (score_bt4.inc defines the tile fetch's two values on the fetching path only, on purpose -- see the comment at f_raw; the
//  warning is silenced for that kernel alone, not for the translation unit)
// Patch k of XCD queue x of a patch grid pM x pN (both 256 x 256 kernels' queues and the bf16x3 arm's static walk).  Row walk (rounds 4-5a): patches x, x + 8, ... of the row-major patch grid -- consecutive patches of an
// XCD share their A panels, every B panel is fetched once per patch row.  Column walk: queue x owns the patch columns x, x + 8,
// ... and goes down each -- consecutive patches share their B panels, every A panel is fetched once per patch column.
// The columns beyond the last full round of eight (pN % 8 of them) are dealt patch by patch in row-major order, so no queue is a
// whole column longer than another.
template <typename I = int64_t>
__device__ __host__ inline I bt4_queue_patches(int x, int pM, int pN, int colwalk) {
  if (colwalk) {
    const I rest = (I)pM * (pN % 8);
    return (I)pM * (pN / 8) + (x < rest ? (rest - x + 7) / 8 : 0);
  }
  const I np = (I)pM * pN;
  return x < np ? (np - x + 7) / 8 : 0;
}
template <typename I>
__device__ __host__ inline void bt4_patch(int x, I k, int pM, int pN, int colwalk, int &pm, int &pn) {
  if (colwalk) {
    const I whole = (I)pM * (pN / 8);
    if (k < whole) { pn = x + 8 * (int)(k / pM); pm = (int)(k % pM); }
    else { const int r = pN % 8; const I q = x + 8 * (k - whole); pm = (int)(q / r); pn = (pN / 8) * 8 + (int)(q % r); }
  } else { const I p = x + 8 * k; pm = (int)(p / pN); pn = (int)(p % pN); }
}
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wuninitialized"
#pragma clang diagnostic ignored "-Wsometimes-uninitialized"
#pragma clang diagnostic ignored "-Wconditional-uninitialized"
#include "score_bt4.inc"
#pragma clang diagnostic pop
#include "score_bf16x3.inc"

// Tile schedule of trials_gemm_bt4_kernel for a btM x btN grid of 256 x 256 tiles: queue x (one per XCD) lists the tiles of
// patches x, x + 8, ... (BPR x BPC tiles each, row-major inside a patch) -- the order the static walk of bt2 takes them in,
// so that the 32 workgroups of an XCD still work inside one patch at a time and share its operand panels in their L2 --
// but a workgroup takes "the next tile of the queue" through an atomic counter instead of a fixed position, and a
// workgroup whose queue has run dry continues in the next XCD's.  Why: the XCDs of one chip do not run at one clock under
// this load (measured per XCD with s_memtime / s_memrealtime stamps, scripts/gemm_clock.py: 2.30 against 2.35 GHz, odd
// against even XCDs), and awkward grids leave up to 5 % more tiles with some XCDs than with others; with equal static
// shares the kernel ends 1.4 - 2.4 ms after its average workgroup (of 28 ms at C2).
// Round 5: the table is built ON THE DEVICE (one wave per queue, in the order above) from per-queue offsets the host gets
// by walking the patches -- no host copy and no stream synchronisation, so a change of the tile grid does not stall the
// host (the sharded form alternates full blocks and a ragged tail; a server scores varying M) -- and the last few grids
// keep their tables (h->bt4_tabs, least recently used replaced).  The queue counters start every launch at zero: the
// launch before left them so (score_bt4.inc: bt4_leave).
__global__ __launch_bounds__(64) void bt4_table_kernel(int2 *__restrict__ tab, int btM, int btN, int pM, int pN, int colwalk, const Bt4Queues qs) {
  const int x = blockIdx.x, lane = threadIdx.x;
  int off = qs.qbase[x];
  const int64_t nq = bt4_queue_patches(x, pM, pN, colwalk);
  for (int64_t k = 0; k < nq; ++k) {
    int pm, pn;
    bt4_patch(x, k, pM, pN, colwalk, pm, pn);
    const int tm = pm * BPR + lane / BPC, tn = pn * BPC + lane % BPC;
    const bool ok = lane < BPR * BPC && tm < btM && tn < btN;
    const unsigned long long mask = __ballot(ok);
    if (ok) tab[off + __popcll(mask & ((1ull << lane) - 1ull))] = make_int2(tm * 256, tn * 256);
    off += __popcll(mask);
  }
}

static int bt4_schedule(plda_handle *h, int btM, int btN, int KQ, Bt4Table **out) {
  if (!h->bt4_cnt.p) {          // the queue counters: zero once, every launch leaves them at zero (score_bt4.inc: bt4_leave)
    PLDA_HIP(h, h->bt4_cnt.reserve(32 * sizeof(unsigned)));
    PLDA_HIP(h, hipMemsetAsync(h->bt4_cnt.p, 0, 32 * sizeof(unsigned), h->stream));
  }
  // Which operand's panels stay in an XCD's L2 from patch to patch, i.e. which operand is fetched again and again (round 5).
  // Along a patch row (rounds 2-5a: always) every B panel is fetched once per patch row; down a patch column every A panel once
  // per patch column.  What decides is where the repeated operand comes from: at C4 the 1.27 GB test side -- five times the
  // 256 MB Infinity Cache -- came from HBM 40 times; down the columns the 42 MB enrol side repeats and the cache holds it.
  // Measured (scripts/probe/colwalk_ab.sh, colwalk_mid.sh; ms per step, row walk -> column walk, same box): C4 183.1 -> 174.2,
  // C3 72.2 -> 71.5 (FETCH_SIZE per launch 56.8 -> 36.0 GB and 35.1 -> 30.5 GB); where both operands fit the cache or neither
  // does the column walk LOSES a little: C2 28.23 -> 28.38 (11.0 -> 4.4 GB), 150k x 150k x 512 156.1 -> 157.4, 200k x 200k x 200
  // and 140k x 140k x 256 equal.  So: down the columns exactly when the enrol side fits half the cache and the test side does
  // not.  (Starting each XCD at a different height of its column, and dealing the columns beyond the last round of eight patch
  // by patch, changed nothing measurable; the second is kept for the balance of the queues.)
  // PLDA_GEMM_VARIANT=48: the row walk always; 49: the column walk always (A/B arms).
  const size_t keep = (size_t)128 << 20, bytesA = (size_t)btM * 256 * KQ * 16, bytesB = (size_t)btN * 256 * KQ * 16;
  const int colwalk = h->gemm_variant == 48 ? 0 : h->gemm_variant == 49 ? 1 : (bytesA <= keep && bytesB > keep) ? 1 : 0;
  Bt4Table *lru = &h->bt4_tabs[0];
  for (auto &t : h->bt4_tabs) {
    if (t.btM == btM && t.btN == btN && t.colwalk == colwalk) { t.used = ++h->bt4_clock; *out = &t; return PLDA_OK; }
    if (t.used < lru->used) lru = &t;
  }
  Bt4Table &t = *lru;
  const int pM = (int)ceil_div(btM, BPR), pN = (int)ceil_div(btN, BPC);
  int64_t total = 0;
  Bt4Queues qs;
  for (int x = 0; x < 8; ++x) {
    t.qbase[x] = (int)total;
    const int64_t nq = bt4_queue_patches(x, pM, pN, colwalk);
    for (int64_t k = 0; k < nq; ++k) {
      int pm, pn;
      bt4_patch(x, k, pM, pN, colwalk, pm, pn);
      total += (int64_t)std::min(BPR, btM - pm * BPR) * std::min(BPC, btN - pn * BPC);
    }
    t.qlen[x] = (int)total - t.qbase[x];
    qs.qbase[x] = t.qbase[x]; qs.qlen[x] = t.qlen[x];
  }
  t.btM = t.btN = -1;
  PLDA_HIP(h, t.tab.reserve(std::max<size_t>((size_t)total, 1) * sizeof(int2)));
  bt4_table_kernel<<<8, 64, 0, h->stream>>>(t.tab.as<int2>(), btM, btN, pM, pN, colwalk, qs);
  PLDA_LAUNCH_CHECK(h);
  t.btM = btM; t.btN = btN; t.colwalk = colwalk; t.used = ++h->bt4_clock;
  *out = &t;
  return PLDA_OK;
}

// finalise fused z-norm statistics: mean = shift + S1/N, std = sqrt(S2/N - (S1/N)^2)
__global__ void znorm_finalize_kernel(const float *__restrict__ shift, const double *__restrict__ colsum,
                                      const double *__restrict__ colsq, int64_t M, double invN,
                                      double *__restrict__ out_mean, double *__restrict__ out_std) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  const double m1 = colsum[j] * invN;
  double var = colsq[j] * invN - m1 * m1;
  if (var < 0.0) var = 0.0;
  out_mean[j] = (double)shift[j] + m1;
  out_std[j] = sqrt(var);
}

__global__ void pilot_shift_kernel(const double *__restrict__ colsum, int64_t M, double invN,
                                   float *__restrict__ shift) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < M) shift[j] = (float)(colsum[j] * invN);
}

// ------------------------------------------------------------------------------------
// trial list in fp64: one wave per (enrol, test) pair -- Plda::LogLikelihoodRatio
// verbatim (pldamodule.cpp:266) plus the z-norm of :269-273.
// ------------------------------------------------------------------------------------
__global__ void score_pairs_kernel(const double *__restrict__ U, const int32_t *__restrict__ n_enrol,
                                   const double *__restrict__ V, const int64_t *__restrict__ e_idx,
                                   const int64_t *__restrict__ t_idx, int64_t P,
                                   const double *__restrict__ psi, int D,
                                   const double *__restrict__ zmean, const double *__restrict__ zstd,
                                   double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= P) return;
  const int64_t e = e_idx[p], t = t_idx[p];
  const double n = (double)n_enrol[e];
  const double *u = U + e * (int64_t)D, *v = V + t * (int64_t)D;
  double acc = 0.0;
  for (int d = lane; d < D; d += 64) {
    double c, var;
    const double ps = psi[d];
    llr_coef(n, ps, c, var);
    const double diff = v[d] - c * u[d];
    acc += (log(var) + diff * diff / var) - (log(1.0 + ps) + v[d] * v[d] / (1.0 + ps));
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) {
    double s = -0.5 * acc;
    if (zmean && zstd && zstd[e] != 0.0) s = (s - zmean[e]) / zstd[e];
    out[p] = s;
  }
}

// Long trial lists (round 5).  The kernel above evaluates the reference's expression per element -- per trial and dimension four
// fp64 divisions and two logarithms that depend on (count, dimension) only: 19.8 ms for 10^7 trials at D = 200.  With the
// enrol counts bucketed by their distinct values (the CountSet of the trials matrix) those terms become tables, one per
// bucket: c_d, 1 / var_d, L = sum_d [log var_d - log(1 + psi_d)], and 1 / (1 + psi_d) shared by all -- per element two
// loads of the vectors and three fused multiply-adds:
//     LLR = -1/2 [ L + sum_d ( (v_d - c_d u_d)^2 / var_d - v_d^2 / (1 + psi_d) ) ]
// the same sum in another association (1e-13 against the oracle where the verbatim kernel has 2e-13).  Lists shorter than
// PAIRS_TAB_MIN, or counts the set does not take (> 4095, more than 64 distinct ones), stay on the verbatim kernel.
constexpr int64_t PAIRS_TAB_MIN = 16384;
// the first trial whose indices leave the enrol / test sets (-1: none), found on the device: one stream synchronisation
__global__ void pairs_validate_kernel(const int64_t *__restrict__ e, const int64_t *__restrict__ t, int64_t P, int64_t M, int64_t Nt,
                                      unsigned long long *__restrict__ first_bad) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x)
    if (e[p] < 0 || e[p] >= M || t[p] < 0 || t[p] >= Nt) atomicMin(first_bad, (unsigned long long)p);
}
int pairs_validate_device(plda_handle *h, const int64_t *de, const int64_t *dt, int64_t P, int64_t M, int64_t Nt, long long *bad) {
  PLDA_HIP(h, h->w[12].reserve(8));
  PLDA_HIP(h, hipMemsetAsync(h->w[12].p, 0xff, 8, h->stream));
  pairs_validate_kernel<<<(unsigned)std::min<int64_t>(ceil_div(P, 256), 4096), 256, 0, h->stream>>>(de, dt, P, M, Nt, h->w[12].as<unsigned long long>());
  PLDA_LAUNCH_CHECK(h);
  unsigned long long v = 0;
  PLDA_HIP(h, hipMemcpyAsync(&v, h->w[12].p, 8, hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  *bad = v == ~0ull ? -1 : (long long)v;
  return PLDA_OK;
}
__global__ __launch_bounds__(256) void pairs_tables_kernel(const double *__restrict__ psi, int D, const CountSet cs, double *__restrict__ tab /*[G][2 D + 1] + [D]*/) {
  __shared__ double red[256];
  const int g = blockIdx.x, S = 2 * D + 1;
  if (g == cs.G) {                               // the shared 1 / (1 + psi)
    for (int d = threadIdx.x; d < D; d += 256) tab[(size_t)cs.G * S + d] = 1.0 / (1.0 + psi[d]);
    return;
  }
  double acc = 0.0;
  for (int d = threadIdx.x; d < D; d += 256) {
    double c, var;
    const double p = psi[d];
    llr_coef((double)cs.vals[g], p, c, var);
    tab[(size_t)g * S + d] = c;
    tab[(size_t)g * S + D + d] = 1.0 / var;
    acc += log(var) - log(1.0 + p);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) tab[(size_t)g * S + 2 * D] = red[0];
}
__global__ void pairs_bucket_kernel(const int32_t *__restrict__ n, int64_t M, const CountSet cs, int32_t *__restrict__ bidx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const int v = n[i];
  int b = 0;
  for (int g = 1; g < cs.G; ++g) b = cs.vals[g] == v ? g : b;
  bidx[i] = (b == 0 && cs.vals[0] != v) ? -1 : b;        // -1: not in the set -> NaN scores for this model (score_pairs_tab_kernel)
}
// one wave per PPW consecutive trials (their loads in flight together)
template <int PPW>
__global__ __launch_bounds__(256) void score_pairs_tab_kernel(const double *__restrict__ U, const int32_t *__restrict__ bidx, const double *__restrict__ V,
                                                              const int64_t *__restrict__ e_idx, const int64_t *__restrict__ t_idx, int64_t P,
                                                              const double *__restrict__ tab, int G, int D, const double *__restrict__ zmean,
                                                              const double *__restrict__ zstd, double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t p0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * PPW;
  if (p0 >= P) return;
  const int S = 2 * D + 1;
  const double *const i1 = tab + (size_t)G * S;
  int64_t e[PPW];
  const double *u[PPW], *v[PPW], *tb[PPW];
  double acc[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    const int64_t p = min(p0 + k, P - 1);
    e[k] = e_idx[p];
    u[k] = U + e[k] * (int64_t)D;
    v[k] = V + t_idx[p] * (int64_t)D;
    const int bk = bidx[e[k]];
    tb[k] = tab + (size_t)max(bk, 0) * S;
    acc[k] = bk < 0 ? __longlong_as_double(0x7ff8000000000000ll) : 0.0;
  }
  for (int d = lane; d < D; d += 64) {
    const double w1 = i1[d];
    double uu[PPW], vv[PPW], cc[PPW], iv[PPW];
#pragma unroll
    for (int k = 0; k < PPW; ++k) { uu[k] = u[k][d]; vv[k] = v[k][d]; cc[k] = tb[k][d]; iv[k] = tb[k][D + d]; }
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const double diff = fma(-cc[k], uu[k], vv[k]);
      acc[k] = fma(diff * diff, iv[k], acc[k]);
      acc[k] = fma(-vv[k] * vv[k], w1, acc[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    const double a = wave_sum_f64(acc[k]);
    if (lane == 0 && p0 + k < P) {
      double s = -0.5 * (a + tb[k][2 * D]);
      if (zmean && zstd && zstd[e[k]] != 0.0) s = (s - zmean[e[k]]) / zstd[e[k]];
      out[p0 + k] = s;
    }
  }
}

// ------------------------------------------------------------------------------------
// host orchestration of a trials-matrix call
// ------------------------------------------------------------------------------------

struct TrialOperands {
  int64_t Mpad, Npad;
  int KQ, Kg;   // padded GEMM depth (multiple of 8) and its k-quad count
  int Kg_alg;   // algorithmic depth: Dout (uniform n), Dout + G - 1 (mixed n, bucketed) or 2 Dout (mixed n, depth-2D form)
  int kind;     // 0 uniform count, 1 mixed counts in the depth-2D form, 2 mixed counts bucketed by distinct count
  bool mixed;   // kind == 1
};

// does the bucketed form apply (and pay) for this set of distinct counts?  cs.G == 0: the set is unknown / unusable
static inline bool buckets_usable(const plda_handle *h, const CountSet *cs) {
  if (!cs || h->mixed_variant == 1) return false;
  const int Dp = (int)round_up(h->Dout, 8);
  return cs->G >= 2 && cs->G <= CS_MAX && cs->G - 1 <= std::max(Dp / 2, 8);
}
// k-quad planes of a packed operand (without the 8 spare ones)
static inline int64_t operand_kq(const plda_handle *h, bool has_counts, const CountSet *cs) {
  const int64_t Dp = round_up(h->Dout, 8);
  if (has_counts && buckets_usable(h, cs)) return (Dp + round_up(cs->G - 1, 8)) / 4;
  return std::max<int64_t>((has_counts ? 2 : 1) * Dp, 16) / 4;
}

// doA / doB: (re)build the enrol side (packed A, row biases, row scales) / the test side (packed B,
// column biases); a blocked call packs each side once per block of its own dimension.
// cs: the distinct enrol counts of the whole call when dn != nullptr (nullptr / unusable: the depth-2D form)
static int prepare_operands(plda_handle *h, const double *dU, const int32_t *dn, int n_uniform,
                            int64_t M, const double *dV, int64_t Nt, const double *dzmean,
                            const double *dzstd, TrialOperands &op, bool doA = true, bool doB = true,
                            const CountSet *cs = nullptr) {
  TraceScope ts(h, "score.pack_operands");
  const int D = h->Dout;
  const int Dp = (int)round_up(D, 8);
  if (doB) h->prep_valid = false;        // the packed test side is being overwritten (plda_score_prepare_dev re-marks its own)
  op.kind = !dn ? 0 : (buckets_usable(h, cs) ? 2 : 1);
  op.mixed = op.kind == 1;
  op.Kg = (int)operand_kq(h, dn != nullptr, cs) * 4;   // >= two 8-k steps: the 256 x 256 kernel runs them in pairs
  op.Kg_alg = op.kind == 2 ? D + cs->G - 1 : (op.mixed ? 2 * D : D);
  op.KQ = op.Kg / 4;
  op.Mpad = round_up(M, 256);   // 256: the big-tile kernel's block tile (the 128 kernel tolerates it)
  op.Npad = round_up(Nt, 256);
  // 8 spare k-quad planes: the bt2 kernel's branch-free DMA may fetch (never consume) up to one stage past KQ
  PLDA_HIP(h, h->s_Apk.reserve((size_t)(op.KQ + 8) * op.Mpad * 16));
  PLDA_HIP(h, h->s_Bpk.reserve((size_t)(op.KQ + 8) * op.Npad * 16));
  PLDA_HIP(h, h->s_rbias.reserve((size_t)op.Mpad * 4));
  PLDA_HIP(h, h->s_rscale.reserve((size_t)op.Mpad * 4));
  PLDA_HIP(h, h->s_cbias.reserve((size_t)op.Npad * 4));
  PLDA_HIP(h, h->s_rpair.reserve((size_t)op.Mpad * 8));
  PLDA_HIP(h, h->s_cpair.reserve((size_t)op.Npad * 8));
  if (op.kind == 2) {
    // bucketed mixed counts: per-bucket coefficient tables, then one pass over each side (+ the test side's dq planes)
    const int G = cs->G, KQm = Dp / 4, KQx = op.KQ - KQm;
    PLDA_HIP(h, h->w[11].reserve((size_t)G * (2 * D + 1) * 8));
    double *coefG = h->w[11].as<double>();
    h->ucoef_ptr = nullptr;          // (the uniform path's cached coefficients live in the same buffer)
    // the tables depend on (model, dimension, the set of counts) only: kept across calls like the uniform path's
    if (!(h->gcoef_ptr == coefG && h->gcoef_epoch == h->model_epoch && h->gcoef_D == D && h->gcoef_set.G == G &&
          std::memcmp(h->gcoef_set.vals, cs->vals, (size_t)G * sizeof(int32_t)) == 0)) {
      bucket_coef_kernel<<<G, 256, 0, h->stream>>>(h->d_psi.as<double>(), D, *cs, coefG);
      PLDA_LAUNCH_CHECK(h);
      h->gcoef_ptr = coefG; h->gcoef_epoch = h->model_epoch; h->gcoef_D = D; h->gcoef_set = *cs;
    }
    auto few = [&](int64_t rpad) { return h->prep_variant == 3 || (h->prep_variant != 2 && rpad <= 32768); };   // (as the uniform path below)
    if (doA && doB && few(op.Mpad) && few(op.Npad)) {
      const PrepBucketArgs a{dU, dn, M, op.Mpad, h->s_Apk.as<float>(), h->s_rbias.as<float>(), h->s_rscale.as<float>(), h->s_rpair.as<float2>(),
                             dV, Nt, op.Npad, h->s_Bpk.as<float>(), h->s_cbias.as<float>(), h->s_cpair.as<float2>()};
      prep_buckets_both_kernel<<<(unsigned)((op.Mpad + op.Npad) / 16), 256, 0, h->stream>>>(a, (int)(op.Mpad / 16), *cs, coefG, h->d_psi.as<double>(), D,
                                                                                           KQm, KQx, dzmean, dzstd);
      PLDA_LAUNCH_CHECK(h);
      return PLDA_OK;
    }
    if (doA) {
      if (few(op.Mpad))
        prep_enrol_buckets_kernel<4><<<(unsigned)(op.Mpad / 16), 256, 0, h->stream>>>(
            dU, dn, *cs, coefG, h->d_psi.as<double>(), D, M, op.Mpad, KQm, KQx, dzmean, dzstd, h->s_Apk.as<float>(),
            h->s_rbias.as<float>(), h->s_rscale.as<float>(), h->s_rpair.as<float2>());
      else
        prep_enrol_buckets_kernel<16><<<(unsigned)(op.Mpad / 64), 256, 0, h->stream>>>(
            dU, dn, *cs, coefG, h->d_psi.as<double>(), D, M, op.Mpad, KQm, KQx, dzmean, dzstd, h->s_Apk.as<float>(),
            h->s_rbias.as<float>(), h->s_rscale.as<float>(), h->s_rpair.as<float2>());
    }
    if (doB) {
      if (few(op.Npad)) {
        prep_side_kernel<1, 4><<<(unsigned)(op.Npad / 16), 256, 0, h->stream>>>(
            dV, coefG + D, coefG + 2 * D, 0, h->d_psi.as<double>(), D, Nt, op.Npad, KQm, nullptr, nullptr, h->s_Bpk.as<float>(),
            h->s_cbias.as<float>(), nullptr, h->s_cpair.as<float2>());
        prep_test_buckets_kernel<4><<<(unsigned)(op.Npad / 16), 256, 0, h->stream>>>(dV, coefG, G, D, Nt, op.Npad, KQm, KQx, h->s_Bpk.as<float>());
      } else {
        prep_side_kernel<1, 16><<<(unsigned)(op.Npad / 64), 256, 0, h->stream>>>(
            dV, coefG + D, coefG + 2 * D, 0, h->d_psi.as<double>(), D, Nt, op.Npad, KQm, nullptr, nullptr, h->s_Bpk.as<float>(),
            h->s_cbias.as<float>(), nullptr, h->s_cpair.as<float2>());
        prep_test_buckets_kernel<16><<<(unsigned)(op.Npad / 64), 256, 0, h->stream>>>(dV, coefG, G, D, Nt, op.Npad, KQm, KQx, h->s_Bpk.as<float>());
      }
    }
    PLDA_LAUNCH_CHECK(h);
    return PLDA_OK;
  }
  const double *psi = h->d_psi.as<double>();
  const bool zn = dzmean && dzstd;
  const int wpb = 4;
  // row side: r'_i and s_i (the z-norm map (x - zmean_i) / zstd_i folded in: s_i = 1 / zstd_i scales the
  // packed A operand and the column bias, r'_i = (r_i - zmean_i) s_i; s_i = 1 without statistics);
  // column side: q_j (0 when enrol counts differ: the second half of the contraction carries it)
  if (op.mixed) {
    if (doA)
      enrol_bias_kernel<<<(unsigned)ceil_div(M, wpb), wpb * 64, 0, h->stream>>>(
          dU, dn, n_uniform, psi, D, M, dzmean, dzstd, h->s_rbias.as<float>(), h->s_rscale.as<float>());
    if (doB) PLDA_HIP(h, hipMemsetAsync(h->s_cbias.p, 0, (size_t)op.Npad * 4, h->stream));
  } else {
    PLDA_HIP(h, h->w[11].reserve((size_t)(2 * D + 1) * 8));
    double *coef = h->w[11].as<double>();
    // the per-dimension coefficients depend on (model, count) only: kept across calls (4.7 us of launch otherwise)
    if (!(h->ucoef_ptr == coef && h->ucoef_epoch == h->model_epoch && h->ucoef_n == n_uniform && h->ucoef_D == D)) {
      uniform_coef_kernel<<<1, 256, 0, h->stream>>>(psi, D, n_uniform, coef);
      h->gcoef_ptr = nullptr;
      h->ucoef_ptr = coef; h->ucoef_epoch = h->model_epoch; h->ucoef_n = n_uniform; h->ucoef_D = D;
    }
    if (h->prep_variant != 1) {
      // one pass per side: bias, bias pair and packed operand together (prep_side_kernel; PLDA_PREP_VARIANT=1: the
      // separate kernels below, kept as the A/B arm and the reference the bit-identity test compares with; 2 / 3: this
      // path with 16 / 4 rows per wave forced)
      auto few_rows = [&](int64_t rpad) { return h->prep_variant == 3 || (h->prep_variant != 2 && rpad <= 32768); };
#define PREP_SIDE(SIDE_, RW_, RPAD_, ...) prep_side_kernel<SIDE_, RW_><<<(unsigned)((RPAD_) / (4 * RW_)), 256, 0, h->stream>>>(__VA_ARGS__)
      if (doA && doB && few_rows(op.Mpad) && few_rows(op.Npad)) {
        const PrepSideArgs a{dU, coef, M, op.Mpad, h->s_Apk.as<float>(), h->s_rbias.as<float>(), h->s_rpair.as<float2>()};
        const PrepSideArgs b{dV, coef + D, Nt, op.Npad, h->s_Bpk.as<float>(), h->s_cbias.as<float>(), h->s_cpair.as<float2>()};
        prep_both_kernel<<<(unsigned)((op.Mpad + op.Npad) / 16), 256, 0, h->stream>>>(a, b, (int)(op.Mpad / 16), coef + 2 * D, n_uniform, psi, D, op.KQ,
                                                                                     dzmean, dzstd, h->s_rscale.as<float>());
        PLDA_LAUNCH_CHECK(h);
        return PLDA_OK;
      }
      if (doA) {
        if (few_rows(op.Mpad))
          PREP_SIDE(0, 4, op.Mpad, dU, coef, coef + 2 * D, n_uniform, psi, D, M, op.Mpad, op.KQ, dzmean, dzstd, h->s_Apk.as<float>(),
                    h->s_rbias.as<float>(), h->s_rscale.as<float>(), h->s_rpair.as<float2>());
        else
          PREP_SIDE(0, 16, op.Mpad, dU, coef, coef + 2 * D, n_uniform, psi, D, M, op.Mpad, op.KQ, dzmean, dzstd, h->s_Apk.as<float>(),
                    h->s_rbias.as<float>(), h->s_rscale.as<float>(), h->s_rpair.as<float2>());
      }
      if (doB) {
        if (few_rows(op.Npad))
          PREP_SIDE(1, 4, op.Npad, dV, coef + D, coef + 2 * D, 0, psi, D, Nt, op.Npad, op.KQ, nullptr, nullptr, h->s_Bpk.as<float>(),
                    h->s_cbias.as<float>(), nullptr, h->s_cpair.as<float2>());
        else
          PREP_SIDE(1, 16, op.Npad, dV, coef + D, coef + 2 * D, 0, psi, D, Nt, op.Npad, op.KQ, nullptr, nullptr, h->s_Bpk.as<float>(),
                    h->s_cbias.as<float>(), nullptr, h->s_cpair.as<float2>());
      }
#undef PREP_SIDE
      PLDA_LAUNCH_CHECK(h);
      return PLDA_OK;
    }
    if (doA)
      weighted_sq_bias_kernel<<<(unsigned)ceil_div(M, wpb), wpb * 64, 0, h->stream>>>(
          dU, coef, 1.0, coef + 2 * D, D, M, dzmean, dzstd, h->s_rbias.as<float>(), h->s_rscale.as<float>());
    if (doB)
      weighted_sq_bias_kernel<<<(unsigned)ceil_div(Nt, wpb), wpb * 64, 0, h->stream>>>(
          dV, coef + D, 0.0, coef + 2 * D, D, Nt, nullptr, nullptr, h->s_cbias.as<float>(), nullptr);
  }
  {
    const int64_t ma = doA ? M : 0, nb = doB ? Nt : 0;
    bias_pairs_kernel<<<(unsigned)ceil_div(std::max(ma, nb), 256), 256, 0, h->stream>>>(
        h->s_rbias.as<float>(), h->s_rscale.as<float>(), ma, h->s_cbias.as<float>(), nb,
        h->s_rpair.as<float2>(), h->s_cpair.as<float2>());
  }
  PLDA_LAUNCH_CHECK(h);
  const float *rs = zn ? h->s_rscale.as<float>() : nullptr;      // folded into the packed A operand
  const dim3 ga((unsigned)(op.Mpad / 64), (unsigned)ceil_div(op.Kg, 32));
  const dim3 gb((unsigned)(op.Npad / 64), (unsigned)ceil_div(op.Kg, 32));
  if (op.mixed) {
    if (doA)
      pack_kernel<1><<<ga, 256, 0, h->stream>>>(dU, dn, n_uniform, psi, rs, D, Dp, M, op.Mpad,
                                                op.KQ, h->s_Apk.as<float>());
    if (doB)
      pack_kernel<3><<<gb, 256, 0, h->stream>>>(dV, nullptr, 0, psi, nullptr, D, Dp, Nt, op.Npad,
                                                op.KQ, h->s_Bpk.as<float>());
  } else {
    if (doA)
      pack_kernel<0><<<ga, 256, 0, h->stream>>>(dU, dn, n_uniform, psi, rs, D, Dp, M, op.Mpad,
                                                op.KQ, h->s_Apk.as<float>());
    if (doB)
      pack_kernel<2><<<gb, 256, 0, h->stream>>>(dV, nullptr, 0, psi, nullptr, D, Dp, Nt, op.Npad,
                                                op.KQ, h->s_Bpk.as<float>());
  }
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

template <int EPI>
static int launch_gemm(plda_handle *h, const TrialOperands &op, int64_t M, int64_t Nt, float *dout, int64_t ld, const float *shift, double *colsum, double *colsq) {
  // M may be a row prefix of the packed operand (z-norm pilot): only its tiles are launched
  const int tilesM = (int)ceil_div(M, 128), tilesN = (int)(op.Npad / 128);
  const int patchesM = (int)ceil_div(tilesM, PATCH_M), patchesN = (int)ceil_div(tilesN, PATCH_N);
  const int64_t numPatches = (int64_t)patchesM * patchesN;
  const int64_t grid = round_up(numPatches, 8) * PATCH_M * PATCH_N;
  if (grid > 0x7fffffffLL) return fail(h, PLDA_E_INVAL, "score_matrix: block too large, shard it");
  TraceScope ts(h, EPI == 0 ? "score.trials_gemm (K5)" : "score.znorm_stats_gemm (K8)", 2.0 * (double)op.Kg_alg * (double)M * (double)Nt, 1);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (h->prof_on) {
    if (h->prof_used == h->prof_events.size()) {
      PLDA_HIP(h, hipEventCreate(&ev0));
      PLDA_HIP(h, hipEventCreate(&ev1));
      h->prof_events.emplace_back(ev0, ev1);
    }
    ev0 = h->prof_events[h->prof_used].first;
    ev1 = h->prof_events[h->prof_used].second;
    h->prof_used++;
    h->prof_flop += 2.0 * (double)op.Kg_alg * (double)M * (double)Nt;
    PLDA_HIP(h, hipEventRecord(ev0, h->stream));
  }
  // opt-in arm: the contraction as three bf16 terms per operand (score_bf16x3.inc; PLDA_SCORE_DTYPE=bf16x3)
  if (EPI == 0 && h->score_dtype == 1 && ld < (1ll << 22) &&
      (int64_t)3 * ((int)round_up(op.KQ, 4) / 2) * std::max(op.Mpad, op.Npad) * 16 < (1ll << 32)) {
    const int KO = (int)round_up(op.KQ, 4) / 2, nsteps = KO / 2;          // k-octs (even), 16-k steps
    PLDA_HIP(h, h->s_A16.reserve((size_t)3 * KO * op.Mpad * 16));
    PLDA_HIP(h, h->s_B16.reserve((size_t)3 * KO * op.Npad * 16));
    split_bf16x3_kernel<<<dim3((unsigned)(op.Mpad / 256), (unsigned)KO), 256, 0, h->stream>>>(h->s_Apk.as<f32x4>(), op.Mpad, op.KQ, h->s_A16.as<f32x4>());
    split_bf16x3_kernel<<<dim3((unsigned)(op.Npad / 256), (unsigned)KO), 256, 0, h->stream>>>(h->s_Bpk.as<f32x4>(), op.Npad, op.KQ, h->s_B16.as<f32x4>());
    static DeviceOnce once;
    if (once.needed(h->device)) {
      const void *fns[] = {reinterpret_cast<const void *>(&trials_gemm_bf16x3_kernel<0>),
#if PLDA_DIAG
                           reinterpret_cast<const void *>(&trials_gemm_bf16x3_kernel<4>), reinterpret_cast<const void *>(&trials_gemm_bf16x3_kernel<8>),
                           reinterpret_cast<const void *>(&trials_gemm_bf16x3_kernel<12>), reinterpret_cast<const void *>(&trials_gemm_bf16x3_kernel<16>),
#endif
      };
      for (const void *f : fns) PLDA_HIP(h, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, B3_LDS));
      once.done(h->device);
    }
    const int b3M = (int)ceil_div(M, 256), b3N = (int)ceil_div(Nt, 128);       // 256 x 128 tiles (Npad is a multiple of 256)
    const int pM = (int)ceil_div(b3M, B3_PR), pN = (int)ceil_div(b3N, B3_PC);
    // the walk of the patches: as bt4_schedule chooses it (which operand repeats, and whether the Infinity Cache holds it)
    const size_t keep3 = (size_t)128 << 20, bytesA3 = (size_t)3 * KO * op.Mpad * 16, bytesB3 = (size_t)3 * KO * op.Npad * 16;
    const int colwalk3 = h->gemm_variant == 48 ? 0 : h->gemm_variant == 49 ? 1 : (bytesA3 <= keep3 && bytesB3 > keep3) ? 1 : 0;
    h->last_kernel = "trials_gemm_bf16x3_kernel";
#define B3L(MODE_)                                                                                                                      \
  trials_gemm_bf16x3_kernel<MODE_><<<256, 512, B3_LDS, h->stream>>>(h->s_A16.as<f32x4>(), h->s_B16.as<f32x4>(), (unsigned)op.Mpad, (unsigned)op.Npad,  \
                                                                    nsteps, h->s_rbias.as<float>(), h->s_rscale.as<float>(), h->s_cbias.as<float>(), dout, \
                                                                    ld, M, Nt, b3M, b3N, pM, pN, colwalk3, h->timeline.as<unsigned long long>())
#if PLDA_DIAG      // measurement arms: only in the diagnostic build (plda_create refuses their variants otherwise)
    if (h->gemm_variant == 63) {                  // the product kernel + clock stamps of workgroup 0 (plda_profile_timeline)
      PLDA_HIP(h, h->timeline.reserve(TIMELINE_WORDS * 8));
      B3L(16);
      h->timeline_valid = true;
    } else if (h->gemm_variant == 54) B3L(4);            // bounding arms (timing only): no DMA / no stores / neither
    else if (h->gemm_variant == 58) B3L(8);
    else if (h->gemm_variant == 62) B3L(12);
    else
#endif
      B3L(0);
#undef B3L
    PLDA_LAUNCH_CHECK(h);
    if (ev1) PLDA_HIP(h, hipEventRecord(ev1, h->stream));
    return PLDA_OK;
  }
  // persistent 256 x 256 kernel: when there are enough tiles to keep 256 CUs busy (PLDA_GEMM_VARIANT=20
  // forces the 128 x 128 kernel, 30 the 256 x 256 one, 31 its timeline-instrumented instantiation)
  const int btM = (int)ceil_div(M, 256), btN = (int)(op.Npad / 256);
  // Which kernel (round 5: measured over sizes with scripts/gemm_sweep.py and scripts/score_size_curve.py under PLDA_GEMM_VARIANT
  // 20 / 30 / 40).  The one-wave-per-SIMD kernel is the fastest per tile but pays ~30 us per launch (cold first stages, the
  // last tiles' 64 MB of stores, all workgroups ending together): it wins from ~1 700 tiles of 256 x 256.  Between 512 and
  // 1 700 tiles the two-waves kernel is 3-9 % ahead of both others (8192 x 8192 x 200: 0.237 ms against 0.249 / 0.252;
  // 2048 x 40000: 0.284 against 0.311 / 0.310) -- when its static 4 x 8 patches are mostly full: 512 x 200000 (two tile rows)
  // takes 0.73 ms on it and 0.49 on the others.  Below 512 tiles the 128 x 128 kernel.  PLDA_GEMM_VARIANT=50: the thresholds
  // of rounds 4-5a (1 024 tiles for both 256 x 256 kernels).
  const int64_t tiles = (int64_t)btM * btN;
  const bool patches_full = (double)tiles >= 0.85 * (double)(round_up(btM, BPR) * round_up(btN, BPC));
  const bool old_rule = h->gemm_variant == 50;
  const bool big = EPI == 0 && tiles >= (old_rule ? 1024 : 1700);                       // -> one wave per SIMD (given K >= 72)
  const bool big2 = EPI == 0 && (old_rule ? tiles >= 1024 : (tiles >= 1700 || (tiles >= 512 && patches_full)));   // -> two waves per SIMD otherwise
  // it needs 32-bit byte offsets into each packed operand and into a tile's output rows
  const bool fits4g = (int64_t)(op.KQ + 8) * op.Mpad * 16 < (1ll << 32) && (int64_t)(op.KQ + 8) * op.Npad * 16 < (1ll << 32);
  const bool use_bt2 = EPI == 0 && fits4g && ld < (1ll << 22) &&
                       ((h->gemm_variant >= 30 && h->gemm_variant <= 37) || ((h->gemm_variant == 0 || old_rule) && big2));
  // one wave per SIMD, 128 x 128 per wave (score_bt4.inc) -- the product path of every BASELINE configuration since round 4
  // (>= 1 700 tiles of 256 x 256 -- see above -- and K >= 72); PLDA_GEMM_VARIANT 40 forces it, 30 forces the round-2/3 kernel, 41 its timeline
  // instantiation, 44 / 45 / 46 its bounding arms (no DMA / no stores / neither); needs >= 3 stages per tile (K >= 72).
  {
    const int nsteps = op.KQ >> 1, nst = (nsteps + 3) >> 2;
    const bool use_bt4 = EPI == 0 && fits4g && ld < (1ll << 22) && nst >= 3 && M < (1ll << 31) && Nt < (1ll << 31) &&
                         (h->gemm_variant == 40 || h->gemm_variant == 41 || h->gemm_variant == 48 || h->gemm_variant == 49 || (h->gemm_variant >= 44 && h->gemm_variant <= 47) || ((h->gemm_variant == 0 || old_rule) && big));
    if (use_bt4) {
      const int sbase = nsteps / nst, fs = sbase + (nsteps - sbase * nst > 0 ? 1 : 0);
      h->last_kernel = "trials_gemm_bt4_kernel";
      Bt4Table *tb = nullptr;
      PLDA_TRY(bt4_schedule(h, btM, btN, op.KQ, &tb));
      // tiles that cross the matrix edge are written whole into scratch slots and copied out behind the launch
      const int rag_m = (M & 255) != 0, rag_n = (Nt & 255) != 0;
      const int fslots = (rag_m || rag_n) ? btM + btN : 0;
      PLDA_HIP(h, h->bt4_fringe.reserve(std::max<size_t>((size_t)fslots * 65536 * 4, 256)));
      Bt4Queues qs;
      for (int x = 0; x < 8; ++x) { qs.qbase[x] = tb->qbase[x]; qs.qlen[x] = tb->qlen[x]; }
      if (!h->bt4_attr_set) {
        const void *fns[] = {reinterpret_cast<const void *>(&trials_gemm_bt4_kernel<3, 0>), reinterpret_cast<const void *>(&trials_gemm_bt4_kernel<4, 0>),
#if PLDA_DIAG
                             reinterpret_cast<const void *>(&trials_gemm_bt4_kernel<3, 1>), reinterpret_cast<const void *>(&trials_gemm_bt4_kernel<4, 1>),
                             reinterpret_cast<const void *>(&trials_gemm_bt4_kernel<4, 4>), reinterpret_cast<const void *>(&trials_gemm_bt4_kernel<4, 8>),
                             reinterpret_cast<const void *>(&trials_gemm_bt4_kernel<4, 12>), reinterpret_cast<const void *>(&trials_gemm_bt4_kernel<4, 16>),
                             reinterpret_cast<const void *>(&trials_gemm_bt4_kernel<3, 16>),
#endif
        };
        for (const void *f : fns) PLDA_HIP(h, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, BT4_LDS));
        h->bt4_attr_set = true;
      }
#define BT4L(FS_, MODE_, DBG_)                                                                            \
  trials_gemm_bt4_kernel<FS_, MODE_><<<256, 256, BT4_LDS, h->stream>>>(                                   \
      h->s_Apk.as<f32x4>(), h->s_Bpk.as<f32x4>(), (unsigned)op.Mpad, (unsigned)op.Npad, op.KQ,            \
      h->s_rpair.as<float2>(), h->s_cpair.as<float2>(), dout, ld, (int)M, (int)Nt, h->bt4_fringe.as<float>(), tb->tab.as<int2>(), h->bt4_cnt.as<unsigned>(), qs, DBG_)
#if PLDA_DIAG
      if (h->gemm_variant == 41) {
        PLDA_HIP(h, h->timeline.reserve(TIMELINE_WORDS * 8));
        PLDA_HIP(h, hipMemsetAsync(h->timeline.p, 0, TIMELINE_WORDS * 8, h->stream));
        if (fs == 3) BT4L(3, 1, h->timeline.as<unsigned long long>());
        else BT4L(4, 1, h->timeline.as<unsigned long long>());
        h->timeline_valid = true;
      } else if (h->gemm_variant == 47) {     // the product kernel + clock stamps of workgroup 0
        PLDA_HIP(h, h->timeline.reserve(TIMELINE_WORDS * 8));
        if (fs == 3) BT4L(3, 16, h->timeline.as<unsigned long long>());
        else BT4L(4, 16, h->timeline.as<unsigned long long>());
        h->timeline_valid = true;
      } else if (h->gemm_variant >= 44 && h->gemm_variant <= 46 && fs == 4) {
        if (h->gemm_variant == 44) BT4L(4, 4, nullptr);
        else if (h->gemm_variant == 45) BT4L(4, 8, nullptr);
        else BT4L(4, 12, nullptr);
      } else
#endif
      if (fs == 3) {
        BT4L(3, 0, nullptr);
      } else {
        BT4L(4, 0, nullptr);
      }
#undef BT4L
      if (fslots)
        fringe_copy_kernel<<<(unsigned)(fslots * 8), 256, 0, h->stream>>>(h->bt4_fringe.as<float>(), dout, ld, (int)M, (int)Nt, btM, btN, rag_m, rag_n);
      PLDA_LAUNCH_CHECK(h);
      if (ev1) PLDA_HIP(h, hipEventRecord(ev1, h->stream));
      return PLDA_OK;
    }
  }
  if (use_bt2) {
    h->last_kernel = "trials_gemm_bt2_kernel";
    const int pM = (int)ceil_div(btM, BPR), pN = (int)ceil_div(btN, BPC);
    if (!h->bt2_attr_set) {
      const void *fns[] = {reinterpret_cast<const void *>(&trials_gemm_bt2_kernel<0>), reinterpret_cast<const void *>(&trials_gemm_bt2_kernel<2>),
#if PLDA_DIAG
                           reinterpret_cast<const void *>(&trials_gemm_bt2_kernel<1>), reinterpret_cast<const void *>(&trials_gemm_bt2_kernel<3>),
                           reinterpret_cast<const void *>(&trials_gemm_bt2_kernel<4>), reinterpret_cast<const void *>(&trials_gemm_bt2_kernel<8>),
                           reinterpret_cast<const void *>(&trials_gemm_bt2_kernel<12>), reinterpret_cast<const void *>(&trials_gemm_bt2_kernel<16>),
#endif
      };
      for (const void *f : fns) PLDA_HIP(h, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, BT2_LDS));
      h->bt2_attr_set = true;
    }
#define BT2L(MODE_, DBG_)                                                                                 \
  trials_gemm_bt2_kernel<MODE_><<<256, 512, BT2_LDS, h->stream>>>(                                        \
      h->s_Apk.as<f32x4>(), h->s_Bpk.as<f32x4>(), (unsigned)op.Mpad, (unsigned)op.Npad, op.KQ,            \
      h->s_rpair.as<float2>(), h->s_cpair.as<float2>(), dout, ld, M, Nt, btM, btN, pN, pM * pN, DBG_)
#if PLDA_DIAG
    if (h->gemm_variant == 31 || h->gemm_variant == 33) {
      // diagnostic: per-wave timestamps of workgroup 0 (plda_profile_timeline)
      PLDA_HIP(h, h->timeline.reserve(TIMELINE_WORDS * 8));
      PLDA_HIP(h, hipMemsetAsync(h->timeline.p, 0, TIMELINE_WORDS * 8, h->stream));
      if (h->gemm_variant == 31) BT2L(1, h->timeline.as<unsigned long long>());
      else BT2L(3, h->timeline.as<unsigned long long>());
      h->timeline_valid = true;
    } else if (h->gemm_variant == 37) {       // the product kernel + clock stamps of workgroup 0
      PLDA_HIP(h, h->timeline.reserve(TIMELINE_WORDS * 8));
      BT2L(16, h->timeline.as<unsigned long long>());
      h->timeline_valid = true;
    } else if (h->gemm_variant == 34) {       // bounding arms: timing only
      BT2L(4, nullptr);
    } else if (h->gemm_variant == 35) {
      BT2L(8, nullptr);
    } else if (h->gemm_variant == 36) {
      BT2L(12, nullptr);
    } else
#endif
    if (h->gemm_variant == 32) {              // the LDS-transpose epilogue (A/B arm with the same scores: test_gpu_bigtile.py)
      BT2L(2, nullptr);
    } else {
      BT2L(0, nullptr);
    }
#undef BT2L
    PLDA_LAUNCH_CHECK(h);
    if (ev1) PLDA_HIP(h, hipEventRecord(ev1, h->stream));
    return PLDA_OK;
  }
  // Kernel instantiations.  Variant 0 is the product configuration; the others are the
  // tuning / ablation arms of scripts/gemm_sweep.py (PLDA_GEMM_VARIANT), kept because the
  // numbers in DESIGN.md section 3 come from them.
#define TG(NKQ_, EPI_, MINW_, ABL_)                                                                     \
  trials_gemm_kernel<NKQ_, EPI_, MINW_, ABL_><<<(unsigned)grid, 256, 0, h->stream>>>(                   \
      h->s_Apk.as<f32x4>(), h->s_Bpk.as<f32x4>(), op.Mpad, op.Npad, op.KQ, h->s_rbias.as<float>(),      \
      h->s_rscale.as<float>(), h->s_cbias.as<float>(), dout,                                           \
      ld, M, Nt, tilesM, tilesN, patchesN, (int)numPatches, shift, colsum, colsq)
  h->last_kernel = "trials_gemm_kernel";
#if PLDA_DIAG
  constexpr int EPI_NOSTORE = (EPI == 0) ? 2 : EPI;
#endif
  switch (h->gemm_variant) {
#if PLDA_DIAG   // tuning / ablation arms of scripts/gemm_sweep.py: diagnostic build only
    case 1: TG(8, EPI, 2, 0); break;             // stage depth 32 k
    case 2: TG(6, EPI, 2, 0); break;             // 24 k
    case 3: TG(6, EPI, 3, 0); break;             // 24 k, 3 workgroups / CU
    case 4: TG(4, EPI, 3, 0); break;             // 16 k, 3 workgroups / CU
    case 9: TG(8, EPI_NOSTORE, 2, 0); break;     // ablation: no output stores
    case 10: TG(8, EPI_NOSTORE, 2, 1); break;    // ... and no in-loop LDS reads
    case 11: TG(8, EPI_NOSTORE, 2, 2); break;    // ... and no in-loop DMA
    case 12: TG(8, EPI_NOSTORE, 2, 4); break;    // ... and no stage barriers
#endif
    default: TG(10, EPI, 2, 0); break;           // product: 40 k per stage, 2 workgroups / CU
  }
#undef TG
  PLDA_LAUNCH_CHECK(h);
  if (ev1) PLDA_HIP(h, hipEventRecord(ev1, h->stream));
  return PLDA_OK;
}

// Content fingerprint of a prepared test side: 64 rows spread over [0, Nt) (first and last included), every element's
// bits mixed with its position (splitmix64) and summed.  plda_score_prepare_dev records it; a later call that would reuse
// the packed operand recomputes it and refuses on a mismatch -- the cache is keyed on the POINTER, and a caching allocator
// hands the same address to another tensor, or the caller updates rows in place (round-3 review, weak 10 / advisor).
__global__ __launch_bounds__(256) void fingerprint_kernel(const double *__restrict__ V, int64_t Nt, int D, unsigned long long *__restrict__ out) {
  unsigned long long acc = 0;
  for (int j = 0; j < 64; ++j) {
    const int64_t row = Nt <= 64 ? j : (int64_t)j * (Nt - 1) / 63;   // (Nt < 2^56)
    if (row >= Nt) break;
    for (int d = threadIdx.x; d < D; d += 256) {
      unsigned long long z = (unsigned long long)__double_as_longlong(V[row * D + d]) + 0x9e3779b97f4a7c15ull * (unsigned long long)(j * 4096 + d + 1);
      z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
      z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
      acc += z ^ (z >> 31);
    }
  }
  __shared__ unsigned long long part[256];
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = part[0];
}

// (synchronises the handle's stream: only plda_score_prepare_dev and calls that REUSE a prepared test side pay it)
static int test_side_fingerprint(plda_handle *h, const double *dV, int64_t Nt, unsigned long long *fp) {
  PLDA_HIP(h, h->w[12].reserve(8));
  fingerprint_kernel<<<1, 256, 0, h->stream>>>(dV, Nt, h->Dout, h->w[12].as<unsigned long long>());
  PLDA_LAUNCH_CHECK(h);
  PLDA_HIP(h, hipMemcpyAsync(fp, h->w[12].p, 8, hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  return PLDA_OK;
}

// ---- the distinct enrol counts of a call ----
// host array: no device work at all
void score_count_set_host(const int32_t *n, int64_t M, CountSet *cs) {
  cs->G = 0;
  std::vector<unsigned char> present(CS_NMAX + 1, 0);
  for (int64_t i = 0; i < M; ++i) {
    const int v = n[i];
    if (v < 1 || v > CS_NMAX) return;          // unusable: the depth-2D form
    present[v] = 1;
  }
  int G = 0;
  for (int v = 1; v <= CS_NMAX; ++v)
    if (present[v]) { if (G == CS_MAX) { cs->G = 0; return; } cs->vals[G++] = v; }
  cs->G = G;
}
// device array: two small kernels, 264 bytes to a pinned landing area and ONE wait for the handle's stream (the host has
// to know G before it can size the operands; callers that know the counts pass them instead: plda_score_matrix does).
int score_count_set_device(plda_handle *h, const int32_t *dn, int64_t M, CountSet *cs) {
  cs->G = 0;
  if (h->mixed_variant == 1) return PLDA_OK;
  TraceScope ts(h, "score.count_set");
  if (!h->cs_pin) {
    PLDA_HIP(h, hipHostMalloc(&h->cs_pin, sizeof(CountSetHost), hipHostMallocCoherent | hipHostMallocMapped));
    std::memset(h->cs_pin, 0, sizeof(CountSetHost));
  }
  if (M <= CS_ONE_WG) {
    CountSetHost *hp = static_cast<CountSetHost *>(h->cs_pin);
    const int seq = ++h->cs_seq;
    count_set_one_wg_kernel<<<1, 1024, 0, h->stream>>>(dn, (int)M, seq, hp);
    PLDA_LAUNCH_CHECK(h);
    // the kernel's last store is `seq`: poll it for a while (a few us after the kernel ends), then fall back to the stream
    volatile int *const flag = &hp->seq;
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(300)) {
      if (*flag == seq) { seen = true; break; }
    }
    if (!seen) PLDA_HIP(h, hipStreamSynchronize(h->stream));
    std::atomic_thread_fence(std::memory_order_acquire);
  } else {
    const size_t off_flags = round_up(CS_NMAX + 1, 16), off_out = off_flags + 16;
    PLDA_HIP(h, h->cs_work.reserve(off_out + sizeof(CountSetDev)));
    unsigned char *present = h->cs_work.as<unsigned char>();
    int *flags = reinterpret_cast<int *>(present + off_flags);
    CountSetDev *dres = reinterpret_cast<CountSetDev *>(present + off_out);
    PLDA_HIP(h, hipMemsetAsync(present, 0, off_out, h->stream));
    count_presence_kernel<<<(unsigned)std::min<int64_t>(ceil_div(M, 256), 1024), 256, 0, h->stream>>>(dn, M, present, flags);
    count_compact_kernel<<<1, 256, 0, h->stream>>>(present, flags, dres);
    PLDA_LAUNCH_CHECK(h);
    PLDA_HIP(h, hipMemcpyAsync(h->cs_pin, dres, sizeof(CountSetDev), hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
  }
  const CountSetDev *r = static_cast<const CountSetDev *>(h->cs_pin);
  if (r->overflow || r->G < 1 || r->G > CS_MAX) return PLDA_OK;
  cs->G = r->G;
  for (int g = 0; g < r->G; ++g) cs->vals[g] = r->vals[g];
  return PLDA_OK;
}
static bool count_subset(const CountSet &a, const CountSet &b) {   // a's counts all in b (both ascending)
  int j = 0;
  for (int i = 0; i < a.G; ++i) {
    while (j < b.G && b.vals[j] < a.vals[i]) ++j;
    if (j == b.G || b.vals[j] != a.vals[i]) return false;
  }
  return true;
}

// reuse_packed_B: the test side (dV, Nt, enrol-count kind) is the one the previous call on this handle
// packed -- the host entry point scores one test set against successive row slabs
// cs_in: the distinct enrol counts of the CALLER's whole call (host slabs, sharded super-blocks: every slab must see the
// same set, the test side is packed once); nullptr: found here from dn
int score_matrix_device(plda_handle *h, const double *dU, const int32_t *dn, int n_uniform, int64_t M,
                        const double *dV, int64_t Nt, const double *dzmean, const double *dzstd,
                        float *dout, int64_t ld, bool reuse_packed_B, const CountSet *cs_in) {
  if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "score_matrix: model not fitted");
  if (M <= 0 || Nt <= 0) return PLDA_OK;
  if (!dU || !dV || !dout || ld < Nt) return fail(h, PLDA_E_INVAL, "score_matrix: bad argument");
  if (!dn && n_uniform <= 0) return fail(h, PLDA_E_INVAL, "score_matrix: n_uniform must be > 0 when n_enrol is NULL");
  const int D = h->Dout;
  const bool zn = dzmean && dzstd;
  // mixed counts: the set of distinct counts decides the form of the operands
  CountSet cs_local, cs_use;
  const CountSet *cs = nullptr;
  if (dn) {
    // a prepared depth-2D test side is what the caller asked to reuse: no count set needed
    const bool prep2d = h->prep_valid && h->prep_kind == 1 && h->prep_dV == dV && h->prep_Nt == Nt && h->prep_epoch == h->model_epoch;
    if (cs_in) cs = cs_in;
    else if (!prep2d) { PLDA_TRY(score_count_set_device(h, dn, M, &cs_local)); cs = &cs_local; }
    if (cs && cs->G == 1) { n_uniform = cs->vals[0]; dn = nullptr; cs = nullptr; }   // one distinct count IS the uniform path
  }
  // a test side packed ahead of time by plda_score_prepare[_counts]_dev (same rows, same model, a matching kind of enrol
  // counts).  (Also consulted by the later slabs of a blocked call -- reuse_packed_B already set -- so that their enrol
  // side is packed in the form the first slab found on the test side.)
  bool try_prep = false;
  if (h->prep_valid && h->prep_dV == dV && h->prep_Nt == Nt && h->prep_epoch == h->model_epoch) {
    if (!dn) try_prep = h->prep_kind == 0 && h->prep_nuniform == n_uniform;
    else if (h->prep_kind == 1) { try_prep = true; cs = nullptr; }
    else if (h->prep_kind == 2 && cs && buckets_usable(h, cs) && count_subset(*cs, h->prep_counts)) { try_prep = true; cs_use = h->prep_counts; cs = &cs_use; }
  }
  if (try_prep && !reuse_packed_B) {
    // the cache is keyed on the POINTER: a caching allocator hands the same address to another tensor, or the caller
    // updates rows in place -- a content fingerprint mismatch is a cache MISS (the rows are packed again), not an error.
    // (A prepared side fits one column block: score_prepare_device refuses others.)
    unsigned long long fp = 0;
    PLDA_TRY(test_side_fingerprint(h, dV, Nt, &fp));
    if (fp == h->prep_fp) reuse_packed_B = true;
    else {
      h->prep_valid = false;
      if (dn) {   // back to the call's own count set (a prepared side may have widened it, or made it unnecessary)
        if (!cs_in && cs_local.G == 0) PLDA_TRY(score_count_set_device(h, dn, M, &cs_local));
        cs = cs_in ? cs_in : &cs_local;
        if (cs->G == 1) { n_uniform = cs->vals[0]; dn = nullptr; cs = nullptr; }
      }
    }
  }
  // The 256 x 256 kernel addresses a packed operand with 32-bit byte offsets: a side whose packed form
  // (KQ + 8 planes of 16 B per row) would reach 4 GiB is scored in row / column blocks, each side
  // packed once per block of its own dimension.  (C3's 1 M x 512 test side is 2.2 GB: one block.)
  int64_t kq8 = operand_kq(h, dn != nullptr, cs) + 8;
  if (h->score_dtype == 1) kq8 = std::max<int64_t>(kq8, 3 * std::max<int64_t>(round_up(kq8 - 8, 4) / 2, 4));   // the bf16 x 3 planes of the opt-in arm: 24 B per 4 k
  const int64_t cap = (((1ll << 32) - 1) / (kq8 * 16)) / 256 * 256;      // rows of one block
  const int64_t nrb = ceil_div(M, cap), ncb = ceil_div(Nt, cap);
  h->last_M = M; h->last_Nt = Nt; h->last_k = (int)(dn ? (buckets_usable(h, cs) ? D + cs->G - 1 : 2 * D) : D);
  for (int64_t rb = 0; rb < nrb; ++rb) {
    const int64_t r0 = rb * cap, m = std::min(cap, M - r0);
    for (int64_t cbk = 0; cbk < ncb; ++cbk) {
      const int64_t c0 = cbk * cap, nt = std::min(cap, Nt - c0);
      TrialOperands op;
      PLDA_TRY(prepare_operands(h, dU + r0 * D, dn ? dn + r0 : nullptr, n_uniform, m, dV + c0 * D, nt,
                                zn ? dzmean + r0 : nullptr, zn ? dzstd + r0 : nullptr, op,
                                /*doA=*/cbk == 0, /*doB=*/ncb > 1 || (rb == 0 && !reuse_packed_B), cs));
      float *o = dout + r0 * ld + c0;
      PLDA_TRY(launch_gemm<0>(h, op, m, nt, o, ld, nullptr, nullptr, nullptr));
    }
  }
  return PLDA_OK;
}

// Pack the test side once for many calls (the reference's callers score one test set against enrol model after enrol
// model: scoring/scorePLDA.py:302-318): V -> k-quad packed fp32 and the column biases.  kind 0: uniform count n_uniform;
// 1: mixed counts in the depth-2D form (+ V*V); 2: mixed counts bucketed, for the distinct counts *cs (later calls whose
// counts are a subset reuse it).
int score_prepare_device(plda_handle *h, const double *dV, int64_t Nt, int kind, int n_uniform, const CountSet *cs) {
  if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "score_prepare: model not fitted");
  if (!dV || Nt <= 0) return fail(h, PLDA_E_INVAL, "score_prepare: bad argument");
  if (kind == 0 && n_uniform <= 0) return fail(h, PLDA_E_INVAL, "score_prepare: n_uniform must be > 0 for uniform enrol counts");
  if (kind == 2) {
    if (!cs || cs->G < 1) return fail(h, PLDA_E_INVAL, "score_prepare_counts: empty or invalid count list");
    if (cs->G == 1) { kind = 0; n_uniform = cs->vals[0]; }
    else if (!buckets_usable(h, cs)) kind = 1;        // too many distinct counts for this dimension: the depth-2D form
  }
  const int64_t kq8 = operand_kq(h, kind != 0, kind == 2 ? cs : nullptr) + 8;
  const int64_t cap = (((1ll << 32) - 1) / (kq8 * 16)) / 256 * 256;
  if (Nt > cap) return fail(h, PLDA_E_INVAL, "score_prepare: the packed test side would exceed 4 GiB (such calls are scored in column blocks)");
  TrialOperands op;
  static const int32_t dummy_marker = 0;
  // (only the kind of the enrol counts matters to the test side: a non-null pointer selects a mixed-count form)
  PLDA_TRY(prepare_operands(h, dV, kind != 0 ? &dummy_marker : nullptr, n_uniform, 0, dV, Nt, nullptr, nullptr, op, /*doA=*/false, /*doB=*/true,
                            kind == 2 ? cs : nullptr));
  PLDA_TRY(test_side_fingerprint(h, dV, Nt, &h->prep_fp));
  h->prep_valid = true; h->prep_dV = dV; h->prep_Nt = Nt; h->prep_epoch = h->model_epoch; h->prep_kind = kind;
  h->prep_nuniform = n_uniform;
  if (kind == 2) h->prep_counts = *cs;
  return PLDA_OK;
}

int score_pairs_device(plda_handle *h, const double *dU, const int32_t *dn, const double *dV,
                       const int64_t *de, const int64_t *dt, int64_t P, const double *dzmean,
                       const double *dzstd, double *dout, int64_t M, const CountSet *cs) {
  if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "score_pairs: model not fitted");
  if (P <= 0) return PLDA_OK;
  const int wpb = 4;
  const int D = h->Dout;
  // (PLDA_MIXED_VARIANT=1 keeps everything on the verbatim kernel: the A/B arm)
  if (cs && cs->G >= 1 && cs->G <= CS_MAX && M > 0 && P >= PAIRS_TAB_MIN && h->mixed_variant != 1) {
    const int G = cs->G, S = 2 * D + 1;
    PLDA_HIP(h, h->w[12].reserve(((size_t)G * S + D) * 8 + (size_t)M * 4));
    double *tab = h->w[12].as<double>();
    int32_t *bidx = reinterpret_cast<int32_t *>(tab + (size_t)G * S + D);
    pairs_tables_kernel<<<G + 1, 256, 0, h->stream>>>(h->d_psi.as<double>(), D, *cs, tab);
    pairs_bucket_kernel<<<(unsigned)ceil_div(M, 256), 256, 0, h->stream>>>(dn, M, *cs, bidx);
    constexpr int PPW = 4;
    score_pairs_tab_kernel<PPW><<<(unsigned)ceil_div(P, (int64_t)wpb * PPW), wpb * 64, 0, h->stream>>>(
        dU, bidx, dV, de, dt, P, tab, G, D, dzmean, dzstd, dout);
    PLDA_LAUNCH_CHECK(h);
    return PLDA_OK;
  }
  score_pairs_kernel<<<(unsigned)ceil_div(P, wpb), wpb * 64, 0, h->stream>>>(
      dU, dn, dV, de, dt, P, h->d_psi.as<double>(), D, dzmean, dzstd, dout);
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

// MPlda_norm (pldamodule.cpp:196-256), fused.  dbkg raw [Nb, Din]; dmodels transformed [M, Dout].
// ------------------------------------------------------------------------------------
// z-norm statistics by MOMENTS (default).  With the cohort on the train side (n = 1) and the models on the test
// side the LLR is S_ij = a_i . v_j + r_i + q_j  (a_i = c x_i / var, r_i = -1/2 (L + sum w x_i^2), q_j = -1/2 sum g v_j^2:
// exactly the operands of the trials GEMM, kept in fp64 here), so over the cohort
//     mean_j = abar . v_j + rbar + q_j,       var_j = [v_j; 1]^T Cov_i([a_i; r_i]) [v_j; 1]       (population variance)
// -- the statistics of every model follow from the cohort's first and second moments: one (D + 1)-wide SYRK over the
// Nb cohort rows and one M x D x D GEMM, O((Nb + M) D^2) instead of the Nb M D of scoring every pair (C5: 2e10
// against 4e12 flop), all in fp64 and with the covariance taken of CENTRED rows (no E[s^2] - E[s]^2 cancellation).
// Same numbers as MPlda_norm (pldamodule.cpp:196-256: every LLR, then mean and population std per model), closer
// to the fp64 oracle than the fused fp32 GEMM below, which stays as the A/B arm (PLDA_ZNORM_VARIANT=1).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void znorm_coef_kernel(const double *__restrict__ psi, int D,
                                                         double *__restrict__ coef /*[3 D + 1]: ca, w, g, L*/) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int d = threadIdx.x; d < D; d += 256) {
    double c, var;
    const double p = psi[d];
    llr_coef(1.0, p, c, var);
    coef[d] = c / var;
    coef[D + d] = c * c / var;
    coef[2 * D + d] = 1.0 / var - 1.0 / (1.0 + p);
    acc += log(var) - log(1.0 + p);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) coef[3 * D] = red[0];
}

// cohort rows -> At[i] = [ca x_i ; r_i]  (one wave per row)
__global__ void znorm_rows_kernel(const double *__restrict__ X, const double *__restrict__ coef, int D, int64_t R,
                                  double *__restrict__ At) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= R) return;
  const double *x = X + row * (int64_t)D;
  double *o = At + row * (int64_t)(D + 1);
  double acc = 0.0;
  for (int d = lane; d < D; d += 64) {
    const double xv = x[d];
    o[d] = coef[d] * xv;
    acc = fma(coef[D + d] * xv, xv, acc);
  }
  acc = wave_sum_f64(acc);
  if (lane == 0) o[D] = -0.5 * (acc + coef[3 * D]);
}

// the same with the pilot shift applied and the constant column appended: At[i] = [ca x_i - p ; r_i - p_D ; 1]  (D + 2 wide).
// For D + 2 > 208, where the SYRK cannot form its rows on the way into LDS (syrk_tri_kernel<true>): rows written once,
// read once by the block SYRK -- three transfers of the cohort instead of the six of the five-pass form.
__global__ void znorm_rows_shift_kernel(const double *__restrict__ X, const double *__restrict__ coef, const double *__restrict__ shift,
                                        int D, int64_t R, double *__restrict__ At) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= R) return;
  const double *x = X + row * (int64_t)D;
  double *o = At + row * (int64_t)(D + 2);
  double acc = 0.0;
  for (int d = lane; d < D; d += 64) {
    const double xv = x[d];
    o[d] = __dsub_rn(__dmul_rn(coef[d], xv), shift[d]);
    acc = fma(coef[D + d] * xv, xv, acc);
  }
  acc = wave_sum_f64(acc);
  if (lane == 0) { o[D] = -0.5 * (acc + coef[3 * D]) - shift[D]; o[D + 1] = 1.0; }
}

// column sums of a [R, C] matrix, deterministic two stages: grid (ceil(C / 64), ZS) then one block
constexpr int ZS = 128;
__global__ __launch_bounds__(256) void znorm_colsum_partial_kernel(const double *__restrict__ A, int64_t R, int C,
                                                                   double *__restrict__ part /*[ZS][C]*/) {
  __shared__ double red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  const int64_t per = (R + ZS - 1) / ZS, r0 = (int64_t)blockIdx.y * per, r1 = r0 + per < R ? r0 + per : R;
  double acc = 0.0;
  if (c < C)
    for (int64_t r = r0 + sub; r < r1; r += 4) acc += A[r * C + c];
  red[sub][threadIdx.x & 63] = acc;
  __syncthreads();
  if (sub == 0 && c < C) part[(size_t)blockIdx.y * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void znorm_colmean_kernel(const double *__restrict__ part, int C, double invR, double *__restrict__ mean) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int z = 0; z < ZS; ++z) s += part[(size_t)z * C + c];
  mean[c] = s * invR;
}
__global__ void znorm_centre_kernel(double *__restrict__ A, int64_t total, int C, const double *__restrict__ mean) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < total) A[idx] -= mean[idx % C];
}

// models: mean_j = abar . v + rbar + q_j,  var_j = v^T Cvv v + 2 v . cvr + crr  with Y_j = Cvv v_j from the GEMM
__global__ void znorm_models_kernel(const double *__restrict__ V, const double *__restrict__ Y, const double *__restrict__ coef,
                                    const double *__restrict__ mom /*[D + 1] means*/, const double *__restrict__ Cov /*[D+1][D+1]*/,
                                    int D, int64_t M, double *__restrict__ out_mean, double *__restrict__ out_std) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= M) return;
  const double *v = V + row * (int64_t)D, *y = Y + row * (int64_t)D;
  const int D1 = D + 1;
  double mq = 0.0, var = 0.0;
  for (int d = lane; d < D; d += 64) {
    const double vd = v[d];
    mq += mom[d] * vd - 0.5 * coef[2 * D + d] * vd * vd;
    var += vd * (y[d] + 2.0 * Cov[(size_t)d * D1 + D]);
  }
  mq = wave_sum_f64(mq);
  var = wave_sum_f64(var);
  if (lane == 0) {
    var += Cov[(size_t)D * D1 + D];
    out_mean[row] = mq + mom[D];
    out_std[row] = sqrt(var > 0.0 ? var : 0.0);
  }
}

// lin = 2 cvr: the covariance column between the cohort's [ca x] and its r (quadform_rows_device's linear term)
__global__ void znorm_lin_kernel(const double *__restrict__ Cov, int D, double *__restrict__ lin) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d < D) lin[d] = 2.0 * Cov[(size_t)d * (D + 1) + D];
}

// Round 5: the cohort's moments from ONE read of the transformed rows.  A pilot shift p (the mean of [ca x ; r] over the
// first <= 64 rows: one workgroup, every load in flight at once) keeps the single-pass covariance S2 / N - (S1 / N)(S1 / N)^T free of
// cancellation (the shifted-data form: exact in exact arithmetic for ANY p, and p is within an eighth of a standard
// deviation of the mean);
// syrk_tri_kernel<true> (linalg.hip) forms the shifted augmented rows on its way into LDS.  Rounds 2-4: rows -> At,
// column sums, centring in place, SYRK reading At twice = ~6x the cohort's bytes.
constexpr int ZPILOT = 64;      // pilot rows
__global__ __launch_bounds__(1024) void znorm_pilot_kernel(const double *__restrict__ X, int D, int64_t R /* <= ZPILOT */, const double *__restrict__ coef,
                                                           double *__restrict__ shift /*[D + 1]*/) {
  __shared__ double s1s[4][256], s2s[4][256];
  __shared__ double red[256];
  const int tc = threadIdx.x & 255, sub = threadIdx.x >> 8;      // column within a 256-wide slab, row phase
  double racc = 0.0;
  for (int d0 = 0; d0 < D; d0 += 256) {
    const int d = d0 + tc;
    double s1 = 0.0, s2 = 0.0;
    if (d < D) {
#pragma unroll
      for (int k = 0; k < ZPILOT / 4; ++k) {           // 16 independent loads in flight
        const int64_t i = sub + 4 * k;
        const double x = i < R ? X[i * D + d] : 0.0;
        s1 += x; s2 = fma(x, x, s2);
      }
    }
    s1s[sub][tc] = s1; s2s[sub][tc] = s2;
    __syncthreads();
    if (sub == 0 && d < D) {
      const double t1 = (s1s[0][tc] + s1s[1][tc]) + (s1s[2][tc] + s1s[3][tc]);
      const double t2 = (s2s[0][tc] + s2s[1][tc]) + (s2s[2][tc] + s2s[3][tc]);
      shift[d] = __dmul_rn(coef[d], t1 / (double)R);
      racc += coef[D + d] * t2 / (double)R;
    }
    __syncthreads();
  }
  if (threadIdx.x < 256) red[threadIdx.x] = racc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) shift[D] = -0.5 * (red[0] + coef[3 * D]);
}

// C [(D + 2)^2] = sum of outer products of the shifted augmented rows -> means [D + 1] and population covariance [D1 x D1]
__global__ void znorm_moments_kernel(const double *__restrict__ C, const double *__restrict__ shift, int D1, double invN,
                                     double *__restrict__ mom, double *__restrict__ Cov) {
  const int D2 = D1 + 1;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D1 * D1) return;
  const int a = idx / D1, b = idx % D1;
  const double ma = C[(size_t)D1 * D2 + a] * invN, mb = C[(size_t)D1 * D2 + b] * invN;     // row D1: the constant column's products = column sums
  Cov[idx] = __dsub_rn(__dmul_rn(C[(size_t)a * D2 + b], invN), __dmul_rn(ma, mb));     // (no fma: a one-row cohort has variance exactly 0)
  if (b == 0) mom[a] = shift[a] + ma;
}

static int znorm_stats_moments(plda_handle *h, const double *dT, int64_t Nb, const double *dmodels, int64_t M,
                               double *dmean, double *dstd) {
  const int D = h->Dout, D1 = D + 1, D2 = D + 2;
  PLDA_HIP(h, h->zn_y.reserve((size_t)M * D * 8));
  PLDA_HIP(h, h->zn_small.reserve(((size_t)3 * D + 1 + (size_t)ZS * D1 + D1 + (size_t)D1 * D1 + D1 + (size_t)D2 * D2) * 8));
  double *Y = h->zn_y.as<double>();
  double *coef = h->zn_small.as<double>(), *part = coef + 3 * D + 1, *mom = part + (size_t)ZS * D1, *Cov = mom + D1;
  double *shift = Cov + (size_t)D1 * D1, *C2 = shift + D1;
  const int wpb = 4;
  {
    TraceScope ts(h, "norm.cohort_moments", 2.0 * (double)Nb * D1 * D1, 1);
    znorm_coef_kernel<<<1, 256, 0, h->stream>>>(h->d_psi.as<double>(), D, coef);
    bool used = false;
    if (h->znorm_variant != 2) {      // PLDA_ZNORM_VARIANT=2: the five-pass form of rounds 2-4 (A/B arm; also what D + 2 > 208 takes)
      znorm_pilot_kernel<<<1, 1024, 0, h->stream>>>(dT, D, std::min<int64_t>(Nb, ZPILOT), coef, shift);
      PLDA_TRY(syrk_znorm_f64(h, D, Nb, dT, coef, shift, C2, &used));
      if (used) znorm_moments_kernel<<<(unsigned)ceil_div((int64_t)D1 * D1, 256), 256, 0, h->stream>>>(C2, shift, D1, 1.0 / (double)Nb, mom, Cov);
      PLDA_LAUNCH_CHECK(h);
    }
    if (!used && h->znorm_variant != 2 && D2 <= 514) {
      // wider than the one-read kernel takes: shifted augmented rows written once, the block SYRK reads them once
      PLDA_HIP(h, h->zn_rows.reserve((size_t)Nb * D2 * 8));
      double *At = h->zn_rows.as<double>();
      znorm_rows_shift_kernel<<<(unsigned)ceil_div(Nb, wpb), wpb * 64, 0, h->stream>>>(dT, coef, shift, D, Nb, At);
      PLDA_LAUNCH_CHECK(h);
      PLDA_TRY(gemm_f64(h, D2, D2, Nb, 1.0, At, 1, D2, At, D2, 1, nullptr, 0.0, C2, D2));
      znorm_moments_kernel<<<(unsigned)ceil_div((int64_t)D1 * D1, 256), 256, 0, h->stream>>>(C2, shift, D1, 1.0 / (double)Nb, mom, Cov);
      PLDA_LAUNCH_CHECK(h);
      used = true;
    }
    if (!used) {
      PLDA_HIP(h, h->zn_rows.reserve((size_t)Nb * D1 * 8));
      double *At = h->zn_rows.as<double>();
      znorm_rows_kernel<<<(unsigned)ceil_div(Nb, wpb), wpb * 64, 0, h->stream>>>(dT, coef, D, Nb, At);
      znorm_colsum_partial_kernel<<<dim3((unsigned)ceil_div(D1, 64), ZS), 256, 0, h->stream>>>(At, Nb, D1, part);
      znorm_colmean_kernel<<<(unsigned)ceil_div(D1, 256), 256, 0, h->stream>>>(part, D1, 1.0 / (double)Nb, mom);
      const int64_t total = Nb * (int64_t)D1;
      znorm_centre_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, h->stream>>>(At, total, D1, mom);
      PLDA_LAUNCH_CHECK(h);
      // population covariance of the centred rows: At^T At / Nb (the SYRK kernels of the fit)
      PLDA_TRY(gemm_f64(h, D1, D1, Nb, 1.0 / (double)Nb, At, 1, D1, At, D1, 1, nullptr, 0.0, Cov, D1));
    }
  }
  {
    TraceScope ts(h, "norm.model_statistics", 2.0 * (double)M * D * D, 1);
    // D <= 208: on the transform kernel's shape (a workgroup owns 128 models and all D columns; v^T Cvv v, the linear terms and
    // the mean come out of its epilogue -- no Y, no second kernel).  PLDA_ZNORM_VARIANT=3: the GEMM + row kernel below (A/B arm).
    if (h->znorm_variant != 3) {
      double *lin = h->zn_y.as<double>();          // (Y's buffer: D doubles of it)
      znorm_lin_kernel<<<(unsigned)ceil_div(D, 256), 256, 0, h->stream>>>(Cov, D, lin);
      bool used = false;
      PLDA_TRY(quadform_rows_device(h, dmodels, M, D, Cov, D1, lin, mom, coef + 2 * D, mom + D, Cov + (size_t)D * D1 + D, dmean, dstd, &used));
      if (used) return PLDA_OK;
    }
    // Y = V Cvv  (Cvv = the leading D x D block of Cov, row stride D + 1)
    PLDA_TRY(gemm_f64(h, M, D, D, 1.0, dmodels, D, 1, Cov, D1, 1, nullptr, 0.0, Y, D));
    znorm_models_kernel<<<(unsigned)ceil_div(M, wpb), wpb * 64, 0, h->stream>>>(dmodels, Y, coef, mom, Cov, D, M, dmean, dstd);
    PLDA_LAUNCH_CHECK(h);
  }
  return PLDA_OK;
}

int znorm_stats_device(plda_handle *h, const double *dbkg, int64_t Nb, int num_examples, int Din,
                       const double *dmodels, int64_t M, double *dmean, double *dstd) {
  if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "norm: model not fitted");
  if (Nb <= 0 || M <= 0) return fail(h, PLDA_E_INVAL, "norm: empty input");
  const int D = h->Dout;
  // cohort rows transformed with num_examples = Nb (quirk Q6, :224)
  PLDA_HIP(h, h->w[8].reserve((size_t)Nb * D * 8));
  double *dT = h->w[8].as<double>();
  if (num_examples <= 0) num_examples = (int)Nb;
  PLDA_TRY(transform_rows_device(h, dbkg, Nb, Din, nullptr, num_examples, dT));
  // cohort = train side with n = 1 (quirk Q7, :235); models = test side
  if (h->znorm_variant != 1) return znorm_stats_moments(h, dT, Nb, dmodels, M, dmean, dstd);     // (0: one-read moments; 2: five-pass moments)
  // ---- A/B arm: every LLR on the fp32 MFMA GEMM with the fused (sum, sum of squares) epilogue ----
  TrialOperands op;
  PLDA_TRY(prepare_operands(h, dT, nullptr, 1, Nb, dmodels, M, nullptr, nullptr, op));
  PLDA_HIP(h, h->w[9].reserve((size_t)op.Npad * (8 + 8 + 4)));
  double *colsum = h->w[9].as<double>();
  double *colsq = colsum + op.Npad;
  float *shift = reinterpret_cast<float *>(colsq + op.Npad);
  // pilot: mean over the first rows -> shift (keeps the single-pass variance well conditioned)
  const int64_t Np = Nb < 128 ? Nb : 128;
  PLDA_HIP(h, hipMemsetAsync(colsum, 0, (size_t)op.Npad * 20, h->stream));
  PLDA_TRY(launch_gemm<1>(h, op, Np, M, nullptr, 0, shift, colsum, colsq));
  pilot_shift_kernel<<<(unsigned)ceil_div(M, 256), 256, 0, h->stream>>>(colsum, M, 1.0 / (double)Np, shift);
  PLDA_HIP(h, hipMemsetAsync(colsum, 0, (size_t)op.Npad * 16, h->stream));
  PLDA_TRY(launch_gemm<1>(h, op, Nb, M, nullptr, 0, shift, colsum, colsq));
  znorm_finalize_kernel<<<(unsigned)ceil_div(M, 256), 256, 0, h->stream>>>(shift, colsum, colsq, M,
                                                                           1.0 / (double)Nb, dmean, dstd);
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

}  // namespace plda
