// plda_amd/csrc/linalg.hip -- fp64 building blocks of the PLDA estimator on gfx950.
//
// These replace the Kaldi matrix-library / ATLAS calls that the reference reaches
// through PldaStats / PldaEstimator: AddMat2 (syrk), AddMatMat
// (gemm), AddMat2Sp (congruence), TpMatrix::Cholesky / Invert, SpMatrix::Eig + SortSvd.
//   gemm_f64      v_mfma_f64_16x16x4_f64: LDS-staged 64x64 or 128x128 tiles with register prefetch,
//                 optional per-k weights (weighted SYRK X^T diag(w) X), deterministic split-K, batching;
//                 a panel-resident 32x32 kernel for the small K <= 256 products of the EM
//   chol_small / spd_inverse_small   Cholesky factor / SPD inverse with the matrix resident in the
//                 registers of one workgroup (D <= 256); whiten_blocked extends the Cholesky to any size by
//                 block elimination (GEMMs on the panel kernel), spd_inverse_blocked = T^T T from it
//   tri_invert    forward substitution, one wave per column, solution held in registers
//   sym_eig_f64   one-sided (Hestenes) block Jacobi: one workgroup per pair of 4-row blocks and outer
//                 round (Gram matrix on the MFMA pipe, 8x8 two-sided sweep in one wave, one apply pass);
//                 a sweep's launches are replayed from a hipGraph; optional warm start
#include <cstdlib>

#include "common.hpp"

#include <algorithm>
#include <utility>
#include <cstring>
#include <type_traits>

namespace plda {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------
// gemm_f64
// ------------------------------------------------------------------------------------
constexpr int GB = 64;    // block tile (square), small variant
constexpr int GK = 16;    // k per LDS stage
constexpr int LDK = 17;   // leading dim of an [m][k] tile (k-contiguous source)

// LDS index of element (m, k) of a TB x 16 operand tile: [m][k] for a k-contiguous source, [k][m] otherwise
template <bool KC, int TB>
__device__ __forceinline__ int tile_idx(int m, int k) { return KC ? m * LDK + k : k * (TB + 16) + m; }

// A TB (m) x 16 (k) tile of op(A) is fetched global -> registers (TB/16 doubles per thread) and
// later written registers -> LDS, so that the fetch of tile t+1 overlaps the MFMAs of tile t.
// element (m,k) = A[m*sm + k*sk] * kw[k].  The loads are UNCONDITIONAL on clamped indices (a
// predicated load becomes a branch + wait per load, which serialises the whole fetch); the
// out-of-range elements are zeroed when the tile is written to LDS.
template <bool KC, int TB>
__device__ __forceinline__ void fetch_tile(double (&r)[TB / 16], double (&w)[KC ? 1 : TB / 16],
                                           const double *__restrict__ A, int64_t sm, int64_t sk, int64_t m0,
                                           int64_t M, int64_t k0, int64_t Kend, const double *__restrict__ kw,
                                           int t) {
  if (KC) {
    const int64_t gk = min(k0 + (t & 15), Kend - 1);
    w[0] = kw ? kw[gk] : 1.0;
#pragma unroll
    for (int pass = 0; pass < TB / 16; ++pass) {
      const int64_t gm = min(m0 + (t >> 4) + pass * 16, M - 1);
      r[pass] = A[gm * sm + gk * sk];
    }
  } else {
    const int64_t gm = min(m0 + (t & (TB - 1)), M - 1);
#pragma unroll
    for (int pass = 0; pass < TB / 16; ++pass) {
      const int64_t gk = min(k0 + t / TB + pass * (256 / TB), Kend - 1);
      r[pass] = A[gm * sm + gk * sk];
      w[pass] = kw ? kw[gk] : 1.0;
    }
  }
}

template <bool KC, int TB>
__device__ __forceinline__ void store_tile(double *lds, const double (&r)[TB / 16], const double (&w)[KC ? 1 : TB / 16],
                                           int64_t m0, int64_t M, int64_t k0, int64_t Kend, int t) {
#pragma unroll
  for (int pass = 0; pass < TB / 16; ++pass) {
    if (KC) {
      const bool ok = (m0 + (t >> 4) + pass * 16 < M) && (k0 + (t & 15) < Kend);
      lds[tile_idx<true, TB>((t >> 4) + pass * 16, t & 15)] = ok ? r[pass] * w[0] : 0.0;
    } else {
      const bool ok = (m0 + (t & (TB - 1)) < M) && (k0 + t / TB + pass * (256 / TB) < Kend);
      lds[tile_idx<false, TB>(t & (TB - 1), t / TB + pass * (256 / TB))] = ok ? r[pass] * w[pass] : 0.0;
    }
  }
}

// TB = 64: 4 waves x (32 x 32); TB = 128: 4 waves x (64 x 64) -- 16 MFMAs per 8 fragment reads, for the
// large products (the scatter SYRK at D = 512, the LDA decision matrix), chosen when it still fills the chip.
template <bool AKC, bool BKC, int TB>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(int64_t M, int64_t N, int64_t K, int64_t kchunk,
                                                       double alpha, const double *__restrict__ A,
                                                       int64_t sam, int64_t sak,
                                                       const double *__restrict__ B, int64_t sbk,
                                                       int64_t sbn, const double *__restrict__ kw,
                                                       double beta, double *__restrict__ C, int64_t ldc,
                                                       double *__restrict__ part, int splits, int64_t strideA,
                                                       int64_t strideB, int64_t strideC) {
  constexpr int TM = TB / 32;                   // 16 x 16 MFMA tiles per wave and dimension
  constexpr int TILE = GK * (TB + 16);          // doubles per operand stage (covers both layouts)
  // blockIdx.z = batch * splits + split; batched calls run with splits == 1
  const int zb = (int)blockIdx.z / splits, zs = (int)blockIdx.z % splits;
  A += (int64_t)zb * strideA;
  B += (int64_t)zb * strideB;
  C += (int64_t)zb * strideC;
  __shared__ double As[2][TILE];
  __shared__ double Bs[2][TILE];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * TB, n0 = (int64_t)blockIdx.x * TB;
  const int64_t kbeg = (int64_t)zs * kchunk;
  const int64_t kend = min(K, kbeg + kchunk);

  f64x4 acc[TM][TM];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};

  const int fi = lane & 15, fk = lane >> 4;
  double ra[TB / 16], rb[TB / 16], wa[AKC ? 1 : TB / 16], wb[BKC ? 1 : TB / 16];
  fetch_tile<AKC, TB>(ra, wa, A, sam, sak, m0, M, kbeg, kend, kw, t);
  fetch_tile<BKC, TB>(rb, wb, B, sbn, sbk, n0, N, kbeg, kend, nullptr, t);
  store_tile<AKC, TB>(As[0], ra, wa, m0, M, kbeg, kend, t);
  store_tile<BKC, TB>(Bs[0], rb, wb, n0, N, kbeg, kend, t);
  __syncthreads();
  int cur = 0;
  for (int64_t k0 = kbeg; k0 < kend; k0 += GK) {
    const bool more = k0 + GK < kend;
    if (more) {   // next tile's global loads are in flight during this tile's MFMAs
      fetch_tile<AKC, TB>(ra, wa, A, sam, sak, m0, M, k0 + GK, kend, kw, t);
      fetch_tile<BKC, TB>(rb, wb, B, sbn, sbk, n0, N, k0 + GK, kend, nullptr, t);
    }
#pragma unroll
    for (int kk = 0; kk < GK / 4; ++kk) {
      double a[TM], b[TM];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) a[tm] = As[cur][tile_idx<AKC, TB>(wm * (TB / 2) + tm * 16 + fi, kk * 4 + fk)];
#pragma unroll
      for (int tn = 0; tn < TM; ++tn) b[tn] = Bs[cur][tile_idx<BKC, TB>(wn * (TB / 2) + tn * 16 + fi, kk * 4 + fk)];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TM; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    }
    if (more) {
      store_tile<AKC, TB>(As[cur ^ 1], ra, wa, m0, M, k0 + GK, kend, t);
      store_tile<BKC, TB>(Bs[cur ^ 1], rb, wb, n0, N, k0 + GK, kend, t);
    }
    __syncthreads();
    cur ^= 1;
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TM; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm * (TB / 2) + tm * 16 + (lane >> 4) + 4 * r;
        const int64_t col = n0 + wn * (TB / 2) + tn * 16 + (lane & 15);
        if (row < M && col < N) {
          if (part) {
            part[((int64_t)zs * M + row) * N + col] = acc[tm][tn][r];
          } else {
            double *c = C + row * ldc + col;
            *c = alpha * acc[tm][tn][r] + (beta != 0.0 ? beta * *c : 0.0);
          }
        }
      }
}

__global__ void splitk_reduce_kernel(const double *__restrict__ part, int splits, int64_t M, int64_t N,
                                     double alpha, double beta, double *__restrict__ C, int64_t ldc) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * N) return;
  double s = 0.0;
  for (int z = 0; z < splits; ++z) s += part[(int64_t)z * M * N + idx];
  const int64_t row = idx / N, col = idx % N;
  double *c = C + row * ldc + col;
  *c = alpha * s + (beta != 0.0 ? beta * *c : 0.0);
}

// workgroup barrier that waits for this wave's LDS operations only.  __syncthreads() also drains the vector-memory
// counter (s_waitcnt vmcnt(0)), i.e. every register prefetch in flight.  (Measured on the SYRK kernels: no change --
// the compiler's own counted waits in front of the LDS stores were already the binding ones.)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0); vmcnt and expcnt untouched
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ------------------------------------------------------------------------------------
// K2: symmetric rank-K update  S = alpha X^T diag(w) X (+ beta S)  -- the AddSamples scatter
// (PldaStats::AddSamples reached at pldamodule.cpp:94-98; Kaldi's AddMat2 / ATLAS dsyrk) and the
// scatter matrices of LDA.  X is [K, D] row-major: both GEMM operands are the SAME rows.
//
// The general kernel above treats it as a D x D x K product: at D = 200 it computes a 256 x 256 output
// (1.64 x the algorithmic flops: 128-tiles), twice (no symmetry), reads X once per operand, and its
// 250-way split-K writes 250 full-size partial slabs -- 0.35 of the fp64 MFMA peak at C2.  Here:
//   * only the LOWER-triangular 128 x 128 super-tiles (I >= J) are launched;
//   * inside a super-tile the 16 x 16 MFMA tiles that lie in the padding (>= D) or, on a diagonal
//     super-tile, strictly above the diagonal are skipped, with a wave -> tile mapping that keeps the
//     four waves balanced: diagonal super-tile: wave w owns tile rows {w, nvr-1-w} (r+1 tiles in row r);
//     off-diagonal: wave w owns tile columns {2w, 2w+1} over all valid tile rows.  At D = 200 a row chunk
//     costs 9 + 10 + 6 MFMA tile-steps per wave against 64 before; at D = 512 132 against 256;
//   * a diagonal super-tile fetches its rows from memory once for both MFMA operands;
//   * split-K writes only the valid tiles of a partial (2 KiB each), a second kernel sums the
//     splits in fixed order and mirrors the triangle -- deterministic.
// ------------------------------------------------------------------------------------
template <bool DIAG>   // (two instantiations rather than one kernel with both tile mappings: that one spilled)
__global__ __launch_bounds__(256, 2) void syrk_lower_kernel(int D, int64_t K, int64_t kchunk,
                                                            const double *__restrict__ X, int64_t ldx,
                                                            const double *__restrict__ kw,
                                                            double *__restrict__ part, int nP, int npairs,
                                                            int splits) {
  constexpr int TB = 128, LD = TB + 16;
  __shared__ double As[2][GK * LD];
  __shared__ double Bs[DIAG ? 1 : 2][DIAG ? 1 : GK * LD];   // a diagonal super-tile reads both operands from As
  __shared__ double Ws[2][GK];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // 1-D grid, XCD-aware: workgroup L runs on XCD L % 8, and all super-tile pairs of one row chunk meet in ONE L2
  // (they read the same rows of X: at D = 512 the six strictly-lower pairs use each 128-column slab three times).
  // So chunk % 8 = L % 8: L = 8 (pair + npairs (chunk / 8)) + chunk % 8.  (Measured: no change at C3 -- the kernel
  // is not bound by HBM traffic: SQ_VALU_MFMA_BUSY_CYCLES is 41 % of its cycles, 65 % of the wave cycles wait;
  // MFMAs without the per-tile branches: no change either.)
  const int Lid = (int)blockIdx.x;
  const int pair = (Lid >> 3) % npairs;
  const int chunk = 8 * ((Lid >> 3) / npairs) + (Lid & 7);
  if (chunk >= splits) return;
  // `pair` enumerates the diagonal super-tiles (DIAG) or the strictly-lower pairs (I > J) row by row;
  // `slot` is the pair's index in the partial slabs: I (I + 1) / 2 + J
  int I, J;
  if (DIAG) { I = J = pair; }
  else { I = 1; J = pair; while (J >= I) { J -= I; ++I; } }
  const int slot = I * (I + 1) / 2 + J;
  constexpr bool diag = DIAG;
  const int64_t m0 = (int64_t)I * TB, n0 = (int64_t)J * TB;
  const int nvr = min(8, (int)((D - m0 + 15) / 16)), nvc = min(8, (int)((D - n0 + 15) / 16));
  const int64_t kbeg = (int64_t)chunk * kchunk;
  const int64_t kend = min(K, kbeg + kchunk);
  // this wave's two "lines" (tile rows on a diagonal super-tile, tile columns otherwise) and, per line,
  // how many tiles of it are computed
  int line[2], cnt[2];
  if (diag) {
    line[0] = wave; line[1] = max(nvr - 1 - wave, 0);
    cnt[0] = line[0] < nvr && line[0] <= nvr - 1 - wave ? line[0] + 1 : 0;
    cnt[1] = nvr - 1 - wave > line[0] ? line[1] + 1 : 0;      // (== when nvr is odd: the middle row is line[0]'s)
  } else {
    line[0] = 2 * wave; line[1] = 2 * wave + 1;
    cnt[0] = line[0] < nvc ? nvr : 0;
    cnt[1] = line[1] < nvc ? nvr : 0;
  }
  f64x4 acc[2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};

  const int fi = lane & 15, fk = lane >> 4;
  // a 16 (k) x 128 (columns) tile of X: thread t holds column t & 127 of rows (t >> 7) + 2 pass; loads are
  // unconditional on clamped indices, out-of-range COLUMNS are zeroed on the way into LDS, out-of-range ROWS
  // get weight zero.  The row weights go to LDS next to the tile (16 per stage) and multiply the row operand
  // after its LDS read, so a diagonal super-tile keeps ONE copy of its rows for both operands.
  // Register prefetch runs TWO stages ahead: at D = 200 a stage's MFMAs (~1 300 cycles) are far shorter than a
  // memory round trip under load, and with one stage of look-ahead the loop ran at the speed of the loads
  // (9 600 cycles per stage; K2 at C2 0.49 of the fp64 peak).
  const int tc = t & 127, tk = t >> 7;
  const double *Bsel0 = diag ? As[0] : Bs[0], *Bsel1 = diag ? As[1] : Bs[1];
  auto fetch = [&](double (&r)[8], int64_t c0, int64_t k0) {
    const int64_t gm = min(c0 + tc, (int64_t)D - 1);
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int64_t gk = min(k0 + tk + pass * 2, kend - 1);
      r[pass] = X[gk * ldx + gm];
    }
  };
  auto fetch_w = [&](int64_t k0) {   // threads 0..15: the weight of row k0 + t (0 past the end of the chunk)
    double w = 0.0;
    if (t < GK && k0 + t < kend) w = kw ? kw[k0 + t] : 1.0;
    return w;
  };
  auto store = [&](double *lds, const double (&r)[8], int64_t c0) {
    const bool ok = c0 + tc < D;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) lds[(tk + pass * 2) * LD + tc] = ok ? r[pass] : 0.0;
  };
  double ra0[8], rb0[8], ra1[8], rb1[8], w0 = 0.0, w1 = 0.0;
  // prologue: stage 0 -> LDS buffer 0, stage 1 -> register set 1
  if (kbeg < kend) {
    fetch(ra0, m0, kbeg);
    if (!diag) fetch(rb0, n0, kbeg);
    w0 = fetch_w(kbeg);
    if (kbeg + GK < kend) {
      fetch(ra1, m0, kbeg + GK);
      if (!diag) fetch(rb1, n0, kbeg + GK);
      w1 = fetch_w(kbeg + GK);
    }
    store(As[0], ra0, m0);
    if (!diag) store(Bs[0], rb0, n0);
    if (t < GK) Ws[0][t] = w0;
  }
  __syncthreads();
  // one stage: issue the loads of stage k0 + 2 GK into (rn, wn), run the MFMAs of stage k0 out of LDS buffer
  // `cur`, move stage k0 + GK (loaded one iteration ago into (rs, ws)) to the other buffer
  auto stage = [&](int64_t k0, int cur, double (&rna)[8], double (&rnb)[8], double &wn, const double (&rsa)[8],
                   const double (&rsb)[8], const double &ws) {
    if (k0 + 2 * GK < kend) {
      fetch(rna, m0, k0 + 2 * GK);
      if (!diag) fetch(rnb, n0, k0 + 2 * GK);
      wn = fetch_w(k0 + 2 * GK);
    }
    const double *Ar = As[cur], *Bc = cur ? Bsel1 : Bsel0, *Wr = Ws[cur];
#pragma unroll
    for (int kk = 0; kk < GK / 4; ++kk) {
      const int kb = (kk * 4 + fk) * LD + fi;
      const double wk = Wr[kk * 4 + fk];
      if (diag) {
        double a[2], b[8];
#pragma unroll
        for (int l = 0; l < 2; ++l) a[l] = Ar[kb + line[l] * 16] * wk;
#pragma unroll
        for (int c = 0; c < 8; ++c) b[c] = Bc[kb + c * 16];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          if (c < cnt[0]) acc[0][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[c], acc[0][c], 0, 0, 0);
          if (c < cnt[1]) acc[1][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[c], acc[1][c], 0, 0, 0);
        }
      } else {
        double a[8], b[2];
#pragma unroll
        for (int r = 0; r < 8; ++r) a[r] = Ar[kb + r * 16] * wk;
#pragma unroll
        for (int l = 0; l < 2; ++l) b[l] = Bc[kb + line[l] * 16];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          if (r < cnt[0]) acc[0][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r], b[0], acc[0][r], 0, 0, 0);
          if (r < cnt[1]) acc[1][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r], b[1], acc[1][r], 0, 0, 0);
        }
      }
    }
    if (k0 + GK < kend) {
      store(As[cur ^ 1], rsa, m0);
      if (!diag) store(Bs[cur ^ 1], rsb, n0);
      if (t < GK) Ws[cur ^ 1][t] = ws;
    }
    lds_barrier();
  };
  for (int64_t k0 = kbeg; k0 < kend; k0 += 2 * GK) {
    stage(k0, 0, ra0, rb0, w0, ra1, rb1, w1);
    if (k0 + GK < kend) stage(k0 + GK, 1, ra1, rb1, w1, ra0, rb0, w0);
  }
  // partial slab of this (split, pair): 64 tile slots of 256 doubles; only computed tiles are written
  double *slab = part + ((int64_t)chunk * nP + slot) * (64 * 256);
#pragma unroll
  for (int l = 0; l < 2; ++l)
#pragma unroll
    for (int x = 0; x < 8; ++x)
      if (x < cnt[l]) {
        const int tr = diag ? line[l] : x, tc = diag ? x : line[l];
        double *d = slab + (tr * 8 + tc) * 256 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r * 64] = acc[l][x][r];
      }
}

// sum of the split partials (fixed order), alpha / beta, and the mirror image
__global__ __launch_bounds__(256) void syrk_reduce_kernel(const double *__restrict__ part, int splits_diag,
                                                          int splits_off, int nP, int D,
                                                          double alpha, double beta, double *__restrict__ C,
                                                          int64_t ldc) {
  int I = 0, J = (int)blockIdx.x / 64;
  while (J > I) { J -= I + 1; ++I; }
  const int tile = (int)blockIdx.x % 64, tr = tile >> 3, tc = tile & 7;
  const int lane = threadIdx.x & 63, reg = threadIdx.x >> 6;
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  const int gr = I * 128 + tr * 16 + (lane >> 4) + 4 * reg, gc = J * 128 + tc * 16 + (lane & 15);
  if (gr >= D || gc >= D || gc > gr) return;            // padding, or above the diagonal (skipped or mirrored)
  const double *p = part + (int64_t)(blockIdx.x / 64) * (64 * 256) + tile * 256 + reg * 64 + lane;
  const int splits = I == J ? splits_diag : splits_off;
  // eight loads in flight per thread (a plain loop over up to 512 splits is a chain of memory latencies);
  // the order of the additions is fixed, so the result is deterministic
  const int64_t zs = (int64_t)nP * (64 * 256);
  double s8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int z = 0;
  for (; z + 8 <= splits; z += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s8[u] += p[(int64_t)(z + u) * zs];
  }
  for (; z < splits; ++z) s8[0] += p[(int64_t)z * zs];
  const double s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  double *c1 = C + (int64_t)gr * ldc + gc, *c2 = C + (int64_t)gc * ldc + gr;
  const double v1 = alpha * s + (beta != 0.0 ? beta * *c1 : 0.0);
  const double v2 = alpha * s + (beta != 0.0 ? beta * *c2 : 0.0);
  *c1 = v1;
  if (gr != gc) *c2 = v2;
}

// ------------------------------------------------------------------------------------
// D <= 208 (the i-vector sizes; C2 is 200): ONE workgroup per row chunk computes the whole lower triangle.
// With 128-column super-tiles X is read twice (once by the diagonal, once by the off-diagonal launch) and at
// C2 the two launches ran at the speed of those reads (2.6-3.7 TB/s of 1 KB row segments), not of the MFMAs.
// Here a stage is 16 full rows (coalesced 1.6 KB each), read once; the 13 x 13 / 2 MFMA tiles are dealt to 8 waves
// as tile rows {w, nt-1-w} (nt + 1 tiles per wave, 15 LDS operand reads for 14 MFMAs per k-step); register
// prefetch two stages ahead; partials of the computed tiles only, summed in fixed order by syrk_tri_reduce_kernel.
// ------------------------------------------------------------------------------------
constexpr int TRI_NT = 13;                 // tile rows: D <= 208
constexpr int TRI_LD = TRI_NT * 16 + 16;   // LDS row stride in doubles

// Rows [0, K1) come from X with the weights kw (nullptr: 1), rows [K1, K) from X2 with the one weight w2 (round 3: the
// statistics pass folds - M^T M, the centroids' term of the offset scatter, into the same launch as X^T diag(w) X).
//
// ZN (round 5, MPlda_norm's cohort moments, /root/reference/src/pldamodule.cpp:220-250): the rows are the cohort's
// TRANSFORMED vectors x_i [D0 wide] and the product is taken of the augmented, shifted rows
//     [ ca_d x_id - p_d (d < D0) | r_i - p_D0 | 1 ],      r_i = -1/2 (L + sum_d w_d x_id^2),      D = D0 + 2 <= 208
// formed on the way into LDS: the scale / shift ride on the store, the row sum of w x^2 is taken of the registers the
// rows were fetched into (one wave reduction per row and wave, four partials per row in LDS) and patched into column D0
// behind the stage's barrier (one more barrier per stage).  One read of the rows gives the second moments, the column sums
// (the constant column) and the count -- rounds 2-4 made five passes (rows, column sums, centring, SYRK reading twice).
// zc: [ca (D0) | w (D0) | g (D0) | L]  (znorm_coef_kernel);  zs: the pilot shift p [D0 + 1].
// Tried in round 6 and dropped: a BALANCED tile plan -- every tile row cut into parts of at most six tiles, the parts dealt to the
// eight waves largest first (11-13 tiles per wave, 22-24 per SIMD at nt = 13, where the assignment below gives 28 : 28 : 21 : 14),
// each part with its own column fragments read from LDS.  Same results, NOT faster: K2 0.159 ms at 100k x 200 (0.156), 0.39 of the
// peak at 1M rows (0.39), norm() 0.88 ms (0.86), the EM's rank-k sums 2 % slower -- 242-256 registers, 33-48 scalar spills and one
// more fragment read per six MFMAs cost what the balance gains; the kernel is not bound by its busiest SIMD's MFMAs.
template <bool ZN>
__global__ __launch_bounds__(512) void syrk_tri_kernel(int D, int64_t K, int64_t kchunk, const double *__restrict__ X,
                                                       int64_t ldx, const double *__restrict__ kw, int64_t K1,
                                                       const double *__restrict__ X2, int64_t ldx2, double w2,
                                                       double *__restrict__ part, const double *__restrict__ zc,
                                                       const double *__restrict__ zs, const SyrkChunk *__restrict__ chunks = nullptr) {
  __shared__ double Xs[2][GK * TRI_LD];
  __shared__ double Ws[2][GK];
  __shared__ double Rs[2][GK * 4];
  const int D0 = ZN ? D - 2 : D;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nt = (D + 15) / 16, ntri = nt * (nt + 1) / 2;
  int64_t kbeg = (int64_t)blockIdx.x * kchunk;
  int64_t kend = min(K, kbeg + kchunk);
  // chunks != nullptr: workgroup x multiplies the rows of chunk x of a table (its own row pointer and count, unit weights; the
  // EM's rank-k sums, em_rank_sums_mstep_f64: the chunks' coefficients are applied by the reduction)
  if (chunks) { X = chunks[blockIdx.x].rows; kbeg = 0; kend = chunks[blockIdx.x].nrows; K1 = kend; kw = nullptr; }
  // tile rows of this wave: lo = wave (wave + 1 tiles), hi = nt - 1 - wave (nt - wave tiles) when it is a different row
  const int lo = wave, hi = nt - 1 - wave;
  const int cnt_lo = lo <= hi ? lo + 1 : 0, cnt_hi = hi > lo ? hi + 1 : 0;
  f64x4 acc_lo[8], acc_hi[TRI_NT];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc_lo[c] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int c = 0; c < TRI_NT; ++c) acc_hi[c] = f64x4{0.0, 0.0, 0.0, 0.0};
  const int fi = lane & 15, fk = lane >> 4;
  const int tc = t & 255, tk = t >> 8;      // column, row offset (rows tk + 2 pass)
  const int gc = min(tc, D0 - 1);
  double z_cs = 1.0, z_sh = 0.0, z_w = 0.0, z_L = 0.0, z_shr = 0.0;
  if (ZN) {
    if (tc < D0) { z_cs = zc[tc]; z_sh = zs[tc]; z_w = zc[D0 + tc]; }
    z_L = zc[3 * D0]; z_shr = zs[D0];
  }
  auto fetch = [&](double (&r)[8], int64_t k0) {
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int64_t gk = min(k0 + tk + pass * 2, kend - 1);
      r[pass] = gk < K1 ? X[gk * ldx + gc] : X2[(gk - K1) * ldx2 + gc];
    }
  };
  auto fetch_w = [&](int64_t k0) {
    double w = 0.0;
    if (t < GK && k0 + t < kend) w = k0 + t < K1 ? (kw ? kw[k0 + t] : 1.0) : w2;
    return w;
  };
  auto store = [&](double *lds, const double (&r)[8], double *rs) {
    if (tc < TRI_LD) {
      const bool ok = tc < D;
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        double v = ok ? r[pass] : 0.0;
        // (rounded product, then the subtraction -- not one fma: a one-row cohort must centre to exactly 0, as the pilot computes p)
        if (ZN) v = tc < D0 ? __dsub_rn(__dmul_rn(r[pass], z_cs), z_sh) : (tc == D0 + 1 ? 1.0 : 0.0);
        lds[(tk + pass * 2) * TRI_LD + tc] = v;
      }
    }
    if (ZN) {
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        const double sq = wave_sum_f64(z_w * r[pass] * r[pass]);      // (z_w = 0 beyond column D0)
        if (lane == 0) rs[(tk + pass * 2) * 4 + (wave & 3)] = sq;
      }
    }
  };
  // ZN: column D0 of the stage just stored, behind the barrier that made its four partials per row visible
  auto patch = [&](double *lds, const double *rs) {
    if (t < GK) lds[t * TRI_LD + D0] = -0.5 * (((rs[t * 4] + rs[t * 4 + 1]) + (rs[t * 4 + 2] + rs[t * 4 + 3])) + z_L) - z_shr;
  };
  double r0[8], r1[8], w0 = 0.0, w1 = 0.0;
  if (kbeg < kend) {
    fetch(r0, kbeg);
    w0 = fetch_w(kbeg);
    if (kbeg + GK < kend) {
      fetch(r1, kbeg + GK);
      w1 = fetch_w(kbeg + GK);
    }
    store(Xs[0], r0, Rs[0]);
    if (t < GK) Ws[0][t] = w0;
  }
  __syncthreads();
  if (ZN) {
    if (kbeg < kend) patch(Xs[0], Rs[0]);
    __syncthreads();
  }
  // (round 5) The two waves of a SIMD -- w and w + 4 -- take their staging at OPPOSITE ends of a stage, as the transform
  // kernel's do: waves 4..7 ("early") write the next stage's rows into the other buffer at the START of a stage (fetched one
  // stage earlier, like everyone's), then fetch and multiply; waves 0..3 fetch, multiply and write at the END.  In step, both
  // waves of a SIMD staged at the same time and multiplied at the same time: nothing overlapped (0.43 of the peak in steady state).
  const bool early = wave >= 4;
  auto stage = [&](int64_t k0, int cur, double (&rn)[8], double &wn, const double (&rs)[8], const double &ws) {
    if (early && k0 + GK < kend) store(Xs[cur ^ 1], rs, Rs[cur ^ 1]);
    if (k0 + 2 * GK < kend) {
      fetch(rn, k0 + 2 * GK);
      wn = fetch_w(k0 + 2 * GK);
    }
    const double *Xr = Xs[cur], *Wr = Ws[cur];
#pragma unroll
    for (int kk = 0; kk < GK / 4; ++kk) {
      const int kb = (kk * 4 + fk) * TRI_LD + fi;
      const double wk = Wr[kk * 4 + fk];
      const double a_lo = Xr[kb + lo * 16] * wk, a_hi = Xr[kb + max(hi, 0) * 16] * wk;
      // column operands in two batches (all 13 at once pushed the kernel over 256 VGPRs)
#pragma unroll
      for (int c0 = 0; c0 < TRI_NT; c0 += 7) {
        double b[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) b[c] = c0 + c < TRI_NT ? Xr[kb + (c0 + c) * 16] : 0.0;
#pragma unroll
        for (int c = 0; c < 7; ++c) {
          const int cc = c0 + c;
          if (cc < 8 && cc < cnt_lo) acc_lo[cc < 8 ? cc : 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_lo, b[c], acc_lo[cc < 8 ? cc : 0], 0, 0, 0);
          if (cc < TRI_NT && cc < cnt_hi) acc_hi[cc < TRI_NT ? cc : 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_hi, b[c], acc_hi[cc < TRI_NT ? cc : 0], 0, 0, 0);
        }
      }
    }
    if (k0 + GK < kend) {
      if (!early) store(Xs[cur ^ 1], rs, Rs[cur ^ 1]);
      if (t < GK) Ws[cur ^ 1][t] = ws;
    }
    lds_barrier();
    if (ZN) {
      if (k0 + GK < kend) patch(Xs[cur ^ 1], Rs[cur ^ 1]);
      lds_barrier();
    }
  };
  for (int64_t k0 = kbeg; k0 < kend; k0 += 2 * GK) {
    stage(k0, 0, r0, w0, r1, w1);
    if (k0 + GK < kend) stage(k0 + GK, 1, r1, w1, r0, w0);
  }
  // partial: tile (r, c) at slot r (r + 1) / 2 + c, 256 doubles in the MFMA C layout
  double *slab = part + (int64_t)blockIdx.x * ntri * 256;
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (c < cnt_lo) {
      double *d = slab + (lo * (lo + 1) / 2 + c) * 256 + lane;
#pragma unroll
      for (int r = 0; r < 4; ++r) d[r * 64] = acc_lo[c][r];
    }
#pragma unroll
  for (int c = 0; c < TRI_NT; ++c)
    if (c < cnt_hi) {
      double *d = slab + (hi * (hi + 1) / 2 + c) * 256 + lane;
#pragma unroll
      for (int r = 0; r < 4; ++r) d[r * 64] = acc_hi[c][r];
    }
}

// ------------------------------------------------------------------------------------
// Tried in round 5 and REMOVED in round 6 (it was PLDA_GEMM64_VARIANT=7): the same product on FOUR waves, one per SIMD, a
// lone wave holding 23 of the 91 tiles (184 accumulator registers), all 13 fragments of a k-step read one k-step ahead,
// 23 MFMAs per 13 LDS reads in one branch-free run.  Steady state 0.35 of the fp64 MFMA peak against 0.43 for the
// eight-wave kernel above, 0.166 against 0.151 ms at 100k rows (D = 200).  Why: a lone wave's own vector, scalar, LDS and
// memory instructions do NOT run in the shadow of its own fp64 MFMAs (K4's finding, transform.hip), so a stage's 16 loads,
// 16 LDS writes, 52 fragment reads, 52 weight multiplies and the address arithmetic ADD to its 92 MFMAs; with two waves per
// SIMD one wave's staging overlaps the other's MFMAs.
// ------------------------------------------------------------------------------------

// Round 5: 1024 threads per tile -- element e = t & 255 of the tile, quarter q = t >> 8 of the splits -- so that a tile's
// 256 partials are four chains of <= 64 with sixteen loads in flight each instead of one chain of 256 with eight (the
// reduction was 25-30 us of dependent L2 round trips at C2, a fifth of K2); the four quarter sums meet in LDS and are
// added in a fixed order: deterministic, and exactly symmetric (both mirror elements get the same sum).
__global__ __launch_bounds__(1024) void syrk_tri_reduce_kernel(const double *__restrict__ part, int splits, int D,
                                                               double alpha, double beta, double *__restrict__ C,
                                                               int64_t ldc) {
  __shared__ double qs[4][256];
  const int nt = (D + 15) / 16, ntri = nt * (nt + 1) / 2;
  int tr = 0, tcol = (int)blockIdx.x;
  while (tcol > tr) { tcol -= tr + 1; ++tr; }
  const int e = threadIdx.x & 255, q = threadIdx.x >> 8;
  const int lane = e & 63, reg = e >> 6;
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  const int gr = tr * 16 + (lane >> 4) + 4 * reg, gcol = tcol * 16 + (lane & 15);
  const bool live = gr < D && gcol < D && gcol <= gr;
  const int per = (splits + 3) >> 2, z0 = q * per, z1 = min(splits, z0 + per);
  const double *p = part + (int64_t)blockIdx.x * 256 + e;
  const int64_t zs = (int64_t)ntri * 256;
  double s16[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) s16[u] = 0.0;
  if (live) {
    int z = z0;
    for (; z + 16 <= z1; z += 16) {
#pragma unroll
      for (int u = 0; u < 16; ++u) s16[u] += p[(int64_t)(z + u) * zs];
    }
    for (int u = 0; z < z1; ++z, ++u) s16[u] += p[(int64_t)z * zs];
  }
  qs[q][e] = (((s16[0] + s16[1]) + (s16[2] + s16[3])) + ((s16[4] + s16[5]) + (s16[6] + s16[7]))) +
             (((s16[8] + s16[9]) + (s16[10] + s16[11])) + ((s16[12] + s16[13]) + (s16[14] + s16[15])));
  __syncthreads();
  if (q != 0 || !live) return;
  const double sum = (qs[0][e] + qs[1][e]) + (qs[2][e] + qs[3][e]);
  double *c1 = C + (int64_t)gr * ldc + gcol, *c2 = C + (int64_t)gcol * ldc + gr;
  const double v1 = alpha * sum + (beta != 0.0 ? beta * *c1 : 0.0);
  const double v2 = alpha * sum + (beta != 0.0 ? beta * *c2 : 0.0);
  *c1 = v1;
  if (gr != gcol) *c2 = v2;
}

// The reduction above for the EM's two rank-k sums, with the M-step as its epilogue (em_rows_mstep_kernel's formula).  The
// slabs are products of row CHUNKS (chunk s: coefficients c1_s, c2_s):   P_y = sum_s c_y[s] slab_s,  and
//   blockIdx.y == 0:  W    = (S + sumK Bin + P_1) / cntW          blockIdx.y == 1:  Bout = (cw Bin + P_2) / cntB
// -- one workgroup per tile AND sum (182 at D = 200: the single-sum-pair form kept 91 of 256 CUs busy for 23 us on 48 MB of
// slabs); Bout is another buffer than Bin because the W half reads Bin while the B half writes.
__global__ __launch_bounds__(1024) void em_rank_reduce_mstep_kernel(const double *__restrict__ part, const SyrkChunk *__restrict__ chunks,
                                                                    int splits, int D, const double *__restrict__ S, double sumK,
                                                                    double cw, double cntW, double cntB, double *__restrict__ W,
                                                                    const double *__restrict__ Bin, double *__restrict__ Bout) {
  __shared__ double qs[4][256];
  __shared__ double cs[1024];
  const int nt = (D + 15) / 16, ntri = nt * (nt + 1) / 2;
  const int y = blockIdx.y;
  int tr = 0, tcol = (int)blockIdx.x;
  while (tcol > tr) { tcol -= tr + 1; ++tr; }
  const int e = threadIdx.x & 255, q = threadIdx.x >> 8;
  const int lane = e & 63, reg = e >> 6;
  const int gr = tr * 16 + (lane >> 4) + 4 * reg, gcol = tcol * 16 + (lane & 15);
  const bool live = gr < D && gcol < D && gcol <= gr;
  const int per = (splits + 3) >> 2, z0 = q * per, z1 = min(splits, z0 + per);
  const int64_t zs = (int64_t)ntri * 256;
  const double *p = part + (int64_t)blockIdx.x * 256 + e;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int zb = 0; zb < splits; zb += 1024) {          // the coefficients of this sum, 1024 chunks at a time, through LDS
    __syncthreads();
    if (zb + (int)threadIdx.x < splits) cs[threadIdx.x] = y ? chunks[zb + threadIdx.x].c2 : chunks[zb + threadIdx.x].c1;
    __syncthreads();
    if (live) {
      const int za = max(z0, zb), zc = min(z1, zb + 1024);
      int z = za;
      for (; z + 16 <= zc; z += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = p[(int64_t)(z + u) * zs];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] = fma(cs[z + u - zb], v[u], acc[u & 3]);
      }
      for (; z < zc; ++z) acc[z & 3] = fma(cs[z - zb], p[(int64_t)z * zs], acc[z & 3]);
    }
  }
  qs[q][e] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (q != 0 || !live) return;
  const double pv = (qs[0][e] + qs[1][e]) + (qs[2][e] + qs[3][e]);
  const size_t ij = (size_t)gr * D + gcol, ji = (size_t)gcol * D + gr;
  const double bij = Bin[ij], bji = Bin[ji];
  if (y == 0) {
    const double wij = S[ij] + fma(sumK, bij, pv), wji = S[ji] + fma(sumK, bji, pv);
    const double w = 0.5 * (wij / cntW + wji / cntW);
    W[ij] = w; W[ji] = w;
  } else {
    const double vij = fma(cw, bij, pv), vji = fma(cw, bji, pv);
    const double b = 0.5 * (vij / cntB + vji / cntB);
    Bout[ij] = b; Bout[ji] = b;
  }
}

// The grouped EM's two rank-k sums and its M-step (fit.hip, row form), D <= 208:
//   P1 = sum_g (-K_g n_g) X_g^T X_g + Z^T Z,     P2 = sum_g (-K_g) X_g^T X_g + Wn^T Wn
// Every product of rows is taken ONCE: the rows are cut into chunks that never cross a group of X (`chunks`, built once per fit by
// em_rank_chunks), one workgroup of the triangle kernel per chunk, and the reduction weights chunk s by (c1_s, c2_s) -- X's
// chunks enter both sums, Z's only the first, Wn's only the second.  (Round 6a ran the stacked X rows through the kernel twice, with
// row weights, 555 k MFMAs instead of 392 k at C2's skewed labels.)  *used = false: D > 208, the caller takes two syrk_pair_f64
// and its own M-step.
int em_rank_sums_mstep_f64(plda_handle *h, int D, const SyrkChunk *chunks, int nchunks, const double *S, double sumK, double cw,
                           double cntW, double cntB, double *W, const double *Bin, double *Bout, bool *used) {
  *used = false;
  if (D > TRI_NT * 16 || !(h->gemm64_variant == 0 || h->gemm64_variant == 6) || nchunks < 1) return PLDA_OK;
  const int nt = (int)ceil_div(D, 16), ntri = nt * (nt + 1) / 2;
  PLDA_HIP(h, h->w[15].reserve((size_t)nchunks * ntri * 256 * 8));
  double *part = h->w[15].as<double>();
  syrk_tri_kernel<false><<<(unsigned)nchunks, 512, 0, h->stream>>>(D, 0, 0, nullptr, D, nullptr, 0, nullptr, D, 1.0, part, nullptr,
                                                                   nullptr, chunks);
  em_rank_reduce_mstep_kernel<<<dim3((unsigned)ntri, 2), 1024, 0, h->stream>>>(part, chunks, nchunks, D, S, sumK, cw, cntW, cntB, W, Bin, Bout);
  PLDA_LAUNCH_CHECK(h);
  *used = true;
  return PLDA_OK;
}

// The chunk table of em_rank_sums_mstep_f64 (host side; `out` has room for em_rank_chunk_bound entries): rows per chunk so that
// the launch is about one workgroup per CU, the groups of X cut into equal parts.
int em_rank_chunk_bound(int G, int D, int64_t K, int cus) {
  const int64_t R = (int64_t)G * D + 2 * K;
  const int rc = (int)std::max<int64_t>(32, round_up(ceil_div(R, cus), 16));
  return (int)((int64_t)G * ceil_div(D, rc) + 2 * ceil_div(K, rc));
}
int em_rank_chunks(int G, int D, int64_t K, int cus, const double *X, const double *gn, const double *gk, const double *Z,
                   const double *Wn, SyrkChunk *out) {
  const int64_t R = (int64_t)G * D + 2 * K;
  const int rc = (int)std::max<int64_t>(32, round_up(ceil_div(R, cus), 16));
  int n = 0;
  const int nx = (int)ceil_div(D, rc);
  for (int g = 0; g < G; ++g)
    for (int c = 0; c < nx; ++c) {
      const int r0 = (int)((int64_t)D * c / nx), r1 = (int)((int64_t)D * (c + 1) / nx);
      out[n++] = SyrkChunk{X + ((int64_t)g * D + r0) * D, r1 - r0, 0, -gk[g] * gn[g], -gk[g]};
    }
  for (int64_t r = 0; r < K; r += rc) out[n++] = SyrkChunk{Z + r * D, (int)std::min<int64_t>(rc, K - r), 0, 1.0, 0.0};
  for (int64_t r = 0; r < K; r += rc) out[n++] = SyrkChunk{Wn + r * D, (int)std::min<int64_t>(rc, K - r), 0, 0.0, 1.0};
  return n;
}

#include "syrk_blk.inc"

int syrk_f64(plda_handle *h, int D, int64_t K, double alpha, const double *X, int64_t ldx, const double *kw,
             double beta, double *C, int64_t ldc) {
  if (h->gemm64_variant == 0) {          // 208 < D <= 512: one read of full rows (syrk_blk.inc)
    bool used = false;
    PLDA_TRY(syrk_blk_pair(h, D, K, X, ldx, kw, 0, nullptr, 0, 0.0, alpha, beta, C, ldc, &used));
    if (used) return PLDA_OK;
  }
  if (D <= TRI_NT * 16 && h->gemm64_variant != 3) {   // PLDA_GEMM64_VARIANT=3: super-tile path always (A/B arm)
    const int nt = (int)ceil_div(D, 16), ntri = nt * (nt + 1) / 2;
    int splits = (int)std::max<int64_t>(1, std::min<int64_t>(256, ceil_div(K, 128)));
    const int64_t kchunk = round_up(ceil_div(K, splits), GK);
    splits = (int)ceil_div(K, kchunk);
    PLDA_HIP(h, h->w[15].reserve((size_t)splits * ntri * 256 * 8));
    double *part = h->w[15].as<double>();
    syrk_tri_kernel<false><<<(unsigned)splits, 512, 0, h->stream>>>(D, K, kchunk, X, ldx, kw, K, nullptr, 0, 0.0, part, nullptr, nullptr);
    syrk_tri_reduce_kernel<<<(unsigned)ntri, 1024, 0, h->stream>>>(part, splits, D, alpha, beta, C, ldc);
    PLDA_LAUNCH_CHECK(h);
    return PLDA_OK;
  }
  const int nT = (int)ceil_div(D, 128), nP = nT * (nT + 1) / 2, nO = nP - nT;
  // diagonal and strictly-lower super-tiles are two launches (two tile mappings); each gets its own split of
  // the rows: 512 workgroups (two per CU, all resident at once), chunks of at least 128 rows (a partial costs
  // 2 KiB per computed tile)
  auto plan = [&](int pairs, int &splits, int64_t &kchunk) {
    static const int target = std::getenv("PLDA_SYRK_WGS") ? std::atoi(std::getenv("PLDA_SYRK_WGS")) : 512;
    splits = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(target, pairs), ceil_div(K, 128)));
    kchunk = round_up(ceil_div(K, splits), GK);
    splits = (int)ceil_div(K, kchunk);
  };
  int sd = 1, so = 1;
  int64_t kd = K, ko = K;
  plan(nT, sd, kd);
  if (nO) plan(nO, so, ko);
  PLDA_HIP(h, h->w[15].reserve((size_t)std::max(sd, so) * nP * 64 * 256 * 8));
  double *part = h->w[15].as<double>();
  syrk_lower_kernel<true><<<(unsigned)(round_up(sd, 8) * nT), 256, 0, h->stream>>>(D, K, kd, X, ldx, kw, part, nP, nT, sd);
  if (nO) syrk_lower_kernel<false><<<(unsigned)(round_up(so, 8) * nO), 256, 0, h->stream>>>(D, K, ko, X, ldx, kw, part, nP, nO, so);
  PLDA_LAUNCH_CHECK(h);
  syrk_reduce_kernel<<<(unsigned)(nP * 64), 256, 0, h->stream>>>(part, sd, so, nP, D, alpha, beta, C, ldc);
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

// Cohort moments of MPlda_norm in one read of the transformed rows (syrk_tri_kernel<true>): C [(D0 + 2)^2] = sum over the K
// rows of the outer product of [ca x - p | r - p_D0 | 1].  *used = false when D0 + 2 > 208 (the caller keeps its own path).
int syrk_znorm_f64(plda_handle *h, int D0, int64_t K, const double *X, const double *zc, const double *zs, double *C, bool *used) {
  const int D = D0 + 2;
  *used = false;
  if (D > TRI_NT * 16) return PLDA_OK;
  const int nt = (int)ceil_div(D, 16), ntri = nt * (nt + 1) / 2;
  int splits = (int)std::max<int64_t>(1, std::min<int64_t>(256, ceil_div(K, 128)));
  const int64_t kchunk = round_up(ceil_div(K, splits), GK);
  splits = (int)ceil_div(K, kchunk);
  PLDA_HIP(h, h->w[15].reserve((size_t)splits * ntri * 256 * 8));
  double *part = h->w[15].as<double>();
  syrk_tri_kernel<true><<<(unsigned)splits, 512, 0, h->stream>>>(D, K, kchunk, X, D0, nullptr, K, nullptr, 0, 0.0, part, zc, zs);
  syrk_tri_reduce_kernel<<<(unsigned)ntri, 1024, 0, h->stream>>>(part, splits, D, 1.0, 0.0, C, D);
  PLDA_LAUNCH_CHECK(h);
  *used = true;
  return PLDA_OK;
}

// C = X^T diag(kw) X + w2 X2^T X2 in one pass where the single-launch kernel applies (D <= 208), else as two products
int syrk_pair_f64(plda_handle *h, int D, int64_t K1, const double *X, int64_t ldx, const double *kw, int64_t K2,
                  const double *X2, int64_t ldx2, double w2, double *C, int64_t ldc) {
  if (D <= TRI_NT * 16 && (h->gemm64_variant == 0 || h->gemm64_variant == 6)) {
    const int nt = (int)ceil_div(D, 16), ntri = nt * (nt + 1) / 2;
    const int64_t K = K1 + K2;
    int splits = (int)std::max<int64_t>(1, std::min<int64_t>(256, ceil_div(K, 128)));
    const int64_t kchunk = round_up(ceil_div(K, splits), GK);
    splits = (int)ceil_div(K, kchunk);
    PLDA_HIP(h, h->w[15].reserve((size_t)splits * ntri * 256 * 8));
    double *part = h->w[15].as<double>();
    syrk_tri_kernel<false><<<(unsigned)splits, 512, 0, h->stream>>>(D, K, kchunk, X, ldx, kw, K1, X2, ldx2, w2, part, nullptr, nullptr);
    syrk_tri_reduce_kernel<<<(unsigned)ntri, 1024, 0, h->stream>>>(part, splits, D, 1.0, 0.0, C, ldc);
    PLDA_LAUNCH_CHECK(h);
    return PLDA_OK;
  }
  if (h->gemm64_variant == 0) {          // 208 < D <= 512: both products from one read of full rows (syrk_blk.inc)
    bool used = false;
    PLDA_TRY(syrk_blk_pair(h, D, K1, X, ldx, kw, K2, X2, ldx2, w2, 1.0, 0.0, C, ldc, &used));
    if (used) return PLDA_OK;
  }
  PLDA_TRY(gemm_f64(h, D, D, K1, 1.0, X, 1, ldx, X, ldx, 1, kw, 0.0, C, ldc));
  return gemm_f64(h, D, D, K2, w2, X2, 1, ldx2, X2, ldx2, 1, nullptr, 1.0, C, ldc);
}

// Small products (the D x D x D GEMMs of the EM, K <= 256): the 64 x 64 kernel above would put 16
// workgroups on the chip and walk K in 13 dependent load -> LDS -> MFMA rounds.  Here a workgroup owns a
// 32 x 32 tile, pulls the WHOLE K extent of both operand panels into LDS with every load in flight at
// once (one memory latency), then runs the MFMAs back to back.
// Panel kernel for the small products of the EM, the whitening and GetOutput (M, N <= 1024, few tiles): a workgroup
// owns a 32 x 32 tile of C and stages 32 rows of each operand over a K chunk of 64 NB values in LDS, every load of
// the chunk in flight at once (8 NB per operand and thread, no integer division in the index math: with batches of
// 8 + 8 loads and idx / K the 200^3 products of the EM took 14 us, four memory round trips and ~100 divisions per
// thread).  K > 256 runs in chunks of 256 with the accumulators kept.
template <bool KC, int NB>
__device__ __forceinline__ void panel_fetch(double (&v)[8 * NB], const double *__restrict__ P, int64_t s_row, int64_t s_k,
                                            int r0, int R, int kn, int t) {
#pragma unroll
  for (int u = 0; u < 8 * NB; ++u) {
    int r, k;
    if (KC) { r = (t >> 4) + 16 * (u & 1); k = (t & 15) + 16 * (u >> 1); }   // 16 lanes along k: 128-byte segments
    else { r = t & 31; k = (t >> 5) + 8 * u; }                              // 32 lanes along the rows
    const int gr = r0 + r;
    v[u] = (k < kn && gr < R) ? P[(int64_t)gr * s_row + (int64_t)k * s_k] : 0.0;
  }
}
template <bool KC, int NB>
__device__ __forceinline__ void panel_stage(const double (&v)[8 * NB], double *__restrict__ S, int t) {
  constexpr int ld = 64 * NB + 1;
#pragma unroll
  for (int u = 0; u < 8 * NB; ++u) {
    int r, k;
    if (KC) { r = (t >> 4) + 16 * (u & 1); k = (t & 15) + 16 * (u >> 1); }
    else { r = t & 31; k = (t >> 5) + 8 * u; }
    S[r * ld + k] = v[u];
  }
}

template <bool AKC, bool BKC, int NB>
__global__ __launch_bounds__(256) void gemm_f64_panel_kernel(int M, int N, int K, double alpha,
                                                             const double *__restrict__ A, int64_t sam, int64_t sak,
                                                             const double *__restrict__ B, int64_t sbk, int64_t sbn,
                                                             double beta, double *__restrict__ C, int64_t ldc,
                                                             int64_t strideA, int64_t strideB, int64_t strideC) {
  extern __shared__ __attribute__((aligned(16))) double panel[];   // As[32][ld], Bs[32][ld]
  constexpr int kc = 64 * NB, ld = kc + 1;
  double *As = panel, *Bs = panel + 32 * ld;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  A += (int64_t)blockIdx.z * strideA;
  B += (int64_t)blockIdx.z * strideB;
  C += (int64_t)blockIdx.z * strideC;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int wm = wave >> 1, wn = wave & 1, fi = lane & 15, fk = lane >> 4;
  const double *ap = As + (wm * 16 + fi) * ld + fk, *bp = Bs + (wn * 16 + fi) * ld + fk;
  f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < K; k0 += kc) {
    const int kn = min(kc, K - k0), Kp = (kn + 3) & ~3;
    double va[8 * NB], vb[8 * NB];
    panel_fetch<AKC, NB>(va, A + (int64_t)k0 * sak, sam, sak, m0, M, kn, t);
    panel_fetch<BKC, NB>(vb, B + (int64_t)k0 * sbk, sbn, sbk, n0, N, kn, t);
    if (k0) __syncthreads();        // the previous chunk has been consumed
    panel_stage<AKC, NB>(va, As, t);
    panel_stage<BKC, NB>(vb, Bs, t);
    __syncthreads();
    int kk = 0;
    for (; kk + 8 <= Kp; kk += 8) {
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[kk], bp[kk], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[kk + 4], bp[kk + 4], acc1, 0, 0, 0);
    }
    if (kk < Kp) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[kk], bp[kk], acc0, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + wm * 16 + fk + 4 * r, col = n0 + wn * 16 + fi;
    if (row < M && col < N) {
      double *c = C + (int64_t)row * ldc + col;
      *c = alpha * (acc0[r] + acc1[r]) + (beta != 0.0 ? beta * *c : 0.0);
    }
  }
}

// The D x D x D products of the EM and of GetOutput (M, N, K <= 256): what they cost is not their 16 MFLOP but a
// launch and a dependent chain -- the panel kernel above stages 2 x 51 KB per workgroup through LDS and then runs 50
// MFMAs on every wave (1.3 us on their own), 7.5 us in all.  Here a workgroup owns ONE 16 x 16 tile and its four waves a
// quarter of K each: the operand fragments go from global memory straight to the MFMA operand registers (all loads of a
// wave in flight at once, no LDS stage, no barrier before the arithmetic), 13 MFMAs per wave at K = 200, and the four
// partial tiles meet in 8 KB of LDS.  169 workgroups at D = 200.
struct Tile16Operands {
  double alpha, beta;
  const double *A, *B;
  double *C;
  int64_t sam, sak, sbk, sbn, ldc, strideA, strideB, strideC;
};
// blockIdx.z = set * batch0 + product: up to three independent sets of batched products of the same shape in one launch
// (gemm_f64_multi: the EM's independent D^3 products)
__global__ __launch_bounds__(256) void gemm_f64_tile16_kernel(int M, int N, int K, Tile16Operands o0, Tile16Operands o1,
                                                              int batch0, Tile16Operands o2 = Tile16Operands{}) {
  __shared__ double part[4][256];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fi = lane & 15, fk = lane >> 4;
  const int set = (int)blockIdx.z / batch0;
  const Tile16Operands &o = set == 0 ? o0 : set == 1 ? o1 : o2;
  const int64_t z = (int)blockIdx.z - set * batch0;
  const double *__restrict__ A = o.A + z * o.strideA, *__restrict__ B = o.B + z * o.strideB;
  double *__restrict__ C = o.C + z * o.strideC;
  const int64_t sam = o.sam, sak = o.sak, sbk = o.sbk, sbn = o.sbn, ldc = o.ldc;
  const double alpha = o.alpha, beta = o.beta;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
  const int kq = ((K + 15) / 16) * 4;                 // k per wave, a multiple of the MFMA's 4
  const int k0 = wave * kq, k1 = min(K, k0 + kq);
  const double *ap = A + (int64_t)min(m0 + fi, M - 1) * sam, *bp = B + (int64_t)min(n0 + fi, N - 1) * sbn;
  double va[16], vb[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int k = k0 + 4 * s + fk;
    const bool ok = k < k1;
    va[s] = ok ? ap[(int64_t)k * sak] : 0.0;
    vb[s] = ok ? bp[(int64_t)k * sbk] : 0.0;
  }
  f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < 16; s += 2) {
    if (k0 + 4 * s < k1) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(va[s], vb[s], acc0, 0, 0, 0);
    if (k0 + 4 * s + 4 < k1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(va[s + 1], vb[s + 1], acc1, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) part[wave][(fk + 4 * r) * 16 + fi] = acc0[r] + acc1[r];
  __syncthreads();
  const int row = m0 + (t >> 4), col = n0 + (t & 15);
  if (row < M && col < N) {
    const double x = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    double *c = C + (int64_t)row * ldc + col;
    *c = alpha * x + (beta != 0.0 ? beta * *c : 0.0);
  }
}

// batch > 1: `batch` independent products with operand strides (a stride of 0 shares the operand);
// no split-K in that case -- the batch supplies the parallelism.
int gemm_f64_batched(plda_handle *h, int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t sam,
                     int64_t sak, int64_t strideA, const double *B, int64_t sbk, int64_t sbn, int64_t strideB,
                     const double *kw, double beta, double *C, int64_t ldc, int64_t strideC, int batch) {
  if (M <= 0 || N <= 0 || batch <= 0) return PLDA_OK;
  if (K <= 0) return fail(h, PLDA_E_INVAL, "gemm_f64: K <= 0");
  const bool akc = (sak == 1), bkc = (sbk == 1);
  if ((!akc && sam != 1) || (!bkc && sbn != 1))
    return fail(h, PLDA_E_INVAL, "gemm_f64: operands need one unit stride");
  // X^T diag(w) X with both operands the same rows: the symmetric kernel (PLDA_GEMM64_VARIANT=2 keeps the
  // general one, for A/B measurements)
  if (batch == 1 && A == B && M == N && !akc && !bkc && sak == sbk && sam == 1 && sbn == 1 && K >= 2048 && M >= 32 &&
      M <= 1024 && h->gemm64_variant != 2)
    return syrk_f64(h, (int)M, K, alpha, A, sak, kw, beta, C, ldc);
  const int64_t tiles = ceil_div(M, GB) * ceil_div(N, GB);
  // one 16 x 16 tile per workgroup, K split over its waves: the small square products (PLDA_GEMM64_VARIANT=5: the panel kernel)
  if (M <= 256 && N <= 256 && K <= 256 && !kw && h->gemm64_variant != 5 && h->gemm64_variant != 4) {
    if (batch > 65535) return fail(h, PLDA_E_INVAL, "gemm_f64: batch %d too large", batch);
    const dim3 tgrid((unsigned)ceil_div(N, 16), (unsigned)ceil_div(M, 16), (unsigned)batch);
    const Tile16Operands o{alpha, beta, A, B, C, sam, sak, sbk, sbn, ldc, strideA, strideB, strideC};
    gemm_f64_tile16_kernel<<<tgrid, 256, 0, h->stream>>>((int)M, (int)N, (int)K, o, o, batch);
    PLDA_LAUNCH_CHECK(h);
    return PLDA_OK;
  }
  // the panel kernel: every K <= 256, and deeper products whose 64 x 64 tiles would leave most of the chip idle
  if (M <= 1024 && N <= 1024 && !kw && (K <= 256 || (K <= 2048 && tiles * batch < 256)) && h->gemm64_variant != 4) {
    const int nb = (int)std::min<int64_t>(4, ceil_div(K, 64));
    const size_t lds = (size_t)2 * 32 * (64 * nb + 1) * 8;
    const dim3 pgrid((unsigned)ceil_div(N, 32), (unsigned)ceil_div(M, 32), (unsigned)batch);
    if (batch > 65535) return fail(h, PLDA_E_INVAL, "gemm_f64: batch %d too large", batch);
#define PANEL_LAUNCH(AK, BK, NBB)                                                                             \
  do {                                                                                                        \
    bool &attr_done = h->panel_attr_set[((AK ? 2 : 0) + (BK ? 1 : 0)) * 4 + NBB - 1];                         \
    if (!attr_done) {                                                                                         \
      PLDA_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_f64_panel_kernel<AK, BK, NBB>),    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32 * (64 * NBB + 1) * 8)); \
      attr_done = true;                                                                                       \
    }                                                                                                         \
    gemm_f64_panel_kernel<AK, BK, NBB><<<pgrid, 256, lds, h->stream>>>((int)M, (int)N, (int)K, alpha, A, sam, \
                                                                       sak, B, sbk, sbn, beta, C, ldc,        \
                                                                       strideA, strideB, strideC);            \
  } while (0)
#define PANEL_NB(AK, BK)                                                                                      \
  do {                                                                                                        \
    if (nb == 1) PANEL_LAUNCH(AK, BK, 1);                                                                     \
    else if (nb == 2) PANEL_LAUNCH(AK, BK, 2);                                                                \
    else if (nb == 3) PANEL_LAUNCH(AK, BK, 3);                                                                \
    else PANEL_LAUNCH(AK, BK, 4);                                                                             \
  } while (0)
    if (akc && bkc) PANEL_NB(true, true);
    else if (akc && !bkc) PANEL_NB(true, false);
    else if (!akc && bkc) PANEL_NB(false, true);
    else PANEL_NB(false, false);
#undef PANEL_NB
#undef PANEL_LAUNCH
    PLDA_LAUNCH_CHECK(h);
    return PLDA_OK;
  }
  // 128 x 128 tiles when they still give every CU work (after split-K), 64 x 64 otherwise
  const int64_t tiles128 = ceil_div(M, 128) * ceil_div(N, 128);
  const bool big = h->gemm64_variant != 1 && M >= 128 && N >= 128 &&
                   tiles128 * (K >= 1024 ? std::min<int64_t>(ceil_div(K, 256), 64) : 1) * batch >= 192;
  const int TBsel = big ? 128 : 64;
  const int64_t tl = big ? tiles128 : tiles;
  int splits = 1;
  if (batch == 1 && tl < 512 && K >= 1024) {
    splits = (int)std::min<int64_t>(ceil_div(K, 256), std::max<int64_t>(1, 1024 / tl));
  }
  int64_t kchunk = round_up(ceil_div(K, splits), GK);
  splits = (int)ceil_div(K, kchunk);
  double *part = nullptr;
  if (splits > 1) {
    PLDA_HIP(h, h->w[15].reserve((size_t)splits * M * N * 8));
    part = h->w[15].as<double>();
  }
  if ((int64_t)splits * batch > 65535) return fail(h, PLDA_E_INVAL, "gemm_f64: batch %d too large", batch);
  const dim3 grid((unsigned)ceil_div(N, TBsel), (unsigned)ceil_div(M, TBsel), (unsigned)(splits * batch));
#define GEMM_LAUNCH(AK, BK)                                                                                   \
  do {                                                                                                        \
    if (big)                                                                                                  \
      gemm_f64_kernel<AK, BK, 128><<<grid, 256, 0, h->stream>>>(M, N, K, kchunk, alpha, A, sam, sak, B, sbk,  \
                                                                sbn, kw, beta, C, ldc, part, splits, strideA, \
                                                                strideB, strideC);                            \
    else                                                                                                      \
      gemm_f64_kernel<AK, BK, 64><<<grid, 256, 0, h->stream>>>(M, N, K, kchunk, alpha, A, sam, sak, B, sbk,   \
                                                               sbn, kw, beta, C, ldc, part, splits, strideA,  \
                                                               strideB, strideC);                             \
  } while (0)
  if (akc && bkc) GEMM_LAUNCH(true, true);
  else if (akc && !bkc) GEMM_LAUNCH(true, false);
  else if (!akc && bkc) GEMM_LAUNCH(false, true);
  else GEMM_LAUNCH(false, false);
#undef GEMM_LAUNCH
  PLDA_LAUNCH_CHECK(h);
  if (splits > 1) {
    splitk_reduce_kernel<<<(unsigned)ceil_div(M * N, 256), 256, 0, h->stream>>>(part, splits, M, N, alpha,
                                                                              beta, C, ldc);
    PLDA_LAUNCH_CHECK(h);
  }
  return PLDA_OK;
}

int gemm_f64(plda_handle *h, int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t sam,
             int64_t sak, const double *B, int64_t sbk, int64_t sbn, const double *kw, double beta,
             double *C, int64_t ldc) {
  return gemm_f64_batched(h, M, N, K, alpha, A, sam, sak, 0, B, sbk, sbn, 0, kw, beta, C, ldc, 0, 1);
}

// Up to three independent batched products of the same shape (alpha = 1, beta = 0), in ONE launch where the small-product
// kernel applies (M, N, K <= 256), else one after the other.
int gemm_f64_multi(plda_handle *h, int64_t M, int64_t N, int64_t K, const GemmSet *sets, int nsets, int batch) {
  if (nsets < 1 || nsets > 3) return fail(h, PLDA_E_INVAL, "gemm_f64_multi: 1..3 sets");
  if (M <= 256 && N <= 256 && K <= 256 && K > 0 && M > 0 && N > 0 && batch > 0 && nsets * batch <= 65535 &&
      h->gemm64_variant != 5 && h->gemm64_variant != 4) {
    Tile16Operands o[3];
    for (int i = 0; i < 3; ++i) {
      const GemmSet &g = sets[i < nsets ? i : 0];
      o[i] = Tile16Operands{1.0, 0.0, g.A, g.B, g.C, g.sam, g.sak, g.sbk, g.sbn, g.ldc, g.strideA, g.strideB, g.strideC};
    }
    const dim3 tgrid((unsigned)ceil_div(N, 16), (unsigned)ceil_div(M, 16), (unsigned)(nsets * batch));
    gemm_f64_tile16_kernel<<<tgrid, 256, 0, h->stream>>>((int)M, (int)N, (int)K, o[0], o[1], batch, o[2]);
    PLDA_LAUNCH_CHECK(h);
    return PLDA_OK;
  }
  for (int i = 0; i < nsets; ++i) {
    const GemmSet &g = sets[i];
    PLDA_TRY(gemm_f64_batched(h, M, N, K, 1.0, g.A, g.sam, g.sak, g.strideA, g.B, g.sbk, g.sbn, g.strideB, nullptr, 0.0, g.C, g.ldc,
                              g.strideC, batch));
  }
  return PLDA_OK;
}

// fp64 reciprocal / reciprocal square root: hardware estimate refined by Newton steps to full
// double precision (the inputs here are well inside the normal range)
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}

// ------------------------------------------------------------------------------------
// triangular inverse (TpMatrix::Invert): one WAVE per column j of X = L^{-1}.
// Forward substitution x_i = (delta_ij - sum_{k=j}^{i-1} L[i][k] x_k) / L[i][i]; lane l keeps
// x_k for k = l (mod 64) in registers, so the dot product is E coalesced loads of row i of
// L, E FMAs and one wave reduction per step; nothing goes through memory until the end.
// ------------------------------------------------------------------------------------
template <int E>
__global__ __launch_bounds__(64) void tri_invert_kernel(const double *__restrict__ L, double *__restrict__ X, int D,
                                                        int ldx, int64_t stride_x, int64_t stride_l) {
  const int j = blockIdx.x;
  const int lane = threadIdx.x;
  L += (int64_t)blockIdx.y * stride_l;
  X += (int64_t)blockIdx.y * stride_x;
  double x[E], r0[E], r1[E], r2[E], r3[E];
#pragma unroll
  for (int e = 0; e < E; ++e) x[e] = 0.0;
  // rows i .. i + 3 of L are in registers when step i starts and row i + 4 is requested before the step's
  // reduction: the chain of a step is E FMAs + a DPP wave sum + a Newton reciprocal instead of a memory round trip
  // + six ds_bpermute shuffles + an IEEE division: 130 -> 70 us at D = 200 (one row of look-ahead gives the same: the
  // chain of the wave sum and the reciprocal is what is left)
  auto load_row = [&](int i, double (&r)[E]) {
    const double *Li = L + (size_t)min(i, D - 1) * D;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int k = lane + e * 64;
      r[e] = k < D ? Li[k] : 0.0;
    }
  };
  auto step = [&](int i, double (&lc)[E]) {   // consumes row i (lc), then refills lc with row i + 4
    if (i < D) {
      double part = 0.0, dsel = 0.0;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int k = lane + e * 64;
        part += (k >= j && k < i) ? lc[e] * x[e] : 0.0;
        dsel = e == (i >> 6) ? lc[e] : dsel;
      }
      load_row(i + 4, lc);
      part = wave_sum_f64(part);
      const double di = readlane_f64(dsel, __builtin_amdgcn_readfirstlane(i & 63));
      const double xi = ((i == j ? 1.0 : 0.0) - part) * rcp_nr(di);
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (lane + e * 64 == i) x[e] = xi;
    }
  };
  load_row(j, r0);
  load_row(j + 1, r1);
  load_row(j + 2, r2);
  load_row(j + 3, r3);
  for (int i = j; i < D; i += 4) {
    step(i, r0);
    step(i + 1, r1);
    step(i + 2, r2);
    step(i + 3, r3);
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int k = lane + e * 64;
    if (k < D) X[(size_t)k * ldx + j] = x[e];   // rows k < j are the zeros x[] was initialised with
  }
}

// L: [batch] D x D factors, leading dimension D, batch stride stride_l
int tri_invert_ld(plda_handle *h, const double *L, int64_t stride_l, double *X, int D, int ldx, int64_t stride_x, int batch) {
  const int E = (int)ceil_div(D, 64);
#define TI(EE) tri_invert_kernel<EE><<<dim3(D, batch), 64, 0, h->stream>>>(L, X, D, ldx, stride_x, stride_l)
  if (E <= 1) TI(1);
  else if (E <= 2) TI(2);
  else if (E <= 4) TI(4);
  else if (E <= 8) TI(8);
  else if (E <= 16) TI(16);
  else return fail(h, PLDA_E_INVAL, "tri_invert: D=%d > 1024 unsupported", D);
#undef TI
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

// ------------------------------------------------------------------------------------
// SPD inverse by symmetric sweeps, matrix resident in registers (D <= 256), one matrix per workgroup.
// Computes (W + n_g B)^-1 for the grouped EM.  Sweep k of the (Goodnight) sweep operator:
//   d = A_kk;  A_ij -= A_ik A_kj / d (i, j != k);  A_ik = A_ki = A_ik / d;  A_kk = -1/d;
// after all D sweeps the matrix holds -A^-1.  For an SPD matrix every pivot is positive and no
// pivoting is needed.  The 1024 threads form a 32 x 32 grid; thread (ty, tx) owns the lower-triangle
// elements (i, j) = (32a + ty, 32b + tx), a >= b, so a sweep is NB(NB+1)/2 FMAs per thread on
// registers plus one LDS broadcast of column k (double-buffered: one barrier per sweep).
// (Two pivots per barrier -- both rank-one updates from the two columns as they stand before the pair -- is
// correct but slower, 3.35 against 2.62 ms for the ten EM iterations at D = 200: the second set of row / column
// factors does not fit next to the matrix in 128 registers, and the kernel is bound by fp64 issue with four waves
// per SIMD, not by the barrier.  Removed.)
// ------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(1024) void spd_inverse_sweep_kernel(const double *__restrict__ W,
                                                                 const double *__restrict__ B,
                                                                 const double *__restrict__ gn, int D, int ldin,
                                                                 int64_t stride_in, double *__restrict__ out,
                                                                 int ldout, int64_t stride_out, int *flag) {
  constexpr int NE = NB * (NB + 1) / 2;
  __shared__ double v[2][NB * 32];
  const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
  const double n = gn ? gn[blockIdx.x] : 0.0;
  W += (int64_t)blockIdx.x * stride_in;      // stride_in == 0: W (and B) shared by the batch, only n differs
  if (B) B += (int64_t)blockIdx.x * stride_in;
  out += (int64_t)blockIdx.x * stride_out;
  double r[NE];
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const int i = a * 32 + ty, j = b * 32 + tx;
      double x = 0.0;
      if (i < D && j <= i) {
        x = W[(size_t)i * ldin + j];
        if (B) x = fma(n, B[(size_t)i * ldin + j], x);
      }
      r[a * (a + 1) / 2 + b] = x;
    }
  bool bad = false;
  // the block index kb of the pivot is a compile-time constant inside the unrolled outer loop, so the
  // owners of column / row k are named registers: r[e(a, kb)], a >= kb, and r[e(kb, b)], b <= kb
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    for (int kl = 0; kl < 32; ++kl) {
      const int k = kb * 32 + kl;
      if (k >= D) break;
      double *vv = v[k & 1];
      if (tx == kl) {   // column k, rows >= k
#pragma unroll
        for (int a = kb; a < NB; ++a)
          if (a > kb || ty >= kl) vv[a * 32 + ty] = r[a * (a + 1) / 2 + kb];
      }
      if (ty == kl) {   // row k, columns < k
#pragma unroll
        for (int b = 0; b <= kb; ++b)
          if (b < kb || tx < kl) vv[b * 32 + tx] = r[kb * (kb + 1) / 2 + b];
      }
      __syncthreads();
      const double d = vv[k];
      if (!(d > 0.0)) bad = true;
      const double inv = rcp_nr(d);
      double ui[NB], vj[NB];
#pragma unroll
      for (int a = 0; a < NB; ++a) ui[a] = -vv[a * 32 + ty] * inv;
#pragma unroll
      for (int b = 0; b < NB; ++b) vj[b] = vv[b * 32 + tx];
#pragma unroll
      for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) r[a * (a + 1) / 2 + b] = fma(ui[a], vj[b], r[a * (a + 1) / 2 + b]);
      // owners overwrite column / row k with v / d, the pivot with -1/d
      if (tx == kl) {
#pragma unroll
        for (int a = kb; a < NB; ++a) {
          if (a > kb || ty > kl) r[a * (a + 1) / 2 + kb] = -ui[a];
          else if (ty == kl) r[a * (a + 1) / 2 + kb] = -inv;
        }
      }
      if (ty == kl) {
#pragma unroll
        for (int b = 0; b <= kb; ++b)
          if (b < kb || tx < kl) r[kb * (kb + 1) / 2 + b] = vj[b] * inv;
      }
    }
  }
  if (bad && t == 0) *flag = 1;
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const int i = a * 32 + ty, j = b * 32 + tx;
      if (i < D && j <= i) {
        const double x = -r[a * (a + 1) / 2 + b];
        out[(size_t)i * ldout + j] = x;
        out[(size_t)j * ldout + i] = x;
      }
    }
}

// The same sweep on FOUR waves (round 3): thread (ty, tx) of a 16 x 16 grid owns the lower-triangle elements
// (16a + ty, 16b + tx), a >= b -- 91 doubles at D = 200, in the 512 registers a wave has when it is alone on its SIMD.
// What a pivot costs is not its 20 000 FMAs (312 cycles of the CU's fp64 rate, whatever the number of waves) but the
// exchange around them: with 16 waves every thread re-read its 2 x 7 entries of column k from LDS one double at a time
// -- 224 wave-level LDS reads and a 16-wave barrier per pivot, 0.87 us.  Here the column is published in a permuted order
// (entry i at (i mod 16) * PAD + i / 16) so that a thread's entries are contiguous: 2 x 7 16-byte reads per thread, 56 per
// pivot, a 4-wave barrier: 175 -> ~55 us at D = 200.  One barrier per pivot (column buffers alternate).
template <int NB>
__global__ __launch_bounds__(256) void spd_inverse_sweep16_kernel(const double *__restrict__ W,
                                                                  const double *__restrict__ B,
                                                                  const double *__restrict__ gn, int D, int ldin,
                                                                  int64_t stride_in, double *__restrict__ out,
                                                                  int ldout, int64_t stride_out, int *flag) {
  constexpr int NE = NB * (NB + 1) / 2;
  constexpr int PAD = (NB + 1) & ~1;                     // entries per thread row of the published column, even
  __shared__ __attribute__((aligned(16))) double v[2][16 * PAD];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const double n = gn ? gn[blockIdx.x] : 0.0;
  W += (int64_t)blockIdx.x * stride_in;      // stride_in == 0: W (and B) shared by the batch, only n differs
  if (B) B += (int64_t)blockIdx.x * stride_in;
  out += (int64_t)blockIdx.x * stride_out;
  double r[NE];
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const int i = a * 16 + ty, j = b * 16 + tx;
      double x = (i == j) ? 1.0 : 0.0;        // (rows / columns beyond D: an identity block, swept like the rest)
      if (i < D && j <= i) {
        x = W[(size_t)i * ldin + j];
        if (B) x = fma(n, B[(size_t)i * ldin + j], x);
      }
      r[a * (a + 1) / 2 + b] = x;
    }
  bool bad = false;
  // the block index kb of the pivot is a compile-time constant inside the unrolled outer loop, so the
  // owners of column / row k are named registers: r[e(a, kb)], a >= kb, and r[e(kb, b)], b <= kb
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    for (int kl = 0; kl < 16; ++kl) {
      const int k = kb * 16 + kl;
      if (k >= D) break;
      double *vv = v[k & 1];
      if (tx == kl) {   // column k, rows >= k: entry (16a + ty) -> vv[ty * PAD + a]
#pragma unroll
        for (int a = kb; a < NB; ++a)
          if (a > kb || ty >= kl) vv[ty * PAD + a] = r[a * (a + 1) / 2 + kb];
      }
      if (ty == kl) {   // row k, columns < k
#pragma unroll
        for (int b = 0; b <= kb; ++b)
          if (b < kb || tx < kl) vv[tx * PAD + b] = r[kb * (kb + 1) / 2 + b];
      }
      __syncthreads();
      const double d = vv[kl * PAD + kb];
      if (!(d > 0.0)) bad = true;
      const double inv = rcp_nr(d);
      double ui[PAD], vj[PAD];
      const double2 *pu = reinterpret_cast<const double2 *>(vv + ty * PAD), *pv = reinterpret_cast<const double2 *>(vv + tx * PAD);
#pragma unroll
      for (int a = 0; a < PAD / 2; ++a) {
        const double2 u2 = pu[a], v2 = pv[a];
        ui[2 * a] = -u2.x * inv; ui[2 * a + 1] = -u2.y * inv;
        vj[2 * a] = v2.x; vj[2 * a + 1] = v2.y;
      }
#pragma unroll
      for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) r[a * (a + 1) / 2 + b] = fma(ui[a], vj[b], r[a * (a + 1) / 2 + b]);
      // owners overwrite column / row k with v / d, the pivot with -1/d
      if (tx == kl) {
#pragma unroll
        for (int a = kb; a < NB; ++a) {
          if (a > kb || ty > kl) r[a * (a + 1) / 2 + kb] = -ui[a];
          else if (ty == kl) r[a * (a + 1) / 2 + kb] = -inv;
        }
      }
      if (ty == kl) {
#pragma unroll
        for (int b = 0; b <= kb; ++b)
          if (b < kb || tx < kl) r[kb * (kb + 1) / 2 + b] = vj[b] * inv;
      }
    }
  }
  if (bad && t == 0) *flag = 1;
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const int i = a * 16 + ty, j = b * 16 + tx;
      if (i < D && j <= i) {
        const double x = -r[a * (a + 1) / 2 + b];
        out[(size_t)i * ldout + j] = x;
        out[(size_t)j * ldout + i] = x;
      }
    }
}

// spd_inverse_mfma_kernel<NT>: the sweep in blocks of 16 pivots on the fp64 matrix cores (its own file so that
// scripts/probe/sweep_mfma_probe.hip can build it with phase clocks)
#define SWM_CLOCK(i)
#define SWM_PCLOCK(i)
#include "sweep_mfma.inc"
#undef SWM_CLOCK
#undef SWM_PCLOCK

// mode 0: out = (W + gn B)^-1; mode 1: out = T, the inverse of the Cholesky factor of W (lower triangular); 64 < D <= 256
static int spd_block_mfma(plda_handle *h, int mode, const double *W, const double *B, const double *gn, int D, int ldin,
                          int64_t stride_in, double *out, int ldout, int64_t stride_out, int *dflag, int batch) {
  const int nt = (int)ceil_div(D, 16);
#define SWM(NTT)                                                                                                    \
  do {                                                                                                              \
    constexpr size_t lds = (size_t)((3 * NTT + 8) * 272 + 128) * 8;                                                 \
    if (!h->sweep_mfma_attr[NTT]) {                                                                                 \
      PLDA_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&spd_inverse_mfma_kernel<NTT, 0>),             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
      PLDA_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&spd_inverse_mfma_kernel<NTT, 1>),             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
      h->sweep_mfma_attr[NTT] = true;                                                                               \
    }                                                                                                               \
    if (mode == 0)                                                                                                  \
      spd_inverse_mfma_kernel<NTT, 0><<<batch, 1024, lds, h->stream>>>(W, B, gn, D, ldin, stride_in, out, ldout,    \
                                                                       stride_out, dflag);                          \
    else                                                                                                            \
      spd_inverse_mfma_kernel<NTT, 1><<<batch, 1024, lds, h->stream>>>(W, B, gn, D, ldin, stride_in, out, ldout,    \
                                                                       stride_out, dflag);                          \
  } while (0)
  switch (nt) {
    case 5: SWM(5); break;
    case 6: SWM(6); break;
    case 7: SWM(7); break;
    case 8: SWM(8); break;
    case 9: SWM(9); break;
    case 10: SWM(10); break;
    case 11: SWM(11); break;
    case 12: SWM(12); break;
    case 13: SWM(13); break;
    case 14: SWM(14); break;
    case 15: SWM(15); break;
    default: SWM(16); break;
  }
#undef SWM
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

// out[g] = (W + gn[g] B)^-1 for g < batch (B == nullptr: W[g]^-1 with batch stride `stride_in`); D <= 256
int spd_inverse_small(plda_handle *h, const double *W, const double *B, const double *gn, int D, int ldin,
                      int64_t stride_in, double *out, int ldout, int64_t stride_out, int *dflag, int batch) {
  if (h->sweep_variant == 0 && D > 64 && D <= 256)   // block sweeps on the matrix cores
    return spd_block_mfma(h, 0, W, B, gn, D, ldin, stride_in, out, ldout, stride_out, dflag, batch);
  if (h->sweep_variant == 0 || h->sweep_variant == 2) {     // four waves, 16 x 16 ownership (2: at every size; 1: the 16-wave kernel of round 2)
    const int nb16 = (int)ceil_div(D, 16);
#define SW16(NBB)                                                                                             \
  spd_inverse_sweep16_kernel<NBB><<<batch, 256, 0, h->stream>>>(W, B, gn, D, ldin, stride_in, out, ldout, \
                                                                stride_out, dflag)
    switch (nb16) {
      case 1: SW16(1); break;
      case 2: SW16(2); break;
      case 3: SW16(3); break;
      case 4: SW16(4); break;
      case 5: SW16(5); break;
      case 6: SW16(6); break;
      case 7: SW16(7); break;
      case 8: SW16(8); break;
      case 9: SW16(9); break;
      case 10: SW16(10); break;
      case 11: SW16(11); break;
      case 12: SW16(12); break;
      case 13: SW16(13); break;
      case 14: SW16(14); break;
      case 15: SW16(15); break;
      case 16: SW16(16); break;
      default: return fail(h, PLDA_E_INVAL, "spd_inverse: D=%d > 256 unsupported", D);
    }
#undef SW16
    PLDA_LAUNCH_CHECK(h);
    return PLDA_OK;
  }
  const int nb = (int)ceil_div(D, 32);
#define SW(NBB)                                                                                              \
  spd_inverse_sweep_kernel<NBB><<<batch, 1024, 0, h->stream>>>(W, B, gn, D, ldin, stride_in, out, ldout, \
                                                               stride_out, dflag)
  switch (nb) {
    case 1: SW(1); break;
    case 2: SW(2); break;
    case 3: SW(3); break;
    case 4: SW(4); break;
    case 5: SW(5); break;
    case 6: SW(6); break;
    case 7: SW(7); break;
    case 8: SW(8); break;
    default: return fail(h, PLDA_E_INVAL, "spd_inverse: D=%d > 256 unsupported", D);
  }
#undef SW
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

int spd_inverse_f64(plda_handle *h, const double *W, const double *B, const double *gn, int D, double *out,
                    int *dflag, int batch) {
  return spd_inverse_small(h, W, B, gn, D, D, 0, out, D, (int64_t)D * D, dflag, batch);
}

// Cholesky factor in registers (D <= 256), same ownership as the sweep kernel: thread (ty, tx) of the 32 x 32
// grid owns the lower-triangle elements (32a + ty, 32b + tx).  Column k: the owners publish it through LDS,
// everyone scales by 1/sqrt(A_kk) and applies the rank-1 update to the trailing blocks (a, b >= kb; rows or
// columns <= k get a zero factor, so there is no per-element branch).  L (zeros above the diagonal) goes
// to `out`.  One matrix per workgroup.
template <int NB>
__global__ __launch_bounds__(1024) void chol_small_kernel(const double *__restrict__ A, int D, int ldin,
                                                          int64_t stride_in, double *__restrict__ out, int ldout,
                                                          int64_t stride_out, int *flag) {
  constexpr int NE = NB * (NB + 1) / 2;
  __shared__ double v[2][NB * 32];
  const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
  A += (int64_t)blockIdx.x * stride_in;
  out += (int64_t)blockIdx.x * stride_out;
  double r[NE];
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const int i = a * 32 + ty, j = b * 32 + tx;
      r[a * (a + 1) / 2 + b] = (i < D && j <= i) ? A[(size_t)i * ldin + j] : 0.0;
    }
  bool bad = false;
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    for (int kl = 0; kl < 32; ++kl) {
      const int k = kb * 32 + kl;
      if (k >= D) break;
      double *vv = v[k & 1];
      if (tx == kl) {   // column k, rows >= k
#pragma unroll
        for (int a = kb; a < NB; ++a)
          if (a > kb || ty >= kl) vv[a * 32 + ty] = r[a * (a + 1) / 2 + kb];
      }
      __syncthreads();
      const double d = vv[k];
      if (!(d > 0.0)) bad = true;
      const double inv = rsqrt_nr(d);
      double ui[NB], vj[NB];
#pragma unroll
      for (int a = kb; a < NB; ++a) ui[a] = (a * 32 + ty > k) ? -vv[a * 32 + ty] * inv : 0.0;
#pragma unroll
      for (int b = kb; b < NB; ++b) vj[b] = (b * 32 + tx > k) ? vv[b * 32 + tx] * inv : 0.0;
#pragma unroll
      for (int a = kb; a < NB; ++a)
#pragma unroll
        for (int b = kb; b <= a; ++b) r[a * (a + 1) / 2 + b] = fma(ui[a], vj[b], r[a * (a + 1) / 2 + b]);
      if (tx == kl) {   // column k of L
#pragma unroll
        for (int a = kb; a < NB; ++a) {
          if (a > kb || ty > kl) r[a * (a + 1) / 2 + kb] = -ui[a];
          else if (ty == kl) r[a * (a + 1) / 2 + kb] = d * inv;
        }
      }
    }
  }
  if (bad && t == 0) *flag = 1;
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int i = a * 32 + ty, j = b * 32 + tx;
      if (i < D && j < D) {
        if (b <= a && j <= i) out[(size_t)i * ldout + j] = r[(a * (a + 1) / 2 + b) < NE ? a * (a + 1) / 2 + b : 0];
        else out[(size_t)i * ldout + j] = 0.0;
      }
    }
}

int chol_small(plda_handle *h, const double *A, int D, int ldin, int64_t stride_in, double *out, int ldout,
               int64_t stride_out, int *dflag, int batch) {
  const int nb = (int)ceil_div(D, 32);
#define CS(NBB) chol_small_kernel<NBB><<<batch, 1024, 0, h->stream>>>(A, D, ldin, stride_in, out, ldout, stride_out, dflag)
  switch (nb) {
    case 1: CS(1); break;
    case 2: CS(2); break;
    case 3: CS(3); break;
    case 4: CS(4); break;
    case 5: CS(5); break;
    case 6: CS(6); break;
    case 7: CS(7); break;
    case 8: CS(8); break;
    default: return fail(h, PLDA_E_INVAL, "chol_small: D=%d > 256 unsupported", D);
  }
#undef CS
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

__global__ void copy_block_kernel(const double *__restrict__ src, int lds, int64_t strides, double *__restrict__ dst,
                                  int ldd, int64_t strided, int rows, int cols, bool transpose) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  const int r = idx / cols, c = idx % cols;
  const double v = src[(int64_t)blockIdx.y * strides + (int64_t)r * lds + c];
  double *d = dst + (int64_t)blockIdx.y * strided;
  if (transpose) d[(int64_t)c * ldd + r] = v;
  else d[(int64_t)r * ldd + c] = v;
}

__global__ void zero_block_kernel(double *__restrict__ dst, int ldd, int64_t strided, int rows, int cols) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  dst[(int64_t)blockIdx.y * strided + (int64_t)(idx / cols) * ldd + idx % cols] = 0.0;
}

// Whitening factor T (lower triangular, T A T^T = I, i.e. T = chol(A)^-1) of [batch] SPD matrices of any size by
// block elimination over the register-resident Cholesky:
//   T11 = whiten(A11),  Y = T11 A12,  S = A22 - Y^T Y,  T22 = whiten(S),  T21 = -T22 Y^T T11.
// A: leading dimension lda, batch stride sa; T: ldt, st; `scr`: 2 n^2 doubles per matrix, batch stride sscr.
// The Schur complement is formed from the Cholesky factor of A11 (Y^T Y), as a blocked Cholesky does.  Block
// elimination with the explicit INVERSE of A11 instead (S = A22 - A12^T A11^-1 A12, the first version of the EM's
// D > 256 inverse) loses the small eigen-directions when A11 is itself ill conditioned: with fewer samples than
// dimensions (N - K < n1) the EM's W came out with 1000 x the error in psi of this form or of the unblocked sweep
// (scripts/stress_case.py shape 150 257 4 6 0.2: 6.5e-4 against 6e-7 at cond(W) = 1.4e9).
int whiten_blocked(plda_handle *h, const double *A, int n, int lda, int64_t sa, double *T, int ldt, int64_t st,
                   double *scr, int64_t sscr, int *dflag, int batch) {
  if (n > 64 && n <= 256 && h->sweep_variant == 0)     // Cholesky factor and its inverse in one kernel, on the matrix cores
    return spd_block_mfma(h, 1, A, nullptr, nullptr, n, lda, sa, T, ldt, st, dflag, batch);
  if (n <= 256) {
    PLDA_TRY(chol_small(h, A, n, lda, sa, scr, n, sscr, dflag, batch));
    return tri_invert_ld(h, scr, sscr, T, n, ldt, st, batch);
  }
  const int n1 = (int)round_up((int64_t)ceil_div(n, 2), 32), n2 = n - n1;
  const double *A12 = A + n1, *A22 = A + (int64_t)n1 * lda + n1;
  double *T11 = T, *T12 = T + n1, *T21 = T + (int64_t)n1 * ldt, *T22 = T + (int64_t)n1 * ldt + n1;
  double *Y = scr, *S = Y + (int64_t)n1 * n2, *tmp = S + (int64_t)n2 * n2, *sub = tmp + (int64_t)n2 * n1;
  const unsigned ub = (unsigned)batch;
  PLDA_TRY(whiten_blocked(h, A, n1, lda, sa, T11, ldt, st, sub, sscr, dflag, batch));
  PLDA_TRY(gemm_f64_batched(h, n1, n2, n1, 1.0, T11, ldt, 1, st, A12, lda, 1, sa, nullptr, 0.0, Y, n2, sscr, batch));
  copy_block_kernel<<<dim3((unsigned)ceil_div((int64_t)n2 * n2, 256), ub), 256, 0, h->stream>>>(A22, lda, sa, S, n2, sscr,
                                                                                            n2, n2, false);
  PLDA_LAUNCH_CHECK(h);
  PLDA_TRY(gemm_f64_batched(h, n2, n2, n1, -1.0, Y, 1, n2, sscr, Y, n2, 1, sscr, nullptr, 1.0, S, n2, sscr, batch));
  PLDA_TRY(whiten_blocked(h, S, n2, n2, sscr, T22, ldt, st, sub, sscr, dflag, batch));
  PLDA_TRY(gemm_f64_batched(h, n2, n1, n1, 1.0, Y, 1, n2, sscr, T11, ldt, 1, st, nullptr, 0.0, tmp, n1, sscr, batch));
  PLDA_TRY(gemm_f64_batched(h, n2, n1, n2, -1.0, T22, ldt, 1, st, tmp, n1, 1, sscr, nullptr, 0.0, T21, ldt, st, batch));
  zero_block_kernel<<<dim3((unsigned)ceil_div((int64_t)n1 * n2, 256), ub), 256, 0, h->stream>>>(T12, ldt, st, n1, n2);
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

// A_g = W + n_g B for the whitening of the grouped EM outside 64 < D <= 256 (inside that range the kernel forms it itself)
__global__ void group_sum_kernel(const double *__restrict__ W, const double *__restrict__ B, const double *__restrict__ gn,
                                 int64_t DD, double *__restrict__ A, int64_t sa) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx < DD) A[(int64_t)blockIdx.y * sa + idx] = fma(gn[blockIdx.y], B[idx], W[idx]);
}

// T_g = chol(W + gn[g] B)^-1 (lower triangular, zeros above the diagonal; T_g (W + n_g B) T_g^T = I) for g < batch, T_g at
// T + g D^2.  `scr`: 3 D^2 doubles per group (A_g and the blocked whitening's scratch), touched only outside 64 < D <= 256.
int whiten_groups_f64(plda_handle *h, const double *W, const double *B, const double *gn, int D, double *T, double *scr,
                      int *dflag, int batch) {
  const int64_t sDD = (int64_t)D * D;
  if (h->sweep_variant == 0 && D > 64 && D <= 256) return spd_block_mfma(h, 1, W, B, gn, D, D, 0, T, D, sDD, dflag, batch);
  group_sum_kernel<<<dim3((unsigned)ceil_div(sDD, 256), (unsigned)batch), 256, 0, h->stream>>>(W, B, gn, sDD, scr, 3 * sDD);
  PLDA_LAUNCH_CHECK(h);
  return whiten_blocked(h, scr, D, D, 3 * sDD, T, D, sDD, scr + sDD, 3 * sDD, dflag, batch);
}

// SPD inverse of [batch] matrices of any size: A^-1 = T^T T with T = whiten(A).  out may be A (dead once T exists), not scr;
// `scr`: 3 n^2 doubles per matrix (T + the whitening's scratch), batch stride sscr.
int spd_inverse_blocked(plda_handle *h, const double *A, int n, int lda, int64_t sa, double *out, int ldo,
                        int64_t so, double *scr, int64_t sscr, int *dflag, int batch) {
  if (n <= 256) return spd_inverse_small(h, A, nullptr, nullptr, n, lda, sa, out, ldo, so, dflag, batch);
  double *T = scr;
  PLDA_TRY(whiten_blocked(h, A, n, lda, sa, T, n, sscr, scr + (int64_t)n * n, sscr, dflag, batch));
  // (m, k) of T^T = T[k][m]
  return gemm_f64_batched(h, n, n, n, 1.0, T, 1, n, sscr, T, n, 1, sscr, nullptr, 0.0, out, ldo, so, batch);
}

// ------------------------------------------------------------------------------------
// symmetric eigensolver: one-sided (Hestenes) block Jacobi on the rows of A (= G), V = I.
// After convergence A = V G has mutually orthogonal rows, so the rows of V are the
// eigenvectors of G and lambda_p = a_p . v_p.
// ------------------------------------------------------------------------------------
// Block form of the one-sided Jacobi round: rows are grouped in blocks of JB = 4; a launch
// is one OUTER tournament round over the blocks; each workgroup takes one block pair,
// stages its 8 rows of A and of V in LDS and runs the full INNER tournament on them
// (7 rounds x 4 disjoint pairs, one wave per pair, __syncthreads between rounds), so 28
// rotations are applied per round trip to L2 instead of one.
constexpr int JB = 4;

template <int E>
__global__ __launch_bounds__(256) void jacobi_block_kernel(double *__restrict__ A, double *__restrict__ V,
                                                           int D, int nb_even, int oround, double tol,
                                                           int *__restrict__ rotations) {
  extern __shared__ __attribute__((aligned(16))) double rows[];   // [8][D] of A, then [8][D] of V
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nm1 = nb_even - 1;
  int bp, bq;
  if (blockIdx.x == 0) { bp = nm1; bq = oround; }
  else { bp = (oround + blockIdx.x) % nm1; bq = (oround - (int)blockIdx.x + nm1) % nm1; }
  if (bp > bq) { const int tt = bp; bp = bq; bq = tt; }
  // bye (the partner block does not exist): the rows of bq are treated as missing and the workgroup still
  // runs the INTRA-block pairs of bp -- with a single block (D <= 4) that is the only place they are rotated
  if (bp * JB >= D) return;
  if (rotations[2]) return;   // converged in an earlier sweep of this batch (jacobi_end_kernel)
  double *lA = rows, *lV = rows + 2 * JB * D;
  // global row of local row r (r < JB: block bp, else block bq); rows >= D do not exist
  auto grow = [&](int r) { return (r < JB ? bp * JB + r : bq * JB + (r - JB)); };
#pragma unroll
  for (int r = 0; r < 2 * JB; ++r) {
    const int g = grow(r);
    if (g < D)
      for (int c = t; c < D; c += 256) {
        lA[r * D + c] = A[(size_t)g * D + c];
        lV[r * D + c] = V[(size_t)g * D + c];
      }
  }
  __syncthreads();
  int nrot = 0;
  constexpr int NP = 2 * JB;   // 8 players
#pragma unroll 1
  for (int ir = 0; ir < NP - 1; ++ir) {
    int p, q;
    if (wave == 0) { p = NP - 1; q = ir; }
    else { p = (ir + wave) % (NP - 1); q = (ir - wave + (NP - 1)) % (NP - 1); }
    if (p > q) { const int tt = p; p = q; q = tt; }
    if (grow(p) < D && grow(q) < D) {
      double *ap = lA + p * D, *aq = lA + q * D;
      double x[E], y[E];
      double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int d = lane + e * 64;
        x[e] = d < D ? ap[d] : 0.0;
        y[e] = d < D ? aq[d] : 0.0;
        alpha += x[e] * x[e];
        beta += y[e] * y[e];
        gamma += x[e] * y[e];
      }
      alpha = wave_sum_f64(alpha);
      beta = wave_sum_f64(beta);
      gamma = wave_sum_f64(gamma);
      if (gamma != 0.0 && gamma * gamma > tol * tol * alpha * beta) {
        // tan(theta) = gamma / (a + sign(a) sqrt(a^2 + gamma^2)), a = (beta - alpha)/2 -- the same root
        // as sign(zeta) / (|zeta| + sqrt(1 + zeta^2)), zeta = a/gamma -- with one rsq + one rcp + one
        // rsq (hardware estimate + Newton steps) instead of three IEEE sqrt and three divisions
        const double a = 0.5 * (beta - alpha);
        const double hsq = a * a + gamma * gamma;
        const double h = hsq * rsqrt_nr(hsq);
        const double tn = gamma * rcp_nr(a + (a >= 0.0 ? h : -h));
        const double c = rsqrt_nr(1.0 + tn * tn), sn = c * tn;
        double *vp = lV + p * D, *vq = lV + q * D;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int d = lane + e * 64;
          if (d < D) {
            ap[d] = c * x[e] - sn * y[e];
            aq[d] = sn * x[e] + c * y[e];
            const double vx = vp[d], vy = vq[d];
            vp[d] = c * vx - sn * vy;
            vq[d] = sn * vx + c * vy;
          }
        }
        nrot++;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 2 * JB; ++r) {
    const int g = grow(r);
    if (g < D)
      for (int c = t; c < D; c += 256) {
        A[(size_t)g * D + c] = lA[r * D + c];
        V[(size_t)g * D + c] = lV[r * D + c];
      }
  }
  if (lane == 0 && nrot) atomicAdd(rotations, nrot);
}

// Gram form of the same outer round.  The inner tournament above is a latency chain of 7 rounds of
// (3 wave reductions -> scalar angle math -> rotation through LDS -> barrier).  Here the 8 x 8 Gram matrix
// H = R R^T of the staged rows is formed once on the MFMA pipe (v_mfma_f64_16x16x4_f64 multiplies AND
// reduces over k), ONE wave diagonalises H by cyclic two-sided Jacobi on 64 lanes (lane = entry (i, j);
// no barriers: a single wave's LDS operations execute in order) while accumulating the product Q of
// the rotations, and all threads then apply Q to the rows of A and V in one pass that writes straight to
// memory.  Mathematically the inner sweep is the one-sided sweep of the 8 rows (same angles, same
// threshold); the rows see one 8 x 8 orthogonal transform instead of 28 separate rotations.
// Measured per launch at D = 200 (25 workgroups): launch + staging 4.6 us, Gram 0.05 us, inner sweep
// 2.6 us (7 rounds of ~750 cycles: 12 LDS reads in one batch, two dependent rsq), apply + store 1.35 us
// = 8.6 us against 11.6 us for jacobi_block_kernel; a second inner sweep costs 2 us and saves no outer sweep.
constexpr int JG_MAX_INNER = 1;

__global__ __launch_bounds__(256) void jacobi_gram_kernel(double *__restrict__ A, double *__restrict__ V, int D,
                                                          int nb_even, int oround, double tol,
                                                          int *__restrict__ rotations) {
  extern __shared__ __attribute__((aligned(16))) double rows[];   // [8][D] of A, then [8][D] of V
  __shared__ double Hs[4][64];
  __shared__ double Qs[64];
  __shared__ int any_rot;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nm1 = nb_even - 1;
  int bp, bq;
  if (blockIdx.x == 0) { bp = nm1; bq = oround; }
  else { bp = (oround + blockIdx.x) % nm1; bq = (oround - (int)blockIdx.x + nm1) % nm1; }
  if (bp > bq) { const int tt = bp; bp = bq; bq = tt; }
  // bye (the partner block does not exist): the rows of bq are treated as missing and the workgroup still
  // runs the INTRA-block pairs of bp -- with a single block (D <= 4) that is the only place they are rotated
  if (bp * JB >= D) return;
  if (rotations[2]) return;   // converged in an earlier sweep of this batch (jacobi_end_kernel)
  constexpr int NP = 2 * JB;  // 8 rows
  double *lA = rows, *lV = rows + NP * D;
  auto grow = [&](int r) { return (r < JB ? bp * JB + r : bq * JB + (r - JB)); };
  // all 16 row segments of a column chunk are requested before any is consumed (a missing row is a
  // zero row: never rotated); addresses are clamped so that the loads need no predicate
  for (int c0 = 0; c0 < D; c0 += 256) {
    const int c = min(c0 + t, D - 1);
    double ta[NP], tv[NP];
#pragma unroll
    for (int r = 0; r < NP; ++r) {
      const int g = min(grow(r), D - 1);
      ta[r] = A[(size_t)g * D + c];
      tv[r] = V[(size_t)g * D + c];
    }
    if (c0 + t < D) {
#pragma unroll
      for (int r = 0; r < NP; ++r) {
        const bool ok = grow(r) < D;
        lA[r * D + c] = ok ? ta[r] : 0.0;
        lV[r * D + c] = ok ? tv[r] : 0.0;
      }
    }
  }
  if (t == 0) any_rot = 0;
  __syncthreads();
  // ---- H = lA lA^T: wave w takes the k-quads w, w+4, ... ----
  {
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    const int i = lane & 15, kq = lane >> 4;
    for (int k0 = wave * 4; k0 < D; k0 += 16) {
      const int k = k0 + kq;
      const double a = (i < NP && k < D) ? lA[i * D + k] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
    // C layout: col = lane & 15, row = (lane >> 4) + 4 * reg -> rows 0..7 are regs 0 and 1
    if ((lane & 15) < NP) {
      Hs[wave][((lane >> 4) + 0) * NP + (lane & 15)] = acc[0];
      Hs[wave][((lane >> 4) + 4) * NP + (lane & 15)] = acc[1];
    }
  }
  __syncthreads();
  // ---- wave 0: cyclic Jacobi on H, lane = (i, j) ----
  if (wave == 0) {
    const int i = lane >> 3, j = lane & 7;
    double *H = Hs[0];
    H[lane] = Hs[0][lane] + Hs[1][lane] + Hs[2][lane] + Hs[3][lane];
    Qs[lane] = i == j ? 1.0 : 0.0;
    // round-robin partner of player x in round ir: 7 <-> ir, otherwise x <-> (2 ir - x) mod 7
    int mi_t[NP - 1], mj_t[NP - 1];
#pragma unroll
    for (int ir = 0; ir < NP - 1; ++ir) {
      mi_t[ir] = i == NP - 1 ? ir : (i == ir ? NP - 1 : (2 * ir - i + 2 * (NP - 1)) % (NP - 1));
      mj_t[ir] = j == NP - 1 ? ir : (j == ir ? NP - 1 : (2 * ir - j + 2 * (NP - 1)) % (NP - 1));
    }
    bool ever = false;
    float relmax = 0.f;   // largest gamma^2 / (alpha beta) this workgroup rotated away
#pragma unroll 1
    for (int sweep = 0; sweep < JG_MAX_INNER; ++sweep) {
      bool rotated = false;
#pragma unroll
      for (int ir = 0; ir < NP - 1; ++ir) {
        // the entries a lane reads were written by OTHER lanes of this wave in the previous round: one
        // wave's LDS operations execute in order, the fence only stops the compiler from caching them
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int mi = mi_t[ir], mj = mj_t[ir];
        const int pi = i < mi ? i : mi, qi = i < mi ? mi : i, pj = j < mj ? j : mj, qj = j < mj ? mj : j;
        const double al[2] = {H[pi * NP + pi], H[pj * NP + pj]}, be[2] = {H[qi * NP + qi], H[qj * NP + qj]},
                     ga[2] = {H[pi * NP + qi], H[pj * NP + qj]};
        const double h00 = H[i * NP + j], h10 = H[mi * NP + j], h01 = H[i * NP + mj], h11 = H[mi * NP + mj];
        const double q0 = Qs[i * NP + j], q1 = Qs[mi * NP + j];
        double rs[2], rm[2];   // [0]: the rotation of i's pair, [1]: of j's pair; R_xx and R_x,mate(x)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          double c = 1.0, sn = 0.0;
          if (ga[w] != 0.0 && ga[w] * ga[w] > tol * tol * al[w] * be[w]) {
            relmax = fmaxf(relmax, (float)(ga[w] * ga[w] / (al[w] * be[w])));
            // the same angle as in jacobi_block_kernel through the double angle: with a = (beta - alpha)/2,
            // r = sqrt(a^2 + gamma^2): cos 2t = |a|/r, sin 2t = sign(a) gamma/r, so c^2 = (1 + |a|/r)/2 and
            // s = sin 2t / (2c) -- two dependent rsq instead of rsq -> rcp -> rsq, and c^2 + s^2 = 1 exactly
            const double a = 0.5 * (be[w] - al[w]);
            const double rinv = rsqrt_nr(a * a + ga[w] * ga[w]);
            const double c2 = fma(0.5 * fabs(a), rinv, 0.5);
            const double cinv = rsqrt_nr(c2);
            c = c2 * cinv;
            sn = (a >= 0.0 ? 0.5 : -0.5) * ga[w] * rinv * cinv;
            rotated = true;
          }
          rs[w] = c;
          rm[w] = (w == 0 ? i < mi : j < mj) ? -sn : sn;   // rows: x' = c x - s y (p), y' = s x + c y (q)
        }
        double hn = rs[0] * (h00 * rs[1] + h01 * rm[1]) + rm[0] * (h10 * rs[1] + h11 * rm[1]);
        if (j == mi && rm[0] != 0.0) hn = 0.0;   // the annihilated entry, exactly
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        H[lane] = hn;
        Qs[lane] = rs[0] * q0 + rm[0] * q1;
      }
      if (__ballot(rotated) == 0) break;
      ever = true;
    }
    for (int o = 32; o > 0; o >>= 1) relmax = fmaxf(relmax, __shfl_xor(relmax, o));
    if (lane == 0 && ever) {
      any_rot = 1;
      atomicAdd(rotations, 1);
      atomicMax(reinterpret_cast<unsigned *>(rotations) + 1, __float_as_uint(relmax));   // non-negative floats order as uints
    }
  }
  __syncthreads();
  if (!any_rot) return;
  // ---- rows <- Q rows, written straight to memory ----
  for (int c = t; c < D; c += 256) {
    double a[NP], v[NP];
#pragma unroll
    for (int r = 0; r < NP; ++r) { a[r] = lA[r * D + c]; v[r] = lV[r * D + c]; }
#pragma unroll
    for (int r = 0; r < NP; ++r) {
      const int g = grow(r);
      if (g < D) {
        double sa = 0.0, sv = 0.0;
#pragma unroll
        for (int x = 0; x < NP; ++x) {
          const double q = Qs[r * NP + x];
          sa = fma(q, a[x], sa);
          sv = fma(q, v[x], sv);
        }
        A[(size_t)g * D + c] = sa;
        V[(size_t)g * D + c] = sv;
      }
    }
  }
}

// Convergence is decided on the device: rot[0] rotations of the running sweep, rot[1] its largest rotated
// gamma^2 / (alpha beta) (float bits), rot[2] != 0 once converged (= sweeps it took), rot[3] sweeps run.
// A sweep (one graph replay) is begin -> rounds -> end; after convergence all of them return at once, so the
// host enqueues sweeps in batches and looks at rot[2] once per batch instead of synchronising every sweep.
__global__ void jacobi_begin_kernel(int *rot) {
  if (rot[2]) return;
  rot[0] = 0; rot[1] = 0;
}
__global__ void jacobi_end_kernel(int *rot, int use_relmax) {
  if (rot[2]) return;
  rot[3] += 1;
  // Quadratic convergence: a sweep whose largest rotated off-diagonal was gamma^2/(alpha beta) <= 1e-18
  // (|cos| <= 1e-9) leaves every pair far below the threshold tol ~ 1e-14, so the confirming sweep that
  // would find nothing to rotate is skipped.  (Only the Gram kernel reports the maximum.)
  const float relmax = __uint_as_float((unsigned)rot[1]);
  if (rot[0] == 0 || (use_relmax && relmax > 0.f && relmax <= 1e-18f)) rot[2] = rot[3];
}

__global__ void set_identity_kernel(double *V, int D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < D * D) V[idx] = (idx / D == idx % D) ? 1.0 : 0.0;
}

// lambda_p = a_p . v_p (one wave per row)
__global__ void eig_values_kernel(const double *__restrict__ A, const double *__restrict__ V, int D,
                                  double *__restrict__ lam) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= D) return;
  double acc = 0.0;
  for (int d = lane; d < D; d += 64) acc += A[(size_t)p * D + d] * V[(size_t)p * D + d];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) lam[p] = acc;
}

// rank sort descending (SortSvd), floor at zero (ApplyFloor), permute eigenvector rows
__global__ void eig_sort_kernel(const double *__restrict__ lam, const double *__restrict__ V, int D,
                                double *__restrict__ s, double *__restrict__ Vsorted, bool floor_at_zero) {
  const int p = blockIdx.x;
  const double lp = lam[p];
  __shared__ int rank_s;
  if (threadIdx.x == 0) rank_s = 0;
  __syncthreads();
  int cnt = 0;
  for (int q = threadIdx.x; q < D; q += blockDim.x) {
    const double lq = lam[q];
    if (lq > lp || (lq == lp && q < p)) cnt++;
  }
  if (cnt) atomicAdd(&rank_s, cnt);
  __syncthreads();
  const int r = rank_s;
  if (threadIdx.x == 0) s[r] = (lp > 0.0 || !floor_at_zero) ? lp : 0.0;
  for (int d = threadIdx.x; d < D; d += blockDim.x) Vsorted[(size_t)r * D + d] = V[(size_t)p * D + d];
}

int eig_sort_rows(plda_handle *h, const double *lam, const double *V, int D, double *s, double *Vsorted) {
  eig_sort_kernel<<<D, 64, 0, h->stream>>>(lam, V, D, s, Vsorted, !h->eig_keep_sign);
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

// One sweep = nb_even - 1 outer rounds; its launches are captured once into a hipGraph
// (per handle, re-captured when D or the buffers change) and replayed.
static int jacobi_sweep_graph(plda_handle *h, double *G, double *V, int D, double tol, int *drot) {
  const int nb = (int)ceil_div(D, JB);
  const int nb_even = nb + (nb & 1);
  const int wgs = nb_even / 2;
  const int E = (int)ceil_div(D, 64);
  const size_t lds = (size_t)4 * JB * D * sizeof(double);
  const bool gram = h->jacobi_variant != 1;
  if (h->stream == nullptr) {
    // HIP's legacy default stream cannot be captured: launch the rounds directly
    PLDA_HIP(h, hipFuncSetAttribute(E <= 1 ? reinterpret_cast<const void *>(&jacobi_block_kernel<1>)
                                    : E <= 2 ? reinterpret_cast<const void *>(&jacobi_block_kernel<2>)
                                    : E <= 4 ? reinterpret_cast<const void *>(&jacobi_block_kernel<4>)
                                    : E <= 8 ? reinterpret_cast<const void *>(&jacobi_block_kernel<8>)
                                             : reinterpret_cast<const void *>(&jacobi_block_kernel<16>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    jacobi_begin_kernel<<<1, 1, 0, h->stream>>>(drot);
    if (gram) PLDA_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&jacobi_gram_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int round = 0; round < nb_even - 1; ++round) {
#define JR(EE) jacobi_block_kernel<EE><<<wgs, 256, lds, h->stream>>>(G, V, D, nb_even, round, tol, drot)
      if (gram) jacobi_gram_kernel<<<wgs, 256, lds, h->stream>>>(G, V, D, nb_even, round, tol, drot);
      else if (E <= 1) JR(1);
      else if (E <= 2) JR(2);
      else if (E <= 4) JR(4);
      else if (E <= 8) JR(8);
      else JR(16);
#undef JR
    }
    jacobi_end_kernel<<<1, 1, 0, h->stream>>>(drot, gram ? 1 : 0);
    PLDA_LAUNCH_CHECK(h);
    return PLDA_OK;
  }
  if (h->jac_exec && (h->jac_G != G || h->jac_V != V || h->jac_D != D)) {
    (void)hipGraphExecDestroy(h->jac_exec);
    h->jac_exec = nullptr;
  }
  if (!h->jac_exec) {
#define JATTR(EE) PLDA_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&jacobi_block_kernel<EE>), \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
    if (E <= 1) JATTR(1); else if (E <= 2) JATTR(2); else if (E <= 4) JATTR(4); else if (E <= 8) JATTR(8); else JATTR(16);
#undef JATTR
    PLDA_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&jacobi_gram_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipGraph_t graph = nullptr;
    PLDA_HIP(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed));
    jacobi_begin_kernel<<<1, 1, 0, h->stream>>>(drot);
    for (int round = 0; round < nb_even - 1; ++round) {
#define JR(EE) jacobi_block_kernel<EE><<<wgs, 256, lds, h->stream>>>(G, V, D, nb_even, round, tol, drot)
      if (gram) jacobi_gram_kernel<<<wgs, 256, lds, h->stream>>>(G, V, D, nb_even, round, tol, drot);
      else if (E <= 1) JR(1);
      else if (E <= 2) JR(2);
      else if (E <= 4) JR(4);
      else if (E <= 8) JR(8);
      else JR(16);
#undef JR
    }
    jacobi_end_kernel<<<1, 1, 0, h->stream>>>(drot, gram ? 1 : 0);
    hipError_t ec = hipStreamEndCapture(h->stream, &graph);
    if (ec != hipSuccess) return hip_fail(h, ec, "hipStreamEndCapture(jacobi sweep)", __FILE__, __LINE__);
    ec = hipGraphInstantiate(&h->jac_exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ec != hipSuccess) { h->jac_exec = nullptr; return hip_fail(h, ec, "hipGraphInstantiate(jacobi sweep)", __FILE__, __LINE__); }
    h->jac_G = G; h->jac_V = V; h->jac_D = D;
  }
  PLDA_HIP(h, hipGraphLaunch(h->jac_exec, h->stream));
  return PLDA_OK;
}

// Eigenvectors are returned in the ROWS of Vrows (sorted by descending eigenvalue).
// warm != nullptr: rows of `warm` are an orthonormal guess (the previous EM iteration's
// eigenvectors); the iteration then starts from A = warm G, V = warm instead of A = G, V = I.
int sym_eig_f64(plda_handle *h, double *G, int D, double *s, double *Vrows, int *sweeps_out,
                const double *warm) {
  if (D > 1024)
    return fail(h, PLDA_E_INVAL,
                "sym_eig: D=%d in (1024, 2048] is served only by the direct solver, whose tridiagonalisation needs ceil(D/8) = %d "
                "co-resident workgroups; this device did not run it (fewer usable CUs than that -- CU masking or a partitioned GPU? "
                "-- or the method was forced to Jacobi), and the block Jacobi fallback holds 16 rows in LDS, i.e. stops at D = 1024",
                D, (D + 7) / 8);
  const size_t DD = (size_t)D * D;
  PLDA_HIP(h, h->w[14].reserve(DD * 8 * 2 + (size_t)D * 8 + 64));
  double *V = h->w[14].as<double>();
  double *A = V + DD;
  double *lam = A + DD;
  int *drot = reinterpret_cast<int *>(lam + D);   // 4 ints (jacobi_begin_kernel)
  if (warm) {
    PLDA_HIP(h, hipMemcpyAsync(V, warm, DD * 8, hipMemcpyDeviceToDevice, h->stream));
    PLDA_TRY(gemm_f64(h, D, D, D, 1.0, warm, D, 1, G, D, 1, nullptr, 0.0, A, D));
  } else {
    set_identity_kernel<<<(unsigned)ceil_div((int64_t)DD, 256), 256, 0, h->stream>>>(V, D);
    PLDA_HIP(h, hipMemcpyAsync(A, G, DD * 8, hipMemcpyDeviceToDevice, h->stream));
  }
  PLDA_LAUNCH_CHECK(h);
  const double tol = 2.220446049250313e-16 * 4.0 * sqrt((double)D);
  // sweeps are enqueued in batches and the device-side convergence flag is read once per batch (a cold
  // start needs ~11 sweeps at D = 200, ~13 at D = 512; sweeps past convergence return at once)
  const int max_sweeps = 40;
  int sweeps = 0, hrot[4] = {0, 0, 0, 0};
  PLDA_HIP(h, hipMemsetAsync(drot, 0, 4 * sizeof(int), h->stream));
  for (int enq = 0; D > 1 && enq < max_sweeps && !hrot[2];) {
    const int batch = std::min(max_sweeps - enq, enq == 0 ? (warm ? 3 : 9) : 2);
    for (int b = 0; b < batch; ++b) PLDA_TRY(jacobi_sweep_graph(h, A, V, D, tol, drot));
    enq += batch;
    PLDA_HIP(h, hipMemcpyAsync(hrot, drot, 4 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
  }
  if (D > 1 && !hrot[2]) return fail(h, PLDA_E_NUMERIC, "sym_eig: Jacobi did not converge in %d sweeps", max_sweeps);
  sweeps = D > 1 ? hrot[2] : 0;
  if (sweeps_out) *sweeps_out = sweeps;
  h->jac_total_sweeps += sweeps;
  eig_values_kernel<<<(unsigned)ceil_div(D, 4), 256, 0, h->stream>>>(A, V, D, lam);
  eig_sort_kernel<<<D, 64, 0, h->stream>>>(lam, V, D, s, Vrows, !h->eig_keep_sign);
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

// ------------------------------------------------------------------------------------
// simultaneous diagonalisation (PldaEstimator::GetOutput's transform)
// ------------------------------------------------------------------------------------
__global__ void symmetrize_kernel(double *G, int D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D * D) return;
  const int i = idx / D, j = idx % D;
  if (j < i) {
    const double v = 0.5 * (G[(size_t)i * D + j] + G[(size_t)j * D + i]);
    G[(size_t)i * D + j] = v;
    G[(size_t)j * D + i] = v;
  }
}

// direct method where it applies, block Jacobi otherwise (or when it gives up): what the LDA solvers call
int sym_eig_auto_f64(plda_handle *h, double *G, int D, double *s, double *Vrows) {
  int status = 1;
  if (h->eig_variant != 1) PLDA_TRY(sym_eig_dc_f64(h, G, D, s, Vrows, &status));
  h->eig_last_method = status == 0 ? 2 : 1;
  if (status != 0) PLDA_TRY(sym_eig_f64(h, G, D, s, Vrows, nullptr, nullptr));
  return PLDA_OK;
}

// The work is enqueued in two parts so that GetOutput needs no host round trip of its own:
//   simdiag_enqueue   Cholesky whitening, congruence, eigensolver, T (and Tinv).  With the direct eigensolver
//                     nothing is read back; *pending = true and the device flags are left for
//   simdiag_finish    reads the Cholesky and eigensolver flags (after a synchronisation the caller needs
//                     anyway) and, if the direct method gave up, repeats the decomposition with block Jacobi.
// simdiag_f64 = both, for callers that want the result at once.
int simdiag_finish_with(plda_handle *h, const double *W, const double *B, int D, double *T, double *Tinv, double *psi,
                        int chol_flag, int eig_status, bool *redo);

static int simdiag_run(plda_handle *h, const double *W, const double *B, int D, double *T, double *Tinv, double *psi,
                       bool warm_start, bool allow_direct, bool defer, bool *pending) {
  const size_t DD = (size_t)D * D;
  const size_t need = DD * 8 * 6 + 64;
  const bool fresh = h->w[13].cap < need || h->simdiag_D != D;
  PLDA_HIP(h, h->w[13].reserve(need));
  h->simdiag_D = D;
  double *scr = h->w[13].as<double>();                                    // 2 DD: whitening scratch
  double *T1 = scr + 2 * DD, *tmp = T1 + DD, *G = tmp + DD, *Vr = G + DD;   // Vr persists: next call's warm start
  int *dflag = reinterpret_cast<int *>(Vr + DD);
  const bool warm = warm_start && !fresh && h->simdiag_has_vr;
  const bool direct = allow_direct && !warm && h->eig_variant != 1;
  if (pending) *pending = false;
  PLDA_HIP(h, hipMemsetAsync(dflag, 0, sizeof(int), h->stream));
  // T1 = chol(W)^-1.  (Any T1 with T1 W T1^T = I gives the same final transform up to row signs; this is
  // the Cholesky one, as in the reference's GetOutput.)
  {
    TraceScope ts(h, "getoutput.whiten (chol + inverse)");
    PLDA_TRY(whiten_blocked(h, W, D, D, 0, T1, D, 0, scr, 0, dflag, 1));
  }
  if (!direct) {
    // checked HERE for the Jacobi solver: with a W that is not positive definite T1 is full of NaNs, and it would
    // run its 40 sweeps on garbage and report "did not converge" instead of the actual cause.  (The direct
    // method refuses non-finite input at once and sets its flag: simdiag_finish then finds the Cholesky flag.)
    int hflag = 0;
    PLDA_HIP(h, hipMemcpyAsync(&hflag, dflag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    PLDA_HIP(h, hipStreamSynchronize(h->stream));
    if (hflag) return fail(h, PLDA_E_NUMERIC, "within-class covariance is not positive definite");
  }
  // tmp = T1 B ; G = tmp T1^T
  {
    TraceScope ts(h, "getoutput.congruence", 4.0 * (double)D * D * D, 1);
    PLDA_TRY(gemm_f64(h, D, D, D, 1.0, T1, D, 1, B, D, 1, nullptr, 0.0, tmp, D));
    PLDA_TRY(gemm_f64(h, D, D, D, 1.0, tmp, D, 1, T1, 1, D, nullptr, 0.0, G, D));
    symmetrize_kernel<<<(unsigned)ceil_div((int64_t)DD, 256), 256, 0, h->stream>>>(G, D);
    PLDA_LAUNCH_CHECK(h);
  }
  // direct method first (cold starts: the closed-form EM never diagonalises, so GetOutput always starts cold);
  // block Jacobi when it declines (D > its limit, an iteration cap) or for warm starts of the per-iteration EM arm
  int dc_status = 1;
  if (direct) {
    PLDA_TRY(sym_eig_dc_f64(h, G, D, psi, Vr, defer ? nullptr : &dc_status));
    if (defer) dc_status = 0;   // assumed; simdiag_finish checks
  }
  h->eig_last_method = dc_status == 0 ? 2 : 1;
  if (dc_status != 0) {
    if (direct) {   // the direct method gave up: was it the Cholesky factor?
      int hflag = 0;
      PLDA_HIP(h, hipMemcpy(&hflag, dflag, sizeof(int), hipMemcpyDeviceToHost));
      if (hflag) return fail(h, PLDA_E_NUMERIC, "within-class covariance is not positive definite");
    }
    TraceScope ts(h, "getoutput.eig.jacobi");
    PLDA_TRY(sym_eig_f64(h, G, D, psi, warm ? tmp : Vr, nullptr, warm ? Vr : nullptr));
    if (warm) PLDA_HIP(h, hipMemcpyAsync(Vr, tmp, DD * 8, hipMemcpyDeviceToDevice, h->stream));
  }
  h->simdiag_has_vr = true;
  // T = Vr T1 (rows of Vr are eigenvectors) ; Tinv = T^-1 = W T^T (from T W T^T = I)
  {
    TraceScope ts(h, "getoutput.transform");
    PLDA_TRY(gemm_f64(h, D, D, D, 1.0, Vr, D, 1, T1, D, 1, nullptr, 0.0, T, D));
    if (Tinv) PLDA_TRY(gemm_f64(h, D, D, D, 1.0, W, D, 1, T, 1, D, nullptr, 0.0, Tinv, D));
  }
  if (pending) *pending = direct && defer;
  return PLDA_OK;
}

int simdiag_enqueue(plda_handle *h, const double *W, const double *B, int D, double *T, double *Tinv, double *psi,
                    bool *pending) {
  return simdiag_run(h, W, B, D, T, Tinv, psi, false, true, true, pending);
}

// after the stream has been synchronised by the caller's own copies: *redo = true when the decomposition had to be
// repeated (T / Tinv / psi were rewritten and the caller's copies of them are stale)
int simdiag_finish(plda_handle *h, const double *W, const double *B, int D, double *T, double *Tinv, double *psi,
                   bool *redo) {
  *redo = false;
  const size_t DD = (size_t)D * D;
  const int *dflag = reinterpret_cast<const int *>(h->w[13].as<double>() + 6 * DD);
  int hflag = 0, status = 0;
  PLDA_HIP(h, hipMemcpy(&hflag, dflag, sizeof(int), hipMemcpyDeviceToHost));
  PLDA_TRY(sym_eig_dc_status(h, &status));
  return simdiag_finish_with(h, W, B, D, T, Tinv, psi, hflag, status, redo);
}

// the two device flags simdiag_finish reads (for a caller that fetches them with its own copies: fit's model export)
void simdiag_flags(plda_handle *h, int D, const int **chol_flag, const int **eig_flag) {
  *chol_flag = reinterpret_cast<const int *>(h->w[13].as<double>() + 6 * (size_t)D * D);
  *eig_flag = h->eigdc_flag;          // nullptr: the direct method did not take the problem (status 8)
}

int simdiag_finish_with(plda_handle *h, const double *W, const double *B, int D, double *T, double *Tinv, double *psi,
                        int chol_flag, int eig_status, bool *redo) {
  *redo = false;
  if (chol_flag) return fail(h, PLDA_E_NUMERIC, "within-class covariance is not positive definite");
  if (eig_status == 0) return PLDA_OK;
  *redo = true;
  return simdiag_run(h, W, B, D, T, Tinv, psi, false, false, false, nullptr);
}

int simdiag_f64(plda_handle *h, const double *W, const double *B, int D, double *T, double *Tinv,
                double *psi, bool warm_start) {
  return simdiag_run(h, W, B, D, T, Tinv, psi, warm_start, true, false, nullptr);
}

}  // namespace plda
