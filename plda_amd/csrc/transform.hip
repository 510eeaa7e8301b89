// plda_amd/csrc/transform.hip -- K4: batched Plda::TransformIvector (reached at /root/reference/src/pldamodule.cpp:171
// for every label's mean and at :224 for every cohort row): t = offset + T x, f = sqrt(Dout / sum_d t_d^2 / (psi_d + 1/n)),
// out = f t.  fp64 MFMA bound: 2 R Dout Din flop over 8 R (Din + Dout) bytes.
//
// One pass (Dout <= 512): a workgroup of 8 waves owns 16 * 8 / CH rows and ALL columns -- wave (rg, ch) accumulates
// 16 rows x NT 16-column tiles in v_mfma_f64_16x16x4_f64 accumulators (CH column slices per row group, their row sums
// meet through LDS) -- so the row's sum of t_d^2 / (psi_d + 1/n) is there when the contraction ends and the normalised
// row is written once.  (The general GEMM + length_norm_kernel pair -- still the path for Dout > 512 -- writes T x, reads
// it back and writes it again: 24 N D bytes moved for 16, and 17-25 % of the time.)  Operand stages of 16 k: T's rows
// for all columns + the workgroup's rows of X, k-contiguous with a row pitch of 17 doubles (conflict-free fragment
// reads), fetched global -> registers under the MFMAs of the previous stage and written to the other buffer behind
// them; one barrier per stage.  Same k order and accumulator layout as gemm_f64_kernel.
//
// Round 3 (fraction of the fp64 MFMA peak at 100k x 200 / 1.2M x 256 / 1M x 512: 0.43 / 0.67 / 0.72 -> 0.53 / 0.76 / 0.77):
//   * no tail round.  The persistent grid used to walk over ceil(R / 128) blocks, so 782 blocks on 256 CUs (the C2 shape)
//     took FOUR rounds for 3.05 rounds of work.  Now the main launch covers a whole number of rounds and the rows that are
//     left go to a second launch whose blocks are as small as it takes to occupy every CU once: 64, 32 or 16 rows, the 8
//     waves sharing a block's rows by COLUMN slices (CH = 2, 4, 8).  A tail launch costs ~18 us (it is bound by the
//     thirteen stage round trips, not by MFMAs); small calls (a few hundred rows) gain the same way: 51 -> 18 us.
//   * row pitch 18 doubles where the LDS allows (conflict-free: see TfGeom::LD); 17 had one conflict per 32 lanes.
//   * the zero-padded copy of T is cached per model (rebuilt when fit / set_model / truncate / smooth change it), not
//     rebuilt on every call.
//   * the last stage of a Din that is not a multiple of 16 runs only the k-steps that hold data (D = 200: 2 of 4).
//   * with a uniform count the length-norm weights 1 / (psi_d + 1/n) are formed once per block and column (they were
//     a reciprocal + two Newton steps per ELEMENT); per-row counts are their own instantiation.
//   * the first operand stage of a workgroup's NEXT block is requested before the epilogue of the current one and lands
//     in LDS behind it; the epilogue's LDS scratch lives in the other stage buffer, so no barrier separates the blocks.
// What bounds it now (ablation at 98 304 x 200 = exactly three rounds, 178 us): without the epilogue 163 us, without
// the MFMAs and their fragment reads 98 us -- the global -> registers -> LDS -> barrier skeleton alone takes 55 % of the
// time, 2.6 us per 16-k stage against 2.8 us of MFMA work per SIMD at 13 tiles (3.4 at 16 tiles, which is why D = 256
// and 512 sit at 0.76), and the two overlap only partly.  Measured and dropped in round 3: deeper stages (above);
// touching X's lines three stages ahead so that the stage's own loads hit L2 (no change: it is not HBM latency);
// two 64-row workgroups per CU at 4 waves per SIMD (no change); small blocks whose waves stream their own fragments
// from L2 without LDS (16 cache lines per load instruction: twice as slow as the staged small blocks).
// Tried and dropped (round 2): 4-wave workgroups of 64 rows (T re-read twice as often: 25-50 % slower at D = 200 and
// 256, also where two of them fit a CU); a second fragment register set filled one k-step ahead, with and without
// sched_group_barrier forcing one LDS read between every two MFMAs (3-10 % slower, spills at (16, 2)); 16-byte granules.
#include "common.hpp"

#include <algorithm>

namespace plda {

// ------------------------------------------------------------------------------------
// separate length-norm pass of the two-kernel arm: t = offset + T x (the GEMM wrote T x into out),
// f = sqrt(Dout / sum_d t_d^2 / (psi_d + 1/n)), out = f t.  One wave per row, fp64.
// ------------------------------------------------------------------------------------
__global__ void length_norm_kernel(double *__restrict__ out, int64_t R, int Dout,
                                   const double *__restrict__ offset, const double *__restrict__ psi,
                                   const int32_t *__restrict__ n_arr, int n_uniform) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= R) return;
  const double inv_n = 1.0 / (n_arr ? (double)n_arr[row] : (double)n_uniform);
  double *t = out + row * (int64_t)Dout;
  double acc = 0.0;
  for (int d = lane; d < Dout; d += 64) {
    const double v = t[d] + offset[d];
    acc += v * v / (psi[d] + inv_n);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  const double f = sqrt((double)Dout / acc);
  for (int d = lane; d < Dout; d += 64) t[d] = f * (t[d] + offset[d]);
}

typedef double f64x4s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double tf_rcp(double x) {   // hardware estimate + two Newton steps: full precision
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// geometry of one instantiation, shared by the kernel and its launcher
template <int NT, int CH, int KS, int RT = 1>
struct TfGeom {
  static constexpr int RG = 8 / CH;                    // row groups of 16 RT rows
  static constexpr int ROWS = 16 * RT * RG;
  static constexpr int COLS = 16 * NT * CH;
  static constexpr int RPP = 512 / KS;                 // rows one fetch pass of the 512 threads covers (KS k each)
  static constexpr int TP = (COLS + RPP - 1) / RPP;    // fetch passes over T's rows
  static constexpr int TR = TP * RPP;                  // rows of T's LDS stage (and minimum rows of the padded T)
  static constexpr int XP = (ROWS + RPP - 1) / RPP;
  // row pitch in doubles.  A fragment read is 16 rows x 4 k-quads of 8 bytes, served 32 lanes at a time over 64 banks
  // of 4 bytes: the rows' start banks 2 LD i mod 64 must be 16 different multiples of 4, i.e. LD = 2 (mod 4):
  // KS + 2.  Where that does not fit the LDS, KS + 1 (one conflict per 32-lane group for KS = 16: PMC, round 2).
  static constexpr int LD = ((size_t)(TR + ROWS) * (KS + 2) * 16 <= 160 * 1024) ? KS + 2 : KS + 1;
  static constexpr int STAGE = (TR + ROWS) * LD;
  static constexpr size_t LDS_BYTES = (size_t)2 * STAGE * 8;
};

// T arrives zero-padded ([>= TR rows][Dinp = Din rounded up to KS], pad_transform_kernel), so its loads need no
// clamps and its LDS writes no predicates; X's row pointers are clamped once per block.  (With clamped
// indices and zero-selects at every load and store the stage loop carried 2.3 vector-ALU instructions per MFMA --
// 64-bit address arithmetic, compares, selects -- each costing the SIMD's matrix pipe an issue slot: PMC, MFMA busy
// 70 % of the cycles at D = 256.)
//
// KS, the depth of a stage (a multiple of 4), is a parameter of the geometry; every class runs 16 (see the dispatch).
// RT (round 4, A/B arm only): row tiles per wave.  With one row tile a wave reads 1 + NT fragments per NT MFMAs (D = 200: 14
// for 13) -- 67 bytes per clock of LDS fragment traffic per CU beside the stage writes; with RT = 2 and half the columns it
// reads 2 + NT for 2 NT MFMAs (9 for 14), the same block of 128 rows.  Slower (see transform_class).
// QF (round 6; norm()'s model pass, MPlda_norm pldamodule.cpp:235-250 by moments): the same product with T := C (a symmetric
// D x D matrix), but the epilogue keeps TWO numbers per row x instead of the normalised row:
//     s1 = sum_c x_c (C x + lin)_c = x^T C x + lin . x        s2 = sum_c x_c (m_c - q_c x_c / 2)
// written as out[row] = s2 + *mD (the model's z-norm mean) and out2[row] = sqrt(max(s1 + *crr, 0)) (its std).  `offset` carries
// lin, `psi` carries q.  The rows' own values are read back from global memory (L2: the stage loop just streamed them).
struct TfQuad { const double *m; const double *mD; const double *crr; double *out2; };
template <int NT, int CH, int KS, bool PERROW, int RT = 1, bool QF = false>
__global__ __launch_bounds__(512) void transform_fused_kernel(const double *__restrict__ X, int64_t R, int Din,
                                                              const double *__restrict__ Tpad, int Dinp, int Dout,
                                                              const double *__restrict__ offset,
                                                              const double *__restrict__ psi,
                                                              const int32_t *__restrict__ n_arr, int n_uniform,
                                                              double *__restrict__ out, const TfQuad qf = TfQuad{}) {
  using G = TfGeom<NT, CH, KS, RT>;
  constexpr int RG = G::RG, ROWS = G::ROWS, COLS = G::COLS, RPP = G::RPP, TP = G::TP, TR = G::TR, XP = G::XP;
  constexpr int LD = G::LD, STAGE = G::STAGE, KSTEPS = KS / 4;
  constexpr bool XPART = ROWS % RPP != 0;   // the last X pass covers rows beyond the block: no LDS row for them
  extern __shared__ __attribute__((aligned(16))) double tf_lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int rg = wave % RG, ch = wave / RG;
  const int fi = lane & 15, fk = lane >> 4;
  const int lk = t % KS, lr = t / KS;       // this thread's k and first row inside a fetch pass
  const bool loader = lr < RPP;             // (512 is not a multiple of every KS: a few threads carry nothing)
  const double *tptr = Tpad + (int64_t)min(lr, RPP - 1) * Dinp + lk;
  const int64_t tstep = (int64_t)RPP * Dinp;
  const int tfrag = (ch * NT * 16 + fi) * LD + fk, xfrag = (TR + rg * 16 * RT + fi) * LD + fk;
  const bool early = wave >= 4;
  // Persistent: one workgroup per CU walks over the row blocks.  (One workgroup fills a CU -- registers -- so between
  // two of them the CU stood idle for the whole turnaround, ~17k cycles per 128-row block: wave launch, LDS
  // allocation, the first loads.)
  const int64_t nblocks = (R + ROWS - 1) / ROWS;
  int64_t blk = blockIdx.x;
  if (blk >= nblocks) return;

  double rt[TP], rx[XP];
  const double *xptr[XP];
  auto point = [&](int64_t b) {
#pragma unroll
    for (int p = 0; p < XP; ++p) xptr[p] = X + min(b * ROWS + min(lr, RPP - 1) + RPP * p, R - 1) * (int64_t)Din + lk;
  };
  // fetch half q (q = 4: everything): the loads of a stage are issued in two halves, behind the MFMAs of the first
  // two k-steps (later ones arrive too late for the wave's LDS write and it waits for them).  All at once at the top
  // of a stage they are 48 KB per workgroup through the CU's 64 B/clk vector memory path: ~750 cycles in which both
  // waves of every SIMD stand in load issue and nobody feeds the matrix pipe (phase timing,
  // scripts/probe/transform_tl.hip: stage time = 8 192 MFMA cycles + exactly that).
  auto fetch = [&](int k0, int q) {
#pragma unroll
    for (int p = 0; p < TP; ++p)
      if (q == 4 || (p & 1) == q) rt[p] = tptr[p * tstep + k0];
    // the last stage of a Din that is not a multiple of KS must not read past a row's end (zeroed at the LDS write)
    const int ko = (k0 + KS <= Din) ? k0 : min(k0 + lk, Din - 1) - lk;
#pragma unroll
    for (int p = 0; p < XP; ++p)
      if (q == 4 || (p & 1) == q) rx[p] = xptr[p][ko];
  };
  auto stage = [&](double *buf, int k0) {
    if (!loader) return;
#pragma unroll
    for (int p = 0; p < TP; ++p) buf[(lr + RPP * p) * LD + lk] = rt[p];
    const bool kok = k0 + lk < Din;
#pragma unroll
    for (int p = 0; p < XP; ++p)
      if (!XPART || p < XP - 1 || lr + RPP * p < ROWS) buf[(TR + lr + RPP * p) * LD + lk] = kok ? rx[p] : 0.0;
  };

  // The two waves of a SIMD (w and w + 4) take their non-MFMA work at opposite ends of a stage: the first fetches
  // the next stage, runs its MFMAs and writes the fetched registers to the other buffer at the END; the second
  // writes them at the START (they were fetched one stage earlier), fetches the stage after next and then runs its
  // MFMAs -- so one of the two is feeding the matrix pipe while the other moves data.
  int cur = 0;
  point(blk);
  fetch(0, 4);
  stage(tf_lds, 0);
  if (early && Din > KS) fetch(KS, 4);
  __syncthreads();
  for (;;) {
    const int64_t r0 = blk * ROWS;
    f64x4s acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[rt][i] = f64x4s{0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < Din; k0 += KS) {
      const bool more = k0 + KS < Din;
      if (early && more) stage(tf_lds + (cur ^ 1) * STAGE, k0 + KS);
      const int kf = early ? k0 + 2 * KS : k0 + KS;     // the stage this wave fetches during this one
      const bool dofetch = kf < Din;
      const int ksteps = min(KSTEPS, (Din - k0 + 3) >> 2);   // the last stage of a ragged Din: only the k-steps that hold data
      const double *Ts = tf_lds + cur * STAGE + tfrag, *Xs = tf_lds + cur * STAGE + xfrag;
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        if (kk < ksteps) {
          double a[RT];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) a[rt] = Xs[rt * 16 * LD + kk * 4];
#pragma unroll
          for (int tn = 0; tn < NT; ++tn) {
            const double b = Ts[tn * 16 * LD + kk * 4];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt], b, acc[rt][tn], 0, 0, 0);
          }
        }
        if (dofetch && kk < 2) fetch(kf, kk);
        // fragment reads stay inside their k-step (all steps' reads hoisted to the top of the stage need
        // KSTEPS x (1 + NT) register pairs next to the accumulators and spill); the SIMD's other wave covers them
        asm volatile("" ::: "memory");
      }
      if (!early && more) stage(tf_lds + (cur ^ 1) * STAGE, k0 + KS);
      __syncthreads();
      cur ^= 1;
    }
    // both stage buffers are dead behind the loop's last barrier.  The first stage of this workgroup's next block is
    // requested now and lands in registers under the epilogue; the epilogue's LDS scratch takes buffer cur ^ 1, the
    // next block's first stage goes to buffer cur.
    const int64_t nblk = blk + gridDim.x;
    const bool has_next = nblk < nblocks;
    if (has_next) {
      point(nblk);
      fetch(0, 4);
    }

    // offset and the length-norm weights of every column go through LDS: read from global tile by tile -- the only order
    // that does not spill -- they were 16 dependent L2 round trips, half of the 19k-cycle epilogue of a workgroup
    // that has the CU to itself.  Uniform count: ep = 1 / (psi + 1/n), the weight itself; per-row counts: ep = psi.
    double *eo = tf_lds + (cur ^ 1) * STAGE, *ep = eo + COLS, *red = ep + COLS;
    const double inv_nu = PERROW ? 0.0 : 1.0 / (double)n_uniform;
    for (int c = t; c < COLS; c += 512) {
      eo[c] = c < Dout ? offset[c] : 0.0;
      const double ps = c < Dout ? psi[c] : 1.0;
      ep[c] = QF ? (c < Dout ? ps : 0.0) : PERROW ? ps : tf_rcp(ps + inv_nu);
      if constexpr (QF) red[2 * CH * ROWS + c] = c < Dout ? qf.m[c] : 0.0;
    }
    __syncthreads();
    if constexpr (QF) {
      const double *em = red + 2 * CH * ROWS;
      double p1[4], p2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) p1[r] = p2[r] = 0.0;
      // this lane's four rows of the block, as 32-bit element offsets from the block's first row (clamped to the last row)
      const double *xb = X + r0 * (int64_t)Din;
      const int rloc = rg * 16 + fk, rmax = (int)min((int64_t)ROWS - 1, R - 1 - r0);
      unsigned xo[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) xo[r] = (unsigned)(min(rloc + 4 * r, rmax) * Din);
      // The rows' values come back from L2 two tiles (eight loads) at a time: left alone the scheduler hoists all 4 NT loads to
      // the top and the kernel spills 181 registers beside its NT accumulator tiles (a "memory" clobber does not hold back loads
      // through a const __restrict__ pointer).  `bump` is always 0, but only the empty asm that follows a tile pair's FMAs
      // knows: the next pair's addresses wait for it.
      unsigned bump = 0;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        const int col = (ch * NT + tn) * 16 + fi;
        const bool cok = col < Dout && col < Din;
        const unsigned cc = (unsigned)min(col, Din - 1) + bump;
        const double off = eo[col], q = ep[col], m = em[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double xv = cok ? xb[xo[r] + cc] : 0.0;
          p1[r] = fma(xv, acc[0][tn][r] + off, p1[r]);
          p2[r] = fma(xv, fma(-0.5 * q, xv, m), p2[r]);
        }
        if ((tn & 1) == 1) asm volatile("" : "+v"(bump), "+v"(p1[0]), "+v"(p1[1]), "+v"(p1[2]), "+v"(p1[3]));
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double a = p1[r], b = p2[r];
        a += dpp_f64<0xB1>(a); b += dpp_f64<0xB1>(b);
        a += dpp_f64<0x4E>(a); b += dpp_f64<0x4E>(b);
        a += dpp_f64<0x141>(a); b += dpp_f64<0x141>(b);
        a += dpp_f64<0x140>(a); b += dpp_f64<0x140>(b);
        p1[r] = a; p2[r] = b;
      }
      if (CH > 1) {
        if (fi == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            red[ch * ROWS + rloc + 4 * r] = p1[r];
            red[(CH + ch) * ROWS + rloc + 4 * r] = p2[r];
          }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double a = 0.0, b = 0.0;
#pragma unroll
          for (int c = 0; c < CH; ++c) { a += red[c * ROWS + rloc + 4 * r]; b += red[(CH + c) * ROWS + rloc + 4 * r]; }
          p1[r] = a; p2[r] = b;
        }
      }
      if (fi == 0 && ch == 0) {
        const double mD = *qf.mD, crr = *qf.crr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t g = r0 + rloc + 4 * r;
          if (g < R) {
            const double var = p1[r] + crr;
            out[g] = p2[r] + mD;
            qf.out2[g] = sqrt(var > 0.0 ? var : 0.0);
          }
        }
      }
      if (!has_next) break;
      stage(tf_lds + cur * STAGE, 0);
      if (early && Din > KS) fetch(KS, 4);
      __syncthreads();
      blk = nblk;
      continue;
    }
    // accumulator layout: column = lane & 15 of the tile, row = (lane >> 4) + 4 * reg of the row tile
    double part[RT][4];
    int64_t grow[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { part[rt][r] = 0.0; grow[rt][r] = r0 + rg * 16 * RT + rt * 16 + fk + 4 * r; }
    if constexpr (PERROW) {
      double inv_n[RT][4];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) inv_n[rt][r] = 1.0 / (double)n_arr[min(grow[rt][r], R - 1)];
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        const int col = (ch * NT + tn) * 16 + fi;
        const bool cok = col < Dout;
        const double off = eo[col], ps = ep[col];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double v = cok ? acc[rt][tn][r] + off : 0.0;
            acc[rt][tn][r] = v;
            part[rt][r] = fma(v * v, tf_rcp(ps + inv_n[rt][r]), part[rt][r]);
          }
      }
    } else {
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        const int col = (ch * NT + tn) * 16 + fi;
        const bool cok = col < Dout;
        const double off = eo[col], w = ep[col];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double v = cok ? acc[rt][tn][r] + off : 0.0;
            acc[rt][tn][r] = v;
            part[rt][r] = fma(v * v, w, part[rt][r]);
          }
      }
    }
    // the sum over the 16 column lanes of a row: four DPP steps in the register file (round 4; __shfl_xor of a double is
    // two ds_bpermute round trips per step -- on a part whose fp64 MFMAs share the SIMD with every vector instruction,
    // the epilogue's cycles are not hidden by anything)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double pz = part[rt][r];
        pz += dpp_f64<0xB1>(pz);     // quad_perm [1,0,3,2]
        pz += dpp_f64<0x4E>(pz);     // quad_perm [2,3,0,1]
        pz += dpp_f64<0x141>(pz);    // row_half_mirror
        pz += dpp_f64<0x140>(pz);    // row_mirror: every lane of the row holds the row's sum
        part[rt][r] = pz;
      }
    if (CH > 1) {     // the other column slices of the same rows live in waves (rg, ch'): exchange through LDS
      if (fi == 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[ch * ROWS + rg * 16 * RT + rt * 16 + fk + 4 * r] = part[rt][r];
      }
      __syncthreads();
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double sum = 0.0;
#pragma unroll
          for (int c = 0; c < CH; ++c) sum += red[c * ROWS + rg * 16 * RT + rt * 16 + fk + 4 * r];   // fixed order: every slice gets the same sum
          part[rt][r] = sum;
        }
    }
    const double sqrt_dout = sqrt((double)Dout);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // sqrt(Dout / sum) = sqrt(Dout) * rsqrt(sum): hardware estimate + two Newton steps (a division and a square root
        // were ~70 instructions per row)
        const double tot = part[rt][r];
        double y = __builtin_amdgcn_rsq(tot);
        y = y * fma(-0.5 * tot * y, y, 1.5);
        y = y * fma(-0.5 * tot * y, y, 1.5);
        const double f = sqrt_dout * y;
        if (grow[rt][r] < R) {
          double *o = out + grow[rt][r] * (int64_t)Dout;
#pragma unroll
          for (int tn = 0; tn < NT; ++tn) {
            const int col = (ch * NT + tn) * 16 + fi;
            if (col < Dout) o[col] = f * acc[rt][tn][r];
          }
        }
      }
    if (!has_next) break;
    // (no barrier: the scratch above is in buffer cur ^ 1, which is next written behind the barrier below)
    stage(tf_lds + cur * STAGE, 0);
    if (early && Din > KS) fetch(KS, 4);
    __syncthreads();
    blk = nblk;
  }
}

// ------------------------------------------------------------------------------------
// A/B arm (PLDA_TRANSFORM_VARIANT=7): the same block shapes with the operand stages brought in by LDS DMA
// (`buffer_load_dwordx4 ... lds`) instead of global -> registers -> ds_write: no staging registers, no LDS-write
// instructions, no load issue in the MFMA stream, a ring of three stage buffers with a stage in flight across each
// barrier (raw s_barrier + counted vmcnt: __syncthreads would drain the DMA), the DMA stream running through the epilogue
// into the next block.  Built because the kernel above still spends 55 % of its time when its MFMAs are taken out
// (round-3 ablation) -- and measured 2-6 % SLOWER than it (C2 0.513 against 0.533 of the fp64 peak, C4 0.725 / 0.748,
// C3 0.721 / 0.766): what the stage rhythm costs is not the staging instructions.  Same results bit for bit.
//   * LDS image of a stage: rows of 128 B (16 k), UNPADDED -- a DMA piece is one wave's 64 lanes x 16 B = 1 KiB = 8 rows,
//     lane-linear -- with the 16-byte chunks of a row XOR-swizzled by (row >> 1) & 7 on the SOURCE side (lane l of a piece
//     fetches chunk (l & 7) ^ swizzle of row l >> 3), so that the fragment read of 16 rows x one k-quad hits 32 different
//     bank pairs.  T rows first (COLS of them), then the block's ROWS rows of X.
//   * piece p of a stage belongs to wave p mod 8 (every wave issues the same number: the count of its `vmcnt`); odd and
//     even pieces differ in the swizzle's high bit, and a wave only ever has one parity: one lane offset per operand.
//   * rows of X beyond R come back as zeros (buffer bounds); k beyond Din is cut by skipping whole k-steps and zeroing
//     the A fragment of a partial one (T's padding is zero, but 0 x a neighbour row's NaN would not be).
//   * offset / weight vectors live in LDS for the whole kernel; the DMA stream runs through the epilogue into the next
//     block (its stages land in the ring while the rows are normalised and stored).
// ------------------------------------------------------------------------------------
#define TF_LDS_AS __attribute__((address_space(3)))

template <int NT, int CH>
struct TfDmaGeom {
  static constexpr int RG = 8 / CH, ROWS = 16 * RG, COLS = 16 * NT * CH;
  static constexpr int TPC = COLS / 8, XPC = ROWS / 8, NPC = TPC + XPC, PPW = (NPC + 7) / 8;
  static constexpr int SB = (COLS + ROWS) * 128;                      // bytes of one stage
  static constexpr int SCRB = (2 * COLS + CH * ROWS) * 8;             // offset / weights / row-sum exchange
  static constexpr int NSTG = (3 * SB + SCRB <= 160 * 1024) ? 3 : 2;
  static constexpr size_t LDS_BYTES = (size_t)NSTG * SB + SCRB;
};

template <int NT, int CH, bool PERROW>
__global__ __launch_bounds__(512) void transform_dma_kernel(const double *__restrict__ X, int64_t R, int Din,
                                                            const double *__restrict__ Tpad, int Dinp, int padrows, int Dout,
                                                            const double *__restrict__ offset,
                                                            const double *__restrict__ psi,
                                                            const int32_t *__restrict__ n_arr, int n_uniform,
                                                            double *__restrict__ out) {
  using G = TfDmaGeom<NT, CH>;
  constexpr int RG = G::RG, ROWS = G::ROWS, COLS = G::COLS, TPC = G::TPC, NPC = G::NPC, PPW = G::PPW;
  constexpr int SB = G::SB, NSTG = G::NSTG;
  static_assert(NPC >= 8, "every wave moves at least one piece");
  extern __shared__ __attribute__((aligned(16))) double tf_lds[];
  TF_LDS_AS char *const lds = (TF_LDS_AS char *)tf_lds;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int rg = wave % RG, ch = wave / RG;
  const int fi = lane & 15, fk = lane >> 4;
  const int64_t nblocks = (R + ROWS - 1) / ROWS;
  int64_t blk = blockIdx.x;
  if (blk >= nblocks) return;
  const int nstages = (Din + 15) >> 4;

  // offset and length-norm weights of every column, once.  Uniform count: ep = 1 / (psi + 1/n); per-row counts: ep = psi.
  double *const eo = reinterpret_cast<double *>(reinterpret_cast<char *>(tf_lds) + NSTG * SB), *const ep = eo + COLS, *const red = ep + COLS;
  {
    const double inv_nu = PERROW ? 0.0 : 1.0 / (double)n_uniform;
    for (int c = t; c < COLS; c += 512) {
      eo[c] = c < Dout ? offset[c] : 0.0;
      const double ps = c < Dout ? psi[c] : 1.0;
      ep[c] = PERROW ? ps : tf_rcp(ps + inv_nu);
    }
  }

  // ---- DMA side ----
  const int rl = lane >> 3;
  const int clog = (lane & 7) ^ ((((wave & 1) << 2) + (rl >> 1)) & 7);     // the chunk of its row this lane fetches
  const int vT = rl * Dinp * 8 + clog * 16, vX = rl * Din * 8 + clog * 16;
  const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(Tpad), 0, padrows * Dinp * 8, 0x00020000);
  int64_t dblk = blk;
  int dst = 0, dbuf = 0;
  bool dok = true;
  auto x_rsrc = [&](int64_t b) {
    const int64_t rows = min((int64_t)ROWS, R - b * ROWS);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(X + b * ROWS * (int64_t)Din), 0, (int)(rows * Din * 8), 0x00020000);
  };
  __amdgpu_buffer_rsrc_t rsX = x_rsrc(dblk);
  auto issue = [&]() {       // the cursor's stage into ring slot dbuf, then step the cursor
    if (dok) {
      const int kb = dst * 128;
      TF_LDS_AS char *const sb = lds + dbuf * SB;
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        int p = wave + 8 * j;
        if (p >= NPC) p -= 8;                                           // (a duplicate: equal piece counts for every wave)
        if (p < TPC)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsT, (TF_LDS_AS void *)(sb + p * 1024), 16, vT, p * 8 * Dinp * 8 + kb, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (TF_LDS_AS void *)(sb + p * 1024), 16, vX, (p - TPC) * 8 * Din * 8 + kb, 0, 0);
      }
    }
    dbuf = dbuf + 1 == NSTG ? 0 : dbuf + 1;
    if (++dst == nstages) {
      dst = 0;
      dblk += gridDim.x;
      dok = dok && dblk < nblocks;
      if (dok) rsX = x_rsrc(dblk);
    }
  };

  // ---- compute side ----
  const int sw = (fi >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) koff[kk] = (((kk * 2 + (fk >> 1)) ^ sw) << 4) + ((fk & 1) << 3);
  const int aoff = (COLS + rg * 16 + fi) * 128, boff = (ch * NT * 16 + fi) * 128;
  const int tailk = Din & 3;                                   // a partial last k-step: lanes fk >= tailk carry no data
  int cbuf = 0;
  bool prev_real = false, stores_pending = false;
#pragma unroll
  for (int i = 0; i < NSTG - 1; ++i) { prev_real = dok; issue(); }

  for (;;) {
    const int64_t r0 = blk * ROWS;
    f64x4s acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f64x4s{0.0, 0.0, 0.0, 0.0};
    for (int st = 0; st < nstages; ++st) {
      // this wave's pieces of the stage have landed (one younger stage may stay in flight); then everybody's have,
      // and nobody reads the slot the next DMA overwrites
      if (NSTG >= 3 && prev_real && !stores_pending) __builtin_amdgcn_s_waitcnt(0x0070 | PPW);   // vmcnt(PPW) lgkmcnt(0)
      else __builtin_amdgcn_s_waitcnt(0x0070);                                                   // vmcnt(0) lgkmcnt(0)
      stores_pending = false;
      asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
      prev_real = dok;
      issue();
      const char *const sbase = reinterpret_cast<const char *>(tf_lds) + cbuf * SB;
      const int k0 = st << 4;
      const int ksteps = min(4, (Din - k0 + 3) >> 2);
      const int zkk = (tailk && st == nstages - 1 && fk >= tailk) ? ksteps - 1 : -1;
      // (fragments one k-step ahead on a second register set -- there is room for it here -- were 10-12 % SLOWER again, as
      //  in the register-staged kernel: C2 0.47 against 0.52, C4 0.66 against 0.75)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk < ksteps) {
          double a = *reinterpret_cast<const double *>(sbase + aoff + koff[kk]);
          a = kk == zkk ? 0.0 : a;
#pragma unroll
          for (int tn = 0; tn < NT; ++tn) {
            const double b = *reinterpret_cast<const double *>(sbase + boff + tn * 2048 + koff[kk]);
            acc[tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[tn], 0, 0, 0);
          }
        }
        asm volatile("" ::: "memory");
      }
      cbuf = cbuf + 1 == NSTG ? 0 : cbuf + 1;
    }

    // ---- epilogue (the DMA of the next block's first stages is in flight) ----
    double part[4] = {0.0, 0.0, 0.0, 0.0};
    int64_t grow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) grow[r] = r0 + rg * 16 + fk + 4 * r;
    if constexpr (PERROW) {
      double inv_n[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) inv_n[r] = 1.0 / (double)n_arr[min(grow[r], R - 1)];
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        const int col = (ch * NT + tn) * 16 + fi;
        const bool cok = col < Dout;
        const double off = eo[col], ps = ep[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double v = cok ? acc[tn][r] + off : 0.0;
          acc[tn][r] = v;
          part[r] = fma(v * v, tf_rcp(ps + inv_n[r]), part[r]);
        }
      }
    } else {
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        const int col = (ch * NT + tn) * 16 + fi;
        const bool cok = col < Dout;
        const double off = eo[col], w = ep[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double v = cok ? acc[tn][r] + off : 0.0;
          acc[tn][r] = v;
          part[r] = fma(v * v, w, part[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) part[r] += __shfl_xor(part[r], o);
    }
    if (CH > 1) {     // the other column slices of the same rows live in waves (rg, ch'): exchange through LDS
      if (fi == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[ch * ROWS + rg * 16 + fk + 4 * r] = part[r];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0) only: the DMA stays in flight across this barrier
      asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double sum = 0.0;
#pragma unroll
        for (int c = 0; c < CH; ++c) sum += red[c * ROWS + rg * 16 + fk + 4 * r];   // fixed order: every slice gets the same sum
        part[r] = sum;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double f = sqrt((double)Dout / part[r]);
      if (grow[r] < R) {
        double *o = out + grow[r] * (int64_t)Dout;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) {
          const int col = (ch * NT + tn) * 16 + fi;
          if (col < Dout) o[col] = f * acc[tn][r];
        }
      }
    }
    stores_pending = true;            // stores and loads retire out of order with respect to each other: count nothing
    blk += gridDim.x;
    if (blk >= nblocks) break;
  }
}

// ------------------------------------------------------------------------------------
// transform_treg_kernel (round 4; PLDA_TRANSFORM_VARIANT=6, an A/B arm: Dout in (192, 208], Din = 200, a uniform count --
// the C2 shape) -- T NEVER LEAVES THE CU.  The kernels above re-stage all of T (333 KB at D = 200) for every 128 rows: 62 %
// of what a block moves through the global -> LDS path.  A CU's four SIMDs hold 4 x 512 registers x 64 lanes x 4 B =
// 512 KB -- T's MFMA B fragments (v_mfma_f64_16x16x4_f64: one double per lane; 13 column tiles x 50 k-steps = 650
// doubles per lane over the CU) fit.  Eight waves, two per SIMD (256 registers each), in two roles:
//   wave (s, 0): the column tiles s and s + 4, all 50 k-steps             -- 100 fragments, 100 MFMAs per row group
//                (the last 16 of them in LDS: with 200 registers of fragments the allocator spilled 17 to scratch
//                memory and reloaded them, a vmcnt(0) each, inside the MFMA loop)
//   wave (s, 1): the tile s + 8 and k-steps [13 s, 13 s + 13) of the 13th  --  63 fragments,  63 MFMAs per row group
// so that every SIMD carries 163 MFMAs of 64 cycles per 16 rows, loaded once per launch.  What streams is X alone, one
// MFMA row group (16 rows, 25 KB) per step, brought in by LDS DMA three groups ahead (a ring of four slots): a 1 KiB
// DMA piece is the A operands of two k-steps; lanes 4 m .. 4 m + 3 fetch the four 16-byte k-pairs of row m's 64-byte
// sector (one request per quad at the address unit), XOR-swizzled by (m >> 2) & 3 so that a fragment read -- 16 rows x
// one k-quad -- spreads over all 64 banks of the unpadded image.  Per group: MFMAs; the 13th tile's four partial
// accumulators and every wave's part of the rows' weighted squares (four sums over the 16 column lanes reduced
// TOGETHER: each DPP exchange halves the live values) -> LDS; ONE barrier (vmcnt(0) in front of it: the operands
// requested in this step have landed); totals, one rsqrt + two Newton steps per row instead of a division and a
// square root, stores through a range-checked descriptor (no row / column branches).
// What was measured on the way (scripts/transform_stream_probe.py, profiles/r04_transform_treg_probe.txt):
//   * ONE wave per SIMD holding a quarter of T (326 registers), epilogue behind the MFMAs: 0.24 ms at C2 against 0.19
//     for the kernels above.  Its timing arms are exactly additive -- MFMAs 4.4 us per group (their ideal), everything
//     else 2.8 us, together 7.1 -- also after the epilogue had been software-pipelined into the MFMA stream slice by
//     slice (a few instructions behind every k-step, scheduling barriers between; sched_group_barrier pipelines gave up
//     after five MFMAs): a wave's own vector instructions do not run in the shadow of its own MFMAs; the shadow belongs
//     to the SIMD's other wave.  Hence two waves per SIMD.
//   * __builtin_amdgcn_raw_ptr_buffer_load_lds makes the compiler's wait-count pass put s_waitcnt vmcnt(0) in front of
//     every later ds_read (it cannot tell which LDS reads a DMA write may alias): a memory round trip per piece in the
//     middle of the MFMA stream, ~1 700 cycles each.  The DMA is inline assembly here; the waits are the explicit ones.
//   * all 25 pieces of a group issued together behind the barrier queue at the CU's one address unit (~150 cycles per
//     piece with 64 separate 16-byte requests, fewer with the quad-contiguous mapping): they go out one per 8 k-steps.
//   * both waves of a SIMD stopping for their epilogues at the same barrier leaves the matrix pipe idle; run in different
//     orders around it (below), one wave's epilogue under the other's MFMAs takes three times as long as alone: the step
//     stays at 15 400 cycles for 10 400 of MFMAs (phase stamps, PLDA_TRANSFORM_VARIANT=14).
// Result: C2 (100k rows) 0.21 ms against 0.19, 800k rows 1.26 against 1.24 ms -- level with the kernels above, not
// ahead: a group takes 6.2 us where its MFMAs are 4.4 (at 2.4 GHz), with or without the reordering.  Not the product
// path; kept with its timing arms because the four findings above are what the next attempt starts from.
// Same sums per output as the kernels above for the columns of full tiles (k ascending in one accumulator); the
// thirteenth tile's columns add four partial sums, the row's norm adds its terms in another order and takes its
// square root by Newton steps: last-ulp differences (tests: 1e-12 against the fp64 oracle).
// ------------------------------------------------------------------------------------
template <int NT, int KSTEPS>
struct TregGeom {
  static_assert(NT == 13, "roles: tiles s, s + 4 | tile s + 8 and a quarter of tile 12");
  static constexpr int QS = (KSTEPS + 3) / 4;              // k-steps of the split tile per SIMD
  static constexpr int NPAIR = KSTEPS / 2;                 // DMA pieces (two k-steps of 16 rows) per row group
  static constexpr int PW = (NPAIR + 7) / 8;               // DMA pieces per wave and group (a ragged split's last ones go to a dump)
  static constexpr int GB = NPAIR * 1024;                  // bytes of a row group's slot in LDS
  static constexpr int NBUF = 4;                           // ring of row-group slots
  // scratch per parity: partial tiles [simd][lane] x 4 doubles | row sums [wave][16] | a dump for the lanes that hold no row sum
  static constexpr int SCR = 4 * 64 * 32 + 8 * 16 * 8 + 8 * 64 * 8;
  // the two-tile waves keep the last LK k-steps of their second tile's fragments in LDS, not in registers: 100 fragments
  // (200 registers) beside accumulators and epilogue spilled 17 of them to scratch memory, reloaded with a vmcnt(0)
  // each in the middle of the MFMA stream (3.5 us per group: measured)
  static constexpr int LK = 16;
  static constexpr int TL = 4 * LK * 512;                  // [simd][LK][64 lanes] doubles
  static constexpr int DUMP = 1024;                        // where the DMA pieces that do not exist land
  static constexpr size_t LDS_BYTES = (size_t)NBUF * GB + 2 * SCR + TL + DUMP;
};

// four sums over the 16 lanes of a DPP row at once: after the two merging exchanges (quad_perm xor 1, xor 2) a lane holds
// the partial sum of p[lane & 3]; row_ror:4 and row_ror:8 add the other three lanes of its class
__device__ __forceinline__ double row_sum4_by_class(const double (&p)[4], int lane) {
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
  const double k0 = b0 ? p[1] : p[0], s0 = b0 ? p[0] : p[1];
  const double k1 = b0 ? p[3] : p[2], s1 = b0 ? p[2] : p[3];
  const double w0 = k0 + dpp_f64<0xB1>(s0), w1 = k1 + dpp_f64<0xB1>(s1);
  const double k = b1 ? w1 : w0, sd = b1 ? w0 : w1;
  double y = k + dpp_f64<0x4E>(sd);
  y += dpp_f64<0x124>(y);     // row_ror:4
  y += dpp_f64<0x128>(y);     // row_ror:8
  return y;
}

struct TregArgs {
  const double *X; int64_t R; int Din; const double *Tpad; int Dinp; int Dout;
  const double *offset; const double *psi; int n_uniform; double *out; unsigned long long *dbg;
};

// one wave's loop.  FTW full tiles (tile index simd + 4 f + TB), SPL: also k-steps [simd QS, +QS) of tile NT - 1
template <int NT, int KSTEPS, int MODE, int FTW, int TB, bool SPL>
__device__ __forceinline__ void treg_wave(const TregArgs &A, TF_LDS_AS char *const lds, const int lane, const int wave, const int simd) {
  using G = TregGeom<NT, KSTEPS>;
  constexpr int QS = G::QS, NPAIR = G::NPAIR, PW = G::PW, GB = G::GB, NBUF = G::NBUF, SCR = G::SCR;
  constexpr int NA = FTW + (SPL ? 1 : 0);
  constexpr int LK = FTW == 2 ? G::LK : 0;                    // trailing k-steps of the last full tile whose fragments live in LDS
  constexpr int KR = KSTEPS - LK;                             // ... and the k-steps of it held in registers
  typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
  const int fi = lane & 15, fk = lane >> 4;
  const int64_t R = A.R;
  const int Din = A.Din, Dout = A.Dout, Dinp = A.Dinp;
  const int64_t ng = (R + 15) >> 4;

  // ---- T: this wave's B fragments, once ----
  double tb[FTW][KSTEPS];                                      // (of the last tile only [0, KR) is ever used: the rest is dead code)
  TF_LDS_AS double *const tl = (TF_LDS_AS double *)(lds + NBUF * GB + 2 * SCR) + (simd * G::LK) * 64 + lane;
#pragma unroll
  for (int f = 0; f < FTW; ++f) {
    const double *tp_ = A.Tpad + (int64_t)((simd + 4 * f + TB) * 16 + fi) * Dinp + fk;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if (f == FTW - 1 && ks >= KR) tl[(ks - KR) * 64] = tp_[4 * ks];
      else tb[f][ks] = tp_[4 * ks];
    }
  }
  const int q0 = simd * QS;                                   // first k-step of this SIMD's share of the split tile
  const int qvalid = SPL ? max(0, min(QS, KSTEPS - q0)) : 0;
  double tq[SPL ? QS : 1];
  if (SPL) {
    const double *tp_ = A.Tpad + (int64_t)((NT - 1) * 16 + fi) * Dinp + fk;
#pragma unroll
    for (int j = 0; j < QS; ++j) tq[j] = j < qvalid ? tp_[4 * (q0 + j)] : 0.0;   // (a share that runs past the last k-step: zero fragments)
  }
  // ---- per-column constants of the epilogue (uniform count: the length-norm weight is 1 / (psi + 1/n)) ----
  const double inv_nu = 1.0 / (double)A.n_uniform;
  double offs[FTW], wts[FTW], offq = 0.0, wq = 0.0;
#pragma unroll
  for (int f = 0; f < FTW; ++f) {
    const int col = (simd + 4 * f + TB) * 16 + fi;
    offs[f] = col < Dout ? A.offset[col] : 0.0;
    wts[f] = col < Dout ? tf_rcp(A.psi[col] + inv_nu) : 0.0;
  }
  {   // (both roles: every wave forms the split tile's row sums)
    const int col = (NT - 1) * 16 + fi;
    offq = col < Dout ? A.offset[col] : 0.0;
    wq = col < Dout ? tf_rcp(A.psi[col] + inv_nu) : 0.0;
  }
  const double sqrt_dout = sqrt((double)Dout);

  // ---- X by LDS DMA, fragment order.  Piece j of a group = k-steps 2 j and 2 j + 1: lane l = (ks_sel = l >> 5,
  //      p = (l >> 4) & 1, r = l & 15) fetches X[row r][8 j + 4 ks_sel + 2 p .. + 1] to byte 16 l of the piece.
  //      The whole address rides on the vector offset: it is what the descriptor range-checks (rows past R, pieces that
  //      do not exist and groups past the end land as zeros -- no branch).
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(A.X), 0, (int)(unsigned)(R * Din * 8), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(A.out, 0, (int)(unsigned)(R * (int64_t)Dout * 8), 0x00020000);
  // (lanes 4 m .. 4 m + 3 fetch the four 16-byte k-pairs of row m's 64-byte sector -- one request per quad at the address
  //  unit, not four -- in the order c ^ ((m >> 2) & 3): the piece's LDS image is row-major with 64 bytes per row, and the
  //  swizzle spreads the 16 rows of a fragment read over all 64 banks)
  const unsigned lane_voff = (unsigned)((lane >> 2) * Din + ((lane & 3) ^ ((lane >> 4) & 3)) * 2) * 8u;
  auto dma_piece = [&](int64_t g, int slot, bool live, int p) {
    if (MODE & 1) return;
    const int j = wave + 8 * p;
    const unsigned voff = (live && j < NPAIR) ? lane_voff + (unsigned)((g * 16 * Din + 8 * j) * 8) : 0xfffffff0u;
    // (inline assembly, not __builtin_amdgcn_raw_ptr_buffer_load_lds: the compiler's wait-count pass cannot tell which LDS
    //  reads a DMA write may alias and puts s_waitcnt vmcnt(0) in front of EVERY later ds_read -- a memory round trip per
    //  piece in the middle of the MFMA stream, 1 700 cycles each (measured).  The waits this kernel needs are the two
    //  explicit ones in front of its barriers.)
    const unsigned ldsaddr = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(j < NPAIR ? lds + slot * GB + j * 1024 : lds + NBUF * GB + 2 * SCR + G::TL));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(ldsaddr), "v"(voff), "s"(rsX) : "memory");
  };
  auto dma_group = [&](int64_t g, int slot, bool live) {
#pragma unroll
    for (int p = 0; p < PW; ++p) dma_piece(g, slot, live, p);
  };
  // fragment k-step ks of the group in slot s, lane i = (kk = i >> 4, r = i & 15): k-pair 2 (ks & 1) + (kk >> 1) of row r
  // sits at 16-byte position pair ^ ((r >> 2) & 3) of the row's 64 bytes in piece ks >> 1
  const unsigned fr = (unsigned)(lane & 15), fkk = (unsigned)(lane >> 4);
  const unsigned aoff0 = fr * 64u + (((0u + (fkk >> 1)) ^ ((fr >> 2) & 3u)) * 16u) + (fkk & 1u) * 8u;
  const unsigned aoff1 = fr * 64u + (((2u + (fkk >> 1)) ^ ((fr >> 2) & 3u)) * 16u) + (fkk & 1u) * 8u;

  const int64_t gstep = gridDim.x;
  int64_t gi = blockIdx.x;
#pragma unroll
  for (int s = 0; s < NBUF - 1; ++s) dma_group(gi + s * gstep, s, gi + s * gstep < ng);
  __builtin_amdgcn_s_waitcnt(0x0070);
  asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");

  // ---- the three parts of a group's work (m: its index in this workgroup's sequence, g: its global index) ----
  // MFMAs of k-steps [K0, K1) of group m into ac; with DMA: the pieces of the group `ahead` further on, one every 8 k-steps
  // (issued together behind a barrier, the 25 pieces of a group queue at the CU's one address unit and every wave stands
  // in its load issue with nobody feeding the matrix pipe)
  auto mfma_range = [&](f64x4s (&ac)[NA], int m, int64_t g, int K0, int K1, int ahead) {
    if (MODE & 2) return;
    const int slot = m & (NBUF - 1);
    const TF_LDS_AS char *const xb0 = lds + slot * GB + aoff0, *const xb1 = lds + slot * GB + aoff1;
    auto frag = [&](int ks) { return *(const TF_LDS_AS double *)(((ks & 1) ? xb1 : xb0) + (ks >> 1) * 1024); };
    double a0 = frag(K0), a1 = frag(K0 + 1);                  // fragments two k-steps ahead of their MFMAs
    double t0 = 0.0, t1 = 0.0;                                // ... and the LDS-resident B fragments likewise
    if (LK > 0 && K1 > KR) { t0 = tl[(max(K0, KR) - KR) * 64]; t1 = tl[(max(K0, KR) + 1 - KR) * 64]; }
#pragma unroll
    for (int ks = K0; ks < K1; ++ks) {
      const double a = a0;
      a0 = a1;
      if (ks + 2 < K1) a1 = frag(ks + 2);
#pragma unroll
      for (int f = 0; f < FTW; ++f) {
        if (LK > 0 && f == FTW - 1 && ks >= KR) {
          const double bb = t0;
          t0 = t1;
          if (ks + 2 < KSTEPS) t1 = tl[(ks + 2 - KR) * 64];
          ac[f] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, ac[f], 0, 0, 0);
        } else {
          ac[f] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, tb[f][ks], ac[f], 0, 0, 0);
        }
      }
      if (ahead > 0 && ks % 8 == 2 && ks / 8 < PW)
        dma_piece(g + ahead * gstep, (m + ahead) & (NBUF - 1), g + ahead * gstep < ng, ks / 8);
      if (SPL && ks >= KSTEPS - QS) {                          // the split tile's k-steps ride along with the last QS steps
        const int j = ks - (KSTEPS - QS);
        const int kq = q0 + min(j, max(qvalid - 1, 0));        // (past the share's end: a valid fragment against zeros)
        const double aq = *(const TF_LDS_AS double *)(((kq & 1) ? xb1 : xb0) + (unsigned)((kq >> 1) * 1024));
        ac[FTW] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, tq[j], ac[FTW], 0, 0, 0);
      }
    }
  };
  // epilogue, part 1 (in front of the group's barrier): the split tile's partial sums and this wave's part of the rows'
  // weighted squares -> LDS.  Accumulator layout: column = lane & 15 of the tile, row = (lane >> 4) + 4 * reg of the row group.
  auto epi1 = [&](f64x4s (&ac)[NA], int m) {
    TF_LDS_AS char *const scr = lds + NBUF * GB + (m & 1) * SCR;
    if (SPL) ((TF_LDS_AS f64x4s *)scr)[simd * 64 + lane] = ac[FTW];
    double pz[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double z = 0.0;
#pragma unroll
      for (int f = 0; f < FTW; ++f) {
        const double v = ac[f][r] + offs[f];
        ac[f][r] = v;
        z = fma(v * v, wts[f], z);
      }
      pz[r] = z;
    }
    const double ysum = row_sum4_by_class(pz, lane);          // lane: the sum for reg (lane & 3) of its fk
    ((TF_LDS_AS double *)(scr + 4 * 64 * 32))[fi < 4 ? wave * 16 + fk + 4 * fi : 128 + wave * 64 + lane] = ysum;
  };
  // part 2 (behind the barrier): totals, normalisation, output
  auto epi2 = [&](f64x4s (&ac)[NA], int m, int64_t g) {
    TF_LDS_AS char *const scr = lds + NBUF * GB + (m & 1) * SCR;
    const TF_LDS_AS f64x4s *const P = (const TF_LDS_AS f64x4s *)scr;
    const TF_LDS_AS double *const red = (const TF_LDS_AS double *)(scr + 4 * 64 * 32);
    double vq[4] = {0.0, 0.0, 0.0, 0.0}, tq4[4];
    {
      const f64x4s sq = (P[0 * 64 + lane] + P[1 * 64 + lane]) + (P[2 * 64 + lane] + P[3 * 64 + lane]);
      double pq[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { vq[r] = sq[r] + offq; pq[r] = vq[r] * vq[r] * wq; }
      const double yq = row_sum4_by_class(pq, lane);           // class lane & 3 -> every lane needs all four: quad broadcasts
      tq4[0] = dpp_f64<0x00>(yq); tq4[1] = dpp_f64<0x55>(yq); tq4[2] = dpp_f64<0xAA>(yq); tq4[3] = dpp_f64<0xFF>(yq);
    }
    double fq = 0.0, vsel = 0.0;
    unsigned qrow = 0xfffffff0u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = fk + 4 * r;
      double tot = tq4[r];
#pragma unroll
      for (int w = 0; w < 8; w += 2) tot += red[w * 16 + rr] + red[(w + 1) * 16 + rr];
      // sqrt(Dout / tot) = sqrt(Dout) * rsqrt(tot): hardware estimate + two Newton steps (full precision)
      double y = __builtin_amdgcn_rsq(tot);
      y = y * fma(-0.5 * tot * y, y, 1.5);
      y = y * fma(-0.5 * tot * y, y, 1.5);
      const double fnorm = sqrt_dout * y;
      // stores through a descriptor over `out` (R Dout 8 bytes): a row past R is beyond its range and dropped; no branch
      const unsigned rowoff = (unsigned)((int)(g * 16 + rr) * Dout) * 8u;
#pragma unroll
      for (int f = 0; f < FTW; ++f) {
        const unsigned col = (unsigned)((simd + 4 * f + TB) * 16 + fi);
        const double val = fnorm * ac[f][r];
        if (!(MODE & 4) || val == 123.456)
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2s, val), rsO, (int)(((int)col < Dout) ? rowoff + col * 8u : 0xfffffff0u), 0, 0);
      }
      if (SPL && simd == r) { fq = fnorm; vsel = vq[r]; qrow = rowoff; }
    }
    if (SPL) {                                                 // the split tile's four regs: one per SIMD
      const unsigned col = (unsigned)((NT - 1) * 16 + fi);
      const double val = fq * vsel;
      if (!(MODE & 4) || val == 123.456)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2s, val), rsO, (int)(((int)col < Dout) ? qrow + col * 8u : 0xfffffff0u), 0, 0);
    }
  };
  // MODE bit 3: shader-clock stamps of workgroup 0 ([group 8 .. 23][wave][8]: the phases of a step, see the two loops)
  auto stamp = [&](int m, int k) {
    if (!(MODE & 8)) return;
    const unsigned long long ts = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && lane == 0 && m >= 8 && m < 24) A.dbg[((m - 8) * 8 + wave) * 8 + k] = ts;
  };
  auto zero = [&](f64x4s (&ac)[NA]) {
#pragma unroll
    for (int f = 0; f < NA; ++f) ac[f] = f64x4s{0.0, 0.0, 0.0, 0.0};
  };

  // ---- the two waves of a SIMD run the parts in DIFFERENT orders around the group's one barrier B(m), so that one's
  //      epilogue falls under the other's MFMAs:
  //        two-tile wave:   MFMA(m)  epi1(m)  B(m)  epi2(m)            | MFMA(m+1) ...
  //        tile + share:    ...  B(m)  MFMA(m+1)[k-steps < KSPLIT]  epi2(m)  MFMA(m+1)[rest]  epi1(m+1)  B(m+1) ...
  //      Phase stamps (PLDA_TRANSFORM_VARIANT=14, scripts/transform_timeline.py, profiles/r04_transform_treg_timeline.txt):
  //      a step is 15 400 cycles whatever KSPLIT is, against 10 400 of MFMAs.  With KSPLIT = 16 the two-tile wave's 100
  //      MFMAs run at their full rate (66 cycles each) while the other wave's epilogue part 2 beside them takes 8 700
  //      cycles instead of the 1 600 - 2 900 it takes alone -- fp64 MFMAs and the other wave's vector instructions do
  //      share the SIMD, the MFMAs win and the vector stream gets about a third of its speed -- and that wave's remaining
  //      47 MFMAs then run alone behind it; with KSPLIT = 50 the two waves' MFMAs interleave (163 in 12 500 cycles: 77
  //      each, the tile + share wave being ONE dependent accumulator chain) and both epilogues are exposed.  Either way
  //      10 400 + ~5 000: on this part the fp64 matrix rate equals the fp64 vector rate, and a transform whose epilogue
  //      is ~7 000 vector-unit cycles per SIMD and row group cannot hide it -- it has to get shorter (section 8).
  //      The DMA pieces of a group are shared by all eight waves; a two-tile wave issues its pieces of group m + 3 under
  //      MFMA(m), a tile + share wave its pieces of group m + 2 under MFMA(m) -- both in front of B(m), into the slot all
  //      waves left at B(m - 1) / B(m - 2).
  constexpr int KSPLIT = KSTEPS;
  if (!SPL) {
    for (int m = 0;; ++m) {
      f64x4s acc[NA];
      zero(acc);
      stamp(m, 0);
      mfma_range(acc, m, gi, 0, KSTEPS, NBUF - 1);
      stamp(m, 1);
      epi1(acc, m);
      stamp(m, 2);
      __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0): the operands requested in this step are in LDS
      stamp(m, 3);
      asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
      stamp(m, 4);
      epi2(acc, m, gi);
      stamp(m, 5);
      gi += gstep;
      if (gi >= ng) break;
    }
  } else {
    f64x4s acc[NA];
    zero(acc);
    mfma_range(acc, 0, gi, 0, KSTEPS, 0);     // (group 2's pieces came with the prologue)
    epi1(acc, 0);
    for (int m = 0;; ++m) {
      stamp(m, 6);
      __builtin_amdgcn_s_waitcnt(0x0070);
      stamp(m, 7);
      asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");           // B(m)
      stamp(m, 0);
      const int64_t gn = gi + gstep;
      if (gn >= ng) { epi2(acc, m, gi); break; }
      f64x4s accn[NA];
      zero(accn);
      mfma_range(accn, m + 1, gn, 0, KSPLIT, NBUF - 2);
      stamp(m, 1);
      epi2(acc, m, gi);
      stamp(m, 2);
      mfma_range(accn, m + 1, gn, KSPLIT, KSTEPS, NBUF - 2);
      stamp(m, 3);
#pragma unroll
      for (int f = 0; f < NA; ++f) acc[f] = accn[f];
      epi1(acc, m + 1);
      stamp(m, 4);
      gi = gn;
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0070);
}

// MODE (timing arms, garbage results): 1 no X DMA, 2 no MFMAs, 4 no output stores
template <int NT, int KSTEPS, int MODE = 0>
__global__ __launch_bounds__(512) void transform_treg_kernel(
    const double *__restrict__ X, int64_t R, int Din, const double *__restrict__ Tpad, int Dinp, int Dout,
    const double *__restrict__ offset, const double *__restrict__ psi, int n_uniform, double *__restrict__ out, unsigned long long *__restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) double tf_lds[];
  TF_LDS_AS char *const lds = (TF_LDS_AS char *)tf_lds;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if ((int64_t)blockIdx.x >= ((R + 15) >> 4)) return;
  const TregArgs A{X, R, Din, Tpad, Dinp, Dout, offset, psi, n_uniform, out, dbg};
  // wave w runs on SIMD w & 3 (waves of a workgroup are dealt to the SIMDs round robin): waves 0-3 take the two-tile
  // role, waves 4-7 the tile + split-share role of the same SIMD
  if (wave < 4) treg_wave<NT, KSTEPS, MODE, 2, 0, false>(A, lds, lane, wave, wave);
  else treg_wave<NT, KSTEPS, MODE, 1, 8, true>(A, lds, lane, wave, wave - 4);
}

template <int NT, int KSTEPS, int MODE = 0>
static int launch_transform_treg(plda_handle *h, const double *dX, int64_t R, int Din, int n_uniform, double *dout, int Dinp) {
  using G = TregGeom<NT, KSTEPS>;
  static_assert(G::LDS_BYTES <= 160 * 1024, "batch buffers exceed the LDS of a CU");
  if (MODE & 8) PLDA_HIP(h, h->timeline.reserve((size_t)8 * 16 * 8 * 8 * 8));
  static DeviceOnce attr;
  if (attr.needed(h->device)) {
    PLDA_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&transform_treg_kernel<NT, KSTEPS, MODE>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
    attr.done(h->device);
  }
  transform_treg_kernel<NT, KSTEPS, MODE><<<(unsigned)std::min<int64_t>(ceil_div(R, (int64_t)16), h->num_cus), 512, G::LDS_BYTES, h->stream>>>(
      dX, R, Din, h->tf_pad.as<double>(), Dinp, h->Dout, h->d_offset.as<double>(), h->d_psi.as<double>(), n_uniform, dout,
      (MODE & 8) ? h->timeline.as<unsigned long long>() : nullptr);
  PLDA_LAUNCH_CHECK(h);
  if (MODE & 8) h->timeline_valid = true;
  return PLDA_OK;
}

__global__ void pad_transform_kernel(const double *__restrict__ T, int Dout, int Din, double *__restrict__ Tpad, int rows,
                                     int Dinp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * Dinp) return;
  const int r = idx / Dinp, c = idx % Dinp;
  Tpad[idx] = (r < Dout && c < Din) ? T[(int64_t)r * Din + c] : 0.0;
}

__global__ void pad_matrix_kernel(const double *__restrict__ A, int lda, int rows_in, int cols_in, double *__restrict__ P, int rows,
                                  int Dinp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * Dinp) return;
  const int r = idx / Dinp, c = idx % Dinp;
  P[idx] = (r < rows_in && c < cols_in) ? A[(int64_t)r * lda + c] : 0.0;
}

template <int NT, int CH, int KS, bool PERROW, int RT = 1>
static int launch_transform_fused_t(plda_handle *h, const double *dX, int64_t R, int Din, const int32_t *dn,
                                    int n_uniform, double *dout, int Dinp) {
  using G = TfGeom<NT, CH, KS, RT>;
  static_assert(KS % 4 == 0 && KS >= 8, "a stage is a whole number of 4-k MFMA steps, and at least two of them");
  static_assert(G::LDS_BYTES <= 160 * 1024, "stage buffers exceed the LDS of a CU");
  static_assert((size_t)(2 * G::COLS + CH * G::ROWS) * 8 <= G::LDS_BYTES / 2, "the epilogue's scratch must fit one stage buffer");
  static DeviceOnce attr;          // (per instantiation and device; setting it twice is harmless)
  if (attr.needed(h->device)) {
    PLDA_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&transform_fused_kernel<NT, CH, KS, PERROW, RT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
    attr.done(h->device);
  }
  transform_fused_kernel<NT, CH, KS, PERROW, RT><<<(unsigned)std::min<int64_t>(ceil_div(R, (int64_t)G::ROWS), h->num_cus), 512,
                                               G::LDS_BYTES, h->stream>>>(
      dX, R, Din, h->tf_pad.as<double>(), Dinp, h->Dout, h->d_offset.as<double>(), h->d_psi.as<double>(), dn,
      n_uniform, dout);
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

template <int NT, int CH, bool PERROW>
static int launch_transform_dma_t(plda_handle *h, const double *dX, int64_t R, int Din, const int32_t *dn,
                                  int n_uniform, double *dout, int Dinp, int padrows) {
  using G = TfDmaGeom<NT, CH>;
  static_assert(G::LDS_BYTES <= 160 * 1024, "stage ring exceeds the LDS of a CU");
  static DeviceOnce attr;
  if (attr.needed(h->device)) {
    PLDA_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&transform_dma_kernel<NT, CH, PERROW>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
    attr.done(h->device);
  }
  transform_dma_kernel<NT, CH, PERROW><<<(unsigned)std::min<int64_t>(ceil_div(R, (int64_t)G::ROWS), h->num_cus), 512,
                                         G::LDS_BYTES, h->stream>>>(
      dX, R, Din, h->tf_pad.as<double>(), Dinp, padrows, h->Dout, h->d_offset.as<double>(), h->d_psi.as<double>(), dn,
      n_uniform, dout);
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

// (the per-row-count epilogue is its own instantiation: as a run-time branch beside the uniform one it made every large
// block shape spill, 92-372 bytes per lane)
template <int NT, int CH, int KS, int RT = 1>
static int launch_transform_fused(plda_handle *h, const double *dX, int64_t R, int Din, const int32_t *dn,
                                  int n_uniform, double *dout, int Dinp) {
  if (RT > 1)
    return dn ? launch_transform_fused_t<NT, CH, KS, true, RT>(h, dX, R, Din, dn, n_uniform, dout, Dinp)
              : launch_transform_fused_t<NT, CH, KS, false, RT>(h, dX, R, Din, dn, n_uniform, dout, Dinp);
  // PLDA_TRANSFORM_VARIANT=7: the DMA-staged kernel (A/B arm; measured 2-6 % behind the register-staged one)
  if (KS == 16 && h->transform_variant == 7)
    return dn ? launch_transform_dma_t<NT, CH, true>(h, dX, R, Din, dn, n_uniform, dout, Dinp, h->tf_pad_rows)
              : launch_transform_dma_t<NT, CH, false>(h, dX, R, Din, dn, n_uniform, dout, Dinp, h->tf_pad_rows);
  return dn ? launch_transform_fused_t<NT, CH, KS, true>(h, dX, R, Din, dn, n_uniform, dout, Dinp)
            : launch_transform_fused_t<NT, CH, KS, false>(h, dX, R, Din, dn, n_uniform, dout, Dinp);
}

// QF launches (quadform_rows_device below): own padded matrix, uniform epilogue shape, RT = 1
template <int NT, int CH, int KS>
static int launch_quadform_t(plda_handle *h, const double *dX, int64_t R, int D, const double *Cpad, int Dinp, const double *lin,
                             const double *q, const TfQuad &qf, double *out) {
  using G = TfGeom<NT, CH, KS, 1>;
  static_assert(G::LDS_BYTES <= 160 * 1024, "stage buffers exceed the LDS of a CU");
  static_assert((size_t)(3 * G::COLS + 2 * CH * G::ROWS) * 8 <= G::LDS_BYTES / 2, "the epilogue's scratch must fit one stage buffer");
  static DeviceOnce attr;
  if (attr.needed(h->device)) {
    PLDA_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&transform_fused_kernel<NT, CH, KS, false, 1, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
    attr.done(h->device);
  }
  transform_fused_kernel<NT, CH, KS, false, 1, true><<<(unsigned)std::min<int64_t>(ceil_div(R, (int64_t)G::ROWS), h->num_cus), 512,
                                                      G::LDS_BYTES, h->stream>>>(dX, R, D, Cpad, Dinp, D, lin, q, nullptr, 1, out, qf);
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

template <int A, int B> constexpr int cmax() { return A > B ? A : B; }

// the instantiations of one dimension class: the main block shape <NT0, CH0> (128 rows for CH0 = 1, 64 for CH0 = 2) and
// the smaller tail blocks <NT1, 2> (64 rows; CH0 = 1 only), <NT2, 4> (32 rows), <NT3, 8> (16 rows); KS = stage depth
// <NTW, CHW>: round 4's A/B arm (PLDA_TRANSFORM_VARIANT=9) -- two row tiles per wave (RT = 2), CHW column slices of NTW
// tiles: 9 fragment reads per 14 MFMAs at D = 200 instead of 14 per 13, the same 128-row block.  Measured SLOWER than the
// one-row-tile shape (C2 0.216 against 0.200 ms = 0.47 / 0.51 of the fp64 peak, C4 0.737 / 0.752, interleaved,
// gpurun_out/r4/k4_sweep.log): the LDS fragment traffic is not what bounds this kernel either.  Not the product path.
template <int KS, int NT0, int CH0, int NT1, int NT2, int NT3, int NTW, int CHW>
static int transform_class(plda_handle *h, const double *dX, int64_t R, int Din, const int32_t *dn, int n_uniform,
                           double *dout) {
  // the zero-padded copy of T ([rows >= every block shape's stage rows][Din rounded up to KS]), rebuilt only when the
  // model has changed (or another class's geometry was cached)
  constexpr int TRW = NTW > 0 ? TfGeom<(NTW > 0 ? NTW : 1), (NTW > 0 ? CHW : 1), KS, 2>::TR : 0;
  constexpr int PADROWS = cmax<cmax<cmax<TfGeom<NT0, CH0, KS>::TR, TRW>(), TfGeom<NT1, 2, KS>::TR>(),
                               cmax<TfGeom<NT2, 4, KS>::TR, TfGeom<NT3, 8, KS>::TR>()>();
  const int Dinp = (int)round_up(Din, KS);
  if (h->tf_pad_epoch != h->model_epoch || h->tf_pad_rows != PADROWS || h->tf_pad_dinp != Dinp) {
    PLDA_HIP(h, h->tf_pad.reserve((size_t)PADROWS * Dinp * 8));
    pad_transform_kernel<<<(unsigned)ceil_div((int64_t)PADROWS * Dinp, 256), 256, 0, h->stream>>>(
        h->d_transform.as<double>(), h->Dout, Din, h->tf_pad.as<double>(), PADROWS, Dinp);
    PLDA_LAUNCH_CHECK(h);
    h->tf_pad_epoch = h->model_epoch; h->tf_pad_rows = PADROWS; h->tf_pad_dinp = Dinp;
  }
  // round 4, A/B arm (PLDA_TRANSFORM_VARIANT=6; 10-13: its timing arms): T resident in registers, X streamed
  // (transform_treg_kernel) -- the C2 shape: 13 column tiles, Din = 200, a uniform count, at least eight row groups per
  // CU, 32-bit byte offsets and 16-byte aligned rows for the DMA.  Measured level with the kernels below, not ahead of
  // them (C2 0.21 against 0.19 ms, 800k rows 1.26 against 1.24 ms): not the product path.
  if constexpr (NT0 == 13 && CH0 == 1) {
    const int tv = h->transform_variant;
    if (!dn && (tv == 6 || (tv >= 10 && tv <= 14)) && Din == 200 && h->Dout > 192 && R >= (int64_t)128 * h->num_cus &&
        (reinterpret_cast<uintptr_t>(dX) & 15) == 0 && R * (int64_t)Din * 8 < ((int64_t)1 << 32) - (1 << 20)) {
      if (tv == 10) return launch_transform_treg<13, 50, 1>(h, dX, R, Din, n_uniform, dout, Dinp);   // timing arms
      if (tv == 11) return launch_transform_treg<13, 50, 2>(h, dX, R, Din, n_uniform, dout, Dinp);
      if (tv == 12) return launch_transform_treg<13, 50, 4>(h, dX, R, Din, n_uniform, dout, Dinp);
      if (tv == 13) return launch_transform_treg<13, 50, 7>(h, dX, R, Din, n_uniform, dout, Dinp);
      if (tv == 14) return launch_transform_treg<13, 50, 8>(h, dX, R, Din, n_uniform, dout, Dinp);   // phase stamps of workgroup 0
      return launch_transform_treg<13, 50>(h, dX, R, Din, n_uniform, dout, Dinp);
    }
  }
  // (NTW = 0: no wide shape for this class -- above D = 256 two row tiles per wave spill; per-row counts with 8 tiles per
  //  slice spill 68 bytes per lane: the round-3 shape there)
  const bool wide = NTW > 0 && !(dn && NTW >= 8) && h->transform_variant == 9;
  const int ROWS0 = wide ? 32 * (8 / (CHW > 0 ? CHW : 1)) : 16 * (8 / CH0);
  const int64_t G = h->num_cus;
  // main launch: a whole number of rounds of the persistent grid (PLDA_TRANSFORM_VARIANT=2: everything, as in round 2)
  const int64_t nb = ceil_div(R, (int64_t)ROWS0);
  const int64_t rows_main = h->transform_variant == 2 ? R : std::min(R, nb / G * G * ROWS0);
  if (rows_main > 0) {
    bool done = false;
    if constexpr (NTW > 0) {
      if (wide) { PLDA_TRY((launch_transform_fused<NTW, CHW, KS, 2>(h, dX, rows_main, Din, dn, n_uniform, dout, Dinp))); done = true; }
    }
    if (!done) PLDA_TRY((launch_transform_fused<NT0, CH0, KS>(h, dX, rows_main, Din, dn, n_uniform, dout, Dinp)));
  }
  const int64_t Rt = R - rows_main;
  if (Rt <= 0) return PLDA_OK;
  // the rest: the smallest blocks that still give every CU at most one
  const double *tX = dX + rows_main * Din;
  const int32_t *tn = dn ? dn + rows_main : nullptr;
  double *to = dout + rows_main * (int64_t)h->Dout;
  const int64_t per_cu = ceil_div(Rt, G);
  if (per_cu <= 16) return launch_transform_fused<NT3, 8, KS>(h, tX, Rt, Din, tn, n_uniform, to, Dinp);
  if (per_cu <= 32) return launch_transform_fused<NT2, 4, KS>(h, tX, Rt, Din, tn, n_uniform, to, Dinp);
  if constexpr (CH0 == 1) {
    if (per_cu <= 64) return launch_transform_fused<NT1, 2, KS>(h, tX, Rt, Din, tn, n_uniform, to, Dinp);
  }
  return launch_transform_fused<NT0, CH0, KS>(h, tX, Rt, Din, tn, n_uniform, to, Dinp);
}

int transform_rows_device(plda_handle *h, const double *dX, int64_t R, int Din, const int32_t *dn,
                          int n_uniform, double *dout) {
  if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "transform: model not fitted");
  if (Din != h->Din) return fail(h, PLDA_E_INVAL, "transform: feature dim %d != model dim %d", Din, h->Din);
  if (R <= 0) return PLDA_OK;
  // out[r][o] = sum_k X[r][k] T[o][k]
  TraceScope ts(h, "transform.gemm + length_norm (K4)", 2.0 * (double)R * h->Dout * Din, 1);
  if (h->Dout <= 512 && h->transform_variant != 1 && R < ((int64_t)1 << 31) * 64) {
    const int D = h->Dout;
    // <stage depth; tiles per wave of the main block shape and its column slices; tiles per wave of the 2 / 4 / 8-slice
    // tail blocks>, NT * CH * 16 >= D in every shape.  Stage depth: 16 k everywhere -- deeper stages (20 ... 32 k, as
    // deep as the LDS allows per class) were measured and are no faster (C2 0.519 against 0.527 of the fp64 peak,
    // C4 0.728 against 0.757), so the 2.5 us a stage's data movement takes is not a latency a longer stage amortises.
    if (D <= 128) return transform_class<16, 8, 1, 4, 2, 1, 4, 2>(h, dX, R, Din, dn, n_uniform, dout);
    if (D <= 208) return transform_class<16, 13, 1, 7, 4, 2, 7, 2>(h, dX, R, Din, dn, n_uniform, dout);
    if (D <= 256) return transform_class<16, 16, 1, 8, 4, 2, 8, 2>(h, dX, R, Din, dn, n_uniform, dout);
    if (D <= 384) return transform_class<16, 12, 2, 12, 6, 3, 0, 0>(h, dX, R, Din, dn, n_uniform, dout);
    return transform_class<16, 16, 2, 16, 8, 4, 0, 0>(h, dX, R, Din, dn, n_uniform, dout);
  }
  PLDA_TRY(gemm_f64(h, R, h->Dout, Din, 1.0, dX, Din, 1, h->d_transform.as<double>(), 1, Din,
                    nullptr, 0.0, dout, h->Dout));
  const int wpb = 4;
  length_norm_kernel<<<(unsigned)ceil_div(R, wpb), wpb * 64, 0, h->stream>>>(
      dout, R, h->Dout, h->d_offset.as<double>(), h->d_psi.as<double>(), dn, n_uniform);
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

// norm()'s model pass on K4's kernel shape (round 6): for every row x of dX [R, D]
//     out_mean = sum_c x_c (m_c - q_c x_c / 2) + *mD,      out_std = sqrt(max(x^T C x + lin . x + *crr, 0))
// C [D, D] symmetric with leading dimension ldc.  D <= 208 (*used = false otherwise: the caller keeps its GEMM).  The general
// GEMM + row kernel this replaces ran the 50k x 200 x 200 product of C5 at 0.25 of the fp64 MFMA peak.
int quadform_rows_device(plda_handle *h, const double *dX, int64_t R, int D, const double *C, int ldc, const double *lin,
                         const double *m, const double *q, const double *mD, const double *crr, double *out_mean,
                         double *out_std, bool *used) {
  *used = false;
  if (D > 208 || R <= 0) return PLDA_OK;
  constexpr int KS = 16;
  const int Dinp = (int)round_up(D, KS);
  // rows of the padded matrix: the largest stage of the shapes below (the main shape's)
  const int padrows = D <= 128 ? cmax<TfGeom<8, 1, KS>::TR, cmax<TfGeom<4, 2, KS>::TR, cmax<TfGeom<2, 4, KS>::TR, TfGeom<1, 8, KS>::TR>()>()>()
                               : cmax<TfGeom<13, 1, KS>::TR, cmax<TfGeom<7, 2, KS>::TR, cmax<TfGeom<4, 4, KS>::TR, TfGeom<2, 8, KS>::TR>()>()>();
  PLDA_HIP(h, h->zn_cpad.reserve((size_t)padrows * Dinp * 8));
  double *Cpad = h->zn_cpad.as<double>();
  pad_matrix_kernel<<<(unsigned)ceil_div((int64_t)padrows * Dinp, 256), 256, 0, h->stream>>>(C, ldc, D, D, Cpad, padrows, Dinp);
  PLDA_LAUNCH_CHECK(h);
  const TfQuad qf{m, mD, crr, out_std};
  const int64_t G = h->num_cus;
  const int64_t nb = ceil_div(R, (int64_t)128);
  const int64_t rows_main = std::min(R, nb / G * G * 128);
  const bool small = D <= 128;
  if (rows_main > 0)
    PLDA_TRY(small ? (launch_quadform_t<8, 1, KS>(h, dX, rows_main, D, Cpad, Dinp, lin, q, qf, out_mean))
                   : (launch_quadform_t<13, 1, KS>(h, dX, rows_main, D, Cpad, Dinp, lin, q, qf, out_mean)));
  const int64_t Rt = R - rows_main;
  *used = true;
  if (Rt <= 0) return PLDA_OK;
  const double *tX = dX + rows_main * D;
  const TfQuad qt{m, mD, crr, out_std + rows_main};
  double *to = out_mean + rows_main;
  const int64_t per_cu = ceil_div(Rt, G);
  if (small) {
    if (per_cu <= 16) return launch_quadform_t<1, 8, KS>(h, tX, Rt, D, Cpad, Dinp, lin, q, qt, to);
    if (per_cu <= 32) return launch_quadform_t<2, 4, KS>(h, tX, Rt, D, Cpad, Dinp, lin, q, qt, to);
    if (per_cu <= 64) return launch_quadform_t<4, 2, KS>(h, tX, Rt, D, Cpad, Dinp, lin, q, qt, to);
    return launch_quadform_t<8, 1, KS>(h, tX, Rt, D, Cpad, Dinp, lin, q, qt, to);
  }
  if (per_cu <= 16) return launch_quadform_t<2, 8, KS>(h, tX, Rt, D, Cpad, Dinp, lin, q, qt, to);
  if (per_cu <= 32) return launch_quadform_t<4, 4, KS>(h, tX, Rt, D, Cpad, Dinp, lin, q, qt, to);
  if (per_cu <= 64) return launch_quadform_t<7, 2, KS>(h, tX, Rt, D, Cpad, Dinp, lin, q, qt, to);
  return launch_quadform_t<13, 1, KS>(h, tX, Rt, D, Cpad, Dinp, lin, q, qt, to);
}

}  // namespace plda
