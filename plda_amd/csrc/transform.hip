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
template <int NT, int CH, int KS, bool PERROW, int RT = 1>
__global__ __launch_bounds__(512) void transform_fused_kernel(const double *__restrict__ X, int64_t R, int Din,
                                                              const double *__restrict__ Tpad, int Dinp, int Dout,
                                                              const double *__restrict__ offset,
                                                              const double *__restrict__ psi,
                                                              const int32_t *__restrict__ n_arr, int n_uniform,
                                                              double *__restrict__ out) {
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
      ep[c] = PERROW ? ps : tf_rcp(ps + inv_nu);
    }
    __syncthreads();
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
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) part[rt][r] += __shfl_xor(part[rt][r], o);
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
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double f = sqrt((double)Dout / part[rt][r]);
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
      __builtin_amdgcn_s_barrier();
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
      __builtin_amdgcn_s_barrier();
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

__global__ void pad_transform_kernel(const double *__restrict__ T, int Dout, int Din, double *__restrict__ Tpad, int rows,
                                     int Dinp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * Dinp) return;
  const int r = idx / Dinp, c = idx % Dinp;
  Tpad[idx] = (r < Dout && c < Din) ? T[(int64_t)r * Din + c] : 0.0;
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

}  // namespace plda
