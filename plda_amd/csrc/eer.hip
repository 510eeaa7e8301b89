// plda_amd/csrc/eer.hip -- equal error rate of a trials matrix on the GPU (SURVEY.md section 8f
// rank 4).  Replaces /root/reference/scoring/eer.py:68-73, which calls
// bob.measure.eer_threshold(negatives, positives) and bob.measure.farfrr(...): `bob` is an
// absent, un-pinned third-party dependency, so (as for Kaldi) its published definition is
// restated -- oracle/plda_oracle_np.py:eer -- and parity for this row is UNPINNED:
//   farfrr(neg, pos, t):  FAR = #{neg >= t} / Nn,  FRR = #{pos < t} / Np
//   eer_threshold:        candidate thresholds are the minimum score and the midpoints
//                         between consecutive distinct scores of the union; the one with
//                         minimal |FAR - FRR| wins, the later one on ties.
// Sort-free: g(k) = FRR - FAR "after score k" is non-decreasing in k, so the minimum of |g|
// is at the smallest key k1 with g(k1) >= 0 or at its predecessor.  k1 is found EXACTLY by
// three HBM-bound passes over the fp32 scores that histogram an order-preserving uint32 key
// 11 + 11 + 10 bits at a time (per-block LDS histograms, float4 loads, column strips), with the crossing
// located in integer arithmetic on the host between passes.  Algorithmic bytes: 3 x 4 B per trial read, nothing written.
//
// Round 5, large matrices (one process): ONE pass over the matrix instead of three.  A pilot -- the same exact
// refinement on every 32nd row (3 % of the bytes) -- brackets the crossing: the key range [klo, khi) in which the
// SAMPLE's FRR - FAR passes from -delta to +delta, delta = five standard errors of the sample's rates.  The one full pass
// then counts, per class, the trials below klo and appends the scores inside the window to two compact lists (a few per
// mille of the trials; wave-aggregated appends), and the exact refinement finishes on the lists with the counts below
// as its starting offsets.  The answer is the three-pass one bit for bit whenever the window holds the crossing key and
// both of its neighbours -- checked with the exact counts (g(klo) < 0 <= g(khi), neighbours found inside); otherwise, and
// for small or sharded inputs, the three passes run.  PLDA_EER_VARIANT=1 forces them (A/B arm, and what the test compares with).
#include "common.hpp"

#include <algorithm>
#include <cstring>

namespace plda {

__device__ __forceinline__ unsigned score_key(float f) {
  unsigned u = __float_as_uint(f);
  if (u == 0x80000000u) u = 0u;      // -0.0 == +0.0 as scores: one candidate threshold, not two
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // monotone: a < b  <=>  key(a) < key(b)
}
static inline float key_score(unsigned k) {
  const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

constexpr int EER_BINS = 2048;

// one pass: class c (0 = impostor, 1 = target) histogram of bits [shift, shift + nbits) of the
// keys whose higher bits equal `prefix` (pass 0: every key).  below/above track the largest
// key below and the smallest key above the prefix range (needed for neighbours in pass 2).
typedef float f32x4e __attribute__((ext_vector_type(4)));

// Flat list of scores of one class (the two-list form of scoring/eer.py).
__global__ __launch_bounds__(256) void eer_hist_kernel(const float *__restrict__ scores, int64_t Nt, int fixed_class,
                                                       int shift, int nbits, unsigned prefix, int has_prefix,
                                                       unsigned long long *__restrict__ hist /*[2][EER_BINS]*/,
                                                       unsigned *__restrict__ below, unsigned *__restrict__ above) {
  __shared__ unsigned lh[2][EER_BINS];
  for (int i = threadIdx.x; i < 2 * EER_BINS; i += 256) (&lh[0][0])[i] = 0;
  __syncthreads();
  const unsigned mask = (1u << nbits) - 1u;
  const int hi_shift = shift + nbits;
  unsigned lo_max = 0u, hi_min = 0xffffffffu;

  auto account = [&](bool ok, float sc, int cls) {
    unsigned bin = 0xffffffffu;   // inactive
    if (ok) {
      const unsigned k = score_key(sc);
      if (has_prefix && (k >> hi_shift) != prefix) {
        if ((k >> hi_shift) < prefix) lo_max = k > lo_max ? k : lo_max; else hi_min = k < hi_min ? k : hi_min;
      } else {
        bin = ((k >> shift) & mask) | ((unsigned)cls << 11);
      }
    }
    // plain non-returning LDS atomics: aggregating lanes with equal bins first (12 ballots per element) was slower
    if (bin != 0xffffffffu) atomicAdd(&lh[bin >> 11][bin & 2047u], 1u);
  };

  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx - threadIdx.x < Nt; idx += (int64_t)gridDim.x * 256)
    account(idx < Nt, idx < Nt ? scores[idx] : 0.f, fixed_class);
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * EER_BINS; i += 256) {
    const unsigned v = (&lh[0][0])[i];
    if (v) atomicAdd(hist + i, (unsigned long long)v);
  }
  if (has_prefix) {
    if (lo_max) atomicMax(below, lo_max);
    if (hi_min != 0xffffffffu) atomicMin(above, hi_min);
  }
}

// Matrix pass, strip layout: a workgroup owns EER_STRIP columns -- 4 per thread, whose speaker ids stay in
// registers -- and walks down a slice of the rows, so that the only per-trial memory traffic is the 4-byte score.
// (Workgroups walking along rows re-read the 8-byte speaker id of every column for every row: 12 B of cache
// traffic per trial, 39.4 ms for the three passes over 1e10 trials against 20.4 ms here, identical results:
// profiles/r02_eer_probe.json.)  Four rows are loaded before they are accounted for.
constexpr int EER_STRIP = 1024;
__global__ __launch_bounds__(256) void eer_hist_strip_kernel(const float *__restrict__ scores, int64_t ld, int64_t M,
                                                             int64_t Nt, const int64_t *__restrict__ espk,
                                                             const int64_t *__restrict__ tspk, int64_t rows_per_wg,
                                                             int shift, int nbits, unsigned prefix, int has_prefix,
                                                             unsigned long long *__restrict__ hist,
                                                             unsigned *__restrict__ below, unsigned *__restrict__ above, int64_t row_step) {
  // (row_step > 1: the pilot's sample -- "row" r stands for matrix row r * row_step; M counts sample rows)
  __shared__ unsigned lh[2][EER_BINS];
  for (int i = threadIdx.x; i < 2 * EER_BINS; i += 256) (&lh[0][0])[i] = 0;
  __syncthreads();
  const unsigned mask = (1u << nbits) - 1u;
  const int hi_shift = shift + nbits;
  unsigned lo_max = 0u, hi_min = 0xffffffffu;
  const int64_t strips = (Nt + EER_STRIP - 1) / EER_STRIP;
  const int64_t strip = blockIdx.x % strips, slice = blockIdx.x / strips;
  const int64_t col = strip * EER_STRIP + (int64_t)threadIdx.x * 4;
  const int64_t r0 = slice * rows_per_wg, r1 = (r0 + rows_per_wg < M) ? r0 + rows_per_wg : M;
  int64_t ts[4];
  bool ok[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ok[e] = col + e < Nt;
    ts[e] = ok[e] ? tspk[col + e] : 0;
  }
  const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(scores) & 15) == 0) && ok[3];

  auto account = [&](float sc, int cls) {
    const unsigned k = score_key(sc);
    if (has_prefix && (k >> hi_shift) != prefix) {
      if ((k >> hi_shift) < prefix) lo_max = k > lo_max ? k : lo_max; else hi_min = k < hi_min ? k : hi_min;
    } else {
      atomicAdd(&lh[cls][(k >> shift) & mask], 1u);
    }
  };

  for (int64_t row = r0; row < r1; row += 4) {
    float v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (row + u >= r1) break;
      const float *src = scores + (row + u) * row_step * ld + col;
      if (vec) {
        const f32x4e x = __builtin_nontemporal_load(reinterpret_cast<const f32x4e *>(src));
        v[u][0] = x.x; v[u][1] = x.y; v[u][2] = x.z; v[u][3] = x.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[u][e] = ok[e] ? src[e] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (row + u >= r1) break;
      const int64_t spk = espk[(row + u) * row_step];
#pragma unroll
      for (int e = 0; e < 4; ++e) if (ok[e]) account(v[u][e], ts[e] == spk ? 1 : 0);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * EER_BINS; i += 256) {
    const unsigned c = (&lh[0][0])[i];
    if (c) atomicAdd(hist + i, (unsigned long long)c);
  }
  if (has_prefix) {
    if (lo_max) atomicMax(below, lo_max);
    if (hi_min != 0xffffffffu) atomicMin(above, hi_min);
  }
}

// The one full pass of the windowed form (same strip layout).  Per trial: three float compares against the window's
// ends (the key order is the float order, and -0 == +0 in both), the class, four predicated counters -- below the window
// per class, targets, at-or-above the window -- and, for the few per mille inside it, a slot in the wave's LDS stage
// (an LDS atomic taken by those lanes only), which leaves for the class's list 512+ scores at a time behind ONE global
// atomic (one global atomic per wave and element serialises on the cursor's line: 0.8 s for a window of a tenth of 1e10
// trials, measured).  NaN scores satisfy no compare: the caller sees below + inside + above != M Nt and runs the three
// passes, whose key order places them.  A list that fills up keeps counting (the cursor says by how much it overflowed).
struct EerWindowOut {
  unsigned long long below[2], targets, above, cursor[2];
};
constexpr int EER_WBUF = 1536;       // staged scores per wave and class (48 KB per workgroup); flushed when a batch (<= 1024 more) might not fit
__global__ __launch_bounds__(256) void eer_window_strip_kernel(const float *__restrict__ scores, int64_t ld, int64_t M, int64_t Nt,
                                                               const int64_t *__restrict__ espk, const int64_t *__restrict__ tspk,
                                                               int64_t rows_per_wg, float flo, float fhi, EerWindowOut *__restrict__ wo,
                                                               float *__restrict__ list0, float *__restrict__ list1, unsigned long long cap) {
  __shared__ unsigned long long red[4][4];
  __shared__ float stage[4][2][EER_WBUF];
  __shared__ int fillc[4][2];
  const int64_t strips = (Nt + EER_STRIP - 1) / EER_STRIP;
  const int64_t strip = blockIdx.x % strips, slice = blockIdx.x / strips;
  const int64_t col = strip * EER_STRIP + (int64_t)threadIdx.x * 4;
  const int64_t r0 = slice * rows_per_wg, r1 = (r0 + rows_per_wg < M) ? r0 + rows_per_wg : M;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t ts[4];
  bool ok[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ok[e] = col + e < Nt;
    ts[e] = ok[e] ? tspk[col + e] : 0;
  }
  const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(scores) & 15) == 0) && ok[3];
  const float qnan = __uint_as_float(0x7fc00000u);
  unsigned b0 = 0, b1 = 0, t1 = 0, ab = 0;      // (a thread sees <= 4 * rows_per_wg trials: 32 bits)
  int *const fc = fillc[wave];
  if (lane < 2) fc[lane] = 0;
  auto flush = [&](int c) {
    const int n = __builtin_amdgcn_readfirstlane(fc[c]);
    if (n == 0) return;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&wo->cursor[c], (unsigned long long)n);
    base = __shfl(base, 0);
    float *dst = c ? list1 : list0;
    const float *src = stage[wave][c];
    for (int i = lane; i < n; i += 64)
      if (base + i < cap) dst[base + i] = src[i];
    if (lane == 0) fc[c] = 0;
  };
  for (int64_t row = r0; row < r1; row += 4) {
    float v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (row + u >= r1) break;
      const float *src = scores + (row + u) * ld + col;
      if (vec) {
        const f32x4e x = __builtin_nontemporal_load(reinterpret_cast<const f32x4e *>(src));
        v[u][0] = x.x; v[u][1] = x.y; v[u][2] = x.z; v[u][3] = x.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[u][e] = ok[e] ? src[e] : qnan;     // (columns beyond the matrix: counted nowhere)
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (row + u >= r1) break;
      const int64_t spk = espk[row + u];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = v[u][e];
        const bool tgt = ok[e] && ts[e] == spk;
        const bool lt_lo = x < flo, lt_hi = x < fhi;
        t1 += tgt ? 1u : 0u;
        b1 += (lt_lo && tgt) ? 1u : 0u;
        b0 += (lt_lo && !tgt) ? 1u : 0u;
        ab += (x >= fhi) ? 1u : 0u;
        if (lt_hi && !lt_lo) {
          const int c = tgt ? 1 : 0;
          const int slot = atomicAdd(&fc[c], 1);
          stage[wave][c][slot] = x;
        }
      }
    }
    if (__builtin_amdgcn_readfirstlane(fc[0]) > EER_WBUF - 1024) flush(0);
    if (__builtin_amdgcn_readfirstlane(fc[1]) > EER_WBUF - 1024) flush(1);
  }
  flush(0);
  flush(1);
  unsigned long long c[4] = {b0, b1, t1, ab};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    for (int o = 32; o > 0; o >>= 1) c[q] += __shfl_xor(c[q], o);
    if (lane == 0) red[wave][q] = c[q];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const unsigned long long sum = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    unsigned long long *dst = threadIdx.x == 0 ? &wo->below[0] : threadIdx.x == 1 ? &wo->below[1] : threadIdx.x == 2 ? &wo->targets : &wo->above;
    if (sum) atomicAdd(dst, sum);
  }
}

struct EerSource {
  const float *scores; int64_t ld, M, Nt; const int64_t *espk, *tspk;   // matrix + labels, or
  const float *pos; int64_t np; const float *neg; int64_t nn;            // two flat lists
  // row-sharded matrix: after every local pass the caller's reduction makes the counts global
  // (hist: sum over ranks; below: max; above: min).  nullptr = single process.
  int (*reduce)(void *ctx, unsigned long long *hist, unsigned *below, unsigned *above) = nullptr;
  void *ctx = nullptr;
  int64_t row_step = 1;                        // matrix form: every row_step-th row only (the pilot's sample)
  // windowed lists (the single-pass form): the lists hold the scores of a key window only; the counts below it and
  // the class totals come from the full pass
  bool windowed = false;
  unsigned long long base_p = 0, base_n = 0, tot_p = 0, tot_n = 0;
  // a matrix that exists one row slab at a time (plda_score_eer_dev: the scores are produced, consumed and dropped): `scores`
  // is nullptr, M / Nt / espk / tspk describe the whole matrix
  const struct EerSlabs *slabs = nullptr;
};
// produce: enqueue the scores of rows [r0, r0 + rows) (rows <= slab_rows) on the handle's stream, say where they are;
// sample: the same for every step-th row of the matrix (<= slab_rows of them) together with THOSE rows' speaker ids
struct EerSlabs {
  int64_t slab_rows;
  int (*produce)(void *ctx, int64_t r0, int64_t rows, const float **scores, int64_t *ld);
  int (*sample)(void *ctx, int64_t step, const float **scores, int64_t *ld, const int64_t **espk, int64_t *rows);
  void *ctx;
};

// One histogram pass over the local data -> hh (host).  No reduction here: see eer_device.
static int eer_pass(plda_handle *h, const EerSource &src, int shift, int nbits, unsigned prefix, int has_prefix,
                    unsigned long long *dhist, unsigned *dbelow, unsigned *dabove, std::vector<unsigned long long> &hh) {
  PLDA_HIP(h, hipMemsetAsync(dhist, 0, 2 * EER_BINS * 8, h->stream));
  if (src.slabs) {
    for (int64_t r0 = 0; r0 < src.M; r0 += src.slabs->slab_rows) {
      const int64_t rows = std::min(src.slabs->slab_rows, src.M - r0);
      const float *sc = nullptr;
      int64_t ld = 0;
      PLDA_TRY(src.slabs->produce(src.slabs->ctx, r0, rows, &sc, &ld));
      const int64_t strips = ceil_div(src.Nt, (int64_t)EER_STRIP);
      const int64_t slices = std::max<int64_t>(1, std::min<int64_t>(rows, (256 * 16) / strips));
      const int64_t rows_per_wg = ceil_div(rows, slices);
      eer_hist_strip_kernel<<<(unsigned)(strips * ceil_div(rows, rows_per_wg)), 256, 0, h->stream>>>(
          sc, ld, rows, src.Nt, src.espk + r0, src.tspk, rows_per_wg, shift, nbits, prefix, has_prefix, dhist, dbelow, dabove, 1);
    }
  } else if (src.scores) {
    const int64_t Ms = ceil_div(src.M, src.row_step);          // rows this pass walks
    const unsigned grid = (unsigned)std::min<int64_t>(Ms, 256 * 16);
    if (grid) {
      const int64_t strips = ceil_div(src.Nt, (int64_t)EER_STRIP);
      const int64_t slices = std::max<int64_t>(1, std::min<int64_t>(Ms, (256 * 16) / strips));
      const int64_t rows_per_wg = ceil_div(Ms, slices);
      eer_hist_strip_kernel<<<(unsigned)(strips * ceil_div(Ms, rows_per_wg)), 256, 0, h->stream>>>(
          src.scores, src.ld, Ms, src.Nt, src.espk, src.tspk, rows_per_wg, shift, nbits, prefix, has_prefix, dhist,
          dbelow, dabove, src.row_step);
    }
  } else {
    for (int c = 0; c < 2; ++c) {
      const float *p = c ? src.pos : src.neg;
      const int64_t n = c ? src.np : src.nn;
      const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(n, 256), 256 * 16);
      if (grid) eer_hist_kernel<<<grid, 256, 0, h->stream>>>(p, n, c, shift, nbits, prefix, has_prefix, dhist, dbelow, dabove);
    }
  }
  PLDA_LAUNCH_CHECK(h);
  hh.resize(2 * EER_BINS);
  PLDA_HIP(h, hipMemcpyAsync(hh.data(), dhist, 2 * EER_BINS * 8, hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  return PLDA_OK;
}

// out: [0] threshold, [1] FAR, [2] FRR, [3] EER = (FAR + FRR) / 2, [4] #targets, [5] #impostors
//
// Sharded calls (src.reduce): every rank makes the SAME four reduction calls whatever happens locally.
// A rank that fails (HIP error, inconsistent counts) keeps taking part with a poisoned histogram --
// 2^48 added to counter 0, far above any real count -- so that all ranks see the failure after the
// next sum and return an error together instead of leaving their peers blocked in a collective.
// window_missed (windowed lists only): set when the crossing key or one of its neighbours is not inside the lists --
// the caller then runs the three passes over the matrix; nothing is written to `out`.
int eer_device(plda_handle *h, const EerSource &src, double *out, bool *window_missed = nullptr) {
  typedef unsigned __int128 u128;
  if (window_missed) *window_missed = false;
  constexpr unsigned long long POISON = 1ull << 48;
  PLDA_HIP(h, h->w[10].reserve(2 * EER_BINS * 8 + 64));
  unsigned long long *dhist = h->w[10].as<unsigned long long>();
  unsigned *dbelow = reinterpret_cast<unsigned *>(dhist + 2 * EER_BINS), *dabove = dbelow + 1;
  std::vector<unsigned long long> H;
  // g(k) >= 0  <=>  P(k) * Nn >= (Nn - N(k)) * Np  with P, N = #targets / #impostors with key <= k
  unsigned long long Np = 0, Nn = 0, Pb = 0, Nb = 0;   // totals; counts strictly below the current range
  unsigned prefix = 0;
  const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
  unsigned long long cP = 0, cN = 0;   // counts at k1
  std::vector<unsigned long long> last;
  static const unsigned init[2] = {0u, 0xffffffffu};   // (static: the async copy below may read it after this frame)
  int rc = PLDA_OK;                                    // first failure seen by this rank
  for (int pass = 0; pass < 3; ++pass) {
    if (rc == PLDA_OK && pass == 2) {
      const hipError_t e = hipMemcpyAsync(dbelow, init, 8, hipMemcpyHostToDevice, h->stream);
      if (e != hipSuccess) rc = hip_fail(h, e, "hipMemcpyAsync(dbelow)", __FILE__, __LINE__);
    }
    if (rc == PLDA_OK) rc = eer_pass(h, src, shifts[pass], bits[pass], prefix, pass > 0, dhist, dbelow, dabove, H);
    if (src.reduce) {
      if (rc != PLDA_OK) { H.assign(2 * EER_BINS, 0ull); H[0] = POISON; }
      if (src.reduce(src.ctx, H.data(), nullptr, nullptr) != 0 && rc == PLDA_OK)
        rc = fail(h, PLDA_E_INVAL, "eer: the caller's reduction failed");
      if (rc == PLDA_OK && H[0] >= POISON) rc = fail(h, PLDA_E_NUMERIC, "eer: another rank of the sharded call failed");
    }
    if (rc != PLDA_OK) continue;
    const int nb = 1 << bits[pass];
    if (pass == 0) {
      for (int b = 0; b < nb; ++b) { Nn += H[b]; Np += H[EER_BINS + b]; }
      if (src.windowed) { Np = src.tot_p; Nn = src.tot_n; Pb = src.base_p; Nb = src.base_n; }   // the lists are a window of the data
      if (Np == 0 || Nn == 0) { rc = fail(h, PLDA_E_INVAL, "eer: need at least one target and one impostor trial"); continue; }
    }
    int sel = -1;
    unsigned long long P = Pb, N = Nb;
    for (int b = 0; b < nb; ++b) {
      const unsigned long long p2 = P + H[EER_BINS + b], n2 = N + H[b];
      if ((H[b] | H[EER_BINS + b]) && (u128)p2 * Nn >= (u128)(Nn - n2) * Np) { sel = b; cP = H[EER_BINS + b]; cN = H[b]; break; }
      P = p2; N = n2;
    }
    if (sel < 0 && src.windowed && window_missed) { *window_missed = true; return PLDA_OK; }   // the crossing lies above the window
    if (sel < 0) { rc = fail(h, PLDA_E_NUMERIC, "eer: crossing not found (inconsistent counts)"); continue; }
    Pb = P; Nb = N;
    prefix = (prefix << bits[pass]) | (unsigned)sel;
    if (pass == 2) last = H;
  }
  unsigned hb[2] = {0u, 0xffffffffu};
  if (rc == PLDA_OK) {
    hipError_t e = hipMemcpyAsync(hb, dbelow, 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) rc = hip_fail(h, e, "copy of the neighbour keys", __FILE__, __LINE__);
  }
  if (src.reduce && src.reduce(src.ctx, nullptr, &hb[0], &hb[1]) != 0 && rc == PLDA_OK)
    rc = fail(h, PLDA_E_INVAL, "eer: the caller's reduction failed");
  if (rc != PLDA_OK) return rc;
  const unsigned k1 = prefix;                       // smallest key with g >= 0; Pb/Nb = counts below k1
  // neighbours of k1 among the data keys
  const int b2 = (int)(k1 & 1023u);
  long long k0 = -1, k2 = -1;
  for (int b = b2 - 1; b >= 0; --b) if (last[b] | last[EER_BINS + b]) { k0 = (long long)((k1 & ~1023u) | (unsigned)b); break; }
  if (k0 < 0 && (Pb + Nb) > 0) k0 = hb[0];
  for (int b = b2 + 1; b < 1024; ++b) if (last[b] | last[EER_BINS + b]) { k2 = (long long)((k1 & ~1023u) | (unsigned)b); break; }
  if (k2 < 0 && (Pb + cP + Nb + cN) < (Np + Nn)) k2 = hb[1];
  if (src.windowed) {
    // the neighbours must have been SEEN in the lists: a predecessor / successor that exists in the data (counts say so)
    // but lies outside the window leaves hb at its initial value
    const bool need0 = (Pb + Nb) > 0, need2 = (Pb + cP + Nb + cN) < (Np + Nn);
    if ((need0 && k0 == 0) || (need2 && k2 == 0xffffffffll)) { if (window_missed) *window_missed = true; return PLDA_OK; }
  }
  // g(k1) = (Pb + cP)/Np - (Nn - Nb - cN)/Nn >= 0 ; g(prev) = Pb/Np - (Nn - Nb)/Nn < 0 (prev = k0, or the start).
  // The choice between the two candidates is made on |FAR - FRR| formed in float64 from the float64 rates,
  // exactly as the definition evaluates it (an exact tie such as 393/400 vs 394/400 around 787/800 must
  // stay a tie and go to the later candidate; extended precision breaks it the other way).
  const double far1 = (double)(Nn - Nb - cN) / (double)Nn, frr1 = (double)(Pb + cP) / (double)Np;
  const double far0 = (double)(Nn - Nb) / (double)Nn, frr0 = (double)Pb / (double)Np;
  double thr, far, frr;
  if (fabs(far1 - frr1) <= fabs(far0 - frr0)) {     // later candidate wins ties
    const double s = key_score(k1);
    thr = k2 >= 0 ? s + ((double)key_score((unsigned)k2) - s) / 2.0 : s + 1e-8;
    far = (double)(Nn - Nb - cN) / (double)Nn; frr = (double)(Pb + cP) / (double)Np;
  } else {
    const double s1 = key_score(k1);
    thr = k0 >= 0 ? (double)key_score((unsigned)k0) + (s1 - (double)key_score((unsigned)k0)) / 2.0 : s1;
    far = (double)(Nn - Nb) / (double)Nn; frr = (double)Pb / (double)Np;
  }
  out[0] = thr; out[1] = far; out[2] = frr; out[3] = 0.5 * (far + frr); out[4] = (double)Np; out[5] = (double)Nn;
  return PLDA_OK;
}

// ---- the single-pass form (see the header) ----
constexpr int64_t EER_PILOT_STEP = 32;
// Returns PLDA_OK with *done = true when `out` holds the result; *done = false: run the three passes.
// (full: the whole matrix -- in memory, or as slabs)
static int eer_matrix_windowed(plda_handle *h, const EerSource &full, double *out, bool *done) {
  typedef unsigned __int128 u128;
  *done = false;
  const float *dscores = full.scores;
  const int64_t ld = full.ld, M = full.M, Nt = full.Nt;
  const int64_t *despk = full.espk, *dtspk = full.tspk;
  TraceScope ts(h, "eer.pilot");
  PLDA_HIP(h, h->w[10].reserve(2 * EER_BINS * 8 + 64 + sizeof(EerWindowOut)));
  unsigned long long *dhist = h->w[10].as<unsigned long long>();
  unsigned *dbelow = reinterpret_cast<unsigned *>(dhist + 2 * EER_BINS), *dabove = dbelow + 1;
  EerWindowOut *dwo = reinterpret_cast<EerWindowOut *>(dhist + 2 * EER_BINS + 8);
  EerSource smp{dscores, ld, M, Nt, despk, dtspk, nullptr, 0, nullptr, 0};
  smp.row_step = EER_PILOT_STEP;
  int64_t pilot_step = EER_PILOT_STEP;
  if (full.slabs) {
    // the sample is produced (a small GEMM of its own) and must fit the slab buffer
    pilot_step = std::max<int64_t>(EER_PILOT_STEP, ceil_div(M, full.slabs->slab_rows));
    const float *sc = nullptr; const int64_t *se = nullptr; int64_t sld = 0, srows = 0;
    PLDA_TRY(full.slabs->sample(full.slabs->ctx, pilot_step, &sc, &sld, &se, &srows));
    smp = EerSource{sc, sld, srows, Nt, se, dtspk, nullptr, 0, nullptr, 0};
  }
  // pilot, pass 0 on the sample: per coarse bin (top 11 key bits) the sample's g = FRR - FAR; the band |g| < delta of five
  // standard errors around the sample's crossing starts in coarse bin ca and ends in cb (often the same); one pass over
  // the sample per end then places the window's ends on sub-bins (22 key bits)
  std::vector<unsigned long long> H0, H1;
  PLDA_TRY(eer_pass(h, smp, 21, 11, 0, 0, dhist, dbelow, dabove, H0));
  unsigned long long Np = 0, Nn = 0;
  for (int b = 0; b < EER_BINS; ++b) { Nn += H0[b]; Np += H0[EER_BINS + b]; }
  if (Np < 2000 || Nn < 2000) return PLDA_OK;                       // too few targets in the sample to bracket anything
  auto g_after = [&](unsigned long long p, unsigned long long n) { return (double)p / (double)Np - (double)(Nn - n) / (double)Nn; };
  double delta = 0.0;
  {
    unsigned long long P = 0, N = 0;
    int c0 = -1;
    for (int b = 0; b < EER_BINS; ++b) {
      const unsigned long long p2 = P + H0[EER_BINS + b], n2 = N + H0[b];
      if ((H0[b] | H0[EER_BINS + b]) && (u128)p2 * Nn >= (u128)(Nn - n2) * Np) { c0 = b; break; }
      P = p2; N = n2;
    }
    if (c0 < 0) return PLDA_OK;
    const double pe = std::min(0.5, std::max((double)P / (double)Np, 10.0 / (double)Np));     // ~ the sample's FRR at the crossing
    delta = 5.0 * (std::sqrt(pe * (1.0 - pe) / (double)Np) + std::sqrt(pe * (1.0 - pe) / (double)Nn)) + 2.0 / (double)Np;
  }
  int ca = -1, cb = -1;
  unsigned long long Pa = 0, Na = 0, Pc = 0, Nc = 0;                 // sample counts below coarse bins ca, cb
  {
    unsigned long long P = 0, N = 0;
    for (int b = 0; b < EER_BINS; ++b) {
      const unsigned long long p2 = P + H0[EER_BINS + b], n2 = N + H0[b];
      const double g = g_after(p2, n2);
      if (ca < 0 && g > -delta) { ca = b; Pa = P; Na = N; }
      if (g >= delta) { cb = b; Pc = P; Nc = N; break; }
      P = p2; N = n2;
    }
  }
  if (ca < 0 || cb < 0) return PLDA_OK;
  int blo = 0, bhi = EER_BINS - 1;
  unsigned long long win_lo = Pa + Na, win_hi = 0;                   // sample trials below the window's ends
  PLDA_TRY(eer_pass(h, smp, 10, 11, (unsigned)ca, 1, dhist, dbelow, dabove, H1));
  {
    unsigned long long p = Pa, n = Na;
    for (int b = 0; b < EER_BINS; ++b) {
      const unsigned long long p2 = p + H1[EER_BINS + b], n2 = n + H1[b];
      if (g_after(p2, n2) > -delta) { blo = b; break; }
      p = p2; n = n2;
    }
    blo = std::max(blo - 1, 0);                                      // one sub-bin of margin
    for (int b = 0; b < blo; ++b) win_lo += H1[b] + H1[EER_BINS + b];
  }
  if (cb != ca) PLDA_TRY(eer_pass(h, smp, 10, 11, (unsigned)cb, 1, dhist, dbelow, dabove, H1));
  {
    unsigned long long p = Pc, n = Nc;
    for (int b = 0; b < EER_BINS; ++b) {
      p += H1[EER_BINS + b]; n += H1[b];
      if (g_after(p, n) >= delta) { bhi = b; break; }
    }
    bhi = std::min(bhi + 1, EER_BINS - 1);
    win_hi = Pc + Nc;
    for (int b = 0; b <= bhi; ++b) win_hi += H1[b] + H1[EER_BINS + b];
  }
  const unsigned klo = ((unsigned)ca << 21) | ((unsigned)blo << 10);
  const unsigned long long khi64 = ((unsigned long long)cb << 21) | ((unsigned long long)(bhi + 1) << 10);
  if (khi64 > 0xffffffffull || khi64 <= klo) return PLDA_OK;
  const unsigned khi = (unsigned)khi64;
  const unsigned long long win = win_hi > win_lo ? win_hi - win_lo : 0;
  // the lists: the sample's in-window count scaled up, with room to spare; too wide a window is not worth a list
  const unsigned long long cap = (unsigned long long)((double)win * (double)pilot_step * 2.0) + (1ull << 20);
  if (cap > (unsigned long long)M * (unsigned long long)Nt / 8) return PLDA_OK;
  PLDA_HIP(h, h->eer_list[0].reserve((size_t)cap * 4));
  PLDA_HIP(h, h->eer_list[1].reserve((size_t)cap * 4));
  ts.next("eer.window_pass", (double)M * (double)Nt * 4.0, 2);
  PLDA_HIP(h, hipMemsetAsync(dwo, 0, sizeof(EerWindowOut), h->stream));
  const int64_t slab = full.slabs ? full.slabs->slab_rows : M;
  for (int64_t r0 = 0; r0 < M; r0 += slab) {
    const int64_t rows = std::min(slab, M - r0);
    const float *sc = dscores;
    int64_t sld = ld;
    if (full.slabs) PLDA_TRY(full.slabs->produce(full.slabs->ctx, r0, rows, &sc, &sld));
    const int64_t strips = ceil_div(Nt, (int64_t)EER_STRIP);
    const int64_t slices = std::max<int64_t>(1, std::min<int64_t>(rows, (256 * 16) / strips));
    const int64_t rows_per_wg = ceil_div(rows, slices);
    eer_window_strip_kernel<<<(unsigned)(strips * ceil_div(rows, rows_per_wg)), 256, 0, h->stream>>>(
        sc, sld, rows, Nt, despk + r0, dtspk, rows_per_wg, key_score(klo), key_score(khi), dwo, h->eer_list[0].as<float>(), h->eer_list[1].as<float>(), cap);
    PLDA_LAUNCH_CHECK(h);
  }
  EerWindowOut wo;
  PLDA_HIP(h, hipMemcpyAsync(&wo, dwo, sizeof(wo), hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  ts.next("eer.finish_on_window");
  if (wo.cursor[0] > cap || wo.cursor[1] > cap) return PLDA_OK;     // a list overflowed
  // every trial is below, inside or above the window -- unless its score is a NaN (the three passes place those by key)
  if (wo.below[0] + wo.below[1] + wo.cursor[0] + wo.cursor[1] + wo.above != (unsigned long long)M * (unsigned long long)Nt) return PLDA_OK;
  const unsigned long long TP = wo.targets, TN = (unsigned long long)M * (unsigned long long)Nt - wo.targets;
  if (TP == 0 || TN == 0) return fail(h, PLDA_E_INVAL, "eer: need at least one target and one impostor trial");
  // exact g at the window's ends: below it FRR - FAR must still be negative, at its end non-negative
  if ((u128)wo.below[1] * TN >= (u128)(TN - wo.below[0]) * TP) return PLDA_OK;
  if ((u128)(wo.below[1] + wo.cursor[1]) * TN < (u128)(TN - wo.below[0] - wo.cursor[0]) * TP) return PLDA_OK;
  if (wo.cursor[0] + wo.cursor[1] == 0) return PLDA_OK;
  EerSource lst{nullptr, 0, 0, 0, nullptr, nullptr, h->eer_list[1].as<float>(), (int64_t)wo.cursor[1], h->eer_list[0].as<float>(), (int64_t)wo.cursor[0]};
  lst.windowed = true;
  lst.base_p = wo.below[1]; lst.base_n = wo.below[0]; lst.tot_p = TP; lst.tot_n = TN;
  bool missed = false;
  PLDA_TRY(eer_device(h, lst, out, &missed));
  *done = !missed;
  return PLDA_OK;
}

int eer_matrix_device(plda_handle *h, const float *dscores, int64_t ld, int64_t M, int64_t Nt, const int64_t *despk,
                      const int64_t *dtspk, double *out,
                      int (*reduce)(void *, unsigned long long *, unsigned *, unsigned *), void *ctx) {
  // a rank of a sharded call may own no row at all (M == 0): it still takes part in the reductions
  if (!out || Nt <= 0 || ld < Nt || M < 0 || (M == 0 && !reduce) || (M > 0 && (!dscores || !despk || !dtspk)))
    return fail(h, PLDA_E_INVAL, "eer: bad argument");
  h->eer_last_passes = 3;
  if (!reduce && h->eer_variant != 1 && (h->eer_variant == 2 || (double)M * (double)Nt >= 2.5e8) && M >= 4 * EER_PILOT_STEP) {
    bool done = false;
    const EerSource fullsrc{dscores, ld, M, Nt, despk, dtspk, nullptr, 0, nullptr, 0};
    PLDA_TRY(eer_matrix_windowed(h, fullsrc, out, &done));
    if (done) { h->eer_last_passes = 1; return PLDA_OK; }
  }
  TraceScope ts(h, "eer.three_passes", 3.0 * (double)M * (double)Nt * 4.0, 2);
  EerSource s{M > 0 ? dscores : reinterpret_cast<const float *>(out), ld, M, Nt, despk, dtspk, nullptr, 0, nullptr, 0};
  s.reduce = reduce;
  s.ctx = ctx;
  return eer_device(h, s, out);
}

// ------------------------------------------------------------------------------------
// EER of a trials matrix that is never held (round 5: plda_score_eer_dev).  The reference's caller scores every trial and
// hands the scores to eer.py (scoring/scorePLDA.py:302-318 -> scoring/eer.py:68-76): what it wants is four numbers, not M x Nt
// floats -- C4's matrix is 192 GB.  Here the scores exist one row slab at a time (<= 4 GiB): the pilot's sample is a GEMM of
// every step-th enrol row, then every slab is scored (the test side packed once, the distinct enrol counts found once) and
// consumed by the window pass; the three-pass form, should the window miss, re-scores the slabs per pass.  Identical to
// plda_eer_matrix_dev on the materialised matrix (the kernels give a trial the same bits wherever its tile lies).
// Not yet inside the GEMM's epilogue: the slab is written and read back once, through HBM.
// ------------------------------------------------------------------------------------
__global__ void eer_gather_rows_kernel(const double *__restrict__ X, int D, int64_t step, int64_t rows, double *__restrict__ out) {
  const int64_t r = blockIdx.x;
  if (r >= rows) return;
  for (int d = threadIdx.x; d < D; d += blockDim.x) out[r * D + d] = X[r * step * D + d];
}
__global__ void eer_gather_meta_kernel(const int32_t *__restrict__ n, const double *__restrict__ zm, const double *__restrict__ zs,
                                       const int64_t *__restrict__ spk, int64_t step, int64_t rows, int32_t *__restrict__ on,
                                       double *__restrict__ ozm, double *__restrict__ ozs, int64_t *__restrict__ ospk) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  if (n) on[r] = n[r * step];
  if (zm) { ozm[r] = zm[r * step]; ozs[r] = zs[r * step]; }
  ospk[r] = spk[r * step];
}

struct ScoreEerCtx {
  plda_handle *h;
  const double *dU; const int32_t *dn; int n_uniform; int64_t M; const double *dV; int64_t Nt;
  const double *dzm, *dzs; const int64_t *despk;
  CountSet cs; bool has_cs; bool packedB;
  float *slab; int64_t slab_rows;
};
static int score_eer_produce(void *vc, int64_t r0, int64_t rows, const float **scores, int64_t *ld) {
  auto *c = static_cast<ScoreEerCtx *>(vc);
  const int D = c->h->Dout;
  PLDA_TRY(score_matrix_device(c->h, c->dU + r0 * D, c->dn ? c->dn + r0 : nullptr, c->n_uniform, rows, c->dV, c->Nt,
                               c->dzm ? c->dzm + r0 : nullptr, c->dzs ? c->dzs + r0 : nullptr, c->slab, c->Nt, c->packedB,
                               c->has_cs ? &c->cs : nullptr));
  c->packedB = true;
  *scores = c->slab; *ld = c->Nt;
  return PLDA_OK;
}
static int score_eer_sample(void *vc, int64_t step, const float **scores, int64_t *ld, const int64_t **espk, int64_t *rows) {
  auto *c = static_cast<ScoreEerCtx *>(vc);
  plda_handle *h = c->h;
  const int D = h->Dout;
  const int64_t Ms = ceil_div(c->M, step);
  const size_t oU = 0, oZ = round_up((size_t)Ms * D * 8, 256), oS = oZ + round_up((size_t)Ms * 16, 256), oN = oS + round_up((size_t)Ms * 8, 256);
  PLDA_HIP(h, h->eer_smp.reserve(oN + (size_t)Ms * 4 + 256));
  char *b = h->eer_smp.as<char>();
  double *sU = reinterpret_cast<double *>(b + oU), *szm = reinterpret_cast<double *>(b + oZ), *szs = szm + Ms;
  int64_t *sspk = reinterpret_cast<int64_t *>(b + oS);
  int32_t *sn = reinterpret_cast<int32_t *>(b + oN);
  eer_gather_rows_kernel<<<(unsigned)Ms, 256, 0, h->stream>>>(c->dU, D, step, Ms, sU);
  eer_gather_meta_kernel<<<(unsigned)ceil_div(Ms, 256), 256, 0, h->stream>>>(c->dn, c->dzm, c->dzs, c->despk, step, Ms, sn, szm, szs, sspk);
  PLDA_LAUNCH_CHECK(h);
  PLDA_TRY(score_matrix_device(h, sU, c->dn ? sn : nullptr, c->n_uniform, Ms, c->dV, c->Nt, c->dzm ? szm : nullptr, c->dzm ? szs : nullptr,
                               c->slab, c->Nt, c->packedB, c->has_cs ? &c->cs : nullptr));
  c->packedB = true;
  *scores = c->slab; *ld = c->Nt; *espk = sspk; *rows = Ms;
  return PLDA_OK;
}

int score_eer_device(plda_handle *h, const double *dU, const int32_t *dn, int n_uniform, int64_t M, const double *dV, int64_t Nt,
                     const double *dzmean, const double *dzstd, const int64_t *despk, const int64_t *dtspk, double *out) {
  if (!h->fitted) return fail(h, PLDA_E_NOT_FITTED, "score_eer: model not fitted");
  if (!dU || !dV || !despk || !dtspk || !out || M <= 0 || Nt <= 0) return fail(h, PLDA_E_INVAL, "score_eer: bad argument");
  if (!dn && n_uniform <= 0) return fail(h, PLDA_E_INVAL, "score_eer: n_uniform must be > 0 when n_enrol is NULL");
  ScoreEerCtx c{h, dU, dn, n_uniform, M, dV, Nt, (dzmean && dzstd) ? dzmean : nullptr, (dzmean && dzstd) ? dzstd : nullptr, despk};
  c.has_cs = false; c.packedB = false;
  if (dn) { PLDA_TRY(score_count_set_device(h, dn, M, &c.cs)); c.has_cs = true; }
  // slabs of <= 4 GiB of scores, whole 256-row tiles, at least one tile row
  int64_t rows = std::max<int64_t>(256, (((int64_t)4 << 30) / 4 / Nt) / 256 * 256);
  if (h->eer_slab_rows > 0) rows = round_up(h->eer_slab_rows, 256);      // PLDA_EER_SLAB_ROWS: small slabs for the tests
  rows = std::min(rows, round_up(M, 256));
  c.slab_rows = rows;
  PLDA_HIP(h, h->eer_slab.reserve((size_t)rows * Nt * 4));
  c.slab = h->eer_slab.as<float>();
  const EerSlabs sl{rows, score_eer_produce, score_eer_sample, &c};
  EerSource src{nullptr, Nt, M, Nt, despk, dtspk, nullptr, 0, nullptr, 0};
  src.slabs = &sl;
  h->prep_valid = false;           // (the slabs pack the test side themselves)
  h->eer_last_passes = 3;
  if (h->eer_variant != 1 && (h->eer_variant == 2 || (double)M * (double)Nt >= 2.5e8) && M >= 4 * EER_PILOT_STEP) {
    bool done = false;
    PLDA_TRY(eer_matrix_windowed(h, src, out, &done));
    if (done) { h->eer_last_passes = 1; return PLDA_OK; }
  }
  TraceScope ts(h, "eer.three_passes (slabs re-scored per pass)", 3.0 * (double)M * (double)Nt * 4.0, 2);
  return eer_device(h, src, out);
}

// ------------------------------------------------------------------------------------
// DET points (round 5): the numbers behind scoring/eer.py:34-62's plot -- bob.measure.plot.det(negatives, positives, 100)
// evaluates farfrr at n thresholds spread evenly from the smallest to the largest score (bob absent: definition
// restated in oracle/plda_oracle_np.py:det, PARITY UNPINNED; the plotting itself stays out of scope).  Two passes over
// the scores: the extreme keys, then per class a histogram of k(s) = #{i : t_i <= s} -- a guess from the division,
// corrected against the fp64 threshold table the host accumulated exactly as the definition does -- from which
// FAR_i = #{neg : k >= i + 1} / Nn and FRR_i = #{pos : k <= i} / Np follow as suffix / prefix sums.  n <= 2047.
// ------------------------------------------------------------------------------------
constexpr int DET_MAX = 2047;
__global__ __launch_bounds__(256) void det_minmax_kernel(const float *__restrict__ scores, int64_t ld, int64_t M, int64_t Nt,
                                                         unsigned *__restrict__ mm /*[0] min key, [1] max key*/) {
  unsigned lo = 0xffffffffu, hi = 0u;
  const int64_t total = M * Nt;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const unsigned k = score_key(scores[(idx / Nt) * ld + idx % Nt]);
    lo = k < lo ? k : lo; hi = k > hi ? k : hi;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned a = __shfl_xor(lo, o), b = __shfl_xor(hi, o);
    lo = a < lo ? a : lo; hi = b > hi ? b : hi;
  }
  if ((threadIdx.x & 63) == 0) { atomicMin(mm, lo); atomicMax(mm + 1, hi); }
}
// cls < 0: the class of trial (i, j) is espk[i] == tspk[j]; else every score is of class cls (the two-list form, M = 1)
__global__ __launch_bounds__(256) void det_hist_kernel(const float *__restrict__ scores, int64_t ld, int64_t M, int64_t Nt,
                                                       const int64_t *__restrict__ espk, const int64_t *__restrict__ tspk, int cls,
                                                       const double *__restrict__ thr, int n, double lo, double inv_step,
                                                       unsigned long long *__restrict__ hist /*[2][DET_MAX + 1]*/) {
  __shared__ unsigned lh[2][DET_MAX + 1];
  __shared__ double ts[DET_MAX + 1];
  for (int i = threadIdx.x; i < 2 * (DET_MAX + 1); i += 256) (&lh[0][0])[i] = 0;
  for (int i = threadIdx.x; i < n; i += 256) ts[i] = thr[i];
  __syncthreads();
  const int64_t total = M * Nt;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / Nt, c = idx % Nt;
    const double sc = (double)scores[r * ld + c];
    int k = (int)((sc - lo) * inv_step) + 1;            // guess of #{i : t_i <= s}
    k = k < 0 ? 0 : (k > n ? n : k);
    while (k > 0 && ts[k - 1] > sc) --k;                // t_{k-1} <= s must hold
    while (k < n && ts[k] <= sc) ++k;                   // and t_k > s
    const int cl = cls >= 0 ? cls : (espk[r] == tspk[c] ? 1 : 0);
    atomicAdd(&lh[cl][k], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * (DET_MAX + 1); i += 256) {
    const unsigned v = (&lh[0][0])[i];
    if (v) atomicAdd(hist + i, (unsigned long long)v);
  }
}

// parts: up to two (scores, ld, M, Nt, cls) pieces -- the labelled matrix (cls = -1), or the impostor and target lists
struct DetPart { const float *scores; int64_t ld, M, Nt; int cls; };
static int det_device(plda_handle *h, const DetPart *parts, int nparts, const int64_t *despk, const int64_t *dtspk, int npoints,
                      double *far, double *frr, double *thresholds) {
  if (npoints < 2 || npoints > DET_MAX) return fail(h, PLDA_E_INVAL, "det: 2 <= n_points <= %d", DET_MAX);
  const size_t hb = (size_t)2 * (DET_MAX + 1) * 8;
  PLDA_HIP(h, h->w[10].reserve(hb + 64 + (size_t)DET_MAX * 8 + 64));
  unsigned long long *dhist = h->w[10].as<unsigned long long>();
  unsigned *dmm = reinterpret_cast<unsigned *>(dhist + 2 * (DET_MAX + 1));
  double *dthr = reinterpret_cast<double *>(dmm + 16);
  static const unsigned init[2] = {0xffffffffu, 0u};
  PLDA_HIP(h, hipMemcpyAsync(dmm, init, 8, hipMemcpyHostToDevice, h->stream));
  for (int p = 0; p < nparts; ++p) {
    const int64_t total = parts[p].M * parts[p].Nt;
    if (total > 0) det_minmax_kernel<<<(unsigned)std::min<int64_t>(ceil_div(total, 256), 256 * 32), 256, 0, h->stream>>>(parts[p].scores, parts[p].ld, parts[p].M, parts[p].Nt, dmm);
  }
  PLDA_LAUNCH_CHECK(h);
  unsigned mm[2];
  PLDA_HIP(h, hipMemcpyAsync(mm, dmm, 8, hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  if (mm[0] > mm[1]) return fail(h, PLDA_E_INVAL, "det: no scores");
  // the thresholds exactly as the definition accumulates them (float64 running sum)
  const double lo = (double)key_score(mm[0]), hi = (double)key_score(mm[1]);
  const double step = (hi - lo) / ((double)npoints - 1.0);
  std::vector<double> thr((size_t)npoints);
  double t = lo;
  for (int i = 0; i < npoints; ++i) { thr[(size_t)i] = t; t += step; }
  PLDA_HIP(h, hipMemcpyAsync(dthr, thr.data(), (size_t)npoints * 8, hipMemcpyHostToDevice, h->stream));
  PLDA_HIP(h, hipMemsetAsync(dhist, 0, hb, h->stream));
  for (int p = 0; p < nparts; ++p) {
    const int64_t total = parts[p].M * parts[p].Nt;
    if (total > 0)
      det_hist_kernel<<<(unsigned)std::min<int64_t>(ceil_div(total, 256), 256 * 8), 256, 0, h->stream>>>(
          parts[p].scores, parts[p].ld, parts[p].M, parts[p].Nt, despk, dtspk, parts[p].cls, dthr, npoints, lo, step > 0.0 ? 1.0 / step : 0.0, dhist);
  }
  PLDA_LAUNCH_CHECK(h);
  std::vector<unsigned long long> H((size_t)2 * (DET_MAX + 1));
  PLDA_HIP(h, hipMemcpyAsync(H.data(), dhist, hb, hipMemcpyDeviceToHost, h->stream));
  PLDA_HIP(h, hipStreamSynchronize(h->stream));
  unsigned long long Nn = 0, Np = 0;
  for (int k = 0; k <= npoints; ++k) { Nn += H[(size_t)k]; Np += H[(size_t)(DET_MAX + 1) + k]; }
  if (Nn == 0 || Np == 0) return fail(h, PLDA_E_INVAL, "det: need at least one target and one impostor score");
  // k(s) = #{i : t_i <= s}:  s >= t_i  <=>  k >= i + 1;   s < t_i  <=>  k <= i
  unsigned long long below_p = 0, ge_n = Nn;
  for (int i = 0; i < npoints; ++i) {
    ge_n -= H[(size_t)i];                                   // impostors with k == i are below t_i
    below_p += H[(size_t)(DET_MAX + 1) + i];                // targets with k <= i are below t_i
    far[i] = (double)ge_n / (double)Nn;
    frr[i] = (double)below_p / (double)Np;
    if (thresholds) thresholds[i] = thr[(size_t)i];
  }
  return PLDA_OK;
}

int det_matrix_device(plda_handle *h, const float *dscores, int64_t ld, int64_t M, int64_t Nt, const int64_t *despk, const int64_t *dtspk,
                      int npoints, double *far, double *frr, double *thresholds) {
  if (!dscores || !despk || !dtspk || !far || !frr || M <= 0 || Nt <= 0 || ld < Nt) return fail(h, PLDA_E_INVAL, "det: bad argument");
  const DetPart part{dscores, ld, M, Nt, -1};
  return det_device(h, &part, 1, despk, dtspk, npoints, far, frr, thresholds);
}
int det_lists_device(plda_handle *h, const float *dpos, int64_t np, const float *dneg, int64_t nn, int npoints, double *far, double *frr,
                     double *thresholds) {
  if (!dpos || !dneg || !far || !frr || np <= 0 || nn <= 0) return fail(h, PLDA_E_INVAL, "det: need at least one target and one impostor score");
  const DetPart parts[2] = {{dneg, nn, 1, nn, 0}, {dpos, np, 1, np, 1}};
  return det_device(h, parts, 2, nullptr, nullptr, npoints, far, frr, thresholds);
}

int eer_lists_device(plda_handle *h, const float *dpos, int64_t np, const float *dneg, int64_t nn, double *out) {
  if (!dpos || !dneg || !out || np <= 0 || nn <= 0) return fail(h, PLDA_E_INVAL, "eer: need at least one target and one impostor score");
  EerSource s{nullptr, 0, 0, 0, nullptr, nullptr, dpos, np, dneg, nn};
  return eer_device(h, s, out);
}

}  // namespace plda
