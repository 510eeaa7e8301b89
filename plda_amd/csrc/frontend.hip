// plda_amd/csrc/frontend.hip -- d-vector front-end: the step right
// before the PLDA path.  Replaces /root/reference/scoring/extractdvector.py:19-58:
//   getnormalizedvector  uttvec / ||uttvec||_2 per frame           (:19-29)
//   extractdvectormean / max / var  pooling over the frames        (:32-47)
// and the *_nol2 variants (:50-59).  HBM bound: every frame is read exactly once
// (T * D * sizeof(element) bytes); fp64 accumulation whatever the input type.
// One workgroup (4 waves) per utterance; a wave takes every 4th frame, lane l owns
// columns l, l + 64, ... so a frame read is one contiguous burst; the frame norm is a
// DPP wave reduction; the per-wave partial (sum, sum of squares, max) meet in LDS.
#include "common.hpp"

#include <algorithm>

namespace plda {

constexpr int DV_MAXE = 16;   // D <= 1024

template <typename T>
__global__ __launch_bounds__(256) void dvector_pool_kernel(const T *__restrict__ frames, int D,
                                                           const int64_t *__restrict__ offsets, int method,
                                                           int l2norm, double *__restrict__ out) {
  __shared__ double part[3][4][64 * DV_MAXE / 4];   // sized for D <= 256 per pass; larger D loops passes
  const int u = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t beg = offsets[u], end = offsets[u + 1];
  const int64_t n = end - beg;
  // columns are processed in passes of 256 so that the LDS combine stays small
  for (int d0 = 0; d0 < D; d0 += 256) {
    double sum[4], sq[4], mx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { sum[e] = 0.0; sq[e] = 0.0; mx[e] = -__builtin_huge_val(); }
    for (int64_t f = beg + wave; f < end; f += 4) {
      const T *row = frames + f * (int64_t)D;
      double inv = 1.0;
      if (l2norm) {
        double ss = 0.0;
        for (int d = lane; d < D; d += 64) { const double x = (double)row[d]; ss += x * x; }
        ss = wave_sum_f64(ss);
        inv = 1.0 / sqrt(ss);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = d0 + lane + e * 64;
        if (d < D) {
          const double y = (double)row[d] * inv;
          sum[e] += y; sq[e] += y * y; mx[e] = y > mx[e] || y != y ? y : mx[e];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      part[0][wave][lane + e * 64] = sum[e];
      part[1][wave][lane + e * 64] = sq[e];
      part[2][wave][lane + e * 64] = mx[e];
    }
    __syncthreads();
    const int d = d0 + threadIdx.x;
    if (d < D) {
      const int c = threadIdx.x;
      const double s = (part[0][0][c] + part[0][1][c]) + (part[0][2][c] + part[0][3][c]);
      const double q = (part[1][0][c] + part[1][1][c]) + (part[1][2][c] + part[1][3][c]);
      double m = part[2][0][c];
#pragma unroll
      for (int w = 1; w < 4; ++w) { const double v = part[2][w][c]; m = (v > m || v != v) ? v : m; }
      double r;
      const double nn = (double)n;
      if (n == 0) r = __builtin_nan("");
      else if (method == 0) r = s / nn;
      else if (method == 1) r = m;
      else { const double mean = s / nn; r = q / nn - mean * mean; if (r < 0.0) r = 0.0; }
      out[(int64_t)u * D + d] = r;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------
// Vectorised path: float32 frames with D = 4 G, G a power of two in [4, 64].  A frame is G
// float4 items, so one wave-wide 16-byte load fetches 64 / G whole frames (1 KiB per
// instruction); four such loads are issued back to back before any arithmetic so that
// the reductions of one batch overlap the loads of the next.  The frame norm is a G-lane
// butterfly (DPP inside a 16-lane row, ds_bpermute across rows).
// ------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ double group_sum_f64(double x) {
  if (G >= 2) x += dpp_f64<0xB1>(x);    // xor 1
  if (G >= 4) x += dpp_f64<0x4E>(x);    // xor 2
  if (G >= 8) x += dpp_f64<0x141>(x);   // quads already uniform: half-row mirror == xor 4
  if (G >= 16) x += dpp_f64<0x140>(x);  // row mirror == xor 8
  if (G >= 32) x += __shfl_xor(x, 16);
  if (G >= 64) x += __shfl_xor(x, 32);
  return x;
}

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int G>
__global__ __launch_bounds__(256) void dvector_pool_vec4_kernel(const f32x4v *__restrict__ frames,
                                                                const int64_t *__restrict__ offsets,
                                                                int method, int l2norm,
                                                                double *__restrict__ out) {
  constexpr int FPW = 64 / G;   // frames per wave-wide load
  constexpr int UNR = 4;
  __shared__ double part[3][4][4 * G];
  const int u = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane / G, c = lane % G;
  const int64_t beg = offsets[u], end = offsets[u + 1];
  const int64_t n = end - beg;
  double sum[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
  double mx[4] = {-__builtin_huge_val(), -__builtin_huge_val(), -__builtin_huge_val(), -__builtin_huge_val()};
  for (int64_t base = beg + (int64_t)wave * FPW * UNR; base < end; base += 4 * FPW * UNR) {
    f32x4v x[UNR];
    bool ok[UNR];
#pragma unroll
    for (int k = 0; k < UNR; ++k) {
      const int64_t f = base + k * FPW + fr;
      ok[k] = f < end;
      x[k] = ok[k] ? __builtin_nontemporal_load(frames + f * G + c) : f32x4v{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < UNR; ++k) {
      double y[4] = {(double)x[k].x, (double)x[k].y, (double)x[k].z, (double)x[k].w};
      if (l2norm) {
        const double ss = group_sum_f64<G>(y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3]);
        const double inv = 1.0 / sqrt(ss);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] *= inv;
      }
      if (ok[k]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sum[e] += y[e];
          sq[e] += y[e] * y[e];
          mx[e] = (y[e] > mx[e] || y[e] != y[e]) ? y[e] : mx[e];
        }
      }
    }
  }
  // combine the FPW frame groups of the wave (lanes with equal c), then the 4 waves
#pragma unroll
  for (int o = G; o < 64; o <<= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sum[e] += __shfl_xor(sum[e], o);
      sq[e] += __shfl_xor(sq[e], o);
      const double m2 = __shfl_xor(mx[e], o);
      mx[e] = (m2 > mx[e] || m2 != m2) ? m2 : mx[e];
    }
  }
  if (fr == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      part[0][wave][4 * c + e] = sum[e];
      part[1][wave][4 * c + e] = sq[e];
      part[2][wave][4 * c + e] = mx[e];
    }
  }
  __syncthreads();
  const int d = threadIdx.x;
  if (d < 4 * G) {
    const double s = (part[0][0][d] + part[0][1][d]) + (part[0][2][d] + part[0][3][d]);
    const double q = (part[1][0][d] + part[1][1][d]) + (part[1][2][d] + part[1][3][d]);
    double m = part[2][0][d];
#pragma unroll
    for (int w = 1; w < 4; ++w) { const double v = part[2][w][d]; m = (v > m || v != v) ? v : m; }
    const double nn = (double)n;
    double r;
    if (n == 0) r = __builtin_nan("");
    else if (method == 0) r = s / nn;
    else if (method == 1) r = m;
    else { const double mean = s / nn; r = q / nn - mean * mean; if (r < 0.0) r = 0.0; }
    out[(int64_t)u * (4 * G) + d] = r;
  }
}

int dvector_pool_device(plda_handle *h, const void *dframes, int dtype, int64_t T, int D,
                        const int64_t *doffsets, int64_t U, int method, int l2norm, double *dout) {
  if (U <= 0) return PLDA_OK;
  if (!dframes || !doffsets || !dout || D <= 0 || D > 64 * DV_MAXE || T < 0 || method < 0 || method > 2 ||
      (dtype != 0 && dtype != 1))
    return fail(h, PLDA_E_INVAL, "dvector_pool: bad argument (D must be <= %d, dtype 0=f32/1=f64, method 0..2)", 64 * DV_MAXE);
  if (dtype == 0 && (reinterpret_cast<uintptr_t>(dframes) & 15) == 0 &&
      (D == 16 || D == 32 || D == 64 || D == 128 || D == 256)) {
    const f32x4v *fv = static_cast<const f32x4v *>(dframes);
#define DV(G_) dvector_pool_vec4_kernel<G_><<<(unsigned)U, 256, 0, h->stream>>>(fv, doffsets, method, l2norm, dout)
    if (D == 16) DV(4); else if (D == 32) DV(8); else if (D == 64) DV(16); else if (D == 128) DV(32); else DV(64);
#undef DV
  } else if (dtype == 0)
    dvector_pool_kernel<float><<<(unsigned)U, 256, 0, h->stream>>>(static_cast<const float *>(dframes), D, doffsets,
                                                                  method, l2norm, dout);
  else
    dvector_pool_kernel<double><<<(unsigned)U, 256, 0, h->stream>>>(static_cast<const double *>(dframes), D, doffsets,
                                                                   method, l2norm, dout);
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

// ------------------------------------------------------------------------------------
// HTK feature files (the data format in front of the path): replaces the reference's reader
// chtk::htk_load (/root/reference/chtk/chtk.cpp:38-88; used at src/kaldi-utils.hpp:22).  HBM-bound byte
// work: every 32-bit word of a frame is byte-swapped (big-endian floats) and frame i of the output is the
// concatenation of the file's frames clamp(i - F .. i + F, 0, n - 1).  A call decodes a whole BATCH of
// files from one blob (the data sections, 4-byte aligned, short files zero-padded by the caller the way
// the reference's zero-initialised read buffer behaves): one wave per output frame; the wave locates its
// file by a uniform binary search of the frame offsets, lanes stride over the frame's words.
// ------------------------------------------------------------------------------------
constexpr int HTK_MAX_FR = 256;   // output frames per workgroup (upper bound)

// One workgroup per chunk of FR consecutive output frames (about 32 KiB of output).  Wave 0 locates the
// chunk's first file with a 64-ary search of the frame offsets (three dependent probes for 2.6e5 files
// instead of eighteen), every frame then walks forward from there; (source base, frame index, frame count)
// per frame go to LDS and the copy loop runs over 16-byte vectors when the frame size and the file's
// placement allow it, over single words otherwise.
__global__ __launch_bounds__(256) void htk_frames_kernel(const uint32_t *__restrict__ blob,
                                                         const int64_t *__restrict__ file_off,
                                                         const int64_t *__restrict__ frame_off, int64_t U, int64_t T,
                                                         int W, int F, int FR, int vec_ok, uint32_t *__restrict__ out) {
  __shared__ int64_t s_base[HTK_MAX_FR];
  __shared__ int s_i[HTK_MAX_FR], s_n[HTK_MAX_FR];
  __shared__ int64_t s_u0;
  const int t = threadIdx.x, lane = t & 63;
  const int64_t t0 = (int64_t)blockIdx.x * FR;
  const int nfr = (int)min<int64_t>(FR, T - t0);
  if (t < 64) {
    // largest u in [0, U-1] with frame_off[u] <= t0
    int64_t lo = 0, hi = U - 1;
    while (lo < hi) {
      const int64_t span = hi - lo, step = (span + 63) / 64;       // probes lo + (l+1) * step, clipped to hi
      const int64_t p = min(lo + (int64_t)(lane + 1) * step, hi);
      const unsigned long long ok = __ballot(frame_off[p] <= t0);
      const int cnt = __popcll(ok);                                 // monotone: the first cnt probes satisfy it
      const int64_t nlo = cnt == 0 ? lo : min(lo + (int64_t)cnt * step, hi);
      const int64_t nhi = cnt == 64 ? hi : min(lo + (int64_t)(cnt + 1) * step, hi) - 1;
      lo = nlo;
      hi = max(nlo, nhi);
    }
    if (lane == 0) s_u0 = lo;
  }
  __syncthreads();
  for (int j = t; j < nfr; j += 256) {
    int64_t u = s_u0;
    const int64_t tt = t0 + j;
    while (u + 1 < U && frame_off[u + 1] <= tt) ++u;               // also steps over empty files
    const int64_t f0 = frame_off[u];
    s_base[j] = file_off[u];
    s_i[j] = (int)(tt - f0);
    s_n[j] = (int)(frame_off[u + 1] - f0);
  }
  __syncthreads();
  const int rowW = (2 * F + 1) * W;
  uint32_t *dst = out + t0 * rowW;
  if (vec_ok) {
    const int Wv = W >> 2, rowV = rowW >> 2, total = nfr * rowV;
#pragma unroll 4
    for (int e = t; e < total; e += 256) {
      const int fl = e / rowV, o = e - fl * rowV;
      const int64_t base = s_base[fl];
      if (base & 3) continue;                                       // unaligned file: word loop below
      const int jj = o / Wv, wv = o - jj * Wv;
      int fr = s_i[fl] + jj - F;
      fr = fr < 0 ? 0 : (fr > s_n[fl] - 1 ? s_n[fl] - 1 : fr);
      uint4 v = *reinterpret_cast<const uint4 *>(blob + base + (int64_t)fr * W + wv * 4);
      v.x = __builtin_bswap32(v.x); v.y = __builtin_bswap32(v.y);
      v.z = __builtin_bswap32(v.z); v.w = __builtin_bswap32(v.w);
      *reinterpret_cast<uint4 *>(dst + (int64_t)e * 4) = v;
    }
  }
  {
    const int total = nfr * rowW;
    const bool vec = vec_ok != 0;
#pragma unroll 4
    for (int e = t; e < total; e += 256) {
      const int fl = e / rowW, o = e - fl * rowW;
      const int64_t base = s_base[fl];
      if (vec && (base & 3) == 0) continue;                         // done by the vector loop
      const int jj = o / W, w = o - jj * W;
      int fr = s_i[fl] + jj - F;
      fr = fr < 0 ? 0 : (fr > s_n[fl] - 1 ? s_n[fl] - 1 : fr);
      dst[e] = __builtin_bswap32(blob[base + (int64_t)fr * W + w]);
    }
  }
}

int htk_frames_device(plda_handle *h, const void *dblob, const int64_t *dfile_off, const int64_t *dframe_off,
                      int64_t U, int64_t T, int samplesize, int frm_ext, float *dout) {
  if (U <= 0 || T <= 0) return PLDA_OK;
  if (!dblob || !dfile_off || !dframe_off || !dout || samplesize <= 0 || frm_ext < 0)
    return fail(h, PLDA_E_INVAL, "htk_frames: bad argument");
  if (samplesize % 4) return fail(h, PLDA_E_INVAL, "htk_frames: samplesize %d is not a multiple of 4", samplesize);
  if ((reinterpret_cast<uintptr_t>(dblob) & 3) != 0) return fail(h, PLDA_E_INVAL, "htk_frames: blob must be 4-byte aligned");
  const int W = samplesize / 4;
  const int64_t rowW = (int64_t)(2 * frm_ext + 1) * W;
  if (rowW > (1 << 20)) return fail(h, PLDA_E_INVAL, "htk_frames: output frame of %lld words unsupported", (long long)rowW);
  const int FR = (int)std::min<int64_t>(HTK_MAX_FR, std::max<int64_t>(1, 8192 / rowW));
  if (ceil_div(T, FR) > 0x7fffffffLL) return fail(h, PLDA_E_INVAL, "htk_frames: too many frames in one call");
  // 16-byte path: frame size a multiple of 16 bytes and 16-byte aligned blob / output (files whose data
  // section is not 16-byte aligned inside the blob fall back to words individually)
  const int vec_ok = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(dout) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(dblob) & 15) == 0;
  htk_frames_kernel<<<(unsigned)ceil_div(T, FR), 256, 0, h->stream>>>(
      static_cast<const uint32_t *>(dblob), dfile_off, dframe_off, U, T, W, frm_ext, FR, vec_ok,
      reinterpret_cast<uint32_t *>(dout));
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

}  // namespace plda
