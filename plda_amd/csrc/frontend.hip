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

int dvector_pool_device(plda_handle *h, const void *dframes, int dtype, int64_t T, int D,
                        const int64_t *doffsets, int64_t U, int method, int l2norm, double *dout) {
  if (U <= 0) return PLDA_OK;
  if (!dframes || !doffsets || !dout || D <= 0 || D > 64 * DV_MAXE || T < 0 || method < 0 || method > 2 ||
      (dtype != 0 && dtype != 1))
    return fail(h, PLDA_E_INVAL, "dvector_pool: bad argument (D must be <= %d, dtype 0=f32/1=f64, method 0..2)", 64 * DV_MAXE);
  if (dtype == 0)
    dvector_pool_kernel<float><<<(unsigned)U, 256, 0, h->stream>>>(static_cast<const float *>(dframes), D, doffsets,
                                                                  method, l2norm, dout);
  else
    dvector_pool_kernel<double><<<(unsigned)U, 256, 0, h->stream>>>(static_cast<const double *>(dframes), D, doffsets,
                                                                   method, l2norm, dout);
  PLDA_LAUNCH_CHECK(h);
  return PLDA_OK;
}

}  // namespace plda
