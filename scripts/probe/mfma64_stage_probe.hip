// Development probe: the K4 transform kernel's stage loop rebuilt piece by piece on top of the inner loop that runs at
// 97 % by itself (mfma64_lds_probe.hip), to see which piece costs the 35 %:
//   STAGE 0: fragment reads + MFMAs + one barrier per stage (4 k-steps), two LDS buffers read alternately
//   STAGE 1: + every thread writes 11 doubles per stage into the other buffer (early waves at the stage's start, late
//            waves at its end), values from registers
//   STAGE 2: + those values come from 11 global loads per stage, issued behind k-steps 0 and 1: 7 from a 346 KB
//            L2-resident array every workgroup reads (T), 4 from a streamed array (X, 128 B per row and stage)
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/mfma64_stage_probe.hip -o scripts/probe/mfma64_stage_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int LD = 18, NT = 13, ROWS = 128, CP = 224, STAGEW = (CP + ROWS) * LD, TP = 7, XP = 4, DIN = 208;
template <int STAGE>
__global__ __launch_bounds__(512) void k(double *out, const double *T, const double *X, int nblocks) {
  extern __shared__ double lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fi = lane & 15, fk = lane >> 4, lk = t & 15, lr = t >> 4;
  const bool early = wave >= 4;
  for (int i = t; i < 2 * STAGEW; i += 512) lds[i] = 1.0 + (i % 7) * 1e-3;
  __syncthreads();
  const int tfrag = fi * LD + fk, xfrag = (CP + wave * 16 + fi) * LD + fk;
  double rt[TP], rx[XP];
  for (int p = 0; p < TP; ++p) rt[p] = 1.0 + p;
  for (int p = 0; p < XP; ++p) rx[p] = 2.0 + p;
  const double *tptr = T + (size_t)lr * DIN + lk;
  double sink = 0;
  int cur = 0;
  for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const double *xptr = X + ((size_t)blk * ROWS + lr) * DIN + lk;
    f64x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f64x4{0, 0, 0, 0};
    auto fetch = [&](int k0, int q) {
      if (STAGE < 2) return;
#pragma unroll
      for (int p = 0; p < TP; ++p) if ((p & 1) == q) rt[p] = tptr[(size_t)p * 32 * DIN + k0];
#pragma unroll
      for (int p = 0; p < XP; ++p) if ((p & 1) == q) rx[p] = xptr[(size_t)p * 32 * DIN + k0];
    };
    auto stage = [&](double *buf) {
      if (STAGE < 1) return;
#pragma unroll
      for (int p = 0; p < TP; ++p) buf[(lr + 32 * p) * LD + lk] = rt[p];
#pragma unroll
      for (int p = 0; p < XP; ++p) buf[(CP + lr + 32 * p) * LD + lk] = rx[p];
    };
    for (int k0 = 0; k0 < DIN; k0 += 16) {
      const bool more = k0 + 16 < DIN;
      if (early && more) stage(lds + (cur ^ 1) * STAGEW);
      const int kf = (early ? k0 + 32 : k0 + 16) % DIN;
      const double *Ts = lds + cur * STAGEW + tfrag, *Xs = lds + cur * STAGEW + xfrag;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const double a = Xs[kk * 4];
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) {
          const double b = Ts[tn * 16 * LD + kk * 4];
          acc[tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[tn], 0, 0, 0);
        }
        if (kk < 2) fetch(kf, kk);
        asm volatile("" ::: "memory");
      }
      if (!early && more) stage(lds + (cur ^ 1) * STAGEW);
      __syncthreads();
      cur ^= 1;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) sink += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  }
  out[blockIdx.x * blockDim.x + t] = sink;
}
template <int STAGE>
void run(const char *name, double *out, const double *T, const double *X, int nblocks) {
  const size_t sm = (size_t)2 * STAGEW * 8;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k<STAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<STAGE><<<256, 512, sm>>>(out, T, X, 256);
  hipDeviceSynchronize();
  hipEventRecord(e0); for (int r = 0; r < 5; ++r) k<STAGE><<<256, 512, sm>>>(out, T, X, nblocks); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double mf = (double)nblocks * 8 * (DIN / 4) * NT;
  printf("%-64s %.1f us  %.1f TFLOP/s = %.3f of 78.6\n", name, ms * 1e3, mf * 2048 / ms / 1e9, mf * 2048 / ms / 1e9 / 78.6);
}
int main() {
  const int nblocks = 768;
  double *out, *T, *X;
  hipMalloc(&out, 256 * 512 * 8); hipMalloc(&T, (size_t)CP * DIN * 8); hipMalloc(&X, (size_t)nblocks * ROWS * DIN * 8);
  hipMemset(T, 0, (size_t)CP * DIN * 8); hipMemset(X, 0, (size_t)nblocks * ROWS * DIN * 8);
  run<0>("reads + MFMAs + barrier per stage", out, T, X, nblocks);
  run<1>("+ 11 LDS writes per thread and stage (from registers)", out, T, X, nblocks);
  run<2>("+ 11 global loads per thread and stage behind k-steps 0, 1", out, T, X, nblocks);
  return 0;
}
