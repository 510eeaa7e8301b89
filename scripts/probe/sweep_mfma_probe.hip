// Development probe: the block-sweep SPD inverse of the E-step (plda_amd/csrc/sweep_mfma.inc) built with a clock read
// at every phase boundary -- where do the cycles of a block step go (panel, the next block's cross, update | pivot
// block, and the three barriers)?  Prints cycles per phase summed over the block steps for every wave, the kernel time and
// the residual of the inverse.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probe/sweep_mfma_probe.hip -o scripts/probe/sweep_mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
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
__device__ __forceinline__ double readlane_f64(double x, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ long long *g_clock;     // [16 waves][8 phases], then start / end stamps
// clocks accumulate in registers and are written once at the end (a global read-modify-write per phase costs more than
// the phases); -DNOCLOCK builds the kernel as the product has it
#ifndef NOCLOCK
#define SWM_CLOCK(i)                                                                   \
  do {                                                                                 \
    const long long now_ = clock64();                                                  \
    if ((i) > 0) cacc_[i] += now_ - last_;                                             \
    last_ = now_;                                                                      \
  } while (0)
#define SWM_PCLOCK(i)                                                                  \
  do {                                                                                 \
    const long long now_ = clock64();                                                  \
    if ((i) > 0) pacc_[i] += now_ - plast_;                                            \
    plast_ = now_;                                                                     \
  } while (0)
#define SWM_PROBE_PLOCALS long long plast_ = clock64();
#define SWM_PROBE_LOCALS long long last_ = clock64(), cacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pacc_[4] = {0, 0, 0, 0}; \
  if (t == 0) { g_clock[128] = wall_clock64(); g_clock[130] = clock64(); }
#define SWM_PROBE_END                                                                  \
  if (t == 0) { g_clock[129] = wall_clock64(); g_clock[131] = clock64(); }             \
  if (lane == 0) {                                                                     \
    for (int i_ = 1; i_ < 8; ++i_) g_clock[wave * 8 + i_] = cacc_[i_];                 \
    if (wave == 15) for (int i_ = 1; i_ < 4; ++i_) g_clock[136 + i_] = pacc_[i_];      \
  }
#else
#define SWM_CLOCK(i)
#define SWM_PCLOCK(i)
#endif
#include "../../plda_amd/csrc/sweep_mfma.inc"

int main(int argc, char **argv) {
  const int D = argc > 1 ? atoi(argv[1]) : 200;
  constexpr int NT = 13;
  if ((D + 15) / 16 != NT) { printf("probe is built for 13 tile rows (193..208)\n"); return 1; }
  std::vector<double> A((size_t)D * D), X((size_t)D * D);
  // SPD: G G^T / D + I with a fixed LCG
  std::vector<double> G((size_t)D * D);
  uint64_t s = 12345;
  for (auto &g : G) { s = s * 6364136223846793005ull + 1442695040888963407ull; g = ((s >> 11) * (1.0 / 9007199254740992.0)) - 0.5; }
  for (int i = 0; i < D; ++i)
    for (int j = 0; j < D; ++j) {
      double x = 0;
      for (int k = 0; k < D; ++k) x += G[(size_t)i * D + k] * G[(size_t)j * D + k];
      A[(size_t)i * D + j] = x / D + (i == j ? 0.1 : 0.0);
    }
  double *dA, *dX; int *dF; long long *dC;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dX, A.size() * 8); hipMalloc(&dF, 4); hipMalloc(&dC, 160 * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipMemset(dF, 0, 4);
  hipMemcpyToSymbol(HIP_SYMBOL(g_clock), &dC, sizeof(dC));
  constexpr size_t lds = (size_t)((3 * NT + 8) * 272 + 128) * 8;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&spd_inverse_mfma_kernel<NT, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 20; ++rep) {
    hipMemset(dC, 0, 160 * 8);
    hipEventRecord(e0);
    spd_inverse_mfma_kernel<NT, 0><<<1, 1024, lds>>>(dA, nullptr, nullptr, D, D, 0, dX, D, 0, dF);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
  }
  if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
  long long c[160]; hipMemcpy(c, dC, sizeof(c), hipMemcpyDeviceToHost);
  hipMemcpy(X.data(), dX, X.size() * 8, hipMemcpyDeviceToHost);
  double res = 0;
  for (int i = 0; i < D; ++i)
    for (int j = 0; j < D; ++j) {
      double x = 0;
      for (int k = 0; k < D; ++k) x += X[(size_t)i * D + k] * A[(size_t)k * D + j];
      res = fmax(res, fabs(x - (i == j ? 1.0 : 0.0)));
    }
  printf("D=%d kernel %.1f us (best of 20), residual %.2e; last run: %.1f us by the 100 MHz counter, %lld clock64 ticks (%.2f GHz)\n", D, best * 1e3, res,
         (c[129] - c[128]) * 0.01, c[131] - c[130], (c[131] - c[130]) / ((c[129] - c[128]) * 10.0));
  const char *names[8] = {"", "A panel | tile", "barrier 1", "U update|factor", "barrier 2", "", "", ""};
  printf("%-16s", "cycles");
  for (int w = 0; w < 16; w += (w < 12 ? 3 : 1)) printf(" wave%-6d", w);
  printf("\n");
  for (int p = 1; p < 5; ++p) {
    printf("%-16s", names[p]);
    for (int w = 0; w < 16; w += (w < 12 ? 3 : 1)) printf(" %10lld", c[w * 8 + p]);
    printf("\n");
  }
  printf("pivot block (wave 15): exchange %lld, 4 x 4 factor %lld, row of C + R' + MFMAs %lld cycles\n", c[137], c[138], c[139]);
  return 0;
}
