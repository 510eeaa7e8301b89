// Development probe: the one-barrier tridiagonalisation kernel of csrc/eig_dc.hip (tridiag_full_kernel<13, 7>, n = 200)
// with a clock read at every phase boundary of a Householder step, accumulated in registers and written once.
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iplda_amd/csrc scripts/probe/tridiag_full_probe.hip -o scripts/probe/tridiag_full_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ long long *g_tclock;     // [8 waves][8]
#define TRF_CLOCK(i)                                                                   \
  do {                                                                                 \
    const long long now_ = clock64();                                                  \
    if ((i) > 0) cacc_[i] += now_ - last_;                                             \
    last_ = now_;                                                                      \
  } while (0)
#define TRF_PROBE_LOCALS long long last_ = clock64(), cacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TRF_PROBE_END                                                                  \
  if (lane == 0) for (int i_ = 1; i_ < 8; ++i_) g_tclock[wave * 8 + i_] = cacc_[i_];
#include "eig_dc.hip"
// (the rest of the library is not linked: stubs for what eig_dc.hip's host code refers to)
namespace plda {
int hip_fail(plda_handle *, hipError_t, const char *, const char *, int) { return -1; }
int fail(plda_handle *, int, const char *, ...) { return -1; }
int gemm_f64(plda_handle *, int64_t, int64_t, int64_t, double, const double *, int64_t, int64_t, const double *, int64_t,
             int64_t, const double *, double, double *, int64_t) { return -1; }
int eig_sort_rows(plda_handle *, const double *, const double *, int, double *, double *) { return -1; }
}

int main() {
  using namespace plda;
  const int n = 200;
  std::vector<double> A((size_t)n * n);
  unsigned long long s = 999;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      const double x = ((s >> 11) * (1.0 / 9007199254740992.0)) - 0.5;
      A[(size_t)i * n + j] = A[(size_t)j * n + i] = x;
    }
  double *dA, *dscale, *dd, *ee, *Vh, *tau; long long *dC; int *dflag;
  hipMalloc(&dflag, 4); hipMemset(dflag, 0, 4);
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dscale, 64); hipMalloc(&dd, n * 8); hipMalloc(&ee, n * 8);
  hipMalloc(&Vh, A.size() * 8); hipMalloc(&tau, n * 8); hipMalloc(&dC, 64 * 8);
  const double one = 1.0;
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dscale, &one, 8, hipMemcpyHostToDevice);
  hipMemcpyToSymbol(HIP_SYMBOL(g_tclock), &dC, sizeof(dC));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 10; ++rep) {
    hipEventRecord(e0);
    tridiag_full_kernel<13, 7><<<1, 512>>>(dA, n, dscale, dd, ee, Vh, tau, dflag);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
  }
  long long c[64]; hipMemcpy(c, dC, sizeof(c), hipMemcpyDeviceToHost);
  std::vector<double> d(n), e(n);
  hipMemcpy(d.data(), dd, n * 8, hipMemcpyDeviceToHost); hipMemcpy(e.data(), ee, n * 8, hipMemcpyDeviceToHost);
  double tr = 0, trA = 0, fr = 0, frA = 0;
  for (int i = 0; i < n; ++i) { tr += d[i]; trA += A[(size_t)i * n + i]; fr += d[i] * d[i] + (i + 1 < n ? 2 * e[i] * e[i] : 0); }
  for (double x : A) frA += x * x;
  printf("n=%d kernel %.1f us; trace %.12g / %.12g, Frobenius^2 %.12g / %.12g\n", n, best * 1e3, tr, trA, fr, frA);
  const char *names[6] = {"", "sync", "p = A v", "barrier", "v.p, w, x'", "reflector+update"};
  for (int p = 1; p < 6; ++p) {
    printf("%-12s", names[p]);
    for (int w = 0; w < 8; ++w) printf(" %9lld", c[w * 8 + p]);
    printf("\n");
  }
  return 0;
}
