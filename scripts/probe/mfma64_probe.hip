// Development probe: what v_mfma_f64_16x16x4_f64 sustains with nothing else in the way -- 1 or 2 waves per SIMD, NT independent
// accumulators per wave, operands in registers (no LDS, no memory).  Prints TFLOP/s and cycles per MFMA per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/mfma64_probe.hip -o scripts/probe/mfma64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(512) void k(double *out, int iters, double a0, double b0) {
  f64x4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = f64x4{0, 0, 0, 0};
  double a = a0 + threadIdx.x, b[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) b[i] = b0 + i + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[i], acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NT; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NT>
void run(int threads, const char *name) {
  double *out; hipMalloc(&out, 256 * 512 * 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NT><<<256, threads>>>(out, 100, 1.0, 2.0);
  hipDeviceSynchronize();
  hipEventRecord(e0); k<NT><<<256, threads>>>(out, iters, 1.0, 2.0); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = 256.0 * (threads / 64) * (double)iters * NT;
  printf("%s NT=%d threads=%d: %.3f ms  %.1f TFLOP/s  (%.1f ns per MFMA per SIMD = %.1f cycles @2.4GHz)\n", name, NT, threads, ms,
         mf * 2048 / ms / 1e9, ms * 1e6 / (mf / 1024), ms * 1e6 / (mf / 1024) * 2.4);
  hipFree(out);
}
int main() {
  run<13>(256, "1 wave/SIMD");
  run<13>(512, "2 waves/SIMD");
  run<16>(256, "1 wave/SIMD");
  run<16>(512, "2 waves/SIMD");
  run<4>(512, "2 waves/SIMD");
  run<1>(512, "2 waves/SIMD (dependent chain)");
  run<2>(256, "1 wave/SIMD");
  return 0;
}
