// Development probe: the inner loop of the K4 transform kernel without its staging -- per k-step one A fragment and NT B
// fragments read from LDS (conflict-free pitch), NT v_mfma_f64_16x16x4_f64 -- 2 waves per SIMD, with and without a
// barrier every 4 k-steps, fragments read in their own k-step (MODE 0) or one k-step ahead on a second register set (MODE 1).
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/mfma64_lds_probe.hip -o scripts/probe/mfma64_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int LD = 18;
template <int NT, int MODE, bool BAR>
__global__ __launch_bounds__(512) void k(double *out, int iters) {
  extern __shared__ double lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fi = lane & 15, fk = lane >> 4;
  for (int i = t; i < (NT * 16 + 128) * LD; i += 512) lds[i] = 1.0 + (i % 7) * 1e-3;
  __syncthreads();
  f64x4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = f64x4{0, 0, 0, 0};
  const double *Ts = lds + fi * LD + fk, *Xs = lds + (NT * 16 + wave * 16 + fi) * LD + fk;
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const double a = Xs[kk * 4];
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) {
          const double b = Ts[tn * 16 * LD + kk * 4];
          acc[tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[tn], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
      }
      if (BAR) __syncthreads();
    }
  } else {
    double fa[2], fb[2][NT];
    auto frag = [&](int s, int kk) {
      fa[s] = Xs[kk * 4];
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) fb[s][tn] = Ts[tn * 16 * LD + kk * 4];
    };
    frag(0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (!(BAR && kk == 3)) frag((kk + 1) & 1, (kk + 1) & 3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) acc[tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[kk & 1], fb[kk & 1][tn], acc[tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (BAR) { __syncthreads(); frag(0, 0); }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NT; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + t] = s;
}
template <int NT, int MODE, bool BAR>
void run(const char *name) {
  double *out; hipMalloc(&out, 256 * 512 * 8);
  const int iters = 4000;
  const size_t sm = (size_t)(NT * 16 + 128) * LD * 8;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k<NT, MODE, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NT, MODE, BAR><<<256, 512, sm>>>(out, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0); k<NT, MODE, BAR><<<256, 512, sm>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = 256.0 * 8 * (double)iters * 4 * NT;
  printf("%-44s NT=%2d: %.3f ms  %.1f TFLOP/s = %.3f of 78.6\n", name, NT, ms, mf * 2048 / ms / 1e9, mf * 2048 / ms / 1e9 / 78.6);
  hipFree(out);
}
int main() {
  run<13, 0, false>("reads in their k-step, no barrier");
  run<13, 0, true>("reads in their k-step, barrier / 4 k-steps");
  run<13, 1, false>("reads one k-step ahead, no barrier");
  run<13, 1, true>("reads one k-step ahead, barrier / 4 k-steps");
  run<16, 0, false>("reads in their k-step, no barrier");
  run<16, 0, true>("reads in their k-step, barrier / 4 k-steps");
  run<16, 1, false>("reads one k-step ahead, no barrier");
  run<16, 1, true>("reads one k-step ahead, barrier / 4 k-steps");
  return 0;
}
