// Development probe (round 4): what the fp32 matrix pipe SUSTAINS on this chip with nothing else in the way -- operands in
// registers (no LDS, no memory), 1 wave per SIMD, 256 accumulator registers, launched back to back for ~1.5 s per arm so that
// the power management has settled.  Three instruction shapes of the same 64 flop / cycle / SIMD peak, and two kinds of
// operand data (constants that barely toggle / pseudo-random mantissas).  Prints TFLOP/s (wall clock), MFMA cycles per
// instruction (shader clock, s_memtime) and the shader clock itself (s_memtime against s_memrealtime, 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/mfma32_power_probe.hip -o scripts/probe/mfma32_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rnd(unsigned &s) {
  s = s * 1664525u + 1013904223u;
  return __uint_as_float(0x3f800000u | (s >> 9)) - 1.5f;   // [-0.5, 0.5), random mantissa
}

// KIND 0: v_mfma_f32_32x32x2_f32, 16 blocks of 16 registers; 1: v_mfma_f32_16x16x4_f32, 64 blocks of 4 registers
template <int KIND, bool RANDOM>
__global__ __launch_bounds__(256, 1) void k(float *out, int iters, unsigned long long *stamp) {
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 17u;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = RANDOM ? rnd(s) : 1.0f; b[i] = RANDOM ? rnd(s) : 0.5f; }
  unsigned long long c0 = 0, r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  float sum = 0.f;
  if (KIND == 0) {
    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i >> 2) + (t & 1) * 4], b[(i & 3) + (t >> 1) * 4], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) sum += acc[i][e];
  } else {
    f32x4 acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 64; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i >> 3)], b[(i & 7) ^ t], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 64; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    stamp[0] = __builtin_amdgcn_s_memtime() - c0;
    stamp[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int KIND, bool RANDOM>
void run(const char *name, double seconds) {
  float *out; hipMalloc(&out, 256 * 256 * 4);
  unsigned long long *stamp; hipMalloc(&stamp, 16);
  const int iters = 40000;                                // 64 (KIND 0) or 128 (KIND 1) MFMAs per iteration per wave
  const double flop_per_launch = 256.0 * 4 * (double)iters * (KIND == 0 ? 64 * 4096.0 : 128 * 2048.0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND, RANDOM><<<256, 256>>>(out, 100, stamp);
  hipDeviceSynchronize();
  // sustained: launches back to back until `seconds` have passed; report the first and the last launch
  double first = 0, last = 0, total = 0; int n = 0;
  unsigned long long hs[2] = {0, 0};
  while (total < seconds * 1e3) {
    hipEventRecord(e0); k<KIND, RANDOM><<<256, 256>>>(out, iters, stamp); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (n == 0) first = ms;
    last = ms; total += ms; ++n;
  }
  hipMemcpy(hs, stamp, 16, hipMemcpyDeviceToHost);
  const double mfma_per_simd = (double)iters * (KIND == 0 ? 64 : 128);
  printf("%-28s %3d launches: first %.2f ms %.1f TFLOP/s | last %.2f ms %.1f TFLOP/s (%.1f %% of 157.3) | %.2f shader cycles per MFMA, shader clock %.0f MHz\n",
         name, n, first, flop_per_launch / first / 1e9, last, flop_per_launch / last / 1e9, flop_per_launch / last / 1e9 / 157.3 * 100,
         (double)hs[0] / mfma_per_simd, (double)hs[0] / ((double)hs[1] / 100.0));
  hipFree(out); hipFree(stamp);
}

int main(int argc, char **argv) {
  const double sec = argc > 1 ? atof(argv[1]) : 1.5;
  run<0, false>("32x32x2 constants", sec);
  run<0, true>("32x32x2 random", sec);
  run<1, false>("16x16x4 constants", sec);
  run<1, true>("16x16x4 random", sec);
  run<0, true>("32x32x2 random (again)", sec);
  system("rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\\|power' | head -4");
  return 0;
}
