// Development probe: device -> host copy of 1.6 GB into (a) pageable memory, (b) the same memory after
// hipHostRegister (with the cost of registering), (c) hipHostMalloc'ed memory.
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/pcie_probe.hip -o scripts/probe/pcie_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t bytes = (size_t)1600 << 20;
  void *d; hipMalloc(&d, bytes); hipMemset(d, 1, bytes);
  char *hp = (char *)aligned_alloc(4096, bytes); memset(hp, 0, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now(); hipMemcpy(hp, d, bytes, hipMemcpyDeviceToHost); double t1 = now();
    printf("pageable D2H: %.1f ms = %.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
  }
  double t0 = now(); hipError_t e = hipHostRegister(hp, bytes, hipHostRegisterDefault); double t1 = now();
  printf("hipHostRegister: %s, %.1f ms\n", hipGetErrorString(e), (t1 - t0) * 1e3);
  for (int rep = 0; rep < 2; ++rep) {
    t0 = now(); hipMemcpy(hp, d, bytes, hipMemcpyDeviceToHost); t1 = now();
    printf("registered D2H: %.1f ms = %.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
  }
  t0 = now(); hipHostUnregister(hp); t1 = now();
  printf("hipHostUnregister: %.1f ms\n", (t1 - t0) * 1e3);
  void *pin; t0 = now(); hipHostMalloc(&pin, (size_t)256 << 20, hipHostMallocDefault); t1 = now();
  printf("hipHostMalloc 256 MB: %.1f ms\n", (t1 - t0) * 1e3);
  t0 = now(); hipMemcpy(pin, d, (size_t)256 << 20, hipMemcpyDeviceToHost); t1 = now();
  printf("pinned D2H 256 MB: %.1f ms = %.1f GB/s\n", (t1 - t0) * 1e3, ((size_t)256 << 20) / (t1 - t0) / 1e9);
  t0 = now(); memcpy(hp, pin, (size_t)256 << 20); t1 = now();
  printf("host memcpy 256 MB (1 thread): %.1f ms = %.1f GB/s\n", (t1 - t0) * 1e3, ((size_t)256 << 20) / (t1 - t0) / 1e9);
  return 0;
}
