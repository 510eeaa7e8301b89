// Development probe: (1) which XCD does workgroup i of a launch run on, (2) does a store made with sc0 on one CU
// become visible to sc0 loads of another CU of the same XCD, and how long does a ping-pong take (sc0 vs sc1)?
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/xcd_probe.hip -o scripts/probe/xcd_probe && scripts/probe/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void where_kernel(int *xcc, int *cu) {
  if (threadIdx.x == 0) {
    xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    cu[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
  }
}

template <int MODE>   // 0: sc0, 1: sc1 (agent), 2: plain
__device__ __forceinline__ void st(unsigned long long *p, unsigned long long v) {
  if (MODE == 0) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  else if (MODE == 1) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
template <int MODE>
__device__ __forceinline__ unsigned long long ld(const unsigned long long *p) {
  unsigned long long v;
  if (MODE == 0) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  else if (MODE == 1) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}

// workgroups a and b (both on XCD `want`, found through HW_REG_XCC_ID) play ping-pong on two words
template <int MODE>
__global__ void pingpong_kernel(unsigned long long *words, int *roles, int want, int rounds, long long *cycles, int *status) {
  __shared__ int role;
  if (threadIdx.x == 0) {
    const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    role = -1;
    if (xcc == want) role = atomicAdd(roles, 1);
  }
  __syncthreads();
  if (role < 0 || role > 1 || threadIdx.x != 0) return;
  unsigned long long *mine = words + role * 64, *theirs = words + (1 - role) * 64;
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 1; r <= rounds; ++r) {
    if (role == 0) st<MODE>(mine, r);
    int polls = 0;
    while (ld<MODE>(theirs) < (unsigned long long)r) {
      if (++polls > (1 << 22)) { status[role] = r; return; }
    }
    if (role == 1) st<MODE>(mine, r);
  }
  cycles[role] = __builtin_readcyclecounter() - t0;
  status[role] = 0;
}

template <int MODE>
void run(const char *name, int want) {
  unsigned long long *words; int *roles, *status; long long *cycles;
  hipMalloc(&words, 1024); hipMalloc(&roles, 64); hipMalloc(&status, 64); hipMalloc(&cycles, 64);
  hipMemset(words, 0, 1024); hipMemset(roles, 0, 64); hipMemset(status, 0xff, 64); hipMemset(cycles, 0, 64);
  const int rounds = 1000;
  pingpong_kernel<MODE><<<64, 64>>>(words, roles, want, rounds, cycles, status);
  hipDeviceSynchronize();
  int hs[2]; long long hc[2]; int hr;
  hipMemcpy(hs, status, 8, hipMemcpyDeviceToHost); hipMemcpy(hc, cycles, 16, hipMemcpyDeviceToHost); hipMemcpy(&hr, roles, 4, hipMemcpyDeviceToHost);
  printf("%-6s XCD %d: workgroups on it %d, status %d %d, cycles per round trip %.0f (s_memtime ticks at 100 MHz -> %.2f us)\n", name, want, hr, hs[0], hs[1],
         (double)hc[0] / rounds, (double)hc[0] / rounds / 100.0);
}

int main() {
  const int nb = 64;
  int *xcc, *cu;
  hipMalloc(&xcc, nb * 4); hipMalloc(&cu, nb * 4);
  where_kernel<<<nb, 64>>>(xcc, cu);
  std::vector<int> hx(nb), hc(nb);
  hipMemcpy(hx.data(), xcc, nb * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hc.data(), cu, nb * 4, hipMemcpyDeviceToHost);
  printf("XCC_ID of workgroups 0..%d:", nb - 1);
  for (int i = 0; i < nb; ++i) printf(" %d", hx[i]);
  printf("\nHW_ID:");
  for (int i = 0; i < 16; ++i) printf(" %08x", hc[i]);
  printf("\n");
  run<0>("sc0", 0);
  run<1>("sc1", 0);
  run<2>("plain", 0);
  run<0>("sc0", 3);
  return 0;
}
