// Probe: latency of a data hand-off between workgroups of one launch -- producer stores a double, consumer polls it.
//   chain stride 1: consecutive workgroups (different XCDs under round-robin dispatch); stride 8: same XCD.
//   poll: sc1 loads (agent scope, what the sweeps use)  vs  glc-only loads (L2 of the consumer's XCD).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__device__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ double ld_wg(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ double ld_glc(const double* p) { return __builtin_nontemporal_load(p); }
__device__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
// workgroup b waits for slot[b - stride] (b >= stride), then writes slot[b]; one thread per workgroup does the hand-off
template <int MODE>
__global__ void chain(double* slot, int stride, int* xcd, long long* spins) {
  const int b = blockIdx.x;
  if (threadIdx.x == 0) {
    xcd[b] = (int)xcc_id();
    double v = 1.0;
    long long n = 0;
    if (b >= stride) {
      const double* p = slot + (size_t)(b - stride) * 16;
      for (;;) {
        v = MODE == 0 ? ld_agent(p) : (MODE == 1 ? ld_glc(p) : ((n & 63) == 63 ? ld_agent(p) : ld_wg(p)));
        if (v != 0.0) break;
        if (++n > (1LL << 24)) { v = -1e300; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    spins[b] = n;
    st_agent(slot + (size_t)b * 16, v + 1.0);
  }
}
int main() {
  const int L = 2048;
  double* slot; int* xcd; long long* spins;
  CK(hipMalloc(&slot, sizeof(double) * 16 * L)); CK(hipMalloc(&xcd, sizeof(int) * L)); CK(hipMalloc(&spins, sizeof(long long) * L));
  int hx[L]; long long hs[L]; double last;
  for (int mode = 0; mode < 3; mode++)
    for (int stride : {1, 8}) {
      double best = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(slot, 0, sizeof(double) * 16 * L));
        CK(hipDeviceSynchronize());
        double t0 = now();
        if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(L), dim3(64), 0, 0, slot, stride, xcd, spins);
        else if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(L), dim3(64), 0, 0, slot, stride, xcd, spins);
        else hipLaunchKernelGGL(chain<2>, dim3(L), dim3(64), 0, 0, slot, stride, xcd, spins);
        CK(hipDeviceSynchronize());
        best = std::min(best, now() - t0);
      }
      CK(hipMemcpy(hx, xcd, sizeof(hx), hipMemcpyDeviceToHost)); CK(hipMemcpy(hs, spins, sizeof(hs), hipMemcpyDeviceToHost));
      CK(hipMemcpy(&last, slot + (size_t)(L - 1) * 16, 8, hipMemcpyDeviceToHost));
      int same = 0; long long smax = 0;
      for (int b = stride; b < L; b++) { same += hx[b] == hx[b - stride]; smax = std::max(smax, hs[b]); }
      const int hops = L / stride;   // length of each dependent chain
      printf("poll %-26s stride %d: %8.1f us total, %.2f us per hop (%d hops); same XCD as producer: %d / %d; max spins %lld; last %.0f\n",
             mode == 0 ? "agent scope (sc1)" : (mode == 1 ? "nontemporal" : "workgroup scope + sc1 / 64"), stride, best * 1e6, best * 1e6 / hops, hops, same, L - stride, smax, last);
    }
  printf("xcd of workgroups 0..15:"); for (int b = 0; b < 16; b++) printf(" %d", hx[b]); printf("\n");
  return 0;
}
