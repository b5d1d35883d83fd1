// Probe: can the host store straight into device memory (large BAR), and what does a kernel pay for reading a small
// descriptor from (a) pinned host memory, (b) device memory written by the host through the BAR?
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <immintrin.h>
#include <csetjmp>
#include <csignal>
static sigjmp_buf g_jmp; static volatile int g_armed = 0;
static void on_segv(int) { if (g_armed) siglongjmp(g_jmp, 1); _exit(99); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Desc { const double* src; double* dst; int n; int pad; };
__global__ void chase(const Desc* d, int nd) {
  const Desc p = d[blockIdx.x % nd];
  for (int i = threadIdx.x; i < p.n; i += blockDim.x) p.dst[i] = p.src[i] + 1.0;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int try_write(void* p, const char* what) {
  fflush(stdout);
  pid_t c = fork();
  if (c == 0) { volatile int* q = (volatile int*)p; q[0] = 12345; q[1000] = 7; _exit(q[0] == 12345 ? 0 : 3); }
  int st = 0; waitpid(c, &st, 0);
  printf("%s: child %s (status %d)\n", what, WIFEXITED(st) && WEXITSTATUS(st) == 0 ? "wrote and read back" : "FAILED", st);
  return WIFEXITED(st) && WEXITSTATUS(st) == 0;
}
int main() {
  signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
  int lb = -1;
  CK(hipDeviceGetAttribute(&lb, hipDeviceAttributeIsLargeBar, 0));
  printf("hipDeviceAttributeIsLargeBar = %d\n", lb);
  double *a, *b; CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20));
  CK(hipMemset(a, 0, 1 << 20));
  hipStream_t s; CK(hipStreamCreate(&s));
  Desc* hpin; CK(hipHostMalloc(&hpin, 1 << 16, hipHostMallocDefault));
  Desc* dplain; CK(hipMalloc(&dplain, 1 << 16));
  Desc* dfine = nullptr; hipError_t ef = hipExtMallocWithFlags((void**)&dfine, 1 << 16, hipDeviceMallocFinegrained);
  printf("hipExtMallocWithFlags(Finegrained): %s\n", hipGetErrorString(ef));
  Desc* dunc = nullptr; hipError_t eu = hipExtMallocWithFlags((void**)&dunc, 1 << 16, hipDeviceMallocUncached);
  printf("hipExtMallocWithFlags(Uncached): %s\n", hipGetErrorString(eu));
  // (the fork test runs in a child without a HIP context: it only tells whether the mapping exists in this address space)
  Desc proto{a, b, 256, 0};
  struct { const char* name; Desc* p; bool direct; } arms[] = {{"pinned host (zero copy)", hpin, true}, {"hipMalloc + host store", dplain, true},
                                                              {"fine-grained device + host store", dfine, true}, {"uncached device + host store", dunc, true}};
  for (auto& arm : arms) {
    if (!arm.p) continue;
    if (arm.p != hpin) {
      // does a plain host store work?  (a fault must not kill the probe: SIGSEGV handler + longjmp)
      if (sigsetjmp(g_jmp, 1) == 0) {
        g_armed = 1;
        volatile long long* q = (volatile long long*)arm.p; q[0] = 1;
        g_armed = 0;
      } else {
        g_armed = 0;
        printf("%-36s host store faults -- skipped\n", arm.name);
        continue;
      }
    }
    for (int i = 0; i < 64; i++) arm.p[i] = proto;
    _mm_sfence();
    for (int rep = 0; rep < 3; rep++) {
      CK(hipStreamSynchronize(s));
      const int L = 200;
      double t0 = now();
      for (int l = 0; l < L; l++) {
        arm.p[l % 64].n = 256;   // a fresh store before every launch, as the descriptor ring does
        _mm_sfence();
        hipLaunchKernelGGL(chase, dim3(8), dim3(256), 0, s, arm.p, 64);
      }
      CK(hipStreamSynchronize(s));
      double dt = now() - t0;
      if (rep == 2) printf("%-36s %6.2f us per dependent launch (8 workgroups, 200 launches back to back)\n", arm.name, dt / L * 1e6);
    }
    double chk = 0; CK(hipMemcpy(&chk, b, 8, hipMemcpyDeviceToHost));
    if (chk != 1.0) printf("   WRONG RESULT %g\n", chk);
  }
  return 0;
}
