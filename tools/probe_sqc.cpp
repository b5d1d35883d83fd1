// Probe: throughput of the scalar memory path for uncached (glc) 64-byte loads -- how long one poll round (NB loads in flight,
// one wait) takes when W waves per CU poll side by side while the rest of the CU streams HBM.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_sqc.cpp -o /tmp/probe_sqc && /tmp/probe_sqc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef int i16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ long long now() { return (long long)__builtin_amdgcn_s_memrealtime(); }
template <int NB> __device__ __forceinline__ int poll(const double* p0, const double* p1, const double* p2, const double* p3) {
  i16v a, b, c, d;
  if (NB == 1) { __asm__ volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(a) : "s"(p0) : "memory"); return a[0]; }
  __asm__ volatile("s_load_dwordx16 %0, %4, 0x0 glc\n\ts_load_dwordx16 %1, %5, 0x0 glc\n\ts_load_dwordx16 %2, %6, 0x0 glc\n\ts_load_dwordx16 %3, %7, 0x0 glc\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(p0), "s"(p1), "s"(p2), "s"(p3) : "memory");
  return a[0] + b[0] + c[0] + d[0];
}
// grid = 768 blocks of 256 threads (3 per CU); in every block the first `wpoll` waves poll, the others stream
template <int NB>
__global__ __launch_bounds__(256) void probe(const double* words, long long* rounds, int wpoll, const double* big, size_t big_n, double* sink, long long t_len, int stream_on) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long t0 = now();
  if (wave < wpoll) {
    const double* p = words + ((size_t)(blockIdx.x * 4 + wave) * 4) * 8;   // four lines of its own
    long long n = 0;
    int s = 0;
    while (now() - t0 < t_len) { s += poll<NB>(p, p + 8, p + 16, p + 24); n++; }
    if ((threadIdx.x & 63) == 0) rounds[blockIdx.x * 4 + wave] = n + (s == 12345 ? 1 : 0);
    return;
  }
  if (!stream_on) return;
  double acc = 0;
  size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) % big_n;
  const size_t stride = (size_t)gridDim.x * 256;
  while (now() - t0 < t_len) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { t[u] = big[e]; e += stride; if (e >= big_n) e -= big_n; }
#pragma unroll
    for (int u = 0; u < 8; u++) acc += t[u];
  }
  if (acc == 1.2345e300) sink[0] = acc;
}
int main() {
  const int grid = 768;
  double *words, *big, *sink;
  long long* rounds;
  CK(hipMalloc(&words, (size_t)grid * 4 * 4 * 64));
  CK(hipMemset(words, 0, (size_t)grid * 4 * 4 * 64));
  CK(hipMalloc(&rounds, sizeof(long long) * grid * 4));
  const size_t big_n = (size_t)1 << 28;
  CK(hipMalloc(&big, big_n * 8));
  CK(hipMemset(big, 0, big_n * 8));
  CK(hipMalloc(&sink, 8));
  static long long h[768 * 4];
  const long long t_len = 5000;   // 50 us
  for (int stream_on = 0; stream_on < 2; stream_on++)
    for (int nb = 1; nb <= 4; nb += 3)
      for (int wpoll = 1; wpoll <= 4; wpoll++) {
        CK(hipMemset(rounds, 0, sizeof(long long) * grid * 4));
        if (nb == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, words, rounds, wpoll, big, big_n, sink, t_len, stream_on);
        else hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(256), 0, 0, words, rounds, wpoll, big, big_n, sink, t_len, stream_on);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, rounds, sizeof(h), hipMemcpyDeviceToHost));
        double tot = 0; int cnt = 0; long long mn = 1LL << 60;
        for (int i = 0; i < grid * 4; i++) if (h[i] > 0) { tot += h[i]; cnt++; if (h[i] < mn) mn = h[i]; }
        printf("streaming %d  %d loads in flight  %2d polling waves per CU: round %.2f us (slowest wave %.2f us)\n", stream_on, nb, 3 * wpoll,
               50.0 / (tot / cnt), 50.0 / (double)mn);
      }
  return 0;
}
