// Probe: latency of an in-launch hand-off (producer stores an 8-byte word, consumer polls it) while every CU streams HBM,
// by the consumer's polling path: vector sc1 loads (what the sweeps used) vs SCALAR loads (s_load_dwordx2 glc: scalar cache
// bypassed, a separate path into the L2 from the CU's vector memory queue) on fine-grained / uncached / plain allocations.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_poll.cpp -o tools/_build/probe_poll && tools/_build/probe_poll
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ long long now() { return (long long)__builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ unsigned long long vload(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long sload(const unsigned long long* p) {
  unsigned long long v;
  __asm__ volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}

typedef int i16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ unsigned long long sload16(const unsigned long long* p) {
  i16v v;
  __asm__ volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return ((unsigned long long)(unsigned)v[1] << 32) | (unsigned)v[0];
}
// block roles: b < npairs: producer of pair b; b < 2 npairs: consumer of pair b - npairs; else streamer
// mode 0: vector sc1 poll; 1: scalar poll
__global__ __launch_bounds__(256) void probe(unsigned long long* words, long long* t_store, long long* t_seen, int npairs, int iters,
                                             const double* big, size_t big_n, double* sink, int stream_on, int mode, long long t_len, int shift) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const long long t0 = now();
  if (b < npairs) {
    if (tid == 0)
      for (int i = 1; i <= iters; i++) {
        while (now() - t0 < 1000 + (long long)i * 600) {}   // every 6 us
        t_store[b * iters + i - 1] = now();
        __hip_atomic_store(words + b * 16, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    // the rest of the block streams like everyone else (producer CU loaded)
  } else if (b < 2 * npairs) {
    const int pr = (b - npairs + shift) % npairs;   // shift = 0: producer and consumer on the same XCD (blocks b, b + 32); 3: on different XCDs
    if (tid == 0)
      for (int i = 1; i <= iters; i++) {
        const unsigned long long* w = words + pr * 16;
        bool seen = true;
        if (mode == 0) { while (vload(w) < (unsigned long long)i) { if (now() - t0 > t_len) { seen = false; break; } __builtin_amdgcn_s_sleep(2); } }
        else if (mode == 1) { while (sload(w) < (unsigned long long)i) { if (now() - t0 > t_len) { seen = false; break; } __builtin_amdgcn_s_sleep(2); } }
        else { while (sload16(w) < (unsigned long long)i) { if (now() - t0 > t_len) { seen = false; break; } __builtin_amdgcn_s_sleep(2); } }
        if (seen) t_seen[pr * iters + i - 1] = now();
      }
  }
  if (!stream_on) return;
  if (b < 2 * npairs && tid < 64) return;   // (the polling / producing wave does not stream)
  // stream: 8 loads in flight per thread until t_len has passed
  double acc = 0;
  size_t e = ((size_t)b * 256 + tid) % big_n;
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
  const int iters = 20;
  for (int npairs = 32; npairs <= 256; npairs *= 8) {
  printf("---- %d producer / consumer pairs\n", npairs);
  const int grid = 256 * 3;
  unsigned long long* words[3];
  const char* mname[3] = {"vector", "scalar x2", "scalar x16"};
  const char* names[3] = {"plain hipMalloc", "fine-grained", "uncached"};
  CK(hipMalloc(&words[0], 8192 * 8));
  if (hipExtMallocWithFlags((void**)&words[1], 4096 * 8, hipDeviceMallocFinegrained) != hipSuccess) { words[1] = nullptr; printf("fine-grained alloc failed\n"); }
  if (hipExtMallocWithFlags((void**)&words[2], 4096 * 8, hipDeviceMallocUncached) != hipSuccess) { words[2] = nullptr; printf("uncached alloc failed\n"); }
  long long *ts, *tn;
  CK(hipMalloc(&ts, sizeof(long long) * npairs * iters));
  CK(hipMalloc(&tn, sizeof(long long) * npairs * iters));
  const size_t big_n = (size_t)1 << 28;   // 2 GB
  double *big, *sink;
  CK(hipMalloc(&big, big_n * 8));
  CK(hipMalloc(&sink, 8));
  CK(hipMemset(big, 0, big_n * 8));
  std::vector<long long> hs(npairs * iters), hn(npairs * iters);
  for (int a = 0; a < 1; a++) {
    if (!words[a]) continue;
    for (int shift = 0; shift <= 3; shift += 3)
    for (int stream_on = 0; stream_on < 2; stream_on++)
      for (int mode = 0; mode < 3; mode++) {
        CK(hipMemset(words[a], 0, 8192 * 8));
        CK(hipMemset(tn, 0, sizeof(long long) * npairs * iters));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, words[a], ts, tn, npairs, iters, big, big_n, sink, stream_on, mode, (long long)(1000 + (iters + 2) * 600), shift);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("%s stream %d mode %d: %s\n", names[a], stream_on, mode, hipGetErrorString(e)); return 1; }
        CK(hipMemcpy(hs.data(), ts, sizeof(long long) * npairs * iters, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hn.data(), tn, sizeof(long long) * npairs * iters, hipMemcpyDeviceToHost));
        std::vector<double> lat;
        int missed = 0;
        for (int p = 0; p < npairs; p++)
          for (int i = 2; i < iters; i++) {
            if (hn[p * iters + i] == 0) { missed++; continue; }
            lat.push_back((hn[p * iters + i] - hs[p * iters + i]) / 100.0);
          }
        std::sort(lat.begin(), lat.end());
        if (lat.empty()) { printf("%-16s shift %d streaming %d  %s poll: nothing seen (missed %d)\n", names[a], shift, stream_on, mname[mode], missed); continue; }
        printf("%-16s shift %d streaming %d  %s poll: median %.2f us  p10 %.2f  p90 %.2f  max %.2f  (n %zu, missed %d)\n", names[a], shift, stream_on,
               mname[mode], lat[lat.size() / 2], lat[lat.size() / 10], lat[lat.size() * 9 / 10], lat.back(), lat.size(), missed);
        (void)0;
      }
  }
  hipFree(words[0]); hipFree(words[1]); hipFree(words[2]); hipFree(ts); hipFree(tn); hipFree(big); hipFree(sink);
  }
  return 0;
}
