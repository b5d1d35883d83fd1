// Probe: host-to-device rates on this box -- pinned buffer, pageable buffer through the runtime, two streams.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t GB = size_t(1) << 30, N = 4 * GB;
  char* d; CK(hipMalloc(&d, N));
  char* hp; CK(hipHostMalloc(&hp, N, hipHostMallocDefault));
  memset(hp, 1, N);
  hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now(); CK(hipMemcpyAsync(d, hp, N, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0));
    double dt = now() - t0; if (rep) printf("pinned, one 4 GB copy:              %.1f GB/s\n", N / dt * 1e-9);
  }
  for (size_t piece : {size_t(64) << 20, size_t(256) << 20}) {
    double t0 = now();
    for (size_t o = 0; o < N; o += piece) CK(hipMemcpyAsync(d + o, hp + o, piece, hipMemcpyHostToDevice, s0));
    CK(hipStreamSynchronize(s0));
    printf("pinned, %4zu MB pieces, one stream:   %.1f GB/s\n", piece >> 20, N / (now() - t0) * 1e-9);
    t0 = now();
    int k = 0;
    for (size_t o = 0; o < N; o += piece, k++) CK(hipMemcpyAsync(d + o, hp + o, piece, hipMemcpyHostToDevice, (k & 1) ? s1 : s0));
    CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
    printf("pinned, %4zu MB pieces, two streams:  %.1f GB/s\n", piece >> 20, N / (now() - t0) * 1e-9);
  }
  char* pg = (char*)malloc(N); memset(pg, 2, N);
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now(); CK(hipMemcpy(d, pg, N, hipMemcpyHostToDevice));
    double dt = now() - t0; if (rep) printf("pageable through hipMemcpy:          %.1f GB/s\n", N / dt * 1e-9);
  }
  // host memcpy rate pageable -> pinned with T threads (what a packing pool can deliver)
  for (int T : {8, 16, 24, 32, 48}) {
    double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&, t] { size_t c = N / T; memcpy(hp + c * t, pg + c * t, c); });
    for (auto& x : th) x.join();
    printf("host memcpy pageable -> pinned, %2d threads: %.1f GB/s\n", T, N / (now() - t0) * 1e-9);
  }
  printf("hardware threads: %u\n", std::thread::hardware_concurrency());
  return 0;
}
