// Host-side runtime shim: the only place the engine touches the HIP runtime API.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <stdexcept>
#include <string>

#define HSSK_CHECK(call)                                                                      \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " +  \
                               __FILE__ + ":" + std::to_string(__LINE__));                    \
  } while (0)

namespace hssk_rt {
typedef hipStream_t stream_t;
typedef hipEvent_t event_t;
inline void* dev_malloc(size_t bytes) { void* p = nullptr; if (bytes) HSSK_CHECK(hipMalloc(&p, bytes)); return p; }
inline void dev_free(void* p) { if (p) (void)hipFree(p); }
// hipHostMalloc'ed memory is mapped into the device address space at the same address (unified addressing)
inline bool pinned_is_device_visible() { return true; }
inline void* pinned_malloc(size_t bytes) { void* p = nullptr; if (bytes) HSSK_CHECK(hipHostMalloc(&p, bytes, hipHostMallocDefault)); return p; }
inline void pinned_free(void* p) { if (p) (void)hipHostFree(p); }
inline void h2d(void* d, const void* h, size_t bytes, stream_t s) { if (bytes) HSSK_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s)); }
inline void d2h(void* h, const void* d, size_t bytes, stream_t s) { if (bytes) HSSK_CHECK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s)); }
inline void d2d(void* d, const void* s_, size_t bytes, stream_t s) { if (bytes) HSSK_CHECK(hipMemcpyAsync(d, s_, bytes, hipMemcpyDeviceToDevice, s)); }
inline void h2d_2d(void* d, size_t dpitch, const void* h, size_t hpitch, size_t width, size_t height, stream_t s) {
  if (width && height) HSSK_CHECK(hipMemcpy2DAsync(d, dpitch, h, hpitch, width, height, hipMemcpyHostToDevice, s));
}
inline void d2h_2d(void* h, size_t hpitch, const void* d, size_t dpitch, size_t width, size_t height, stream_t s) {
  if (width && height) HSSK_CHECK(hipMemcpy2DAsync(h, hpitch, d, dpitch, width, height, hipMemcpyDeviceToHost, s));
}
inline void memset_async(void* d, int v, size_t bytes, stream_t s) { if (bytes) HSSK_CHECK(hipMemsetAsync(d, v, bytes, s)); }
inline void sync(stream_t s) { HSSK_CHECK(hipStreamSynchronize(s)); }
inline void check_launch() { HSSK_CHECK(hipGetLastError()); }
// dynamic LDS above the 64 KB default (gfx950: 160 KB per CU)
template <typename K> inline void allow_dynamic_lds(K kernel, size_t bytes) {
  HSSK_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}
// largest LDS allocation one workgroup can get (static + dynamic; gfx950: 160 KB, gfx90a / gfx942: 64 KB)
inline size_t max_lds_per_workgroup() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 65536; }
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeSharedMemPerBlockOptin, dev) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) { (void)hipGetLastError(); return 65536; }
  }
  return n > 0 ? (size_t)n : 65536;
}
inline stream_t stream_create() { stream_t s; HSSK_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); return s; }
inline void stream_destroy(stream_t s) { (void)hipStreamDestroy(s); }
inline event_t event_create() { event_t e; HSSK_CHECK(hipEventCreate(&e)); return e; }
inline void event_destroy(event_t e) { (void)hipEventDestroy(e); }
inline void event_record(event_t e, stream_t s) { HSSK_CHECK(hipEventRecord(e, s)); }
inline float event_elapsed_ms(event_t a, event_t b) { float ms = 0; HSSK_CHECK(hipEventSynchronize(b)); HSSK_CHECK(hipEventElapsedTime(&ms, a, b)); return ms; }
inline void stream_wait_event(stream_t s, event_t e) { HSSK_CHECK(hipStreamWaitEvent(s, e, 0)); }
inline void event_sync(event_t e) { HSSK_CHECK(hipEventSynchronize(e)); }
// host memory that the DMA engines can read directly (hipHostMalloc / hipHostRegister)
inline bool is_pinned_host_pointer(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}
// compute units of the current device (256 on an MI355X in SPX mode, 32 per partition in CPX mode)
inline int cu_count() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
  return n > 0 ? n : 256;
}
// workgroups of one launch that are certain to be resident together (kernels whose workgroups wait for each other)
inline int coresident_workgroups() { return cu_count(); }
// total HBM of the current device in bytes (0 if unknown)
inline size_t device_total_bytes() {
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return tot;
}
inline int device_count() { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
inline void set_device(int d) { HSSK_CHECK(hipSetDevice(d)); }
inline bool is_device_pointer(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeDevice;
}
}  // namespace hssk_rt
