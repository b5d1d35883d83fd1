// TEST INFRASTRUCTURE ONLY.  CPU stand-in for strumpack_amd/csrc/hip/hssk_rt.h ("device" memory
// is host memory, streams are no-ops).  See hssk_device.h in this directory.
#pragma once
#include <thread>
#include <cstdlib>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

namespace hssk_rt {
typedef void* stream_t;
typedef double* event_t;
inline void* dev_malloc(size_t bytes) { return bytes ? std::malloc(bytes) : nullptr; }
inline void dev_free(void* p) { std::free(p); }
inline bool pinned_is_device_visible() { return true; }
inline void* pinned_malloc(size_t bytes) { return bytes ? std::malloc(bytes) : nullptr; }
inline void pinned_free(void* p) { std::free(p); }
inline void h2d(void* d, const void* h, size_t bytes, stream_t) { if (bytes) std::memcpy(d, h, bytes); }
inline void d2h(void* h, const void* d, size_t bytes, stream_t) { if (bytes) std::memcpy(h, d, bytes); }
inline void d2d(void* d, const void* s, size_t bytes, stream_t) { if (bytes) std::memmove(d, s, bytes); }
inline void h2d_2d(void* d, size_t dpitch, const void* h, size_t hpitch, size_t width, size_t height, stream_t) {
  for (size_t r = 0; r < height; r++) std::memcpy((char*)d + r * dpitch, (const char*)h + r * hpitch, width);
}
inline void d2h_2d(void* h, size_t hpitch, const void* d, size_t dpitch, size_t width, size_t height, stream_t) {
  for (size_t r = 0; r < height; r++) std::memcpy((char*)h + r * hpitch, (const char*)d + r * dpitch, width);
}
inline void memset_async(void* d, int v, size_t bytes, stream_t) { if (bytes) std::memset(d, v, bytes); }
inline void sync(stream_t) {}
inline void check_launch() {}
template <typename K> inline void allow_dynamic_lds(K, size_t) {}
inline size_t max_lds_per_workgroup() { return 160 * 1024; }
inline stream_t stream_create() { return nullptr; }
inline void stream_destroy(stream_t) {}
inline event_t event_create() { return new double(0); }
inline void event_destroy(event_t e) { delete e; }
inline void event_record(event_t e, stream_t) {
  *e = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline float event_elapsed_ms(event_t a, event_t b) { return (float)(*b - *a); }
inline void stream_wait_event(stream_t, event_t) {}
inline void event_sync(event_t) {}
inline bool is_pinned_host_pointer(const void*) { return false; }
inline int cu_count() { return 256; }
// the emulator runs a launch's workgroups on a pool of host threads (emu_runtime.cpp: at most 16)
inline int coresident_workgroups() {
  const char* e = std::getenv("HSSK_EMU_THREADS");
  const int nt = e ? std::atoi(e) : (int)std::thread::hardware_concurrency();
  return nt < 16 ? nt : 16;
}
inline size_t device_total_bytes() { return (size_t)16 << 30; }
inline int device_count() { return 1; }
inline void set_device(int) {}
inline bool is_device_pointer(const void*) { return false; }
}  // namespace hssk_rt
