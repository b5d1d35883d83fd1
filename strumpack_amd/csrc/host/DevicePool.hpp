// Process-wide pool of device chunks, shared by the HSS and BLR engines: arenas return their chunks here instead of hipFree, so
// that repeated constructions (solver loops, the fronts of a multifrontal factorization, benchmarks) do not pay hipMalloc /
// hipFree page-table work -- a free of a multi-gigabyte block is a synchronous call of tens of milliseconds.
#pragma once
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "hssk.h"

namespace strumpack {

class DevicePool {
 public:
  static DevicePool& get() { static DevicePool p; return p; }
  void* acquire(size_t bytes) {
    {
      std::lock_guard<std::mutex> g(mu_);
      auto it = free_.find(bytes);
      if (it != free_.end() && !it->second.empty()) { void* p = it->second.back(); it->second.pop_back(); cached_ -= bytes; return p; }
    }
    void* p = hssk_malloc((long long)bytes);
    if (!p) {  // memory pressure: drop the cache and retry once
      trim();
      p = hssk_malloc((long long)bytes);
    }
    return p;
  }
  void release(void* p, size_t bytes) {
    std::lock_guard<std::mutex> g(mu_);
    if (cached_ + bytes > limit_) { hssk_free(p); return; }
    free_[bytes].push_back(p);
    cached_ += bytes;
  }
  void trim() {
    std::lock_guard<std::mutex> g(mu_);
    for (auto& kv : free_) for (void* p : kv.second) hssk_free(p);
    free_.clear();
    cached_ = 0;
  }
  ~DevicePool() { for (auto& kv : free_) for (void* p : kv.second) hssk_free(p); }

 private:
  std::mutex mu_;
  std::map<size_t, std::vector<void*>> free_;
  // bytes kept for reuse (STRUMPACK_AMD_POOL_GB; default 48 of the 288 GB: the working array + block products of a
  // 60000-row BLR front are ~45 GB)
  size_t cached_ = 0, limit_ = [] { const char* e = std::getenv("STRUMPACK_AMD_POOL_GB"); return (size_t)(e ? std::max(0, std::atoi(e)) : 48) << 30; }();
};

namespace HSS { using strumpack::DevicePool; }

}  // namespace strumpack
