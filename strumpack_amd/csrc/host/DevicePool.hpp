// Process-wide pool of device chunks, shared by the HSS and BLR engines: arenas return their chunks here instead of hipFree, so
// that repeated constructions (solver loops, the fronts of a multifrontal factorization, benchmarks) do not pay hipMalloc /
// hipFree page-table work -- a free of a multi-gigabyte block is a synchronous call of tens of milliseconds.
#pragma once
#include <algorithm>
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
    ensure_limit();
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
  // bytes currently kept for reuse / the cap on them; set_limit frees chunks, largest first, until the cache is within the new
  // cap (the small chunks, which a running solver loop asks for most often, stay)
  size_t cached() { std::lock_guard<std::mutex> g(mu_); return cached_; }
  size_t limit() { std::lock_guard<std::mutex> g(mu_); ensure_limit(); return limit_; }
  void set_limit(size_t bytes) {
    std::lock_guard<std::mutex> g(mu_);
    limit_ = bytes;
    limit_set_ = true;
    while (cached_ > limit_ && !free_.empty()) {
      auto it = std::prev(free_.end());   // (the map is keyed by chunk size)
      if (it->second.empty()) { free_.erase(it); continue; }
      hssk_free(it->second.back());
      it->second.pop_back();
      cached_ -= it->first;
      if (it->second.empty()) free_.erase(it);
    }
  }
  ~DevicePool() { for (auto& kv : free_) for (void* p : kv.second) hssk_free(p); }

 private:
  std::mutex mu_;
  std::map<size_t, std::vector<void*>> free_;
  // Bytes kept for reuse.  The cache is only ever trimmed by this library (on its own failed allocation, set_limit, trim), so
  // what it holds is invisible to the other allocators of the process (torch, RCCL): the default cap is a FIFTH of the device's
  // memory -- of the device that is current on the thread that first releases a chunk; the cap is one number for the process,
  // whichever devices' chunks are cached (one process per GPU is how this library is run; a process that drives several
  // devices sets the cap itself) -- (57 of the 288 GB of an MI355X -- the working array + block products of a 60000-row BLR front are ~48 GB: with less,
  // every factorization of such a front pays a hipFree and a hipMalloc of tens of GB, half a second --, 13 GB of a 64 GB part), STRUMPACK_AMD_POOL_GB overrides it, SPX_device_pool_trim / SPX_device_pool_set_limit_gb manage it at run time.
  void ensure_limit() {
    if (limit_set_) return;
    limit_set_ = true;
    const char* e = std::getenv("STRUMPACK_AMD_POOL_GB");
    if (e) { limit_ = (size_t)std::max(0, std::atoi(e)) << 30; return; }
    const long long tot = hssk_device_total_bytes();
    limit_ = tot > 0 ? (size_t)tot / 5 : (size_t)8 << 30;
  }
  size_t cached_ = 0, limit_ = 0;
  bool limit_set_ = false;
};

namespace HSS { using strumpack::DevicePool; }

}  // namespace strumpack
