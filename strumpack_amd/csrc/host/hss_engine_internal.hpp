// Internals shared by the translation units of DeviceHSS (hss_engine.cpp: construction and introspection; hss_sources.cpp:
// operand sources; hss_dist.cpp: subtree ownership and exchanges; hss_compress.cpp / hss_compress_kernel.cpp: compression;
// hss_apply.cpp, hss_factor.cpp, hss_solve.cpp, hss_schur.cpp, hss_io.cpp).  Not part of the public surface.
#pragma once
#include "hss_engine.hpp"
#include "Comm.hpp"
#include "LinearNormal.hpp"
#include "DevicePool.hpp"

#include <unistd.h>
#include <atomic>
#include <functional>
#include <condition_variable>
#include <thread>
#include <exception>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <iostream>
#include <istream>
#include <ostream>
#include <map>
#include <mutex>
#include <random>
#include <stdexcept>

namespace strumpack {
namespace HSS {

inline void ck(int rc) {
  if (rc) throw std::runtime_error(std::string("hssk: ") + hssk_last_error());
}
inline double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// right-hand sides from which mult / solve run their LEAF level as batched MFMA GEMMs over all right-hand sides and only
// the inner levels as the single-launch sweep (STRUMPACK_AMD_HYBRID_NRHS overrides; a value above 64 switches it off)
inline int hybrid_nrhs() {
  static const int v = [] { const char* e = std::getenv("STRUMPACK_AMD_HYBRID_NRHS"); return e ? std::atoi(e) : 12; }();
  return v;
}

// most right-hand sides mult / solve still run through the single-launch sweeps (in groups of 64 inside the launch); beyond,
// the batched launches per level (STRUMPACK_AMD_FUSE_MAX_NRHS overrides)
inline int fuse_max_nrhs() {
  static const int v = [] { const char* e = std::getenv("STRUMPACK_AMD_FUSE_MAX_NRHS"); return e ? std::atoi(e) : 256; }();
  return v;
}

// host random stream of the reference (misc/RandomWrapper.hpp:128-191): engine seeded with 0
struct HostRng {
  std::default_random_engine sj{0};   // SJLT patterns (the reference seeds its generator from the clock, sketch.hpp:266-270)
  std::minstd_rand lin{0};
  LinearNormal linnorm{0};   // lin under a normal distribution (the default), generated on all host threads
  std::mt19937 mer{0};
  std::normal_distribution<double> nd;
  std::uniform_real_distribution<double> ud;
};

// bump allocator over large device chunks
// run fn(0..n-1) on the host's hardware threads (per-node index work of a tree level, host-side gathers).  The threads
// are persistent: a level's work is a few hundred microseconds, starting up to 32 threads per call cost more than that
// (the tree phase of the host-operand path: 13 ms, half of it thread start-up).
class HostPool {
 public:
  static HostPool& get() { static HostPool p; return p; }
  // runs body() on every worker and on the caller; returns when all are done.  One job at a time: a second caller
  // (another matrix on another thread) finds the pool busy and runs its loop alone.
  bool run(const std::function<void()>& body) {
    if (getpid() != pid_) return false;   // (a forked child has the pool object but not its threads: it works alone)
    static thread_local bool inside = false;   // a loop started from inside a pool job runs on its own thread
    if (inside) return false;
    std::unique_lock<std::mutex> own(owner_, std::try_to_lock);
    if (!own.owns_lock() || th_.empty()) return false;
    struct Mark { bool& f; Mark(bool& x) : f(x) { f = true; } ~Mark() { f = false; } } mark(inside);
    {
      std::lock_guard<std::mutex> g(mu_);
      body_ = &body; pending_ = th_.size(); gen_++;
    }
    cv_.notify_all();
    body();
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return pending_ == 0; });
    body_ = nullptr;
    return true;
  }
  ~HostPool() {
    if (getpid() != pid_) { for (auto& t : th_) t.detach(); return; }
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; gen_++; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }

 private:
  HostPool() : pid_(getpid()) {
    const unsigned n = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    for (unsigned t = 1; t < n; t++) th_.emplace_back([this] { loop(); });
  }
  void loop() {
    unsigned long seen = 0;
    for (;;) {
      const std::function<void()>* f;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        f = body_;
      }
      (*f)();
      { std::lock_guard<std::mutex> g(mu_); if (--pending_ == 0) done_.notify_all(); }
    }
  }
  const pid_t pid_;
  std::vector<std::thread> th_;
  std::mutex owner_, mu_;
  std::condition_variable cv_, done_;
  const std::function<void()>* body_ = nullptr;
  unsigned long gen_ = 0;
  size_t pending_ = 0;
  bool stop_ = false;
};
template <class F> void host_parallel_for(size_t n, F&& fn) {
  if (n <= 1) { for (size_t i = 0; i < n; i++) fn(i); return; }
  std::atomic<size_t> next{0};
  std::exception_ptr err;
  std::mutex mu;
  const std::function<void()> body = [&] {
    try {
      for (size_t i = next++; i < n; i = next++) fn(i);
    } catch (...) { std::lock_guard<std::mutex> g(mu); err = std::current_exception(); }
  };
  if (!HostPool::get().run(body)) body();
  if (err) std::rethrow_exception(err);
}

class Arena {
 public:
  explicit Arena(size_t chunk = size_t(64) << 20) : chunk_(chunk) {}
  ~Arena() { reset(); }
  void* alloc(size_t bytes) {
    bytes = (std::max<size_t>(bytes, 8) + 255) & ~size_t(255);
    while (bytes > left_) {
      // try the next chunk kept from before a rewind(), else get a new one
      if (next_ < chunks_.size()) {
        cur_ = (char*)chunks_[next_].first;
        left_ = chunks_[next_].second;
        next_++;
        continue;
      }
      const size_t gran = size_t(64) << 20;
      size_t c = (std::max(chunk_, bytes) + gran - 1) / gran * gran;
      void* p = DevicePool::get().acquire(c);
      if (!p) throw std::runtime_error(std::string("device allocation failed: ") + hssk_last_error());
      chunks_.emplace_back(p, c);
      next_ = chunks_.size();
      cur_ = (char*)p;
      left_ = c;
    }
    void* r = cur_;
    cur_ += bytes;
    left_ -= bytes;
    used_ += bytes;
    return r;
  }
  double* dbl(size_t count) { return (double*)alloc(sizeof(double) * count); }
  int* ints(size_t count) { return (int*)alloc(sizeof(int) * count); }
  // forget all allocations but keep the chunks (caller guarantees the device is done with them)
  void rewind() { next_ = 0; cur_ = nullptr; left_ = 0; used_ = 0; }
  // forget the allocations made since mark() (same guarantee; the chunks stay)
  struct Mark { size_t left, used, next; char* cur; };
  Mark mark() const { return Mark{left_, used_, next_, cur_}; }
  void rewind(const Mark& m) { left_ = m.left; used_ = m.used; next_ = m.next; cur_ = m.cur; }
  void reset() {
    for (auto& c : chunks_) DevicePool::get().release(c.first, c.second);
    chunks_.clear();
    rewind();
  }
  size_t used() const { return used_; }

 private:
  size_t chunk_, left_ = 0, used_ = 0, next_ = 0;
  char* cur_ = nullptr;
  std::vector<std::pair<void*, size_t>> chunks_;
};

// ---------------------------------------------------------------------------------------------
// sample / element sources (hss_sources.cpp)
// ---------------------------------------------------------------------------------------------
struct ElemReq {
  const int* dI;  // device index arrays (may be null: contiguous from i0/j0)
  const int* dJ;
  const std::vector<int>* hI;  // host copies (for host-callback sources)
  const std::vector<int>* hJ;
  int i0, j0, m, n;
  double* dB;
  int ldb;
};

struct DeviceHSS::Source {
  virtual ~Source() {}
  // Srt[r0:r0+dn, :] = (A R)^T, Sct[r0:r0+dn, :] = (A^T R)^T for the sample rows [r0, r0+dn) of Rt
  virtual void sample(DeviceHSS& H, int r0, int dn) = 0;
  virtual void extract(DeviceHSS& H, const std::vector<ElemReq>& reqs) = 0;
  // extract() needs nothing sample() produces or uploads (operand resident on the device, or a formula): the leaves' diagonal
  // blocks are then taken out BEFORE the sketch instead of behind it -- off the latency chain of the tree levels
  virtual bool extract_before_sample() const { return false; }
  // sketch products per round that sample() executes (2; 1 when the operand is symmetric by the caller's word: flop count)
  virtual int products(const DeviceHSS&) const { return 2; }
  // flops of the sample() call just made for dn new samples: per product 2 N^2 dn, or 2 N^2 nnz when the SJLT pattern was
  // streamed -- decided by the route sample() actually took (a source that multiplies with the dense form of the pattern, or
  // copies the second product of a symmetric operand, says so here)
  virtual double sketch_flops(const DeviceHSS& H, int dn) const;
  // scattered entries of the operand among this rank's OWN nodes straight from device memory or a formula (the single-launch
  // tree pass gathers its coupling blocks itself, hssk_tree_inner); false: only extract() can serve them
  virtual bool device_elems(const DeviceHSS&, hssk_elem_src*) const { return false; }
};

}  // namespace HSS
}  // namespace strumpack
