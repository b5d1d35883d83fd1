// TaskTimer: the wall-clock stopwatch the reference's drivers use around their phases (misc/TaskTimer.hpp:120-160:
// TaskTimer(name), start(), stop(), elapsed()).  The reference's per-thread task tracing behind the same class is a
// profiling aid of its OpenMP task runtime and has no counterpart here.
#pragma once
#include <chrono>
#include <string>

namespace strumpack {

class TaskTimer {
 public:
  explicit TaskTimer(const std::string& name = "", int depth = 1) : name_(name) { (void)depth; }
  void start() { t0_ = clock::now(); running_ = true; }
  void stop() { if (running_) { t1_ = clock::now(); running_ = false; } }
  // seconds since start() (up to now while running, up to stop() afterwards)
  double elapsed() {
    const auto end = running_ ? clock::now() : t1_;
    return std::chrono::duration<double>(end - t0_).count();
  }
  const std::string& name() const { return name_; }

 private:
  using clock = std::chrono::steady_clock;
  std::string name_;
  clock::time_point t0_ = clock::now(), t1_ = clock::now();
  bool running_ = false;
};

}  // namespace strumpack
