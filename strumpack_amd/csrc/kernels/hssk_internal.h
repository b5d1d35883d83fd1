// Internal: context object shared by the kernel translation units (not part of the C-ABI).
#pragma once
#include <functional>
#include <vector>
#include <algorithm>
#include <utility>
#include <cstring>
#include <stdexcept>
#include <string>

#include "hssk.h"
#include "hssk_rt.h"

namespace hssk_rec {
extern thread_local std::vector<std::function<void()>>* sink;   // non-null while a plan is being recorded (hssk_device.h)
}

// A recorded sequence of batched launches (one apply / solve sweep) with its descriptor arrays kept in a private
// pinned block: replaying it costs one kernel launch per step and no host-side descriptor work.
struct hssk_plan {
  std::vector<std::function<void()>> launches;
  // descriptor storage of the recorded launches: DEVICE memory, with a pinned shadow the descriptors are staged through (one
  // stream-ordered copy per table while the plan is recorded).  The first version kept the tables in pinned host memory and
  // let the kernels read them in place: every replayed launch then fetched its descriptors across PCIe -- ~15 us on each of
  // the batched launches of a many-right-hand-side sweep, and the first microseconds of every workgroup of the single-launch
  // sweeps.
  struct Block { char* dev; char* shadow; size_t size; };
  std::vector<Block> blocks;
  size_t off = 0;
  char* alloc(size_t bytes, char** shadow) {
    const size_t need = (bytes + 255) & ~size_t(255);
    if (blocks.empty() || off + need > blocks.back().size) {
      const size_t sz = std::max<size_t>(need, size_t(1) << 20);
      blocks.push_back(Block{(char*)hssk_rt::dev_malloc(sz), (char*)hssk_rt::pinned_malloc(sz), sz});
      off = 0;
    }
    char* p = blocks.back().dev + off;
    *shadow = blocks.back().shadow + off;
    off += need;
    return p;
  }
  ~hssk_plan() { for (auto& b : blocks) { hssk_rt::dev_free(b.dev); hssk_rt::pinned_free(b.shadow); } }
};

struct hssk_uploader;
struct hssk_ctx {
  int device = 0;
  hssk_rt::stream_t stream{};
  // descriptor staging ring: pinned host mirror + device buffer
  char* h_ring = nullptr;
  char* d_ring = nullptr;
  size_t ring_bytes = 0, ring_off = 0;
  size_t zero_copy_bytes = 0;   // descriptor arrays up to this size are read in place from the pinned ring
  hssk_rt::event_t ev0{}, ev1{};
  hssk_rt::event_t ev_sync{};   // hssk_stream_wait: recorded on this context's stream, waited for by another's
  // side stream (hssk_side_begin / _end / _join): work that does not depend on what the main stream is doing
  hssk_rt::stream_t side{}, main_saved{};
  hssk_rt::event_t ev_fork{}, ev_join{};
  bool on_side = false, side_made = false;
  bool require_mma = false;   // hssk_sweep_require_mma: the sweeps refuse (code 2) what their matrix-core form cannot take
  bool dgemm_timed = false;
  double dgemm_timed_flops = 0.;  // algorithmic flops of the launch bracketed by ev0 / ev1
  // brackets set aside by hssk_dgemm_timing_defer (read by _collect at the caller's next synchronisation)
  struct TimedBracket { hssk_rt::event_t a, b; double flops; };
  std::vector<TimedBracket> dgemm_deferred;
  long long dgemm_trace_wgs = 0;  // workgroups of the last main launch (trace records behind d_clk + 4)
  long long* d_clk = nullptr;   // device: {shader cycles, 100 MHz ticks} of workgroup 0 of the last dgemm
  double* d_scratch = nullptr;  // split-K partials of hssk_dgemm
  // dependency flags of the single-launch sweeps (device, all zero between launches) and their error word (pinned)
  struct hssk_uploader* uploader = nullptr;   // copy stream + pinned bounce slots of hssk_h2d_block_async (hssk_ctx.hip)
  int* d_sweep_flags = nullptr;
  int* h_sweep_err = nullptr;
  // stopwatches (hssk_watch_*): recorded event pairs per id, recycled through a free list
  std::vector<std::pair<hssk_rt::event_t, hssk_rt::event_t>> watch[8];
  std::vector<hssk_rt::event_t> watch_free;
  bool watch_open[8] = {false, false, false, false, false, false, false, false};
  size_t sweep_cap = size_t(1) << 20;
  size_t scratch_bytes = 0;

  // copies `bytes` of host data into the ring and returns the device address (valid for kernels
  // enqueued on `stream` after this call)
  hssk_plan* recording = nullptr;
  void* stage(const void* host, size_t bytes) {
    if (recording) {   // descriptors of a recorded sweep live as long as the plan, in device memory
      char* shadow = nullptr;
      char* p = recording->alloc(bytes, &shadow);
      std::memcpy(shadow, host, bytes);
      hssk_rt::h2d(p, shadow, bytes, stream);
      return p;
    }
    size_t need = (bytes + 255) & ~size_t(255);
    if (need > ring_bytes) throw std::runtime_error("hssk: descriptor batch exceeds staging ring");
    if (ring_off + need > ring_bytes) {
      hssk_rt::sync(stream);
      if (side_made) {   // (launches on the other stream of the pair may still read their descriptors)
        hssk_rt::sync(side);
        if (on_side) hssk_rt::sync(main_saved);
      }
      ring_off = 0;
    }
    std::memcpy(h_ring + ring_off, host, bytes);
    void* d;
    if (bytes <= zero_copy_bytes) {
      // small descriptor arrays are read by the kernel straight from the pinned host ring (it is mapped into the
      // device address space): no copy kernel, no extra launch on the critical path of the latency-bound tree levels
      d = h_ring + ring_off;
    } else {
      hssk_rt::h2d(d_ring + ring_off, h_ring + ring_off, bytes, stream);
      d = d_ring + ring_off;
    }
    ring_off += need;
    return d;
  }
  // second work buffer (blocked triangular solves: inverted diagonal blocks + a copy of the block being solved); separate from
  // `scratch`, whose contents a caller may still need
  double* d_aux = nullptr;
  size_t aux_bytes = 0;
  // every stream of the context idle: a buffer about to be freed may still be read by launches on the side stream
  void sync_all() {
    hssk_rt::sync(stream);
    if (side_made) {
      hssk_rt::sync(side);
      if (on_side) hssk_rt::sync(main_saved);
    }
  }
  double* aux(size_t bytes) {
    if (bytes > aux_bytes) {
      sync_all();
      hssk_rt::dev_free(d_aux);
      d_aux = (double*)hssk_rt::dev_malloc(bytes);
      aux_bytes = bytes;
    }
    return d_aux;
  }
  // written-out column blocks of a generated operand (hssk_sketch_gen: ragged rest columns, shapes outside the fused kernel)
  double* d_gen = nullptr;
  size_t gen_bytes = 0;
  double* gen_block(size_t bytes) {
    if (bytes > gen_bytes) {
      sync_all();
      hssk_rt::dev_free(d_gen);
      d_gen = (double*)hssk_rt::dev_malloc(bytes);
      gen_bytes = bytes;
    }
    return d_gen;
  }
  double* scratch(size_t bytes) {
    if (bytes > scratch_bytes) {
      sync_all();
      hssk_rt::dev_free(d_scratch);
      d_scratch = (double*)hssk_rt::dev_malloc(bytes);
      scratch_bytes = bytes;
    }
    return d_scratch;
  }
};

void hssk_set_error(const std::string& msg);

// "this entry point does not take these operands" (status 2): callers with a fallback path test for 2, the others go
// through ck() and must find a message, never a stale one
#define HSSK_UNSUPPORTED(msg) do { hssk_set_error(std::string(__func__) + ": " + (msg)); return 2; } while (0)
#define HSSK_API_BEGIN try {
#define HSSK_API_END                 \
  return 0;                          \
  }                                  \
  catch (const std::exception& e) {  \
    hssk_set_error(e.what());        \
    return 1;                        \
  }
