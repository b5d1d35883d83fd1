// Context, error state and memory helpers of the hssk C-ABI (include/hssk.h).
#include "hssk_internal.h"

#include <cstdlib>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

static thread_local std::string g_err;
// retired contexts are kept (stream, events, 64 MB pinned + device staging rings) and handed out
// again by hssk_ctx_create: creating pinned memory costs milliseconds, solvers create many matrices
static std::mutex g_pool_mu;
static std::vector<hssk_ctx*> g_pool;
void hssk_set_error(const std::string& msg) { g_err = msg; }

thread_local std::vector<std::function<void()>>* hssk_rec::sink = nullptr;

// ---- pipelined host -> device uploads (hssk_h2d_block_async) ----------------------------------------------------------
struct hssk_uploader {
  static constexpr int SLOTS = 4;
  static constexpr size_t CHUNK = size_t(256) << 20;   // (large pieces: a piece costs one round of host-thread start-up)
  hssk_rt::stream_t copy{};
  hssk_rt::event_t ev_copy{}, ev_compute{};
  hssk_rt::event_t ev_mark[2];
  bool marked[2] = {false, false};
  char* pinned[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
  hssk_rt::event_t slot_ev[SLOTS];
  bool slot_busy[SLOTS] = {false, false, false, false};
  int next = 0;
  hssk_uploader() {
    copy = hssk_rt::stream_create();
    ev_copy = hssk_rt::event_create();
    ev_compute = hssk_rt::event_create();
    ev_mark[0] = hssk_rt::event_create();
    ev_mark[1] = hssk_rt::event_create();
    for (int i = 0; i < SLOTS; i++) slot_ev[i] = hssk_rt::event_create();
  }
  // the bounce slots (4 x 256 MiB of pinned host memory) exist only while the packing path of h2d_bytes uses them: taken on
  // first use, given back when the context is retired (a pooled context keeps its streams and events, not the slots)
  char* slot(int s) {
    if (!pinned[s]) pinned[s] = (char*)hssk_rt::pinned_malloc(CHUNK);
    return pinned[s];
  }
  void release_slots() {
    try { hssk_rt::sync(copy); } catch (...) {}
    for (int i = 0; i < SLOTS; i++) {
      if (pinned[i]) hssk_rt::pinned_free(pinned[i]);
      pinned[i] = nullptr;
      slot_busy[i] = false;
    }
  }
  ~hssk_uploader() {
    try { hssk_rt::sync(copy); } catch (...) {}
    for (int i = 0; i < SLOTS; i++) { if (pinned[i]) hssk_rt::pinned_free(pinned[i]); hssk_rt::event_destroy(slot_ev[i]); }
    hssk_rt::event_destroy(ev_copy);
    hssk_rt::event_destroy(ev_compute);
    hssk_rt::event_destroy(ev_mark[0]);
    hssk_rt::event_destroy(ev_mark[1]);
    hssk_rt::stream_destroy(copy);
  }
};
static hssk_uploader* uploader(hssk_ctx* c) {
  if (!c->uploader) c->uploader = new hssk_uploader();
  return c->uploader;
}
// persistent host threads for the packing (a piece is packed in ~5 ms: starting threads per piece would cost as much)
class PackPool {
 public:
  explicit PackPool(unsigned n) : n_(n), pid_(getpid()) {
    for (unsigned t = 0; t < n_; t++) th_.emplace_back([this, t] { loop(t); });
  }
  ~PackPool() {
    if (getpid() != pid_) { for (auto& t : th_) t.detach(); return; }   // (a forked child has the object but not the threads)
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; gen_++; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  unsigned size() const { return n_; }
  // runs fn(t) for t < size() on the pool and returns when all are done
  void run(const std::function<void(unsigned)>& fn) {
    if (getpid() != pid_) { for (unsigned t = 0; t < n_; t++) fn(t); return; }   // forked child: the caller does the work
    std::lock_guard<std::mutex> own(owner_);   // one job at a time (contexts on several host threads share the pool)
    std::unique_lock<std::mutex> lk(mu_);
    fn_ = &fn; pending_ = n_; gen_++;
    cv_.notify_all();
    done_.wait(lk, [&] { return pending_ == 0; });
  }

 private:
  void loop(unsigned t) {
    unsigned long seen = 0;
    for (;;) {
      const std::function<void(unsigned)>* f;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        f = fn_;
      }
      (*f)(t);
      { std::lock_guard<std::mutex> g(mu_); if (--pending_ == 0) done_.notify_all(); }
    }
  }
  unsigned n_;
  const pid_t pid_;
  std::vector<std::thread> th_;
  std::mutex mu_, owner_;
  std::condition_variable cv_, done_;
  const std::function<void(unsigned)>* fn_ = nullptr;
  unsigned long gen_ = 0;
  unsigned pending_ = 0;
  bool stop_ = false;
};
static PackPool& pack_pool() {
  static PackPool p(std::max(1u, std::min(24u, std::thread::hardware_concurrency())));
  return p;
}
// columns [c0, c1) (colb bytes each, spitch bytes apart) of a column-major host block into a compact pinned buffer, on the
// host's hardware threads
static void host_pack(char* dst, const char* src, size_t spitch, size_t colb, long long c0, long long c1) {
  const long long ncol = c1 - c0;
  const size_t total = colb * (size_t)ncol;
  if (total < (size_t(8) << 20)) {
    for (long long j = 0; j < ncol; j++) std::memcpy(dst + colb * j, src + (size_t)(c0 + j) * spitch, colb);
    return;
  }
  PackPool& pool = pack_pool();
  const unsigned nt = pool.size();
  // split by bytes, not by columns: a block may be one very long column
  const size_t per = (total + nt - 1) / nt;
  pool.run([=](unsigned t) {
    size_t b0 = std::min(total, per * t), b1 = std::min(total, b0 + per);
    while (b0 < b1) {
      const size_t j = b0 / colb, o = b0 % colb, len = std::min(colb - o, b1 - b0);
      std::memcpy(dst + b0, src + (size_t)(c0 + (long long)j) * spitch + o, len);
      b0 += len;
    }
  });
}
// `cols` columns of `colb` bytes from host memory (pinned: DMA in place; pageable: the runtime's staged copy for a contiguous
// source, else the pinned bounce ring) to the device, on the copy stream
static void h2d_bytes(hssk_ctx* c, char* dst, size_t dpitch, const char* src, size_t spitch, size_t colb, long long cols) {
  hssk_uploader* u = uploader(c);
  if (hssk_rt::is_pinned_host_pointer(src)) {   // DMA straight from the caller's pinned buffer
    if (dpitch == colb && spitch == colb) hssk_rt::h2d(dst, src, colb * (size_t)cols, u->copy);
    else hssk_rt::h2d_2d(dst, dpitch, src, spitch, colb, (size_t)cols, u->copy);
    return;
  }
  // A contiguous pageable source goes through the runtime's own staged copy: measured at N = 1e5 (80 GB) 56.4 GB/s against
  // 52-53 GB/s for the packing pool below (the link gives a pinned buffer 57.6 GB/s); the call returns when the source has
  // been staged, which the caller's loop allows for (HostBlockSource::sample).  HSSK_H2D_DIRECT=0 forces the pool.
  static const bool direct = [] { const char* e = std::getenv("HSSK_H2D_DIRECT"); return !(e && e[0] == '0'); }();
  if (direct && dpitch == colb && spitch == colb) {
    hssk_rt::h2d(dst, src, colb * (size_t)cols, u->copy);
    return;
  }
  if (colb > hssk_uploader::CHUNK) {   // columns longer than a bounce slot: pieces of one column at a time
    for (long long j = 0; j < cols; j++)
      for (size_t o = 0; o < colb; o += hssk_uploader::CHUNK) {
        const size_t len = std::min(hssk_uploader::CHUNK, colb - o);
        const int s = u->next; u->next = (s + 1) % hssk_uploader::SLOTS;
        if (u->slot_busy[s]) hssk_rt::event_sync(u->slot_ev[s]);
        std::memcpy(u->slot(s), src + (size_t)j * spitch + o, len);
        hssk_rt::h2d(dst + (size_t)j * dpitch + o, u->slot(s), len, u->copy);
        hssk_rt::event_record(u->slot_ev[s], u->copy);
        u->slot_busy[s] = true;
      }
    return;
  }
  const long long cpc = std::max<long long>(1, (long long)(hssk_uploader::CHUNK / colb));   // columns per bounce slot
  for (long long c0 = 0; c0 < cols; c0 += cpc) {
    const long long c1 = std::min(cols, c0 + cpc);
    const int s = u->next; u->next = (s + 1) % hssk_uploader::SLOTS;
    if (u->slot_busy[s]) hssk_rt::event_sync(u->slot_ev[s]);   // the DMA that last read this slot has finished
    host_pack(u->slot(s), src, spitch, colb, c0, c1);
    if (dpitch == colb)   // contiguous on the device: one linear DMA (the rectangular copy path is markedly slower)
      hssk_rt::h2d(dst + (size_t)c0 * dpitch, u->slot(s), colb * (size_t)(c1 - c0), u->copy);
    else
      hssk_rt::h2d_2d(dst + (size_t)c0 * dpitch, dpitch, u->slot(s), colb, colb, (size_t)(c1 - c0), u->copy);
    hssk_rt::event_record(u->slot_ev[s], u->copy);
    u->slot_busy[s] = true;
  }
}

extern "C" {

int hssk_h2d_block_async(hssk_ctx* c, double* dst, long long ldd, const double* src, long long lds, long long rows,
                         long long cols) {
  HSSK_API_BEGIN
  if (rows <= 0 || cols <= 0) return 0;
  h2d_bytes(c, (char*)dst, sizeof(double) * (size_t)ldd, (const char*)src, sizeof(double) * (size_t)lds, sizeof(double) * (size_t)rows, cols);
  HSSK_API_END
}
int hssk_h2d_bytes_async(hssk_ctx* c, void* dst, long long dpitch, const void* src, long long spitch, long long width, long long cols) {
  HSSK_API_BEGIN
  if (width <= 0 || cols <= 0) return 0;
  if (dpitch < width || spitch < width) HSSK_UNSUPPORTED("pitch smaller than the row width");
  h2d_bytes(c, (char*)dst, (size_t)dpitch, (const char*)src, (size_t)spitch, (size_t)width, cols);
  HSSK_API_END
}
int hssk_copy_fence(hssk_ctx* c) {
  HSSK_API_BEGIN
  hssk_uploader* u = uploader(c);
  hssk_rt::event_record(u->ev_copy, u->copy);
  hssk_rt::stream_wait_event(c->stream, u->ev_copy);
  HSSK_API_END
}
int hssk_compute_fence(hssk_ctx* c) {
  HSSK_API_BEGIN
  hssk_uploader* u = uploader(c);
  hssk_rt::event_record(u->ev_compute, c->stream);
  hssk_rt::stream_wait_event(u->copy, u->ev_compute);
  HSSK_API_END
}
int hssk_compute_mark(hssk_ctx* c, int slot) {
  HSSK_API_BEGIN
  hssk_uploader* u = uploader(c);
  hssk_rt::event_record(u->ev_mark[slot & 1], c->stream);
  u->marked[slot & 1] = true;
  HSSK_API_END
}
int hssk_copy_wait(hssk_ctx* c, int slot) {
  HSSK_API_BEGIN
  hssk_uploader* u = uploader(c);
  if (u->marked[slot & 1]) hssk_rt::stream_wait_event(u->copy, u->ev_mark[slot & 1]);
  HSSK_API_END
}

const char* hssk_last_error(void) { return g_err.c_str(); }

int hssk_ctx_create(hssk_ctx** out, int device) {
  HSSK_API_BEGIN
  if (hssk_rt::device_count() <= device)
    throw std::runtime_error("hssk_ctx_create: no HIP device " + std::to_string(device) +
                             " (this library has no CPU fallback)");
  hssk_rt::set_device(device);
  {
    std::lock_guard<std::mutex> g(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); i++)
      if (g_pool[i]->device == device) {
        *out = g_pool[i];
        g_pool.erase(g_pool.begin() + i);
        return 0;
      }
  }
  hssk_ctx* c = new hssk_ctx;
  c->device = device;
  c->stream = hssk_rt::stream_create();
  c->ring_bytes = size_t(64) << 20;
  c->h_ring = (char*)hssk_rt::pinned_malloc(c->ring_bytes);
  c->zero_copy_bytes = hssk_rt::pinned_is_device_visible() ? 65536 : 0;   // measured: tree+factor+solve 13.3 -> 11.7 ms at N = 1e5
  if (const char* e = std::getenv("HSSK_ZERO_COPY_BYTES")) c->zero_copy_bytes = (size_t)std::atoll(e);
  c->d_ring = (char*)hssk_rt::dev_malloc(c->ring_bytes);
  c->ev0 = hssk_rt::event_create();
  c->ev1 = hssk_rt::event_create();
  c->ev_sync = hssk_rt::event_create();
  *out = c;
  HSSK_API_END
}

void hssk_ctx_destroy(hssk_ctx* c) {
  if (!c) return;
  try { hssk_rt::sync(c->stream); } catch (...) {}
  {
    std::lock_guard<std::mutex> g(g_pool_mu);
    if (g_pool.size() < 8) {
      if (c->uploader) c->uploader->release_slots();
      c->ring_off = 0;
      c->dgemm_timed = false;
      // a context abandoned in the middle of something (an exception between a stopwatch's start and stop, or inside a
      // side-stream section) goes back to the pool in its neutral state
      if (c->on_side) { c->stream = c->main_saved; c->on_side = false; }
      for (int i = 0; i < 8; i++) {
        for (auto& p : c->watch[i]) { c->watch_free.push_back(p.first); c->watch_free.push_back(p.second); }
        c->watch[i].clear();
        c->watch_open[i] = false;
      }
      c->recording = nullptr;
      c->require_mma = false;
      g_pool.push_back(c);
      return;
    }
  }
  hssk_rt::pinned_free(c->h_ring);
  hssk_rt::dev_free(c->d_ring);
  hssk_rt::dev_free(c->d_scratch);
  hssk_rt::dev_free(c->d_aux);
  hssk_rt::dev_free(c->d_gen);
  delete c->uploader;
  hssk_rt::pinned_free(c->h_sweep_err);
  for (auto& w : c->watch) for (auto& p : w) { hssk_rt::event_destroy(p.first); hssk_rt::event_destroy(p.second); }
  for (auto& b : c->dgemm_deferred) { hssk_rt::event_destroy(b.a); hssk_rt::event_destroy(b.b); }
  for (auto e : c->watch_free) hssk_rt::event_destroy(e);
  hssk_rt::event_destroy(c->ev0);
  hssk_rt::event_destroy(c->ev_sync);
  if (c->side_made) { hssk_rt::event_destroy(c->ev_fork); hssk_rt::event_destroy(c->ev_join); hssk_rt::stream_destroy(c->side); }
  hssk_rt::event_destroy(c->ev1);
  hssk_rt::stream_destroy(c->stream);
  delete c;
}

void* hssk_ctx_stream(hssk_ctx* c) { return (void*)c->stream; }

int hssk_stream_wait(hssk_ctx* waiter, hssk_ctx* on) {
  HSSK_API_BEGIN
  if (!waiter || !on) throw std::invalid_argument("hssk_stream_wait: null context");
  if (waiter == on) return 0;
  hssk_rt::event_record(on->ev_sync, on->stream);
  hssk_rt::stream_wait_event(waiter->stream, on->ev_sync);
  HSSK_API_END
}
// ---- side stream: launches between _begin and _end go to a second stream that first waits for everything issued on the
// main stream so far; _join makes the main stream wait for them.  Recorded plans replay the same routing.
int hssk_side_begin(hssk_ctx* c) {
  HSSK_API_BEGIN
  if (c->on_side) throw std::logic_error("hssk_side_begin: already on the side stream");
  if (!c->side_made) {
    c->side = hssk_rt::stream_create();
    c->ev_fork = hssk_rt::event_create();
    c->ev_join = hssk_rt::event_create();
    c->side_made = true;
  }
  auto f = [c]() {
    hssk_rt::event_record(c->ev_fork, c->stream);
    hssk_rt::stream_wait_event(c->side, c->ev_fork);
    c->main_saved = c->stream;
    c->stream = c->side;
    c->on_side = true;
  };
  f();
  if (hssk_rec::sink) hssk_rec::sink->push_back(f);
  HSSK_API_END
}
int hssk_side_end(hssk_ctx* c) {
  HSSK_API_BEGIN
  if (!c->on_side) throw std::logic_error("hssk_side_end: not on the side stream");
  auto f = [c]() { c->stream = c->main_saved; c->on_side = false; };
  f();
  if (hssk_rec::sink) hssk_rec::sink->push_back(f);
  HSSK_API_END
}
int hssk_side_join(hssk_ctx* c) {
  HSSK_API_BEGIN
  if (c->on_side) throw std::logic_error("hssk_side_join: still on the side stream");
  if (!c->side_made) return 0;
  auto f = [c]() {
    hssk_rt::event_record(c->ev_join, c->side);
    hssk_rt::stream_wait_event(c->stream, c->ev_join);
  };
  f();
  if (hssk_rec::sink) hssk_rec::sink->push_back(f);
  HSSK_API_END
}
static hssk_rt::event_t watch_event(hssk_ctx* c) {
  if (c->watch_free.empty()) return hssk_rt::event_create();
  hssk_rt::event_t e = c->watch_free.back();
  c->watch_free.pop_back();
  return e;
}
int hssk_watch_start(hssk_ctx* c, int id) {
  HSSK_API_BEGIN
  if (id < 0 || id >= 8) throw std::invalid_argument("hssk_watch_start: id out of range");
  if (c->watch_open[id]) {
    // left open by a caller that threw between start and stop: the interval is dropped, the stopwatch starts over
    auto p = c->watch[id].back();
    c->watch[id].pop_back();
    c->watch_free.push_back(p.first);
    c->watch_free.push_back(p.second);
    c->watch_open[id] = false;
  }
  hssk_rt::event_t a = watch_event(c), b = watch_event(c);
  hssk_rt::event_record(a, c->stream);
  c->watch[id].emplace_back(a, b);
  c->watch_open[id] = true;
  HSSK_API_END
}
int hssk_watch_stop(hssk_ctx* c, int id) {
  HSSK_API_BEGIN
  if (id < 0 || id >= 8 || !c->watch_open[id]) throw std::logic_error("hssk_watch_stop: stopwatch is not running");
  hssk_rt::event_record(c->watch[id].back().second, c->stream);
  c->watch_open[id] = false;
  HSSK_API_END
}
double hssk_watch_read_ms(hssk_ctx* c, int id, int* pairs) {
  if (pairs) *pairs = 0;
  if (!c || id < 0 || id >= 8) return 0.;
  try {
    if (c->watch_open[id]) { hssk_rt::event_record(c->watch[id].back().second, c->stream); c->watch_open[id] = false; }
    hssk_rt::sync(c->stream);
    double ms = 0.;
    for (auto& p : c->watch[id]) {
      ms += hssk_rt::event_elapsed_ms(p.first, p.second);
      c->watch_free.push_back(p.first);
      c->watch_free.push_back(p.second);
    }
    if (pairs) *pairs = (int)c->watch[id].size();
    c->watch[id].clear();
    return ms;
  } catch (const std::exception& e) { hssk_set_error(e.what()); return -1.; }
}

int hssk_sync(hssk_ctx* c) {
  HSSK_API_BEGIN
  hssk_rt::sync(c->stream);
  HSSK_API_END
}

void* hssk_malloc(long long bytes) {
  try { return hssk_rt::dev_malloc((size_t)bytes); } catch (const std::exception& e) { hssk_set_error(e.what()); return nullptr; }
}
void hssk_free(void* p) { hssk_rt::dev_free(p); }
long long hssk_device_total_bytes(void) {
  try { return (long long)hssk_rt::device_total_bytes(); } catch (...) { return 0; }
}

int hssk_memcpy_h2d(hssk_ctx* c, void* dst, const void* src, long long bytes) {
  HSSK_API_BEGIN
  hssk_rt::h2d(dst, src, (size_t)bytes, c->stream);
  hssk_rt::sync(c->stream);  // src may be pageable: make the call synchronous for FFI safety
  HSSK_API_END
}
int hssk_upload_async(hssk_ctx* c, void* dst, const void* src, long long bytes) {
  HSSK_API_BEGIN
  if (bytes <= 0) return 0;
  const size_t need = ((size_t)bytes + 255) & ~size_t(255);
  if (need > c->ring_bytes) {   // larger than the ring: plain synchronous copy
    hssk_rt::h2d(dst, src, (size_t)bytes, c->stream);
    hssk_rt::sync(c->stream);
    return 0;
  }
  if (c->ring_off + need > c->ring_bytes) { hssk_rt::sync(c->stream); c->ring_off = 0; }
  std::memcpy(c->h_ring + c->ring_off, src, (size_t)bytes);
  hssk_rt::h2d(dst, c->h_ring + c->ring_off, (size_t)bytes, c->stream);
  c->ring_off += need;
  HSSK_API_END
}
int hssk_memcpy_d2h(hssk_ctx* c, void* dst, const void* src, long long bytes) {
  HSSK_API_BEGIN
  hssk_rt::d2h(dst, src, (size_t)bytes, c->stream);
  hssk_rt::sync(c->stream);
  HSSK_API_END
}

int hssk_memcpy_d2d(hssk_ctx* c, void* dst, const void* src, long long bytes) {
  HSSK_API_BEGIN
  hssk_rt::d2d(dst, src, (size_t)bytes, c->stream);
  HSSK_API_END
}
int hssk_memcpy2d_h2d(hssk_ctx* c, void* dst, long long dpitch, const void* src, long long spitch,
                      long long width, long long height) {
  HSSK_API_BEGIN
  hssk_rt::h2d_2d(dst, (size_t)dpitch, src, (size_t)spitch, (size_t)width, (size_t)height, c->stream);
  hssk_rt::sync(c->stream);
  HSSK_API_END
}
int hssk_memcpy2d_d2h(hssk_ctx* c, void* dst, long long dpitch, const void* src, long long spitch,
                      long long width, long long height) {
  HSSK_API_BEGIN
  hssk_rt::d2h_2d(dst, (size_t)dpitch, src, (size_t)spitch, (size_t)width, (size_t)height, c->stream);
  hssk_rt::sync(c->stream);
  HSSK_API_END
}
int hssk_memset_zero(hssk_ctx* c, void* dst, long long bytes) {
  HSSK_API_BEGIN
  hssk_rt::memset_async(dst, 0, (size_t)bytes, c->stream);
  HSSK_API_END
}
int hssk_is_device_pointer(const void* p) { return hssk_rt::is_device_pointer(p) ? 1 : 0; }

double hssk_last_dgemm_clock_ghz(hssk_ctx* c) {
  try {
    if (!c->d_clk) return 0.;
    long long h[2] = {0, 0};
    hssk_rt::d2h(h, c->d_clk, sizeof(h), c->stream);
    hssk_rt::sync(c->stream);
    return h[1] > 0 ? (double)h[0] / ((double)h[1] / 100e6) * 1e-9 : 0.;
  } catch (...) { return 0.; }
}

long long hssk_last_dgemm_trace(hssk_ctx* c, long long* out, long long max_wgs) {
  try {
    if (!c->d_clk || c->dgemm_trace_wgs <= 0) return 0;
    const long long nw = c->dgemm_trace_wgs < max_wgs ? c->dgemm_trace_wgs : max_wgs;
    if (nw > 0) { hssk_rt::d2h(out, c->d_clk + 4, sizeof(long long) * 4 * nw, c->stream); hssk_rt::sync(c->stream); }
    return nw;
  } catch (...) { return -1; }
}

// ---- sweep plans: record a sequence of batched launches once, replay it without host-side descriptor work
int hssk_plan_begin(hssk_ctx* c, hssk_plan** out) {
  try {
    if (c->recording || hssk_rec::sink) throw std::logic_error("hssk_plan_begin: a plan is already being recorded");
    if (!hssk_rt::pinned_is_device_visible()) throw std::runtime_error("hssk_plan_begin: pinned memory is not device-visible here");
    hssk_plan* p = new hssk_plan();
    c->recording = p;
    hssk_rec::sink = &p->launches;
    *out = p;
    return 0;
  } catch (const std::exception& e) { hssk_set_error(e.what()); return 1; }
}
int hssk_plan_end(hssk_ctx* c) {
  c->recording = nullptr;
  hssk_rec::sink = nullptr;
  return 0;
}
int hssk_plan_replay(hssk_ctx* c, hssk_plan* p) {
  try {
    if (c->recording) throw std::logic_error("hssk_plan_replay: cannot replay while recording");
    for (auto& f : p->launches) f();
    hssk_rt::check_launch();
    return 0;
  } catch (const std::exception& e) { hssk_set_error(e.what()); return 1; }
}
void hssk_plan_destroy(hssk_plan* p) { delete p; }
int hssk_plan_size(const hssk_plan* p) { return p ? (int)p->launches.size() : 0; }

double hssk_last_dgemm_flops(hssk_ctx* c) { return c->dgemm_timed ? c->dgemm_timed_flops : 0.; }

float hssk_last_dgemm_ms(hssk_ctx* c) {
  if (!c->dgemm_timed) return -1.f;
  try { return hssk_rt::event_elapsed_ms(c->ev0, c->ev1); } catch (...) { return -1.f; }
}

// The bracket of the last timed launch is set aside (the context gets fresh events) instead of being read: reading needs a
// synchronisation behind the launch, and a caller with more launches to enqueue -- the second sketch product, the leaf level
// behind it -- would leave the device idle for the sync's return and its own host work (35 - 80 us per product at N = 1e5).
int hssk_dgemm_timing_defer(hssk_ctx* c) {
  HSSK_API_BEGIN
  if (!c->dgemm_timed) return 0;
  c->dgemm_deferred.push_back(hssk_ctx::TimedBracket{c->ev0, c->ev1, c->dgemm_timed_flops});
  c->ev0 = watch_event(c);
  c->ev1 = watch_event(c);
  c->dgemm_timed = false;
  HSSK_API_END
}
// synchronises; sums and releases the brackets set aside since the last call
int hssk_dgemm_timing_collect(hssk_ctx* c, double* ms, double* flops, int* launches) {
  HSSK_API_BEGIN
  double t = 0., f = 0.;
  int n = 0;
  if (!c->dgemm_deferred.empty()) hssk_rt::sync(c->stream);
  for (auto& b : c->dgemm_deferred) {
    const float e = hssk_rt::event_elapsed_ms(b.a, b.b);
    if (e > 0) { t += e; f += b.flops; n++; }
    c->watch_free.push_back(b.a);
    c->watch_free.push_back(b.b);
  }
  c->dgemm_deferred.clear();
  if (ms) *ms = t;
  if (flops) *flops = f;
  if (launches) *launches = n;
  HSSK_API_END
}

}  // extern "C"
