// TEST INFRASTRUCTURE ONLY.  Fiber-based SIMT emulator: every GPU thread of a workgroup is a
// ucontext fiber; the fibers of one workgroup run round-robin on one OS thread and switch at
// __syncthreads()/wave-collective points; workgroups are distributed over a few OS threads.
// Deterministic, no data races inside a workgroup; catches indexing/logic errors, not memory-model
// ones.  Wave collectives require every live lane of the wave to participate (as on hardware with
// a full exec mask).
// (fortified longjmp rejects jumps between fiber stacks)
#undef _FORTIFY_SOURCE
#include <setjmp.h>
#include <ucontext.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "hssk_device.h"

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
#include <sched.h>

namespace emu {
namespace {
constexpr size_t STACK = 256 * 1024;
// A fiber is entered once through makecontext / setcontext; every later switch is _setjmp / _longjmp, which -- unlike
// swapcontext -- does not save and restore the signal mask (one system call per switch).
struct Fiber {
  ucontext_t ctx;
  jmp_buf jb;
  char* stack = nullptr;
  bool done = false, started = false;
};
struct Wave {
  int count = 0, live = 0;
  unsigned gen = 0;
  double buf[64], buf2[64];
};
struct Worker {
  jmp_buf sched;
  ucontext_t sched_ctx;
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  int cur = 0, live = 0, bcount = 0;
  unsigned bgen = 0;
  const std::function<void()>* body = nullptr;
  std::vector<char> shmem;
  dim3 bdim;
};
thread_local Worker* W = nullptr;

// -DEMU_SWAPCONTEXT (sanitizer builds, tools/asan_emu.sh): plain swapcontext switches, which the sanitizer runtime follows
#ifdef EMU_SWAPCONTEXT
#define EMU_TO_SCHED(w, f) swapcontext(&(f).ctx, &(w)->sched_ctx)
#else
#define EMU_TO_SCHED(w, f) _longjmp((w)->sched, 1)
#endif
void yield() {
  Worker* w = W;
#ifdef EMU_SWAPCONTEXT
  swapcontext(&w->fibers[w->cur].ctx, &w->sched_ctx);
#else
  if (!_setjmp(w->fibers[w->cur].jb)) _longjmp(w->sched, 1);
#endif
}
}  // namespace
}  // namespace emu
// a polling lane gives way to the other workgroups (OS threads) AND to the other lanes of its own workgroup: on the device the
// waves of a workgroup progress independently, so a wave may poll for a word whose store another wave of a partner workgroup
// has yet to issue
void hssk_pause() {
  sched_yield();
  emu::yield();
}
namespace emu {
namespace {
void trampoline() {
  Worker* w = W;
  (*w->body)();
  Fiber& f = w->fibers[w->cur];
  f.done = true;
  w->live--;
  w->waves[w->cur / 64].live--;
  EMU_TO_SCHED(w, f);
}
void set_tid(Worker* w, int t) {
  threadIdx.x = t % w->bdim.x;
  threadIdx.y = (t / w->bdim.x) % w->bdim.y;
  threadIdx.z = t / (w->bdim.x * w->bdim.y);
}
void run_block(Worker* w, dim3 grid, dim3 block, unsigned bid) {
  int T = block.x * block.y * block.z;
  if ((int)w->fibers.size() < T) {
    size_t old = w->fibers.size();
    w->fibers.resize(T);
    for (size_t i = old; i < (size_t)T; i++) w->fibers[i].stack = (char*)std::malloc(STACK);
  }
  w->waves.assign((T + 63) / 64, Wave());
  for (int t = 0; t < T; t++) w->waves[t / 64].live++;
  w->live = T; w->bcount = 0; w->bgen = 0; w->bdim = block;
  blockIdx.x = bid % grid.x; blockIdx.y = (bid / grid.x) % grid.y; blockIdx.z = bid / (grid.x * grid.y);
  blockDim = block; gridDim = grid;
  for (int t = 0; t < T; t++) {
    Fiber& f = w->fibers[t];
    f.done = false;
    f.started = false;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = STACK;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, trampoline, 0);
  }
  long spins = 0;
  while (w->live > 0) {
    for (int t = 0; t < T; t++) {
      if (w->fibers[t].done) continue;
      w->cur = t;
      set_tid(w, t);
#ifdef EMU_SWAPCONTEXT
      swapcontext(&w->sched_ctx, &w->fibers[t].ctx);
#else
      if (!_setjmp(w->sched)) {
        Fiber& f = w->fibers[t];
        if (!f.started) { f.started = true; setcontext(&f.ctx); }
        else _longjmp(f.jb, 1);
      }
#endif
    }
    if (++spins > 200000000L) { std::fprintf(stderr, "emu: deadlock in block %u\n", bid); std::abort(); }
  }
}
}  // namespace

void* dyn_shared() { return W->shmem.data(); }

void block_barrier() {
  Worker* w = W;
  unsigned g = w->bgen;
  w->bcount++;
  for (;;) {
    if (w->bgen != g) return;
    if (w->bcount >= w->live) { w->bcount = 0; w->bgen++; return; }
    yield();
    set_tid(w, w->cur);
  }
}

static void wave_barrier(Worker* w, Wave& wv) {
  unsigned g = wv.gen;
  wv.count++;
  for (;;) {
    if (wv.gen != g) return;
    if (wv.count >= wv.live) { wv.count = 0; wv.gen++; return; }
    yield();
  }
}

double wave_xchg(double v, int src_lane) {
  Worker* w = W;
  int t = w->cur;
  Wave& wv = w->waves[t / 64];
  wv.buf[t % 64] = v;
  wave_barrier(w, wv);
  double r = wv.buf[src_lane];
  wave_barrier(w, wv);
  return r;
}

hssk_f16v mfma_f32_32x32x2(float a, float b, hssk_f16v c) {
  Worker* w = W;
  int t = w->cur, l = t % 64;
  Wave& wv = w->waves[t / 64];
  wv.buf[l] = a;
  wv.buf2[l] = b;
  wave_barrier(w, wv);
  const int col = l & 31;
  for (int r = 0; r < 16; r++) {
    const int row = 8 * (r / 4) + 4 * (l >> 5) + r % 4;
    float s = c[r];
    for (int k = 0; k < 2; k++) s = std::fmaf((float)wv.buf[row + 32 * k], (float)wv.buf2[col + 32 * k], s);
    c[r] = s;
  }
  wave_barrier(w, wv);
  return c;
}

unsigned long long wave_ballot(int pred) {
  Worker* w = W;
  int t = w->cur;
  Wave& wv = w->waves[t / 64];
  wv.buf[t % 64] = pred ? 1. : 0.;
  wave_barrier(w, wv);
  const int T = (int)(w->bdim.x * w->bdim.y * w->bdim.z), lanes = std::min(64, T - (t / 64) * 64);
  unsigned long long m = 0;
  for (int l = 0; l < lanes; l++)
    if (wv.buf[l] != 0.) m |= 1ULL << l;
  wave_barrier(w, wv);
  return m;
}

hssk_d4 mfma_f64_16x16x4(double a, double b, hssk_d4 c) {
  Worker* w = W;
  int t = w->cur, l = t % 64;
  Wave& wv = w->waves[t / 64];
  wv.buf[l] = a;
  wv.buf2[l] = b;
  wave_barrier(w, wv);
  int col = l & 15;
  for (int r = 0; r < 4; r++) {
    int row = (l >> 4) + 4 * r;
    double s = c[r];
    for (int k = 0; k < 4; k++) s = std::fma(wv.buf[row + 16 * k], wv.buf2[col + 16 * k], s);
    c[r] = s;
  }
  wave_barrier(w, wv);
  return c;
}

// Persistent worker threads: the fiber stacks of a worker (256 KB per emulated thread) are allocated once and reused by
// every launch (a fresh std::thread per launch used to allocate -- and never free -- them again).
namespace {
struct Pool {
  std::mutex m;
  std::condition_variable cv, done;
  const std::function<void()>* job = nullptr;
  unsigned long gen = 0;
  int pending = 0, size = 0;
  void worker_loop() {
    unsigned long seen = 0;
    for (;;) {
      const std::function<void()>* j;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return gen != seen; });
        seen = gen;
        j = job;
      }
      (*j)();
      {
        std::lock_guard<std::mutex> lk(m);
        if (--pending == 0) done.notify_all();
      }
    }
  }
  void run(const std::function<void()>& f) {
    std::unique_lock<std::mutex> lk(m);
    job = &f;
    pending = size;
    gen++;
    cv.notify_all();
    done.wait(lk, [&] { return pending == 0; });
  }
};
Pool* pool(int n) {   // leaked on purpose: the workers wait for work until the process ends
  static Pool* p = [n] {
    Pool* q = new Pool;
    q->size = n;
    for (int i = 0; i < n; i++) std::thread([q] { q->worker_loop(); }).detach();
    return q;
  }();
  return p;
}
std::mutex launch_mutex;   // one emulated launch at a time (several host threads may launch)
}  // namespace

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  unsigned nblocks = grid.x * grid.y * grid.z;
  if (!nblocks) return;
  static int nthreads = [] {
    const char* e = std::getenv("HSSK_EMU_THREADS");
    int n = e ? std::atoi(e) : (int)std::thread::hardware_concurrency();
    return std::max(1, std::min(n, 16));
  }();
  std::atomic<unsigned> next{0};
  std::function<void()> work = [&]() {
    static thread_local Worker worker;
    W = &worker;
    worker.body = &body;
    worker.shmem.assign(shmem + 64, 0);
    for (;;) {
      unsigned b = next.fetch_add(1);
      if (b >= nblocks) break;
      run_block(&worker, grid, block, b);
    }
  };
  if (nthreads <= 1 || nblocks <= 1) { work(); return; }
  std::lock_guard<std::mutex> lk(launch_mutex);
  pool(nthreads)->run(work);
}
}  // namespace emu
