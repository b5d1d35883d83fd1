// DeviceHSS implementation: level-synchronous compression / apply / ULV factor / solve on the device.
// Reference behaviour followed (all under /root/reference/src/HSS unless noted):
//   HSSMatrix.cpp:60-82                     tree construction
//   HSSMatrix.compress_stable.hpp:100-442   adaptive stable compression (default)
//   HSSMatrix.compress.hpp:100-165,300-368,524-724  original compression, local samples, reduce
//   HSSBasisID.hpp:146-203                  interpolative basis apply / applyC / dense
//   HSSMatrix.apply.hpp:55-220              mat-vec
//   HSSMatrix.factor.hpp:51-147             ULV factorization
//   HSSMatrix.solve.hpp:69-238              ULV solve
// Sibling nodes are independent, so the reference's post-order recursion is executed here as one
// batched kernel launch per step and tree height (cf. its own level-wise variant,
// HSSMatrix.compress_stable.hpp:234-277).
#include "hss_engine.hpp"
#include "Comm.hpp"

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

namespace {
inline void ck(int rc) {
  if (rc) throw std::runtime_error(std::string("hssk: ") + hssk_last_error());
}
inline double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

void CommSpec::apply(EngineOptions& e) const {
  e.world = world; e.rank = rank; e.comm_user = user;
  e.allgather = nullptr; e.allgather_stream = nullptr; e.allreduce_stream = nullptr; e.reduce_scatter_stream = nullptr;
  if (world <= 1) return;
  if (native) {
    e.allgather_stream = comm::rccl_allgather_hook;
    e.allreduce_stream = comm::rccl_allreduce_hook;
    e.reduce_scatter_stream = comm::rccl_reduce_scatter_hook;
  } else {
    e.allgather = allgather;
  }
}

// host random stream of the reference (misc/RandomWrapper.hpp:128-191): engine seeded with 0
struct HostRng {
  std::default_random_engine sj{0};   // SJLT patterns (the reference seeds its generator from the clock, sketch.hpp:266-270)
  std::minstd_rand lin{0};
  std::mt19937 mer{0};
  std::normal_distribution<double> nd;
  std::uniform_real_distribution<double> ud;
};

// process-wide pool of device chunks: arenas return their chunks here instead of hipFree, so that
// repeated constructions (solver loops, benchmarks) do not pay hipMalloc / hipFree page-table work
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
  size_t cached_ = 0, limit_ = size_t(8) << 30;
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
// sample / element sources
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
};

struct DeviceHSS::DenseDeviceSource : DeviceHSS::Source {
  const double* dA;
  long long lda;
  DenseDeviceSource(const double* a, long long l) : dA(a), lda(l) {}
  void sample(DeviceHSS& H, int r0, int dn) override {
    const long long N = H.n_;
    // AFunctor::operator()(Rr,Rc,Sr,Sc), HSSExtra.hpp:236-239, in the transposed sample layout.
    // Multi-GPU: this rank computes the sample columns [j0, j1) only (rows j0:j1 of A for Sr,
    // columns j0:j1 of A for Sc).
    long long j0 = 0, j1 = N;
    if (H.dist_subtree_) {           // this rank's subtree range: nothing is exchanged here
      const Node& c = H.nodes_[H.cut_nodes_[H.o_.rank]];
      j0 = c.lo; j1 = c.lo + c.m;
    } else if (H.o_.world > 1) {     // fallback: equal column shards + all-gather of the samples
      j0 = std::min(N, H.cols_per_rank_ * H.o_.rank); j1 = std::min(N, j0 + H.cols_per_rank_);
    }
    const long long nloc = j1 - j0;
    // SJLT sketch: stream A once per product instead of a dense GEMM (blocks wider than the kernel's LDS tile, or
    // STRUMPACK_AMD_SJLT_DENSE=1, multiply with the dense form of the pattern)
    static const bool sj_dense = std::getenv("STRUMPACK_AMD_SJLT_DENSE") && std::atoi(std::getenv("STRUMPACK_AMD_SJLT_DENSE"));
    if (nloc > 0 && H.sj_pat_ && dn <= 1024 && !sj_dense) {
      for (int t = 0; t < 2; t++) {
        const double* Aop = t == 0 ? dA + j0 : dA + j0 * lda;
        double* St = (t == 0 ? H.Srt_ : H.Sct_) + r0 + j0 * H.dcap_;
        ck(hssk_sjlt_sketch(H.ctx_, t, nloc, N, Aop, lda, H.sj_pat_, H.sj_nnz_, dn, St, H.dcap_));
        ck(hssk_sync(H.ctx_));
        float ms = hssk_last_dgemm_ms(H.ctx_);
        if (ms > 0) {
          H.stats_.sketch_kernel_ms += ms; H.stats_.sketch_launches++;
          H.stats_.sketch_kernel_flops += hssk_last_dgemm_flops(H.ctx_);
          H.stats_.sketch_kernel_bytes += 8.0 * (double)nloc * (double)N;
        }
      }
    } else if (nloc > 0) {
      ck(hssk_dgemm(H.ctx_, 1, dn, nloc, N, 1.0, H.Rt_ + r0, H.dcap_, dA + j0, lda, 0.0, H.Srt_ + r0 + j0 * H.dcap_, H.dcap_));
      ck(hssk_sync(H.ctx_));
      float ms = hssk_last_dgemm_ms(H.ctx_);
      if (ms > 0) { H.stats_.sketch_kernel_ms += ms; H.stats_.sketch_launches++; H.stats_.sketch_kernel_flops += hssk_last_dgemm_flops(H.ctx_); }
      ck(hssk_dgemm(H.ctx_, 0, dn, nloc, N, 1.0, H.Rt_ + r0, H.dcap_, dA + j0 * lda, lda, 0.0, H.Sct_ + r0 + j0 * H.dcap_, H.dcap_));
      ck(hssk_sync(H.ctx_));
      ms = hssk_last_dgemm_ms(H.ctx_);
      if (ms > 0) { H.stats_.sketch_kernel_ms += ms; H.stats_.sketch_launches++; H.stats_.sketch_kernel_flops += hssk_last_dgemm_flops(H.ctx_); }
    }
    if (H.o_.world > 1 && !H.dist_subtree_) {
      const long long bytes = (long long)sizeof(double) * H.dcap_ * H.cols_per_rank_;
      H.comm(H.Srt_, bytes);
      H.comm(H.Sct_, bytes);
    }
  }
  void extract(DeviceHSS& H, const std::vector<ElemReq>& reqs) override {
    std::vector<hssk_elem_desc> d;
    d.reserve(reqs.size());
    for (auto& r : reqs)
      if (r.m > 0 && r.n > 0) d.push_back(hssk_elem_desc{dA, lda, r.dI, r.dJ, r.i0, r.j0, r.dB, r.m, r.n, r.ldb, 0});
    if (!d.empty()) ck(hssk_gather_elems(H.ctx_, d.data(), (int)d.size()));
  }
};

// Sharded dense operand (one process per GPU, subtree ownership): this rank holds the columns [j0, j1) of its subtree
// (n x nloc) and, optionally, the rows [j0, j1) (nloc x n) -- never the whole matrix.
//   sketch:  Sc(j0:j1, :) = A(:, j0:j1)^H R                                   -- local
//            Sr(j0:j1, :) = A(j0:j1, :) R                                     -- local when the row block is given, else
//            Sr = sum_g A(:, cols_g) R(cols_g, :): every rank multiplies its column block with its rows of R and the
//            partial N x d products are summed to the owners of the rows (reduce-scatter over xGMI, SURVEY.md 8(e)(5):
//            the "reduce of off-diagonal contributions"; per-rank flops are the same 2 n nloc d either way)
//   elements: blocks inside the subtree come from the column block; the coupling blocks of the replicated top nodes
//            B01 = A(Ir_0, Ic_1) straddle the ranks: every rank fills in the rows (columns) it holds, zeros elsewhere,
//            and the partial blocks are summed over the ranks (a few r x r blocks per top level)
struct DeviceHSS::ShardedDenseSource : DeviceHSS::Source {
  const double* dRows;
  long long ldr;
  const double* dCols;
  long long ldc;
  ShardedDenseSource(const double* r, long long lr, const double* c, long long lc) : dRows(r), ldr(lr), dCols(c), ldc(lc) {}
  void sample(DeviceHSS& H, int r0, int dn) override {
    const bool single = H.o_.world == 1;   // one rank: its "shard" is the whole operand (same code path, no collective)
    if (!single && !H.dist_subtree_) throw std::invalid_argument("sharded operand: the tree cannot be cut into one subtree per rank (world must be a power of two and the tree complete down to that depth)");
    if (H.sj_pat_) throw std::invalid_argument("sharded operand: the SJLT sketch needs the replicated-operand interface");
    const long long N = H.n_;
    const Node& c = H.nodes_[single ? 0 : H.cut_nodes_[H.o_.rank]];
    const long long j0 = c.lo, nloc = c.m;
    auto timed = [&] {
      ck(hssk_sync(H.ctx_));
      const float ms = hssk_last_dgemm_ms(H.ctx_);
      if (ms > 0) { H.stats_.sketch_kernel_ms += ms; H.stats_.sketch_launches++; H.stats_.sketch_kernel_flops += hssk_last_dgemm_flops(H.ctx_); }
    };
    ck(hssk_dgemm(H.ctx_, 0, dn, nloc, N, 1.0, H.Rt_ + r0, H.dcap_, dCols, ldc, 0.0, H.Sct_ + r0 + j0 * H.dcap_, H.dcap_));
    timed();
    if (dRows) {
      ck(hssk_dgemm(H.ctx_, 1, dn, nloc, N, 1.0, H.Rt_ + r0, H.dcap_, dRows, ldr, 0.0, H.Srt_ + r0 + j0 * H.dcap_, H.dcap_));
      timed();
    } else {
      // P (dn x N) = R(cols, :)^T A(:, cols)^T : the contribution of this rank's columns to every row of Sr
      H.tmp_->rewind();
      double* P = H.tmp_->dbl((size_t)dn * N);
      double* mineP = H.tmp_->dbl((size_t)dn * nloc);
      ck(hssk_dgemm(H.ctx_, 1, dn, N, nloc, 1.0, H.Rt_ + r0 + j0 * H.dcap_, H.dcap_, dCols, ldc, 0.0, P, dn));
      timed();
      const double tc = now();
      std::vector<long long> offs(H.o_.world), counts(H.o_.world);
      for (int g = 0; g < H.o_.world; g++) {
        const Node& cg = H.nodes_[single ? 0 : H.cut_nodes_[g]];
        offs[g] = (long long)cg.lo * dn; counts[g] = (long long)cg.m * dn;
      }
      H.reduce_scatter_sum(P, offs, counts, mineP);
      hssk_colgather_desc cp{mineP, H.Srt_ + r0 + j0 * H.dcap_, nullptr, dn, (int)nloc, dn, H.dcap_, 0};
      ck(hssk_gather_cols(H.ctx_, &cp, 1));
      ck(hssk_sync(H.ctx_));
      H.stats_.t_comm += now() - tc;
    }
  }
  void extract(DeviceHSS& H, const std::vector<ElemReq>& reqs) override {
    const Node& c = H.nodes_[H.o_.world == 1 ? 0 : H.cut_nodes_[H.o_.rank]];
    const int j0 = c.lo, j1 = c.lo + c.m;
    auto inside = [&](const std::vector<int>* h, int i0, int cnt) {
      if (!h) return i0 >= j0 && i0 + cnt <= j1;
      for (int i = 0; i < cnt; i++) if ((*h)[i] < j0 || (*h)[i] >= j1) return false;
      return true;
    };
    std::vector<hssk_elem_desc> own, part, back;
    size_t tot = 0;
    for (auto& r : reqs) if (r.m > 0 && r.n > 0 && !(inside(r.hI, r.i0, r.m) && inside(r.hJ, r.j0, r.n))) tot += (size_t)r.m * r.n;
    double* stage = tot ? H.comm_arena_->dbl(tot) : nullptr;
    size_t off = 0;
    for (auto& r : reqs) {
      if (r.m <= 0 || r.n <= 0) continue;
      if (inside(r.hI, r.i0, r.m) && inside(r.hJ, r.j0, r.n)) {
        // column block addressed with global column indices: A'(i, j) = dCols[i + (j - j0) ldc]
        own.push_back(hssk_elem_desc{dCols - (long long)j0 * ldc, ldc, r.dI, r.dJ, r.i0, r.j0, r.dB, r.m, r.n, r.ldb, 0, 0, 0, 0, 0});
      } else {
        double* T = stage + off;
        off += (size_t)r.m * r.n;
        if (dRows) part.push_back(hssk_elem_desc{dRows - j0, ldr, r.dI, r.dJ, r.i0, r.j0, T, r.m, r.n, r.m, 0, j0, j1, 0, 0});
        else part.push_back(hssk_elem_desc{dCols - (long long)j0 * ldc, ldc, r.dI, r.dJ, r.i0, r.j0, T, r.m, r.n, r.m, 0, 0, 0, j0, j1});
        back.push_back(hssk_elem_desc{T, r.m, nullptr, nullptr, 0, 0, r.dB, r.m, r.n, r.ldb, 0, 0, 0, 0, 0});
      }
    }
    if (!own.empty()) ck(hssk_gather_elems(H.ctx_, own.data(), (int)own.size()));
    if (!part.empty()) {
      const double tc = now();
      ck(hssk_gather_elems(H.ctx_, part.data(), (int)part.size()));
      H.allreduce_sum(stage, (long long)tot);
      ck(hssk_gather_elems(H.ctx_, back.data(), (int)back.size()));
      H.stats_.t_comm += now() - tc;
    }
  }
};

// Host-resident operand, streamed: column blocks A(:, c0:c1) cross PCIe once per sampling round through two device
// buffers; the upload of block b+1 (copy stream, pinned bounce buffers filled by host threads) overlaps the two sketch
// GEMMs of block b (compute stream):
//   Sc(c0:c1, :)  = A(:, c0:c1)^H R            -- complete for these columns
//   Sr(:, :)     += A(:, c0:c1) R(c0:c1, :)    -- the block's contribution to every row
// The diagonal blocks and the coupling blocks are read from the host operand afterwards (contiguous 2-D copies, resp. a
// multi-threaded host gather of the few scattered entries + one upload).  At most 2 x n x nb doubles of A are in HBM.
// An operand of another scalar type (float, complex<float>, complex<double>: `dtype`, the reference's other instantiations,
// HSS/HSSMatrix.cpp:513-516) crosses the link in ITS format -- half / a quarter of the bytes of its double-precision real image --
// into two staging buffers; hssk_expand_image writes the image of a block (interleaved [re -im; im re] for complex scalars,
// HSSMatrixPromoted.hpp) into the one block buffer the GEMMs read.
struct DeviceHSS::HostBlockSource : DeviceHSS::Source {
  const void* hA;          // column-major host matrix (scalars of `dtype`), or null when `fill` evaluates the columns
  long long lda;           // in scalars
  int dtype = HSSK_DT_F64;
  const host_fill_t* fill;
  const host_elem_t* elem;
  double* dBuf[2] = {nullptr, nullptr};
  double* dNat[2] = {nullptr, nullptr};   // staging of the native blocks (dtype != HSSK_DT_F64)
  long long nb = 0;
  int gen = -1;   // compression attempt the buffers were carved in (a restart resets the work arena)
  // diagonal blocks of the leaves, copied out of the column blocks while they pass through the device (first sample of an
  // attempt): extract() then serves them from here instead of gathering them from host memory again
  std::vector<double*> dcache;   // by node id
  HostBlockSource(const void* a, long long l, const host_fill_t* f, const host_elem_t* e, int dt = HSSK_DT_F64)
      : hA(a), lda(l), dtype(dt), fill(f), elem(e) {}
  long long reals() const { return dtype == HSSK_DT_C32 || dtype == HSSK_DT_C64 ? 2 : 1; }   // image rows per scalar row
  size_t esize() const { return dtype == HSSK_DT_F32 ? 4 : dtype == HSSK_DT_C64 ? 16 : 8; }
  // entry (I, J) of the image
  double image_at(size_t I, size_t J) const {
    switch (dtype) {
      case HSSK_DT_F32: return (double)((const float*)hA)[I + J * (size_t)lda];
      case HSSK_DT_C32: {
        const float* z = (const float*)hA + 2 * (I / 2 + (J / 2) * (size_t)lda);
        return (I & 1) == (J & 1) ? (double)z[0] : ((I & 1) ? (double)z[1] : -(double)z[1]);
      }
      case HSSK_DT_C64: {
        const double* z = (const double*)hA + 2 * (I / 2 + (J / 2) * (size_t)lda);
        return (I & 1) == (J & 1) ? z[0] : ((I & 1) ? z[1] : -z[1]);
      }
      default: return ((const double*)hA)[I + J * (size_t)lda];
    }
  }
  void sample(DeviceHSS& H, int r0, int dn) override {
    if (H.o_.world > 1) throw std::invalid_argument("host-resident operands are single-GPU (use the device / sharded interfaces)");
    // (an SJLT sketching matrix is applied in its dense form here -- Rt_ holds it, DeviceHSS::fill_random: the streaming
    // SJLT kernels overwrite their output, the blocks of a streamed operand have to accumulate)
    const long long N = H.n_;
    const bool typed = hA && dtype != HSSK_DT_F64;
    const long long W = typed ? reals() : 1, ns = N / W;   // scalar rows
    if (typed && N % W) throw std::logic_error("image dimension of a complex operand must be even");
    const bool first = gen != H.attempt_;
    if (first) {
      gen = H.attempt_;
      // ~1.5 GB per buffer (STRUMPACK_AMD_HOST_BLOCK_MB, or _KB for small operands, to change), whole 64-column tiles of the
      // sketch GEMM
      long long bytes = 1536LL << 20;
      if (const char* e = std::getenv("STRUMPACK_AMD_HOST_BLOCK_MB")) bytes = std::max(1LL, std::atoll(e)) << 20;
      if (const char* e = std::getenv("STRUMPACK_AMD_HOST_BLOCK_KB")) bytes = std::max(1LL, std::atoll(e)) << 10;
      nb = std::max<long long>(64, bytes / (8 * std::max<long long>(N, 1)) / 64 * 64);
      nb = std::min(nb, (N + 63) / 64 * 64);
      dBuf[0] = H.work_->dbl((size_t)N * nb);
      if (!typed) dBuf[1] = H.work_->dbl((size_t)N * nb);
      else for (int k = 0; k < 2; k++) dNat[k] = H.work_->dbl(((size_t)ns * (size_t)(nb / W) * esize() + 7) / 8);
    }
    const long long nblk = (N + nb - 1) / nb;
    const bool capture = first;
    if (capture) {
      dcache.assign(H.nodes_.size(), nullptr);
      for (size_t id = 0; id < H.nodes_.size(); id++)
        if (H.nodes_[id].leaf() && H.nodes_[id].m > 0) dcache[id] = H.work_->dbl((size_t)H.nodes_[id].m * H.nodes_[id].m);
    }
    std::vector<double> tmp;   // columns evaluated by `fill` (packed into the pinned ring before the call returns)
    auto upload = [&](long long b) {
      const long long c0 = b * nb, c1 = std::min(N, c0 + nb);
      if (typed)   // (c0 and c1 are even: nb is a multiple of 64, N = 2 ns)
        ck(hssk_h2d_bytes_async(H.ctx_, dNat[b & 1], (long long)(ns * esize()), (const char*)hA + (size_t)(c0 / W) * lda * esize(),
                                (long long)(lda * esize()), (long long)(ns * esize()), (c1 - c0) / W));
      else if (hA) ck(hssk_h2d_block_async(H.ctx_, dBuf[b & 1], N, (const double*)hA + (size_t)c0 * lda, lda, N, c1 - c0));
      else {
        tmp.resize((size_t)N * (c1 - c0));
        (*fill)(c0, c1, tmp.data());
        ck(hssk_h2d_block_async(H.ctx_, dBuf[b & 1], N, tmp.data(), N, N, c1 - c0));
      }
    };
    upload(0);
    for (long long b = 0; b < nblk; b++) {
      const long long c0 = b * nb, c1 = std::min(N, c0 + nb);
      ck(hssk_copy_fence(H.ctx_));          // the GEMMs below wait for block b
      const double* Ab = typed ? dBuf[0] : dBuf[b & 1];
      if (typed) {
        // the image of block b (behind the GEMMs of block b - 1 in stream order); its staging buffer is free from here on
        ck(hssk_expand_image(H.ctx_, dBuf[0], N, dNat[b & 1], ns, ns, (c1 - c0) / W, dtype));
        ck(hssk_compute_mark(H.ctx_, (int)(b & 1)));
      }
      ck(hssk_dgemm(H.ctx_, 0, dn, c1 - c0, N, 1.0, H.Rt_ + r0, H.dcap_, Ab, N, 0.0, H.Sct_ + r0 + c0 * H.dcap_, H.dcap_));
      ck(hssk_dgemm(H.ctx_, 1, dn, N, c1 - c0, 1.0, H.Rt_ + r0 + c0 * H.dcap_, H.dcap_, Ab, N, b ? 1.0 : 0.0, H.Srt_ + r0, H.dcap_));
      if (capture) {
        // the columns of the leaves' diagonal blocks that lie in this column block
        std::vector<hssk_colgather_desc> dg;
        for (size_t id = 0; id < H.nodes_.size(); id++) {
          const Node& nd = H.nodes_[id];
          if (!dcache[id]) continue;
          const long long a = std::max<long long>(nd.lo, c0), e = std::min<long long>(nd.lo + nd.m, c1);
          if (a >= e) continue;
          dg.push_back(hssk_colgather_desc{Ab + nd.lo + (size_t)(a - c0) * N, dcache[id] + (size_t)(a - nd.lo) * nd.m, nullptr, nd.m, (int)(e - a),
                                           (int)N, nd.m, 0});
        }
        if (!dg.empty()) ck(hssk_gather_cols(H.ctx_, dg.data(), (int)dg.size()));
      }
      // block b + 1 overwrites the buffer the work of block b - 1 read -- and only that: the upload (whose packing blocks
      // this thread for most of its duration) is issued AFTER the GEMMs of block b, which then run under it, and it does
      // not wait for them.  (Issued before them, every block stalled the copy stream for the 2.4 ms of its GEMMs.)
      if (!typed) ck(hssk_compute_mark(H.ctx_, (int)(b & 1)));
      if (b + 1 < nblk) {
        ck(hssk_copy_wait(H.ctx_, (int)((b + 1) & 1)));
        upload(b + 1);
      }
    }
    ck(hssk_sync(H.ctx_));
  }
  void extract(DeviceHSS& H, const std::vector<ElemReq>& reqs) override {
    // every requested block is compact (ldb == m): gather on the host threads into one staging image, one upload each
    std::vector<size_t> off(reqs.size() + 1, 0);
    for (size_t k = 0; k < reqs.size(); k++) off[k + 1] = off[k] + (size_t)std::max(reqs[k].m, 0) * std::max(reqs[k].n, 0);
    // blocks served from the device-side cache of leaf diagonal blocks
    std::vector<const double*> hit(reqs.size(), nullptr);
    if (!dcache.empty()) {
      std::vector<std::pair<int, size_t>> by_lo;   // (lo, node) of the cached leaves, for the lookups below
      for (size_t id = 0; id < H.nodes_.size(); id++) if (dcache[id]) by_lo.push_back({H.nodes_[id].lo, id});
      std::sort(by_lo.begin(), by_lo.end());
      std::vector<hssk_colgather_desc> cp;
      for (size_t k = 0; k < reqs.size(); k++) {
        const ElemReq& r = reqs[k];
        if (r.hI || r.hJ || r.i0 != r.j0 || r.m != r.n || r.m <= 0 || gen != H.attempt_) continue;
        auto it = std::lower_bound(by_lo.begin(), by_lo.end(), std::make_pair(r.i0, size_t(0)));
        if (it == by_lo.end() || it->first != r.i0 || H.nodes_[it->second].m != r.m) continue;
        hit[k] = dcache[it->second];
        cp.push_back(hssk_colgather_desc{hit[k], r.dB, nullptr, r.m, r.n, r.m, r.ldb, 0});
      }
      if (!cp.empty()) ck(hssk_gather_cols(H.ctx_, cp.data(), (int)cp.size()));
    }
    for (size_t k = 0; k < reqs.size(); k++)
      if (hit[k]) off[k + 1] = off[k];   // (no staging space for served requests)
      else off[k + 1] = off[k] + (size_t)std::max(reqs[k].m, 0) * std::max(reqs[k].n, 0);
    std::unique_ptr<double[]> img_store(new double[std::max<size_t>(off.back(), 1)]);   // (not value-initialised: every element is written)
    struct { double* p; double* data() { return p; } } img{img_store.get()};
    host_parallel_for(reqs.size(), [&](size_t k) {
      const ElemReq& r = reqs[k];
      if (r.m <= 0 || r.n <= 0 || hit[k]) return;
      double* B = img.data() + off[k];
      if (hA && dtype != HSSK_DT_F64) {
        for (int j = 0; j < r.n; j++) {
          const size_t J = (size_t)(r.hJ ? (*r.hJ)[j] : r.j0 + j);
          for (int i = 0; i < r.m; i++) B[i + (size_t)j * r.m] = image_at((size_t)(r.hI ? (*r.hI)[i] : r.i0 + i), J);
        }
      } else if (hA) {
        for (int j = 0; j < r.n; j++) {
          const double* col = (const double*)hA + (size_t)(r.hJ ? (*r.hJ)[j] : r.j0 + j) * lda;
          if (r.hI) for (int i = 0; i < r.m; i++) B[i + (size_t)j * r.m] = col[(*r.hI)[i]];
          else std::memcpy(B + (size_t)j * r.m, col + r.i0, sizeof(double) * r.m);
        }
      } else {
        std::vector<int> I(r.m), J(r.n);
        for (int i = 0; i < r.m; i++) I[i] = r.hI ? (*r.hI)[i] : r.i0 + i;
        for (int j = 0; j < r.n; j++) J[j] = r.hJ ? (*r.hJ)[j] : r.j0 + j;
        (*elem)(r.m, I.data(), r.n, J.data(), B, r.m);
      }
    });
    for (size_t k = 0; k < reqs.size(); k++) {
      const ElemReq& r = reqs[k];
      if (r.m <= 0 || r.n <= 0 || hit[k]) continue;
      if (r.ldb == r.m) ck(hssk_upload_async(H.ctx_, r.dB, img.data() + off[k], (long long)(sizeof(double) * (off[k + 1] - off[k]))));
      else ck(hssk_memcpy2d_h2d(H.ctx_, r.dB, sizeof(double) * r.ldb, img.data() + off[k], sizeof(double) * r.m, sizeof(double) * r.m, r.n));
    }
  }
};

struct DeviceHSS::CallbackSource : DeviceHSS::Source {
  const host_mult_t& mult;
  const host_elem_t& elem;
  CallbackSource(const host_mult_t& m, const host_elem_t& e) : mult(m), elem(e) {}
  void sample(DeviceHSS& H, int r0, int dn) override {
    if (H.o_.world > 1) throw std::invalid_argument("the host-callback interface is single-GPU");
    const int N = H.n_;
    std::vector<double> Rt((size_t)dn * N), R((size_t)N * dn), S((size_t)N * dn), St((size_t)dn * N);
    ck(hssk_memcpy2d_d2h(H.ctx_, Rt.data(), sizeof(double) * dn, H.Rt_ + r0, sizeof(double) * H.dcap_, sizeof(double) * dn, N));
    for (int j = 0; j < N; j++) for (int i = 0; i < dn; i++) R[j + (size_t)i * N] = Rt[i + (size_t)j * dn];
    for (int pass = 0; pass < 2; pass++) {
      mult(pass == 0 ? 'N' : 'C', N, dn, R.data(), N, S.data(), N);
      for (int j = 0; j < N; j++) for (int i = 0; i < dn; i++) St[i + (size_t)j * dn] = S[j + (size_t)i * N];
      double* dst = (pass == 0 ? H.Srt_ : H.Sct_) + r0;
      ck(hssk_memcpy2d_h2d(H.ctx_, dst, sizeof(double) * H.dcap_, St.data(), sizeof(double) * dn, sizeof(double) * dn, N));
    }
  }
  void extract(DeviceHSS& H, const std::vector<ElemReq>& reqs) override {
    for (auto& r : reqs) {
      if (r.m <= 0 || r.n <= 0) continue;
      std::vector<int> I(r.m), J(r.n);
      for (int i = 0; i < r.m; i++) I[i] = r.hI ? (*r.hI)[i] : r.i0 + i;
      for (int j = 0; j < r.n; j++) J[j] = r.hJ ? (*r.hJ)[j] : r.j0 + j;
      std::vector<double> B((size_t)r.m * r.n);
      elem(r.m, I.data(), r.n, J.data(), B.data(), r.m);
      ck(hssk_memcpy2d_h2d(H.ctx_, r.dB, sizeof(double) * r.ldb, B.data(), sizeof(double) * r.m, sizeof(double) * r.m, r.n));
    }
  }
};

// ---------------------------------------------------------------------------------------------
// construction
// ---------------------------------------------------------------------------------------------
DeviceHSS::DeviceHSS(int n, const EngineOptions& opts, const structured::ClusterTree* tree) : n_(n), o_(opts) {
  ck(hssk_ctx_create(&ctx_, o_.device));
  persist_.reset(new Arena(size_t(64) << 20));
  work_.reset(new Arena(size_t(256) << 20));
  fact_.reset(new Arena(size_t(64) << 20));
  tmp_.reset(new Arena(size_t(64) << 20));
  comm_arena_.reset(new Arena(size_t(64) << 20));
  plan_arena_.reset(new Arena(size_t(64) << 20));
  schur_.reset(new Arena(size_t(64) << 20));
  build_tree(tree);
  setup_ownership();
}

DeviceHSS::~DeviceHSS() {
  if (ctx_) hssk_sync(ctx_);
  drop_plans();
  plan_arena_.reset();
  persist_.reset();
  work_.reset();
  fact_.reset();
  tmp_.reset();
  comm_arena_.reset();
  schur_.reset();
  hssk_ctx_destroy(ctx_);
}

void DeviceHSS::build_tree(const structured::ClusterTree* tree) {
  nodes_.clear();
  // pre-order; HSSMatrix.cpp:60-70 (bisection while size > leaf) or :72-82 (given cluster tree)
  std::function<int(int, int, int, int, const structured::ClusterTree*)> rec =
      [&](int lo, int m, int lvl, int parent, const structured::ClusterTree* t) -> int {
    int id = (int)nodes_.size();
    nodes_.emplace_back();
    nodes_[id].lo = lo; nodes_[id].m = m; nodes_[id].lvl = lvl; nodes_[id].parent = parent;
    bool split = t ? !t->c.empty() : (m > o_.leaf_size);
    if (split) {
      int m0 = t ? t->c[0].size : m / 2;
      int c0 = rec(lo, m0, lvl + 1, id, t ? &t->c[0] : nullptr);
      int c1 = rec(lo + m0, m - m0, lvl + 1, id, t ? &t->c[1] : nullptr);
      nodes_[id].c0 = c0; nodes_[id].c1 = c1;
      nodes_[id].height = 1 + std::max(nodes_[c0].height, nodes_[c1].height);
    }
    return id;
  };
  if (tree && tree->size != n_) throw std::invalid_argument("cluster tree size does not match the matrix dimension");
  rec(0, n_, 0, -1, tree);
  int H = nodes_[0].height, Dp = 0;
  for (auto& nd : nodes_) Dp = std::max(Dp, nd.lvl);
  by_height_.assign(H + 1, {});
  by_depth_.assign(Dp + 1, {});
  for (int i = 0; i < (int)nodes_.size(); i++) {
    by_height_[nodes_[i].height].push_back(i);
    by_depth_[nodes_[i].lvl].push_back(i);
  }
  d_ranks_ = persist_->ints(2 * nodes_.size() + 2);
}


// ---------------------------------------------------------------------------------------------
// multi-GPU: subtree ownership.  With G = 2^c ranks and a tree that is complete down to depth c,
// rank g owns the subtree rooted at the g-th node of depth c (sketch columns, compression, ULV
// factors, solve / apply sweeps of that subtree: no communication); the 2^c - 1 nodes above the cut
// are processed redundantly by every rank after one small all-gather of the cut nodes' reduced blocks
// per phase (the reference's MPI code splits the tree the same way,
// HSSMatrixMPI.hpp:408-414).  Otherwise (G not a power of two / shallow tree) only the sketch is
// sharded and the whole tree is replicated.
// ---------------------------------------------------------------------------------------------
void DeviceHSS::setup_ownership() {
  const size_t nn = nodes_.size();
  owner_.assign(nn, -1);
  cut_nodes_.clear();
  dist_subtree_ = false;
  const int G = o_.world;
  if (G > 1 && (G & (G - 1)) == 0) {
    int c = 0;
    while ((1 << c) < G) c++;
    std::vector<int> cut;
    bool ok = true;
    for (size_t i = 0; i < nn; i++) {
      if (nodes_[i].lvl == c) cut.push_back((int)i);
      if (nodes_[i].lvl < c && nodes_[i].leaf()) ok = false;
    }
    if (ok && (int)cut.size() == G) {
      dist_subtree_ = true;
      cut_nodes_ = cut;  // pre-order == left-to-right
      for (size_t i = 0; i < nn; i++) {
        if (nodes_[i].lvl < c) continue;
        int a = (int)i;
        while (nodes_[a].lvl > c) a = nodes_[a].parent;
        for (int g = 0; g < G; g++) if (cut[g] == a) owner_[i] = g;
      }
    }
  }
  auto split = [&](const std::vector<std::vector<int>>& all, std::vector<std::vector<int>>& own,
                   std::vector<std::vector<int>>& top) {
    own.assign(all.size(), {});
    top.assign(all.size(), {});
    for (size_t l = 0; l < all.size(); l++)
      for (int id : all[l]) {
        if (!dist_subtree_) own[l].push_back(id);
        else if (owner_[id] < 0) top[l].push_back(id);
        else if (owner_[id] == o_.rank) own[l].push_back(id);
      }
  };
  split(by_height_, own_by_height_, top_by_height_);
  split(by_depth_, own_by_depth_, top_by_depth_);
}

// STRUMPACK_AMD_TIME_COMM=1: bracket every collective with stream synchronisations and add its wall time to
// stats().t_comm (diagnostic: the stream-ordered collectives are otherwise invisible to the host clock)
static bool time_comm() {
  const char* e = std::getenv("STRUMPACK_AMD_TIME_COMM");
  return e && e[0] == '1';
}

void DeviceHSS::comm(void* dbuf, long long bytes_per_rank) {
  if (o_.world <= 1) return;
  if (o_.allgather_stream) {   // RCCL on the engine's stream: ordered with the kernels, no host synchronisation
    if (time_comm()) {
      ck(hssk_sync(ctx_));
      const double t0 = now();
      o_.allgather_stream(o_.comm_user, dbuf, bytes_per_rank, hssk_ctx_stream(ctx_));
      ck(hssk_sync(ctx_));
      stats_.t_comm += now() - t0;
      return;
    }
    o_.allgather_stream(o_.comm_user, dbuf, bytes_per_rank, hssk_ctx_stream(ctx_));
    return;
  }
  if (!o_.allgather) throw std::logic_error("multi-GPU operation needs an all-gather hook");
  ck(hssk_sync(ctx_));
  const double t0 = now();
  o_.allgather(o_.comm_user, dbuf, bytes_per_rank);   // (host-synchronous by contract)
  if (time_comm()) stats_.t_comm += now() - t0;
}

// dbuf[0:count) <- sum over the ranks
void DeviceHSS::allreduce_sum(double* dbuf, long long count) {
  if (o_.world <= 1 || count <= 0) return;
  if (o_.allreduce_stream) { o_.allreduce_stream(o_.comm_user, dbuf, count, hssk_ctx_stream(ctx_)); return; }
  // all-gather hook only: gather every rank's partial block, sum locally
  double* slabs = comm_arena_->dbl((size_t)count * o_.world);
  ck(hssk_memcpy_d2d(ctx_, slabs + (size_t)count * o_.rank, dbuf, (long long)sizeof(double) * count));
  comm(slabs, (long long)sizeof(double) * count);
  ck(hssk_sum_slabs(ctx_, slabs, count, count, o_.world, dbuf));
}

// recv[0:counts[me]) <- sum over the ranks of send[offs[me] : offs[me] + counts[me])
void DeviceHSS::reduce_scatter_sum(const double* send, const std::vector<long long>& offs, const std::vector<long long>& counts,
                                   double* recv) {
  const int me = o_.rank;
  if (o_.world <= 1) { ck(hssk_memcpy_d2d(ctx_, recv, send + offs[me], (long long)sizeof(double) * counts[me])); return; }
  if (o_.reduce_scatter_stream) {
    o_.reduce_scatter_stream(o_.comm_user, send, offs.data(), counts.data(), recv, hssk_ctx_stream(ctx_));
    return;
  }
  // all-gather hook only: every rank publishes the slice each other rank needs (padded to the largest slice)
  long long cmax = 0;
  for (long long c : counts) cmax = std::max(cmax, c);
  const int G = o_.world;
  // slab layout: [destination g][source rank] blocks of cmax doubles; one all-gather per destination keeps the hook simple
  double* slabs = comm_arena_->dbl((size_t)cmax * G);
  for (int g = 0; g < G; g++) {
    if (counts[g] <= 0) continue;
    ck(hssk_memset_zero(ctx_, slabs + (size_t)cmax * me, (long long)sizeof(double) * cmax));
    ck(hssk_memcpy_d2d(ctx_, slabs + (size_t)cmax * me, send + offs[g], (long long)sizeof(double) * counts[g]));
    comm(slabs, (long long)sizeof(double) * cmax);
    if (g == me) ck(hssk_sum_slabs(ctx_, slabs, counts[g], cmax, G, recv));
    ck(hssk_sync(ctx_));
  }
}

// v holds world * per_rank ints; this rank's block is valid on entry, all blocks on return
void DeviceHSS::allgather_ints(std::vector<int>& v, int per_rank) {
  const size_t bytes = sizeof(int) * (size_t)per_rank;
  int* d = comm_arena_->ints((size_t)per_rank * o_.world);
  ck(hssk_memcpy_h2d(ctx_, d + (size_t)per_rank * o_.rank, v.data() + (size_t)per_rank * o_.rank, (long long)bytes));
  comm(d, (long long)bytes);
  ck(hssk_memcpy_d2h(ctx_, v.data(), d, (long long)(bytes * o_.world)));
}

// after the owned subtrees of a compression round: publish the cut nodes to every rank
void DeviceHSS::exchange_cut_compress(int dtot) {
  const int G = o_.world, me = o_.rank;
  std::vector<int> meta(4 * (size_t)G, 0);
  {
    const Node& c = nodes_[cut_nodes_[me]];
    meta[4 * me] = c.Ustate; meta[4 * me + 1] = c.Vstate; meta[4 * me + 2] = c.rU; meta[4 * me + 3] = c.rV;
  }
  allgather_ints(meta, 4);
  int rmax = 0;
  for (int g = 0; g < G; g++) {
    Node& c = nodes_[cut_nodes_[g]];
    if (g != me) { c.Ustate = meta[4 * g]; c.Vstate = meta[4 * g + 1]; c.rU = meta[4 * g + 2]; c.rV = meta[4 * g + 3]; }
    if (c.compressed()) rmax = std::max(rmax, std::max(c.rU, c.rV));
  }
  if (rmax == 0) return;
  // index sets (host -> all ranks)
  std::vector<int> idx(2 * (size_t)rmax * G, 0);
  {
    const Node& c = nodes_[cut_nodes_[me]];
    if (c.compressed()) {
      std::copy(c.Ir.begin(), c.Ir.end(), idx.begin() + 2 * (size_t)rmax * me);
      std::copy(c.Ic.begin(), c.Ic.end(), idx.begin() + 2 * (size_t)rmax * me + rmax);
    }
  }
  allgather_ints(idx, 2 * rmax);
  // the exchange buffers are carved once per compression attempt and reused by the adaptive rounds (every round re-sends
  // the panels of all samples and re-points the remote cut nodes, so nothing of the previous round is read again); a round
  // that needs more room -- the ranks grew -- carves a larger pair
  const size_t need_idx = 2 * (size_t)rmax * G + rmax, need_buf = 4 * (size_t)dcap_ * rmax * G;
  if (cut_gen_ != attempt_ || need_idx > cut_idx_cap_ || need_buf > cut_buf_cap_) {
    const bool grow = cut_gen_ == attempt_;   // (a new attempt starts from a reset arena: the old pair is gone)
    cut_gen_ = attempt_;
    cut_idx_cap_ = std::max(need_idx, grow ? 2 * cut_idx_cap_ : size_t(0));
    cut_buf_cap_ = std::max(need_buf, grow ? 2 * cut_buf_cap_ : size_t(0));
    cut_idx_ = work_->ints(cut_idx_cap_);
    cut_buf_ = work_->dbl(cut_buf_cap_);
  }
  int* didx = cut_idx_;
  ck(hssk_memcpy_h2d(ctx_, didx, idx.data(), (long long)(sizeof(int) * idx.size())));
  std::vector<int> iota(rmax);
  for (int i = 0; i < rmax; i++) iota[i] = i;
  int* diota = didx + 2 * (size_t)rmax * G;
  ck(hssk_memcpy_h2d(ctx_, diota, iota.data(), (long long)(sizeof(int) * rmax)));
  // panels: [Srt(:, Jr) | Sct(:, Jc) | RrtRed | RctRed], each dcap x rmax, leading dimension dcap
  const size_t pan = (size_t)dcap_ * rmax, blk = 4 * pan;
  double* buf = cut_buf_;
  {
    Node& c = nodes_[cut_nodes_[me]];
    if (c.compressed()) {
      double* slot = buf + blk * me;
      std::vector<hssk_colgather_desc> g;
      g.push_back(hssk_colgather_desc{c.Srt, slot, c.permU, dtot, c.rU, dcap_, dcap_, 0});
      g.push_back(hssk_colgather_desc{c.Sct, slot + pan, c.permV, dtot, c.rV, dcap_, dcap_, 0});
      g.push_back(hssk_colgather_desc{c.RrtRed, slot + 2 * pan, nullptr, dtot, c.rV, dcap_, dcap_, 0});
      g.push_back(hssk_colgather_desc{c.RctRed, slot + 3 * pan, nullptr, dtot, c.rU, dcap_, dcap_, 0});
      ck(hssk_gather_cols(ctx_, g.data(), (int)g.size()));
    }
  }
  comm(buf, (long long)(sizeof(double) * blk));
  for (int g = 0; g < G; g++) {
    if (g == me) continue;
    Node& c = nodes_[cut_nodes_[g]];
    if (!c.compressed()) continue;
    double* slot = buf + blk * g;
    c.Srt = slot; c.Sct = slot + pan; c.RrtRed = slot + 2 * pan; c.RctRed = slot + 3 * pan;
    c.permU = c.permV = diota;  // the received panels hold the skeleton columns only, in order
    c.dIr = didx + 2 * (size_t)rmax * g;
    c.dIc = c.dIr + rmax;
    c.Ir.assign(idx.begin() + 2 * (size_t)rmax * g, idx.begin() + 2 * (size_t)rmax * g + c.rU);
    c.Ic.assign(idx.begin() + 2 * (size_t)rmax * g + rmax, idx.begin() + 2 * (size_t)rmax * g + rmax + c.rV);
    c.panels = true;
  }
}

// kernel-matrix compression: publish the cut nodes (rank, skeleton ids, column set) to every rank; returns the
// all-ranks OR of `failed` so that every process takes the same decision about another round
bool DeviceHSS::exchange_cut_kernel(std::vector<std::vector<int>>& cols, bool failed) {
  const int G = o_.world, me = o_.rank;
  std::vector<int> meta(4 * (size_t)G, 0);
  {
    const Node& c = nodes_[cut_nodes_[me]];
    meta[4 * me] = c.compressed(); meta[4 * me + 1] = c.rU; meta[4 * me + 2] = (int)cols[cut_nodes_[me]].size(); meta[4 * me + 3] = failed;
  }
  allgather_ints(meta, 4);
  int rmax = 0, cmax = 0;
  bool any_failed = false;
  for (int g = 0; g < G; g++) {
    rmax = std::max(rmax, meta[4 * g + 1]); cmax = std::max(cmax, meta[4 * g + 2]);
    any_failed = any_failed || meta[4 * g + 3] || !meta[4 * g];
  }
  if (any_failed) return true;
  const size_t per = (size_t)rmax + cmax;
  if (per == 0) return false;
  std::vector<int> idx(per * G, 0);
  {
    const Node& c = nodes_[cut_nodes_[me]];
    std::copy(c.Ir.begin(), c.Ir.end(), idx.begin() + per * me);
    std::copy(cols[cut_nodes_[me]].begin(), cols[cut_nodes_[me]].end(), idx.begin() + per * me + rmax);
  }
  allgather_ints(idx, (int)per);
  int* didx = work_->ints(per * G);
  ck(hssk_memcpy_h2d(ctx_, didx, idx.data(), (long long)(sizeof(int) * idx.size())));
  for (int g = 0; g < G; g++) {
    if (g == me) continue;
    Node& c = nodes_[cut_nodes_[g]];
    c.rU = c.rV = meta[4 * g + 1];
    c.Ustate = c.Vstate = 2;
    c.Ir.assign(idx.begin() + per * g, idx.begin() + per * g + c.rU);
    c.Ic = c.Ir;
    c.dIr = c.dIc = didx + per * g;
    cols[cut_nodes_[g]].assign(idx.begin() + per * g + rmax, idx.begin() + per * g + rmax + meta[4 * g + 2]);
  }
  return false;
}

// ranks / basis sizes of every node, for introspection and buffer sizing on all ranks
void DeviceHSS::exchange_node_table() {
  const size_t nn = nodes_.size();
  std::vector<int> t(4 * nn * (size_t)o_.world, 0);
  int* mineblk = t.data() + 4 * nn * (size_t)o_.rank;
  for (size_t i = 0; i < nn; i++)
    if (owner_[i] == o_.rank) { mineblk[4 * i] = nodes_[i].rU; mineblk[4 * i + 1] = nodes_[i].rV; mineblk[4 * i + 2] = nodes_[i].mU; mineblk[4 * i + 3] = nodes_[i].mV; }
  allgather_ints(t, (int)(4 * nn));
  for (size_t i = 0; i < nn; i++) {
    const int g = owner_[i];
    if (g < 0 || g == o_.rank) continue;
    const int* b = t.data() + 4 * nn * (size_t)g + 4 * i;
    nodes_[i].rU = b[0]; nodes_[i].rV = b[1]; nodes_[i].mU = b[2]; nodes_[i].mV = b[3];
    nodes_[i].Ustate = nodes_[i].Vstate = 2;
  }
}

// after the owned subtrees of the ULV factorization: Dt (rU x rU) and Vt1 (rU x rV) of the cut nodes
void DeviceHSS::exchange_cut_factor() {
  const int G = o_.world, me = o_.rank;
  size_t blk = 1;
  for (int g = 0; g < G; g++) {
    const Node& c = nodes_[cut_nodes_[g]];
    blk = std::max(blk, (size_t)c.rU * c.rU + (size_t)c.rU * c.rV);
  }
  double* buf = fact_->dbl(blk * G);
  {
    const Node& c = nodes_[cut_nodes_[me]];
    double* slot = buf + blk * me;
    if (c.rU) ck(hssk_memcpy_d2d(ctx_, slot, c.Dt, (long long)(sizeof(double) * c.rU * c.rU)));
    if (c.rU && c.rV) ck(hssk_memcpy_d2d(ctx_, slot + (size_t)c.rU * c.rU, c.Vt1, (long long)(sizeof(double) * c.rU * c.rV)));
  }
  comm(buf, (long long)(sizeof(double) * blk));
  for (int g = 0; g < G; g++) {
    if (g == me) continue;
    Node& c = nodes_[cut_nodes_[g]];
    c.Dt = buf + blk * g;
    c.Vt1 = c.Dt + (size_t)c.rU * c.rU;
  }
}

// every rank holds its own row range of dx (n x nrhs, ldx): make all ranges available everywhere
void DeviceHSS::allgather_rows(double* dx, long long ldx, int nrhs) {
  const int G = o_.world, me = o_.rank;
  int mmax = 0;
  for (int g = 0; g < G; g++) mmax = std::max(mmax, nodes_[cut_nodes_[g]].m);
  const size_t blk = (size_t)mmax * nrhs;
  double* buf = tmp_->dbl(blk * G);
  const Node& c = nodes_[cut_nodes_[me]];
  hssk_rowgather_desc pk{dx + c.lo, buf + blk * me, nullptr, c.m, nrhs, (int)ldx, mmax, 0, 0};
  ck(hssk_gather_rows(ctx_, &pk, 1));
  comm(buf, (long long)(sizeof(double) * blk));
  std::vector<hssk_rowgather_desc> up;
  for (int g = 0; g < G; g++) {
    if (g == me) continue;
    const Node& o = nodes_[cut_nodes_[g]];
    up.push_back(hssk_rowgather_desc{buf + blk * g, dx + o.lo, nullptr, o.m, nrhs, mmax, (int)ldx, 0, 0});
  }
  if (!up.empty()) ck(hssk_gather_rows(ctx_, up.data(), (int)up.size()));
  ck(hssk_sync(ctx_));
}

bool DeviceHSS::is_compressed() const { return nodes_[0].compressed(); }
int DeviceHSS::levels() const { return nodes_[0].height + 1; }
int DeviceHSS::rank() const { return rank(0); }
long long DeviceHSS::nonzeros() const { return nonzeros(0); }
long long DeviceHSS::memory() const { return memory(0); }
// the same over the sub-tree of a node (HSSMatrix::child(c)->rank() ...; the node's own basis belongs to it, as in the
// reference, where a child carries its U and V)
int DeviceHSS::rank(int node) const {
  int r = 0;
  for (int i = node, e = subtree_end(node); i < e; i++) r = std::max(r, std::max(nodes_[i].rU, nodes_[i].rV));
  return r;
}
long long DeviceHSS::nonzeros(int node) const {
  long long t = 0;
  for (int i = node, e = subtree_end(node); i < e; i++) {
    const Node& nd = nodes_[i];
    if (nd.leaf()) t += (long long)nd.m * nd.m;
    else t += (long long)nodes_[nd.c0].rU * nodes_[nd.c1].rV + (long long)nodes_[nd.c1].rU * nodes_[nd.c0].rV;
    if (nd.lvl > 0) t += (long long)nd.rU * (nd.mU - nd.rU) + nd.mU + (long long)nd.rV * (nd.mV - nd.rV) + nd.mV;
  }
  return t;
}
long long DeviceHSS::memory(int node) const {
  long long t = 0;
  for (int i = node, e = subtree_end(node); i < e; i++) {
    const Node& nd = nodes_[i];
    if (nd.leaf()) t += 8LL * nd.m * nd.m;
    else t += 8LL * ((long long)nodes_[nd.c0].rU * nodes_[nd.c1].rV + (long long)nodes_[nd.c1].rU * nodes_[nd.c0].rV);
    if (nd.lvl > 0) t += 8LL * ((long long)nd.rU * (nd.mU - nd.rU) + (long long)nd.rV * (nd.mV - nd.rV)) + 4LL * (nd.mU + nd.mV);
  }
  return t;
}
long long DeviceHSS::factor_memory() const { return (long long)fact_->used(); }
void DeviceHSS::node_info(int* out) const {
  for (size_t i = 0; i < nodes_.size(); i++) {
    const Node& nd = nodes_[i];
    int* o = out + 6 * i;
    o[0] = nd.lo; o[1] = nd.m; o[2] = nd.lvl ? nd.mU : 0; o[3] = nd.rU; o[4] = nd.rV; o[5] = nd.leaf();
  }
}
void DeviceHSS::ensure_ready(const char* what) const {
  if (!is_compressed()) throw std::logic_error(std::string(what) + ": the HSS matrix is not compressed");
}

// ---------------------------------------------------------------------------------------------
// compression driver
// ---------------------------------------------------------------------------------------------
void DeviceHSS::compress_dense_device(const double* dA, long long lda) {
  DenseDeviceSource s(dA, lda);
  compress(s);
}
void DeviceHSS::compress_dense_host(const double* A, long long lda) {
  HostBlockSource s(A, lda, nullptr, nullptr);
  compress(s);
}
void DeviceHSS::compress_dense_host_typed(const void* A, long long lda, int dtype) {
  if (dtype != HSSK_DT_F32 && dtype != HSSK_DT_C32 && dtype != HSSK_DT_C64) throw std::invalid_argument("compress_dense_host_typed: unknown scalar type");
  HostBlockSource s(A, lda, nullptr, nullptr, dtype);
  compress(s);
}
void DeviceHSS::compress_host_blocks(const host_fill_t& fill, const host_elem_t& elem) {
  HostBlockSource s(nullptr, 0, &fill, &elem);
  compress(s);
}
void DeviceHSS::compress_dense_device_sharded(const double* dRows, long long ldr, const double* dCols, long long ldc) {
  if (!dCols) throw std::invalid_argument("sharded operand: the column block is required");
  ShardedDenseSource s(dRows, ldr, dCols, ldc);
  compress(s);
}
bool DeviceHSS::shard_range(int rank, int& lo, int& hi) const {
  if (o_.world == 1) { lo = 0; hi = n_; return true; }
  if (!dist_subtree_ || rank < 0 || rank >= (int)cut_nodes_.size()) return false;
  lo = nodes_[cut_nodes_[rank]].lo;
  hi = lo + nodes_[cut_nodes_[rank]].m;
  return true;
}
void DeviceHSS::compress_callbacks(const host_mult_t& mult, const host_elem_t& elem) {
  CallbackSource s(mult, elem);
  compress(s);
}

void DeviceHSS::drop_plans() {
  for (auto& kv : plans_) if (kv.second.plan) hssk_plan_destroy(kv.second.plan);
  plans_.clear();
  if (plan_arena_) plan_arena_->reset();
}
bool DeviceHSS::plans_enabled() const {
  static const bool off = [] { const char* e = std::getenv("STRUMPACK_AMD_NO_PLANS"); return e && e[0] == '1'; }();
  return !off;
}

void DeviceHSS::reset_compression() {
  book_ = PendingBook();   // (a compression that threw may have left an ID commit half done)
  defer_book_ = false;
  drop_plans();
  for (auto& nd : nodes_) {
    int lo = nd.lo, m = nd.m, lvl = nd.lvl, h = nd.height, c0 = nd.c0, c1 = nd.c1, p = nd.parent;
    nd = Node();
    nd.lo = lo; nd.m = m; nd.lvl = lvl; nd.height = h; nd.c0 = c0; nd.c1 = c1; nd.parent = p;
  }
  persist_->reset();
  work_->reset();
  fact_->reset();
  factored_ = false;
  d_ranks_ = persist_->ints(2 * nodes_.size() + 2);
}

// hard restart (compress.hpp:289-294, reset()): every node back to UNTOUCHED, the first d_have sample rows of Srt_ / Sct_
// back to what the sampling produced; the sample arrays themselves are kept
void DeviceHSS::restart_nodes(int d_have) {
  ck(hssk_sync(ctx_));
  for (auto& nd : nodes_) {
    int lo = nd.lo, m = nd.m, lvl = nd.lvl, h = nd.height, c0 = nd.c0, c1 = nd.c1, p = nd.parent;
    nd = Node();
    nd.lo = lo; nd.m = m; nd.lvl = lvl; nd.height = h; nd.c0 = c0; nd.c1 = c1; nd.parent = p;
  }
  persist_->reset();
  d_ranks_ = persist_->ints(2 * nodes_.size() + 2);
  if (d_have > 0) {
    std::vector<hssk_colgather_desc> cp;
    const int ncols = (int)std::min<long long>(n_, 0x7fffffff);
    cp.push_back(hssk_colgather_desc{Srt0_, Srt_, nullptr, d_have, ncols, dcap_, dcap_, 0});
    cp.push_back(hssk_colgather_desc{Sct0_, Sct_, nullptr, d_have, ncols, dcap_, dcap_, 0});
    ck(hssk_gather_cols(ctx_, cp.data(), 2));
  }
}

void DeviceHSS::free_compress_workspace() {
  ck(hssk_sync(ctx_));
  work_->reset();
  Rt_ = Srt_ = Sct_ = Srt0_ = Sct0_ = nullptr;
  sj_pat_ = nullptr;
  for (auto& nd : nodes_) { nd.Srt = nd.Sct = nd.Rrt = nd.Rct = nd.RrtRed = nd.RctRed = nd.Qr = nd.Qc = nullptr; nd.panels = false; }
}

void DeviceHSS::compress(Source& src) {
  OpGuard op_guard(op_mu_);
  double t0 = now();
  stats_ = PhaseStats();
  int dcap = o_.algorithm != 1 ? o_.d0 + o_.p : o_.d0 + o_.dd;
  dcap = std::max(16, (dcap + 15) / 16 * 16);
  for (;;) {
    if (compress_attempt(src, dcap)) break;
    dcap *= 2;  // the sample capacity was too small: restart (the random stream is seeded, so the
                // restarted run retraces the same samples and continues past the old capacity)
    if (o_.verbose) std::cout << "# HSS compression: growing the sample capacity to " << dcap << std::endl;
  }
  if (dist_subtree_) exchange_node_table();
  free_compress_workspace();
  comm_arena_->reset();
  stats_.t_compress = now() - t0;
  stats_.t_tree = stats_.t_compress - stats_.t_sketch - stats_.t_random;
}

void DeviceHSS::fill_random(int r0, int dn) {
  double t0 = now();
  const long long N = n_;
  sj_pat_ = nullptr;
  if (o_.sketch == 1) {
    // SJLT (HSSMatrix.compress_stable.hpp:39-97, HSSMatrix.sketch.hpp): every row of the N x dn block gets nnz entries
    // +-1 -- nnz0 in the first d0 + dd columns, nnz in each further block (S.add_columns / SJLTMatrix(g, nnz, n, dnew)).
    // CHUNK (sketch.hpp:419-441): one nonzero in each of nnz chunks of dn / nnz columns; PERM (:316-341): the first
    // nnz entries of a random permutation of the columns (drawn here as a partial Fisher-Yates shuffle).  Only the
    // pattern (nnz ints per row) crosses PCIe; the dense block the tree levels need is expanded on the device.
    if (r0 == 0 || !rng_) rng_.reset(new HostRng());
    auto& e = rng_->sj;
    if ((r0 == 0 ? o_.nnz0 : o_.nnz) > 8)
      throw std::invalid_argument("SJLT sketch: more than 8 nonzeros per row (--hss_nnz0 / --hss_nnz) are not supported by the device pattern");
    const int nnz = std::max(1, std::min(r0 == 0 ? o_.nnz0 : o_.nnz, dn));
    const int nq = nnz <= 4 ? 4 : 8;            // ints per row in the device pattern (hssk.h); unused ones point at column dn
    std::vector<int> pat((size_t)nq * N, dn);
    std::uniform_int_distribution<int> sign(0, 1);
    if (o_.sjlt_algo == 0) {
      const int chunk = dn / nnz;
      std::uniform_int_distribution<int> shift(0, chunk - 1);
      for (long long k = 0; k < N; k++)
        for (int q = 0; q < nnz; q++) {
          const int c = shift(e) + chunk * q;
          pat[(size_t)k * nq + q] = sign(e) == 0 ? c : (c | (int)0x80000000);
        }
    } else {
      std::vector<int> cols(dn);
      for (int j = 0; j < dn; j++) cols[j] = j;
      for (long long k = 0; k < N; k++)
        for (int q = 0; q < nnz; q++) {
          std::uniform_int_distribution<int> pick(q, dn - 1);
          std::swap(cols[q], cols[pick(e)]);
          pat[(size_t)k * nq + q] = sign(e) == 0 ? cols[q] : (cols[q] | (int)0x80000000);
        }
    }
    int* dp = work_->ints((size_t)nq * N);
    ck(hssk_memcpy_h2d(ctx_, dp, pat.data(), (long long)sizeof(int) * nq * N));
    ck(hssk_sjlt_dense(ctx_, Rt_ + r0, dn, N, dcap_, dp, nnz));
    sj_pat_ = dp;
    sj_nnz_ = nnz;
  } else if (o_.random_engine == 2) {
    // device Philox: element (sample s, column c) is a pure function of (seed, s * N + c)
    if (o_.random_dist != 0) throw std::invalid_argument("philox engine implements the normal distribution only");
    ck(hssk_randn(ctx_, Rt_ + r0, dn, N, dcap_, r0, N, 0x5354524dull));
  } else {
    // reference-identical host stream: DenseMatrix::random fills the N x dn block column-major,
    // i.e. sample by sample (dense/DenseMatrix.cpp:172-181); the generator persists across rounds
    // (HSSMatrix.compress_stable.hpp:108-112).
    if (r0 == 0 || !rng_) rng_.reset(new HostRng());
    std::minstd_rand* lin = &rng_->lin;
    std::mt19937* mer = &rng_->mer;
    auto& nd = rng_->nd;
    auto& ud = rng_->ud;
    std::vector<double> buf((size_t)dn * N);
    for (int s = 0; s < dn; s++)
      for (long long c = 0; c < N; c++) {
        double v;
        if (o_.random_engine == 0) v = o_.random_dist == 0 ? nd(*lin) : ud(*lin);
        else v = o_.random_dist == 0 ? nd(*mer) : ud(*mer);
        buf[s + (size_t)c * dn] = v;
      }
    ck(hssk_memcpy2d_h2d(ctx_, Rt_ + r0, sizeof(double) * dcap_, buf.data(), sizeof(double) * dn, sizeof(double) * dn, N));
  }
  ck(hssk_sync(ctx_));
  stats_.t_random += now() - t0;
}

bool DeviceHSS::compress_attempt(Source& src, int dcap) {
  reset_compression();
  attempt_++;
  dcap_ = dcap;
  const size_t N = n_;
  // sample arrays; with several GPUs the column count is padded to world * cols_per_rank so that
  // every rank's shard is one contiguous, equally sized block (in-place all-gather)
  cols_per_rank_ = o_.world > 1 ? ((long long)N + o_.world - 1) / o_.world : (long long)N;
  const size_t Npad = o_.world > 1 ? (size_t)cols_per_rank_ * o_.world : N;
  Rt_ = work_->dbl((size_t)dcap * N);
  Srt_ = work_->dbl((size_t)dcap * Npad);
  Sct_ = work_->dbl((size_t)dcap * Npad);
  if (o_.algorithm == 2) { Srt0_ = work_->dbl((size_t)dcap * Npad); Sct0_ = work_->dbl((size_t)dcap * Npad); }
  stats_.rounds = 0;
  stats_.f_sketch = stats_.f_local = stats_.f_reduce = stats_.f_id = stats_.f_ortho = 0;
  const bool original = (o_.algorithm != 1);
  if (!original) {
    // compress_stable(Amult, Aelem, opts), HSSMatrix.compress_stable.hpp:100-163
    int d = o_.d0, dd = o_.dd;
    while (!is_compressed()) {
      int c = (d == o_.d0) ? 0 : d;
      int dnew = (d == o_.d0) ? d + dd : dd;
      if (c + dnew > dcap) return false;
      fill_random(c, dnew);
      double t0 = now();
      src.sample(*this, c, dnew);
      ck(hssk_sync(ctx_));
      stats_.t_sketch += now() - t0;
      stats_.f_sketch += 4.0 * (double)N * (double)N * (sj_pat_ ? sj_nnz_ : dnew);   // SJLT: 2 nnz flops per element and product
      if (o_.verbose) std::cout << "# compressing with d+dd = " << d << "+" << dd << " (stable)" << std::endl;
      stats_.rounds++;
      for (auto& ids : own_by_height_) process_level(src, ids, d, dd, false);
      if (dist_subtree_) {
        exchange_cut_compress(d + dd);
        for (auto& ids : top_by_height_) process_level(src, ids, d, dd, false);
      }
      stats_.d_final = d + dd;
      if (!is_compressed()) {
        d += dd;
        dd = std::min(dd, o_.max_rank - d);
        if (dd <= 0) break;  // cannot add samples: compression failed (is_compressed() stays false)
      }
    }
  } else {
    // compress_original, HSSMatrix.compress.hpp:100-165
    int d_old = 0, d = o_.d0 + o_.p;
    while (!is_compressed()) {
      if (d > dcap) return false;
      fill_random(d_old, d - d_old);
      double t0 = now();
      src.sample(*this, d_old, d - d_old);
      ck(hssk_sync(ctx_));
      stats_.t_sketch += now() - t0;
      stats_.f_sketch += 4.0 * (double)N * (double)N * (sj_pat_ ? sj_nnz_ : d - d_old);
      if (o_.verbose) std::cout << "# compressing with d = " << d - o_.p << " + " << o_.p << (o_.algorithm == 2 ? " (original, hard restart)" : " (original)") << std::endl;
      if (o_.algorithm == 2) {   // keep the new samples as drawn
        if (dist_subtree_ || o_.world > 1) throw std::invalid_argument("hard restart is a single-GPU option");
        std::vector<hssk_colgather_desc> cp;
        cp.push_back(hssk_colgather_desc{Srt_ + d_old, Srt0_ + d_old, nullptr, d - d_old, (int)N, dcap_, dcap_, 0});
        cp.push_back(hssk_colgather_desc{Sct_ + d_old, Sct0_ + d_old, nullptr, d - d_old, (int)N, dcap_, dcap_, 0});
        ck(hssk_gather_cols(ctx_, cp.data(), 2));
      }
      stats_.rounds++;
      for (auto& ids : own_by_height_) process_level(src, ids, d, d - d_old, true);
      if (dist_subtree_) {
        exchange_cut_compress(d);
        for (auto& ids : top_by_height_) process_level(src, ids, d, d - d_old, true);
      }
      stats_.d_final = d;
      if (!is_compressed()) {
        d_old = d;
        d = 2 * (d_old - o_.p) + o_.p;
        if (d_old >= 4 * n_ + o_.p + 64) break;
        if (o_.algorithm == 2 && d <= dcap) restart_nodes(d_old);
      }
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// one tree height of one compression round
//   stable:   d, dd as in compress_recursive_stable (samples [0,d+dd), new ones [d,d+dd))
//   original: d = total samples, dd = newly added ones
// ---------------------------------------------------------------------------------------------
void DeviceHSS::process_level(Source& src, const std::vector<int>& ids_all, int d, int dd, bool original) {
  const int dtot = original ? d : d + dd;
  const int dnew0 = original ? d - dd : d;  // first new sample row
  std::vector<int> ids;
  for (int id : ids_all) {
    Node& nd = nodes_[id];
    if (!nd.leaf() && !(nodes_[nd.c0].compressed() && nodes_[nd.c1].compressed())) continue;
    if (nd.lvl == 0 && nd.compressed()) continue;
    ids.push_back(id);
  }
  if (ids.empty()) return;
  // --- extraction of D / B01 / B10 for untouched nodes (compress_stable.hpp:171-182, 204-217)
  std::vector<int> fresh;
  std::vector<char> was_untouched(nodes_.size(), 0), was_compressed(nodes_.size(), 0);
  for (int id : ids) {
    Node& nd = nodes_[id];
    was_untouched[id] = nd.untouched();
    was_compressed[id] = nd.compressed();
    if (nd.untouched()) fresh.push_back(id);
  }
  extract_blocks(src, fresh);
  std::vector<int> work_ids, r0s, dns;
  for (int id : ids) {
    Node& nd = nodes_[id];
    if (nd.lvl == 0) { nd.Ustate = nd.Vstate = 2; continue; }
    if (!nd.panels) {
      if (nd.leaf()) {
        nd.mU = nd.mV = nd.m;
        nd.Srt = Srt_ + (size_t)nd.lo * dcap_;
        nd.Sct = Sct_ + (size_t)nd.lo * dcap_;
        nd.Rrt = nd.Rct = Rt_ + (size_t)nd.lo * dcap_;
      } else {
        nd.mU = nodes_[nd.c0].rU + nodes_[nd.c1].rU;
        nd.mV = nodes_[nd.c0].rV + nodes_[nd.c1].rV;
        nd.Srt = work_->dbl((size_t)dcap_ * std::max(nd.mU, 1));
        nd.Sct = work_->dbl((size_t)dcap_ * std::max(nd.mV, 1));
        nd.Rrt = work_->dbl((size_t)dcap_ * std::max(nd.mV, 1));
        nd.Rct = work_->dbl((size_t)dcap_ * std::max(nd.mU, 1));
      }
      nd.panels = true;
    }
    work_ids.push_back(id);
    r0s.push_back(was_untouched[id] ? 0 : dnew0);
    dns.push_back(was_untouched[id] ? dtot : dtot - dnew0);
  }
  if (work_ids.empty()) return;
  local_samples(work_ids, r0s, dns);

  // --- bases
  std::vector<int> id_nodes, id_which, ot_nodes, ot_which;
  for (int id : work_ids) {
    Node& nd = nodes_[id];
    if (was_compressed[id]) continue;
    for (int w = 0; w < 2; w++) {
      int st = w == 0 ? nd.Ustate : nd.Vstate;
      if (st == 2) continue;
      int rows = w == 0 ? nd.mU : nd.mV;
      if (original || dtot >= o_.max_rank || dtot >= rows) { id_nodes.push_back(id); id_which.push_back(w); }
      else { ot_nodes.push_back(id); ot_which.push_back(w); }
    }
  }
  if (!ot_nodes.empty()) {
    std::vector<char> resolved;
    ortho_test(ot_nodes, ot_which, d, dd, resolved);
    for (size_t i = 0; i < ot_nodes.size(); i++) {
      if (resolved[i]) { id_nodes.push_back(ot_nodes[i]); id_which.push_back(ot_which[i]); }
      else {
        Node& nd = nodes_[ot_nodes[i]];
        (ot_which[i] == 0 ? nd.Ustate : nd.Vstate) = 1;
      }
    }
  }
  // (the host-side bookkeeping of the ID -- index sets, permutations -- is finished behind the launches of the sample
  // reduction below, which only need what is on the device; the ORIGINAL algorithm inspects and resets ranks first)
  defer_book_ = !original;
  run_id(id_nodes, id_which, dtot);
  defer_book_ = false;
  if (original) {
    // compute_U_V_bases acceptance, HSSMatrix.compress.hpp:663-686
    for (int id : work_ids) {
      Node& nd = nodes_[id];
      if (was_compressed[id]) continue;
      bool ok = (dtot - o_.p >= o_.max_rank) || (nd.rU < dtot - o_.p && nd.rV < dtot - o_.p);
      if (!ok) { nd.Ustate = nd.Vstate = 1; nd.rU = nd.rV = 0; nd.Ir.clear(); nd.Ic.clear(); }
    }
  }
  // --- reduce (reduce_local_samples, HSSMatrix.compress.hpp:689-724)
  std::vector<int> rd_ids, rd_r0, rd_dn;
  for (int id : work_ids) {
    Node& nd = nodes_[id];
    if (!nd.compressed()) continue;
    if (!was_compressed[id]) {
      nd.RrtRed = work_->dbl((size_t)dcap_ * std::max(nd.rV, 1));
      nd.RctRed = work_->dbl((size_t)dcap_ * std::max(nd.rU, 1));
      rd_ids.push_back(id); rd_r0.push_back(0); rd_dn.push_back(dtot);
    } else {
      rd_ids.push_back(id); rd_r0.push_back(dnew0); rd_dn.push_back(dtot - dnew0);
    }
  }
  reduce_samples(rd_ids, rd_r0, rd_dn);
  finish_id_bookkeeping();
}

void DeviceHSS::extract_blocks(Source& src, const std::vector<int>& ids) {
  std::vector<ElemReq> reqs;
  for (int id : ids) {
    Node& nd = nodes_[id];
    if (nd.leaf()) {
      nd.D = persist_->dbl((size_t)nd.m * nd.m);
      reqs.push_back(ElemReq{nullptr, nullptr, nullptr, nullptr, nd.lo, nd.lo, nd.m, nd.m, nd.D, nd.m});
    } else {
      Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
      nd.B01 = persist_->dbl((size_t)std::max(a.rU, 1) * std::max(b.rV, 1));
      nd.B10 = persist_->dbl((size_t)std::max(b.rU, 1) * std::max(a.rV, 1));
      reqs.push_back(ElemReq{a.dIr, b.dIc, &a.Ir, &b.Ic, 0, 0, a.rU, b.rV, nd.B01, std::max(a.rU, 1)});
      reqs.push_back(ElemReq{b.dIr, a.dIc, &b.Ir, &a.Ic, 0, 0, b.rU, a.rV, nd.B10, std::max(b.rU, 1)});
    }
  }
  if (!reqs.empty()) src.extract(*this, reqs);
}

// compute_local_samples (HSSMatrix.compress.hpp:524-629) on sample rows [r0, r0+dn) of each node
void DeviceHSS::local_samples(const std::vector<int>& ids, const std::vector<int>& r0s, const std::vector<int>& dns) {
  std::vector<hssk_combine_desc> cb;
  std::vector<hssk_gemm_desc> mm;
  std::vector<hssk_leaf_update_desc> lu;   // fused Sr / Sc update of the leaves (both share the R panel)
  for (size_t k = 0; k < ids.size(); k++) {
    Node& nd = nodes_[ids[k]];
    const int r0 = r0s[k], dn = dns[k];
    if (dn <= 0) continue;
    if (nd.leaf()) {
      const int m = nd.m;
      // Sr_loc -= D Rr_loc  ->  Srt -= Rt D^T ;  Sc_loc -= D^T Rc_loc  ->  Sct -= Rt D
      if (dn <= 192 && dn % 2 == 0 && r0 % 2 == 0 && nd.Rrt == nd.Rct)
        lu.push_back(hssk_leaf_update_desc{nd.Rrt + r0, nd.D, nd.Srt + r0, nd.Sct + r0, dn, m, dcap_, m, dcap_});
      else {
        mm.push_back(hssk_gemm_desc{nd.Rrt + r0, nd.D, nd.Srt + r0, dn, m, m, dcap_, m, dcap_, 0, 1, -1.0, 1.0});
        mm.push_back(hssk_gemm_desc{nd.Rct + r0, nd.D, nd.Sct + r0, dn, m, m, dcap_, m, dcap_, 0, 0, -1.0, 1.0});
      }
      stats_.f_local += 4.0 * m * (double)m * dn;
    } else {
      Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
      // the children's skeleton rows (extract_rows, compress.hpp:563-566, 611-614) minus the coupling terms, one fused
      // gather + product per block:  Sr0 = Sr_a(Jr_a) - B01 Rr1 ; Sr1 = Sr_b(Jr_b) - B10 Rr0 ;
      //                              Sc0 = Sc_a(Jc_a) - B10^T Rc1 ; Sc1 = Sc_b(Jc_b) - B01^T Rc0       (all transposed)
      const int none = 0x7fffffff;
      const int l01 = std::max(a.rU, 1), l10 = std::max(b.rU, 1);
      cb.push_back(hssk_combine_desc{a.Srt + r0, nullptr, dcap_, none, a.permU, b.RrtRed + r0, nullptr, dcap_, none, nullptr,
                                     nd.B01, 1, l01, -1.0, nd.Srt + r0, dcap_, dn, a.rU, b.rV});
      cb.push_back(hssk_combine_desc{b.Srt + r0, nullptr, dcap_, none, b.permU, a.RrtRed + r0, nullptr, dcap_, none, nullptr,
                                     nd.B10, 1, l10, -1.0, nd.Srt + r0 + (size_t)a.rU * dcap_, dcap_, dn, b.rU, a.rV});
      cb.push_back(hssk_combine_desc{a.Sct + r0, nullptr, dcap_, none, a.permV, b.RctRed + r0, nullptr, dcap_, none, nullptr,
                                     nd.B10, l10, 1, -1.0, nd.Sct + r0, dcap_, dn, a.rV, b.rU});
      cb.push_back(hssk_combine_desc{b.Sct + r0, nullptr, dcap_, none, b.permV, a.RctRed + r0, nullptr, dcap_, none, nullptr,
                                     nd.B01, l01, 1, -1.0, nd.Sct + r0 + (size_t)a.rV * dcap_, dcap_, dn, b.rV, a.rU});
      stats_.f_local += 4.0 * ((double)a.rU * b.rV + (double)b.rU * a.rV) * dn;
    }
  }
  if (!cb.empty()) ck(hssk_gather_combine(ctx_, cb.data(), (int)cb.size()));
  if (!lu.empty()) {
    int rc = hssk_leaf_update_vbatched(ctx_, lu.data(), (int)lu.size());
    if (rc == 2) {  // layout not eligible for the fused kernel: two plain GEMMs per leaf
      for (auto& u : lu) {
        mm.push_back(hssk_gemm_desc{u.R, u.D, u.Sr, u.d, u.m, u.m, u.ldr, u.ldd, u.lds, 0, 1, -1.0, 1.0});
        mm.push_back(hssk_gemm_desc{u.R, u.D, u.Sc, u.d, u.m, u.m, u.ldr, u.ldd, u.lds, 0, 0, -1.0, 1.0});
      }
    } else ck(rc);
  }
  if (!mm.empty()) ck(hssk_gemm_vbatched(ctx_, mm.data(), (int)mm.size()));
}

// reduce_local_samples: Rr_loc <- V^H Rr_loc, Rc_loc <- U^H Rc_loc (HSSBasisID::applyC), transposed
void DeviceHSS::reduce_samples(const std::vector<int>& ids, const std::vector<int>& r0s, const std::vector<int>& dns) {
  if (ids.empty()) return;
  // Rr_red = Rr(Jc, :) + XV Rr(rest, :) (transposed: columns of Rrt), where Rr of an inner node is the stack of its
  // children's reduced samples -- read in place from the two children ([a | b] with the split at a's rank), one fused
  // gather + product per (node, side)
  std::vector<hssk_combine_desc> cb;
  const int none = 0x7fffffff;
  for (size_t k = 0; k < ids.size(); k++) {
    Node& nd = nodes_[ids[k]];
    const int r0 = r0s[k], dn = dns[k];
    if (dn <= 0) continue;
    const double *rr0, *rr1 = nullptr, *rc0, *rc1 = nullptr;
    int sr = none, sc = none;
    if (nd.leaf()) { rr0 = nd.Rrt + r0; rc0 = nd.Rct + r0; }
    else {
      Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
      rr0 = a.RrtRed + r0; rr1 = b.RrtRed + r0; sr = a.rV;
      rc0 = a.RctRed + r0; rc1 = b.RctRed + r0; sc = a.rU;
    }
    {
      const int m = nd.mV, r = nd.rV, K = (m > r && r > 0) ? m - r : 0;
      if (r > 0)
        cb.push_back(hssk_combine_desc{rr0, rr1, dcap_, sr, nd.permV, rr0, rr1, dcap_, sr, nd.permV + r, nd.XV, 1, r, 1.0,
                                       nd.RrtRed + r0, dcap_, dn, r, K});
      stats_.f_reduce += 2.0 * r * (double)K * dn;
    }
    {
      const int m = nd.mU, r = nd.rU, K = (m > r && r > 0) ? m - r : 0;
      if (r > 0)
        cb.push_back(hssk_combine_desc{rc0, rc1, dcap_, sc, nd.permU, rc0, rc1, dcap_, sc, nd.permU + r, nd.XU, 1, r, 1.0,
                                       nd.RctRed + r0, dcap_, dn, r, K});
      stats_.f_reduce += 2.0 * r * (double)K * dn;
    }
  }
  if (!cb.empty()) ck(hssk_gather_combine(ctx_, cb.data(), (int)cb.size()));
}

// ID of the listed (node, basis) pairs on all dtot samples; commits ranks, X, perm, index sets
void DeviceHSS::run_id(const std::vector<int>& ids, const std::vector<int>& which, int dtot) {
  if (ids.empty()) return;
  Arena& tmp = *tmp_;
  tmp.rewind();
  const size_t cnt = ids.size();
  std::vector<double*> Ws(cnt, nullptr);
  std::vector<const double*> srcs(cnt, nullptr);
  std::vector<int> ds(cnt, dtot);
  for (size_t k = 0; k < cnt; k++) {
    Node& nd = nodes_[ids[k]];
    const int m = which[k] == 0 ? nd.mU : nd.mV;
    const double* S = which[k] == 0 ? nd.Srt : nd.Sct;
    if (m == 0) continue;
    Ws[k] = tmp.dbl((size_t)dtot * m);
    srcs[k] = S;
  }
  // (the samples stay where they are: the ID reads them in place and writes its factors to the panel in tmp_)
  id_panels(ids, which, Ws, ds, &srcs, dcap_);
}

// Tall panels (d >> m, the kernel-matrix path: d = thousands of sampled columns): the pivoted QR of W (d x m) only
// depends on R of W = Q R, so W is first reduced to its m x m triangular factor by an unpivoted Householder TSQR
// -- row chunks factored independently (register-resident / blocked batched QR), the R factors stacked pairwise
// and re-factored until one is left -- and the ID then runs on that small panel in the register kernels.  Same
// pivots, ranks and X = R11^{-1} R12 as the direct QRCP up to rounding; backward stable (all Householder).
static bool tsqr_staircase() {   // (read per call: the tests compare both paths in one process)
  const char* e = std::getenv("STRUMPACK_AMD_TSQR_DENSE");
  return !(e && std::atoi(e));
}

void DeviceHSS::tsqr_reduce(const std::vector<int>& ids, const std::vector<int>& which, std::vector<double*>& Ws,
                            std::vector<int>& ds) {
  Arena& tmp = *tmp_;
  const size_t cnt = ids.size();
  struct Piece { double* p; int ld, rows; };
  std::vector<std::vector<Piece>> pieces(cnt);
  std::vector<int> ms(cnt, 0);
  bool any = false;
  std::vector<hssk_qr_desc> qr;
  for (size_t k = 0; k < cnt; k++) {
    const Node& nd = nodes_[ids[k]];
    const int m = which[k] == 0 ? nd.mU : nd.mV, d = ds[k];
    ms[k] = m;
    if (m <= 0 || !Ws[k] || d <= std::max(256, 2 * m)) continue;
    // chunk rows: register QR (<= 256 rows x 192 columns, <= 208 rows x 208 columns: the 16-lanes-per-column kernels of
    // hssk_qr.hip), the 512-row blocked path, or the tall blocked path for wide panels
    const int chunk = m <= 192 ? 256 : (m <= 208 ? 208 : (m <= 256 ? 512 : 2 * m));
    for (int r0 = 0; r0 < d; r0 += chunk) {
      const int cr = std::min(chunk, d - r0);
      double* wk = tmp.dbl((size_t)cr + m);
      qr.push_back(hssk_qr_desc{Ws[k] + r0, d, cr, m, nullptr, 0, 0, nullptr, wk, 0, 0., 0., 1});   // (only R is read again)
      pieces[k].push_back(Piece{Ws[k] + r0, d, std::min(cr, m)});
    }
    any = true;
  }
  if (!any) return;
  ck(hssk_qr_vbatched(ctx_, qr.data(), (int)qr.size()));
  // pairs of full triangles are merged by hssk_tpqr_vbatched: in place over the first one, no stacking, one launch per tree
  // level (STRUMPACK_AMD_TSQR_PAIRS=0 or STRUMPACK_AMD_TSQR_DENSE=1: the stacked blocked QR below for everything)
  static const bool pairs_off = [] { const char* e = std::getenv("STRUMPACK_AMD_TSQR_PAIRS"); return e && e[0] == '0'; }();
  for (;;) {
    std::vector<hssk_triu_desc> cp;
    std::vector<hssk_tpqr_desc> tp;
    qr.clear();
    bool more = false;
    for (size_t k = 0; k < cnt; k++) {
      std::vector<Piece>& pc = pieces[k];
      if (pc.size() <= 1) continue;
      const int m = ms[k];
      // fan-in: as many triangles as fit the register QR (256 rows) for narrow panels; wider ones either pairwise
      // (390-row blocked QR per tree level) or all at once through the tall blocked path (STRUMPACK_AMD_TSQR_FANIN)
      static const int fan_env = std::getenv("STRUMPACK_AMD_TSQR_FANIN") ? std::atoi(std::getenv("STRUMPACK_AMD_TSQR_FANIN")) : 0;
      const size_t fan_wide = fan_env >= 2 ? (size_t)fan_env : 2;
      const size_t fan = m <= 128 ? std::max<size_t>(2, 256 / std::max(m, 1)) : fan_wide;
      std::vector<Piece> next;
      for (size_t i = 0; i < pc.size(); i += fan) {
        const size_t cntp = std::min(fan, pc.size() - i);
        if (cntp == 1) { next.push_back(pc[i]); continue; }
        int rows = 0;
        bool full = true;   // every piece a full m x m triangle
        for (size_t t = 0; t < cntp; t++) { rows += pc[i + t].rows; full = full && pc[i + t].rows == m; }
        if (cntp == 2 && full && m <= 224 && !pairs_off && tsqr_staircase()) {
          tp.push_back(hssk_tpqr_desc{pc[i].p, pc[i].ld, pc[i + 1].p, pc[i + 1].ld, m});
          next.push_back(pc[i]);
          continue;
        }
        double* dst = tmp.dbl((size_t)rows * m);
        // full triangles are stacked with their rows interleaved (row r of piece t -> row cntp r + t): column j of the
        // stack is then zero from row cntp (j + 1) on, and the blocked QR only sweeps that staircase (hssk_qr_desc::stair)
        const bool stair = full && rows > 256 && tsqr_staircase();
        int r0 = 0;
        for (size_t t = 0; t < cntp; t++) {
          if (stair) cp.push_back(hssk_triu_desc{pc[i + t].p, dst + t, pc[i + t].rows, m, pc[i + t].ld, rows, (int)cntp});
          else cp.push_back(hssk_triu_desc{pc[i + t].p, dst + r0, pc[i + t].rows, m, pc[i + t].ld, rows, 1});
          r0 += pc[i + t].rows;
        }
        double* wk = tmp.dbl((size_t)rows + m);
        qr.push_back(hssk_qr_desc{dst, rows, rows, m, nullptr, 0, 0, nullptr, wk, stair ? (int)cntp : 0, 0., 0., 1});
        next.push_back(Piece{dst, rows, std::min(rows, m)});
      }
      pc.swap(next);
      more = more || pc.size() > 1;
    }
    if (cp.empty() && tp.empty()) break;
    if (!tp.empty()) ck(hssk_tpqr_vbatched(ctx_, tp.data(), (int)tp.size()));
    if (!cp.empty()) {
      ck(hssk_copy_triu(ctx_, cp.data(), (int)cp.size()));
      ck(hssk_qr_vbatched(ctx_, qr.data(), (int)qr.size()));
    }
    if (!more) break;
  }
  // clean m x m (or shorter) triangular panels for the ID
  std::vector<hssk_triu_desc> fin;
  for (size_t k = 0; k < cnt; k++) {
    if (pieces[k].empty()) continue;
    const Piece& pc = pieces[k][0];
    double* R = tmp.dbl((size_t)pc.rows * ms[k]);
    fin.push_back(hssk_triu_desc{pc.p, R, pc.rows, ms[k], pc.ld, pc.rows});
    Ws[k] = R;
    ds[k] = pc.rows;
  }
  if (!fin.empty()) ck(hssk_copy_triu(ctx_, fin.data(), (int)fin.size()));
}

// Row ID of the listed (node, basis) pairs from prepared panels W_k = S_k^T (ds[k] x m_k, contiguous, in tmp_):
// truncated QRCP + X = R11^{-1} R12 on the device, then the commit of rank, permutation, skeleton indices.
void DeviceHSS::id_panels(const std::vector<int>& ids, const std::vector<int>& which, const std::vector<double*>& Ws_in,
                          const std::vector<int>& ds_in, const std::vector<const double*>* srcs, int ldsrc) {
  Arena& tmp = *tmp_;
  const size_t cnt = ids.size();
  std::vector<double*> Ws(Ws_in);
  std::vector<int> ds(ds_in);
  if (srcs) {
    // panels that take the TSQR pre-reduction (more than 256 sample rows) are reduced in place: those need their copy
    std::vector<hssk_colgather_desc> cp;
    bool tall = false;
    for (size_t k = 0; k < cnt; k++) tall = tall || ds[k] > 256;
    if (tall) {
      for (size_t k = 0; k < cnt; k++) {
        const int m = which[k] == 0 ? nodes_[ids[k]].mU : nodes_[ids[k]].mV;
        if (Ws[k] && m) cp.push_back(hssk_colgather_desc{(*srcs)[k], Ws[k], nullptr, ds[k], m, ldsrc, ds[k], 0});
      }
      if (!cp.empty()) ck(hssk_gather_cols(ctx_, cp.data(), (int)cp.size()));
      srcs = nullptr;
    }
  }
  tsqr_reduce(ids, which, Ws, ds);
  std::vector<hssk_id_desc> idd;
  int id_dmax = 0, id_mmax = 0;
  std::vector<int*> perms(cnt, nullptr);
  size_t perm_total = 0;
  for (size_t k = 0; k < cnt; k++) perm_total += (which[k] == 0 ? nodes_[ids[k]].mU : nodes_[ids[k]].mV);
  // ranks and permutations of the level in ONE device block: one read-back (one host synchronisation) per level
  int* rank_block = persist_->ints(cnt + std::max<size_t>(perm_total, 1));
  int* perm_block = rank_block + cnt;
  size_t poff = 0;
  for (size_t k = 0; k < cnt; k++) {
    Node& nd = nodes_[ids[k]];
    const int m = which[k] == 0 ? nd.mU : nd.mV;
    perms[k] = perm_block + poff;
    poff += m;
    if (m == 0) continue;
    double* wk = tmp.dbl(3 * (size_t)m);
    // (defer_x: X = R11^{-1} R12 is computed behind the read-back of the ranks, straight into its final place)
    idd.push_back(hssk_id_desc{Ws[k], ds[k], ds[k], m, o_.rel_tol / nd.lvl, o_.abs_tol / nd.lvl, o_.max_rank, perms[k], rank_block + k, wk,
                               srcs ? (*srcs)[k] : nullptr, ldsrc, 1});
    id_dmax = std::max(id_dmax, ds[k]);
    id_mmax = std::max(id_mmax, m);
  }
  if (!idd.empty()) ck(hssk_id_vbatched(ctx_, idd.data(), (int)idd.size()));
  const int x_solved = hssk_id_solves_inline(id_dmax, id_mmax);
  std::vector<int> hall(cnt + std::max<size_t>(perm_total, 1));
  ck(hssk_memcpy_d2h(ctx_, hall.data(), rank_block, (long long)sizeof(int) * (cnt + perm_total)));
  const int* hranks = hall.data();
  const int* hperm = hall.data() + cnt;
  // commit, in the order that puts the device back to work first: (A) ranks -> final places of X -> the X solves are
  // launched; (B) the host-side bookkeeping (permutations, global skeleton indices: vectors per node) while they run;
  // then the one index upload of the level
  std::vector<hssk_xsolve_desc> xc;
  std::vector<size_t> idx_off(cnt), perm_off(cnt);
  size_t idx_total = 0;
  poff = 0;
  for (size_t k = 0; k < cnt; k++) {
    Node& nd = nodes_[ids[k]];
    const int w = which[k];
    const int m = w == 0 ? nd.mU : nd.mV;
    const int dtot = ds[k];
    const int r = m ? hranks[k] : 0;
    perm_off[k] = poff;
    poff += m;
    idx_off[k] = idx_total;
    idx_total += r;
    double* X = persist_->dbl((size_t)std::max(r, 1) * std::max(m - r, 1));
    if (r > 0 && m > r) xc.push_back(hssk_xsolve_desc{Ws[k], dtot, r, m, X, r, x_solved});
    if (w == 0) { nd.rU = r; nd.XU = X; nd.permU = perms[k]; nd.Ustate = 2; }
    else { nd.rV = r; nd.XV = X; nd.permV = perms[k]; nd.Vstate = 2; }
    stats_.f_id += 2.0 * (4.0 * m * (double)dtot * r - 2.0 * (m + dtot) * (double)r * r + 4.0 * r * (double)r * r / 3.0 + (double)r * r * (m - r));
  }
  if (!xc.empty()) ck(hssk_id_xsolve_vbatched(ctx_, xc.data(), (int)xc.size()));
  book_.ids = ids; book_.which = which; book_.hall = std::move(hall);
  book_.idx_off = std::move(idx_off); book_.perm_off = std::move(perm_off);
  book_.cnt = cnt; book_.idx_total = idx_total; book_.active = true;
  if (!defer_book_) finish_id_bookkeeping();
  // (no synchronisation: everything that reuses the W panels in tmp_ is enqueued behind these launches on the same stream)
}

// Second half of id_panels' commit: permutations and global skeleton indices of the level on the host (vectors per node),
// the one index upload of the level.  Nothing on the device waits for it except the next level's block extraction.
void DeviceHSS::finish_id_bookkeeping() {
  if (!book_.active) return;
  book_.active = false;
  const std::vector<int>&ids = book_.ids, &which = book_.which;
  const size_t cnt = book_.cnt, idx_total = book_.idx_total;
  const int* hperm = book_.hall.data() + cnt;
  std::vector<int> idx_host(std::max<size_t>(idx_total, 1));   // all skeleton index sets of this level: one upload
  // (on this thread: waking the host pool costs ~100 us, as much as the widest level's bookkeeping itself -- measured)
  for (size_t k = 0; k < cnt; k++) {
    Node& nd = nodes_[ids[k]];
    const int w = which[k];
    const int m = w == 0 ? nd.mU : nd.mV;
    const int r = w == 0 ? nd.rU : nd.rV;
    std::vector<int> perm(hperm + book_.perm_off[k], hperm + book_.perm_off[k] + m);
    // global skeleton indices (compress_stable.hpp:299-306, 334-341)
    std::vector<int> I(r);
    if (nd.leaf()) for (int i = 0; i < r; i++) I[i] = nd.lo + perm[i];
    else {
      const std::vector<int>& ia = w == 0 ? nodes_[nd.c0].Ir : nodes_[nd.c0].Ic;
      const std::vector<int>& ib = w == 0 ? nodes_[nd.c1].Ir : nodes_[nd.c1].Ic;
      const int r0 = (int)ia.size();
      for (int i = 0; i < r; i++) I[i] = perm[i] < r0 ? ia[perm[i]] : ib[perm[i] - r0];
    }
    std::copy(I.begin(), I.end(), idx_host.begin() + book_.idx_off[k]);
    if (w == 0) { nd.hpermU = std::move(perm); nd.Ir = std::move(I); }
    else { nd.hpermV = std::move(perm); nd.Ic = std::move(I); }
  }
  int* idx_dev = persist_->ints(std::max<size_t>(idx_total, 1));
  if (idx_total) ck(hssk_upload_async(ctx_, idx_dev, idx_host.data(), (long long)sizeof(int) * idx_total));
  for (size_t k = 0; k < cnt; k++) {
    Node& nd = nodes_[ids[k]];
    (which[k] == 0 ? nd.dIr : nd.dIc) = idx_dev + book_.idx_off[k];
  }
}

// ---------------------------------------------------------------------------------------------
// Kernel-matrix compression from point coordinates:
// HSSMatrix::compress_with_coordinates / compress_recursive_ann / compute_local_samples_ann /
// compute_U_V_bases_ann (HSS/HSSMatrix.compress_kernel.hpp:50-293), level-synchronous.
// No random sketch: the sample of a node is S = K(I, cols) with cols = the neighbours of the node's points
// that lie outside the node (leaf) resp. the union of the children's column sets outside the node (inner),
// I = the node's rows (leaf) resp. its children's skeleton rows; symmetric: V = U, B10 = B01^T.
// ---------------------------------------------------------------------------------------------
void DeviceHSS::compress_kernel(const KernelSpec& ks, const int* user_ann, int user_k) {
  OpGuard op_guard(op_mu_);
  double t0 = now();
  stats_ = PhaseStats();
  const int N = n_, dim = ks.d;
  if (dim <= 0 || !ks.X) throw std::invalid_argument("compress_kernel: no points");
  int k = std::min(N, std::max(1, user_ann ? user_k : ks.ann));
  for (;;) {
    reset_compression();
    stats_.rounds++;
    // points and neighbour lists
    double* dX = work_->dbl((size_t)dim * N);
    ck(hssk_memcpy_h2d(ctx_, dX, ks.X, (long long)sizeof(double) * dim * N));
    hssk_kernel_spec spec{dX, N, dim, ks.type, ks.p, ks.h, ks.lambda};
    double tk0 = now();
    std::vector<int> ann((size_t)k * N);
    if (user_ann && k == user_k) std::copy(user_ann, user_ann + (size_t)k * N, ann.begin());
    else if (ks.neighbors) ks.neighbors(k, ann.data());
    else {
      // one process per GPU: neighbours of this rank's own points only (its subtree's leaves are all that read them)
      int q0 = 0, q1 = N;
      if (dist_subtree_) { const Node& c = nodes_[cut_nodes_[o_.rank]]; q0 = c.lo; q1 = c.lo + c.m; }
      int* dann = work_->ints((size_t)k * N);
      ck(hssk_knn(ctx_, dX, dim, N, k, q0, q1, dann));
      ck(hssk_memcpy_d2h(ctx_, ann.data() + (size_t)k * q0, dann + (size_t)k * q0, (long long)sizeof(int) * k * (q1 - q0)));
    }
    stats_.t_random += now() - tk0;   // neighbour search (reported in the 'random' slot: it replaces the random sketch)
    std::vector<std::vector<int>> cols(nodes_.size());   // per node: sorted unique column ids outside the node
    bool failed = false;
    auto do_level = [&](const std::vector<int>& ids) {
      if (ids.empty() || failed) return;
      tmp_->rewind();
      double tl0 = now();
      // ---- column sets and row sets (host), one index upload per level
      std::vector<int> hidx;
      std::vector<size_t> roff(ids.size()), coff(ids.size());
      std::vector<std::vector<int>> rows(ids.size());
      // the nodes of a level are independent: host threads build their row / column sets side by side
      host_parallel_for(ids.size(), [&](size_t q) {
        Node& nd = nodes_[ids[q]];
        std::vector<int>& cs = cols[ids[q]];
        const int lo = nd.lo, hi = nd.lo + nd.m;
        if (nd.leaf()) {
          nd.mU = nd.mV = nd.m;
          rows[q].resize(nd.m);
          for (int i = 0; i < nd.m; i++) rows[q][i] = lo + i;
          if (nd.lvl > 0) {
            // sorted, duplicate-free ids outside the node: marked in a bitmap over the point set and read back in order
            // (m k ~ 1e4 ids per leaf: cheaper than sorting them)
            std::vector<unsigned long long> bits(((size_t)N + 63) / 64, 0ULL);
            size_t marked = 0;
            for (int i = lo; i < hi; i++)
              for (int j = 0; j < k; j++) {
                const int g = ann[(size_t)i * k + j];
                if (g >= 0 && (g < lo || g >= hi)) {
                  unsigned long long& wd = bits[(size_t)g >> 6];
                  const unsigned long long b = 1ULL << (g & 63);
                  marked += !(wd & b);
                  wd |= b;
                }
              }
            cs.reserve(marked);
            for (size_t wi = 0; wi < bits.size(); wi++) {
              unsigned long long wd = bits[wi];
              while (wd) {
                cs.push_back((int)(wi * 64 + (size_t)__builtin_ctzll(wd)));
                wd &= wd - 1;
              }
            }
          }
        } else {
          Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
          nd.mU = nd.mV = a.rU + b.rU;
          rows[q] = a.Ir;
          rows[q].insert(rows[q].end(), b.Ir.begin(), b.Ir.end());
          if (nd.lvl > 0) {
            // union of the children's (sorted, duplicate-free) sets without the ids inside this node
            const std::vector<int>&ca = cols[nd.c0], &cb = cols[nd.c1];
            cs.reserve(ca.size() + cb.size());
            size_t i = 0, j = 0;
            auto keep = [&](int g) { if (g < lo || g >= hi) cs.push_back(g); };
            while (i < ca.size() && j < cb.size()) {
              if (ca[i] < cb[j]) keep(ca[i++]);
              else if (cb[j] < ca[i]) keep(cb[j++]);
              else { keep(ca[i]); i++; j++; }
            }
            while (i < ca.size()) keep(ca[i++]);
            while (j < cb.size()) keep(cb[j++]);
          }
        }
      });
      for (size_t q = 0; q < ids.size(); q++) {
        const std::vector<int>& cs = cols[ids[q]];
        roff[q] = hidx.size(); hidx.insert(hidx.end(), rows[q].begin(), rows[q].end());
        coff[q] = hidx.size(); hidx.insert(hidx.end(), cs.begin(), cs.end());
      }
      // the children's column sets are not needed above this level
        for (int id : ids) {
          if (nodes_[id].leaf()) continue;
          std::vector<int>().swap(cols[nodes_[id].c0]);
          std::vector<int>().swap(cols[nodes_[id].c1]);
        }
      int* didx = tmp_->ints(std::max<size_t>(hidx.size(), 1));
      if (!hidx.empty()) ck(hssk_memcpy_h2d(ctx_, didx, hidx.data(), (long long)sizeof(int) * hidx.size()));
      stats_.t_sketch += now() - tl0;   // host column-set construction (the 'sketch' slot of this path)
      // ---- D (leaves), B01 / B10 (inner nodes), sample panels W = K(cols, rows)  [= S^T]
      std::vector<hssk_keval_desc> ev;
      std::vector<hssk_transpose_desc> tr;
      std::vector<int> idn, which;
      std::vector<double*> Ws;
      std::vector<int> ds;
      for (size_t q = 0; q < ids.size(); q++) {
        Node& nd = nodes_[ids[q]];
        if (nd.leaf()) {
          nd.D = persist_->dbl((size_t)nd.m * nd.m);
          ev.push_back(hssk_keval_desc{nullptr, nullptr, nd.D, nd.m, nd.m, nd.m, nd.lo, nd.lo});
        } else {
          Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
          nd.B01 = persist_->dbl((size_t)std::max(a.rU, 1) * std::max(b.rV, 1));
          nd.B10 = persist_->dbl((size_t)std::max(b.rU, 1) * std::max(a.rV, 1));
          if (a.rU > 0 && b.rV > 0) {
            ev.push_back(hssk_keval_desc{a.dIr, b.dIc, nd.B01, a.rU, b.rV, a.rU, 0, 0});
            tr.push_back(hssk_transpose_desc{nd.B01, nd.B10, a.rU, b.rV, a.rU, b.rU});
          }
        }
        if (nd.lvl == 0) { nd.Ustate = nd.Vstate = 2; continue; }
        const int m = nd.mU, d = (int)cols[ids[q]].size();
        idn.push_back(ids[q]); which.push_back(0); ds.push_back(d);
        double* W = (m > 0 && d > 0) ? tmp_->dbl((size_t)d * m) : nullptr;
        Ws.push_back(W);
        if (W) ev.push_back(hssk_keval_desc{didx + coff[q], didx + roff[q], W, d, m, d, 0, 0});
      }
      if (!ev.empty()) ck(hssk_kernel_eval_vbatched(ctx_, &spec, ev.data(), (int)ev.size()));
      if (!tr.empty()) ck(hssk_transpose(ctx_, tr.data(), (int)tr.size()));
      if (idn.empty()) return;
      // nodes with an empty column set (d == 0) get rank 0 through a 1 x m zero panel
      for (size_t q = 0; q < idn.size(); q++)
        if (!Ws[q] && nodes_[idn[q]].mU > 0) {
          Ws[q] = tmp_->dbl(nodes_[idn[q]].mU);
          ck(hssk_memset_zero(ctx_, Ws[q], (long long)sizeof(double) * nodes_[idn[q]].mU));
          ds[q] = 1;
        }
      id_panels(idn, which, Ws, ds);
      // symmetric: V = U; acceptance test of compute_U_V_bases_ann (:262-272)
      for (size_t q = 0; q < idn.size(); q++) {
        Node& nd = nodes_[idn[q]];
        nd.rV = nd.rU; nd.XV = nd.XU; nd.permV = nd.permU; nd.hpermV = nd.hpermU; nd.Ic = nd.Ir; nd.dIc = nd.dIr; nd.Vstate = nd.Ustate;
        const int d = (int)cols[idn[q]].size();
        if (!(d >= nd.m || d >= o_.max_rank || nd.rU + o_.p < d)) failed = true;
      }
    };
    for (auto& ids : own_by_height_) do_level(ids);
    if (dist_subtree_) {
      failed = exchange_cut_kernel(cols, failed);
      for (auto& ids : top_by_height_) do_level(ids);
    }
    ck(hssk_sync(ctx_));
    if (!failed) break;
    if (k >= N) throw std::runtime_error("compress_kernel: the ID did not reach the required accuracy with all points as neighbours");
    k = std::min(2 * k, N);   // compress_with_coordinates: ann_number doubles until the tree compresses (:75)
    if (o_.verbose) std::cout << "# HSS kernel compression: increasing the neighbour count to " << k << std::endl;
  }
  if (dist_subtree_) exchange_node_table();
  free_compress_workspace();
  comm_arena_->reset();
  stats_.d_final = k;
  stats_.t_compress = now() - t0;
  stats_.t_tree = stats_.t_compress - stats_.t_sketch - stats_.t_random;
  if (o_.verbose)
    std::cout << "# HSS kernel compression: neighbours " << stats_.t_random << " s, column sets " << stats_.t_sketch << " s, blocks + ID "
              << stats_.t_tree << " s" << std::endl;
}

// update_orthogonal_basis (HSSMatrix.compress_stable.hpp:390-442) for the listed (node, basis) pairs
void DeviceHSS::ortho_test(const std::vector<int>& ids, const std::vector<int>& which, int d, int dd,
                           std::vector<char>& resolved) {
  const size_t cnt = ids.size();
  resolved.assign(cnt, 0);
  Arena& tmp = *tmp_;
  tmp.rewind();
  std::vector<hssk_transpose_desc> tr;
  std::vector<hssk_colgather_desc> cp;
  std::vector<hssk_qr_desc> qr;
  double* rdiag = tmp.dbl(2 * cnt);
  std::vector<char> untouched(cnt);
  // The QR of the first d sample columns only has to deliver max / min |R_ii| (DenseMatrix::orthogonalize,
  // dense/DenseMatrix.cpp:721-744); its explicit Q is needed by the Gram-Schmidt step alone, i.e. for the nodes
  // the R-diagonal test leaves undecided -- it is formed for those (hssk_formq_vbatched) after the read-back.
  for (size_t k = 0; k < cnt; k++) {
    Node& nd = nodes_[ids[k]];
    const int w = which[k];
    const int m = w == 0 ? nd.mU : nd.mV;
    const double* S = w == 0 ? nd.Srt : nd.Sct;
    double* Q = w == 0 ? nd.Qr : nd.Qc;
    untouched[k] = (w == 0 ? nd.Ustate : nd.Vstate) == 0;
    int c2, n2;
    if (untouched[k]) { c2 = 0; n2 = std::min(d, m); }
    else { c2 = d - dd; n2 = std::min(dd, m - (d - dd)); }
    double* T = tmp.dbl((size_t)m * std::max(n2, 1));
    if (untouched[k]) tr.push_back(hssk_transpose_desc{S, T, n2, m, dcap_, m});
    else cp.push_back(hssk_colgather_desc{Q + (size_t)c2 * m, T, nullptr, m, n2, m, m, 0});
    double* wk = tmp.dbl((size_t)m + n2);
    // the R-diagonal test below only needs to know whether SOME |R_ii| falls under the tolerance: the factorisation may stop
    // at the first one that does (hssk_qr_desc.stop_rel; the 1 - 1e-12 keeps the device's product form on the safe side of the
    // host's quotient form).  A node the test leaves undecided has run the full factorisation, which formq then uses.
    {
      const double atol = o_.abs_tol / nd.lvl, rtol = o_.rel_tol / nd.lvl;
      const double r0 = w == 0 ? nd.Ur_max : nd.Vr_max;
      hssk_qr_desc q{T, m, m, n2, nullptr, m, 0, rdiag + 2 * k, wk, 0, 0., 0.};
      if (untouched[k]) { q.stop_rel = rtol * (1. - 1e-12); q.stop_abs = atol; }
      else q.stop_abs = std::max(atol, rtol * std::abs(r0) * (1. - 1e-12));
      qr.push_back(q);
    }
    stats_.f_ortho += 4.0 * m * (double)n2 * n2;
  }
  if (!cp.empty()) ck(hssk_gather_cols(ctx_, cp.data(), (int)cp.size()));
  if (!tr.empty()) ck(hssk_transpose(ctx_, tr.data(), (int)tr.size()));
  ck(hssk_qr_vbatched(ctx_, qr.data(), (int)qr.size()));
  std::vector<double> hr(2 * cnt);
  ck(hssk_memcpy_d2h(ctx_, hr.data(), rdiag, (long long)sizeof(double) * 2 * cnt));
  std::vector<size_t> pend;
  for (size_t k = 0; k < cnt; k++) {
    Node& nd = nodes_[ids[k]];
    const int w = which[k];
    double r_max = hr[2 * k], r_min = hr[2 * k + 1];
    double& r_max_0 = w == 0 ? nd.Ur_max : nd.Vr_max;
    if (untouched[k]) r_max_0 = r_max;
    const double atol = o_.abs_tol / nd.lvl, rtol = o_.rel_tol / nd.lvl;
    if (std::abs(r_min) < atol || std::abs(r_min / r_max_0) < rtol) resolved[k] = 1;
    else pend.push_back(k);
  }
  if (pend.empty()) return;
  // undecided nodes: Q12 block from the stored reflectors, and Q(:, d:d+dd) = S(:, d:d+dd)
  {
    std::vector<hssk_qr_desc> fq;
    tr.clear();
    for (size_t k : pend) {
      Node& nd = nodes_[ids[k]];
      const int w = which[k];
      const int m = w == 0 ? nd.mU : nd.mV;
      const double* S = w == 0 ? nd.Srt : nd.Sct;
      double*& Q = w == 0 ? nd.Qr : nd.Qc;
      if (!Q) Q = work_->dbl((size_t)m * dcap_);
      const int c2 = untouched[k] ? 0 : d - dd;
      hssk_qr_desc q = qr[k];
      q.Q = Q + (size_t)c2 * m; q.ldq = m; q.nq = q.cols; q.rdiag = nullptr;
      fq.push_back(q);
      tr.push_back(hssk_transpose_desc{S + d, Q + (size_t)d * m, dd, m, dcap_, m});
    }
    ck(hssk_formq_vbatched(ctx_, fq.data(), (int)fq.size()));
    ck(hssk_transpose(ctx_, tr.data(), (int)tr.size()));
  }
  // iterated classical Gram-Schmidt of the dd new columns against Q12, norms of the first p columns
  const int pc = std::min(dd, o_.p);
  double* nrm = tmp.dbl(2 * pend.size());
  std::vector<hssk_norm_desc> n0, n1;
  std::vector<hssk_gemm_desc> g1, g2;
  for (size_t i = 0; i < pend.size(); i++) {
    size_t k = pend[i];
    Node& nd = nodes_[ids[k]];
    const int w = which[k];
    const int m = w == 0 ? nd.mU : nd.mV;
    double* Q = w == 0 ? nd.Qr : nd.Qc;
    const int q12 = std::min(d, m);
    double* Q3 = Q + (size_t)d * m;
    double* P = tmp.dbl((size_t)q12 * dd);
    n0.push_back(hssk_norm_desc{Q3, m, pc, m, nrm + 2 * i});
    g1.push_back(hssk_gemm_desc{Q, Q3, P, q12, dd, m, m, m, q12, 1, 0, 1.0, 0.0});
    g2.push_back(hssk_gemm_desc{Q, P, Q3, m, dd, q12, m, q12, m, 0, 0, -1.0, 1.0});
    n1.push_back(hssk_norm_desc{Q3, m, pc, m, nrm + 2 * i + 1});
    stats_.f_ortho += 8.0 * m * (double)q12 * dd;
  }
  ck(hssk_sumsq_vbatched(ctx_, n0.data(), (int)n0.size()));
  for (int it = 0; it < 2; it++) {
    ck(hssk_gemm_vbatched(ctx_, g1.data(), (int)g1.size()));
    ck(hssk_gemm_vbatched(ctx_, g2.data(), (int)g2.size()));
  }
  ck(hssk_sumsq_vbatched(ctx_, n1.data(), (int)n1.size()));
  std::vector<double> hn(2 * pend.size());
  ck(hssk_memcpy_d2h(ctx_, hn.data(), nrm, (long long)sizeof(double) * hn.size()));
  for (size_t i = 0; i < pend.size(); i++) {
    size_t k = pend[i];
    Node& nd = nodes_[ids[k]];
    const double atol = o_.abs_tol / nd.lvl, rtol = o_.rel_tol / nd.lvl;
    double S3 = std::sqrt(hn[2 * i]), Q3 = std::sqrt(hn[2 * i + 1]);
    if (Q3 / std::sqrt(double(dd)) < atol || Q3 / S3 < rtol) resolved[k] = 1;
  }
}

// ---------------------------------------------------------------------------------------------
// save / load  (HSSMatrix::write / read, HSS/HSSMatrix.cpp:438-510).  Own, self-describing binary layout: the
// reference writes raw object images (sizeof(DenseMatrix) including its vtable and data pointers), which only the
// binary that wrote them can read back, so there is no common file format to follow.
//   "HSSAMD01" | int n, nnodes | per node (pre-order): 13 ints {lo, m, lvl, height, c0, c1, parent, Ustate, Vstate,
//   rU, rV, mU, mV}, then nine blocks, each as (int64 count, payload): D, B01, B10, XU, permU, Ir, XV, permV, Ic
//   (doubles / ints, little-endian, column-major; count 0 when the node does not have the block).
// ---------------------------------------------------------------------------------------------
namespace {
template <class T> void put(std::ostream& os, const T* p, size_t n) {
  const long long c = (long long)n;
  os.write((const char*)&c, sizeof(c));
  if (n) os.write((const char*)p, sizeof(T) * n);
}
template <class T> std::vector<T> get(std::istream& is) {
  long long c = -1;
  is.read((char*)&c, sizeof(c));
  if (!is || c < 0 || c > (1LL << 40)) throw std::runtime_error("HSS file is truncated or corrupt");
  std::vector<T> v((size_t)c);
  if (c) is.read((char*)v.data(), sizeof(T) * (size_t)c);
  if (!is) throw std::runtime_error("HSS file is truncated");
  return v;
}
}  // namespace

void DeviceHSS::save(std::ostream& os) const {
  if (o_.world > 1) throw std::invalid_argument("write: not supported for a matrix sharded over several processes");
  ck(hssk_sync(ctx_));
  os.write("HSSAMD01", 8);
  const int hdr[2] = {n_, (int)nodes_.size()};
  os.write((const char*)hdr, sizeof(hdr));
  auto dump_d = [&](const double* d, size_t cnt) {
    std::vector<double> buf(d ? cnt : 0);
    if (!buf.empty()) ck(hssk_memcpy_d2h(ctx_, buf.data(), d, (long long)(sizeof(double) * buf.size())));
    put(os, buf.data(), buf.size());
  };
  auto dump_i = [&](const int* d, size_t cnt) {
    std::vector<int> buf(d ? cnt : 0);
    if (!buf.empty()) ck(hssk_memcpy_d2h(ctx_, buf.data(), d, (long long)(sizeof(int) * buf.size())));
    put(os, buf.data(), buf.size());
  };
  for (const Node& nd : nodes_) {
    const int f[13] = {nd.lo, nd.m, nd.lvl, nd.height, nd.c0, nd.c1, nd.parent, nd.Ustate, nd.Vstate, nd.rU, nd.rV, nd.mU, nd.mV};
    os.write((const char*)f, sizeof(f));
    const bool basis = nd.lvl > 0 && nd.compressed();
    dump_d(nd.leaf() ? nd.D : nullptr, (size_t)nd.m * nd.m);
    dump_d(nd.leaf() ? nullptr : nd.B01, nd.leaf() ? 0 : (size_t)nodes_[nd.c0].rU * nodes_[nd.c1].rV);
    dump_d(nd.leaf() ? nullptr : nd.B10, nd.leaf() ? 0 : (size_t)nodes_[nd.c1].rU * nodes_[nd.c0].rV);
    dump_d(basis ? nd.XU : nullptr, (size_t)nd.rU * std::max(nd.mU - nd.rU, 0));
    dump_i(basis ? nd.permU : nullptr, nd.mU);
    put(os, nd.Ir.data(), basis ? nd.Ir.size() : 0);
    dump_d(basis ? nd.XV : nullptr, (size_t)nd.rV * std::max(nd.mV - nd.rV, 0));
    dump_i(basis ? nd.permV : nullptr, nd.mV);
    put(os, nd.Ic.data(), basis ? nd.Ic.size() : 0);
  }
  if (!os) throw std::runtime_error("write: I/O error");
}

std::unique_ptr<DeviceHSS> DeviceHSS::load(std::istream& is, const EngineOptions& opts) {
  char magic[8];
  is.read(magic, 8);
  if (!is || std::memcmp(magic, "HSSAMD01", 8)) throw std::runtime_error("not an HSS matrix file of this library");
  int hdr[2] = {0, 0};
  is.read((char*)hdr, sizeof(hdr));
  const int n = hdr[0], nn = hdr[1];
  if (!is || n < 0 || nn < 1) throw std::runtime_error("corrupt HSS file header");
  struct Rec { int f[13]; std::vector<double> D, B01, B10, XU, XV; std::vector<int> pU, pV, Ir, Ic; };
  std::vector<Rec> recs(nn);
  for (auto& r : recs) {
    is.read((char*)r.f, sizeof(r.f));
    if (!is) throw std::runtime_error("HSS file is truncated");
    r.D = get<double>(is); r.B01 = get<double>(is); r.B10 = get<double>(is);
    r.XU = get<double>(is); r.pU = get<int>(is); r.Ir = get<int>(is);
    r.XV = get<double>(is); r.pV = get<int>(is); r.Ic = get<int>(is);
  }
  // cluster tree from the node table (children follow their parent in pre-order)
  std::function<structured::ClusterTree(int)> tree_of = [&](int i) {
    if (i < 0 || i >= nn) throw std::runtime_error("corrupt HSS file (tree)");
    structured::ClusterTree t(recs[i].f[1]);
    if (recs[i].f[4] >= 0) {
      if (recs[i].f[4] <= i || recs[i].f[5] <= i) throw std::runtime_error("corrupt HSS file (tree)");
      t.c.resize(2);
      t.c[0] = tree_of(recs[i].f[4]);
      t.c[1] = tree_of(recs[i].f[5]);
    }
    return t;
  };
  structured::ClusterTree tree = tree_of(0);
  if (tree.size != n) throw std::runtime_error("corrupt HSS file (size)");
  std::unique_ptr<DeviceHSS> H(new DeviceHSS(n, opts, &tree));
  if ((int)H->nodes_.size() != nn) throw std::runtime_error("corrupt HSS file (node count)");
  Arena& P = *H->persist_;
  auto up_d = [&](const std::vector<double>& v) -> double* {
    double* d = P.dbl(std::max<size_t>(v.size(), 1));
    if (!v.empty()) ck(hssk_memcpy_h2d(H->ctx_, d, v.data(), (long long)(sizeof(double) * v.size())));
    return d;
  };
  auto up_i = [&](const std::vector<int>& v) -> int* {
    int* d = P.ints(std::max<size_t>(v.size(), 1));
    if (!v.empty()) ck(hssk_memcpy_h2d(H->ctx_, d, v.data(), (long long)(sizeof(int) * v.size())));
    return d;
  };
  for (int i = 0; i < nn; i++) {
    Node& nd = H->nodes_[i];
    const Rec& r = recs[i];
    if (nd.lo != r.f[0] || nd.m != r.f[1] || nd.c0 != r.f[4] || nd.c1 != r.f[5]) throw std::runtime_error("corrupt HSS file (node table)");
    nd.Ustate = r.f[7]; nd.Vstate = r.f[8]; nd.rU = r.f[9]; nd.rV = r.f[10]; nd.mU = r.f[11]; nd.mV = r.f[12];
    // every block is checked against the node table before it reaches the device: a truncated-but-parseable or foreign
    // file must fail here, not as an out-of-bounds read in the first mult / factor
    if (nd.Ustate < 0 || nd.Ustate > 2 || nd.Vstate < 0 || nd.Vstate > 2) throw std::runtime_error("corrupt HSS file (node state)");
    if (nd.rU < 0 || nd.rV < 0 || nd.mU < 0 || nd.mV < 0 || nd.rU > nd.mU || nd.rV > nd.mV) throw std::runtime_error("corrupt HSS file (ranks)");
    const bool used = nd.compressed() || nd.lvl == 0;   // blocks exist once the node has been processed
    if (nd.leaf()) {
      if (r.D.size() != (used || !r.D.empty() ? (size_t)nd.m * nd.m : 0)) throw std::runtime_error("corrupt HSS file (leaf block size)");
      if (nd.lvl > 0 && nd.compressed() && (nd.mU != nd.m || nd.mV != nd.m)) throw std::runtime_error("corrupt HSS file (leaf basis rows)");
      if (!r.D.empty()) nd.D = up_d(r.D);
    } else {
      const Rec &a = recs[nd.c0], &b = recs[nd.c1];   // (rU, rV) of the children: f[9], f[10]
      if (used) {
        if (r.B01.size() != (size_t)a.f[9] * b.f[10] || r.B10.size() != (size_t)b.f[9] * a.f[10])
          throw std::runtime_error("corrupt HSS file (coupling block sizes)");
        if (nd.lvl > 0 && (nd.mU != a.f[9] + b.f[9] || nd.mV != a.f[10] + b.f[10])) throw std::runtime_error("corrupt HSS file (basis rows)");
      } else if (!r.B01.empty() || !r.B10.empty()) throw std::runtime_error("corrupt HSS file (coupling blocks of an untouched node)");
      nd.B01 = up_d(r.B01); nd.B10 = up_d(r.B10);
    }
    if (nd.lvl > 0 && nd.compressed()) {
      if ((int)r.pU.size() != nd.mU || (int)r.pV.size() != nd.mV || (int)r.Ir.size() != nd.rU || (int)r.Ic.size() != nd.rV ||
          r.XU.size() != (size_t)nd.rU * std::max(nd.mU - nd.rU, 0) || r.XV.size() != (size_t)nd.rV * std::max(nd.mV - nd.rV, 0))
        throw std::runtime_error("corrupt HSS file (basis sizes)");
      auto is_perm = [](const std::vector<int>& p) {
        std::vector<char> seen(p.size(), 0);
        for (int v : p) { if (v < 0 || v >= (int)p.size() || seen[v]) return false; seen[v] = 1; }
        return true;
      };
      if (!is_perm(r.pU) || !is_perm(r.pV)) throw std::runtime_error("corrupt HSS file (permutation)");
      for (int v : r.Ir) if (v < 0 || v >= n) throw std::runtime_error("corrupt HSS file (row index set)");
      for (int v : r.Ic) if (v < 0 || v >= n) throw std::runtime_error("corrupt HSS file (column index set)");
      nd.XU = up_d(r.XU); nd.permU = up_i(r.pU); nd.hpermU = r.pU; nd.Ir = r.Ir; nd.dIr = up_i(r.Ir);
      nd.XV = up_d(r.XV); nd.permV = up_i(r.pV); nd.hpermV = r.pV; nd.Ic = r.Ic; nd.dIc = up_i(r.Ic);
    }
  }
  ck(hssk_sync(H->ctx_));
  return H;
}

// ---------------------------------------------------------------------------------------------
// shift
// ---------------------------------------------------------------------------------------------
void DeviceHSS::shift(double sigma) {
  OpGuard op_guard(op_mu_);
  std::vector<hssk_shift_desc> d;
  for (auto& nd : nodes_)
    if (nd.leaf() && nd.D) d.push_back(hssk_shift_desc{nd.D, nd.m, nd.m});
  if (!d.empty()) ck(hssk_shift_diag(ctx_, d.data(), (int)d.size(), sigma));
  ck(hssk_sync(ctx_));
  factored_ = partial_factored_ = schur_ready_ = false;  // the ULV factors are stale (examples/dense/testStructured.cpp:199)
  drop_plans();
}

void DeviceHSS::shift_cplx(double re, double im) {
  OpGuard op_guard(op_mu_);
  std::vector<hssk_shift_desc> d;
  for (auto& nd : nodes_)
    if (nd.leaf() && nd.D) {
      if ((nd.lo | nd.m) & 1) throw std::logic_error("shift_cplx: leaf boundaries must be even (embedded complex matrix)");
      d.push_back(hssk_shift_desc{nd.D, nd.m, nd.m});
    }
  if (!d.empty()) ck(hssk_shift_diag_cplx(ctx_, d.data(), (int)d.size(), re, im));
  ck(hssk_sync(ctx_));
  factored_ = partial_factored_ = schur_ready_ = false;
  drop_plans();
}

// ---------------------------------------------------------------------------------------------
// mult: apply_HSS (HSSMatrix.cpp:419-435, HSSMatrix.apply.hpp:55-220)
// ---------------------------------------------------------------------------------------------
void DeviceHSS::mult(char trans, int nrhs, const double* x, long long ldx, double* y, long long ldy,
                     bool on_device, double beta) {
  OpGuard op_guard(op_mu_);
  mult_sub(0, trans, nrhs, x, ldx, y, ldy, on_device, beta);
}

// op(H_sr) x for the HSS sub-matrix rooted at node sr (sr = 0: the whole matrix; sr = a child of the root: the
// diagonal block H00 / H11 that the reference reaches through child(c)->apply_fwd / apply_bwd, HSSMatrix.Schur.hpp:81-82).
// x / y have rows(sr) rows.
void DeviceHSS::mult_sub(int sr, char trans, int nrhs, const double* x, long long ldx, double* y, long long ldy,
                         bool on_device, double beta) {
  ensure_ready("mult");
  if (nrhs <= 0 || n_ == 0) return;
  if (sr != 0 && o_.world != 1) throw std::logic_error("mult_sub: sub-matrix products need a single-process matrix");
  double t0 = now();
  const bool T = !(trans == 'N' || trans == 'n');
  const bool plannable = sr == 0 && on_device && o_.world == 1 && plans_enabled();
  const PlanKey key{0, T ? 'T' : 'N', nrhs, (const void*)x, (void*)y, ldx, ldy, beta};
  if (plannable) {
    auto it = plans_.find(key);
    if (it != plans_.end() && it->second.plan) {
      ck(hssk_plan_replay(ctx_, it->second.plan));
      ck(hssk_sync(ctx_));
      if (hssk_sweep_status(ctx_)) throw std::runtime_error(std::string("mult: ") + hssk_last_error());
      stats_.t_mult = now() - t0;
      return;
    }
  }
  hssk_plan* rec = nullptr;
  if (plannable && plans_.size() > 32) drop_plans();   // many different buffers: start over rather than grow
  if (plannable && ++plans_[key].seen == 2) ck(hssk_plan_begin(ctx_, &rec));
  struct EndRec { hssk_ctx* c; hssk_plan* p; bool done = false; ~EndRec() { if (p && !done) { hssk_plan_end(c); hssk_plan_destroy(p); } } } guard{ctx_, rec};
  Arena& tmp = rec ? *plan_arena_ : *tmp_;
  if (!rec) tmp.rewind();
  const int N = nodes_[sr].m, lo0 = nodes_[sr].lo;
  const int sr_end = subtree_end(sr);
  const double* dx = x;
  double* dy = y;
  long long lx = ldx, ly = ldy;
  if (!on_device) {
    double* bx = tmp.dbl((size_t)N * nrhs);
    double* by = tmp.dbl((size_t)N * nrhs);
    ck(hssk_memcpy2d_h2d(ctx_, bx, sizeof(double) * N, x, sizeof(double) * ldx, sizeof(double) * N, nrhs));
    if (beta != 0.0) ck(hssk_memcpy2d_h2d(ctx_, by, sizeof(double) * N, y, sizeof(double) * ldy, sizeof(double) * N, nrhs));
    dx = bx; dy = by; lx = ly = N;
  }
  if (lx > 0x7fffffffLL || ly > 0x7fffffffLL) throw std::invalid_argument("mult: leading dimension too large");
  // per-node buffers: cat (children's V^H results, rows of the "in" basis), t (U tmp2, rows of "out" basis)
  const size_t nn = nodes_.size();
  std::vector<double*> cat(nn, nullptr), tbuf(nn, nullptr);
  auto rin = [&](const Node& nd) { return T ? nd.rU : nd.rV; };   // rank of the basis applied to the input
  auto rout = [&](const Node& nd) { return T ? nd.rV : nd.rU; };
  auto min_ = [&](const Node& nd) { return T ? nd.mU : nd.mV; };
  auto mout = [&](const Node& nd) { return T ? nd.mV : nd.mU; };
  // (one block: these are the vectors handed from node to node; the single-launch sweep arms the block with a sentinel)
  size_t hand_total = 0;
  for (size_t i = 0; i < nn; i++) {
    const Node& nd = nodes_[i];
    if (nd.leaf() || !mine((int)i) || (int)i < sr || (int)i >= sr_end) continue;
    hand_total += (size_t)std::max(rin(nodes_[nd.c0]) + rin(nodes_[nd.c1]), 1) * nrhs + (size_t)std::max(rout(nodes_[nd.c0]) + rout(nodes_[nd.c1]), 1) * nrhs;
  }
  double* hand = tmp.dbl(std::max<size_t>(hand_total, 1));
  {
    size_t off = 0;
    for (size_t i = 0; i < nn; i++) {
      const Node& nd = nodes_[i];
      if (nd.leaf() || !mine((int)i) || (int)i < sr || (int)i >= sr_end) continue;
      int ci = rin(nodes_[nd.c0]) + rin(nodes_[nd.c1]);
      int co = rout(nodes_[nd.c0]) + rout(nodes_[nd.c1]);
      cat[i] = hand + off; off += (size_t)std::max(ci, 1) * nrhs;
      tbuf[i] = hand + off; off += (size_t)std::max(co, 1) * nrhs;
    }
  }
  // ---- up-sweep, one height: tmp1 = Vin^H [..]
  auto up = [&](const std::vector<int>& ids) {
    std::vector<hssk_rowgather_desc> g;
    std::vector<hssk_gemm_desc> mm;
    for (int id : ids) {
      const Node& nd = nodes_[id];
      if (id == sr) continue;
      const Node& pa = nodes_[nd.parent];
      const int m = min_(nd), r = rin(nd);
      const int* perm = T ? nd.permU : nd.permV;
      const double* X = T ? nd.XU : nd.XV;
      const double* src = nd.leaf() ? dx + (nd.lo - lo0) : cat[id];
      const int lds = nd.leaf() ? (int)lx : std::max(m, 1);
      const int pci = rin(nodes_[pa.c0]) + rin(nodes_[pa.c1]);
      double* dst = cat[nd.parent] + (id == pa.c0 ? 0 : rin(nodes_[pa.c0]));
      const int ldd = std::max(pci, 1);
      if (r == 0) continue;
      g.push_back(hssk_rowgather_desc{src, dst, perm, r, nrhs, lds, ldd, 0, 0});
      if (m > r) {
        double* Tm = tmp.dbl((size_t)(m - r) * nrhs);
        g.push_back(hssk_rowgather_desc{src, Tm, perm + r, m - r, nrhs, lds, m - r, 0, 0});
        mm.push_back(hssk_gemm_desc{X, Tm, dst, r, nrhs, m - r, r, m - r, ldd, 0, 0, 1.0, 1.0});
      }
    }
    if (!g.empty()) ck(hssk_gather_rows(ctx_, g.data(), (int)g.size()));
    if (!mm.empty()) ck(hssk_gemm_vbatched(ctx_, mm.data(), (int)mm.size()));
  };
  // ---- down-sweep, one depth
  auto down = [&](const std::vector<int>& ids) {
    std::vector<hssk_gemm_desc> m1, leafmm, innermm;  // m1: basis expansion X^T tmp2
    std::vector<hssk_rowgather_desc> sc;
    for (int id : ids) {
      const Node& nd = nodes_[id];
      const int mo = mout(nd), ro = rout(nd);
      const int* perm = T ? nd.permV : nd.permU;
      const double* X = T ? nd.XV : nd.XU;
      // tmp2 of this node lives in the parent's t buffer
      const double* tmp2 = nullptr;
      int ld2 = 1;
      if (id != sr) {
        const Node& pa = nodes_[nd.parent];
        tmp2 = tbuf[nd.parent] + (id == pa.c0 ? 0 : rout(nodes_[pa.c0]));
        ld2 = std::max(rout(nodes_[pa.c0]) + rout(nodes_[pa.c1]), 1);
      }
      double* out = nd.leaf() ? dy + (nd.lo - lo0) : tbuf[id];
      // (the root has no basis, so its mU / mV are unset: size t from the children's ranks)
      const int ldo = nd.leaf() ? (int)ly : std::max(rout(nodes_[nd.c0]) + rout(nodes_[nd.c1]), 1);
      const bool expand = id != sr && ro > 0;
      if (nd.leaf()) {
        // c = D b + beta c (+ U tmp2)
        leafmm.push_back(hssk_gemm_desc{nd.D, dx + (nd.lo - lo0), out, nd.m, nrhs, nd.m, nd.m, (int)lx, ldo, T ? 1 : 0, 0, 1.0, beta});
      } else {
        const Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
        const int ri_a = rin(a), ri_b = rin(b), ro_a = rout(a), ro_b = rout(b);
        const int lc = std::max(ri_a + ri_b, 1);
        const double* t1a = cat[id];
        const double* t1b = cat[id] + ri_a;
        const double bet = expand ? 1.0 : 0.0;
        if (!T) {  // tmp2_0 = B01 tmp1_1 ; tmp2_1 = B10 tmp1_0
          innermm.push_back(hssk_gemm_desc{nd.B01, t1b, out, ro_a, nrhs, ri_b, std::max(ro_a, 1), lc, ldo, 0, 0, 1.0, bet});
          innermm.push_back(hssk_gemm_desc{nd.B10, t1a, out + ro_a, ro_b, nrhs, ri_a, std::max(ro_b, 1), lc, ldo, 0, 0, 1.0, bet});
        } else {   // tmp2_0 = B10^T tmp1_1 ; tmp2_1 = B01^T tmp1_0   (ranks: B10 is rU1 x rV0, B01 is rU0 x rV1)
          innermm.push_back(hssk_gemm_desc{nd.B10, t1b, out, ro_a, nrhs, ri_b, std::max(ri_b, 1), lc, ldo, 1, 0, 1.0, bet});
          innermm.push_back(hssk_gemm_desc{nd.B01, t1a, out + ro_a, ro_b, nrhs, ri_a, std::max(ri_a, 1), lc, ldo, 1, 0, 1.0, bet});
        }
      }
      if (expand) {
        // out(perm[:r]) (+)= tmp2 ; out(perm[r:]) (+)= X^T tmp2     (HSSBasisID::apply)
        const int acc = nd.leaf() ? 1 : 0;  // leaves add onto D b; inner nodes initialise t
        sc.push_back(hssk_rowgather_desc{tmp2, out, perm, ro, nrhs, ld2, ldo, 1, acc});
        if (mo > ro) {
          double* E2 = tmp.dbl((size_t)(mo - ro) * nrhs);
          m1.push_back(hssk_gemm_desc{X, tmp2, E2, mo - ro, nrhs, ro, ro, ld2, mo - ro, 1, 0, 1.0, 0.0});
          sc.push_back(hssk_rowgather_desc{E2, out, perm + ro, mo - ro, nrhs, mo - ro, ldo, 1, acc});
        }
      }
    }
    // order: leaves need D b before the accumulate-scatter; inner nodes need the scatter (which
    // initialises t) before the beta = 1 coupling gemm.
    if (!leafmm.empty()) ck(hssk_gemm_vbatched(ctx_, leafmm.data(), (int)leafmm.size()));
    if (!m1.empty()) ck(hssk_gemm_vbatched(ctx_, m1.data(), (int)m1.size()));
    if (!sc.empty()) ck(hssk_gather_rows(ctx_, sc.data(), (int)sc.size()));
    if (!innermm.empty()) ck(hssk_gemm_vbatched(ctx_, innermm.data(), (int)innermm.size()));
  };
  // few right-hand sides: up-sweep and down-sweep of a set of levels as ONE launch (hssk_apply_sweep: a workgroup per
  // node and direction, dependency flags between them) instead of two to four batched launches per level
  static const bool no_fuse = [] { const char* e = std::getenv("STRUMPACK_AMD_NO_FUSED_APPLY"); return e && e[0] == '1'; }();
  const bool fuse = nrhs <= 64 && !no_fuse;   // (more right-hand sides: the batched MFMA launches per level)
  if (fuse) ck(hssk_sweep_arm(ctx_, hand, (long long)hand_total));
  typedef std::vector<std::vector<int>> Levels;
  auto sweep = [&](const Levels* ups, const Levels* downs) -> bool {
    std::vector<hssk_apply_up_desc> U;
    std::vector<hssk_apply_down_desc> Dn;
    std::vector<int> wu(nn, -1), wd(nn, -1);
    if (ups)
      for (auto& ids : *ups)
        for (int id : ids) {
          if (id == sr) continue;
          const Node& nd = nodes_[id];
          const Node& pa = nodes_[nd.parent];
          hssk_apply_up_desc d{};
          d.m = min_(nd); d.r = rin(nd);
          d.perm = T ? nd.permU : nd.permV;
          d.X = T ? nd.XU : nd.XV;
          d.src = nd.leaf() ? dx + (nd.lo - lo0) : cat[id];
          d.lds = nd.leaf() ? (int)lx : std::max(d.m, 1);
          d.dst = cat[nd.parent] + (id == pa.c0 ? 0 : rin(nodes_[pa.c0]));
          d.ldd = std::max(rin(nodes_[pa.c0]) + rin(nodes_[pa.c1]), 1);
          d.inner = nd.leaf() ? 0 : 1;
          d.wait0 = nd.leaf() ? -1 : wu[nd.c0];
          d.wait1 = nd.leaf() ? -1 : wu[nd.c1];
          wu[id] = (int)U.size();
          U.push_back(d);
        }
    const int nup = (int)U.size();
    if (downs)
      for (auto& ids : *downs)
        for (int id : ids) {
          const Node& nd = nodes_[id];
          hssk_apply_down_desc d{};
          d.wait0 = d.wait1 = d.wait2 = -1;
          if (id != sr) {
            const Node& pa = nodes_[nd.parent];
            d.tmp2 = tbuf[nd.parent] + (id == pa.c0 ? 0 : rout(nodes_[pa.c0]));
            d.ld2 = std::max(rout(nodes_[pa.c0]) + rout(nodes_[pa.c1]), 1);
            d.perm = T ? nd.permV : nd.permU;
            d.X = T ? nd.XV : nd.XU;
            d.mo = mout(nd); d.ro = rout(nd);
            d.wait0 = wd[nd.parent];
          }
          d.trans = T ? 1 : 0;
          if (nd.leaf()) {
            d.D = nd.D; d.x = dx + (nd.lo - lo0); d.ldx = (int)lx; d.m = nd.m; d.beta = beta;
            d.out = dy + (nd.lo - lo0); d.ldo = (int)ly;
            if (!d.D) return false;
          } else {
            const Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
            d.B01 = nd.B01; d.B10 = nd.B10; d.t1 = cat[id];
            d.ri_a = rin(a); d.ri_b = rin(b); d.ro_a = rout(a); d.ro_b = rout(b);
            d.ldt1 = std::max(d.ri_a + d.ri_b, 1);
            d.out = tbuf[id]; d.ldo = std::max(d.ro_a + d.ro_b, 1);
            d.wait1 = wu[nd.c0]; d.wait2 = wu[nd.c1];
            if (!d.B01 || !d.B10) return false;
          }
          wd[id] = nup + (int)Dn.size();
          Dn.push_back(d);
        }
    if (U.empty() && Dn.empty()) return true;
    const int rc = hssk_apply_sweep(ctx_, U.data(), nup, Dn.data(), (int)Dn.size(), nrhs);
    if (rc == 2) return false;
    ck(rc);
    return true;
  };
  std::vector<std::vector<int>> sub_h, sub_d;
  if (sr != 0) { sub_h = sublists(own_by_height_, sr); sub_d = sublists(own_by_depth_, sr); }
  const Levels& ups_own = sr ? sub_h : own_by_height_;
  const Levels& downs_own = sr ? sub_d : own_by_depth_;
  // single process: the whole product is one launch
  const bool whole = fuse && !dist_subtree_ && sweep(&ups_own, &downs_own);
  if (!whole && !(fuse && dist_subtree_ && sweep(&ups_own, nullptr)))
    for (auto& ids : ups_own) up(ids);
  if (dist_subtree_) {
    // publish tmp1 (rin x nrhs) of the cut nodes into every rank's top buffers
    const int G = o_.world, me = o_.rank;
    int rm = 1;
    for (int g = 0; g < G; g++) rm = std::max(rm, rin(nodes_[cut_nodes_[g]]));
    const size_t blk = (size_t)rm * nrhs;
    double* buf = tmp.dbl(blk * G);
    auto slice = [&](int g, double*& p1, int& ld1) {
      const int id = cut_nodes_[g];
      const Node& pa = nodes_[nodes_[id].parent];
      p1 = cat[nodes_[id].parent] + (id == pa.c0 ? 0 : rin(nodes_[pa.c0]));
      ld1 = std::max(rin(nodes_[pa.c0]) + rin(nodes_[pa.c1]), 1);
    };
    {
      double* p1; int ld1;
      slice(me, p1, ld1);
      const int r = rin(nodes_[cut_nodes_[me]]);
      if (r) { hssk_rowgather_desc pk{p1, buf + blk * me, nullptr, r, nrhs, ld1, rm, 0, 0}; ck(hssk_gather_rows(ctx_, &pk, 1)); }
    }
    comm(buf, (long long)(sizeof(double) * blk));
    std::vector<hssk_rowgather_desc> upk;
    for (int g = 0; g < G; g++) {
      if (g == me) continue;
      double* p1; int ld1;
      slice(g, p1, ld1);
      const int r = rin(nodes_[cut_nodes_[g]]);
      if (r) upk.push_back(hssk_rowgather_desc{buf + blk * g, p1, nullptr, r, nrhs, rm, ld1, 0, 0});
    }
    if (!upk.empty()) ck(hssk_gather_rows(ctx_, upk.data(), (int)upk.size()));
    if (!(fuse && sweep(&top_by_height_, &top_by_depth_))) {
      for (auto& ids : top_by_height_) up(ids);
      for (auto& ids : top_by_depth_) down(ids);
    }
  }
  if (!whole && !(fuse && dist_subtree_ && sweep(nullptr, &downs_own)))
    for (auto& ids : downs_own) down(ids);
  if (dist_subtree_) allgather_rows(dy, ly, nrhs);
  if (!on_device) ck(hssk_memcpy2d_d2h(ctx_, y, sizeof(double) * ldy, dy, sizeof(double) * N, sizeof(double) * N, nrhs));
  if (rec) { ck(hssk_plan_end(ctx_)); guard.done = true; plans_[key].plan = rec; }
  ck(hssk_sync(ctx_));
  if (hssk_sweep_status(ctx_)) throw std::runtime_error(std::string("mult: ") + hssk_last_error());
  stats_.t_mult = now() - t0;
  {
    double bm = 0;
    for (int i = sr; i < sr_end; i++) {
      const Node& nd = nodes_[i];
      if (nd.leaf()) bm += (double)nd.m * nd.m;
      else bm += (double)nodes_[nd.c0].rU * nodes_[nd.c1].rV + (double)nodes_[nd.c1].rU * nodes_[nd.c0].rV;
      if (i != sr) bm += (double)nd.rU * (nd.mU - nd.rU) + (double)nd.rV * (nd.mV - nd.rV);
    }
    stats_.b_mult = 8.0 * bm;
  }
}

// ---------------------------------------------------------------------------------------------
// ULV factorization (HSSMatrix.factor.hpp:51-147)
// ---------------------------------------------------------------------------------------------
void DeviceHSS::factor() { factor_sub(0, false); }

// HSSMatrix::partial_factor (HSSMatrix.factor.hpp:43-49): ULV-factor the (0,0) block only -- child(0) is eliminated as
// the root of its own subtree -- and keep its reduced column basis Vhat (HSSFactors::Vhat(), HSSExtra.hpp:191) for
// the Schur complement update of the (1,1) block.
void DeviceHSS::partial_factor() {
  OpGuard op_guard(op_mu_);
  if (nodes_[0].leaf()) return;
  if (o_.world != 1) throw std::logic_error("partial_factor: needs a single-process matrix");
  factor_sub(nodes_[0].c0, true);
}

void DeviceHSS::factor_sub(int sr, bool partial) {
  OpGuard op_guard(op_mu_);
  ensure_ready("factor");
  double t0 = now();
  ck(hssk_sync(ctx_));
  drop_plans();   // recorded sweeps reference the old factors
  fact_->reset();
  stats_.f_ulv = 0;
  for (auto& nd : nodes_) nd.Qt = nd.Rlq = nd.W1 = nd.Vt0 = nd.Dt = nd.Vt1 = nd.LU = nd.WQ = nd.Tinv = nd.TinvU = nd.Vt0T = nullptr, nd.piv = nullptr;
  const size_t nn = nodes_.size();
  std::vector<double*> Dh(nn, nullptr), Vh(nn, nullptr), Vd(nn, nullptr);
  // inverted diagonal blocks for the single-launch solve sweeps: nothing in the factorization reads them, so the
  // descriptors of all levels are collected and take ONE launch at the end (a launch per level was 10-20 us each)
  std::vector<hssk_trtri_desc> ti;
  // A node's reduced block Dt (rU x rU) is written by its products straight into the diagonal block of the PARENT's Dh
  // (allocated here, ahead of the parent's level): no copy launch per level.  The cut nodes of a distributed tree keep a
  // compact Dt of their own -- it travels through exchange_cut_factor().
  std::vector<char> is_cut(nn, 0);
  if (dist_subtree_) for (int c : cut_nodes_) is_cut[c] = 1;
  auto dt_slot = [&](int id, int r, int& ld) -> double* {
    const Node& nd = nodes_[id];
    if (nd.parent < 0 || is_cut[id]) { ld = std::max(r, 1); return fact_->dbl((size_t)ld * ld); }
    const Node& pa = nodes_[nd.parent];
    const int mu = nodes_[pa.c0].rU + nodes_[pa.c1].rU;
    ld = std::max(mu, 1);
    if (!Dh[nd.parent]) Dh[nd.parent] = fact_->dbl((size_t)ld * ld);
    const int off = id == pa.c0 ? 0 : nodes_[pa.c0].rU;
    return Dh[nd.parent] + off + (size_t)off * ld;
  };
  auto level = [&](const std::vector<int>& ids) {
    if (ids.empty()) return;
    // ---- assemble Dh (mU x mU) and Vh (mU x rV)
    std::vector<hssk_colgather_desc> cp;
    std::vector<hssk_gemm_desc> g0, g1;
    Arena& tmp = *tmp_;   // (rewound once per factorization: the levels are enqueued back to back, no host sync between them)
    for (int id : ids) {
      Node& nd = nodes_[id];
      const bool root = id == sr;
      const int mu = nd.leaf() ? nd.m : nodes_[nd.c0].rU + nodes_[nd.c1].rU;
      if (nd.leaf() && !root) {
        Dh[id] = nd.D;   // only read below (the root's block is factored in place: it gets a copy)
      } else if (nd.leaf()) {
        Dh[id] = fact_->dbl((size_t)std::max(mu, 1) * std::max(mu, 1));
        cp.push_back(hssk_colgather_desc{nd.D, Dh[id], nullptr, nd.m, nd.m, nd.m, nd.m, 0});
      } else {
        Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
        // D = [Dt0, B01 Vt1_1^T ; B10 Vt1_0^T, Dt1]; the diagonal blocks are already in place (dt_slot) unless the children
        // are cut nodes
        if (!Dh[id]) {
          Dh[id] = fact_->dbl((size_t)std::max(mu, 1) * std::max(mu, 1));
          cp.push_back(hssk_colgather_desc{a.Dt, Dh[id], nullptr, a.rU, a.rU, std::max(a.rU, 1), std::max(mu, 1), 0});
          cp.push_back(hssk_colgather_desc{b.Dt, Dh[id] + a.rU + (size_t)a.rU * mu, nullptr, b.rU, b.rU, std::max(b.rU, 1), std::max(mu, 1), 0});
        }
        g0.push_back(hssk_gemm_desc{nd.B01, b.Vt1, Dh[id] + (size_t)a.rU * mu, a.rU, b.rU, b.rV, std::max(a.rU, 1), std::max(b.rU, 1), std::max(mu, 1), 0, 1, 1.0, 0.0});
        g0.push_back(hssk_gemm_desc{nd.B10, a.Vt1, Dh[id] + a.rU, b.rU, a.rU, a.rV, std::max(b.rU, 1), std::max(a.rU, 1), std::max(mu, 1), 0, 1, 1.0, 0.0});
        stats_.f_ulv += 2.0 * a.rU * (double)b.rU * (a.rV + b.rV);
      }
      if ((!root || partial) && !nd.leaf() && nd.rV) {
        Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
        // Vh = [Vt1_0 Vd(0:rV0, :) ; Vt1_1 Vd(rV0:, :)]   (Vd: the node's dense column basis, formed ahead of the levels)
        g1.push_back(hssk_gemm_desc{a.Vt1, Vd[id], Vh[id], a.rU, nd.rV, a.rV, std::max(a.rU, 1), nd.mV, nd.mU, 0, 0, 1.0, 0.0});
        g1.push_back(hssk_gemm_desc{b.Vt1, Vd[id] + a.rV, Vh[id] + a.rU, b.rU, nd.rV, b.rV, std::max(b.rU, 1), nd.mV, nd.mU, 0, 0, 1.0, 0.0});
        stats_.f_ulv += 2.0 * nd.rV * ((double)a.rU * a.rV + (double)b.rU * b.rV);
      }
    }
    if (!cp.empty()) ck(hssk_gather_cols(ctx_, cp.data(), (int)cp.size()));
    // (the coupling products into Dh and the products that build Vh are independent of each other: one batched launch)
    g0.insert(g0.end(), g1.begin(), g1.end());
    if (!g0.empty()) ck(hssk_gemm_vbatched(ctx_, g0.data(), (int)g0.size()));
    // ---- eliminate
    std::vector<hssk_elem_desc> ge;
    std::vector<hssk_ulvsplit_desc> us;
    std::vector<hssk_gemm_desc> g2, g3;
    std::vector<hssk_qr_desc> qr;
    std::vector<hssk_lu_desc> lu;
    for (int id : ids) {
      Node& nd = nodes_[id];
      if (id == sr) {
        const int mu = nd.leaf() ? nd.m : nodes_[nd.c0].rU + nodes_[nd.c1].rU;
        nd.LU = Dh[id];
        if (partial) nd.Vt0 = Vh[id];   // Vhat: mu x rV, the column basis in the reduced unknowns
        nd.piv = (int*)fact_->alloc(sizeof(int) * (std::max(mu, 1) + 1));
        if (mu) lu.push_back(hssk_lu_desc{nd.LU, mu, mu, nd.piv, nd.piv + mu});
        if (mu && mu <= 256) {
          const size_t nblk = (size_t)(mu + 63) / 64;
          nd.Tinv = fact_->dbl(nblk * 4096);
          nd.TinvU = fact_->dbl(nblk * 4096);
          ti.push_back(hssk_trtri_desc{nd.LU, nd.Tinv, mu, mu, 2});
          ti.push_back(hssk_trtri_desc{nd.LU, nd.TinvU, mu, mu, 1});
        }
        stats_.f_ulv += 2.0 / 3.0 * mu * (double)mu * mu;
        continue;
      }
      const int m = nd.mU, r = nd.rU, rv = nd.rV;
      if (m > r) {
        // W1 = (P^T D)(0:r, :) ; W0^T = (P^T D)(r:, :)^T - W1^T X       (factor.hpp:109-118)
        nd.W1 = fact_->dbl((size_t)std::max(r, 1) * m);
        nd.Rlq = fact_->dbl((size_t)m * (m - r));
        nd.Qt = fact_->dbl((size_t)m * m);
        nd.Vt1 = fact_->dbl((size_t)std::max(r, 1) * std::max(rv, 1));
        int ldt = 1;
        nd.Dt = dt_slot(id, r, ldt);
        if (m <= 256) {   // one fused launch (hssk_ulv_split); larger blocks: two row gathers and a product
          us.push_back(hssk_ulvsplit_desc{Dh[id], m, m, r, nd.permU, nd.XU, std::max(r, 1), nd.W1, std::max(r, 1), nd.Rlq, m});
        } else {
          if (r) ge.push_back(hssk_elem_desc{Dh[id], m, nd.permU, nullptr, 0, 0, nd.W1, r, m, r, 0});
          ge.push_back(hssk_elem_desc{Dh[id], m, nd.permU + r, nullptr, 0, 0, nd.Rlq, m - r, m, m, 1});
          if (r) g2.push_back(hssk_gemm_desc{nd.W1, nd.XU, nd.Rlq, m, m - r, r, r, r, m, 1, 0, -1.0, 1.0});
        }
        // LQ(W0) == QR(W0^T): Q~ (m x m) = Q^T, R~ = L^T                  (factor.hpp:122)
        double* wk = tmp.dbl((size_t)2 * m);
        qr.push_back(hssk_qr_desc{nd.Rlq, m, m, m - r, nd.Qt, m, m, nullptr, wk});
        // Vt0 = Q0 Vh = Q~(:, :m-r)^T Vh, kept TRANSPOSED (Vt0^T = Vh^T Q~(:, 0:m-r): rows contiguous for the solve sweep; the
        // per-level solve reads the same array) ; Vt1 = Q~(:, m-r:)^T Vh ; Dt = W1 Q1^T = W1 Q~(:, m-r:)
        if (rv) {
          nd.Vt0T = fact_->dbl((size_t)rv * (m - r));
          g3.push_back(hssk_gemm_desc{Vh[id], nd.Qt, nd.Vt0T, rv, m - r, m, m, m, rv, 1, 0, 1.0, 0.0});
          if (r) g3.push_back(hssk_gemm_desc{nd.Qt + (size_t)(m - r) * m, Vh[id], nd.Vt1, r, rv, m, m, m, r, 1, 0, 1.0, 0.0});
        }
        if (r) g3.push_back(hssk_gemm_desc{nd.W1, nd.Qt + (size_t)(m - r) * m, nd.Dt, r, r, m, r, m, ldt, 0, 0, 1.0, 0.0});
        // derived factors of the solve sweeps: WQ = W1 Q~(:, 0:m-r) and the inverted diagonal blocks of R~^T
        if (r) {
          nd.WQ = fact_->dbl((size_t)r * (m - r));
          g3.push_back(hssk_gemm_desc{nd.W1, nd.Qt, nd.WQ, r, m - r, m, r, m, r, 0, 0, 1.0, 0.0});
        }
        if (m <= 256) {
          nd.Tinv = fact_->dbl((size_t)((m - r + 63) / 64) * 4096);
          ti.push_back(hssk_trtri_desc{nd.Rlq, nd.Tinv, m - r, m, 0});
        }
        const double k = m - r;
        stats_.f_ulv += 2.0 * k * r * m + (2.0 * m * k * k - 2.0 / 3.0 * k * k * k) + (4.0 * m * m * k - 2.0 * m * k * k) / 1.0 * 0.5 + 2.0 * m * m * rv + 2.0 * r * (double)r * m;
      } else {
        // nothing to eliminate: Dt = P^T D, Vt1 = Vh   (factor.hpp:138-141)
        int ldt = 1;
        nd.Dt = dt_slot(id, m, ldt);
        nd.Vt1 = Vh[id];
        if (m) ge.push_back(hssk_elem_desc{Dh[id], m, nd.permU, nullptr, 0, 0, nd.Dt, m, m, ldt, 0});
      }
    }
    if (!us.empty()) ck(hssk_ulv_split(ctx_, us.data(), (int)us.size()));
    if (!ge.empty()) ck(hssk_gather_elems(ctx_, ge.data(), (int)ge.size()));
    if (!g2.empty()) ck(hssk_gemm_vbatched(ctx_, g2.data(), (int)g2.size()));
    if (!qr.empty()) ck(hssk_qr_vbatched(ctx_, qr.data(), (int)qr.size()));
    if (!g3.empty()) ck(hssk_gemm_vbatched(ctx_, g3.data(), (int)g3.size()));
    if (!lu.empty()) ck(hssk_getrf_vbatched(ctx_, lu.data(), (int)lu.size()));
  };
  std::vector<std::vector<int>> sub_h;
  if (sr != 0) sub_h = sublists(own_by_height_, sr);
  tmp_->rewind();
  {
    // the dense column bases [I; X^T] in row order (leaves: straight into Vh; inner nodes: Vd, multiplied by the children's
    // Vt1 at the node's level) depend on the compression only: one launch for the whole tree instead of one per level
    std::vector<hssk_basis_desc> bd;
    auto prep = [&](const std::vector<int>& ids) {
      for (int id : ids) {
        const Node& nd = nodes_[id];
        if (id == sr && !partial) continue;
        Vh[id] = fact_->dbl((size_t)std::max(nd.mU, 1) * std::max(nd.rV, 1));
        if (!nd.rV) continue;
        double* out = Vh[id];
        if (!nd.leaf()) out = Vd[id] = tmp_->dbl((size_t)nd.mV * nd.rV);
        bd.push_back(hssk_basis_desc{nd.XV, nd.permV, out, nd.mV, nd.rV, nd.rV, nd.mV});
      }
    };
    for (auto& ids : (sr ? sub_h : own_by_height_)) prep(ids);
    if (dist_subtree_) for (auto& ids : top_by_height_) prep(ids);
    if (!bd.empty()) ck(hssk_basis_dense(ctx_, bd.data(), (int)bd.size()));
  }
  for (auto& ids : (sr ? sub_h : own_by_height_)) level(ids);
  if (dist_subtree_) {
    exchange_cut_factor();
    for (auto& ids : top_by_height_) level(ids);
  }
  if (!ti.empty()) ck(hssk_trtri_diag_vbatched(ctx_, ti.data(), (int)ti.size()));
  ck(hssk_sync(ctx_));
  factored_ = sr == 0;
  partial_factored_ = partial;
  schur_ready_ = false;
  stats_.t_factor = now() - t0;
}

// ---------------------------------------------------------------------------------------------
// Sub-tree helpers and the Schur complement of the (0,0) block (HSSMatrix.Schur.hpp)
// ---------------------------------------------------------------------------------------------
int DeviceHSS::subtree_end(int sr) const {
  int id = sr;
  while (!nodes_[id].leaf()) id = nodes_[id].c1;
  return id + 1;
}

std::vector<std::vector<int>> DeviceHSS::sublists(const std::vector<std::vector<int>>& lists, int sr) const {
  const int end = subtree_end(sr);
  std::vector<std::vector<int>> out;
  for (auto& l : lists) {
    std::vector<int> f;
    for (int id : l) if (id >= sr && id < end) f.push_back(id);
    if (!f.empty()) out.push_back(std::move(f));
  }
  return out;
}

void DeviceHSS::mult_child(int c, char trans, int nrhs, const double* x, long long ldx, double* y, long long ldy,
                           bool on_device) {
  OpGuard op_guard(op_mu_);
  if (nodes_[0].leaf()) throw std::logic_error("mult_child: the root is a leaf");
  mult_sub(c == 0 ? nodes_[0].c0 : nodes_[0].c1, trans, nrhs, x, ldx, y, ldy, on_device, 0.0);
}

void DeviceHSS::mult_node(int node, char trans, int nrhs, const double* x, long long ldx, double* y, long long ldy,
                          bool on_device) {
  OpGuard op_guard(op_mu_);
  if (node < 0 || node >= (int)nodes_.size()) throw std::invalid_argument("mult_node: no such node");
  mult_sub(node, trans, nrhs, x, ldx, y, ldy, on_device, 0.0);
}

void DeviceHSS::basis_up(int sr, bool useU, const double* dA, long long lda, int c, double* dOut, int ldout, Arena& wk) {
  if (c <= 0) return;
  if (lda > 0x7fffffffLL) throw std::invalid_argument("basis_up: leading dimension too large");
  const int lo0 = nodes_[sr].lo, end = subtree_end(sr);
  auto rk = [&](const Node& nd) { return useU ? nd.rU : nd.rV; };
  auto rows = [&](const Node& nd) { return nd.leaf() ? nd.m : rk(nodes_[nd.c0]) + rk(nodes_[nd.c1]); };
  std::vector<double*> cat(nodes_.size(), nullptr);
  for (int id = sr; id < end; id++)
    if (!nodes_[id].leaf()) cat[id] = wk.dbl((size_t)std::max(rows(nodes_[id]), 1) * c);
  for (auto& ids : sublists(by_height_, sr)) {
    std::vector<hssk_rowgather_desc> g;
    std::vector<hssk_gemm_desc> mm;
    for (int id : ids) {
      const Node& nd = nodes_[id];
      const int m = rows(nd), r = rk(nd);
      if (r == 0) continue;
      const int* perm = useU ? nd.permU : nd.permV;
      const double* X = useU ? nd.XU : nd.XV;
      const double* src = nd.leaf() ? dA + (nd.lo - lo0) : cat[id];
      const int lds = nd.leaf() ? (int)lda : std::max(m, 1);
      double* dst = dOut;
      int ldd = ldout;
      if (id != sr) {
        const Node& pa = nodes_[nd.parent];
        dst = cat[nd.parent] + (id == pa.c0 ? 0 : rk(nodes_[pa.c0]));
        ldd = std::max(rows(pa), 1);
      }
      g.push_back(hssk_rowgather_desc{src, dst, perm, r, c, lds, ldd, 0, 0});
      if (m > r) {
        double* Tm = wk.dbl((size_t)(m - r) * c);
        g.push_back(hssk_rowgather_desc{src, Tm, perm + r, m - r, c, lds, m - r, 0, 0});
        mm.push_back(hssk_gemm_desc{X, Tm, dst, r, c, m - r, r, m - r, ldd, 0, 0, 1.0, 1.0});
      }
    }
    if (!g.empty()) ck(hssk_gather_rows(ctx_, g.data(), (int)g.size()));
    if (!mm.empty()) ck(hssk_gemm_vbatched(ctx_, mm.data(), (int)mm.size()));
  }
}

void DeviceHSS::basis_down(int sr, bool useU, const double* dIn, int ldin, int c, double* dOut, long long ldo, Arena& wk,
                           bool recurse) {
  if (c <= 0) return;
  if (ldo > 0x7fffffffLL) throw std::invalid_argument("basis_down: leading dimension too large");
  const int lo0 = nodes_[sr].lo;
  auto rk = [&](const Node& nd) { return useU ? nd.rU : nd.rV; };
  auto rows = [&](const Node& nd) { return nd.leaf() ? nd.m : rk(nodes_[nd.c0]) + rk(nodes_[nd.c1]); };
  std::vector<double*> t(nodes_.size(), nullptr);
  std::vector<std::vector<int>> lists;
  if (recurse) lists = sublists(by_depth_, sr);
  else lists.push_back(std::vector<int>{sr});
  for (auto& ids : lists) {
    std::vector<hssk_rowgather_desc> sc;
    std::vector<hssk_gemm_desc> mm, zero;
    for (int id : ids) {
      const Node& nd = nodes_[id];
      const int mo = rows(nd), r = rk(nd);
      if (mo == 0) continue;
      double* out;
      int ld;
      if (nd.leaf() || !recurse) {
        out = dOut + (recurse ? nd.lo - lo0 : 0);
        ld = (int)ldo;
      } else {
        out = t[id] = wk.dbl((size_t)mo * c);
        ld = mo;
      }
      const double* in = dIn;
      int ldi = ldin;
      if (id != sr) {
        const Node& pa = nodes_[nd.parent];
        in = t[nd.parent] + (id == pa.c0 ? 0 : rk(nodes_[pa.c0]));
        ldi = std::max(rows(pa), 1);
      }
      if (r == 0) {   // no basis: this block row of the product is zero (Schur.hpp:262, :269)
        zero.push_back(hssk_gemm_desc{out, out, out, mo, c, 0, ld, 1, ld, 0, 0, 0.0, 0.0});
        continue;
      }
      const int* perm = useU ? nd.permU : nd.permV;
      const double* X = useU ? nd.XU : nd.XV;
      // out(perm[:r]) = in ; out(perm[r:]) = X^T in      (HSSBasisID::apply)
      sc.push_back(hssk_rowgather_desc{in, out, perm, r, c, ldi, ld, 1, 0});
      if (mo > r) {
        double* E2 = wk.dbl((size_t)(mo - r) * c);
        mm.push_back(hssk_gemm_desc{X, in, E2, mo - r, c, r, r, ldi, mo - r, 1, 0, 1.0, 0.0});
        sc.push_back(hssk_rowgather_desc{E2, out, perm + r, mo - r, c, mo - r, ld, 1, 0});
      }
    }
    if (!zero.empty()) ck(hssk_gemm_vbatched(ctx_, zero.data(), (int)zero.size()));
    if (!mm.empty()) ck(hssk_gemm_vbatched(ctx_, mm.data(), (int)mm.size()));
    if (!sc.empty()) ck(hssk_gather_rows(ctx_, sc.data(), (int)sc.size()));
  }
}

DeviceHSS::SchurDims DeviceHSS::schur_dims() const {
  SchurDims d;
  if (nodes_[0].leaf()) return d;
  const Node &a = nodes_[nodes_[0].c0], &b = nodes_[nodes_[0].c1];
  d.n0 = a.m; d.n1 = b.m;
  d.rV0 = a.rV; d.rU0 = a.rU; d.rV1 = b.rV; d.rU1 = b.rU;
  d.mu0 = a.leaf() ? a.m : nodes_[a.c0].rU + nodes_[a.c1].rU;
  return d;
}

void DeviceHSS::schur_update(double* Theta, long long ldt, double* DUB01, long long ldd, double* Phi, long long ldp,
                             double* Vhat, long long ldv) {
  OpGuard op_guard(op_mu_);
  ensure_ready("Schur_update");
  if (nodes_[0].leaf()) return;    // Schur.hpp:42
  if (!partial_factored_) throw std::logic_error("Schur_update: partial_factor() has not been called");
  const Node& root = nodes_[0];
  const Node &a = nodes_[root.c0], &b = nodes_[root.c1];
  const SchurDims d = schur_dims();
  ck(hssk_sync(ctx_));
  schur_->rewind();
  Arena wk;
  auto L = [](int x) { return std::max(x, 1); };
  sDUB01_ = schur_->dbl((size_t)L(d.mu0) * L(d.rV1));
  sTheta_ = schur_->dbl((size_t)L(d.n1) * L(d.rV0));
  sPhi_ = schur_->dbl((size_t)L(d.n1) * L(d.mu0));
  sVtDUB01_ = schur_->dbl((size_t)L(d.rV0) * L(d.rV1));
  sW_ = schur_->dbl((size_t)L(d.rU1) * L(d.rV1));
  // DUB01 = D00^{-1} (U0 B01)                                         (Schur.hpp:46-48)
  basis_down(root.c0, true, root.B01, L(d.rU0), d.rV1, sDUB01_, L(d.mu0), wk, false);
  if (d.mu0 && d.rV1) {
    hssk_lusolve_desc ls{a.LU, a.piv, sDUB01_, d.mu0, d.rV1, d.mu0, L(d.mu0)};
    ck(hssk_getrs_vbatched(ctx_, &ls, 1));
  }
  // Theta = U1big B10 ; Phi = V1big DUB01^T                          (Schur.hpp:52-58)
  basis_down(root.c1, true, root.B10, L(d.rU1), d.rV0, sTheta_, L(d.n1), wk);
  double* Dt = wk.dbl((size_t)L(d.rV1) * L(d.mu0));
  if (d.mu0 && d.rV1) {
    hssk_transpose_desc tr{sDUB01_, Dt, d.mu0, d.rV1, L(d.mu0), L(d.rV1)};
    ck(hssk_transpose(ctx_, &tr, 1));
  }
  basis_down(root.c1, false, Dt, L(d.rV1), d.mu0, sPhi_, L(d.n1), wk);
  // small products reused by every Schur_product_*: Vhat^T DUB01 (rV0 x rV1) and W = B10 Vhat^T DUB01 (rU1 x rV1)
  std::vector<hssk_gemm_desc> g;
  g.push_back(hssk_gemm_desc{a.Vt0, sDUB01_, sVtDUB01_, d.rV0, d.rV1, d.mu0, L(d.mu0), L(d.mu0), L(d.rV0), 1, 0, 1.0, 0.0});
  ck(hssk_gemm_vbatched(ctx_, g.data(), 1));
  g[0] = hssk_gemm_desc{root.B10, sVtDUB01_, sW_, d.rU1, d.rV1, d.rV0, L(d.rU1), L(d.rV0), L(d.rU1), 0, 0, 1.0, 0.0};
  ck(hssk_gemm_vbatched(ctx_, g.data(), 1));
  ck(hssk_sync(ctx_));
  schur_ready_ = true;
  auto get = [&](double* h, long long ldh, const double* dsrc, int rows, int cols) {
    if (h && rows > 0 && cols > 0)
      ck(hssk_memcpy2d_d2h(ctx_, h, sizeof(double) * ldh, dsrc, sizeof(double) * rows, sizeof(double) * rows, cols));
  };
  get(Theta, ldt, sTheta_, d.n1, d.rV0);
  get(DUB01, ldd, sDUB01_, d.mu0, d.rV1);
  get(Phi, ldp, sPhi_, d.n1, d.mu0);
  get(Vhat, ldv, a.Vt0, d.mu0, d.rV0);
}

void DeviceHSS::schur_product_direct(int c, const double* R, long long ldr, double* Sr, long long ldsr, double* Sc,
                                     long long ldsc, bool on_device) {
  OpGuard op_guard(op_mu_);
  if (!schur_ready_) throw std::logic_error("Schur_product_direct: Schur_update() has not been called");
  if (c <= 0) return;
  const Node& root = nodes_[0];
  const Node& a = nodes_[root.c0];
  const SchurDims d = schur_dims();
  Arena wk;
  auto L = [](int x) { return std::max(x, 1); };
  const int n1 = d.n1;
  const double* dR = R;
  double *dSr = Sr, *dSc = Sc;
  long long lr = ldr, lsr = ldsr, lsc = ldsc;
  if (!on_device) {
    double* b = wk.dbl((size_t)n1 * c);
    ck(hssk_memcpy2d_h2d(ctx_, b, sizeof(double) * n1, R, sizeof(double) * ldr, sizeof(double) * n1, c));
    dR = b; dSr = wk.dbl((size_t)n1 * c); dSc = wk.dbl((size_t)n1 * c);
    lr = lsr = lsc = n1;
  }
  if (lsr > 0x7fffffffLL || lsc > 0x7fffffffLL) throw std::invalid_argument("Schur_product_direct: leading dimension too large");
  // Sr = H11 R, Sc = H11^T R; the basis products V1big^T R / U1big^T R are the forward halves of those applies
  mult_sub(root.c1, 'N', c, dR, lr, dSr, lsr, true, 0.0);
  mult_sub(root.c1, 'T', c, dR, lr, dSc, lsc, true, 0.0);
  double* V1tR = wk.dbl((size_t)L(d.rV1) * c);
  double* U1tR = wk.dbl((size_t)L(d.rU1) * c);
  basis_up(root.c1, false, dR, lr, c, V1tR, L(d.rV1), wk);
  basis_up(root.c1, true, dR, lr, c, U1tR, L(d.rU1), wk);
  // Sr -= Theta (Vhat^T DUB01) (V1big^T R) ;  Sc -= Phi Vhat B10^T (U1big^T R)        (Schur.hpp:60-71)
  double* t1 = wk.dbl((size_t)L(d.rV0) * c);
  double* t2 = wk.dbl((size_t)L(d.rV0) * c);
  double* t3 = wk.dbl((size_t)L(d.mu0) * c);
  std::vector<hssk_gemm_desc> g(2);
  g[0] = hssk_gemm_desc{sVtDUB01_, V1tR, t1, d.rV0, c, d.rV1, L(d.rV0), L(d.rV1), L(d.rV0), 0, 0, 1.0, 0.0};
  g[1] = hssk_gemm_desc{root.B10, U1tR, t2, d.rV0, c, d.rU1, L(d.rU1), L(d.rU1), L(d.rV0), 1, 0, 1.0, 0.0};
  ck(hssk_gemm_vbatched(ctx_, g.data(), 2));
  g[0] = hssk_gemm_desc{a.Vt0, t2, t3, d.mu0, c, d.rV0, L(d.mu0), L(d.rV0), L(d.mu0), 0, 0, 1.0, 0.0};
  ck(hssk_gemm_vbatched(ctx_, g.data(), 1));
  g[0] = hssk_gemm_desc{sTheta_, t1, dSr, n1, c, d.rV0, L(n1), L(d.rV0), (int)lsr, 0, 0, -1.0, 1.0};
  g[1] = hssk_gemm_desc{sPhi_, t3, dSc, n1, c, d.mu0, L(n1), L(d.mu0), (int)lsc, 0, 0, -1.0, 1.0};
  ck(hssk_gemm_vbatched(ctx_, g.data(), 2));
  if (!on_device) {
    ck(hssk_memcpy2d_d2h(ctx_, Sr, sizeof(double) * ldsr, dSr, sizeof(double) * n1, sizeof(double) * n1, c));
    ck(hssk_memcpy2d_d2h(ctx_, Sc, sizeof(double) * ldsc, dSc, sizeof(double) * n1, sizeof(double) * n1, c));
  }
  ck(hssk_sync(ctx_));
}

void DeviceHSS::schur_product_indirect(int c, const double* R0, long long ldr0, const double* R1, long long ldr1,
                                       const double* Sr1, long long ldsr1, const double* Sc1, long long ldsc1,
                                       double* Sr, long long ldsr, double* Sc, long long ldsc, bool on_device) {
  OpGuard op_guard(op_mu_);
  if (nodes_[0].leaf()) return;   // Schur.hpp:158
  if (!schur_ready_) throw std::logic_error("Schur_product_indirect: Schur_update() has not been called");
  if (c <= 0) return;
  const Node& root = nodes_[0];
  const SchurDims d = schur_dims();
  Arena wk;
  auto L = [](int x) { return std::max(x, 1); };
  const int n0 = d.n0, n1 = d.n1;
  const double *dR0 = R0, *dR1 = R1;
  double *dSr = Sr, *dSc = Sc;
  long long l0 = ldr0, l1 = ldr1, lsr = ldsr, lsc = ldsc;
  auto up = [&](const double* h, long long ldh, int rows) {
    double* b = wk.dbl((size_t)L(rows) * c);
    if (rows) ck(hssk_memcpy2d_h2d(ctx_, b, sizeof(double) * rows, h, sizeof(double) * ldh, sizeof(double) * rows, c));
    return b;
  };
  if (!on_device) {
    dR0 = up(R0, ldr0, n0); dR1 = up(R1, ldr1, n1);
    dSr = up(Sr1, ldsr1, n1); dSc = up(Sc1, ldsc1, n1);
    l0 = n0; l1 = lsr = lsc = n1;
  } else {
    // start from Sr1 / Sc1
    if (Sr != Sr1) { hssk_rowgather_desc cp{Sr1, Sr, nullptr, n1, c, (int)ldsr1, (int)ldsr, 0, 0}; ck(hssk_gather_rows(ctx_, &cp, 1)); }
    if (Sc != Sc1) { hssk_rowgather_desc cp{Sc1, Sc, nullptr, n1, c, (int)ldsc1, (int)ldsc, 0, 0}; ck(hssk_gather_rows(ctx_, &cp, 1)); }
  }
  double* V0tR0 = wk.dbl((size_t)L(d.rV0) * c);
  double* U0tR0 = wk.dbl((size_t)L(d.rU0) * c);
  double* V1tR1 = wk.dbl((size_t)L(d.rV1) * c);
  double* U1tR1 = wk.dbl((size_t)L(d.rU1) * c);
  basis_up(root.c0, false, dR0, l0, c, V0tR0, L(d.rV0), wk);
  basis_up(root.c0, true, dR0, l0, c, U0tR0, L(d.rU0), wk);
  basis_up(root.c1, false, dR1, l1, c, V1tR1, L(d.rV1), wk);
  basis_up(root.c1, true, dR1, l1, c, U1tR1, L(d.rU1), wk);
  // P = -(B10 V0big^T R0 + W V1big^T R1)  (rU1 x c) ; Q = -(B01^T U0big^T R0 + W^T U1big^T R1)  (rV1 x c)
  double* P = wk.dbl((size_t)L(d.rU1) * c);
  double* Q = wk.dbl((size_t)L(d.rV1) * c);
  std::vector<hssk_gemm_desc> g(2);
  g[0] = hssk_gemm_desc{root.B10, V0tR0, P, d.rU1, c, d.rV0, L(d.rU1), L(d.rV0), L(d.rU1), 0, 0, -1.0, 0.0};
  g[1] = hssk_gemm_desc{root.B01, U0tR0, Q, d.rV1, c, d.rU0, L(d.rU0), L(d.rU0), L(d.rV1), 1, 0, -1.0, 0.0};
  ck(hssk_gemm_vbatched(ctx_, g.data(), 2));
  g[0] = hssk_gemm_desc{sW_, V1tR1, P, d.rU1, c, d.rV1, L(d.rU1), L(d.rV1), L(d.rU1), 0, 0, -1.0, 1.0};
  g[1] = hssk_gemm_desc{sW_, U1tR1, Q, d.rV1, c, d.rU1, L(d.rU1), L(d.rU1), L(d.rV1), 1, 0, -1.0, 1.0};
  ck(hssk_gemm_vbatched(ctx_, g.data(), 2));
  // Sr = Sr1 + U1big P ; Sc = Sc1 + V1big Q                           (Schur.hpp:213-218)
  double* E = wk.dbl((size_t)L(n1) * c);
  basis_down(root.c1, true, P, L(d.rU1), c, E, L(n1), wk);
  { hssk_rowgather_desc ad{E, dSr, nullptr, n1, c, L(n1), (int)lsr, 0, 1}; ck(hssk_gather_rows(ctx_, &ad, 1)); }
  basis_down(root.c1, false, Q, L(d.rV1), c, E, L(n1), wk);
  { hssk_rowgather_desc ad{E, dSc, nullptr, n1, c, L(n1), (int)lsc, 0, 1}; ck(hssk_gather_rows(ctx_, &ad, 1)); }
  if (!on_device) {
    ck(hssk_memcpy2d_d2h(ctx_, Sr, sizeof(double) * ldsr, dSr, sizeof(double) * n1, sizeof(double) * n1, c));
    ck(hssk_memcpy2d_d2h(ctx_, Sc, sizeof(double) * ldsc, dSc, sizeof(double) * n1, sizeof(double) * n1, c));
  }
  ck(hssk_sync(ctx_));
}

// ---------------------------------------------------------------------------------------------
// ULV solve (HSSMatrix.solve.hpp:69-238)
// ---------------------------------------------------------------------------------------------
void DeviceHSS::solve(int nrhs, double* b, long long ldb, bool on_device) {
  OpGuard op_guard(op_mu_);
  ensure_ready("solve");
  if (!factored_) throw std::logic_error("solve: factor() has not been called (or shift() invalidated the factors)");
  if (nrhs <= 0 || n_ == 0) return;
  double t0 = now();
  // repeated solve on the same device buffer: replay the recorded sweep (no descriptor building, no staging)
  const bool plannable = on_device && o_.world == 1 && plans_enabled();
  const PlanKey key{1, 'N', nrhs, (const void*)b, (void*)b, ldb, ldb, 0.};
  if (plannable) {
    auto it = plans_.find(key);
    if (it != plans_.end() && it->second.plan) {
      ck(hssk_plan_replay(ctx_, it->second.plan));
      ck(hssk_sync(ctx_));
      if (hssk_sweep_status(ctx_)) throw std::runtime_error(std::string("solve: ") + hssk_last_error());
      stats_.t_solve = now() - t0;
      return;
    }
  }
  hssk_plan* rec = nullptr;
  // the first call on a buffer runs normally; the second one is recorded while it runs, later ones replay
  if (plannable && plans_.size() > 32) drop_plans();   // many different buffers: start over rather than grow
  if (plannable && ++plans_[key].seen == 2) ck(hssk_plan_begin(ctx_, &rec));
  struct EndRec { hssk_ctx* c; hssk_plan* p; bool done = false; ~EndRec() { if (p && !done) { hssk_plan_end(c); hssk_plan_destroy(p); } } } guard{ctx_, rec};
  Arena& tmp = rec ? *plan_arena_ : *tmp_;   // a recorded sweep keeps its own work vectors
  if (!rec) tmp.rewind();
  const int N = n_;
  double* db = b;
  long long lb = ldb;
  if (!on_device) {
    db = tmp.dbl((size_t)N * nrhs);
    ck(hssk_memcpy2d_h2d(ctx_, db, sizeof(double) * N, b, sizeof(double) * ldb, sizeof(double) * N, nrhs));
    lb = N;
  }
  if (lb > 0x7fffffffLL) throw std::invalid_argument("solve: leading dimension too large");
  const size_t nn = nodes_.size();
  // f: assembled right-hand side of an inner node (mU rows; children write ft1 into it);
  // y: (mU - rU) rows; zc: children's z stacked (mV rows); xb: solution in the node's basis (mU rows)
  std::vector<double*> f(nn, nullptr), y(nn, nullptr), zc(nn, nullptr), xb(nn, nullptr);
  // (f, zc, xb are handed from node to node: carved from one block that the single-launch sweeps arm with a sentinel)
  size_t hand_total = 0;
  for (size_t i = 0; i < nn; i++) {
    if (!mine((int)i) || nodes_[i].leaf()) continue;
    const Node& nd = nodes_[i];
    const int mu = nodes_[nd.c0].rU + nodes_[nd.c1].rU, mv = nodes_[nd.c0].rV + nodes_[nd.c1].rV;
    hand_total += (size_t)(2 * std::max(mu, 1) + std::max(mv, 1)) * nrhs;
  }
  double* hand = tmp.dbl(std::max<size_t>(hand_total, 1));
  {
    size_t off = 0;
    for (size_t i = 0; i < nn; i++) {
      if (!mine((int)i)) continue;
      const Node& nd = nodes_[i];
      if (!nd.leaf()) {
        const int mu = nodes_[nd.c0].rU + nodes_[nd.c1].rU, mv = nodes_[nd.c0].rV + nodes_[nd.c1].rV;
        f[i] = hand + off; off += (size_t)std::max(mu, 1) * nrhs;
        zc[i] = hand + off; off += (size_t)std::max(mv, 1) * nrhs;
        xb[i] = hand + off; off += (size_t)std::max(mu, 1) * nrhs;
      }
      if (nd.lvl > 0 && nd.mU > nd.rU) y[i] = tmp.dbl((size_t)(nd.mU - nd.rU) * nrhs);
    }
  }
  // few right-hand sides: the whole forward sweep (root solve included) and the whole backward sweep are ONE launch each
  // (hssk_ulv_fwd_sweep / _bwd_sweep: a workgroup per node, dependency flags between them) instead of 7 / 3 batched
  // launches per level
  static const bool no_fuse = [] { const char* e = std::getenv("STRUMPACK_AMD_NO_FUSED_SOLVE"); return e && e[0] == '1'; }();
  const bool fuse = nrhs <= 64 && !no_fuse;   // (more right-hand sides: the batched MFMA launches per level)
  if (fuse) ck(hssk_sweep_arm(ctx_, hand, (long long)hand_total));
  typedef std::vector<std::vector<int>> Levels;
  auto fwd_sweep = [&](const Levels& levels) -> bool {
    std::vector<hssk_sweep_fwd_desc> fd;
    std::vector<int> where(nn, -1);
    for (auto& ids : levels)
      for (int id : ids) {
        const Node& nd = nodes_[id];
        hssk_sweep_fwd_desc d{};
        d.wait0 = d.wait1 = -1;
        d.mv = nd.leaf() ? nd.m : nodes_[nd.c0].rV + nodes_[nd.c1].rV;
        if (nd.leaf()) { d.fsrc = db + nd.lo; d.ldf = (int)lb; }
        else {
          const Node &a = nodes_[nd.c0], &c = nodes_[nd.c1];
          d.fsrc = f[id]; d.ldf = std::max(a.rU + c.rU, 1);
          d.B01 = nd.B01; d.B10 = nd.B10; d.zc = zc[id];
          d.rU0 = a.rU; d.rU1 = c.rU; d.rV0 = a.rV; d.rV1 = c.rV; d.ldz_in = std::max(a.rV + c.rV, 1);
          d.permV = nd.permV; d.XV = nd.XV;
          if (!d.B01 || !d.B10) return false;
          d.wait0 = where[nd.c0]; d.wait1 = where[nd.c1];
        }
        if (nd.lvl == 0) {
          // root: x = LU^{-1} f
          d.m = nd.leaf() ? nd.m : nodes_[nd.c0].rU + nodes_[nd.c1].rU;
          if (d.m == 0) continue;
          d.LU = nd.LU; d.piv = nd.piv; d.TinvL = nd.Tinv; d.TinvU = nd.TinvU;
          if (!d.LU || !d.TinvL || !d.TinvU) return false;
          d.xroot = nd.leaf() ? db + nd.lo : xb[id];
          d.ldxr = nd.leaf() ? (int)lb : std::max(d.m, 1);
        } else {
          const Node& pa = nodes_[nd.parent];
          d.m = nd.mU; d.r = nd.rU; d.rv = nd.rV;
          if (!nd.leaf()) d.mv = nd.mV;
          d.permU = nd.permU; d.XU = nd.XU; d.Rlq = nd.Rlq; d.Tinv = nd.Tinv; d.WQ = nd.WQ; d.Vt0T = nd.Vt0T;
          d.ft1 = f[nd.parent] + (id == pa.c0 ? 0 : nodes_[pa.c0].rU);
          d.ldp = std::max(nodes_[pa.c0].rU + nodes_[pa.c1].rU, 1);
          d.z = zc[nd.parent] + (id == pa.c0 ? 0 : nodes_[pa.c0].rV);
          d.ldz = std::max(nodes_[pa.c0].rV + nodes_[pa.c1].rV, 1);
          d.y = y[id];
          if (d.m > d.r && (!d.y || !d.Rlq || !d.Tinv || (d.r && !d.WQ) || (d.rv && !d.Vt0T))) return false;
        }
        where[id] = (int)fd.size();
        fd.push_back(d);
      }
    if (fd.empty()) return true;
    const int rc = hssk_ulv_fwd_sweep(ctx_, fd.data(), (int)fd.size(), nrhs);
    if (rc == 2) return false;
    ck(rc);
    return true;
  };
  auto bwd_sweep = [&](const Levels& levels) -> bool {
    std::vector<hssk_sweep_bwd_desc> bd;
    std::vector<int> where(nn, -1);
    for (auto& ids : levels)
      for (int id : ids) {
        const Node& nd = nodes_[id];
        if (nd.leaf()) continue;
        const Node& a = nodes_[nd.c0];
        const int cid[2] = {nd.c0, nd.c1};
        for (int q = 0; q < 2; q++) {
          if (!mine(cid[q])) continue;
          const Node& cn = nodes_[cid[q]];
          if (cn.mU == 0) continue;
          hssk_sweep_bwd_desc d{};
          d.Qt = cn.Qt; d.y = y[cid[q]]; d.xpart = xb[id] + (q ? a.rU : 0);
          d.out = cn.leaf() ? db + cn.lo : xb[cid[q]];
          d.m = cn.mU; d.r = cn.rU; d.ldx = std::max(a.rU + nodes_[nd.c1].rU, 1); d.ldo = cn.leaf() ? (int)lb : std::max(cn.mU, 1);
          d.wait0 = where[id];
          if (d.m > d.r && (!d.Qt || !d.y)) return false;
          where[cid[q]] = (int)bd.size();
          bd.push_back(d);
        }
      }
    if (bd.empty()) return true;
    const int rc = hssk_ulv_bwd_sweep(ctx_, bd.data(), (int)bd.size(), nrhs);
    if (rc == 2) return false;
    ck(rc);
    return true;
  };
  // ---- forward, one tree height
  auto fwd = [&](const std::vector<int>& ids) {
    if (ids.empty()) return;
    std::vector<hssk_gemm_desc> ga, gb, gc, gd, ge;
    std::vector<hssk_rowgather_desc> rg;
    std::vector<hssk_trsm_desc> ts;
    std::vector<hssk_lusolve_desc> ls;
    for (int id : ids) {
      const Node& nd = nodes_[id];
      if (nd.leaf()) continue;
      const Node &a = nodes_[nd.c0], &c = nodes_[nd.c1];
      const int ldf = std::max(a.rU + c.rU, 1), lz = std::max(a.rV + c.rV, 1);
      // f0 = ft1_0 - B01 z_1 ; f1 = ft1_1 - B10 z_0   (solve.hpp:88-99).  The children already wrote
      // ft1 - W1 (Q0^T y) into f (the -W1 Q0^T y term of solve.hpp:100-131 only needs child data).
      ga.push_back(hssk_gemm_desc{nd.B01, zc[id] + a.rV, f[id], a.rU, nrhs, c.rV, std::max(a.rU, 1), lz, ldf, 0, 0, -1.0, 1.0});
      ga.push_back(hssk_gemm_desc{nd.B10, zc[id], f[id] + a.rU, c.rU, nrhs, a.rV, std::max(c.rU, 1), lz, ldf, 0, 0, -1.0, 1.0});
    }
    if (!ga.empty()) ck(hssk_gemm_vbatched(ctx_, ga.data(), (int)ga.size()));
    for (int id : ids) {
      const Node& nd = nodes_[id];
      const double* fsrc = nd.leaf() ? db + nd.lo : f[id];
      const int mu = nd.leaf() ? nd.m : nodes_[nd.c0].rU + nodes_[nd.c1].rU;
      const int ldf = nd.leaf() ? (int)lb : std::max(mu, 1);
      if (nd.lvl == 0) {
        // x = LU^{-1} f (solve.hpp:133-135)
        if (nd.leaf()) { if (mu) ls.push_back(hssk_lusolve_desc{nd.LU, nd.piv, db + nd.lo, mu, nrhs, mu, (int)lb}); }
        else {
          rg.push_back(hssk_rowgather_desc{f[id], xb[id], nullptr, mu, nrhs, ldf, std::max(mu, 1), 0, 0});
          if (mu) ls.push_back(hssk_lusolve_desc{nd.LU, nd.piv, xb[id], mu, nrhs, mu, std::max(mu, 1)});
        }
        continue;
      }
      const Node& pa = nodes_[nd.parent];
      const int m = nd.mU, r = nd.rU;
      double* ft1 = f[nd.parent] + (id == pa.c0 ? 0 : nodes_[pa.c0].rU);
      const int ldp = std::max(nodes_[pa.c0].rU + nodes_[pa.c1].rU, 1);
      double* z = zc[nd.parent] + (id == pa.c0 ? 0 : nodes_[pa.c0].rV);
      const int ldz = std::max(nodes_[pa.c0].rV + nodes_[pa.c1].rV, 1);
      // f <- P^T f ; ft1 = f(0:r) ; y = L^{-1} (f(r:) - E ft1)    (solve.hpp:153-163)
      if (r) rg.push_back(hssk_rowgather_desc{fsrc, ft1, nd.permU, r, nrhs, ldf, ldp, 0, 0});
      if (m > r) {
        rg.push_back(hssk_rowgather_desc{fsrc, y[id], nd.permU + r, m - r, nrhs, ldf, m - r, 0, 0});
        if (r) gd.push_back(hssk_gemm_desc{nd.XU, ft1, y[id], m - r, nrhs, r, r, ldp, m - r, 1, 0, -1.0, 1.0});
        ts.push_back(hssk_trsm_desc{nd.Rlq, y[id], m - r, nrhs, m, m - r, 0, 1, 0});
        if (r) {
          // ft1 -= W1 (Q0^T y),  Q0^T y = Q~(:, :m-r) y
          double* t = tmp.dbl((size_t)m * nrhs);
          gb.push_back(hssk_gemm_desc{nd.Qt, y[id], t, m, nrhs, m - r, m, m - r, m, 0, 0, 1.0, 0.0});
          gc.push_back(hssk_gemm_desc{nd.W1, t, ft1, r, nrhs, m, r, m, ldp, 0, 0, -1.0, 1.0});
        }
      }
      // z = V^H [z0; z1] + Vt0^H y   (leaf: z = Vt0^H y)          (solve.hpp:164-192)
      const int rv = nd.rV;
      if (rv) {
        double zbeta = 0.0;
        if (!nd.leaf()) {
          const int mv = nd.mV;
          rg.push_back(hssk_rowgather_desc{zc[id], z, nd.permV, rv, nrhs, std::max(mv, 1), ldz, 0, 0});
          if (mv > rv) {
            double* t = tmp.dbl((size_t)(mv - rv) * nrhs);
            rg.push_back(hssk_rowgather_desc{zc[id], t, nd.permV + rv, mv - rv, nrhs, std::max(mv, 1), mv - rv, 0, 0});
            gd.push_back(hssk_gemm_desc{nd.XV, t, z, rv, nrhs, mv - rv, rv, mv - rv, ldz, 0, 0, 1.0, 1.0});
          }
          zbeta = 1.0;
        }
        if (m > r && rv > 0) ge.push_back(hssk_gemm_desc{nd.Vt0T, y[id], z, rv, nrhs, m - r, rv, m - r, ldz, 0, 0, 1.0, zbeta});   // z (+)= Vt0^T y
        else if (nd.leaf()) ge.push_back(hssk_gemm_desc{z, z, z, rv, nrhs, 0, 1, 1, ldz, 0, 0, 1.0, 0.0});  // z = 0
      }
    }
    if (!rg.empty()) ck(hssk_gather_rows(ctx_, rg.data(), (int)rg.size()));
    if (!gd.empty()) ck(hssk_gemm_vbatched(ctx_, gd.data(), (int)gd.size()));
    if (!ts.empty()) ck(hssk_trsm_vbatched(ctx_, ts.data(), (int)ts.size()));
    if (!ge.empty()) ck(hssk_gemm_vbatched(ctx_, ge.data(), (int)ge.size()));
    if (!gb.empty()) ck(hssk_gemm_vbatched(ctx_, gb.data(), (int)gb.size()));
    if (!gc.empty()) ck(hssk_gemm_vbatched(ctx_, gc.data(), (int)gc.size()));
    if (!ls.empty()) ck(hssk_getrs_vbatched(ctx_, ls.data(), (int)ls.size()));
  };
  // ---- backward, one depth (solve.hpp:199-238): x_c = Q_c^H [y_c ; x(part)] = Q~(:, :mc-rc) y_c + Q~(:, mc-rc:) xpart
  auto bwd = [&](const std::vector<int>& ids) {
    std::vector<hssk_gemm_desc> g1, g2;
    std::vector<hssk_rowgather_desc> cp;
    for (int id : ids) {
      const Node& nd = nodes_[id];
      if (nd.leaf()) continue;
      const Node &a = nodes_[nd.c0], &c = nodes_[nd.c1];
      const int mu = a.rU + c.rU;
      const double* x = xb[id];
      const int ldx = std::max(mu, 1);
      const Node* ch[2] = {&a, &c};
      const int cid[2] = {nd.c0, nd.c1};
      for (int q = 0; q < 2; q++) {
        if (!mine(cid[q])) continue;  // the other ranks' subtrees continue on their owners
        const Node& cn = *ch[q];
        const int mc = cn.mU, rc = cn.rU;
        const double* xpart = x + (q ? a.rU : 0);
        double* out = cn.leaf() ? db + cn.lo : xb[cid[q]];
        const int ldo = cn.leaf() ? (int)lb : std::max(mc, 1);
        if (mc > rc) {
          g1.push_back(hssk_gemm_desc{cn.Qt, y[cid[q]], out, mc, nrhs, mc - rc, mc, mc - rc, ldo, 0, 0, 1.0, 0.0});
          if (rc) g2.push_back(hssk_gemm_desc{cn.Qt + (size_t)(mc - rc) * mc, xpart, out, mc, nrhs, rc, mc, ldx, ldo, 0, 0, 1.0, 1.0});
        } else if (mc) {
          cp.push_back(hssk_rowgather_desc{xpart, out, nullptr, mc, nrhs, ldx, ldo, 0, 0});
        }
      }
    }
    if (!g1.empty()) ck(hssk_gemm_vbatched(ctx_, g1.data(), (int)g1.size()));
    if (!g2.empty()) ck(hssk_gemm_vbatched(ctx_, g2.data(), (int)g2.size()));
    if (!cp.empty()) ck(hssk_gather_rows(ctx_, cp.data(), (int)cp.size()));
  };
  if (!(fuse && fwd_sweep(own_by_height_)))
    for (auto& ids : own_by_height_) fwd(ids);
  if (dist_subtree_) {
    // publish ft1' (rU x nrhs) and z (rV x nrhs) of the cut nodes into every rank's top buffers
    const int G = o_.world, me = o_.rank;
    int ru = 1, rv = 1;
    for (int g = 0; g < G; g++) { ru = std::max(ru, nodes_[cut_nodes_[g]].rU); rv = std::max(rv, nodes_[cut_nodes_[g]].rV); }
    const size_t blk = (size_t)(ru + rv) * nrhs;
    double* buf = tmp.dbl(blk * G);
    auto slices = [&](int g, double*& pf, int& ldf, double*& pz, int& ldz) {
      const int id = cut_nodes_[g];
      const Node& pa = nodes_[nodes_[id].parent];
      pf = f[nodes_[id].parent] + (id == pa.c0 ? 0 : nodes_[pa.c0].rU);
      ldf = std::max(nodes_[pa.c0].rU + nodes_[pa.c1].rU, 1);
      pz = zc[nodes_[id].parent] + (id == pa.c0 ? 0 : nodes_[pa.c0].rV);
      ldz = std::max(nodes_[pa.c0].rV + nodes_[pa.c1].rV, 1);
    };
    {
      double *pf, *pz; int ldf, ldz;
      slices(me, pf, ldf, pz, ldz);
      const Node& c = nodes_[cut_nodes_[me]];
      std::vector<hssk_rowgather_desc> pk;
      if (c.rU) pk.push_back(hssk_rowgather_desc{pf, buf + blk * me, nullptr, c.rU, nrhs, ldf, ru, 0, 0});
      if (c.rV) pk.push_back(hssk_rowgather_desc{pz, buf + blk * me + (size_t)ru * nrhs, nullptr, c.rV, nrhs, ldz, rv, 0, 0});
      if (!pk.empty()) ck(hssk_gather_rows(ctx_, pk.data(), (int)pk.size()));
    }
    comm(buf, (long long)(sizeof(double) * blk));
    std::vector<hssk_rowgather_desc> up;
    for (int g = 0; g < G; g++) {
      if (g == me) continue;
      double *pf, *pz; int ldf, ldz;
      slices(g, pf, ldf, pz, ldz);
      const Node& c = nodes_[cut_nodes_[g]];
      if (c.rU) up.push_back(hssk_rowgather_desc{buf + blk * g, pf, nullptr, c.rU, nrhs, ru, ldf, 0, 0});
      if (c.rV) up.push_back(hssk_rowgather_desc{buf + blk * g + (size_t)ru * nrhs, pz, nullptr, c.rV, nrhs, rv, ldz, 0, 0});
    }
    if (!up.empty()) ck(hssk_gather_rows(ctx_, up.data(), (int)up.size()));
    if (!(fuse && fwd_sweep(top_by_height_)))
      for (auto& ids : top_by_height_) fwd(ids);
    if (!(fuse && bwd_sweep(top_by_depth_)))
      for (auto& ids : top_by_depth_) bwd(ids);
  }
  if (!(fuse && bwd_sweep(own_by_depth_)))
    for (auto& ids : own_by_depth_) bwd(ids);
  if (dist_subtree_) allgather_rows(db, lb, nrhs);
  if (!on_device) ck(hssk_memcpy2d_d2h(ctx_, b, sizeof(double) * ldb, db, sizeof(double) * N, sizeof(double) * N, nrhs));
  if (rec) { ck(hssk_plan_end(ctx_)); guard.done = true; plans_[key].plan = rec; }
  ck(hssk_sync(ctx_));
  if (hssk_sweep_status(ctx_)) throw std::runtime_error(std::string("solve: ") + hssk_last_error());
  stats_.t_solve = now() - t0;
  {
    double fs = 0;
    for (auto& nd : nodes_) {
      if (nd.lvl == 0) { const int mu = nd.leaf() ? nd.m : nodes_[nd.c0].rU + nodes_[nd.c1].rU; fs += 2.0 * mu * (double)mu; continue; }
      const double m = nd.mU, r = nd.rU, k = m - r, rv = nd.rV;
      fs += 2.0 * k * r + k * k + 2.0 * k * rv + 2.0 * m * m + 2.0 * k * m;
      if (!nd.leaf()) fs += 4.0 * nodes_[nd.c0].rU * (double)nodes_[nd.c1].rV;
    }
    stats_.f_solve = fs * nrhs;
    // blocks read by the sweeps (fused path: X, the off-diagonal part of R~ + its inverted diagonal blocks, WQ, Vt0, B, XV
    // going up, Q~ going down; the unfused path reads W1 and Q~(:, 0:q) instead of WQ)
    double bs = 0;
    for (auto& nd : nodes_) {
      if (nd.lvl == 0) { const double mu = nd.leaf() ? nd.m : nodes_[nd.c0].rU + nodes_[nd.c1].rU; bs += mu * mu; continue; }
      const double m = nd.mU, r = nd.rU, k = m - r, rv = nd.rV;
      bs += r * k + k * (k + 1) / 2 + (fuse ? r * k : r * m + m * k) + k * rv + m * m;   // (per group of four right-hand sides when fused)
      if (!nd.leaf()) bs += (double)nodes_[nd.c0].rU * nodes_[nd.c1].rV + (double)nodes_[nd.c1].rU * nodes_[nd.c0].rV + rv * (nd.mV - rv);
    }
    stats_.b_solve = 8.0 * bs;
  }
}

}  // namespace HSS
}  // namespace strumpack
