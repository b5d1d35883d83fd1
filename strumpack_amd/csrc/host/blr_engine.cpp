// DeviceBLR implementation (see blr_engine.hpp for the reference behaviour it follows).
#include "blr_engine.hpp"
#include "DevicePool.hpp"

#include <random>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <string>

namespace strumpack {
namespace BLR {

namespace {
inline void ck(int rc) {
  if (rc) throw std::runtime_error(std::string("hssk: ") + hssk_last_error());
}
inline double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

// bump allocator over device chunks from the process-wide pool (DevicePool.hpp); rewind() keeps the chunks
class Arena2 {
 public:
  explicit Arena2(size_t chunk) : chunk_(chunk) {}
  ~Arena2() { for (auto& c : chunks_) DevicePool::get().release(c.first, c.second); }
  void* alloc(size_t bytes) {
    bytes = (std::max<size_t>(bytes, 8) + 255) & ~size_t(255);
    while (bytes > left_) {
      if (next_ < chunks_.size()) { cur_ = (char*)chunks_[next_].first; left_ = chunks_[next_].second; next_++; continue; }
      const size_t gran = size_t(64) << 20;   // (sizes on a grid: the next factorization of a front this size finds them in the pool)
      const size_t c = (std::max(chunk_, bytes) + gran - 1) / gran * gran;
      void* p = DevicePool::get().acquire(c);
      if (!p) throw std::runtime_error(std::string("device allocation failed: ") + hssk_last_error());
      chunks_.emplace_back(p, c);
      next_ = chunks_.size();
      cur_ = (char*)p; left_ = c;
    }
    void* r = cur_;
    cur_ += bytes; left_ -= bytes; used_ += bytes;
    return r;
  }
  double* dbl(size_t n) { return (double*)alloc(sizeof(double) * std::max<size_t>(n, 1)); }
  int* ints(size_t n) { return (int*)alloc(sizeof(int) * std::max<size_t>(n, 1)); }
  void rewind() { next_ = 0; cur_ = nullptr; left_ = 0; used_ = 0; }
  size_t used() const { return used_; }

 private:
  size_t chunk_, left_ = 0, used_ = 0, next_ = 0;
  char* cur_ = nullptr;
  std::vector<std::pair<void*, size_t>> chunks_;
};

DeviceBLR::DeviceBLR(int m, const std::vector<int>& rowtiles, int n, const std::vector<int>& coltiles, const BLREngineOptions& o)
    : m_(m), n_(n), o_(o) {
  roff_.assign(1, 0);
  for (int t : rowtiles) roff_.push_back(roff_.back() + t);
  coff_.assign(1, 0);
  for (int t : coltiles) coff_.push_back(coff_.back() + t);
  if (roff_.back() != m || coff_.back() != n) throw std::invalid_argument("BLR: the tile sizes do not add up to the matrix dimensions");
  ck(hssk_ctx_create(&ctx_, o_.device));
  store_.reset(new Arena2(size_t(256) << 20));
  tmp_.reset(new Arena2(size_t(256) << 20));
  blk_.reset(new Arena2(size_t(256) << 20));
  tiles_.assign((size_t)rowblocks() * colblocks(), Tile());
}

DeviceBLR::~DeviceBLR() {
  if (ctx2_) { hssk_sync(ctx2_); hssk_ctx_destroy(ctx2_); }
  if (ctx_) hssk_sync(ctx_);
  store_.reset();
  tmp_.reset();
  blk_.reset();
  free_array();
  hssk_ctx_destroy(ctx_);
}

void DeviceBLR::free_array() {
  if (dA_) DevicePool::get().release(dA_, dA_bytes_);
  dA_ = nullptr;
  dA_bytes_ = 0;
}

void DeviceBLR::alloc_array() {
  free_array();
  ld_ = std::max(m_, 1);
  const size_t gran = size_t(64) << 20;
  dA_bytes_ = (sizeof(double) * (size_t)ld_ * std::max(n_, 1) + gran - 1) / gran * gran;
  dA_ = (double*)DevicePool::get().acquire(dA_bytes_);
  if (!dA_) throw std::runtime_error("BLR: device allocation of the operand failed");
  store_->rewind();
  sweep_tab_[0] = sweep_tab_[1] = SweepTables();
  tiles_.assign(tiles_.size(), Tile());
  dpiv_ = store_->ints((size_t)std::max(m_, 1) + rowblocks() + 1);
  compressed_ = factored_ = false;
  nsteps_ = 0;
}

// rows x cols block of the working array at (r0, c0) from a host or device source (null: zeros)
void DeviceBLR::put_block(int r0, int c0, int rows, int cols, const double* src, long long lds, bool on_device) {
  if (rows <= 0 || cols <= 0) return;
  double* dst = dA_ + r0 + (size_t)c0 * ld_;
  if (!src) {
    if (rows == ld_) { ck(hssk_memset_zero(ctx_, dst, (long long)sizeof(double) * ld_ * cols)); return; }
    hssk_gemm_desc z{dst, dst, dst, rows, cols, 0, (int)ld_, (int)ld_, (int)ld_, 0, 0, 0.0, 0.0};   // C = 0 * C
    ck(hssk_gemm_vbatched(ctx_, &z, 1));
    return;
  }
  if (lds < rows) throw std::invalid_argument("BLR: leading dimension smaller than the block");
  if (on_device) {
    if (lds > 0x7fffffffLL) throw std::invalid_argument("BLR: leading dimension too large");
    hssk_colgather_desc cp{src, dst, nullptr, rows, cols, (int)lds, (int)ld_, 0};
    ck(hssk_gather_cols(ctx_, &cp, 1));
  } else {
    ck(hssk_h2d_block_async(ctx_, dst, ld_, src, lds, rows, cols));
  }
}

void DeviceBLR::load(const double* A, long long lda, bool on_device) {
  alloc_array();
  put_block(0, 0, m_, n_, A, lda, on_device);
  // (hssk_h2d_block_async runs on the copy stream; the fence orders the kernels that follow behind it)
  if (!on_device) ck(hssk_copy_fence(ctx_));
}

// Truncated RRQR of the listed tiles (batched).  A tile whose rank does not pay (r (m + n) > m n, BLRMatrix.cpp:568) or
// that is not admissible is kept dense -- stored as U = T P, V = P (r = n) so that every later step treats all tiles alike.
void DeviceBLR::compress_tiles(const std::vector<std::pair<int, int>>& ij, const char* adm) {
  if (ij.empty()) return;
  const size_t cnt = ij.size();
  Arena2& tmp = *tmp_;
  std::vector<hssk_colgather_desc> cp;
  std::vector<hssk_id_desc> idd;
  std::vector<double*> W(cnt, nullptr);
  std::vector<int*> perm(cnt, nullptr);
  int nmax = 0;
  for (auto& t : ij) nmax = std::max(nmax, tn(t.second));
  int* ranks = tmp.ints(cnt);
  // identity permutation for the tiles that are not compressed
  std::vector<int> iota(nmax);
  std::iota(iota.begin(), iota.end(), 0);
  int* diota = tmp.ints(nmax);
  ck(hssk_upload_async(ctx_, diota, iota.data(), (long long)sizeof(int) * nmax));
  std::vector<char> want(cnt, 0);
  for (size_t k = 0; k < cnt; k++) {
    const int i = ij[k].first, j = ij[k].second, m = tm(i), n = tn(j);
    want[k] = (!adm || adm[(size_t)i + (size_t)j * rowblocks()]) && m > 0 && n > 0;
    if (!want[k]) continue;
    W[k] = tmp.dbl((size_t)m * n);
    perm[k] = tmp.ints(n);
    cp.push_back(hssk_colgather_desc{blk(i, j), W[k], nullptr, m, n, (int)ld_, m, 0});
    // a tile is kept dense as soon as r (m + n) > m n (BLRMatrix.cpp:568): the factorization may stop one step beyond that rank
    const int rpay = (int)std::min<long long>(((long long)m * n) / (m + n) + 1, (long long)o_.max_rank);
    idd.push_back(hssk_id_desc{W[k], m, m, n, o_.rel_tol, o_.abs_tol, rpay, perm[k], ranks + k, tmp.dbl(3 * (size_t)n)});
  }
  std::vector<int> hr(cnt, 0);
  if (o_.lr_algo == 1) {
    // adaptive cross approximation (BLR/LRTile.cpp:66-74 -> dense/ACA.cpp:41-118): U and V straight from rows and columns of
    // the tile in the array; the factors that pay (rank (m + n) <= m n) are copied to their place
    std::vector<hssk_aca_desc> ad;
    std::vector<double*> Ua(cnt, nullptr), Va(cnt, nullptr);
    std::vector<int> rcap(cnt, 0);
    for (size_t k = 0; k < cnt; k++) {
      if (!want[k]) continue;
      const int i = ij[k].first, j = ij[k].second, m = tm(i), n = tn(j);
      const int rpay = (int)std::min<long long>(std::min<long long>(((long long)m * n) / (m + n) + 1, (long long)o_.max_rank), std::min(m, n));
      rcap[k] = std::max(rpay, 1);
      Ua[k] = tmp.dbl((size_t)m * rcap[k]);
      Va[k] = tmp.dbl((size_t)n * rcap[k]);
      // the reference's first row: a default-seeded std::mt19937 drawn on [0, m) (ACA.cpp:55-57)
      std::mt19937 mt;
      std::uniform_int_distribution<int> rgen(0, m - 1);
      ad.push_back(hssk_aca_desc{blk(i, j), (int)ld_, m, n, o_.rel_tol, o_.abs_tol, rcap[k], rgen(mt), Ua[k], m, Va[k], n, ranks + k});
    }
    if (!ad.empty()) {
      ck(hssk_aca_vbatched(ctx_, ad.data(), (int)ad.size()));
      ck(hssk_memcpy_d2h(ctx_, hr.data(), ranks, (long long)sizeof(int) * cnt));
    }
    std::vector<hssk_colgather_desc> cu;
    for (size_t k = 0; k < cnt; k++) {
      const int i = ij[k].first, j = ij[k].second, m = tm(i), n = tn(j);
      Tile& t = tile(i, j);
      const int r = want[k] ? hr[k] : n;
      t.lowrank = want[k] && (long long)r * (m + n) <= (long long)m * n;
      t.r = t.lowrank ? r : n;
      if (!t.U) t.U = store_->dbl((size_t)m * t.r);
      t.V = store_->dbl((size_t)n * t.r);
      if (t.lowrank) {
        if (t.r > 0 && m > 0) cu.push_back(hssk_colgather_desc{Ua[k], t.U, nullptr, m, t.r, m, m, 0});
        if (t.r > 0 && n > 0) cu.push_back(hssk_colgather_desc{Va[k], t.V, nullptr, n, t.r, n, n, 0});
      } else {   // kept dense: U = the tile, V = I
        if (m > 0 && n > 0) cu.push_back(hssk_colgather_desc{blk(i, j), t.U, nullptr, m, n, (int)ld_, m, 0});
      }
    }
    if (!cu.empty()) ck(hssk_gather_cols(ctx_, cu.data(), (int)cu.size()));
    std::vector<hssk_basis_desc> eye;
    for (size_t k = 0; k < cnt; k++) {
      const int i = ij[k].first, j = ij[k].second, n = tn(j);
      Tile& t = tile(i, j);
      if (!t.lowrank && t.r > 0 && n > 0) eye.push_back(hssk_basis_desc{nullptr, diota, t.V, n, t.r, tm(i), n});
    }
    if (!eye.empty()) ck(hssk_basis_dense(ctx_, eye.data(), (int)eye.size()));
    return;
  }
  if (!idd.empty()) {
    ck(hssk_gather_cols(ctx_, cp.data(), (int)cp.size()));
    ck(hssk_id_vbatched(ctx_, idd.data(), (int)idd.size()));
    ck(hssk_memcpy_d2h(ctx_, hr.data(), ranks, (long long)sizeof(int) * cnt));
    // (the cooperative ID's workgroups poll each other with a bounded spin; a timeout must not pass as a rank)
    if (hssk_sweep_status(ctx_)) throw std::runtime_error(std::string("BLR tile compression: ") + hssk_last_error());
  }
  std::vector<hssk_colgather_desc> gu;
  std::vector<hssk_basis_desc> bv;
  for (size_t k = 0; k < cnt; k++) {
    const int i = ij[k].first, j = ij[k].second, m = tm(i), n = tn(j);
    Tile& t = tile(i, j);
    const int r = want[k] ? hr[k] : n;
    t.lowrank = want[k] && (long long)r * (m + n) <= (long long)m * n;
    t.r = t.lowrank ? r : n;
    const int* p = want[k] ? perm[k] : diota;
    if (!t.U) t.U = store_->dbl((size_t)m * t.r);   // (the factorization pre-assigns U inside a block row's panel)
    t.V = store_->dbl((size_t)n * t.r);
    if (t.r > 0 && m > 0) gu.push_back(hssk_colgather_desc{blk(i, j), t.U, p, m, t.r, (int)ld_, m, 0});
    if (t.r > 0 && n > 0) bv.push_back(hssk_basis_desc{want[k] ? W[k] + (size_t)t.r * m : nullptr, p, t.V, n, t.r, m, n});
  }
  if (!gu.empty()) ck(hssk_gather_cols(ctx_, gu.data(), (int)gu.size()));
  if (!bv.empty()) ck(hssk_basis_dense(ctx_, bv.data(), (int)bv.size()));
}

void DeviceBLR::compress_host(const double* A, long long lda, const char* adm) {
  std::lock_guard<std::recursive_mutex> op_guard(op_mu_);   // operations on one matrix take turns
  const double t0 = now();
  load(A, lda, false);
  // (hssk_h2d_block_async runs on the copy stream; the fence in load() orders the kernels below behind it)
  for (int j = 0; j < colblocks(); j++) {   // one block column per batch: bounds the workspace
    tmp_->rewind();
    std::vector<std::pair<int, int>> ij;
    for (int i = 0; i < rowblocks(); i++) ij.emplace_back(i, j);
    compress_tiles(ij, adm);
  }
  ck(hssk_sync(ctx_));
  free_array();
  compressed_ = true;
  t_compress = now() - t0;
}
void DeviceBLR::compress_device(const double* dA, long long lda, const char* adm) {
  std::lock_guard<std::recursive_mutex> op_guard(op_mu_);   // operations on one matrix take turns
  const double t0 = now();
  load(dA, lda, true);
  for (int j = 0; j < colblocks(); j++) {
    tmp_->rewind();
    std::vector<std::pair<int, int>> ij;
    for (int i = 0; i < rowblocks(); i++) ij.emplace_back(i, j);
    compress_tiles(ij, adm);
  }
  ck(hssk_sync(ctx_));
  free_array();
  compressed_ = true;
  t_compress = now() - t0;
}

void DeviceBLR::compress_and_factor_host(const double* A, long long lda, const char* adm) {
  std::lock_guard<std::recursive_mutex> op_guard(op_mu_);   // operations on one matrix take turns
  if (m_ != n_ || roff_ != coff_) throw std::invalid_argument("BLR factorization needs a square matrix with the same clusters for rows and columns");
  load(A, lda, false);
  factor_rl(adm, rowblocks());
}
void DeviceBLR::compress_and_factor_device(const double* dA, long long lda, const char* adm) {
  std::lock_guard<std::recursive_mutex> op_guard(op_mu_);   // operations on one matrix take turns
  if (m_ != n_ || roff_ != coff_) throw std::invalid_argument("BLR factorization needs a square matrix with the same clusters for rows and columns");
  load(dA, lda, true);
  factor_rl(adm, rowblocks());
}

// BLRMatrix::construct_and_partial_factor (BLRMatrix.cpp:740-1037, RL): the front [F11 F12; F21 F22] is ONE working array, so
// the elimination of the separator is the block LU above stopped after sep_blocks steps: the block row of a step runs through
// F11 and F12, its block column through F11 and F21, and the trailing update reaches the rest of F11, F12, F21 and all of F22.
void DeviceBLR::partial_factor(int sep_blocks, const double* F11, long long ld11, const double* F12, long long ld12,
                               const double* F21, long long ld21, const double* F22, long long ld22, const char* adm11,
                               bool on_device) {
  std::lock_guard<std::recursive_mutex> op_guard(op_mu_);   // operations on one matrix take turns
  if (m_ != n_ || roff_ != coff_) throw std::invalid_argument("BLR front: needs the same clusters for rows and columns");
  const int rb = rowblocks();
  if (sep_blocks < 0 || sep_blocks > rb) throw std::invalid_argument("BLR front: number of separator tiles out of range");
  const int ds = roff_[sep_blocks], du = m_ - ds;
  if (ds > 0 && !F11) throw std::invalid_argument("BLR front: no F11");
  alloc_array();
  put_block(0, 0, ds, ds, F11, ld11, on_device);
  put_block(0, ds, ds, du, F12, ld12, on_device);
  put_block(ds, 0, du, ds, F21, ld21, on_device);
  put_block(ds, ds, du, du, F22, ld22, on_device);
  if (!on_device) ck(hssk_copy_fence(ctx_));
  // admissibility of the whole front: F11 as given (all but the diagonal when null), F12 / F21 always (BLRMatrix.cpp:818-845)
  std::vector<char> adm((size_t)rb * rb, 1);
  if (adm11)
    for (int j = 0; j < sep_blocks; j++)
      for (int i = 0; i < sep_blocks; i++) adm[(size_t)i + (size_t)j * rb] = adm11[(size_t)i + (size_t)j * sep_blocks];
  factor_rl(adm.data(), sep_blocks);
}
void DeviceBLR::partial_factor_host(int sep_blocks, const double* F11, long long ld11, const double* F12, long long ld12,
                                    const double* F21, long long ld21, const double* F22, long long ld22, const char* adm11) {
  partial_factor(sep_blocks, F11, ld11, F12, ld12, F21, ld21, F22, ld22, adm11, false);
}
void DeviceBLR::partial_factor_device(int sep_blocks, const double* F11, long long ld11, const double* F12, long long ld12,
                                      const double* F21, long long ld21, const double* F22, long long ld22, const char* adm11) {
  partial_factor(sep_blocks, F11, ld11, F12, ld12, F21, ld21, F22, ld22, adm11, true);
}

void DeviceBLR::schur_host(double* F22, long long ld) const {
  std::lock_guard<std::recursive_mutex> op_guard(op_mu_);
  if (!factored_) throw std::logic_error("BLR front: not factored");
  const int du = upd_rows();
  if (du <= 0) return;
  if (!F22 || ld < du) throw std::invalid_argument("BLR front: bad output for the Schur complement");
  ck(hssk_memcpy2d_d2h(ctx_, F22, sizeof(double) * ld, blk(nsteps_, nsteps_), sizeof(double) * ld_, sizeof(double) * du, du));
}

void DeviceBLR::tile_ranks(int* out) const {
  for (int j = 0; j < colblocks(); j++)
    for (int i = 0; i < rowblocks(); i++) {
      const Tile& t = tile(i, j);
      out[(size_t)i + (size_t)j * rowblocks()] = (t.r >= 0 && t.lowrank) ? t.r : -1;
    }
}

// BLRMatrix::compress_and_factor, algorithm RL (BLRMatrix.cpp:114-175), one block step = a handful of batched launches;
// stopped after `nsteps` block steps it is the partial factorization of a front (BLRMatrix.cpp:740-1037)
void DeviceBLR::factor_rl(const char* adm, int nsteps) {
  const double t0 = now();
  const int rb = rowblocks();
  int* info = dpiv_ + m_;
  for (double& p : phase_ms) p = 0;
  static const bool one_stream = [] { const char* e = std::getenv("STRUMPACK_AMD_BLR_ONE_STREAM"); return e && e[0] == '1'; }();
  const bool two_streams = !one_stream;
  if (two_streams && !ctx2_) ck(hssk_ctx_create(&ctx2_, o_.device));
  f_schur = f_total = b_schur = 0;
  schur_launches = 0;
  // phases on the device clock (time_phases): stopwatches 0..3 of the context bracket the launches of each phase
  auto watch = [&](int id, bool start) {
    if (!time_phases) return;
    ck(start ? hssk_watch_start(ctx_, id) : hssk_watch_stop(ctx_, id));
  };
  std::vector<char> adm_nd;   // the diagonal is never compressed
  if (adm) adm_nd.assign(adm, adm + (size_t)rb * rb); else adm_nd.assign((size_t)rb * rb, 1);
  for (int i = 0; i < rb; i++) adm_nd[(size_t)i + (size_t)i * rb] = 0;
  invL_.assign(rb, nullptr);
  invU_.assign(rb, nullptr);
  // look-ahead of the Schur updates (below): block steps per deferred update of the trailing array; 1 = every step updates
  // everything (the right-looking schedule as written)
  // Depth: with nt block rows / columns left, a block of `la` steps reads and writes ~ la (nt) tiles in its strips per step and
  // nt^2 / la in its deferred update: least near la = sqrt(nt) (measured on the 96 x 96 plane's front, nt = 136: Schur phase
  // 104 ms at depth 1, 41 ms at 8 and at 16).  STRUMPACK_AMD_BLR_LOOKAHEAD fixes it.
  const int la_fixed = [] { const char* e = std::getenv("STRUMPACK_AMD_BLR_LOOKAHEAD"); return e ? std::max(1, std::atoi(e)) : 0; }();
  auto depth = [&](int first) { return la_fixed ? la_fixed : std::max(4, std::min(24, (int)std::lround(std::sqrt((double)(rb - first))))); };
  int la = depth(0);
  const bool defer = la_fixed != 1;
  struct Pending { int p; double* T; int ldT; std::vector<int> off; };   // step, its product T_p (rows > p), column offset of tile (p, j) at [j - p - 1]
  std::vector<Pending> pend;
  int b0 = 0, b1 = std::min(la, nsteps);
  blk_->rewind();
  // ---- LU of a diagonal tile, in place -- on the second stream: the compression of its block row / column neither reads the
  // tile nor waits for its factors; the triangular solves join the two streams.  The tile is final as soon as the previous
  // step's update of THAT tile is enqueued: the update is taken out of the step's products and launched first, and the LU
  // is issued right behind it (lu_issued) -- it then runs beside the REST of the previous step's updates as well, not only
  // beside its own step's compression (a front of many small-rank tiles is a chain of diagonal LUs: 0.46 ms per 156-row tile
  // on one workgroup, 119 of the 215 ms of the 200 x 200 root front).
  hssk_ctx* lctx = (two_streams && !time_phases) ? ctx2_ : ctx_;
  std::vector<char> lu_issued(rb + 1, 0);
  auto issue_lu = [&](int i) {
    if (lu_issued[i]) return;
    lu_issued[i] = 1;
    const int mi = tm(i);
    hssk_lu_desc lu{blk(i, i), mi, (int)ld_, dpiv_ + roff_[i], info + i};
    if (lctx != ctx_) ck(hssk_stream_wait(lctx, ctx_));
    watch(0, true);
    if (mi) ck(hssk_getrf_vbatched(lctx, &lu, 1));
    if (mi > 0) {
      // the inverted diagonal blocks of the tile's L and U, once: this step's two triangular solves (tiles of 128 rows and more)
      // and every later solve phase with the tile (the single-launch sweeps: any size) take them from here
      // (hssk_trsm_desc::Tinv, hssk_blr_row::Tinv) instead of inverting per call
      const size_t per = (size_t)((mi + 63) / 64) * 64 * 64;
      double* inv = store_->dbl(2 * per);
      hssk_trtri_desc tt[2] = {{blk(i, i), inv, mi, (int)ld_, 2}, {blk(i, i), inv + per, mi, (int)ld_, 1}};
      ck(hssk_trtri_diag_vbatched(lctx, tt, 2));
      invL_[i] = inv;
      invU_[i] = inv + per;
    }
    watch(0, false);
    f_total += (2.0 / 3.0) * mi * (double)mi * mi;
  };
  static const bool lu_ahead = [] { const char* e = std::getenv("STRUMPACK_AMD_BLR_NO_LU_AHEAD"); return !(e && e[0] == '1'); }();
  for (int i = 0; i < nsteps; i++) {
    tmp_->rewind();
    const int mi = tm(i);
    issue_lu(i);
    if (i + 1 == rb) { if (lctx != ctx_) ck(hssk_stream_wait(ctx_, lctx)); break; }
    // ---- compress the block row and the block column of this step from the running Schur complement
    std::vector<std::pair<int, int>> ij;
    for (int j = i + 1; j < rb; j++) { ij.emplace_back(i, j); ij.emplace_back(j, i); }
    // the U factors of the block row are carved side by side in one m_i x R panel (used as ONE operand below and in the
    // backward solve); their ranks are not known yet: reserve the dense bound and trim the view afterwards
    watch(1, true);
    compress_tiles(ij, adm_nd.data());
    watch(1, false);
    int R = 0;
    for (int j = i + 1; j < rb; j++) R += tile(i, j).r;
    double* Ucat = store_->dbl((size_t)mi * std::max(R, 1));
    {
      std::vector<hssk_colgather_desc> mv;
      int off = 0;
      for (int j = i + 1; j < rb; j++) {
        Tile& t = tile(i, j);
        if (t.r > 0 && mi > 0) mv.push_back(hssk_colgather_desc{t.U, Ucat + (size_t)off * mi, nullptr, mi, t.r, mi, mi, 0});
        t.U = Ucat + (size_t)off * mi;
        off += t.r;
      }
      if (!mv.empty()) ck(hssk_gather_cols(ctx_, mv.data(), (int)mv.size()));
    }
    // ---- block row: U <- L^{-1} P U (all tiles at once: the panel);  block column: V <- U_ii^{-T} V
    if (lctx != ctx_) ck(hssk_stream_wait(ctx_, lctx));
    watch(2, true);
    if (R > 0 && mi > 0) {
      hssk_lusolve_desc sw{blk(i, i), dpiv_ + roff_[i], Ucat, mi, R, (int)ld_, mi};
      ck(hssk_laswp_vbatched(ctx_, &sw, 1));
      hssk_trsm_desc tl{blk(i, i), Ucat, mi, R, (int)ld_, mi, 1, 0, 1, invL_[i]};
      ck(hssk_trsm_vbatched(ctx_, &tl, 1));
    }
    {
      std::vector<hssk_trsm_desc> tu;
      for (int k = i + 1; k < rb; k++) {
        Tile& t = tile(k, i);
        if (t.r > 0 && mi > 0) tu.push_back(hssk_trsm_desc{blk(i, i), t.V, mi, t.r, (int)ld_, mi, 0, 1, 0, invU_[i]});
      }
      if (!tu.empty()) ck(hssk_trsm_vbatched(ctx_, tu.data(), (int)tu.size()));
      for (auto& d : tu) f_total += (double)d.n * d.n * d.nrhs;
      f_total += (double)mi * mi * R;
    }
    watch(2, false);
    // ---- Schur update of the trailing array, always into full rank (BLRMatrix.cpp:160-175):
    //   A_kj -= U_ki (V_ki^T U_ij) V_ij^T  for k, j > i, as three batched GEMMs over the block column / row:
    //   G_k = V_ki^T [U_i,i+1 .. U_i,rb)  (r_ki x R);  T(k rows, :) = U_ki G_k;  A(i+1:, j cols) -= T(:, R_j) V_ij^T
    // Applied step by step, the last product reads and writes the WHOLE trailing array once per block step (16 bytes per
    // 2 r flops at tile rank r: HBM-bound, and most of a large front's time).  The steps are therefore taken in blocks of
    // `la` (look-ahead): inside a block a step updates at once only what the block's own later steps need -- the block
    // columns (i, b1) in full and the block rows (i, b1) of the other columns -- and keeps T_i; after the block's last step
    // the rest of the array (rows and columns >= b1) receives all `la` updates in ONE product per block column,
    // A(b1:, j) -= [T_b0(:, R_b0,j) | ... | T_b1-1(:, R_b1-1,j)] [V_b0,j | ... | V_b1-1,j]^T: read and written once per block,
    // with an inner dimension la times as long (the same sums in another order: left-looking inside the trailing part).
    if (i == b1) { b0 = b1; la = depth(b0); b1 = std::min(b0 + la, nsteps); }
    (void)b0;
    const int nrest = m_ - roff_[i + 1];
    if (R > 0 && nrest > 0) {
      double* T = (defer ? blk_ : tmp_)->dbl((size_t)nrest * R);
      std::vector<hssk_gemm_desc> gG, gT, gF, gP;
      for (int k = i + 1; k < rb; k++) {
        Tile& t = tile(k, i);
        const int mk = tm(k);
        double* Trow = T + (roff_[k] - roff_[i + 1]);
        if (t.r > 0 && mk > 0) {
          double* G = tmp_->dbl((size_t)t.r * R);
          gG.push_back(hssk_gemm_desc{t.V, Ucat, G, t.r, R, mi, mi, mi, t.r, 1, 0, 1.0, 0.0});
          gT.push_back(hssk_gemm_desc{t.U, G, Trow, mk, R, t.r, mk, t.r, nrest, 0, 0, 1.0, 0.0});
        } else if (mk > 0) {
          gT.push_back(hssk_gemm_desc{Trow, Trow, Trow, mk, R, 0, nrest, nrest, nrest, 0, 0, 0.0, 0.0});   // zero block
        }
      }
      const int rows_now = defer ? roff_[b1] - roff_[i + 1] : nrest;   // rows of the columns >= b1 that cannot wait
      Pending pd{i, T, nrest, {}};
      int off = 0;
      for (int j = i + 1; j < rb; j++) {
        Tile& t = tile(i, j);
        const int nj = tn(j);
        const int rows = (!defer || j < b1) ? nrest : rows_now;
        // (the next diagonal tile's share of the update goes first, on its own: its LU starts behind it)
        const int top = (lu_ahead && j == i + 1 && i + 1 < nsteps) ? std::min(rows, tm(i + 1)) : 0;
        if (t.r > 0 && nj > 0 && top > 0)
          gP.push_back(hssk_gemm_desc{T + (size_t)off * nrest, t.V, dA_ + roff_[i + 1] + (size_t)coff_[j] * ld_, top, nj, t.r,
                                      nrest, nj, (int)ld_, 0, 1, -1.0, 1.0});
        if (t.r > 0 && nj > 0 && rows > top)
          gF.push_back(hssk_gemm_desc{T + (size_t)off * nrest + top, t.V, dA_ + roff_[i + 1] + top + (size_t)coff_[j] * ld_, rows - top, nj, t.r,
                                      nrest, nj, (int)ld_, 0, 1, -1.0, 1.0});
        pd.off.push_back(off);
        off += t.r;
      }
      if (defer) pend.push_back(std::move(pd));
      watch(3, true);
      if (!gG.empty()) ck(hssk_gemm_vbatched(ctx_, gG.data(), (int)gG.size()));
      if (!gT.empty()) ck(hssk_gemm_vbatched(ctx_, gT.data(), (int)gT.size()));
      if (!gP.empty()) ck(hssk_gemm_vbatched(ctx_, gP.data(), (int)gP.size()));
      watch(3, false);
      // the next diagonal tile is final unless it still waits for this block's deferred update (then: behind the flush below)
      if (lu_ahead && i + 1 < nsteps && !(defer && i + 1 == b1)) issue_lu(i + 1);
      watch(3, true);
      if (!gF.empty()) ck(hssk_gemm_vbatched(ctx_, gF.data(), (int)gF.size()));
      watch(3, false);
      schur_launches += (!gG.empty()) + (!gT.empty()) + (!gF.empty()) + (!gP.empty());
      for (auto* gl : {&gG, &gT, &gF, &gP})
        for (auto& d : *gl) {
          f_schur += 2.0 * d.m * (double)d.n * d.k;
          b_schur += 8.0 * ((double)d.m * d.k + (double)d.k * d.n + (d.beta != 0.0 ? 2.0 : 1.0) * d.m * (double)d.n);
        }
    }
    if (defer && i + 1 == b1) {
      // ---- the block's deferred updates: rows and columns >= b1
      const int r0 = roff_[b1], nrows = m_ - r0;
      std::vector<hssk_colgather_desc> cp, cpP;
      std::vector<hssk_gemm_desc> gD, gDP;
      if (nrows > 0)
        for (int j = b1; j < rb; j++) {
          const int nj = tn(j);
          int K = 0;
          for (auto& q : pend) K += tile(q.p, j).r;
          if (K <= 0 || nj <= 0) continue;
          double* Tcat = blk_->dbl((size_t)nrows * K);
          double* Vcat = blk_->dbl((size_t)nj * K);
          int col = 0;
          for (auto& q : pend) {
            const Tile& t = tile(q.p, j);
            if (t.r <= 0) continue;
            auto& cpl = (lu_ahead && j == b1 && b1 < nsteps) ? cpP : cp;   // (the next diagonal tile's block column first)
            cpl.push_back(hssk_colgather_desc{q.T + (r0 - roff_[q.p + 1]) + (size_t)q.off[j - q.p - 1] * q.ldT, Tcat + (size_t)col * nrows, nullptr,
                                              nrows, t.r, q.ldT, nrows, 0});
            cpl.push_back(hssk_colgather_desc{t.V, Vcat + (size_t)col * nj, nullptr, nj, t.r, nj, nj, 0});
            col += t.r;
          }
          const int top = (lu_ahead && j == b1 && b1 < nsteps) ? std::min(nrows, tm(b1)) : 0;
          if (top > 0) gDP.push_back(hssk_gemm_desc{Tcat, Vcat, dA_ + r0 + (size_t)coff_[j] * ld_, top, nj, K, nrows, nj, (int)ld_, 0, 1, -1.0, 1.0});
          if (nrows > top)
            gD.push_back(hssk_gemm_desc{Tcat + top, Vcat, dA_ + r0 + top + (size_t)coff_[j] * ld_, nrows - top, nj, K, nrows, nj, (int)ld_, 0, 1, -1.0, 1.0});
        }
      watch(3, true);
      if (!cpP.empty()) ck(hssk_gather_cols(ctx_, cpP.data(), (int)cpP.size()));
      if (!gDP.empty()) ck(hssk_gemm_vbatched(ctx_, gDP.data(), (int)gDP.size()));
      watch(3, false);
      if (lu_ahead && b1 < nsteps) issue_lu(b1);
      watch(3, true);
      if (!cp.empty()) ck(hssk_gather_cols(ctx_, cp.data(), (int)cp.size()));
      if (!gD.empty()) ck(hssk_gemm_vbatched(ctx_, gD.data(), (int)gD.size()));
      watch(3, false);
      schur_launches += !gD.empty();
      gD.insert(gD.end(), gDP.begin(), gDP.end());
      for (auto& d : gD) {
        f_schur += 2.0 * d.m * (double)d.n * d.k;
        b_schur += 8.0 * (3.0 * (double)d.m * d.k + 3.0 * (double)d.k * d.n + 2.0 * d.m * (double)d.n);   // (the gathered operands: read, written, read)
      }
      pend.clear();
      blk_->rewind();
    }
  }
  f_total += f_schur;
  if (time_phases)
    for (int p = 0; p < 4; p++) phase_ms[p] = hssk_watch_read_ms(ctx_, p, nullptr);
  ck(hssk_sync(ctx_));
  nsteps_ = nsteps;
  std::vector<int> hinfo(rb, 0);
  ck(hssk_memcpy_d2h(ctx_, hinfo.data(), info, (long long)sizeof(int) * rb));
  for (int i = 0; i < nsteps; i++)
    if (hinfo[i] > 0) throw std::runtime_error("BLR factorization: zero pivot in diagonal tile " + std::to_string(i));
  compressed_ = factored_ = true;
  blk_.reset(new Arena2(size_t(256) << 20));   // (the block products of a large front are gigabytes: not kept)
  t_factor = now() - t0;
}

// y = op(B) x.  Tiles are visited in rounds s: round s pairs block row i with block column (i + s) mod cb (transposed: the
// other way round), so the outputs of one round are distinct and its tiles are two batched GEMMs.
void DeviceBLR::mult(char trans, int nrhs, const double* x, long long ldx, double* y, long long ldy) const {
  std::lock_guard<std::recursive_mutex> op_guard(op_mu_);   // operations on one matrix take turns
  if (!compressed_ || factored_) throw std::logic_error("BLR mult: needs a compressed, unfactored matrix (construct_from_dense)");
  if (nrhs <= 0) return;
  const bool T = !(trans == 'N' || trans == 'n');
  const int rb = rowblocks(), cb = colblocks();
  const int nin = T ? m_ : n_, nout = T ? n_ : m_;
  Arena2& tmp = *tmp_;
  tmp.rewind();
  double* dx = tmp.dbl((size_t)nin * nrhs);
  double* dy = tmp.dbl((size_t)nout * nrhs);
  ck(hssk_memcpy2d_h2d(ctx_, dx, sizeof(double) * nin, x, sizeof(double) * ldx, sizeof(double) * nin, nrhs));
  ck(hssk_memset_zero(ctx_, dy, (long long)sizeof(double) * nout * nrhs));
  const int rounds = T ? rb : cb;
  for (int s = 0; s < rounds; s++) {
    std::vector<hssk_gemm_desc> g1, g2;
    for (int o = 0; o < (T ? cb : rb); o++) {
      const int i = T ? (o + s) % rb : o, j = T ? o : (o + s) % cb;
      if (T && s >= rb) continue;
      const Tile& t = tile(i, j);
      const int m = tm(i), n = tn(j);
      if (t.r <= 0 || m == 0 || n == 0) continue;
      double* tt = tmp.dbl((size_t)t.r * nrhs);
      if (!T) {
        g1.push_back(hssk_gemm_desc{t.V, dx + coff_[j], tt, t.r, nrhs, n, n, nin, t.r, 1, 0, 1.0, 0.0});
        g2.push_back(hssk_gemm_desc{t.U, tt, dy + roff_[i], m, nrhs, t.r, m, t.r, nout, 0, 0, 1.0, 1.0});
      } else {
        g1.push_back(hssk_gemm_desc{t.U, dx + roff_[i], tt, t.r, nrhs, m, m, nin, t.r, 1, 0, 1.0, 0.0});
        g2.push_back(hssk_gemm_desc{t.V, tt, dy + coff_[j], n, nrhs, t.r, n, t.r, nout, 0, 0, 1.0, 1.0});
      }
    }
    if (!g1.empty()) ck(hssk_gemm_vbatched(ctx_, g1.data(), (int)g1.size()));
    if (!g2.empty()) ck(hssk_gemm_vbatched(ctx_, g2.data(), (int)g2.size()));
  }
  ck(hssk_memcpy2d_d2h(ctx_, y, sizeof(double) * ldy, dy, sizeof(double) * nout, sizeof(double) * nout, nrhs));
}

// largest summed rank of a block row / block column of the eliminated steps (size of the sweeps' scratch vector)
int DeviceBLR::rmax() const {
  const int rb = rowblocks();
  int Rmax = 1;
  for (int i = 0; i < nsteps_; i++) {
    int R = 0, C = 0;
    for (int j = i + 1; j < rb; j++) { R += std::max(tile(i, j).r, 0); C += std::max(tile(j, i).r, 0); }
    Rmax = std::max(Rmax, std::max(R, C));
  }
  return Rmax;
}

// block forward substitution over the eliminated steps, X (n_ x nrhs, device): x_i <- L_ii^{-1} P_i x_i, then
// x_k -= U_ki (V_ki^T x_i) for every block row k below -- the rest of the separator AND the update rows (B21)
// The two substitutions for ONE right-hand side as one launch each (hssk_blr_sweep: a workgroup per block row, waiting for the
// rows it depends on through its tiles of non-zero rank); false: not applicable (tiles beyond 512 rows, a tile without its
// inverted diagonal blocks, STRUMPACK_AMD_BLR_SWEEP=0) -- the caller walks the block steps.
bool DeviceBLR::sweep(double* X, bool backward) const {
  static const bool off = [] { const char* e = std::getenv("STRUMPACK_AMD_BLR_SWEEP"); return e && e[0] == '0'; }();
  if (off) return false;
  const int rb = rowblocks(), ns = nsteps_;
  {
    const SweepTables& T = sweep_tab_[backward ? 1 : 0];
    if (T.built) {
      if (!T.ok) return false;
      int* flags = tmp_->ints((size_t)T.nrows + 1);
      ck(hssk_blr_sweep_resident(ctx_, T.rows, T.nrows, T.terms, X, flags));
      return true;
    }
  }
  std::vector<hssk_blr_row> rows;
  std::vector<hssk_blr_term> terms;
  for (int i = 0; i < ns; i++)
    if (tm(i) > 512 || (tm(i) > 0 && (!invL_[i] || !invU_[i]))) return false;
  for (int k = ns; k < rb; k++) if (tm(k) > 512) return false;
  if (!backward) {
    // block row k = workgroup k: terms (k, i), i < min(k, steps), in ascending order of the step they wait for
    for (int k = 0; k < rb; k++) {
      hssk_blr_row r{};
      r.first_term = (int)terms.size();
      r.off = roff_[k]; r.m = tm(k);
      for (int i = 0; i < std::min(k, ns); i++) {
        const Tile& t = tile(k, i);
        if (t.r <= 0 || tm(k) == 0 || tm(i) == 0) continue;
        terms.push_back(hssk_blr_term{t.U, t.V, t.r, tm(i), roff_[i], i});
      }
      r.nterms = (int)terms.size() - r.first_term;
      if (k < ns && r.m > 0) { r.LU = blk(k, k); r.lda = (int)ld_; r.mode = 0; r.piv = dpiv_ + roff_[k]; r.Tinv = invL_[k]; }
      rows.push_back(r);
    }
  } else {
    // block row i = workgroup steps - 1 - i: terms (i, j), j > i; the update columns are inputs, the separator's come from
    // the workgroups in front
    for (int w = 0; w < ns; w++) {
      const int i = ns - 1 - w;
      hssk_blr_row r{};
      r.first_term = (int)terms.size();
      r.off = roff_[i]; r.m = tm(i);
      for (int j = rb - 1; j > i; j--) {   // (inputs first, then the rows finished longest ago)
        const Tile& t = tile(i, j);
        if (t.r <= 0 || tm(i) == 0 || tn(j) == 0) continue;
        terms.push_back(hssk_blr_term{t.U, t.V, t.r, tn(j), coff_[j], j < ns ? ns - 1 - j : -1});
      }
      r.nterms = (int)terms.size() - r.first_term;
      if (r.m > 0) { r.LU = blk(i, i); r.lda = (int)ld_; r.mode = 1; r.piv = nullptr; r.Tinv = invU_[i]; }
      rows.push_back(r);
    }
  }
  // the tables go to the device once per factorization: the factors they describe do not change between solves
  SweepTables& T = sweep_tab_[backward ? 1 : 0];
  T.built = true;
  T.nrows = (int)rows.size();
  if (hssk_blr_sweep_check(rows.data(), (int)rows.size(), terms.data(), (int)terms.size())) return false;   // (ok stays false)
  T.rows = (hssk_blr_row*)store_->alloc(sizeof(hssk_blr_row) * std::max<size_t>(rows.size(), 1));
  T.terms = (hssk_blr_term*)store_->alloc(sizeof(hssk_blr_term) * std::max<size_t>(terms.size(), 1));
  ck(hssk_memcpy_h2d(ctx_, T.rows, rows.data(), (long long)(sizeof(hssk_blr_row) * rows.size())));
  if (!terms.empty()) ck(hssk_memcpy_h2d(ctx_, T.terms, terms.data(), (long long)(sizeof(hssk_blr_term) * terms.size())));
  T.ok = true;
  int* flags = tmp_->ints(rows.size() + 1);
  ck(hssk_blr_sweep_resident(ctx_, T.rows, T.nrows, T.terms, X, flags));
  return true;
}

void DeviceBLR::fwd(double* X, int nrhs, double* t, int Rmax) const {
  const int rb = rowblocks();
  if (nrhs == 1 && sweep(X, false)) return;
  for (int i = 0; i < nsteps_; i++) {
    const int mi = tm(i);
    if (!mi) continue;
    double* Xi = X + roff_[i];
    hssk_lusolve_desc sw{blk(i, i), dpiv_ + roff_[i], Xi, mi, nrhs, (int)ld_, n_};
    ck(hssk_laswp_vbatched(ctx_, &sw, 1));
    hssk_trsm_desc tl{blk(i, i), Xi, mi, nrhs, (int)ld_, n_, 1, 0, 1, invL_[i]};
    ck(hssk_trsm_vbatched(ctx_, &tl, 1));
    std::vector<hssk_gemm_desc> g1, g2;
    int off = 0;
    for (int k = i + 1; k < rb; k++) {   // x_k -= U_ki (V_ki^T x_i)
      const Tile& tk = tile(k, i);
      if (tk.r <= 0 || tm(k) == 0) continue;
      g1.push_back(hssk_gemm_desc{tk.V, Xi, t + off, tk.r, nrhs, mi, mi, n_, Rmax, 1, 0, 1.0, 0.0});
      g2.push_back(hssk_gemm_desc{tk.U, t + off, X + roff_[k], tm(k), nrhs, tk.r, tm(k), Rmax, n_, 0, 0, -1.0, 1.0});
      off += tk.r;
    }
    if (!g1.empty()) ck(hssk_gemm_vbatched(ctx_, g1.data(), (int)g1.size()));
    if (!g2.empty()) ck(hssk_gemm_vbatched(ctx_, g2.data(), (int)g2.size()));
  }
}

// block backward substitution over the eliminated steps: x_i <- U_ii^{-1} (x_i - sum_{j > i} U_ij V_ij^T x_j), j through the
// rest of the separator and the update columns (B12)
void DeviceBLR::bwd(double* X, int nrhs, double* t, int Rmax) const {
  const int rb = rowblocks();
  if (nrhs == 1 && sweep(X, true)) return;
  for (int i = nsteps_ - 1; i >= 0; i--) {
    const int mi = tm(i);
    if (!mi) continue;
    double* Xi = X + roff_[i];
    std::vector<hssk_gemm_desc> g1;
    int R = 0;
    const double* Ucat = nullptr;
    for (int j = i + 1; j < rb; j++) {   // t = [V_ij^T x_j]_j stacked;  x_i -= [U_ij]_j t   (the U's of a block row are one panel)
      const Tile& tj = tile(i, j);
      if (tj.r <= 0) continue;
      if (!Ucat) Ucat = tj.U;
      g1.push_back(hssk_gemm_desc{tj.V, X + coff_[j], t + R, tj.r, nrhs, tn(j), tn(j), n_, Rmax, 1, 0, 1.0, 0.0});
      R += tj.r;
    }
    if (!g1.empty()) {
      ck(hssk_gemm_vbatched(ctx_, g1.data(), (int)g1.size()));
      hssk_gemm_desc g2{Ucat, t, Xi, mi, nrhs, R, mi, Rmax, n_, 0, 0, -1.0, 1.0};
      ck(hssk_gemm_vbatched(ctx_, &g2, 1));
    }
    hssk_trsm_desc tu{blk(i, i), Xi, mi, nrhs, (int)ld_, n_, 0, 0, 0, invU_[i]};
    ck(hssk_trsm_vbatched(ctx_, &tu, 1));
  }
}

// BLRMatrix::solve (BLRMatrix.hpp:118-122): x <- P x, block forward substitution with the unit lower factor, block
// backward substitution with the upper factor
void DeviceBLR::solve(int nrhs, double* b, long long ldb) const {
  std::lock_guard<std::recursive_mutex> op_guard(op_mu_);   // operations on one matrix take turns
  if (!factored_) throw std::logic_error("BLR solve: the matrix has not been factored (construct_and_factor_from_dense)");
  if (nsteps_ != rowblocks()) throw std::logic_error("BLR solve: the matrix is a partially factored front (use front_forward / front_backward)");
  if (nrhs <= 0 || n_ == 0) return;
  Arena2& tmp = *tmp_;
  tmp.rewind();
  double* X = tmp.dbl((size_t)n_ * nrhs);
  ck(hssk_memcpy2d_h2d(ctx_, X, sizeof(double) * n_, b, sizeof(double) * ldb, sizeof(double) * n_, nrhs));
  const int Rmax = rmax();
  double* t = tmp.dbl((size_t)Rmax * nrhs);
  fwd(X, nrhs, t, Rmax);
  bwd(X, nrhs, t, Rmax);
  ck(hssk_memcpy2d_d2h(ctx_, b, sizeof(double) * ldb, X, sizeof(double) * n_, sizeof(double) * n_, nrhs));
  if (hssk_sweep_status(ctx_)) throw std::runtime_error(std::string("BLR solve: ") + hssk_last_error());
}

// FrontBLR::fwd_solve_phase2 (FrontBLR.cpp:525-547): bsep <- L11^{-1} P bsep, bupd <- bupd - B21 bsep
void DeviceBLR::front_forward(int nrhs, double* bsep, long long ldb, double* bupd, long long ldu) const {
  std::lock_guard<std::recursive_mutex> op_guard(op_mu_);
  if (!factored_) throw std::logic_error("BLR front: not factored");
  const int ds = sep_rows(), du = upd_rows();
  if (nrhs <= 0 || ds == 0) return;
  if (du > 0 && !bupd) throw std::invalid_argument("BLR front: no update part of the right-hand side");
  Arena2& tmp = *tmp_;
  tmp.rewind();
  double* X = tmp.dbl((size_t)n_ * nrhs);
  ck(hssk_memcpy2d_h2d(ctx_, X, sizeof(double) * n_, bsep, sizeof(double) * ldb, sizeof(double) * ds, nrhs));
  if (du > 0) ck(hssk_memcpy2d_h2d(ctx_, X + ds, sizeof(double) * n_, bupd, sizeof(double) * ldu, sizeof(double) * du, nrhs));
  const int Rmax = rmax();
  double* t = tmp.dbl((size_t)Rmax * nrhs);
  fwd(X, nrhs, t, Rmax);
  ck(hssk_memcpy2d_d2h(ctx_, bsep, sizeof(double) * ldb, X, sizeof(double) * n_, sizeof(double) * ds, nrhs));
  if (du > 0) ck(hssk_memcpy2d_d2h(ctx_, bupd, sizeof(double) * ldu, X + ds, sizeof(double) * n_, sizeof(double) * du, nrhs));
  if (hssk_sweep_status(ctx_)) throw std::runtime_error(std::string("BLR front, forward phase: ") + hssk_last_error());
}

// FrontBLR::bwd_solve_phase1 (FrontBLR.cpp:550-570): ysep <- U11^{-1} (ysep - B12 yupd)
void DeviceBLR::front_backward(int nrhs, double* ysep, long long ldy, const double* yupd, long long ldu) const {
  std::lock_guard<std::recursive_mutex> op_guard(op_mu_);
  if (!factored_) throw std::logic_error("BLR front: not factored");
  const int ds = sep_rows(), du = upd_rows();
  if (nrhs <= 0 || ds == 0) return;
  if (du > 0 && !yupd) throw std::invalid_argument("BLR front: no update part of the solution");
  Arena2& tmp = *tmp_;
  tmp.rewind();
  double* X = tmp.dbl((size_t)n_ * nrhs);
  ck(hssk_memcpy2d_h2d(ctx_, X, sizeof(double) * n_, ysep, sizeof(double) * ldy, sizeof(double) * ds, nrhs));
  if (du > 0) ck(hssk_memcpy2d_h2d(ctx_, X + ds, sizeof(double) * n_, yupd, sizeof(double) * ldu, sizeof(double) * du, nrhs));
  const int Rmax = rmax();
  double* t = tmp.dbl((size_t)Rmax * nrhs);
  bwd(X, nrhs, t, Rmax);
  ck(hssk_memcpy2d_d2h(ctx_, ysep, sizeof(double) * ldy, X, sizeof(double) * n_, sizeof(double) * ds, nrhs));
  if (hssk_sweep_status(ctx_)) throw std::runtime_error(std::string("BLR front, backward phase: ") + hssk_last_error());
}

void DeviceBLR::dense(double* A, long long lda) const {
  std::lock_guard<std::recursive_mutex> op_guard(op_mu_);   // operations on one matrix take turns
  if (!compressed_ || factored_) throw std::logic_error("BLR dense: needs a compressed, unfactored matrix");
  Arena2& tmp = *tmp_;
  tmp.rewind();
  double* D = tmp.dbl((size_t)std::max(m_, 1) * std::max(n_, 1));
  ck(hssk_memset_zero(ctx_, D, (long long)sizeof(double) * std::max(m_, 1) * std::max(n_, 1)));
  std::vector<hssk_gemm_desc> g;
  for (int j = 0; j < colblocks(); j++)
    for (int i = 0; i < rowblocks(); i++) {
      const Tile& t = tile(i, j);
      if (t.r > 0 && tm(i) && tn(j))
        g.push_back(hssk_gemm_desc{t.U, t.V, D + roff_[i] + (size_t)coff_[j] * m_, tm(i), tn(j), t.r, tm(i), tn(j), m_, 0, 1, 1.0, 0.0});
    }
  if (!g.empty()) ck(hssk_gemm_vbatched(ctx_, g.data(), (int)g.size()));
  ck(hssk_memcpy2d_d2h(ctx_, A, sizeof(double) * lda, D, sizeof(double) * m_, sizeof(double) * m_, n_));
}

int DeviceBLR::rank() const {
  int r = 0;
  for (auto& t : tiles_) if (t.lowrank) r = std::max(r, t.r);
  return r;
}
long long DeviceBLR::nonzeros() const {
  long long nz = 0;
  for (int j = 0; j < colblocks(); j++)
    for (int i = 0; i < rowblocks(); i++) {
      const Tile& t = tile(i, j);
      if (t.r < 0 || !t.lowrank) nz += (long long)tm(i) * tn(j);   // dense (diagonal tiles of a factorization included)
      else nz += (long long)t.r * (tm(i) + tn(j));
    }
  return nz;
}
void DeviceBLR::front_nonzeros(long long out[3]) const {
  out[0] = out[1] = out[2] = 0;
  const int rb = rowblocks(), ns = nsteps_;
  for (int j = 0; j < rb; j++)
    for (int i = 0; i < rb; i++) {
      if (i >= ns && j >= ns) continue;   // F22: not part of the factors
      const Tile& t = tile(i, j);
      const long long nz = (t.r < 0 || !t.lowrank) ? (long long)tm(i) * tn(j) : (long long)t.r * (tm(i) + tn(j));
      out[(i < ns && j < ns) ? 0 : (i < ns ? 1 : 2)] += nz;
    }
}
long long DeviceBLR::memory() const { return nonzeros() * (long long)sizeof(double); }

}  // namespace BLR
}  // namespace strumpack
