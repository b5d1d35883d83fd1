// DeviceBLR implementation (see blr_engine.hpp for the reference behaviour it follows).
#include "blr_engine.hpp"

#include <algorithm>
#include <chrono>
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

// bump allocator over device chunks (hssk_malloc); rewind() keeps the chunks
class Arena2 {
 public:
  explicit Arena2(size_t chunk) : chunk_(chunk) {}
  ~Arena2() { for (auto& c : chunks_) hssk_free(c.first); }
  void* alloc(size_t bytes) {
    bytes = (std::max<size_t>(bytes, 8) + 255) & ~size_t(255);
    while (bytes > left_) {
      if (next_ < chunks_.size()) { cur_ = (char*)chunks_[next_].first; left_ = chunks_[next_].second; next_++; continue; }
      const size_t c = std::max(chunk_, bytes);
      void* p = hssk_malloc((long long)c);
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
  tiles_.assign((size_t)rowblocks() * colblocks(), Tile());
}

DeviceBLR::~DeviceBLR() {
  if (ctx_) hssk_sync(ctx_);
  store_.reset();
  tmp_.reset();
  if (dA_) hssk_free(dA_);
  hssk_ctx_destroy(ctx_);
}

void DeviceBLR::load(const double* A, long long lda, bool on_device) {
  if (dA_) { hssk_free(dA_); dA_ = nullptr; }
  ld_ = std::max(m_, 1);
  dA_ = (double*)hssk_malloc((long long)sizeof(double) * ld_ * std::max(n_, 1));
  if (!dA_) throw std::runtime_error("BLR: device allocation of the operand failed");
  if (on_device) {
    hssk_colgather_desc cp{A, dA_, nullptr, m_, n_, (int)lda, (int)ld_, 0};
    if (lda > 0x7fffffffLL) throw std::invalid_argument("BLR: leading dimension too large");
    ck(hssk_gather_cols(ctx_, &cp, 1));
  } else {
    ck(hssk_h2d_block_async(ctx_, dA_, ld_, A, lda, m_, n_));
    ck(hssk_copy_fence(ctx_));
  }
  store_->rewind();
  tiles_.assign(tiles_.size(), Tile());
  dpiv_ = store_->ints((size_t)std::max(m_, 1) + rowblocks() + 1);
  compressed_ = factored_ = false;
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
    idd.push_back(hssk_id_desc{W[k], m, m, n, o_.rel_tol, o_.abs_tol, o_.max_rank, perm[k], ranks + k, tmp.dbl(3 * (size_t)n)});
  }
  std::vector<int> hr(cnt, 0);
  if (!idd.empty()) {
    ck(hssk_gather_cols(ctx_, cp.data(), (int)cp.size()));
    ck(hssk_id_vbatched(ctx_, idd.data(), (int)idd.size()));
    ck(hssk_memcpy_d2h(ctx_, hr.data(), ranks, (long long)sizeof(int) * cnt));
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
  hssk_free(dA_);
  dA_ = nullptr;
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
  hssk_free(dA_);
  dA_ = nullptr;
  compressed_ = true;
  t_compress = now() - t0;
}

void DeviceBLR::compress_and_factor_host(const double* A, long long lda, const char* adm) {
  std::lock_guard<std::recursive_mutex> op_guard(op_mu_);   // operations on one matrix take turns
  load(A, lda, false);
  factor_rl(adm);
}
void DeviceBLR::compress_and_factor_device(const double* dA, long long lda, const char* adm) {
  std::lock_guard<std::recursive_mutex> op_guard(op_mu_);   // operations on one matrix take turns
  load(dA, lda, true);
  factor_rl(adm);
}

// BLRMatrix::compress_and_factor, algorithm RL (BLRMatrix.cpp:114-175), one block step = a handful of batched launches
void DeviceBLR::factor_rl(const char* adm) {
  if (m_ != n_ || roff_ != coff_) throw std::invalid_argument("BLR factorization needs a square matrix with the same clusters for rows and columns");
  const double t0 = now();
  const int rb = rowblocks();
  int* info = dpiv_ + m_;
  std::vector<char> adm_nd;   // the diagonal is never compressed
  if (adm) adm_nd.assign(adm, adm + (size_t)rb * rb); else adm_nd.assign((size_t)rb * rb, 1);
  for (int i = 0; i < rb; i++) adm_nd[(size_t)i + (size_t)i * rb] = 0;
  for (int i = 0; i < rb; i++) {
    tmp_->rewind();
    const int mi = tm(i);
    // ---- LU of the diagonal tile, in place
    hssk_lu_desc lu{blk(i, i), mi, (int)ld_, dpiv_ + roff_[i], info + i};
    if (mi) ck(hssk_getrf_vbatched(ctx_, &lu, 1));
    if (i + 1 == rb) break;
    // ---- compress the block row and the block column of this step from the running Schur complement
    std::vector<std::pair<int, int>> ij;
    for (int j = i + 1; j < rb; j++) { ij.emplace_back(i, j); ij.emplace_back(j, i); }
    // the U factors of the block row are carved side by side in one m_i x R panel (used as ONE operand below and in the
    // backward solve); their ranks are not known yet: reserve the dense bound and trim the view afterwards
    compress_tiles(ij, adm_nd.data());
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
    if (R > 0 && mi > 0) {
      hssk_lusolve_desc sw{blk(i, i), dpiv_ + roff_[i], Ucat, mi, R, (int)ld_, mi};
      ck(hssk_laswp_vbatched(ctx_, &sw, 1));
      hssk_trsm_desc tl{blk(i, i), Ucat, mi, R, (int)ld_, mi, 1, 0, 1};
      ck(hssk_trsm_vbatched(ctx_, &tl, 1));
    }
    {
      std::vector<hssk_trsm_desc> tu;
      for (int k = i + 1; k < rb; k++) {
        Tile& t = tile(k, i);
        if (t.r > 0 && mi > 0) tu.push_back(hssk_trsm_desc{blk(i, i), t.V, mi, t.r, (int)ld_, mi, 0, 1, 0});
      }
      if (!tu.empty()) ck(hssk_trsm_vbatched(ctx_, tu.data(), (int)tu.size()));
    }
    // ---- Schur update of the trailing array, always into full rank (BLRMatrix.cpp:160-175):
    //   A_kj -= U_ki (V_ki^T U_ij) V_ij^T  for k, j > i, as three batched GEMMs over the block column / row:
    //   G_k = V_ki^T [U_i,i+1 .. U_i,rb)  (r_ki x R);  T(k rows, :) = U_ki G_k;  A(i+1:, j cols) -= T(:, R_j) V_ij^T
    const int nrest = m_ - roff_[i + 1];
    if (R > 0 && nrest > 0) {
      double* T = tmp_->dbl((size_t)nrest * R);
      std::vector<hssk_gemm_desc> gG, gT, gF;
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
      int off = 0;
      for (int j = i + 1; j < rb; j++) {
        Tile& t = tile(i, j);
        const int nj = tn(j);
        if (t.r > 0 && nj > 0)
          gF.push_back(hssk_gemm_desc{T + (size_t)off * nrest, t.V, dA_ + roff_[i + 1] + (size_t)coff_[j] * ld_, nrest, nj, t.r,
                                      nrest, nj, (int)ld_, 0, 1, -1.0, 1.0});
        off += t.r;
      }
      if (!gG.empty()) ck(hssk_gemm_vbatched(ctx_, gG.data(), (int)gG.size()));
      if (!gT.empty()) ck(hssk_gemm_vbatched(ctx_, gT.data(), (int)gT.size()));
      if (!gF.empty()) ck(hssk_gemm_vbatched(ctx_, gF.data(), (int)gF.size()));
    }
  }
  ck(hssk_sync(ctx_));
  std::vector<int> hinfo(rb);
  ck(hssk_memcpy_d2h(ctx_, hinfo.data(), info, (long long)sizeof(int) * rb));
  for (int i = 0; i < rb; i++)
    if (hinfo[i] > 0) throw std::runtime_error("BLR factorization: zero pivot in diagonal tile " + std::to_string(i));
  compressed_ = factored_ = true;
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

// BLRMatrix::solve (BLRMatrix.hpp:118-122): x <- P x, block forward substitution with the unit lower factor, block
// backward substitution with the upper factor
void DeviceBLR::solve(int nrhs, double* b, long long ldb) const {
  std::lock_guard<std::recursive_mutex> op_guard(op_mu_);   // operations on one matrix take turns
  if (!factored_) throw std::logic_error("BLR solve: the matrix has not been factored (construct_and_factor_from_dense)");
  if (nrhs <= 0 || n_ == 0) return;
  const int rb = rowblocks();
  Arena2& tmp = *tmp_;
  tmp.rewind();
  double* X = tmp.dbl((size_t)n_ * nrhs);
  ck(hssk_memcpy2d_h2d(ctx_, X, sizeof(double) * n_, b, sizeof(double) * ldb, sizeof(double) * n_, nrhs));
  int Rmax = 1;
  for (int i = 0; i < rb; i++) {
    int R = 0, C = 0;
    for (int j = i + 1; j < rb; j++) { R += tile(i, j).r; C += tile(j, i).r; }
    Rmax = std::max(Rmax, std::max(R, C));
  }
  double* t = tmp.dbl((size_t)Rmax * nrhs);
  for (int i = 0; i < rb; i++) {
    const int mi = tm(i);
    if (!mi) continue;
    double* Xi = X + roff_[i];
    hssk_lusolve_desc sw{blk(i, i), dpiv_ + roff_[i], Xi, mi, nrhs, (int)ld_, n_};
    ck(hssk_laswp_vbatched(ctx_, &sw, 1));
    hssk_trsm_desc tl{blk(i, i), Xi, mi, nrhs, (int)ld_, n_, 1, 0, 1};
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
  for (int i = rb - 1; i >= 0; i--) {
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
    hssk_trsm_desc tu{blk(i, i), Xi, mi, nrhs, (int)ld_, n_, 0, 0, 0};
    ck(hssk_trsm_vbatched(ctx_, &tu, 1));
  }
  ck(hssk_memcpy2d_d2h(ctx_, b, sizeof(double) * ldb, X, sizeof(double) * n_, sizeof(double) * n_, nrhs));
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
long long DeviceBLR::memory() const { return nonzeros() * (long long)sizeof(double); }

}  // namespace BLR
}  // namespace strumpack
