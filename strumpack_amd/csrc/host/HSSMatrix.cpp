#include <atomic>
#include <thread>
#include <functional>
#include <chrono>
#include <fstream>

#include <mutex>
#include <exception>
#include "HSSMatrix.hpp"
#include "Kernel.hpp"
#include "NeighborSearch.hpp"

#include <sstream>

namespace strumpack {
namespace HSS {

EngineOptions HSSMatrix<double>::engine_options(const opts_t& o) {
  EngineOptions e;
  e.rel_tol = o.rel_tol(); e.abs_tol = o.abs_tol(); e.leaf_size = o.leaf_size(); e.max_rank = o.max_rank();
  e.d0 = o.d0(); e.dd = o.dd(); e.p = o.p();
  // HARD_RESTART (compress.hpp:235-298): the acceptance rule of ORIGINAL, but a failed round resets every node and the
  // next round recompresses the whole tree on all samples
  e.algorithm = o.compression_algorithm() == CompressionAlgorithm::STABLE ? 1 : (o.compression_algorithm() == CompressionAlgorithm::HARD_RESTART ? 2 : 0);
  // (user_defined_random -- the random block filled in by the user's multiplication routine, compress_stable.hpp:126-141 --
  // only has a meaning for compress(Amult, Aelem), which switches it on; the other constructors never call back for samples)
  e.random_engine = o.random_engine() == random::RandomEngine::LINEAR ? 0 : (o.random_engine() == random::RandomEngine::MERSENNE ? 1 : 2);
  e.random_dist = o.random_distribution() == random::RandomDistribution::NORMAL ? 0 : 1;
  e.sketch = o.compression_sketch() == CompressionSketch::SJLT ? 1 : 0;
  e.sjlt_algo = o.SJLT_algo() == SJLTAlgo::CHUNK ? 0 : 1;
  e.nnz0 = o.nnz0(); e.nnz = o.nnz();
  e.verbose = o.verbose();
  e.factor_ahead = o.factor_ahead();
  e.symmetric = o.symmetric_operand();
  if (const char* fa = std::getenv("STRUMPACK_AMD_FACTOR_AHEAD")) e.factor_ahead = std::atoi(fa) != 0;
  if (const char* d = std::getenv("STRUMPACK_AMD_DEVICE")) e.device = std::atoi(d);
  return e;
}

HSSMatrix<double>::HSSMatrix(std::size_t m, std::size_t n, const opts_t& opts) : rows_(m), cols_(n) {
  if (m != n) throw std::invalid_argument("HSS compression only supported for square matrices.");
  make_engine(opts, nullptr);
}
HSSMatrix<double>::HSSMatrix(const structured::ClusterTree& t, const opts_t& opts) : rows_(t.size), cols_(t.size) {
  tree_.reset(new structured::ClusterTree(t));
  make_engine(opts, tree_.get());
}
HSSMatrix<double>::~HSSMatrix() {}

void HSSMatrix<double>::write(const std::string& fname) const {
  owner("write");
  if (!eng_) throw std::invalid_argument("write: empty matrix");
  std::ofstream f(fname, std::ios::out | std::ios::trunc | std::ios::binary);
  if (!f) throw std::runtime_error("write: cannot open " + fname);
  eng_->save(f);
}
HSSMatrix<double> HSSMatrix<double>::read(const std::string& fname) {
  std::ifstream f(fname, std::ios::in | std::ios::binary);
  if (!f) throw std::runtime_error("read: cannot open " + fname);
  HSSMatrix<double> H;
  opts_t o;
  H.eng_ = DeviceHSS::load(f, engine_options(o));
  H.rows_ = H.cols_ = H.eng_->rows();
  return H;
}

// HSSMatrix(kernel::Kernel&, opts): HSS/HSSMatrix.cpp:88-106
HSSMatrix<double>::HSSMatrix(kernel::Kernel<double>& K, const opts_t& opts) : HSSMatrix(K, opts, 1, 0, nullptr, nullptr) {}
static CommSpec callback_group(int world, int rank, void (*fn)(void*, void*, long long), void* user) {
  CommSpec pg;
  pg.world = world; pg.rank = rank; pg.allgather = fn; pg.user = user; pg.native = false;
  return pg;
}
HSSMatrix<double>::HSSMatrix(kernel::Kernel<double>& K, const opts_t& opts, int world, int rank,
                             void (*fn)(void*, void*, long long), void* user) : HSSMatrix(K, opts, callback_group(world, rank, fn, user)) {}
HSSMatrix<double>::HSSMatrix(kernel::Kernel<double>& K, const opts_t& opts, const CommSpec& pg) : rows_(K.n()), cols_(K.n()) {
  auto tc0 = std::chrono::steady_clock::now();
  // (cobble / kd of a large point set: on the device, one launch per tree level -- host/Clustering.hpp, kernels/hssk_cluster.hip)
  bool on_device = false;
  DevicePoints dpts;   // (the points the device clustering reordered stay there for the compression: no second upload)
  struct Release { DevicePoints& d; ~Release() { d.release(); } } release_dpts{dpts};
  auto t = binary_tree_clustering(opts.clustering_algorithm(), K.data(), K.permutation(), opts.leaf_size(), engine_options(opts).device, &on_device, &dpts);
  K.permute();
  if (opts.verbose())
    std::cout << "# clustering (" << get_name(opts.clustering_algorithm()) << (on_device ? ", device" : ", host") << ") time = "
              << std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count() << std::endl;
  tree_.reset(new structured::ClusterTree(t));
  EngineOptions e = engine_options(opts);
  pg.apply(e);
  eng_.reset(new DeviceHSS(int(rows_), e, tree_.get()));
  device_points_ = dpts.X();
  try { compress(K, opts); } catch (...) { device_points_ = nullptr; throw; }
  device_points_ = nullptr;
}
void HSSMatrix<double>::compress(const kernel::Kernel<double>& K, const opts_t& opts) {
  compress_with_neighbors(K, opts, K.neighbors(), K.neighbor_count());
}
void HSSMatrix<double>::compress_with_neighbors(const kernel::Kernel<double>& K, const opts_t& opts, const int* ann, int k) {
  if (K.n() != rows_) throw std::invalid_argument("compress: kernel size does not match");
  if (K.data().ld() != int(K.d())) throw std::invalid_argument("compress(Kernel): the point matrix must be contiguous");
  make_engine(opts, tree_.get());
  DeviceHSS::KernelSpec ks;
  ks.X = K.data().data(); ks.dX = device_points_; ks.d = int(K.d()); ks.type = K.device_type(); ks.p = K.degree();
  ks.h = K.width(); ks.lambda = K.lambda(); ks.ann = std::min<int>(int(K.n()), opts.approximate_neighbors());
  // a user-defined Kernel subclass (kernel/Kernel.hpp:73-170: only its virtual eval is known): blocks evaluated on the host
  if (K.device_type() < 0) {
    ks.type = 0;
    const kernel::Kernel<double>* Kp = &K;
    ks.eval = [Kp](int i, int j) { return Kp->eval((std::size_t)i, (std::size_t)j); };
  }
  if (opts.neighbor_search() == NeighborSearch::ANN) {
    const double* X = K.data().data();
    const std::size_t dim = K.d(), n = K.n(), iters = opts.ann_iterations();
    ks.neighbors = [X, dim, n, iters](int kk, int* out) {
      // leaves of a tree sample are independent: brute-force them on the host's threads
      auto parfor = [](std::size_t cnt, const std::function<void(std::size_t)>& f) {
        const unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
        std::atomic<std::size_t> next{0};
        std::vector<std::thread> th;
        for (unsigned t = 0; t < std::min<std::size_t>(nt, cnt); t++)
          th.emplace_back([&] { for (std::size_t i = next++; i < cnt; i = next++) f(i); });
        for (auto& t : th) t.join();
      };
      auto L = ann_detail::search(X, dim, n, iters, (std::size_t)kk, parfor);
      for (std::size_t i = 0; i < L.id.size(); i++) out[i] = (int)L.id[i];
    };
  }
  eng_->compress_kernel(ks, ann, k);
}

void HSSMatrix<double>::make_engine(const opts_t& opts, const structured::ClusterTree* t) {
  owner("compress");
  EngineOptions e = engine_options(opts);
  if (eng_ && eng_->options().leaf_size == e.leaf_size && eng_->options().device == e.device) {
    const EngineOptions& cur = eng_->options();   // same tree: keep the engine (device context, process group), new knobs
    e.world = cur.world; e.rank = cur.rank; e.allgather = cur.allgather; e.comm_user = cur.comm_user;
    e.allgather_stream = cur.allgather_stream; e.allreduce_stream = cur.allreduce_stream; e.reduce_scatter_stream = cur.reduce_scatter_stream;
    eng_->set_options(e);
    return;
  }
  eng_.reset(new DeviceHSS(int(rows_), e, t));
}

void HSSMatrix<double>::compress(const DenseM_t& A, const opts_t& opts) {
  if (A.rows() != rows_ || A.cols() != cols_) throw std::invalid_argument("compress: matrix dimensions do not match");
  make_engine(opts, tree_.get());
  eng_->compress_dense_host(A.data(), A.ld());
}
void HSSMatrix<double>::compress_image(const void* A, std::size_t lda, int dtype, const opts_t& opts) {
  make_engine(opts, tree_.get());
  eng_->compress_dense_host_typed(A, (long long)lda, dtype);
}
void HSSMatrix<double>::compress_device(const double* dA, long long lda, const opts_t& opts) {
  make_engine(opts, tree_.get());
  eng_->compress_dense_device(dA, lda);
}
void HSSMatrix<double>::compress_device_sharded(const double* dA, long long lda, const opts_t& opts, int world, int rank,
                                                void (*fn)(void*, void*, long long), void* user) {
  compress_device_sharded(dA, lda, opts, callback_group(world, rank, fn, user));
}
void HSSMatrix<double>::compress_device_sharded(const double* dA, long long lda, const opts_t& opts, const CommSpec& pg) {
  owner("compress");
  EngineOptions e = engine_options(opts);
  pg.apply(e);
  eng_.reset(new DeviceHSS(int(rows_), e, tree_.get()));
  eng_->compress_dense_device(dA, lda);
}
void HSSMatrix<double>::compress_generator(int kind, const opts_t& opts) {
  make_engine(opts, tree_.get());
  hssk_gen g{kind, 0, {0., 0., 0., 0.}};
  eng_->compress_generator(g);
}
void HSSMatrix<double>::compress_generator(int kind, const opts_t& opts, const CommSpec& pg) {
  owner("compress");
  EngineOptions e = engine_options(opts);
  pg.apply(e);
  eng_.reset(new DeviceHSS(int(rows_), e, tree_.get()));
  hssk_gen g{kind, 0, {0., 0., 0., 0.}};
  eng_->compress_generator(g);
}
void HSSMatrix<double>::compress_device_blocks(const double* dRows, long long ldr, const double* dCols, long long ldc,
                                               const opts_t& opts, const CommSpec& pg) {
  owner("compress");
  EngineOptions e = engine_options(opts);
  pg.apply(e);
  eng_.reset(new DeviceHSS(int(rows_), e, tree_.get()));
  eng_->compress_dense_device_sharded(dRows, ldr, dCols, ldc);
}
void HSSMatrix<double>::compress(const mult_t& Amult, const elem_t& Aelem, const opts_t& opts) {
  make_engine(opts, tree_.get());
  const int N = int(rows_);
  if (opts.user_defined_random()) {
    struct Flag {   // on for this call only, whatever way it ends
      DeviceHSS* h;
      explicit Flag(DeviceHSS* h_) : h(h_) { EngineOptions e = h->options(); e.user_random = true; h->set_options(e); }
      ~Flag() { EngineOptions e = h->options(); e.user_random = false; h->set_options(e); }
    } flag(eng_.get());
    // Amult fills Rr and Rc as well as Sr and Sc (the sparse HSS fronts sample their children this way,
    // sparse/fronts/FrontHSS.cpp:367-385).  The engine keeps ONE random block for the row and the column samples, as every
    // caller in the reference does (Rc is a copy of Rr); distinct blocks are refused rather than silently merged.
    host_sample_t us = [&](int n, int nrhs, double* R, double* Sr, double* Sc) {
      DenseMW_t Rr(n, nrhs, R, n), Srw(n, nrhs, Sr, n), Scw(n, nrhs, Sc, n);
      DenseM_t Rc(n, nrhs);
      Rr.zero();
      Amult(Rr, Rc, Srw, Scw);
      for (int j = 0; j < nrhs; j++)
        for (int i = 0; i < n; i++)
          if (Rc(i, j) != Rr(i, j)) throw std::invalid_argument("user_defined_random: Rr and Rc must be the same random block");
    };
    host_elem_t he0 = [&](int m, const int* I, int n, const int* J, double* B, int ldb) {
      std::vector<std::size_t> Iv(I, I + m), Jv(J, J + n);
      DenseM_t Bm(m, n);
      Aelem(Iv, Jv, Bm);
      for (int j = 0; j < n; j++) std::memcpy(B + (size_t)j * ldb, Bm.ptr(0, j), sizeof(double) * m);
    };
    eng_->compress_callbacks_user_random(us, he0);
    return;
  }
  // the reference hands both sample blocks to the user in one call (HSSExtra.hpp:231-239); the
  // engine asks for the two products separately, so cache the pair
  DenseM_t Sr_cache, Sc_cache;
  host_mult_t hm = [&](char trans, int n, int nrhs, const double* R, int ldr, double* S, int lds) {
    if (trans == 'N') {
      DenseM_t Rr(n, nrhs, R, ldr), Rc(Rr);
      Sr_cache = DenseM_t(n, nrhs);
      Sc_cache = DenseM_t(n, nrhs);
      Amult(Rr, Rc, Sr_cache, Sc_cache);
    }
    const DenseM_t& src = trans == 'N' ? Sr_cache : Sc_cache;
    for (int j = 0; j < nrhs; j++) std::memcpy(S + (size_t)j * lds, src.ptr(0, j), sizeof(double) * n);
  };
  host_elem_t he = [&](int m, const int* I, int n, const int* J, double* B, int ldb) {
    std::vector<std::size_t> Iv(I, I + m), Jv(J, J + n);
    DenseM_t Bm(m, n);
    Aelem(Iv, Jv, Bm);
    for (int j = 0; j < n; j++) std::memcpy(B + (size_t)j * ldb, Bm.ptr(0, j), sizeof(double) * m);
  };
  (void)N;
  eng_->compress_callbacks(hm, he);
}

void HSSMatrix<double>::compress_from_elements(const elem_t& Aelem, const opts_t& opts) {
  make_engine(opts, tree_.get());
  const std::size_t N = rows_;
  host_elem_t he = [&](int m, const int* I, int n, const int* J, double* B, int ldb) {
    std::vector<std::size_t> Iv(I, I + m), Jv(J, J + n);
    DenseM_t Bm(m, n);
    Aelem(Iv, Jv, Bm);
    for (int j = 0; j < n; j++) std::memcpy(B + (size_t)j * ldb, Bm.ptr(0, j), sizeof(double) * m);
  };
  // A(:, c0:c1) in 1024 x 1024 tiles, tiles in parallel on the host threads (the reference evaluates its tiles from
  // concurrent OpenMP tasks, StructuredMatrix.cpp:226-233)
  DeviceHSS::host_fill_t fill = [&](long long c0, long long c1, double* dst) {
    const std::size_t T = 1024;
    const std::size_t tr = (N + T - 1) / T, tc = (std::size_t(c1 - c0) + T - 1) / T;
    std::atomic<std::size_t> next{0};
    std::exception_ptr err;
    std::mutex mu;
    auto work = [&] {
      try {
        for (std::size_t t = next++; t < tr * tc; t = next++) {
          const std::size_t i0 = (t % tr) * T, j0 = std::size_t(c0) + (t / tr) * T;
          const std::size_t mb = std::min(T, N - i0), nbk = std::min(T, std::size_t(c1) - j0);
          std::vector<std::size_t> I(mb), J(nbk);
          for (std::size_t i = 0; i < mb; i++) I[i] = i0 + i;
          for (std::size_t j = 0; j < nbk; j++) J[j] = j0 + j;
          DenseM_t Tm(mb, nbk);
          Aelem(I, J, Tm);
          for (std::size_t j = 0; j < nbk; j++) std::memcpy(dst + i0 + (j0 - std::size_t(c0) + j) * N, Tm.ptr(0, j), sizeof(double) * mb);
        }
      } catch (...) { std::lock_guard<std::mutex> g(mu); err = std::current_exception(); }
    };
    const unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (unsigned t = 1; t < std::min<std::size_t>(nt, tr * tc); t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (err) std::rethrow_exception(err);
  };
  eng_->compress_host_blocks(fill, he);
}

std::size_t HSSMatrix<double>::memory() const {
  if (veng_) return std::size_t(veng_->memory(vnode_));
  return eng_ ? std::size_t(eng_->memory()) : 0;
}
std::size_t HSSMatrix<double>::nonzeros() const {
  if (veng_) return std::size_t(veng_->nonzeros(vnode_));
  return eng_ ? std::size_t(eng_->nonzeros()) : 0;
}
std::size_t HSSMatrix<double>::rank() const {
  if (veng_) return std::size_t(veng_->rank(vnode_));
  return eng_ ? std::size_t(eng_->rank()) : 0;
}
std::size_t HSSMatrix<double>::levels() const {
  if (veng_) return std::size_t(veng_->nodes()[vnode_].height + 1);
  return eng_ ? std::size_t(eng_->levels()) : 0;
}
bool HSSMatrix<double>::is_compressed() const {
  if (veng_) return veng_->nodes()[vnode_].compressed();
  return eng_ && eng_->is_compressed();
}
bool HSSMatrix<double>::leaf() const {
  if (veng_) return veng_->nodes()[vnode_].leaf();
  return !eng_ || eng_->num_nodes() == 1;
}

void HSSMatrix<double>::mult(Trans op, const DenseM_t& x, DenseM_t& y) const {
  apply_HSS(op, *this, x, 0., y);
}
DenseMatrix<double> HSSMatrix<double>::apply(const DenseM_t& b) const {
  DenseM_t c(rows_, b.cols());
  apply_HSS(Trans::N, *this, b, 0., c);
  return c;
}
DenseMatrix<double> HSSMatrix<double>::applyC(const DenseM_t& b) const {
  DenseM_t c(cols_, b.cols());
  apply_HSS(Trans::C, *this, b, 0., c);
  return c;
}
// (a child view: the ULV factors of its subtree with the child as the root -- "a child of an HSS matrix is itself an HSS
//  matrix", HSSMatrix.hpp:194-202.  They live in the nodes, as the reference's ULV_ members do: factoring the whole matrix
//  afterwards replaces them, and the other way round.)
void HSSMatrix<double>::factor() {
  if (veng_) veng_->factor_node(vnode_);
  else eng_->factor();
}
void HSSMatrix<double>::solve(DenseM_t& b) const {
  if (b.rows() != rows_) throw std::invalid_argument("solve: right-hand side has the wrong number of rows");
  if (veng_) veng_->solve_node(vnode_, int(b.cols()), b.data(), b.ld(), false);
  else eng_->solve(int(b.cols()), b.data(), b.ld(), false);
}
std::size_t HSSMatrix<double>::factor_nonzeros() const {
  DeviceHSS* e = engine();
  if (!e) return 0;
  const auto& nd = e->nodes();
  const bool whole = e->node_is_factored(0);
  const bool partial = e->is_partially_factored() && !nd[0].leaf();
  // the node the factorization in place is rooted at: the whole tree, child 0 (partial_factor), or a child factored on its own
  int root = -1;
  if (whole) root = 0;
  else if (partial) root = nd[0].c0;
  else for (std::size_t i = 1; i < nd.size(); i++) if (e->node_is_factored(int(i))) root = int(i);
  if (root < 0) return 0;
  std::size_t nnz = 0;
  std::function<void(int, bool)> walk = [&](int i, bool counted) {   // counted: inside the factored subtree
    const auto& n = nd[i];
    const bool in = counted || i == root;
    if (in) {
      if (i == root) {
        const std::size_t mu = n.leaf() ? std::size_t(n.m) : std::size_t(nd[n.c0].rU + nd[n.c1].rU);
        nnz += mu * mu + (partial && !whole ? mu * std::size_t(n.rV) : 0);
      } else if (n.mU > n.rU) {
        const std::size_t m = n.mU, r = n.rU, q = m - r, rv = n.rV;
        nnz += q * q + q * rv + r * m + m * m;
      }
    }
    if (!n.leaf()) { walk(n.c0, in); walk(n.c1, in); }
  };
  // a view reports the factors of its own nodes: the factored subtree lies inside the view (count from its root on), the view
  // lies inside the factored subtree (every node of the view counts), or they are disjoint
  auto below = [&](int i, int a) { while (i > 0 && i != a) i = nd[i].parent; return i == a; };   // is a an ancestor of (or equal to) i
  const bool view_inside = below(vnode_, root) && vnode_ != root;
  if (!below(root, vnode_) && !view_inside) return 0;
  walk(vnode_, view_inside);
  return nnz;
}
void HSSMatrix<double>::forward_solve(WorkSolve<double>& w, const DenseM_t& b, bool partial) const {
  DeviceHSS* e = engine();
  if (!e) throw std::logic_error("forward_solve: empty matrix");
  if (b.rows() != rows_) throw std::invalid_argument("forward_solve: right-hand side has the wrong number of rows");
  const auto& nd = e->nodes()[vnode_];
  const std::size_t mu = nd.leaf() ? std::size_t(nd.m) : std::size_t(e->nodes()[nd.c0].rU + e->nodes()[nd.c1].rU);
  w.x = DenseM_t(mu, b.cols());
  w.reduced_rhs = partial ? DenseM_t(std::size_t(nd.rV), b.cols()) : DenseM_t();
  e->forward_solve_node(vnode_, w.state_, int(b.cols()), b.data(), b.ld(), partial, w.x.data(), w.x.ld(),
                        partial ? w.reduced_rhs.data() : nullptr, w.reduced_rhs.ld());
}
void HSSMatrix<double>::backward_solve(WorkSolve<double>& w, DenseM_t& x) const {
  DeviceHSS* e = engine();
  if (!e) throw std::logic_error("backward_solve: empty matrix");
  if (x.rows() != rows_ || int(x.cols()) != w.state_.nrhs) throw std::invalid_argument("backward_solve: the solution block has the wrong shape");
  e->backward_solve_node(vnode_, w.state_, w.x.data(), w.x.ld(), x.data(), x.ld());
  w.x = DenseM_t();
}
void HSSMatrix<double>::shift(scalar_t sigma) { owner("shift"); eng_->shift(sigma); }
void HSSMatrix<double>::mult_device(Trans op, int nrhs, const double* dx, long long ldx, double* dy, long long ldy, double beta) const {
  if (veng_) veng_->mult_node(vnode_, op == Trans::N ? 'N' : 'C', nrhs, dx, ldx, dy, ldy, true, beta);
  else eng_->mult(op == Trans::N ? 'N' : 'C', nrhs, dx, ldx, dy, ldy, true, beta);
}
void HSSMatrix<double>::solve_device(int nrhs, double* db, long long ldb) const {
  if (veng_) veng_->solve_node(vnode_, nrhs, db, ldb, true);
  else eng_->solve(nrhs, db, ldb, true);
}

DenseMatrix<double> HSSMatrix<double>::dense() const {
  // H * I in column blocks (HSSMatrix.cpp:188-260 re-expands recursively; test-only, O(N^2 r))
  DenseM_t D(rows_, cols_);
  const std::size_t bs = 256;
  for (std::size_t j0 = 0; j0 < cols_; j0 += bs) {
    std::size_t nb = std::min(bs, cols_ - j0);
    DenseM_t E(cols_, nb);
    for (std::size_t j = 0; j < nb; j++) E(j0 + j, j) = 1.;
    DenseMW_t Dj(rows_, nb, D, 0, j0);
    apply_HSS(Trans::N, *this, E, 0., Dj);
  }
  return D;
}

// H(I, J) (HSSMatrix.extract.hpp:36-104).  Small requests -- the sub-blocks a front's assembly asks for -- walk the tree on the
// device (DeviceHSS::extract_blocks: O(r^2 (|I| + |J|) log N)); requests that cover a large part of the matrix (dense()
// through extract, whole block rows) are cheaper as |J| products with the matrix
DenseMatrix<double> HSSMatrix<double>::extract(const std::vector<std::size_t>& I, const std::vector<std::size_t>& J) const {
  DenseM_t B(I.size(), J.size());
  for (std::size_t i = 0; i < I.size(); i++) if (I[i] >= rows_) throw std::invalid_argument("extract: row index out of range");
  for (std::size_t j = 0; j < J.size(); j++) if (J[j] >= cols_) throw std::invalid_argument("extract: column index out of range");
  if (I.empty() || J.empty()) return B;
  DeviceHSS* e = engine();
  if (e->extract_by_traversal_ok(vnode_, (long long)I.size(), (long long)J.size())) {
    std::vector<int> r(I.begin(), I.end()), c(J.begin(), J.end());
    const int roff[2] = {0, (int)r.size()}, coff[2] = {0, (int)c.size()};
    double* out = B.data();
    const int ld = B.ld();
    e->extract_blocks(vnode_, 1, r.data(), roff, c.data(), coff, &out, &ld, false, false);
    return B;
  }
  DenseM_t E(cols_, J.size()), HE(rows_, J.size());
  for (std::size_t j = 0; j < J.size(); j++) E(J[j], j) = 1.;
  apply_HSS(Trans::N, *this, E, 0., HE);
  for (std::size_t j = 0; j < J.size(); j++)
    for (std::size_t i = 0; i < I.size(); i++) B(i, j) = HE(I[i], j);
  return B;
}
double HSSMatrix<double>::get(std::size_t i, std::size_t j) const { return extract({i}, {j})(0, 0); }
void HSSMatrix<double>::extract_add(const std::vector<std::size_t>& I, const std::vector<std::size_t>& J, DenseM_t& B) const {
  if (B.rows() != I.size() || B.cols() != J.size()) throw std::invalid_argument("extract_add: B has the wrong shape");
  DenseM_t E = extract(I, J);
  for (std::size_t j = 0; j < J.size(); j++)
    for (std::size_t i = 0; i < I.size(); i++) B(i, j) += E(i, j);
}
// a batch of requests in one pair of launches (extension): B[b] (+)= H(I[b], J[b])
void HSSMatrix<double>::extract_blocks(const std::vector<std::vector<std::size_t>>& I, const std::vector<std::vector<std::size_t>>& J,
                                       std::vector<DenseM_t>& B, bool add) const {
  if (I.size() != J.size()) throw std::invalid_argument("extract_blocks: as many row sets as column sets");
  const int nb = (int)I.size();
  if (!add) { B.clear(); for (int b = 0; b < nb; b++) B.emplace_back(I[b].size(), J[b].size()); }
  if ((int)B.size() != nb) throw std::invalid_argument("extract_blocks: one output per request");
  std::vector<int> r, c, roff(nb + 1, 0), coff(nb + 1, 0), ld(nb);
  std::vector<double*> out(nb);
  for (int b = 0; b < nb; b++) {
    if (B[b].rows() != I[b].size() || B[b].cols() != J[b].size()) throw std::invalid_argument("extract_blocks: output of the wrong shape");
    r.insert(r.end(), I[b].begin(), I[b].end());
    c.insert(c.end(), J[b].begin(), J[b].end());
    roff[b + 1] = (int)r.size(); coff[b + 1] = (int)c.size();
    out[b] = B[b].data(); ld[b] = B[b].ld();
  }
  if (nb) engine()->extract_blocks(vnode_, nb, r.data(), roff.data(), c.data(), coff.data(), out.data(), ld.data(), false, add);
}

// ---- Schur complement of the (0,0) block (HSSMatrix.Schur.hpp) ----------------------------------------------
void HSSMatrix<double>::partial_factor() { owner("partial_factor"); eng_->partial_factor(); }
void HSSMatrix<double>::Schur_update(DenseM_t& Theta, DenseM_t& DUB01, DenseM_t& Phi) const {
  owner("Schur_update");
  if (leaf()) return;
  const auto d = eng_->schur_dims();
  Theta = DenseM_t(d.n1, d.rV0);
  DUB01 = DenseM_t(d.mu0, d.rV1);
  Phi = DenseM_t(d.n1, d.mu0);
  eng_->schur_update(Theta.data(), Theta.ld(), DUB01.data(), DUB01.ld(), Phi.data(), Phi.ld(), nullptr, 1);
}
DenseMatrix<double> HSSMatrix<double>::Vhat() const {
  owner("Vhat");
  if (leaf() || !eng_->is_partially_factored()) throw std::logic_error("Vhat: partial_factor() has not been called");
  const auto d = eng_->schur_dims();
  DenseM_t V(d.mu0, d.rV0);
  const auto& a = eng_->nodes()[eng_->nodes()[0].c0];
  if (d.mu0 && d.rV0)
    if (hssk_memcpy2d_d2h(eng_->ctx(), V.data(), sizeof(double) * V.ld(), a.Vt0, sizeof(double) * d.mu0, sizeof(double) * d.mu0, d.rV0))
      throw std::runtime_error(hssk_last_error());
  return V;
}
void HSSMatrix<double>::Schur_product_direct(const DenseM_t& Theta, const DenseM_t& DUB01, const DenseM_t& Phi,
                                             const DenseM_t&, const DenseM_t& R, DenseM_t& Sr, DenseM_t& Sc) const {
  owner("Schur_product_direct");
  const auto d = eng_->schur_dims();
  if (Theta.rows() != std::size_t(d.n1) || Theta.cols() != std::size_t(d.rV0) || DUB01.rows() != std::size_t(d.mu0) ||
      DUB01.cols() != std::size_t(d.rV1) || Phi.rows() != std::size_t(d.n1) || Phi.cols() != std::size_t(d.mu0))
    throw std::invalid_argument("Schur_product_direct: Theta / DUB01 / Phi are not the matrices returned by Schur_update");
  if (R.rows() != std::size_t(d.n1)) throw std::invalid_argument("Schur_product_direct: R has the wrong number of rows");
  if (Sr.rows() != R.rows() || Sr.cols() != R.cols()) Sr = DenseM_t(R.rows(), R.cols());
  if (Sc.rows() != R.rows() || Sc.cols() != R.cols()) Sc = DenseM_t(R.rows(), R.cols());
  eng_->schur_product_direct(int(R.cols()), R.data(), R.ld(), Sr.data(), Sr.ld(), Sc.data(), Sc.ld(), false);
}
void HSSMatrix<double>::Schur_product_indirect(const DenseM_t& DUB01, const DenseM_t& R0, const DenseM_t& R1,
                                               const DenseM_t& Sr1, const DenseM_t& Sc1, DenseM_t& Sr, DenseM_t& Sc) const {
  owner("Schur_product_indirect");
  if (leaf()) return;
  const auto d = eng_->schur_dims();
  if (DUB01.rows() != std::size_t(d.mu0) || DUB01.cols() != std::size_t(d.rV1))
    throw std::invalid_argument("Schur_product_indirect: DUB01 is not the matrix returned by Schur_update");
  if (R0.rows() != std::size_t(d.n0) || R1.rows() != std::size_t(d.n1) || R0.cols() != R1.cols() ||
      Sr1.rows() != R1.rows() || Sc1.rows() != R1.rows() || Sr1.cols() != R1.cols() || Sc1.cols() != R1.cols())
    throw std::invalid_argument("Schur_product_indirect: operand shapes do not match");
  Sr = DenseM_t(R1.rows(), R1.cols());
  Sc = DenseM_t(R1.rows(), R1.cols());
  eng_->schur_product_indirect(int(R1.cols()), R0.data(), R0.ld(), R1.data(), R1.ld(), Sr1.data(), Sr1.ld(), Sc1.data(),
                               Sc1.ld(), Sr.data(), Sr.ld(), Sc.data(), Sc.ld(), false);
}
DenseMatrix<double> HSSMatrix<double>::apply_child(int c, Trans op, const DenseM_t& x) const {
  owner("apply_child");
  if (leaf()) throw std::logic_error("apply_child: the matrix is a single leaf");
  const auto d = eng_->schur_dims();
  const std::size_t n = c == 0 ? d.n0 : d.n1;
  if (x.rows() != n) throw std::invalid_argument("apply_child: x has the wrong number of rows");
  DenseM_t y(n, x.cols());
  eng_->mult_child(c, op == Trans::N ? 'N' : 'C', int(x.cols()), x.data(), x.ld(), y.data(), y.ld(), false);
  return y;
}

// ---- child views (HSS/HSSMatrix.hpp:194-202) ----------------------------------------------------------------
HSSMatrix<double>::HSSMatrix(DeviceHSS* parent_engine, int node) : veng_(parent_engine), vnode_(node) {
  if (!parent_engine || node < 0 || node >= parent_engine->num_nodes()) throw std::invalid_argument("HSSMatrix view: no such node");
  rows_ = cols_ = parent_engine->nodes()[node].m;
}
const HSSMatrix<double>* HSSMatrix<double>::child(int c) const {
  if (leaf()) throw std::logic_error("child: the matrix is a leaf");
  if (c != 0 && c != 1) throw std::invalid_argument("child: c must be 0 or 1");
  DeviceHSS* e = engine();
  const auto& root = e->nodes()[vnode_];
  const int id = c == 0 ? root.c0 : root.c1;
  if (!ch_[c] || ch_[c]->vnode_ != id || ch_[c]->veng_ != e) ch_[c].reset(new HSSMatrix<double>(e, id));
  return ch_[c].get();
}
HSSMatrix<double>* HSSMatrix<double>::child(int c) { return const_cast<HSSMatrix<double>*>(static_cast<const HSSMatrix<double>*>(this)->child(c)); }
std::size_t HSSMatrix<double>::U_rank() const { return engine() ? engine()->nodes()[vnode_].rU : 0; }
std::size_t HSSMatrix<double>::V_rank() const { return engine() ? engine()->nodes()[vnode_].rV : 0; }
std::size_t HSSMatrix<double>::U_rows() const { return engine() && engine()->nodes()[vnode_].lvl ? engine()->nodes()[vnode_].mU : 0; }
std::size_t HSSMatrix<double>::V_rows() const { return engine() && engine()->nodes()[vnode_].lvl ? engine()->nodes()[vnode_].mV : 0; }
DenseMatrix<double> HSSMatrix<double>::Factors::Vhat() const {
  DeviceHSS* eng = self->engine();
  if (!eng) throw std::logic_error("Vhat: empty matrix");
  const auto& root = eng->nodes()[0];
  if (root.leaf() || self->vnode_ != root.c0 || !eng->is_partially_factored())
    throw std::logic_error("Vhat: only child(0) carries it, after partial_factor()");
  const auto d = eng->schur_dims();
  DenseM_t V(d.mu0, d.rV0);
  const auto& a = eng->nodes()[root.c0];
  if (d.mu0 && d.rV0)
    if (hssk_memcpy2d_d2h(eng->ctx(), V.data(), sizeof(double) * V.ld(), a.Vt0, sizeof(double) * d.mu0, sizeof(double) * d.mu0, d.rV0))
      throw std::runtime_error(hssk_last_error());
  return V;
}

void HSSMatrix<double>::print_info(std::ostream& out, std::size_t roff, std::size_t coff) const {
  const DeviceHSS* e = engine();
  if (!e) return;
  const int lo0 = e->nodes()[vnode_].lo;
  for (int i = vnode_, end = e->node_end(vnode_); i < end; i++) {  // pre-order, same line format as HSSMatrix.cpp:344-350
    const auto& nd = e->nodes()[i];
    out << "SEQ rank=0 b = [" << roff + nd.lo - lo0 << "," << roff + nd.lo - lo0 + nd.m << " x " << coff + nd.lo - lo0 << ","
        << coff + nd.lo - lo0 + nd.m << "]  U = " << (nd.lvl ? nd.mU : 0) << " x " << nd.rU << " V = "
        << (nd.lvl ? nd.mV : 0) << " x " << nd.rV << (nd.leaf() ? " leaf" : " non-leaf") << std::endl;
  }
}

void HSSMatrix<double>::draw(std::ostream& of, std::size_t rlo, std::size_t clo) const {
  const DeviceHSS* e = engine();
  if (!e) return;
  // pre-order walk of the sub-tree rooted at this matrix: same rectangles and colours as the reference's recursion
  std::function<void(int, std::size_t, std::size_t)> rec = [&](int id, std::size_t r0, std::size_t c0) {
    const auto& nd = e->nodes()[id];
    if (nd.leaf()) {
      of << "set obj rect from " << r0 << ", " << c0 << " to " << r0 + nd.m << ", " << c0 + nd.m << " fc rgb 'red'" << std::endl;
      return;
    }
    const auto &a = e->nodes()[nd.c0], &b = e->nodes()[nd.c1];
    const int rank0 = std::max(a.rU, b.rV), rank1 = std::max(b.rU, a.rV), minmn = std::max(nd.m, 1);
    auto colour = [&](int rk) {
      const int red = int(std::floor(255.0 * rk / minmn)), blue = 255 - red;
      char buf[16];
      std::snprintf(buf, sizeof(buf), "%02x00%02x", std::max(0, std::min(255, red)), std::max(0, std::min(255, blue)));
      return std::string(buf);
    };
    of << "set obj rect from " << r0 << ", " << c0 + a.m << " to " << r0 + a.m << ", " << c0 + nd.m << " fc rgb '#" << colour(rank0) << "'" << std::endl;
    of << "set obj rect from " << r0 + a.m << ", " << c0 << " to " << r0 + nd.m << ", " << c0 + a.m << " fc rgb '#" << colour(rank1) << "'" << std::endl;
    rec(nd.c0, r0, c0);
    rec(nd.c1, r0 + a.m, c0 + a.m);
  };
  rec(vnode_, rlo, clo);
}
void draw(const HSSMatrix<double>& H, const std::string& name) {
  std::ofstream of("plot" + name + ".gnuplot");
  of << "set terminal pdf enhanced color size 5,4" << std::endl;
  of << "set output '" << name << ".pdf'" << std::endl;
  H.draw(of);
  of << "set xrange [0:" << H.cols() << "]" << std::endl;
  of << "set yrange [" << H.rows() << ":0]" << std::endl;
  of << "plot x lt -1 notitle" << std::endl;
}
std::unique_ptr<HSSMatrix<double>> HSSMatrix<double>::clone() const {
  owner("clone");
  if (!eng_) return std::unique_ptr<HSSMatrix<double>>(new HSSMatrix<double>());
  std::stringstream ss(std::ios::in | std::ios::out | std::ios::binary);
  eng_->save(ss);
  std::unique_ptr<HSSMatrix<double>> H(new HSSMatrix<double>());
  H->eng_ = DeviceHSS::load(ss, eng_->options());
  H->rows_ = H->cols_ = H->eng_->rows();
  return H;
}
void HSSMatrix<double>::reset() {
  owner("reset");
  if (eng_) eng_->reset();
  ch_[0].reset(); ch_[1].reset();
  trailing_deleted_ = false;
}

void apply_HSS(Trans op, const HSSMatrix<double>& A, const DenseMatrix<double>& B, double beta, DenseMatrix<double>& C) {
  if (B.rows() != A.rows() || C.rows() != A.rows() || B.cols() != C.cols())
    throw std::invalid_argument("apply_HSS: dimension mismatch");
  if (A.trailing_block_deleted()) throw std::logic_error("apply_HSS: the trailing block of this matrix has been deleted (delete_trailing_block)");
  if (A.is_view()) A.engine()->mult_node(A.node(), op == Trans::N ? 'N' : 'C', int(B.cols()), B.data(), B.ld(), C.data(), C.ld(), false, beta);
  else A.engine()->mult(op == Trans::N ? 'N' : 'C', int(B.cols()), B.data(), B.ld(), C.data(), C.ld(), false, beta);
}

}  // namespace HSS
}  // namespace strumpack
