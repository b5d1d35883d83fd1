// Binary-tree clustering of point sets: the row/column ordering and the cluster tree of a kernel matrix
// (reference: clustering/Clustering.hpp:143-168 binary_tree_clustering, clustering/KMeans.cpp,
// clustering/KDTree.cpp, clustering/CobblePartitioning.cpp).
//
// Host code on purpose: it runs once per matrix over d x n doubles (n log n work) and, more importantly,
// the reference's trees are defined through libstdc++'s std::mt19937 / std::uniform_int_distribution /
// std::discrete_distribution / std::nth_element; using the same library calls on the same data reproduces
// the reference's permutation exactly (tests/test_kernel_*.py compare against fixtures made by the reference).
//
// All partitioners have one shape -- label every point 0 / 1, move the 0-points to the front with the
// reference's swap sequence (which also defines the order inside each half), recurse -- so only the labelling
// differs between them.
#pragma once
#include <atomic>
#include <memory>
#include <system_error>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <numeric>
#include <random>
#include <exception>
#include <string>
#include <thread>
#include <vector>

#include "ClusterTree.hpp"
#include "DenseMatrix.hpp"
#include "DevicePool.hpp"
#include "hssk.h"

namespace strumpack {

enum class ClusteringAlgorithm { NATURAL, TWO_MEANS, KD_TREE, PCA, COBBLE };

inline std::string get_name(ClusteringAlgorithm c) {
  switch (c) {
    case ClusteringAlgorithm::NATURAL: return "natural";
    case ClusteringAlgorithm::TWO_MEANS: return "2means";
    case ClusteringAlgorithm::KD_TREE: return "kdtree";
    case ClusteringAlgorithm::PCA: return "PCA";
    case ClusteringAlgorithm::COBBLE: return "cobble";
  }
  return "unknown";
}

inline ClusteringAlgorithm get_clustering_algorithm(const std::string& c) {
  if (c == "natural") return ClusteringAlgorithm::NATURAL;
  if (c == "2means") return ClusteringAlgorithm::TWO_MEANS;
  if (c == "kdtree") return ClusteringAlgorithm::KD_TREE;
  if (c == "pca") return ClusteringAlgorithm::PCA;
  if (c == "cobble") return ClusteringAlgorithm::COBBLE;
  std::cerr << "WARNING: binary tree clustering not recognized, setting to recursive 2 means (2means)." << std::endl;
  return ClusteringAlgorithm::TWO_MEANS;
}

namespace clustering_detail {

// a d x n block of points (column = point) inside a larger column-major array
struct Points {
  double* x;
  int d, n;
  double* pt(int i) const { return x + (size_t)i * d; }
};

inline double dist2(int d, const double* a, const double* b) {  // kernel/Metrics.hpp:41-50
  double k = 0.;
  for (int i = 0; i < d; i++) { double t = a[i] - b[i]; k += t * t; }
  return k;
}
inline double dist(int d, const double* a, const double* b) { return std::sqrt(dist2(d, a, b)); }

// move the points labelled 0 to the front (swap sequence of KMeans.cpp:170-182 and the other partitioners)
inline void group_zero_first(const Points& p, std::vector<int>& label, int n0, int* perm) {
  int ct = 0, cj = 0;
  for (int j = 0; j < n0; j++) {
    while (label[cj] != 0) cj++;
    if (cj != ct) {
      std::swap_ranges(p.pt(cj), p.pt(cj) + p.d, p.pt(ct));
      std::swap(perm[cj], perm[ct]);
      label[cj] = label[ct];
      label[ct] = 0;
    }
    cj++;
    ct++;
  }
}

// 2-means with the "random, distance-maximised" start (KMeans.cpp:44-63, 118-168)
inline void label_two_means(const Points& p, std::vector<int>& label, int nc[2], std::mt19937& gen) {
  const int n = p.n, d = p.d;
  std::uniform_int_distribution<std::size_t> pick(0, n - 1);
  const std::size_t t = pick(gen);
  std::vector<double> w(n);
  for (int i = 0; i < n; i++) w[i] = dist2(d, p.pt(i), p.pt((int)t));
  std::discrete_distribution<int> far(w.begin(), w.end());
  const int t2 = far(gen);
  std::vector<double> c0(p.pt((int)t), p.pt((int)t) + d), c1(p.pt(t2), p.pt(t2) + d);
  label.assign(n, 0);
  bool changes = true;
  for (int iter = 0; changes && iter < 100; iter++) {
    changes = false;
    for (int i = 0; i < n; i++) {
      const int ci = dist(d, p.pt(i), c1.data()) < dist(d, p.pt(i), c0.data()) ? 1 : 0;
      if (ci != label[i]) changes = true;
      label[i] = ci;
    }
    nc[0] = nc[1] = 0;
    std::fill(c0.begin(), c0.end(), 0.);
    std::fill(c1.begin(), c1.end(), 0.);
    for (int i = 0; i < n; i++) {
      std::vector<double>& c = label[i] ? c1 : c0;
      nc[label[i]]++;
      for (int j = 0; j < d; j++) c[j] += p.pt(i)[j];
    }
    for (int j = 0; j < d; j++) { c0[j] /= nc[0]; c1[j] /= nc[1]; }
  }
}

// Labels of a median split: the n / 2 smallest keys get 0, as the reference's
//   std::nth_element(idx.begin(), idx.begin() + n / 2, idx.end(), [&](a, b) { return key[a] < key[b]; })
// leaves them.  WHICH elements end up below position n / 2 is the same for every selection algorithm unless equal keys
// straddle that position, so the selection runs on a copy of the keys (contiguous doubles: a third of the time of the
// indirect comparisons, and the serial part of every level of the clustering); only if the keys below the median value
// do not number exactly n / 2 is the reference's own call made, whose arrangement then decides the ties.
inline void median_labels(const std::vector<double>& key, std::vector<int>& label, int nc[2]) {
  const int n = (int)key.size(), h = n / 2;
  nc[0] = h; nc[1] = n - h;
  label.assign(n, 0);
  if (n < 2) return;
  // (STRUMPACK_AMD_CLUSTER_EXACT_SELECT=1: always the reference's call -- the tests compare the two)
  static const bool exact = [] { const char* e = std::getenv("STRUMPACK_AMD_CLUSTER_EXACT_SELECT"); return e && e[0] == '1'; }();
  std::vector<double> tmp(key);
  std::nth_element(tmp.begin(), tmp.begin() + h, tmp.end());
  const double v = tmp[h];
  int below = 0;
  for (int i = 0; i < n; i++) below += key[i] < v;
  if (below == h && !exact) {
    for (int i = 0; i < n; i++) label[i] = key[i] < v ? 0 : 1;
    return;
  }
  std::vector<std::size_t> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  std::nth_element(idx.begin(), idx.begin() + h, idx.end(), [&](const std::size_t& a, const std::size_t& b) { return key[a] < key[b]; });
  for (int i = h; i < n; i++) label[idx[i]] = 1;
}

// median split along the coordinate of largest extent (KDTree.cpp:36-57, 83-95)
inline void label_kd(const Points& p, std::vector<int>& label, int nc[2]) {
  const int n = p.n, d = p.d;
  std::vector<double> mx(p.pt(0), p.pt(0) + d), mn(mx);
  for (int i = 1; i < n; i++)
    for (int j = 0; j < d; j++) { mx[j] = std::max(p.pt(i)[j], mx[j]); mn[j] = std::min(p.pt(i)[j], mn[j]); }
  int dim = 0;
  double ext = mx[0] - mn[0];
  for (int j = 1; j < d; j++)
    if (mx[j] - mn[j] > ext) { ext = mx[j] - mn[j]; dim = j; }
  std::vector<double> key(n);
  for (int i = 0; i < n; i++) key[i] = p.pt(i)[dim];
  median_labels(key, label, nc);
}

// fn(lo, hi) over [0, n) in contiguous pieces, on a few host threads when the range is long (the top levels of a large
// point set: the two halves of a split only fork BELOW it)
// Host threads forked by the clustering at any one time are capped (nested fan-out: the halves of the top splits fork, and
// each labels its points in pieces): a piece that finds the budget spent, or whose thread cannot be started, runs on the
// caller's thread instead.  Exceptions of a piece are carried to the caller after every started thread has been joined.
inline std::atomic<int>& live_threads() { static std::atomic<int> n{0}; return n; }
inline int thread_budget() { return (int)std::max(1u, std::min(512u, 2 * std::thread::hardware_concurrency())); }
struct ThreadSlot {   // one unit of the budget, held while a forked thread lives
  bool ok;
  ThreadSlot() : ok(live_threads().fetch_add(1) < thread_budget()) { if (!ok) live_threads().fetch_sub(1); }
  ~ThreadSlot() { if (ok) live_threads().fetch_sub(1); }
};

template <class F> inline void for_pieces(int n, F&& fn) {
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const int pieces = n >= 32768 ? (int)std::min<unsigned>(8, hw) : 1;
  if (pieces <= 1) { fn(0, n, 0); return; }
  std::vector<std::thread> th;
  std::vector<std::unique_ptr<ThreadSlot>> slots;
  std::vector<std::exception_ptr> err(pieces);
  auto piece = [&](int t) {
    try { fn((int)((long long)n * t / pieces), (int)((long long)n * (t + 1) / pieces), t); }
    catch (...) { err[t] = std::current_exception(); }
  };
  std::vector<int> inline_pieces{0};
  for (int t = 1; t < pieces; t++) {
    std::unique_ptr<ThreadSlot> sl(new ThreadSlot());
    bool started = false;
    if (sl->ok) {
      try { th.emplace_back(piece, t); started = true; } catch (const std::system_error&) {}
    }
    if (started) slots.push_back(std::move(sl)); else inline_pieces.push_back(t);
  }
  for (int t : inline_pieces) piece(t);
  for (auto& x : th) x.join();
  for (auto& e : err) if (e) std::rethrow_exception(e);
}

// median split by distance from the point farthest from the centroid (CobblePartitioning.cpp:36-78)
inline void label_cobble(const Points& p, std::vector<int>& label, int nc[2]) {
  const int n = p.n, d = p.d;
  std::vector<double> cen(d, 0.);
  for (int i = 0; i < n; i++)   // (serial: the sum keeps the reference's order)
    for (int j = 0; j < d; j++) cen[j] += p.pt(i)[j];
  for (int j = 0; j < d; j++) cen[j] /= n;
  // the farthest point, the FIRST one among equals: per piece, then over the pieces in order
  int pfirst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double pfar[8] = {-1., -1., -1., -1., -1., -1., -1., -1.};
  for_pieces(n, [&](int lo, int hi, int t) {
    int f = lo;
    double far = -1.;
    for (int i = lo; i < hi; i++) {
      const double dd = dist(d, p.pt(i), cen.data());
      if (dd > far) { far = dd; f = i; }
    }
    pfirst[t] = f; pfar[t] = far;
  });
  int first = 0;
  double far = -1.;
  for (int t = 0; t < 8; t++)
    if (pfar[t] > far) { far = pfar[t]; first = pfirst[t]; }
  std::vector<double> ds(n);
  for_pieces(n, [&](int lo, int hi, int) {
    for (int i = lo; i < hi; i++) ds[i] = dist(d, p.pt(i), p.pt(first));
  });
  median_labels(ds, label, nc);
}

// median split along the principal direction of the (uncentred) second-moment matrix p p^T, as the reference does
// (PCAPartitioning.cpp:36-95; it takes the eigenvector from LAPACK syevx, whose sign is implementation-defined --
// here: cyclic Jacobi on the d x d matrix, eigenvector normalised so that its largest component is positive; the
// tree is the reference's up to mirroring of a split)
inline void label_pca(const Points& p, std::vector<int>& label, int nc[2]) {
  const int n = p.n, d = p.d;
  std::vector<double> A((size_t)d * d, 0.), V((size_t)d * d, 0.);
  for (int i = 0; i < n; i++)
    for (int a = 0; a < d; a++)
      for (int b = 0; b <= a; b++) A[a + (size_t)b * d] += p.pt(i)[a] * p.pt(i)[b];
  for (int a = 0; a < d; a++) {
    V[a + (size_t)a * d] = 1.;
    for (int b = 0; b < a; b++) A[b + (size_t)a * d] = A[a + (size_t)b * d];
  }
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0.;
    for (int a = 0; a < d; a++)
      for (int b = 0; b < a; b++) off += A[a + (size_t)b * d] * A[a + (size_t)b * d];
    if (off < 1e-30) break;
    for (int q = 1; q < d; q++)
      for (int r = 0; r < q; r++) {
        const double apq = A[r + (size_t)q * d];
        if (apq == 0.) continue;
        const double theta = (A[q + (size_t)q * d] - A[r + (size_t)r * d]) / (2. * apq);
        const double t = (theta >= 0. ? 1. : -1.) / (std::fabs(theta) + std::sqrt(theta * theta + 1.));
        const double c = 1. / std::sqrt(t * t + 1.), sn = t * c;
        for (int k = 0; k < d; k++) {   // A <- A J, V <- V J
          const double akr = A[k + (size_t)r * d], akq = A[k + (size_t)q * d];
          A[k + (size_t)r * d] = c * akr - sn * akq; A[k + (size_t)q * d] = sn * akr + c * akq;
          const double vkr = V[k + (size_t)r * d], vkq = V[k + (size_t)q * d];
          V[k + (size_t)r * d] = c * vkr - sn * vkq; V[k + (size_t)q * d] = sn * vkr + c * vkq;
        }
        for (int k = 0; k < d; k++) {   // A <- J^T A
          const double ark = A[r + (size_t)k * d], aqk = A[q + (size_t)k * d];
          A[r + (size_t)k * d] = c * ark - sn * aqk; A[q + (size_t)k * d] = sn * ark + c * aqk;
        }
      }
  }
  int top = 0;
  for (int a = 1; a < d; a++)
    if (A[a + (size_t)a * d] > A[top + (size_t)top * d]) top = a;
  std::vector<double> z(V.begin() + (size_t)top * d, V.begin() + (size_t)(top + 1) * d);
  int big = 0;
  for (int a = 1; a < d; a++)
    if (std::fabs(z[a]) > std::fabs(z[big])) big = a;
  if (z[big] < 0.) for (auto& v : z) v = -v;
  std::vector<double> x(n, 0.);
  for (int i = 0; i < n; i++)
    for (int a = 0; a < d; a++) x[i] += p.pt(i)[a] * z[a];
  median_labels(x, label, nc);
}

using labeller_t = std::function<void(const Points&, std::vector<int>&, int*)>;

// `fork`: levels below this one whose two halves may run on two host threads (the halves of a split share nothing: disjoint
// ranges of the points and of the permutation; labellers with a state of their own -- the random stream of 2-means --
// pass 0).  Serial, the 100 000 points of BASELINE configs[3] took 20 ms of the 102 ms step: every level is four passes
// over all points.
inline structured::ClusterTree recurse(const Points& p, int cluster_size, int* perm, const labeller_t& lab, int fork = 0) {
  structured::ClusterTree tree(p.n);
  if (p.n < cluster_size) return tree;
  std::vector<int> label;
  int nc[2] = {0, 0};
  lab(p, label, nc);
  group_zero_first(p, label, nc[0], perm);
  if (!nc[0] || !nc[1]) return tree;
  tree.c.resize(2);
  bool forked = false;
  std::unique_ptr<ThreadSlot> slot;
  if (fork > 0 && p.n >= 4096) slot.reset(new ThreadSlot());
  if (slot && slot->ok) {
    std::exception_ptr err;
    std::thread other;
    try {
      other = std::thread([&] {
        try { tree.c[0] = recurse(Points{p.x, p.d, nc[0]}, cluster_size, perm, lab, fork - 1); }
        catch (...) { err = std::current_exception(); }
      });
      forked = true;
    } catch (const std::system_error&) {}
    if (forked) {
      try { tree.c[1] = recurse(Points{p.pt(nc[0]), p.d, nc[1]}, cluster_size, perm + nc[0], lab, fork - 1); }
      catch (...) { other.join(); throw; }
      other.join();
      if (err) std::rethrow_exception(err);
    }
  }
  if (!forked) {
    tree.c[0] = recurse(Points{p.x, p.d, nc[0]}, cluster_size, perm, lab);
    tree.c[1] = recurse(Points{p.pt(nc[0]), p.d, nc[1]}, cluster_size, perm + nc[0], lab);
  }
  return tree;
}
// levels to fork: up to the host's hardware threads
inline int fork_levels() {
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  int l = 0;
  while ((2u << l) <= hw && l < 6) l++;
  return l;
}

}  // namespace clustering_detail

// Reorders the points p (d x n, one point per column) in place, fills perm (1-based, LAPACK lapmt
// convention: new column i is old column perm[i]) and returns the cluster tree.
inline structured::ClusterTree binary_tree_clustering(ClusteringAlgorithm algo, DenseMatrix<double>& p,
                                                      std::vector<int>& perm, std::size_t cluster_size) {
  namespace cd = clustering_detail;
  const int n = (int)p.cols(), d = (int)p.rows();
  if (p.ld() != d) throw std::invalid_argument("binary_tree_clustering: points must be stored contiguously");
  perm.resize(n);
  std::iota(perm.begin(), perm.end(), 1);
  cd::Points pts{p.data(), d, n};
  switch (algo) {
    case ClusteringAlgorithm::NATURAL: {
      structured::ClusterTree t(n);
      t.refine((int)cluster_size);
      return t;
    }
    case ClusteringAlgorithm::PCA:
      return cd::recurse(pts, (int)cluster_size, perm.data(), cd::label_pca, cd::fork_levels());
    case ClusteringAlgorithm::TWO_MEANS: {
      std::mt19937 gen(1);  // reproducible, as in the reference
      return cd::recurse(pts, (int)cluster_size, perm.data(),
                         [&gen](const cd::Points& q, std::vector<int>& l, int* nc) { cd::label_two_means(q, l, nc, gen); });
    }
    case ClusteringAlgorithm::KD_TREE:
      return cd::recurse(pts, (int)cluster_size, perm.data(), cd::label_kd, cd::fork_levels());
    case ClusteringAlgorithm::COBBLE:
      return cd::recurse(pts, (int)cluster_size, perm.data(), cd::label_cobble, cd::fork_levels());
  }
  return structured::ClusterTree(n);
}

// ---- the device form of the median-split partitioners (cobble, kd): kernels/hssk_cluster.hip ------------------------------
// The tree of a median split follows from n and the cluster size alone (n / 2 | n - n / 2 while n >= cluster_size).
inline structured::ClusterTree median_split_tree(int n, int cluster_size) {
  structured::ClusterTree t(n);
  if (n < cluster_size || n / 2 == 0) return t;
  t.c.resize(2);
  t.c[0] = median_split_tree(n / 2, cluster_size);
  t.c[1] = median_split_tree(n - n / 2, cluster_size);
  return t;
}

// Reorders p and fills perm (1-based) like binary_tree_clustering, one launch per tree level.  Returns 0 when done; -1 when
// this form does not apply (algorithm, dimension, layout); > 0 when the device met a tie at a median or at the farthest point
// or a long displacement chain (hssk_cluster_median) -- p and perm are untouched then and the host form decides.
// A device block handed on by the clustering: the reordered points (d x n doubles at its start) stay on the device for the
// compression that follows; release() gives the block back to the pool.
struct DevicePoints {
  void* block = nullptr;
  std::size_t bytes = 0;
  const double* X() const { return (const double*)block; }
  void release() { if (block) DevicePool::get().release(block, bytes); block = nullptr; bytes = 0; }
};

inline int binary_tree_clustering_device(ClusteringAlgorithm algo, DenseMatrix<double>& p, std::vector<int>& perm,
                                         std::size_t cluster_size, int device, structured::ClusterTree& tree, DevicePoints* keep = nullptr) {
  const int n = (int)p.cols(), d = (int)p.rows();
  if (algo != ClusteringAlgorithm::COBBLE && algo != ClusteringAlgorithm::KD_TREE) return -1;
  if (n <= 0 || d <= 0 || d > 64 || p.ld() != d || cluster_size < 2 || cluster_size > (std::size_t)n) return -1;
  hssk_ctx* ctx = nullptr;
  if (hssk_ctx_create(&ctx, device)) throw std::runtime_error(std::string("binary_tree_clustering: ") + hssk_last_error());
  const size_t bytes = sizeof(double) * (size_t)d * n + sizeof(int) * (size_t)n;
  const size_t chunk = ((bytes + (size_t(64) << 20) - 1) >> 26) << 26;   // (the arenas' granularity: the pool hands the block on)
  void* blk = DevicePool::get().acquire(chunk);
  if (!blk) { hssk_ctx_destroy(ctx); throw std::runtime_error("binary_tree_clustering: out of device memory"); }
  double* dX = (double*)blk;
  int* dperm = (int*)(dX + (size_t)d * n);
  int status = 0, rc = hssk_memcpy_h2d(ctx, dX, p.data(), (long long)sizeof(double) * d * n);
  if (!rc) rc = hssk_cluster_median(ctx, dX, d, n, algo == ClusteringAlgorithm::COBBLE ? 4 : 2, (int)cluster_size, dperm, &status);
  std::vector<int> pm;
  if (!rc && !status) {
    pm.resize(n);
    rc = hssk_memcpy_d2h(ctx, pm.data(), dperm, (long long)sizeof(int) * n);
    if (!rc) rc = hssk_memcpy_d2h(ctx, p.data(), dX, (long long)sizeof(double) * d * n);
  }
  std::string err = rc ? hssk_last_error() : "";
  if (keep && !rc && !status) { keep->block = blk; keep->bytes = chunk; }   // (the caller's, with the points in cluster order)
  else DevicePool::get().release(blk, chunk);
  hssk_ctx_destroy(ctx);
  if (rc == 2) return -1;
  if (rc) throw std::runtime_error("binary_tree_clustering (device): " + err);
  if (status) return status;
  perm.resize(n);
  for (int i = 0; i < n; i++) perm[i] = pm[i] + 1;
  tree = median_split_tree(n, (int)cluster_size);
  return 0;
}

// binary_tree_clustering with the median-split partitioners on the device when the point set is large enough to pay for the
// launches (STRUMPACK_AMD_CLUSTER_DEVICE_MIN points, default 8192; STRUMPACK_AMD_CLUSTER_HOST=1: always the host form).
// *used_device (may be null): whether the device form produced the result.
inline structured::ClusterTree binary_tree_clustering(ClusteringAlgorithm algo, DenseMatrix<double>& p, std::vector<int>& perm,
                                                      std::size_t cluster_size, int device, bool* used_device = nullptr,
                                                      DevicePoints* keep = nullptr) {
  static const bool host_only = [] { const char* e = std::getenv("STRUMPACK_AMD_CLUSTER_HOST"); return e && e[0] == '1'; }();
  static const long long dev_min = [] { const char* e = std::getenv("STRUMPACK_AMD_CLUSTER_DEVICE_MIN"); return e ? std::atoll(e) : 8192LL; }();
  if (used_device) *used_device = false;
  if (!host_only && (long long)p.cols() >= dev_min) {
    structured::ClusterTree t(0);
    if (binary_tree_clustering_device(algo, p, perm, cluster_size, device, t, keep) == 0) {
      if (used_device) *used_device = true;
      return t;
    }
  }
  return binary_tree_clustering(algo, p, perm, cluster_size);
}

}  // namespace strumpack
