// find_approximate_neighbors: approximate k nearest neighbours by random projection trees
// (reference: clustering/NeighborSearch.hpp, clustering/NeighborSearch.cpp:88-345).
//
// The default neighbour search of the kernel-matrix compression is the exact one on the device (hssk_knn).  This
// host routine is the reference's own randomized algorithm, kept (a) because it is part of the reference's public
// surface and (b) as the neighbour search that makes a kernel HSS matrix come out *identical* to the reference's
// (HSSOptions::set_neighbor_search(NeighborSearch::ANN), --hss_neighbor_search ann): same std::mt19937(1) stream,
// same std::normal_distribution / std::sort / std::partial_sort calls on the same data.  The only freedom left is
// the rounding of the d-term projection dot product (the reference calls BLAS ddot), which can reorder exact ties.
//
// Algorithm: a tree sample splits the points recursively at the median of their projection on a random direction
// until a node has fewer than 6 k points; inside every leaf the k nearest points of each point are found by brute
// force; further tree samples are merged in (the k best of both lists) until the lists contain >= 99 % of the true
// neighbours of 100 random sample points or num_iters extra samples have been drawn.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <random>
#include <vector>

#include "DenseMatrix.hpp"

namespace strumpack {
namespace ann_detail {

inline double dist2(std::size_t d, const double* a, const double* b) {
  double k = 0.;
  for (std::size_t i = 0; i < d; i++) { double t = a[i] - b[i]; k += t * t; }
  return k;
}

struct Lists {   // k x n, column c = point c, ordered by (distance, id)
  std::size_t k = 0, n = 0;
  std::vector<std::uint32_t> id;
  std::vector<double> score;
  Lists(std::size_t k_, std::size_t n_) : k(k_), n(n_), id(k_ * n_, 0), score(k_ * n_, 0.) {}
};

inline void projection_tree(const double* X, std::size_t d, std::size_t min_leaf, std::vector<std::uint32_t>& cur,
                            std::size_t start, std::size_t size, std::vector<std::size_t>& leaves,
                            std::vector<std::size_t>& leaf_ptr, std::mt19937& gen) {
  if (size < min_leaf) {
    leaf_ptr.push_back(leaf_ptr.back() + size);
    for (std::size_t i = 0; i < size; i++) leaves.push_back(cur[start + i]);
    return;
  }
  std::vector<double> dir(d);
  std::normal_distribution<double> normal(0.0, 1.0);   // a fresh distribution per node, as in the reference
  for (std::size_t i = 0; i < d; i++) dir[i] = normal(gen);
  double nrm = 0.;
  for (std::size_t i = 0; i < d; i++) nrm += dir[i] * dir[i];
  nrm = std::sqrt(nrm);
  for (std::size_t i = 0; i < d; i++) dir[i] /= nrm;
  std::vector<double> rel(size, 0.);
  for (std::size_t i = 0; i < size; i++) {
    const double* x = X + (std::size_t)cur[start + i] * d;
    double s = 0.;
    for (std::size_t j = 0; j < d; j++) s += x[j] * dir[j];
    rel[i] = s;
  }
  std::vector<std::uint32_t> idx(size);
  std::iota(idx.begin(), idx.end(), 0);
  const std::uint32_t half = (std::uint32_t)size / 2;
  std::sort(idx.begin(), idx.end(), [&](const std::uint32_t& a, const std::uint32_t& b) {
    return (rel[a] < rel[b]) || ((rel[a] == rel[b]) && (a < b)); });
  std::vector<std::uint32_t> sorted(size);
  for (std::size_t i = 0; i < size; i++) sorted[i] = cur[start + idx[i]];
  std::copy(sorted.begin(), sorted.end(), cur.begin() + start);
  projection_tree(X, d, min_leaf, cur, start, half, leaves, leaf_ptr, gen);
  projection_tree(X, d, min_leaf, cur, start + half, size - half, leaves, leaf_ptr, gen);
}

template <class ParFor>
inline void tree_sample(const double* X, std::size_t d, Lists& L, std::mt19937& gen, ParFor&& parfor) {
  const std::size_t n = L.n, k = L.k;
  std::vector<std::size_t> leaves, leaf_ptr;
  leaves.reserve(n);
  leaf_ptr.push_back(0);
  std::vector<std::uint32_t> cur(n);
  std::iota(cur.begin(), cur.end(), 0);
  projection_tree(X, d, 6 * k, cur, 0, n, leaves, leaf_ptr, gen);
  parfor(leaf_ptr.size() - 1, [&](std::size_t leaf) {
    const std::size_t sz = leaf_ptr[leaf + 1] - leaf_ptr[leaf];
    const std::size_t* pts = leaves.data() + leaf_ptr[leaf];
    std::vector<double> D(sz * sz);
    for (std::size_t i = 0; i < sz; i++) {
      D[i + i * sz] = 0.;
      for (std::size_t j = i + 1; j < sz; j++) D[j + i * sz] = D[i + j * sz] = dist2(d, X + pts[i] * d, X + pts[j] * d);
    }
    std::vector<std::uint32_t> idx(sz);
    const std::size_t kk = std::min(k, sz);
    for (std::size_t i = 0; i < sz; i++) {
      std::iota(idx.begin(), idx.end(), 0);
      std::partial_sort(idx.begin(), idx.begin() + kk, idx.end(), [&](const std::uint32_t& a, const std::uint32_t& b) {
        return (D[i + a * sz] < D[i + b * sz]) || ((D[i + a * sz] == D[i + b * sz]) && (a < b)); });
      for (std::size_t j = 0; j < kk; j++) {
        L.id[j + pts[i] * k] = (std::uint32_t)pts[idx[j]];
        L.score[j + pts[i] * k] = D[i + idx[j] * sz];
      }
    }
  });
}

// the k best of two (distance-ordered) lists per point
inline void merge_best(Lists& A, const Lists& B) {
  const std::size_t k = A.k;
  std::vector<std::uint32_t> ci(k);
  std::vector<double> cs(k);
  for (std::size_t c = 0; c < A.n; c++) {
    std::uint32_t* ai = A.id.data() + c * k;
    double* as = A.score.data() + c * k;
    const std::uint32_t* bi = B.id.data() + c * k;
    const double* bs = B.score.data() + c * k;
    std::size_t r1 = 0, r2 = 0, cur = 0;
    while (r1 < k && r2 < k && cur < k) {
      if (as[r1] > bs[r2]) { ci[cur] = bi[r2]; cs[cur] = bs[r2]; r2++; }
      else {
        ci[cur] = ai[r1]; cs[cur] = as[r1];
        if (ai[r1] == bi[r2]) r2++;
        r1++;
      }
      cur++;
    }
    while (cur < k) {
      if (r1 == k) { ci[cur] = bi[r2]; cs[cur] = bs[r2]; r2++; }
      else { ci[cur] = ai[r1]; cs[cur] = as[r1]; r1++; }
      cur++;
    }
    std::copy(ci.begin(), ci.end(), ai);
    std::copy(cs.begin(), cs.end(), as);
  }
}

// average fraction of the true k nearest neighbours of 100 random points found in the lists
inline double quality(const double* X, std::size_t d, const Lists& L, std::mt19937& gen) {
  const std::size_t n = L.n, k = L.k, ns = 100;
  std::vector<std::size_t> samples(ns);
  {
    std::uniform_int_distribution<std::size_t> uni(0, n - 1);
    for (auto& s : samples) s = uni(gen);
  }
  std::vector<double> dist(n);
  std::vector<std::uint32_t> idx(n);
  double q = 0.;
  for (std::size_t j = 0; j < ns; j++) {
    const std::size_t i = samples[j];
    for (std::size_t c = 0; c < n; c++) dist[c] = dist2(d, X + i * d, X + c * d);
    std::iota(idx.begin(), idx.end(), 0);
    std::partial_sort(idx.begin(), idx.begin() + std::min(k, n), idx.end(), [&](const std::uint32_t& a, const std::uint32_t& b) {
      return (dist[a] < dist[b]) || ((dist[a] == dist[b]) && (a < b)); });
    std::size_t r1 = 0, r2 = 0;
    int found = 0;
    while (r2 < k)
      if (L.id[r1 + i * k] == idx[r2]) { r1++; r2++; found++; }
      else r2++;
    q += (double)found / k;
  }
  return q / ns;
}

struct Serial {
  template <class F> void operator()(std::size_t n, F&& f) const { for (std::size_t i = 0; i < n; i++) f(i); }
};

template <class ParFor>
inline Lists search(const double* X, std::size_t d, std::size_t n, std::size_t num_iters, std::size_t k, ParFor&& parfor) {
  Lists L(k, n);
  std::mt19937 gen(1);   // reproducible, as in the reference
  tree_sample(X, d, L, gen, parfor);
  double q = quality(X, d, L, gen);
  for (std::size_t it = 0; it < num_iters && q < 0.99; it++) {
    Lists M(k, n);
    tree_sample(X, d, M, gen, parfor);
    merge_best(L, M);
    q = quality(X, d, L, gen);
  }
  return L;
}

}  // namespace ann_detail

// neighbors / scores: ann_number x n (column c = neighbours of point c, nearest first, the point itself included)
inline void find_approximate_neighbors(const DenseMatrix<double>& data, std::size_t num_iters, std::size_t ann_number,
                                       DenseMatrix<std::uint32_t>& neighbors, DenseMatrix<double>& scores) {
  if (data.ld() != int(data.rows())) throw std::invalid_argument("find_approximate_neighbors: points must be contiguous");
  auto L = ann_detail::search(data.data(), data.rows(), data.cols(), num_iters, ann_number, ann_detail::Serial());
  neighbors = DenseMatrix<std::uint32_t>(ann_number, data.cols(), L.id.data(), ann_number);
  scores = DenseMatrix<double>(ann_number, data.cols(), L.score.data(), ann_number);
}

}  // namespace strumpack
