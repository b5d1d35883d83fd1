// structured::ClusterTree (reference: structured/ClusterTree.hpp:60-125): the binary partition tree
// that defines the HSS block structure.
#pragma once
#include <cassert>
#include <iostream>
#include <vector>

namespace strumpack {
namespace structured {

class ClusterTree {
 public:
  int size;
  std::vector<ClusterTree> c;
  ClusterTree() : size(0) {}
  ClusterTree(int n) : size(n) {}
  // split while size >= 2*leaf_size (reference :104-114)
  const ClusterTree& refine(int leaf_size) {
    assert(c.empty());
    if (size >= 2 * leaf_size) {
      c.resize(2);
      c[0].size = size / 2;
      c[0].refine(leaf_size);
      c[1].size = size - size / 2;
      c[1].refine(leaf_size);
    }
    return *this;
  }
  // sizes of the leaves, left to right (reference :152-158)
  template <typename T = int> std::vector<T> leaf_sizes() const {
    std::vector<T> l;
    collect(l);
    return l;
  }
  void print() const { for (auto& ch : c) ch.print(); std::cout << size << " "; }
  bool is_complete() const {
    if (c.empty()) return true;
    return c.size() == 2 && c[0].is_complete() && c[1].is_complete();
  }
  template <typename T> void collect(std::vector<T>& l) const {
    if (c.empty()) l.push_back(T(size));
    else for (auto& ch : c) ch.collect(l);
  }
  int levels() const { int l = 0; for (auto& ch : c) l = std::max(l, ch.levels()); return l + 1; }
};

}  // namespace structured
}  // namespace strumpack
