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
// This file: construction, tree, introspection.  The rest of the class: see hss_engine_internal.hpp.
#include "hss_engine_internal.hpp"

namespace strumpack {
namespace HSS {

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
  if (fctx_) hssk_sync(fctx_);
  if (ctx_) hssk_sync(ctx_);
  drop_plans();
  plan_arena_.reset();
  persist_.reset();
  work_.reset();
  fact_.reset();
  tmp_.reset();
  comm_arena_.reset();
  schur_.reset();
  if (fctx_) hssk_ctx_destroy(fctx_);
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
// memory(): bytes of D, B01, B10, the E factors and the pivot vectors PLUS the reference's per-node bookkeeping, so that the
// number is the one HSSMatrix::memory() of the reference reports (HSS/HSSMatrix.cpp:308-314 adds sizeof(*this) per node: 592
// bytes for HSSMatrix<double> in its g++ / libstdc++ build; the node table of this engine is of the same order).
// nonzeros() is the count of stored scalars and pivot entries only: the reference adds the same sizeof(*this) BYTES to its
// scalar count as well (HSSMatrix.cpp:317-323) -- its nonzeros() = this nonzeros() + 592 * nodes; that quirk is not reproduced.
long long DeviceHSS::memory(int node) const {
  long long t = 0;
  for (int i = node, e = subtree_end(node); i < e; i++) {
    const Node& nd = nodes_[i];
    t += kRefNodeBytes;
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

}  // namespace HSS
}  // namespace strumpack
