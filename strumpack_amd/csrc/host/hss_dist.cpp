// DeviceHSS: subtree ownership of a multi-GPU run and the exchanges at the cut level.
#include "hss_engine_internal.hpp"

namespace strumpack {
namespace HSS {

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

}  // namespace HSS
}  // namespace strumpack
