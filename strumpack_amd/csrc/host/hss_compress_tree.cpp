// DeviceHSS: the inner levels of a compression round as ONE launch (hssk_tree_inner, kernels/hssk_tree.hip) -- the host side:
// eligibility, the speculated rank bound, the device node table, the single read-back and the commit of the node table.
// HSSMatrix.compress_stable.hpp:165-348 above the leaves; the level-synchronous form of the same steps is process_level
// (hss_compress.cpp), which stays the path for everything this one declines.
#include "hss_engine_internal.hpp"

namespace strumpack {
namespace HSS {

static std::atomic<long long> g_tree_launches{0}, g_tree_fallbacks{0};
long long tree_pass_launches() { return g_tree_launches; }
long long tree_pass_fallbacks() { return g_tree_fallbacks; }

static bool tree_pass_enabled() {   // (read per call: the tests compare both paths in one process)
  const char* e = std::getenv("STRUMPACK_AMD_TREE_LAUNCH");
  return !(e && e[0] == '0');
}

// All of this rank's inner nodes (heights >= 1 of own_by_height_) in one launch.  Returns false -- with nothing of the
// matrix touched -- when the pass does not apply (the caller then walks the levels) :
//   * a source that cannot serve scattered entries from device memory / a formula,
//   * leaves that are not all compressed yet, or inner nodes that already hold state of an earlier round,
//   * more than 256 samples, ranks beyond the kernel's bounds,
// or when a rank came out above the speculated bound (status of the launch).
bool DeviceHSS::tree_pass(Source& src, int d, int dd) {
  if (!tree_pass_enabled() || o_.algorithm != 1) return false;
  const int dtot = d + dd;
  if (dtot > 256 || own_by_height_.size() < 2) return false;
  hssk_elem_src es{};
  if (!src.device_elems(*this, &es)) return false;
  int lmax = 0, ninner = 0;
  for (size_t h = 0; h < own_by_height_.size(); h++)
    for (int id : own_by_height_[h]) {
      const Node& nd = nodes_[id];
      if (nd.leaf()) {
        if (!nd.compressed() || nd.lvl == 0) return false;
        lmax = std::max(lmax, std::max(nd.rU, nd.rV));
      } else {
        if (!nd.untouched() || nd.panels) return false;
        ninner++;
      }
    }
  if (!ninner) return false;
  // Rank bound: ranks grow slowly up a tree (N = 32768: 16 at the leaves, 31 at the top; N = 1e5: ~20 and 41), so 2.5 x the
  // leaves' largest covers the trees this engine meets; the launch reports the ones it does not.  m = r0 + r1 <= 2 rcap rows
  // enter a decomposition and must not outnumber the samples (below that the stable algorithm runs its stopping test first,
  // which is the level-synchronous path's business).
  int rcap = ((std::max(32, (5 * lmax + 1) / 2) + 15) / 16) * 16;
  if (const char* e = std::getenv("STRUMPACK_AMD_TREE_RCAP")) rcap = std::atoi(e);   // (tests: force an overflow)
  rcap = std::min(rcap, (dtot / 2) / 16 * 16);
  // (the kernel's variants: 32 / 48 / 64, as far as the device's LDS goes -- anything else is the level path's, decided HERE,
  //  before anything is carved from an arena)
  if ((rcap != 32 && rcap != 48 && rcap != 64) || rcap > hssk_tree_rcap_max() || lmax > rcap) return false;
  const int mcap = 2 * rcap;

  // ---- device node table: this rank's nodes, children before parents (own_by_height_ order)
  std::vector<int> tix(nodes_.size(), -1);
  std::vector<int> tnodes;
  for (auto& ids : own_by_height_) for (int id : ids) { tix[id] = (int)tnodes.size(); tnodes.push_back(id); }
  for (int id : tnodes)
    if (!nodes_[id].leaf() && (tix[nodes_[id].c0] < 0 || tix[nodes_[id].c1] < 0)) return false;   // a child outside this rank's levels: not a tree this pass knows
  const size_t nt = tnodes.size();
  std::vector<hssk_tnode> tab(nt);
  // storage of the inner nodes at the rank bound -- panels, workspace, integer arrays -- from the compression's work arena;
  // what the matrix keeps is copied out once the ranks are known (X, B01, B10 to exact-size blocks, the integer arrays in one
  // piece); a pass that ends in the level path gives its storage back (the matrix's own arena is not touched before the commit)
  const Arena::Mark work_mark = work_->mark();
  int* iblock = work_->ints(4 * nt + (size_t)ninner * (2 * mcap + 2 * rcap));
  int* ires = iblock;
  int* inext = iblock + 4 * nt;
  struct Tmp { double *X[2], *B01, *B10; };
  std::vector<Tmp> tmpv(nt);
  std::vector<int> order;
  for (size_t t = 0; t < nt; t++) {
    const int id = tnodes[t];
    Node& nd = nodes_[id];
    hssk_tnode& q = tab[t];
    std::memset(&q, 0, sizeof(q));
    q.lvl = nd.lvl;
    if (nd.leaf()) {
      q.c0 = q.c1 = -1;
      q.S[0] = nd.Srt; q.S[1] = nd.Sct;
      q.perm[0] = nd.permU; q.perm[1] = nd.permV;
      q.Rred[0] = nd.RrtRed; q.Rred[1] = nd.RctRed;
      q.I[0] = nd.dIr; q.I[1] = nd.dIc;
      q.r[0] = nd.rU; q.r[1] = nd.rV; q.m[0] = nd.mU; q.m[1] = nd.mV;
      q.flag[0] = q.flag[1] = 1;
      continue;
    }
    q.c0 = tix[nd.c0]; q.c1 = tix[nd.c1];
    Tmp& tv = tmpv[t];
    tv.B01 = q.B01 = work_->dbl((size_t)rcap * rcap);
    tv.B10 = q.B10 = work_->dbl((size_t)rcap * rcap);
    if (nd.lvl != 0) {
      for (int s = 0; s < 2; s++) {
        q.S[s] = work_->dbl((size_t)dcap_ * mcap);
        q.Rred[s] = work_->dbl((size_t)dcap_ * rcap);
        q.W[s] = work_->dbl((size_t)mcap * mcap);
        tv.X[s] = q.X[s] = work_->dbl((size_t)rcap * mcap);
        q.perm[s] = inext; inext += mcap;
        q.I[s] = inext; inext += rcap;
      }
      order.push_back((int)(t << 1));
      order.push_back((int)(t << 1) | 1);
    } else order.push_back((int)(t << 1));
  }
  hssk_tnode* dtab = (hssk_tnode*)work_->alloc(sizeof(hssk_tnode) * nt);
  ck(hssk_upload_async(ctx_, dtab, tab.data(), (long long)(sizeof(hssk_tnode) * nt)));
  int rc = hssk_tree_inner(ctx_, dtab, order.data(), (int)order.size(), dtot, dcap_, rcap, o_.rel_tol, o_.abs_tol, o_.max_rank, &es, ires);
  if (rc == 2) { work_->rewind(work_mark); return false; }
  ck(rc);
  g_tree_launches++;
  // ---- the read-back of the tree: ranks and statuses (a few KB) first -- with them the exact-size blocks of the matrix are
  // carved and the copies into them enqueued -- and the pivoted orders and skeleton indices for the host's tables while the
  // device copies; small trees (a rank's subtree of a multi-GPU run) take everything in one piece
  const size_t nints = (size_t)(inext - iblock);
  const bool one_piece = nints * sizeof(int) <= (size_t(128) << 10);
  std::vector<int> hall(nints);
  ck(hssk_memcpy_d2h(ctx_, hall.data(), iblock, (long long)(sizeof(int) * (one_piece ? nints : 4 * nt))));
  if (hssk_sweep_status(ctx_)) throw std::runtime_error(std::string("compress: single-launch tree pass: ") + hssk_last_error());
  for (int e : order) {
    const int t = e >> 1, sd = e & 1;
    if (hall[4 * (size_t)t + 2 + sd]) {   // a rank above the bound: the level-synchronous path redoes the inner levels
      g_tree_fallbacks++;
      ck(hssk_sync(ctx_));                // (nothing of the launch is in flight when its storage is handed out again)
      work_->rewind(work_mark);
      return false;
    }
  }
  // the integer arrays the matrix keeps (pivoted orders, skeleton indices: at the rank bound, as the kernel laid them out)
  int* ikeep = nints > 4 * nt ? persist_->ints(nints - 4 * nt) : nullptr;
  if (ikeep) ck(hssk_memcpy_d2d(ctx_, ikeep, iblock + 4 * nt, (long long)(sizeof(int) * (nints - 4 * nt))));
  auto kept = [&](int* p) { return ikeep + (p - (iblock + 4 * nt)); };
  // ---- commit: the node table as process_level leaves it
  std::vector<hssk_colgather_desc> cp;
  cp.reserve(4 * (size_t)ninner);
  auto keep = [&](const double* from, int rows, int cols) -> double* {   // exact-size block of the matrix (leading dimension max(rows, 1))
    double* to = persist_->dbl((size_t)std::max(rows, 1) * std::max(cols, 1));
    if (rows > 0 && cols > 0) cp.push_back(hssk_colgather_desc{from, to, nullptr, rows, cols, rows, rows, 0});
    return to;
  };
  for (size_t t = 0; t < nt; t++) {
    const int id = tnodes[t];
    Node& nd = nodes_[id];
    if (nd.leaf()) continue;
    const hssk_tnode& q = tab[t];
    Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
    if (nd.lvl != 0) {   // (ranks before the parent's blocks are sized: own_by_height_ order has the children first)
      nd.mU = a.rU + b.rU; nd.mV = a.rV + b.rV;
      nd.rU = hall[4 * t]; nd.rV = hall[4 * t + 1];
    }
    nd.B01 = keep(tmpv[t].B01, a.rU, b.rV);
    nd.B10 = keep(tmpv[t].B10, b.rU, a.rV);
    stats_.f_local += 4.0 * ((double)a.rU * b.rV + (double)b.rU * a.rV) * dtot;
    if (nd.lvl == 0) { nd.Ustate = nd.Vstate = 2; continue; }
    nd.Srt = q.S[0]; nd.Sct = q.S[1];
    nd.Rrt = nd.Rct = nullptr;   // (an inner node's samples are its children's reduced ones, read in place)
    nd.RrtRed = q.Rred[0]; nd.RctRed = q.Rred[1];
    nd.permU = kept(q.perm[0]); nd.permV = kept(q.perm[1]);
    nd.dIr = kept(q.I[0]); nd.dIc = kept(q.I[1]);
    nd.panels = true;
    for (int s = 0; s < 2; s++) {
      const int m = s == 0 ? nd.mU : nd.mV, r = s == 0 ? nd.rU : nd.rV;
      (s == 0 ? nd.XU : nd.XV) = keep(tmpv[t].X[s], r, m - r);
      (s == 0 ? nd.Ustate : nd.Vstate) = 2;
      const int K = (m > r && r > 0) ? m - r : 0;
      stats_.f_reduce += 2.0 * r * (double)K * dtot;
      stats_.f_id += 2.0 * (4.0 * m * (double)dtot * r - 2.0 * (m + dtot) * (double)r * r + 4.0 * r * (double)r * r / 3.0 + (double)r * r * (m - r));
    }
  }
  if (!cp.empty()) ck(hssk_gather_cols(ctx_, cp.data(), (int)cp.size()));
  // the host's copies of the pivoted orders and skeleton indices (extraction, serialization, the exchanges of a distributed tree)
  if (!one_piece) ck(hssk_memcpy_d2h(ctx_, hall.data() + 4 * nt, iblock + 4 * nt, (long long)(sizeof(int) * (nints - 4 * nt))));
  for (size_t t = 0; t < nt; t++) {
    Node& nd = nodes_[tnodes[t]];
    if (nd.leaf() || nd.lvl == 0) continue;
    const hssk_tnode& q = tab[t];
    for (int s = 0; s < 2; s++) {
      const int m = s == 0 ? nd.mU : nd.mV, r = s == 0 ? nd.rU : nd.rV;
      const int* hp = hall.data() + (q.perm[s] - iblock);
      const int* hi = hall.data() + (q.I[s] - iblock);
      (s == 0 ? nd.hpermU : nd.hpermV).assign(hp, hp + m);
      (s == 0 ? nd.Ir : nd.Ic).assign(hi, hi + r);
    }
  }
  return true;
}

}  // namespace HSS
}  // namespace strumpack
