// DeviceHSS: shift and the hierarchical mat-vec (HSSMatrix.apply.hpp:55-220).
#include "hss_engine_internal.hpp"

namespace strumpack {
namespace HSS {

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
  invalidate_factors();  // the ULV factors are stale (examples/dense/testStructured.cpp:199)
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
  invalidate_factors();
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
  // part: 0 everything; 1 only the leaves' op(D) x + beta y (independent of the tree: it can run next to the sweeps, on the side
  // stream); 2 everything but that
  auto down = [&](const std::vector<int>& ids, int part = 0) {
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
      if (part == 1 && !nd.leaf()) continue;
      if (nd.leaf()) {
        // c = D b + beta c (+ U tmp2)
        if (part != 2) leafmm.push_back(hssk_gemm_desc{nd.D, dx + (nd.lo - lo0), out, nd.m, nrhs, nd.m, nd.m, (int)lx, ldo, T ? 1 : 0, 0, 1.0, beta});
        if (part == 1) continue;
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
  const bool fuse = nrhs <= fuse_max_nrhs() && !no_fuse;   // (more right-hand sides: the batched MFMA launches per level)
  if (fuse) ck(hssk_sweep_arm(ctx_, hand, (long long)hand_total));
  typedef std::vector<std::vector<int>> Levels;
  // leaf_acc: the leaves of `downs` only add U tmp2 onto what is in y (their op(D) x is already there)
  auto sweep = [&](const Levels* ups, const Levels* downs, bool leaf_acc = false) -> bool {
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
          if (nd.leaf() && leaf_acc) {
            d.acc = 1; d.m = nd.m;
            d.out = dy + (nd.lo - lo0); d.ldo = (int)ly;
          } else if (nd.leaf()) {
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
  // Many right-hand sides (hybrid): the single-launch sweep works on groups of four right-hand sides, each group streaming
  // the blocks again -- at nrhs = 64 that is 16 passes over the leaves' diagonal blocks, 88 % of all bytes.  The LEAF level
  // therefore runs as batched MFMA GEMMs over all right-hand sides at once (every block read once), and only the inner
  // levels -- small blocks, a chain of dependent levels -- stay in the single launch; the leaves' results reach it through
  // the hand-off buffers like any child's (a value that is already there is taken without waiting).
  bool whole = false;
  // (also whenever the leaves are beyond the sweep kernels' 256 rows -- leaf size 512 --: their level as batched launches, the
  // inner levels, whose nodes are small, still in the single launch instead of two to four launches per level)
  bool big_leaves = false;
  for (int id : ups_own.empty() ? std::vector<int>() : ups_own[0]) big_leaves = big_leaves || nodes_[id].m > 256;
  if (fuse && !dist_subtree_ && (nrhs >= hybrid_nrhs() || big_leaves) && !ups_own.empty() && ups_own.size() > 1) {
    Levels up_in(ups_own.begin() + 1, ups_own.end()), down_in;
    for (auto& ids : downs_own) {
      std::vector<int> v;
      for (int id : ids) if (!nodes_[id].leaf()) v.push_back(id);
      down_in.push_back(std::move(v));
    }
    // The leaves' op(D) x -- most of the bytes and flops of the product, and independent of the tree -- goes to the side stream,
    // next to the sweeps; what is left of the leaves' step is y += U tmp2.
    static const bool no_side = [] { const char* e = std::getenv("STRUMPACK_AMD_NO_SIDE_STREAM"); return e && e[0] == '1'; }();
    const bool side = !no_side;
    if (side) {
      ck(hssk_side_begin(ctx_));
      struct End { hssk_ctx* c; ~End() { hssk_side_end(c); } } end{ctx_};
      down(ups_own[0], 1);
    }
    const int rest = side ? 2 : 0;
    if (side && nrhs >= hssk_sweep_mma_min_nrhs() && !big_leaves) {
      // matrix-core form (kernels/hssk_sweep_mma.h): every level, the leaves included, in the one launch; the leaves' y += U tmp2
      // as a second launch behind the side stream's op(D) x
      struct Req { hssk_ctx* c; Req(hssk_ctx* c_) : c(c_) { hssk_sweep_require_mma(c, 1); } ~Req() { hssk_sweep_require_mma(c, 0); } } req(ctx_);
      if (sweep(&ups_own, &down_in)) {
        ck(hssk_side_join(ctx_));
        Levels leaf_level{ups_own[0]};
        if (!sweep(nullptr, &leaf_level, true)) down(ups_own[0], rest);
        whole = true;
      }
    }
    if (!whole) {
      up(ups_own[0]);
      if (sweep(&up_in, &down_in)) {
        if (side) ck(hssk_side_join(ctx_));
        down(ups_own[0], rest);
      } else {   // (a node beyond the sweep's limits: the batched launches for the rest as well)
        for (auto& ids : up_in) up(ids);
        if (side) ck(hssk_side_join(ctx_));
        for (auto& ids : downs_own) down(ids, rest);
      }
      whole = true;
    }
  }
  // single process: the whole product is one launch
  if (!whole) whole = fuse && !dist_subtree_ && sweep(&ups_own, &downs_own);
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

}  // namespace HSS
}  // namespace strumpack
