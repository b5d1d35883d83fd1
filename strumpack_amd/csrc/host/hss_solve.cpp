// DeviceHSS: ULV solve (HSSMatrix.solve.hpp:69-238).
#include "hss_engine_internal.hpp"

namespace strumpack {
namespace HSS {

// ---------------------------------------------------------------------------------------------
// ULV solve (HSSMatrix.solve.hpp:69-238)
// ---------------------------------------------------------------------------------------------
void DeviceHSS::solve(int nrhs, double* b, long long ldb, bool on_device) { solve_sub(0, nrhs, b, ldb, on_device); }

// HSSMatrix::forward_solve / backward_solve (HSS/HSSMatrix.solve.hpp:52-66): the two halves of a solve with the state between
// them (HSS::WorkSolve) in `w`.  Forward: the elimination sweep up to the root's solve, x_root out (w.x of the reference);
// partial (the matrix was partial_factor()ed and `node` is child 0, or node was factored on its own): also
// reduced_rhs = Vhat^* x_root + V^* [z_0; z_1] (solve.hpp:139-154), what a sparse front subtracts Theta times from its update
// part (sparse/fronts/FrontHSS.cpp:458-461).  Backward: x_root in (the front has subtracted Phi^* y_upd from it, :489-493),
// the solution of the node's rows out.
void DeviceHSS::forward_solve_node(int node, SolveWork& w, int nrhs, const double* b, long long ldb, bool partial, double* xroot,
                                   long long ldx, double* reduced, long long ldr) {
  w.partial = partial;
  w.xroot = xroot; w.ldx = ldx; w.reduced = reduced; w.ldr = ldr;
  solve_sub(node, nrhs, const_cast<double*>(b), ldb, false, 1, &w);
}
void DeviceHSS::backward_solve_node(int node, SolveWork& w, const double* xroot, long long ldx, double* x, long long ldxo) {
  if (!w.valid || w.sr != node) throw std::logic_error("backward_solve: no forward_solve of this matrix went before");
  w.xroot = const_cast<double*>(xroot); w.ldx = ldx;
  solve_sub(node, w.nrhs, x, ldxo, false, 2, &w);
}

// Chain blocks (hssk_sweep_fwd_desc::G, kernels/hssk_sweep.hip): for every inner node below the root the matrix G with
// [ft1; z] = G [f; zc] -- the node's whole forward step (solve.hpp:88-192: coupling products, P^T, E, L^{-1}, W1 Q0, V^*, Vt0^*) as
// what it is for the parent, one linear map of what the children hand over.  G is not assembled from the factors: the forward
// sweep ITSELF is run on the columns of an identity, every node on its own (no waits), sixteen columns per launch, and writes G
// where it would write the hand-off -- whatever arithmetic the sweep does, the block reproduces.  Cost: a dozen small launches
// and (rU + rV)(mU + mV) doubles per node (57 MB at N = 1e5 against 270 MB of factors), paid when a single-vector solve on a
// device buffer comes the second time (the call that is recorded for replay); STRUMPACK_AMD_NO_CHAIN=1 keeps the plain sweep.
void DeviceHSS::chain_blocks() {
  if (chain_built_) return;
  chain_built_ = true;
  static const bool off = [] { const char* e = std::getenv("STRUMPACK_AMD_NO_CHAIN"); return e && e[0] == '1'; }();
  if (off || o_.world != 1 || dist_subtree_ || !factored_) return;
  const int CH = 16;
  auto pad = [&](int k) { return (k + CH - 1) / CH * CH; };
  std::vector<int> ids;
  int kmax = 0;
  for (size_t i = 1; i < nodes_.size(); i++) {
    const Node& nd = nodes_[i];
    if (nd.leaf() || nd.mU <= nd.rU || nd.mU > 256 || nd.mV > 256) continue;
    if (!nd.B01 || !nd.B10 || !nd.Rlq || !nd.Tinv || (nd.rU && !nd.WQ) || (nd.rV && !nd.Vt0T)) continue;
    if (!hssk_sweep_chain_ok(nd.mU, nd.rU, nd.mV, nd.rV)) continue;
    ids.push_back((int)i);
    kmax = std::max(kmax, nd.mU + nd.mV);
  }
  if (ids.empty()) return;
  const int kpad = pad(kmax);
  if (!chain_arena_) chain_arena_.reset(new Arena());
  const Arena::Mark mk = tmp_->mark();
  std::vector<double> eye((size_t)kpad * kpad, 0.);
  for (int i = 0; i < kpad; i++) eye[(size_t)i * kpad + i] = 1.;
  double* dI = tmp_->dbl(eye.size());
  ck(hssk_memcpy_h2d(ctx_, dI, eye.data(), (long long)(sizeof(double) * eye.size())));
  std::vector<double*> ys(ids.size());
  for (size_t k = 0; k < ids.size(); k++) {
    Node& nd = nodes_[ids[k]];
    nd.Gc = chain_arena_->dbl((size_t)(nd.rU + nd.rV) * pad(nd.mU + nd.mV));
    ys[k] = tmp_->dbl((size_t)(nd.mU - nd.rU) * CH);
  }
  bool ok = true;
  for (int c0 = 0; c0 < kpad && ok; c0 += CH) {
    std::vector<hssk_sweep_fwd_desc> fd;
    for (size_t k = 0; k < ids.size(); k++) {
      const Node& nd = nodes_[ids[k]];
      if (c0 >= pad(nd.mU + nd.mV)) continue;
      const Node &a = nodes_[nd.c0], &c = nodes_[nd.c1];
      const int ldg = nd.rU + nd.rV;
      hssk_sweep_fwd_desc d{};
      d.wait0 = d.wait1 = -1;
      d.fsrc = dI + (size_t)c0 * kpad; d.ldf = kpad;
      d.zc = dI + nd.mU + (size_t)c0 * kpad; d.ldz_in = kpad;
      d.B01 = nd.B01; d.B10 = nd.B10;
      d.rU0 = a.rU; d.rU1 = c.rU; d.rV0 = a.rV; d.rV1 = c.rV;
      d.permV = nd.permV; d.XV = nd.XV;
      d.m = nd.mU; d.r = nd.rU; d.rv = nd.rV; d.mv = nd.mV;
      d.permU = nd.permU; d.XU = nd.XU; d.Rlq = nd.Rlq; d.Tinv = nd.Tinv; d.WQ = nd.WQ; d.Vt0T = nd.Vt0T;
      d.ft1 = nd.Gc + (size_t)c0 * ldg; d.ldp = ldg;
      d.z = nd.Gc + nd.rU + (size_t)c0 * ldg; d.ldz = ldg;
      d.y = ys[k];
      fd.push_back(d);
    }
    if (fd.empty()) continue;
    const int rc = hssk_ulv_fwd_sweep(ctx_, fd.data(), (int)fd.size(), CH);
    if (rc == 2) ok = false;
    else ck(rc);
  }
  ck(hssk_sync(ctx_));
  tmp_->rewind(mk);
  if (!ok || hssk_sweep_status(ctx_)) {
    for (int id : ids) nodes_[id].Gc = nullptr;
    chain_arena_->reset();
  }
}

static int solve_side_mode() {
  static const int v = [] { const char* e = std::getenv("STRUMPACK_AMD_SOLVE_SIDE"); return e ? std::atoi(e) : 0; }();
  return v;
}

// sr != 0: the subtree of node sr as a matrix of its own (factor_node): rows of b = the node's rows
// phase 0: the whole solve; 1: forward half, state kept in *ws; 2: backward half from *ws
void DeviceHSS::solve_sub(int sr, int nrhs, double* b, long long ldb, bool on_device, int phase, SolveWork* ws) {
  OpGuard op_guard(op_mu_);
  ensure_ready("solve");
  if (sr < 0 || sr >= (int)nodes_.size()) throw std::invalid_argument("solve: no such node");
  if (sr == 0 && !factored_) throw std::logic_error("solve: factor() has not been called (or shift() invalidated the factors)");
  const bool partial_root = partial_factored_ && !nodes_[0].leaf() && sr == nodes_[0].c0;   // child 0 after partial_factor()
  if (sr != 0 && sub_factored_ != sr && !(phase != 0 && partial_root))
    throw std::logic_error("solve: this child has not been factored (or a later factorization replaced its factors)");
  if (phase != 0 && (!ws || o_.world != 1)) throw std::logic_error("forward_solve / backward_solve: need a work object and a single-process matrix");
  if (sr != 0 && o_.world != 1) throw std::logic_error("solve: a child on its own needs a single-process matrix");
  if (nrhs <= 0 || nodes_[sr].m == 0) return;
  const int lo0 = nodes_[sr].lo;
  std::vector<std::vector<int>> sub_h, sub_d;
  if (sr != 0) { sub_h = sublists(own_by_height_, sr); sub_d = sublists(own_by_depth_, sr); }
  // the levels of the solve: the whole tree's, or those of the subtree below sr
  const std::vector<std::vector<int>>& lv_height = sr ? sub_h : own_by_height_;
  const std::vector<std::vector<int>>& lv_depth = sr ? sub_d : own_by_depth_;
  double t0 = now();
  // repeated solve on the same device buffer: replay the recorded sweep (no descriptor building, no staging)
  const bool plannable = on_device && o_.world == 1 && plans_enabled();
  const PlanKey key{1 + 4 * sr, 'N', nrhs, (const void*)b, (void*)b, ldb, ldb, 0.};
  if (plannable) {
    auto it = plans_.find(key);
    if (it != plans_.end() && it->second.plan) {
      ck(hssk_plan_replay(ctx_, it->second.plan));
      ck(hssk_sync(ctx_));
      if (hssk_sweep_status(ctx_)) throw std::runtime_error(std::string("solve: ") + hssk_last_error());
      stats_.t_solve = now() - t0;
      return;
    }
  }
  hssk_plan* rec = nullptr;
  // the first call on a buffer runs normally; the second one is recorded while it runs, later ones replay
  if (plannable && plans_.size() > 32) drop_plans();   // many different buffers: start over rather than grow
  if (plannable && ++plans_[key].seen == 2) {
    if (nrhs == 1 && sr == 0 && phase == 0) chain_blocks();   // (before the recording starts: its launches are not the sweep's)
    ck(hssk_plan_begin(ctx_, &rec));
  }
  struct EndRec { hssk_ctx* c; hssk_plan* p; bool done = false; ~EndRec() { if (p && !done) { hssk_plan_end(c); hssk_plan_destroy(p); } } } guard{ctx_, rec};
  if (phase == 1) {   // the state of a split solve lives in its own arena until the backward half has run
    ws->arena.reset(new Arena());
    ws->valid = false;
  }
  Arena& tmp = phase ? *ws->arena : (rec ? *plan_arena_ : *tmp_);   // a recorded sweep keeps its own work vectors
  if (!rec && !phase) tmp.rewind();
  const int N = nodes_[sr].m;
  double* db = b;
  long long lb = ldb;
  if (phase == 2) { db = ws->db; lb = N; }
  else if (!on_device) {
    db = tmp.dbl((size_t)N * nrhs);
    ck(hssk_memcpy2d_h2d(ctx_, db, sizeof(double) * N, b, sizeof(double) * ldb, sizeof(double) * N, nrhs));
    lb = N;
  }
  if (lb > 0x7fffffffLL) throw std::invalid_argument("solve: leading dimension too large");
  const size_t nn = nodes_.size();
  // f: assembled right-hand side of an inner node (mU rows; children write ft1 into it);
  // y: (mU - rU) rows; zc: children's z stacked (mV rows); xb: solution in the node's basis (mU rows)
  std::vector<double*> f_(nn, nullptr), y_(nn, nullptr), zc_(nn, nullptr), xb_(nn, nullptr);
  if (phase == 1) { ws->f = f_; ws->y = y_; ws->zc = zc_; ws->xb = xb_; }
  std::vector<double*>&f = phase ? ws->f : f_, &y = phase ? ws->y : y_, &zc = phase ? ws->zc : zc_, &xb = phase ? ws->xb : xb_;
  // (f, zc, xb are handed from node to node: carved from one block that the single-launch sweeps arm with a sentinel)
  size_t hand_total = 0;
  if (phase != 2)
  for (size_t i = 0; i < nn; i++) {
    if (!mine((int)i) || nodes_[i].leaf()) continue;
    const Node& nd = nodes_[i];
    const int mu = nodes_[nd.c0].rU + nodes_[nd.c1].rU, mv = nodes_[nd.c0].rV + nodes_[nd.c1].rV;
    hand_total += (size_t)(2 * std::max(mu, 1) + std::max(mv, 1)) * nrhs;
  }
  double* hand = phase == 2 ? nullptr : tmp.dbl(std::max<size_t>(hand_total, 1));
  if (phase != 2) {
    size_t off = 0;
    for (size_t i = 0; i < nn; i++) {
      if (!mine((int)i)) continue;
      const Node& nd = nodes_[i];
      if (!nd.leaf()) {
        const int mu = nodes_[nd.c0].rU + nodes_[nd.c1].rU, mv = nodes_[nd.c0].rV + nodes_[nd.c1].rV;
        f[i] = hand + off; off += (size_t)std::max(mu, 1) * nrhs;
        zc[i] = hand + off; off += (size_t)std::max(mv, 1) * nrhs;
        xb[i] = hand + off; off += (size_t)std::max(mu, 1) * nrhs;
      }
      if (nd.lvl > 0 && nd.mU > nd.rU) y[i] = tmp.dbl((size_t)(nd.mU - nd.rU) * nrhs);
    }
  }
  // few right-hand sides: the whole forward sweep (root solve included) and the whole backward sweep are ONE launch each
  // (hssk_ulv_fwd_sweep / _bwd_sweep: a workgroup per node, dependency flags between them) instead of 7 / 3 batched
  // launches per level
  static const bool no_fuse = [] { const char* e = std::getenv("STRUMPACK_AMD_NO_FUSED_SOLVE"); return e && e[0] == '1'; }();
  const bool fuse = nrhs <= fuse_max_nrhs() && !no_fuse;   // (more right-hand sides: the batched MFMA launches per level)
  if (fuse && phase != 2) ck(hssk_sweep_arm(ctx_, hand, (long long)hand_total));
  typedef std::vector<std::vector<int>> Levels;
  auto fwd_sweep = [&](const Levels& levels) -> bool {
    std::vector<hssk_sweep_fwd_desc> fd;
    std::vector<int> where(nn, -1);
    for (auto& ids : levels)
      for (int id : ids) {
        const Node& nd = nodes_[id];
        hssk_sweep_fwd_desc d{};
        d.wait0 = d.wait1 = -1;
        d.mv = nd.leaf() ? nd.m : nodes_[nd.c0].rV + nodes_[nd.c1].rV;
        if (nd.leaf()) { d.fsrc = db + (nd.lo - lo0); d.ldf = (int)lb; }
        else {
          const Node &a = nodes_[nd.c0], &c = nodes_[nd.c1];
          d.fsrc = f[id]; d.ldf = std::max(a.rU + c.rU, 1);
          d.B01 = nd.B01; d.B10 = nd.B10; d.zc = zc[id];
          d.rU0 = a.rU; d.rU1 = c.rU; d.rV0 = a.rV; d.rV1 = c.rV; d.ldz_in = std::max(a.rV + c.rV, 1);
          d.permV = nd.permV; d.XV = nd.XV;
          if (!d.B01 || !d.B10) return false;
          d.wait0 = where[nd.c0]; d.wait1 = where[nd.c1];
        }
        if (id == sr) {
          // root: x = LU^{-1} f
          d.m = nd.leaf() ? nd.m : nodes_[nd.c0].rU + nodes_[nd.c1].rU;
          if (d.m == 0) continue;
          d.LU = nd.LU; d.piv = nd.piv; d.TinvL = nd.Tinv; d.TinvU = nd.TinvU;
          if (!d.LU || !d.TinvL || !d.TinvU) return false;
          d.xroot = nd.leaf() ? db + (nd.lo - lo0) : xb[id];
          d.ldxr = nd.leaf() ? (int)lb : std::max(d.m, 1);
        } else {
          const Node& pa = nodes_[nd.parent];
          d.m = nd.mU; d.r = nd.rU; d.rv = nd.rV;
          if (!nd.leaf()) d.mv = nd.mV;
          d.permU = nd.permU; d.XU = nd.XU; d.Rlq = nd.Rlq; d.Tinv = nd.Tinv; d.WQ = nd.WQ; d.Vt0T = nd.Vt0T;
          d.ft1 = f[nd.parent] + (id == pa.c0 ? 0 : nodes_[pa.c0].rU);
          d.ldp = std::max(nodes_[pa.c0].rU + nodes_[pa.c1].rU, 1);
          d.z = zc[nd.parent] + (id == pa.c0 ? 0 : nodes_[pa.c0].rV);
          d.ldz = std::max(nodes_[pa.c0].rV + nodes_[pa.c1].rV, 1);
          d.y = y[id];
          if (nrhs == 1 && sr == 0 && !nd.leaf() && nd.Gc) { d.G = nd.Gc; d.ldg = nd.rU + nd.rV; }
          if (d.m > d.r && (!d.y || !d.Rlq || !d.Tinv || (d.r && !d.WQ) || (d.rv && !d.Vt0T))) return false;
        }
        where[id] = (int)fd.size();
        fd.push_back(d);
      }
    if (fd.empty()) return true;
    const int rc = hssk_ulv_fwd_sweep(ctx_, fd.data(), (int)fd.size(), nrhs);
    if (rc == 2) return false;
    ck(rc);
    return true;
  };
  // mode 0: every child; 1: inner children only (the leaves are finished by bwd(.., 2)); 2: (batched form only) leaf children
  auto bwd_sweep = [&](const Levels& levels, int mode) -> bool {
    std::vector<hssk_sweep_bwd_desc> bd;
    std::vector<int> where(nn, -1);
    for (auto& ids : levels)
      for (int id : ids) {
        const Node& nd = nodes_[id];
        if (nd.leaf()) continue;
        const Node& a = nodes_[nd.c0];
        const int cid[2] = {nd.c0, nd.c1};
        for (int q = 0; q < 2; q++) {
          if (!mine(cid[q])) continue;
          const Node& cn = nodes_[cid[q]];
          if (cn.mU == 0) continue;
          if (mode == 1 && cn.leaf()) continue;
          hssk_sweep_bwd_desc d{};
          d.Qt = cn.Qt; d.y = y[cid[q]]; d.xpart = xb[id] + (q ? a.rU : 0);
          d.out = cn.leaf() ? db + (cn.lo - lo0) : xb[cid[q]];
          d.m = cn.mU; d.r = cn.rU; d.ldx = std::max(a.rU + nodes_[nd.c1].rU, 1); d.ldo = cn.leaf() ? (int)lb : std::max(cn.mU, 1);
          d.wait0 = where[id];
          if (d.m > d.r && (!d.Qt || !d.y)) return false;
          where[cid[q]] = (int)bd.size();
          bd.push_back(d);
        }
      }
    if (bd.empty()) return true;
    const int rc = hssk_ulv_bwd_sweep(ctx_, bd.data(), (int)bd.size(), nrhs);
    if (rc == 2) return false;
    ck(rc);
    return true;
  };
  // ---- forward, one tree height
  auto fwd = [&](const std::vector<int>& ids, bool blocked = false) {
    if (ids.empty()) return;
    std::vector<hssk_gemm_desc> ga, gb, gc, gd, ge;
    std::vector<std::vector<hssk_gemm_desc>> sub;         // block substitution: stage 2 b = diagonal block b, 2 b + 1 = rows below
    std::vector<std::vector<hssk_rowgather_desc>> cpb;    // copies of y_b in front of stage 2 b
    std::vector<hssk_rowgather_desc> rg;
    std::vector<hssk_trsm_desc> ts;
    std::vector<hssk_lusolve_desc> ls;
    for (int id : ids) {
      const Node& nd = nodes_[id];
      if (nd.leaf()) continue;
      const Node &a = nodes_[nd.c0], &c = nodes_[nd.c1];
      const int ldf = std::max(a.rU + c.rU, 1), lz = std::max(a.rV + c.rV, 1);
      // f0 = ft1_0 - B01 z_1 ; f1 = ft1_1 - B10 z_0   (solve.hpp:88-99).  The children already wrote
      // ft1 - W1 (Q0^T y) into f (the -W1 Q0^T y term of solve.hpp:100-131 only needs child data).
      ga.push_back(hssk_gemm_desc{nd.B01, zc[id] + a.rV, f[id], a.rU, nrhs, c.rV, std::max(a.rU, 1), lz, ldf, 0, 0, -1.0, 1.0});
      ga.push_back(hssk_gemm_desc{nd.B10, zc[id], f[id] + a.rU, c.rU, nrhs, a.rV, std::max(c.rU, 1), lz, ldf, 0, 0, -1.0, 1.0});
    }
    if (!ga.empty()) ck(hssk_gemm_vbatched(ctx_, ga.data(), (int)ga.size()));
    for (int id : ids) {
      const Node& nd = nodes_[id];
      const double* fsrc = nd.leaf() ? db + (nd.lo - lo0) : f[id];
      const int mu = nd.leaf() ? nd.m : nodes_[nd.c0].rU + nodes_[nd.c1].rU;
      const int ldf = nd.leaf() ? (int)lb : std::max(mu, 1);
      if (id == sr) {
        // x = LU^{-1} f (solve.hpp:133-135)
        if (nd.leaf()) { if (mu) ls.push_back(hssk_lusolve_desc{nd.LU, nd.piv, db + (nd.lo - lo0), mu, nrhs, mu, (int)lb}); }
        else {
          rg.push_back(hssk_rowgather_desc{f[id], xb[id], nullptr, mu, nrhs, ldf, std::max(mu, 1), 0, 0});
          if (mu) ls.push_back(hssk_lusolve_desc{nd.LU, nd.piv, xb[id], mu, nrhs, mu, std::max(mu, 1)});
        }
        continue;
      }
      const Node& pa = nodes_[nd.parent];
      const int m = nd.mU, r = nd.rU;
      double* ft1 = f[nd.parent] + (id == pa.c0 ? 0 : nodes_[pa.c0].rU);
      const int ldp = std::max(nodes_[pa.c0].rU + nodes_[pa.c1].rU, 1);
      double* z = zc[nd.parent] + (id == pa.c0 ? 0 : nodes_[pa.c0].rV);
      const int ldz = std::max(nodes_[pa.c0].rV + nodes_[pa.c1].rV, 1);
      // f <- P^T f ; ft1 = f(0:r) ; y = L^{-1} (f(r:) - E ft1)    (solve.hpp:153-163)
      if (r) rg.push_back(hssk_rowgather_desc{fsrc, ft1, nd.permU, r, nrhs, ldf, ldp, 0, 0});
      if (m > r) {
        rg.push_back(hssk_rowgather_desc{fsrc, y[id], nd.permU + r, m - r, nrhs, ldf, m - r, 0, 0});
        if (r) gd.push_back(hssk_gemm_desc{nd.XU, ft1, y[id], m - r, nrhs, r, r, ldp, m - r, 1, 0, -1.0, 1.0});
        if (blocked && nd.Tinv) {
          // y <- R~^{-T} y on 64-row blocks with the diagonal blocks inverted at factor time: q / 64 pairs of GEMMs over
          // all right-hand sides instead of q dependent steps per group of four
          const int q = m - r;
          for (int b0 = 0, blk = 0; b0 < q; b0 += 64, blk++) {
            const int nb = std::min(64, q - b0);
            if ((int)sub.size() <= 2 * blk + 1) sub.resize(2 * blk + 2);
            double* t = tmp.dbl((size_t)nb * nrhs);
            cpb.resize(sub.size());
            cpb[2 * blk].push_back(hssk_rowgather_desc{y[id] + b0, t, nullptr, nb, nrhs, q, nb, 0, 0});
            sub[2 * blk].push_back(hssk_gemm_desc{nd.Tinv + (size_t)blk * 64 * 64, t, y[id] + b0, nb, nrhs, nb, 64, nb, q, 0, 0, 1.0, 0.0});
            const int rest = q - b0 - nb;
            if (rest > 0)   // rows below -= R~(b, below)^T y_b
              sub[2 * blk + 1].push_back(hssk_gemm_desc{nd.Rlq + b0 + (size_t)(b0 + nb) * m, y[id] + b0, y[id] + b0 + nb, rest, nrhs, nb, m, q, q, 1, 0, -1.0, 1.0});
          }
        } else {
          ts.push_back(hssk_trsm_desc{nd.Rlq, y[id], m - r, nrhs, m, m - r, 0, 1, 0});
        }
        if (r && nd.WQ) {
          gc.push_back(hssk_gemm_desc{nd.WQ, y[id], ft1, r, nrhs, m - r, r, m - r, ldp, 0, 0, -1.0, 1.0});   // ft1 -= WQ y
        } else if (r) {
          // ft1 -= W1 (Q0^T y),  Q0^T y = Q~(:, :m-r) y
          double* t = tmp.dbl((size_t)m * nrhs);
          gb.push_back(hssk_gemm_desc{nd.Qt, y[id], t, m, nrhs, m - r, m, m - r, m, 0, 0, 1.0, 0.0});
          gc.push_back(hssk_gemm_desc{nd.W1, t, ft1, r, nrhs, m, r, m, ldp, 0, 0, -1.0, 1.0});
        }
      }
      // z = V^H [z0; z1] + Vt0^H y   (leaf: z = Vt0^H y)          (solve.hpp:164-192)
      const int rv = nd.rV;
      if (rv) {
        double zbeta = 0.0;
        if (!nd.leaf()) {
          const int mv = nd.mV;
          rg.push_back(hssk_rowgather_desc{zc[id], z, nd.permV, rv, nrhs, std::max(mv, 1), ldz, 0, 0});
          if (mv > rv) {
            double* t = tmp.dbl((size_t)(mv - rv) * nrhs);
            rg.push_back(hssk_rowgather_desc{zc[id], t, nd.permV + rv, mv - rv, nrhs, std::max(mv, 1), mv - rv, 0, 0});
            gd.push_back(hssk_gemm_desc{nd.XV, t, z, rv, nrhs, mv - rv, rv, mv - rv, ldz, 0, 0, 1.0, 1.0});
          }
          zbeta = 1.0;
        }
        if (m > r && rv > 0) ge.push_back(hssk_gemm_desc{nd.Vt0T, y[id], z, rv, nrhs, m - r, rv, m - r, ldz, 0, 0, 1.0, zbeta});   // z (+)= Vt0^T y
        else if (nd.leaf()) ge.push_back(hssk_gemm_desc{z, z, z, rv, nrhs, 0, 1, 1, ldz, 0, 0, 1.0, 0.0});  // z = 0
      }
    }
    if (!rg.empty()) ck(hssk_gather_rows(ctx_, rg.data(), (int)rg.size()));
    if (!gd.empty()) ck(hssk_gemm_vbatched(ctx_, gd.data(), (int)gd.size()));
    if (!ts.empty()) ck(hssk_trsm_vbatched(ctx_, ts.data(), (int)ts.size()));
    for (size_t st = 0; st < sub.size(); st++) {
      if (st < cpb.size() && !cpb[st].empty()) ck(hssk_gather_rows(ctx_, cpb[st].data(), (int)cpb[st].size()));
      if (!sub[st].empty()) ck(hssk_gemm_vbatched(ctx_, sub[st].data(), (int)sub[st].size()));
    }
    if (!ge.empty()) ck(hssk_gemm_vbatched(ctx_, ge.data(), (int)ge.size()));
    if (!gb.empty()) ck(hssk_gemm_vbatched(ctx_, gb.data(), (int)gb.size()));
    if (!gc.empty()) ck(hssk_gemm_vbatched(ctx_, gc.data(), (int)gc.size()));
    if (!ls.empty()) ck(hssk_getrs_vbatched(ctx_, ls.data(), (int)ls.size()));
  };
  // ---- backward, one depth (solve.hpp:199-238): x_c = Q_c^H [y_c ; x(part)] = Q~(:, :mc-rc) y_c + Q~(:, mc-rc:) xpart
  auto bwd = [&](const std::vector<int>& ids, int mode = 0) {
    std::vector<hssk_gemm_desc> g1, g2;
    std::vector<hssk_rowgather_desc> cp;
    for (int id : ids) {
      const Node& nd = nodes_[id];
      if (nd.leaf()) continue;
      const Node &a = nodes_[nd.c0], &c = nodes_[nd.c1];
      const int mu = a.rU + c.rU;
      const double* x = xb[id];
      const int ldx = std::max(mu, 1);
      const Node* ch[2] = {&a, &c};
      const int cid[2] = {nd.c0, nd.c1};
      for (int q = 0; q < 2; q++) {
        if (!mine(cid[q])) continue;  // the other ranks' subtrees continue on their owners
        const Node& cn = *ch[q];
        // mode 1: inner children only; 2: leaves only; 3: leaves only, the part that needs no parent (Q~(:, 0:q) y); 4: leaves
        // only, the rest
        if ((mode == 1 && cn.leaf()) || (mode >= 2 && !cn.leaf())) continue;
        const int mc = cn.mU, rc = cn.rU;
        const double* xpart = x + (q ? a.rU : 0);
        double* out = cn.leaf() ? db + (cn.lo - lo0) : xb[cid[q]];
        const int ldo = cn.leaf() ? (int)lb : std::max(mc, 1);
        if (mc > rc) {
          if (mode != 4) g1.push_back(hssk_gemm_desc{cn.Qt, y[cid[q]], out, mc, nrhs, mc - rc, mc, mc - rc, ldo, 0, 0, 1.0, 0.0});
          if (rc && mode != 3) g2.push_back(hssk_gemm_desc{cn.Qt + (size_t)(mc - rc) * mc, xpart, out, mc, nrhs, rc, mc, ldx, ldo, 0, 0, 1.0, 1.0});
        } else if (mc && mode != 3) {
          cp.push_back(hssk_rowgather_desc{xpart, out, nullptr, mc, nrhs, ldx, ldo, 0, 0});
        }
      }
    }
    if (!g1.empty()) ck(hssk_gemm_vbatched(ctx_, g1.data(), (int)g1.size()));
    if (!g2.empty()) ck(hssk_gemm_vbatched(ctx_, g2.data(), (int)g2.size()));
    if (!cp.empty()) ck(hssk_gather_rows(ctx_, cp.data(), (int)cp.size()));
  };
  // many right-hand sides (hybrid, see mult_sub): the leaf level as batched MFMA GEMMs over all right-hand sides (the
  // blocks of the leaves -- X, R~, WQ, Vt0, Q~: most of the bytes -- read once), the inner levels in the single launches
  bool big_leaves = false;   // (leaves beyond the sweep kernels' 256 rows -- leaf size 512 -- take the same route for any nrhs)
  if (!lv_height.empty())
    for (int id : lv_height[0]) big_leaves = big_leaves || nodes_[id].m > 256;
  const bool hybrid = fuse && !dist_subtree_ && (nrhs >= hybrid_nrhs() || big_leaves) && lv_height.size() > 1;
  bool fwd_done = phase == 2;   // (the backward half: the forward sweep ran in forward_solve)
  static const bool no_side = [] { const char* e = std::getenv("STRUMPACK_AMD_NO_SIDE_STREAM"); return e && e[0] == '1'; }();
  const bool side = hybrid && !no_side && phase == 0 && solve_side_mode() != 0;   // (a split solve keeps every launch on the main stream)
  std::vector<int> leaf_parents;   // the leaves' parents, whatever their depth: one batch
  if (hybrid)
    for (auto& ids : lv_depth) leaf_parents.insert(leaf_parents.end(), ids.begin(), ids.end());
  if (hybrid && phase != 2) {
    Levels inner(lv_height.begin() + 1, lv_height.end());
    // the leaf level: one launch in the matrix-core form of the sweep when it takes the leaves (kernels/hssk_sweep_mma.h),
    // else batched launches over all right-hand sides
    bool leaves_done = false;
    if (nrhs >= hssk_sweep_mma_min_nrhs() && !big_leaves) {
      struct Req { hssk_ctx* c; Req(hssk_ctx* c_) : c(c_) { hssk_sweep_require_mma(c, 1); } ~Req() { hssk_sweep_require_mma(c, 0); } } req(ctx_);
      Levels leaf_level{lv_height[0]};
      leaves_done = fwd_sweep(leaf_level);
    }
    if (!leaves_done) fwd(lv_height[0], true);
    // the leaves' Q~(:, 0:q) y -- most of the backward step, and independent of the levels above -- can run on the side stream
    // next to the inner levels (STRUMPACK_AMD_SOLVE_SIDE=1: next to the forward sweep's, =2: next to the backward sweep's).  Off
    // since round 6: the product now keeps two workgroups per CU busy (gemm_tall_kernel: 0.16 instead of 0.25 ms) and takes the
    // memory system from the chain of inner levels -- 0.543 (=1) / 0.570 (=2) against 0.534 ms one after the other at N = 1e5
    // (gpurun_out/sweeps_n64_a.txt); the mat-vec, whose leaf product needs nothing from the tree, keeps its side stream (0.248
    // against 0.316 ms).
    const bool side_early = solve_side_mode() == 1;
    auto side_leaves = [&] {
      ck(hssk_side_begin(ctx_));
      struct End { hssk_ctx* c; ~End() { hssk_side_end(c); } } end{ctx_};
      bwd(leaf_parents, 3);
    };
    if (side && side_early) side_leaves();
    if (!fwd_sweep(inner))
      for (auto& ids : inner) fwd(ids);
    if (side && !side_early) side_leaves();
    fwd_done = true;
  }
  if (!fwd_done && !(fuse && fwd_sweep(lv_height)))
    for (auto& ids : lv_height) fwd(ids);
  if (dist_subtree_) {
    // publish ft1' (rU x nrhs) and z (rV x nrhs) of the cut nodes into every rank's top buffers
    const int G = o_.world, me = o_.rank;
    int ru = 1, rv = 1;
    for (int g = 0; g < G; g++) { ru = std::max(ru, nodes_[cut_nodes_[g]].rU); rv = std::max(rv, nodes_[cut_nodes_[g]].rV); }
    const size_t blk = (size_t)(ru + rv) * nrhs;
    double* buf = tmp.dbl(blk * G);
    auto slices = [&](int g, double*& pf, int& ldf, double*& pz, int& ldz) {
      const int id = cut_nodes_[g];
      const Node& pa = nodes_[nodes_[id].parent];
      pf = f[nodes_[id].parent] + (id == pa.c0 ? 0 : nodes_[pa.c0].rU);
      ldf = std::max(nodes_[pa.c0].rU + nodes_[pa.c1].rU, 1);
      pz = zc[nodes_[id].parent] + (id == pa.c0 ? 0 : nodes_[pa.c0].rV);
      ldz = std::max(nodes_[pa.c0].rV + nodes_[pa.c1].rV, 1);
    };
    {
      double *pf, *pz; int ldf, ldz;
      slices(me, pf, ldf, pz, ldz);
      const Node& c = nodes_[cut_nodes_[me]];
      std::vector<hssk_rowgather_desc> pk;
      if (c.rU) pk.push_back(hssk_rowgather_desc{pf, buf + blk * me, nullptr, c.rU, nrhs, ldf, ru, 0, 0});
      if (c.rV) pk.push_back(hssk_rowgather_desc{pz, buf + blk * me + (size_t)ru * nrhs, nullptr, c.rV, nrhs, ldz, rv, 0, 0});
      if (!pk.empty()) ck(hssk_gather_rows(ctx_, pk.data(), (int)pk.size()));
    }
    comm(buf, (long long)(sizeof(double) * blk));
    std::vector<hssk_rowgather_desc> up;
    for (int g = 0; g < G; g++) {
      if (g == me) continue;
      double *pf, *pz; int ldf, ldz;
      slices(g, pf, ldf, pz, ldz);
      const Node& c = nodes_[cut_nodes_[g]];
      if (c.rU) up.push_back(hssk_rowgather_desc{buf + blk * g, pf, nullptr, c.rU, nrhs, ru, ldf, 0, 0});
      if (c.rV) up.push_back(hssk_rowgather_desc{buf + blk * g + (size_t)ru * nrhs, pz, nullptr, c.rV, nrhs, rv, ldz, 0, 0});
    }
    if (!up.empty()) ck(hssk_gather_rows(ctx_, up.data(), (int)up.size()));
    if (!(fuse && fwd_sweep(top_by_height_)))
      for (auto& ids : top_by_height_) fwd(ids);
    if (!(fuse && bwd_sweep(top_by_depth_, 0)))
      for (auto& ids : top_by_depth_) bwd(ids);
  }
  if (phase == 1) {
    // ---- forward half: x_root (and the reduced right-hand side of a partial solve) to the host, the state stays in *ws
    const Node& rt = nodes_[sr];
    const int mu = rt.leaf() ? rt.m : nodes_[rt.c0].rU + nodes_[rt.c1].rU;
    const double* dxr = rt.leaf() ? db : xb[sr];
    const long long ldr_dev = rt.leaf() ? lb : std::max(mu, 1);
    if (mu && ws->xroot) ck(hssk_memcpy2d_d2h(ctx_, ws->xroot, sizeof(double) * ws->ldx, dxr, sizeof(double) * ldr_dev, sizeof(double) * mu, nrhs));
    if (ws->partial && ws->reduced) {
      // reduced_rhs = Vhat^* x_root + V^* [z_0; z_1]   (HSSMatrix.solve.hpp:139-154; Vhat = the root's Vt0 of a partial factorization)
      const int rv = rt.rV;
      if (sr == 0) throw std::logic_error("forward_solve: a partial solve needs a node with a column basis (child 0 of the root)");
      if (rv) {
        if (!rt.Vt0) throw std::logic_error("forward_solve: partial, but the factorization kept no Vhat (partial_factor() first)");
        double* red = tmp.dbl((size_t)rv * nrhs);
        if (mu) {
          hssk_gemm_desc g{rt.Vt0, dxr, red, rv, nrhs, mu, std::max(rt.mU, 1), (int)ldr_dev, rv, 1, 0, 1.0, 0.0};
          ck(hssk_gemm_vbatched(ctx_, &g, 1));
        } else ck(hssk_memset_zero(ctx_, red, (long long)sizeof(double) * rv * nrhs));
        if (!rt.leaf()) {
          const int mv = rt.mV;
          hssk_rowgather_desc r0{zc[sr], red, rt.permV, rv, nrhs, std::max(mv, 1), rv, 0, 1};
          ck(hssk_gather_rows(ctx_, &r0, 1));
          if (mv > rv) {
            double* t = tmp.dbl((size_t)(mv - rv) * nrhs);
            hssk_rowgather_desc r1{zc[sr], t, rt.permV + rv, mv - rv, nrhs, std::max(mv, 1), mv - rv, 0, 0};
            ck(hssk_gather_rows(ctx_, &r1, 1));
            hssk_gemm_desc g{rt.XV, t, red, rv, nrhs, mv - rv, rv, mv - rv, rv, 0, 0, 1.0, 1.0};
            ck(hssk_gemm_vbatched(ctx_, &g, 1));
          }
        }
        ck(hssk_memcpy2d_d2h(ctx_, ws->reduced, sizeof(double) * ws->ldr, red, sizeof(double) * rv, sizeof(double) * rv, nrhs));
      }
    }
    ck(hssk_sync(ctx_));
    if (hssk_sweep_status(ctx_)) throw std::runtime_error(std::string("forward_solve: ") + hssk_last_error());
    ws->db = db; ws->nrhs = nrhs; ws->sr = sr; ws->valid = true;
    return;
  }
  if (phase == 2) {
    // ---- backward half: the caller's x_root (possibly updated) back into the root's slot
    const Node& rt = nodes_[sr];
    const int mu = rt.leaf() ? rt.m : nodes_[rt.c0].rU + nodes_[rt.c1].rU;
    double* dxr = rt.leaf() ? db : xb[sr];
    const long long ldr_dev = rt.leaf() ? lb : std::max(mu, 1);
    if (mu) ck(hssk_memcpy2d_h2d(ctx_, dxr, sizeof(double) * ldr_dev, ws->xroot, sizeof(double) * ws->ldx, sizeof(double) * mu, nrhs));
  }
  if (hybrid) {
    if (!bwd_sweep(lv_depth, 1))
      for (auto& ids : lv_depth) bwd(ids, 1);
    if (side) ck(hssk_side_join(ctx_));
    bwd(leaf_parents, side ? 4 : 2);
  } else if (!(fuse && bwd_sweep(lv_depth, 0)))
    for (auto& ids : lv_depth) bwd(ids);
  if (dist_subtree_) allgather_rows(db, lb, nrhs);
  if (!on_device) ck(hssk_memcpy2d_d2h(ctx_, b, sizeof(double) * ldb, db, sizeof(double) * N, sizeof(double) * N, nrhs));
  if (rec) { ck(hssk_plan_end(ctx_)); guard.done = true; plans_[key].plan = rec; }
  ck(hssk_sync(ctx_));
  if (phase == 2) { ws->arena.reset(); ws->valid = false; }
  if (hssk_sweep_status(ctx_)) throw std::runtime_error(std::string("solve: ") + hssk_last_error());
  stats_.t_solve = now() - t0;
  {
    double fs = 0;
    for (auto& nd : nodes_) {
      if (nd.lvl == 0) { const int mu = nd.leaf() ? nd.m : nodes_[nd.c0].rU + nodes_[nd.c1].rU; fs += 2.0 * mu * (double)mu; continue; }
      const double m = nd.mU, r = nd.rU, k = m - r, rv = nd.rV;
      fs += 2.0 * k * r + k * k + 2.0 * k * rv + 2.0 * m * m + 2.0 * k * m;
      if (!nd.leaf()) fs += 4.0 * nodes_[nd.c0].rU * (double)nodes_[nd.c1].rV;
    }
    stats_.f_solve = fs * nrhs;
    // blocks read by the sweeps (fused path: X, the off-diagonal part of R~ + its inverted diagonal blocks, WQ, Vt0, B, XV
    // going up, Q~ going down; the unfused path reads W1 and Q~(:, 0:q) instead of WQ)
    double bs = 0;
    for (auto& nd : nodes_) {
      if (nd.lvl == 0) { const double mu = nd.leaf() ? nd.m : nodes_[nd.c0].rU + nodes_[nd.c1].rU; bs += mu * mu; continue; }
      const double m = nd.mU, r = nd.rU, k = m - r, rv = nd.rV;
      bs += r * k + k * (k + 1) / 2 + (fuse ? r * k : r * m + m * k) + k * rv + m * m;   // (per group of four right-hand sides when fused)
      if (!nd.leaf()) bs += (double)nodes_[nd.c0].rU * nodes_[nd.c1].rV + (double)nodes_[nd.c1].rU * nodes_[nd.c0].rV + rv * (nd.mV - rv);
    }
    stats_.b_solve = 8.0 * bs;
  }
}

}  // namespace HSS
}  // namespace strumpack
