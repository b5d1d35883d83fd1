// DeviceHSS: ULV factorization (HSSMatrix.factor.hpp:51-147).
#include "hss_engine_internal.hpp"
#include <cstdio>

namespace strumpack {
namespace HSS {

// ---------------------------------------------------------------------------------------------
// ULV factorization (HSSMatrix.factor.hpp:51-147)
// ---------------------------------------------------------------------------------------------
void DeviceHSS::factor() { factor_sub(0, false); }

// HSSMatrix::partial_factor (HSSMatrix.factor.hpp:43-49): ULV-factor the (0,0) block only -- child(0) is eliminated as
// the root of its own subtree -- and keep its reduced column basis Vhat (HSSFactors::Vhat(), HSSExtra.hpp:191) for
// the Schur complement update of the (1,1) block.
void DeviceHSS::partial_factor() {
  OpGuard op_guard(op_mu_);
  if (nodes_[0].leaf()) return;
  if (o_.world != 1) throw std::logic_error("partial_factor: needs a single-process matrix");
  factor_sub(nodes_[0].c0, true);
}

// The factorization is level-synchronous with no host synchronisation between its levels; its state lives in frun_ so that
// the levels can be enqueued one by one -- by factor_sub all at once, or by the compression as it settles them (factor ahead).
void DeviceHSS::factor_begin(int sr, bool partial, hssk_ctx* cx) {
  drop_plans();   // recorded sweeps reference the old factors
  fact_->reset();
  stats_.f_ulv = 0;
  for (auto& nd : nodes_) nd.Qt = nd.Rlq = nd.W1 = nd.Vt0 = nd.Dt = nd.Vt1 = nd.LU = nd.WQ = nd.Tinv = nd.TinvU = nd.Vt0T = nd.Gc = nullptr, nd.piv = nullptr;
  chain_built_ = false;
  if (chain_arena_) chain_arena_->reset();
  const size_t nn = nodes_.size();
  frun_ = FactorRun();
  frun_.active = true;
  frun_.sr = sr; frun_.partial = partial; frun_.cx = cx;
  frun_.Dh.assign(nn, nullptr); frun_.Vh.assign(nn, nullptr); frun_.Vd.assign(nn, nullptr);
  // The cut nodes of a distributed tree keep a compact Dt of their own -- it travels through exchange_cut_factor().
  frun_.is_cut.assign(nn, 0);
  if (dist_subtree_) for (int c : cut_nodes_) frun_.is_cut[c] = 1;
}

// the dense column bases [I; X^T] in row order (leaves: straight into Vh; inner nodes: Vd, multiplied by the children's Vt1 at
// the node's level): one launch for all the nodes listed
void DeviceHSS::factor_prep(const std::vector<int>& ids) {
  FactorRun& f = frun_;
  std::vector<hssk_basis_desc> bd;
  for (int id : ids) {
    const Node& nd = nodes_[id];
    if (id == f.sr && !f.partial) continue;
    f.Vh[id] = fact_->dbl((size_t)std::max(nd.mU, 1) * std::max(nd.rV, 1));
    if (!nd.rV) continue;
    double* out = f.Vh[id];
    if (!nd.leaf()) out = f.Vd[id] = fact_->dbl((size_t)nd.mV * nd.rV);
    bd.push_back(hssk_basis_desc{nd.XV, nd.permV, out, nd.mV, nd.rV, nd.rV, nd.mV});
  }
  if (!bd.empty()) ck(hssk_basis_dense(f.cx, bd.data(), (int)bd.size()));
}

void DeviceHSS::factor_level(const std::vector<int>& ids) {
  if (ids.empty()) return;
  FactorRun& f = frun_;
  hssk_ctx* cx = f.cx;
  const int sr = f.sr;
  const bool partial = f.partial;
  std::vector<double*>&Dh = f.Dh, &Vh = f.Vh, &Vd = f.Vd;
  // inverted diagonal blocks for the single-launch solve sweeps: nothing in the factorization reads them, so the
  // descriptors of all levels are collected (f.ti) and take ONE launch at the end (a launch per level was 10-20 us each)
  // A node's reduced block Dt (rU x rU) is written by its products straight into the diagonal block of the PARENT's Dh
  // (allocated here, ahead of the parent's level): no copy launch per level.
  auto dt_slot = [&](int id, int r, int& ld) -> double* {
    const Node& nd = nodes_[id];
    if (nd.parent < 0 || f.is_cut[id]) { ld = std::max(r, 1); return fact_->dbl((size_t)ld * ld); }
    const Node& pa = nodes_[nd.parent];
    const int mu = nodes_[pa.c0].rU + nodes_[pa.c1].rU;
    ld = std::max(mu, 1);
    if (!Dh[nd.parent]) Dh[nd.parent] = fact_->dbl((size_t)ld * ld);
    const int off = id == pa.c0 ? 0 : nodes_[pa.c0].rU;
    return Dh[nd.parent] + off + (size_t)off * ld;
  };
  // Inner levels whose nodes fit one workgroup each: the level is ONE launch (hssk_ulv_node_vbatched: assembly, split, QR, Q~
  // and the products behind it per node) instead of five -- 65 - 75 us per level at N = 1e5 in launches of 10 - 17 us for a few
  // 80-row blocks.  STRUMPACK_AMD_NO_ULV_NODE=1 keeps the batched steps.
  static const bool no_node = [] { const char* e = std::getenv("STRUMPACK_AMD_NO_ULV_NODE"); return e && e[0] == '1'; }();
  bool fused = !no_node;
  for (int id : ids) {
    const Node& nd = nodes_[id];
    if (!fused) break;
    if (nd.leaf()) { fused = false; break; }
    const Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
    const int mu = a.rU + b.rU;
    const bool vh = (id != sr || partial) && nd.rV > 0;
    if (mu <= 0 || !a.Vt1 || !b.Vt1) fused = false;
    if (id != sr && (nd.mU != mu || nd.mU <= nd.rU)) fused = false;
    if (fused && !hssk_ulv_node_fits(mu, id == sr ? 0 : nd.rU, vh ? nd.rV : 0, a.rU, b.rU, a.rV, b.rV)) fused = false;
  }
  std::vector<hssk_ulvnode_desc> un;
  // ---- assemble Dh (mU x mU) and Vh (mU x rV)
  std::vector<hssk_colgather_desc> cp;
  std::vector<hssk_gemm_desc> g0, g1;
  for (int id : ids) {
    Node& nd = nodes_[id];
    const bool root = id == sr;
    const int mu = nd.leaf() ? nd.m : nodes_[nd.c0].rU + nodes_[nd.c1].rU;
    if (nd.leaf() && !root) {
      Dh[id] = nd.D;   // only read below (the root's block is factored in place: it gets a copy)
    } else if (nd.leaf()) {
      Dh[id] = fact_->dbl((size_t)std::max(mu, 1) * std::max(mu, 1));
      cp.push_back(hssk_colgather_desc{nd.D, Dh[id], nullptr, nd.m, nd.m, nd.m, nd.m, 0});
    } else {
      Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
      // D = [Dt0, B01 Vt1_1^T ; B10 Vt1_0^T, Dt1]; the diagonal blocks are already in place (dt_slot) unless the children
      // are cut nodes
      if (!Dh[id]) {
        Dh[id] = fact_->dbl((size_t)std::max(mu, 1) * std::max(mu, 1));
        cp.push_back(hssk_colgather_desc{a.Dt, Dh[id], nullptr, a.rU, a.rU, std::max(a.rU, 1), std::max(mu, 1), 0});
        cp.push_back(hssk_colgather_desc{b.Dt, Dh[id] + a.rU + (size_t)a.rU * mu, nullptr, b.rU, b.rU, std::max(b.rU, 1), std::max(mu, 1), 0});
      }
      if (!fused) {
        g0.push_back(hssk_gemm_desc{nd.B01, b.Vt1, Dh[id] + (size_t)a.rU * mu, a.rU, b.rU, b.rV, std::max(a.rU, 1), std::max(b.rU, 1), std::max(mu, 1), 0, 1, 1.0, 0.0});
        g0.push_back(hssk_gemm_desc{nd.B10, a.Vt1, Dh[id] + a.rU, b.rU, a.rU, a.rV, std::max(b.rU, 1), std::max(a.rU, 1), std::max(mu, 1), 0, 1, 1.0, 0.0});
      }
      stats_.f_ulv += 2.0 * a.rU * (double)b.rU * (a.rV + b.rV);
    }
    if ((!root || partial) && !nd.leaf() && nd.rV) {
      Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
      // Vh = [Vt1_0 Vd(0:rV0, :) ; Vt1_1 Vd(rV0:, :)]   (Vd: the node's dense column basis, factor_prep)
      if (!fused) {
        g1.push_back(hssk_gemm_desc{a.Vt1, Vd[id], Vh[id], a.rU, nd.rV, a.rV, std::max(a.rU, 1), nd.mV, nd.mU, 0, 0, 1.0, 0.0});
        g1.push_back(hssk_gemm_desc{b.Vt1, Vd[id] + a.rV, Vh[id] + a.rU, b.rU, nd.rV, b.rV, std::max(b.rU, 1), nd.mV, nd.mU, 0, 0, 1.0, 0.0});
      }
      stats_.f_ulv += 2.0 * nd.rV * ((double)a.rU * a.rV + (double)b.rU * b.rV);
    }
  }
  // the assembly half of a fused node's descriptor
  auto node_desc = [&](int id) {
    const Node& nd = nodes_[id];
    const Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
    hssk_ulvnode_desc d{};
    d.B01 = nd.B01; d.B10 = nd.B10; d.Vt1a = a.Vt1; d.Vt1b = b.Vt1;
    d.ra = a.rU; d.rb = b.rU; d.rva = a.rV; d.rvb = b.rV;
    d.m = a.rU + b.rU;
    d.Dh = Dh[id];
    const bool vh = (id != sr || partial) && nd.rV > 0;
    d.Vd = vh ? Vd[id] : nullptr;
    d.Vh = vh ? Vh[id] : nullptr;
    d.rv = vh ? nd.rV : 0;
    return d;
  };
  if (!cp.empty()) ck(hssk_gather_cols(cx, cp.data(), (int)cp.size()));
  // (the coupling products into Dh and the products that build Vh are independent of each other: one batched launch)
  g0.insert(g0.end(), g1.begin(), g1.end());
  if (!g0.empty()) ck(hssk_gemm_vbatched(cx, g0.data(), (int)g0.size()));
  // ---- eliminate
  std::vector<hssk_elem_desc> ge;
  std::vector<hssk_ulvsplit_desc> us;
  std::vector<hssk_gemm_desc> g2, g3;
  std::vector<hssk_qr_desc> qr;
  std::vector<hssk_lu_desc> lu;
  for (int id : ids) {
    Node& nd = nodes_[id];
    if (id == sr) {
      const int mu = nd.leaf() ? nd.m : nodes_[nd.c0].rU + nodes_[nd.c1].rU;
      nd.LU = Dh[id];
      if (partial) nd.Vt0 = Vh[id];   // Vhat: mu x rV, the column basis in the reduced unknowns
      nd.piv = (int*)fact_->alloc(sizeof(int) * (std::max(mu, 1) + 1));
      if (mu) lu.push_back(hssk_lu_desc{nd.LU, mu, mu, nd.piv, nd.piv + mu});
      if (mu && mu <= 256) {
        const size_t nblk = (size_t)(mu + 63) / 64;
        nd.Tinv = fact_->dbl(nblk * 4096);
        nd.TinvU = fact_->dbl(nblk * 4096);
        f.ti.push_back(hssk_trtri_desc{nd.LU, nd.Tinv, mu, mu, 2});
        f.ti.push_back(hssk_trtri_desc{nd.LU, nd.TinvU, mu, mu, 1});
      }
      stats_.f_ulv += 2.0 / 3.0 * mu * (double)mu * mu;
      if (fused) un.push_back(node_desc(id));   // (eliminate = 0: the assembly; the LU follows below)
      continue;
    }
    const int m = nd.mU, r = nd.rU, rv = nd.rV;
    if (m > r) {
      // W1 = (P^T D)(0:r, :) ; W0^T = (P^T D)(r:, :)^T - W1^T X       (factor.hpp:109-118)
      nd.W1 = fact_->dbl((size_t)std::max(r, 1) * m);
      nd.Rlq = fact_->dbl((size_t)m * (m - r));
      nd.Qt = fact_->dbl((size_t)m * m);
      nd.Vt1 = fact_->dbl((size_t)std::max(r, 1) * std::max(rv, 1));
      int ldt = 1;
      nd.Dt = dt_slot(id, r, ldt);
      if (fused) {
        hssk_ulvnode_desc d = node_desc(id);
        d.eliminate = 1;
        d.perm = nd.permU; d.X = nd.XU; d.r = r;
        d.W1 = nd.W1; d.Rlq = nd.Rlq; d.Qt = nd.Qt; d.tau = fact_->dbl((size_t)2 * m);
        d.Vt1 = nd.Vt1; d.Dt = nd.Dt; d.ldt = ldt;
        if (rv) { nd.Vt0T = fact_->dbl((size_t)rv * (m - r)); d.Vt0T = nd.Vt0T; }
        if (r) { nd.WQ = fact_->dbl((size_t)r * (m - r)); d.WQ = nd.WQ; }
        un.push_back(d);
        nd.Tinv = fact_->dbl((size_t)((m - r + 63) / 64) * 4096);
        f.ti.push_back(hssk_trtri_desc{nd.Rlq, nd.Tinv, m - r, m, 0});
        const double k = m - r;
        stats_.f_ulv += 2.0 * k * r * m + (2.0 * m * k * k - 2.0 / 3.0 * k * k * k) + (4.0 * m * m * k - 2.0 * m * k * k) / 1.0 * 0.5 + 2.0 * m * m * rv + 2.0 * r * (double)r * m;
        continue;
      }
      if (m <= 256) {   // one fused launch (hssk_ulv_split); larger blocks: two row gathers and a product
        us.push_back(hssk_ulvsplit_desc{Dh[id], m, m, r, nd.permU, nd.XU, std::max(r, 1), nd.W1, std::max(r, 1), nd.Rlq, m});
      } else {
        if (r) ge.push_back(hssk_elem_desc{Dh[id], m, nd.permU, nullptr, 0, 0, nd.W1, r, m, r, 0});
        ge.push_back(hssk_elem_desc{Dh[id], m, nd.permU + r, nullptr, 0, 0, nd.Rlq, m - r, m, m, 1});
        if (r) g2.push_back(hssk_gemm_desc{nd.W1, nd.XU, nd.Rlq, m, m - r, r, r, r, m, 1, 0, -1.0, 1.0});
      }
      // LQ(W0) == QR(W0^T): Q~ (m x m) = Q^T, R~ = L^T                  (factor.hpp:122)
      double* wk = fact_->dbl((size_t)2 * m);   // (from the factors' own arena: the levels are enqueued back to back, and a run ahead shares nothing with the compression)
      qr.push_back(hssk_qr_desc{nd.Rlq, m, m, m - r, nd.Qt, m, m, nullptr, wk});
      // Vt0 = Q0 Vh = Q~(:, :m-r)^T Vh, kept TRANSPOSED (Vt0^T = Vh^T Q~(:, 0:m-r): rows contiguous for the solve sweep; the
      // per-level solve reads the same array) ; Vt1 = Q~(:, m-r:)^T Vh ; Dt = W1 Q1^T = W1 Q~(:, m-r:)
      if (rv) {
        nd.Vt0T = fact_->dbl((size_t)rv * (m - r));
        g3.push_back(hssk_gemm_desc{Vh[id], nd.Qt, nd.Vt0T, rv, m - r, m, m, m, rv, 1, 0, 1.0, 0.0});
        if (r) g3.push_back(hssk_gemm_desc{nd.Qt + (size_t)(m - r) * m, Vh[id], nd.Vt1, r, rv, m, m, m, r, 1, 0, 1.0, 0.0});
      }
      if (r) g3.push_back(hssk_gemm_desc{nd.W1, nd.Qt + (size_t)(m - r) * m, nd.Dt, r, r, m, r, m, ldt, 0, 0, 1.0, 0.0});
      // derived factors of the solve sweeps: WQ = W1 Q~(:, 0:m-r) and the inverted diagonal blocks of R~^T
      if (r) {
        nd.WQ = fact_->dbl((size_t)r * (m - r));
        g3.push_back(hssk_gemm_desc{nd.W1, nd.Qt, nd.WQ, r, m - r, m, r, m, r, 0, 0, 1.0, 0.0});
      }
      if (m <= 256) {
        nd.Tinv = fact_->dbl((size_t)((m - r + 63) / 64) * 4096);
        f.ti.push_back(hssk_trtri_desc{nd.Rlq, nd.Tinv, m - r, m, 0});
      }
      const double k = m - r;
      stats_.f_ulv += 2.0 * k * r * m + (2.0 * m * k * k - 2.0 / 3.0 * k * k * k) + (4.0 * m * m * k - 2.0 * m * k * k) / 1.0 * 0.5 + 2.0 * m * m * rv + 2.0 * r * (double)r * m;
    } else {
      // nothing to eliminate: Dt = P^T D, Vt1 = Vh   (factor.hpp:138-141)
      int ldt = 1;
      nd.Dt = dt_slot(id, m, ldt);
      nd.Vt1 = Vh[id];
      if (m) ge.push_back(hssk_elem_desc{Dh[id], m, nd.permU, nullptr, 0, 0, nd.Dt, m, m, ldt, 0});
    }
  }
  if (!un.empty()) ck(hssk_ulv_node_vbatched(cx, un.data(), (int)un.size()));
  if (!us.empty()) ck(hssk_ulv_split(cx, us.data(), (int)us.size()));
  if (!ge.empty()) ck(hssk_gather_elems(cx, ge.data(), (int)ge.size()));
  if (!g2.empty()) ck(hssk_gemm_vbatched(cx, g2.data(), (int)g2.size()));
  if (!qr.empty()) ck(hssk_qr_vbatched(cx, qr.data(), (int)qr.size()));
  if (!g3.empty()) ck(hssk_gemm_vbatched(cx, g3.data(), (int)g3.size()));
  if (!lu.empty()) ck(hssk_getrf_vbatched(cx, lu.data(), (int)lu.size()));
}

// ---- factor ahead (EngineOptions::factor_ahead): called by the compression after it has processed height h of the rank's own
// levels.  Heights are taken strictly in order, each as soon as every one of its nodes is compressed (a compressed node's
// bases are final: later rounds only touch the others); its launches go to the second context, whose stream waits for what
// the compression has enqueued so far.
void DeviceHSS::factor_ahead_level(size_t h) {
  if (!o_.factor_ahead || o_.algorithm != 1) return;
  if (o_.world > 1 && !dist_subtree_) return;
  if (!frun_.active) {
    if (h != 0) return;
    if (!fctx_) ck(hssk_ctx_create(&fctx_, o_.device));
    factor_begin(0, false, fctx_);
    frun_.ahead = true;
  }
  if (!frun_.ahead || h >= own_by_height_.size()) return;
  // every height up to h whose turn has come (a height held back earlier -- see below -- is caught up with here)
  while (frun_.done <= h) {
    const std::vector<int>& ids = own_by_height_[frun_.done];
    // A node's reduced block Dt goes straight into its PARENT's Dh (dt_slot), whose size and offsets are the ranks of BOTH
    // children: on a tree whose siblings differ in height the taller one is settled later, so the level also waits for the
    // siblings of its nodes (cut nodes keep a compact Dt of their own and need not).
    for (int id : ids) {
      const Node& nd = nodes_[id];
      if (!nd.compressed()) return;
      if (nd.parent >= 0 && !frun_.is_cut[id]) {
        const Node& pa = nodes_[nd.parent];
        if (!nodes_[pa.c0].compressed() || !nodes_[pa.c1].compressed()) return;
      }
    }
    ck(hssk_stream_wait(fctx_, ctx_));
    factor_prep(ids);
    factor_level(ids);
    // (the inverted diagonal blocks of the level's triangles: on this stream a launch per level costs nothing)
    if (!frun_.ti.empty()) { ck(hssk_trtri_diag_vbatched(fctx_, frun_.ti.data(), (int)frun_.ti.size())); frun_.ti.clear(); }
    frun_.done++;
  }
}

void DeviceHSS::factor_cancel() {
  if (frun_.active && frun_.ahead && fctx_) hssk_sync(fctx_);
  frun_ = FactorRun();
}

void DeviceHSS::factor_sub(int sr, bool partial) {
  OpGuard op_guard(op_mu_);
  ensure_ready("factor");
  double t0 = now();
  std::vector<std::vector<int>> sub_h;
  if (sr != 0) sub_h = sublists(own_by_height_, sr);
  const std::vector<std::vector<int>>& levels = sr ? sub_h : own_by_height_;
  const bool resume = sr == 0 && !partial && frun_.active && frun_.ahead;
  if (resume) {
    // the levels enqueued behind the compression are done or running on the other stream: wait, take the rest from here
    ck(hssk_sync(fctx_));
    ck(hssk_sync(ctx_));
    frun_.cx = ctx_;
  } else {
    static const bool trace_host = [] { const char* e = std::getenv("STRUMPACK_AMD_TRACE_HOST"); return e && e[0] == '1'; }();
    const double th0 = now();
    factor_cancel();
    ck(hssk_sync(ctx_));
    const double th1 = now();
    factor_begin(sr, partial, ctx_);
    const double th2 = now();
    if (trace_host) std::fprintf(stderr, "# host: factor: entry %.1f us, cancel + sync %.1f us, begin %.1f us\n", (th0 - t0) * 1e6, (th1 - th0) * 1e6, (th2 - th1) * 1e6);
    // the dense column bases depend on the compression only: one launch for the lowest level -- whose elimination, the bulk of
    // the factorization, is then enqueued before the host has touched the rest of the tree -- and one for everything above it
    // instead of one per level
    if (!levels.empty()) {
      const double tp0 = now();
      factor_prep(levels[0]);
      const double tp1 = now();
      factor_level(levels[0]);
      frun_.done = 1;
      if (trace_host) std::fprintf(stderr, "# host: factor: prep of the lowest level %.1f us, its launches %.1f us; since the end of compress() %.1f us\n", (tp1 - tp0) * 1e6, (now() - tp1) * 1e6, (tp0 - stats_.t_mark) * 1e6);
    }
    std::vector<int> all;
    for (size_t h = 1; h < levels.size(); h++) all.insert(all.end(), levels[h].begin(), levels[h].end());
    if (dist_subtree_) for (auto& ids : top_by_height_) all.insert(all.end(), ids.begin(), ids.end());
    factor_prep(all);
  }
  for (size_t h = frun_.done; h < levels.size(); h++) {
    if (resume) factor_prep(levels[h]);
    factor_level(levels[h]);
  }
  if (dist_subtree_) {
    exchange_cut_factor();
    for (auto& ids : top_by_height_) {
      if (resume) factor_prep(ids);
      factor_level(ids);
    }
  }
  if (!frun_.ti.empty()) ck(hssk_trtri_diag_vbatched(ctx_, frun_.ti.data(), (int)frun_.ti.size()));
  ck(hssk_sync(ctx_));
  frun_ = FactorRun();
  factored_ = sr == 0;
  sub_factored_ = (sr != 0 && !partial) ? sr : -1;
  partial_factored_ = partial;
  schur_ready_ = false;
  stats_.t_factor = now() - t0;
}

}  // namespace HSS
}  // namespace strumpack
