// DeviceHSS: the adaptive randomized compression (driver, tree levels, ID / TSQR, stopping test).
// HSSMatrix.compress_stable.hpp:100-442, HSSMatrix.compress.hpp:100-165,300-368,524-724.
#include "hss_engine_internal.hpp"
#include <cstdio>

namespace strumpack {
namespace HSS {

void DeviceHSS::drop_plans() {
  for (auto& kv : plans_) if (kv.second.plan) hssk_plan_destroy(kv.second.plan);
  plans_.clear();
  if (plan_arena_) plan_arena_->reset();
}
bool DeviceHSS::plans_enabled() const {
  static const bool off = [] { const char* e = std::getenv("STRUMPACK_AMD_NO_PLANS"); return e && e[0] == '1'; }();
  return !off;
}

void DeviceHSS::reset_compression() {
  book_ = PendingBook();   // (a compression that threw may have left an ID commit half done)
  defer_book_ = false;
  drop_plans();
  for (auto& nd : nodes_) {
    int lo = nd.lo, m = nd.m, lvl = nd.lvl, h = nd.height, c0 = nd.c0, c1 = nd.c1, p = nd.parent;
    nd = Node();
    nd.lo = lo; nd.m = m; nd.lvl = lvl; nd.height = h; nd.c0 = c0; nd.c1 = c1; nd.parent = p;
  }
  factor_cancel();   // (a factorization running ahead writes into the arenas reset below)
  persist_->reset();
  dev_tree_ = nullptr;
  work_->reset();
  fact_->reset();
  invalidate_factors();
  d_ranks_ = persist_->ints(2 * nodes_.size() + 2);
}

// hard restart (compress.hpp:289-294, reset()): every node back to UNTOUCHED, the first d_have sample rows of Srt_ / Sct_
// back to what the sampling produced; the sample arrays themselves are kept
void DeviceHSS::restart_nodes(int d_have) {
  ck(hssk_sync(ctx_));
  factor_cancel();
  for (auto& nd : nodes_) {
    int lo = nd.lo, m = nd.m, lvl = nd.lvl, h = nd.height, c0 = nd.c0, c1 = nd.c1, p = nd.parent;
    nd = Node();
    nd.lo = lo; nd.m = m; nd.lvl = lvl; nd.height = h; nd.c0 = c0; nd.c1 = c1; nd.parent = p;
  }
  persist_->reset();
  dev_tree_ = nullptr;
  invalidate_factors();
  d_ranks_ = persist_->ints(2 * nodes_.size() + 2);
  if (d_have > 0) {
    std::vector<hssk_colgather_desc> cp;
    const int ncols = (int)std::min<long long>(n_, 0x7fffffff);
    cp.push_back(hssk_colgather_desc{Srt0_, Srt_, nullptr, d_have, ncols, dcap_, dcap_, 0});
    cp.push_back(hssk_colgather_desc{Sct0_, Sct_, nullptr, d_have, ncols, dcap_, dcap_, 0});
    ck(hssk_gather_cols(ctx_, cp.data(), 2));
  }
}

void DeviceHSS::free_compress_workspace() {
  ck(hssk_sync(ctx_));
  work_->reset();
  Rt_ = Srt_ = Sct_ = Srt0_ = Sct0_ = nullptr;
  sj_pat_ = nullptr;
  for (auto& nd : nodes_) { nd.Srt = nd.Sct = nd.Rrt = nd.Rct = nd.RrtRed = nd.RctRed = nd.Qr = nd.Qc = nullptr; nd.panels = false; }
}

void DeviceHSS::compress(Source& src) {
  OpGuard op_guard(op_mu_);
  double t0 = now();
  stats_ = PhaseStats();
  int dcap = o_.algorithm != 1 ? o_.d0 + o_.p : o_.d0 + o_.dd;
  dcap = std::max(16, (dcap + 15) / 16 * 16);
  for (;;) {
    if (compress_attempt(src, dcap)) break;
    dcap *= 2;  // the sample capacity was too small: restart (the random stream is seeded, so the
                // restarted run retraces the same samples and continues past the old capacity)
    if (o_.verbose) std::cout << "# HSS compression: growing the sample capacity to " << dcap << std::endl;
  }
  static const bool trace_host = [] { const char* e = std::getenv("STRUMPACK_AMD_TRACE_HOST"); return e && e[0] == '1'; }();
  const double th0 = now();
  if (dist_subtree_) exchange_node_table();
  free_compress_workspace();
  comm_arena_->reset();
  const double th1 = now();
  {
    const double ms = hssk_watch_read_ms(ctx_, 5, nullptr);   // every product of every round and attempt
    if (ms > 0.) stats_.t_sketch = ms * 1e-3;
    double kms = 0., kfl = 0.;
    int kn = 0;
    ck(hssk_dgemm_timing_collect(ctx_, &kms, &kfl, &kn));     // the main launches of the products (hssk_dgemm_timing_defer)
    stats_.sketch_kernel_ms += kms; stats_.sketch_kernel_flops += kfl; stats_.sketch_launches += kn;
  }
  stats_.t_compress = now() - t0;
  stats_.t_tree = stats_.t_compress - stats_.t_sketch - stats_.t_random;
  stats_.t_mark = now();
  if (trace_host) std::fprintf(stderr, "# host: compress end: free workspace %.1f us, timers %.1f us\n", (th1 - th0) * 1e6, (now() - th1) * 1e6);
}

void DeviceHSS::fill_random(int r0, int dn) {
  if (o_.user_random) { sj_pat_ = nullptr; return; }   // the source's sample() delivers the random block with the products
  double t0 = now();
  const long long N = n_;
  sj_pat_ = nullptr;
  if (o_.sketch == 1) {
    // SJLT (HSSMatrix.compress_stable.hpp:39-97, HSSMatrix.sketch.hpp): every row of the N x dn block gets nnz entries
    // +-1 -- nnz0 in the first d0 + dd columns, nnz in each further block (S.add_columns / SJLTMatrix(g, nnz, n, dnew)).
    // CHUNK (sketch.hpp:419-441): one nonzero in each of nnz chunks of dn / nnz columns; PERM (:316-341): the first
    // nnz entries of a random permutation of the columns (drawn here as a partial Fisher-Yates shuffle).  Only the
    // pattern (nnz ints per row) crosses PCIe; the dense block the tree levels need is expanded on the device.
    if (r0 == 0 || !rng_) rng_.reset(new HostRng());
    auto& e = rng_->sj;
    const int nnz = std::max(1, std::min(r0 == 0 ? o_.nnz0 : o_.nnz, dn));
    // the device pattern holds up to 8 entries per row (hssk.h); denser sketching matrices (--hss_nnz0 / --hss_nnz > 8) keep the
    // same distribution but travel as the dense N x dn block and take the dense sketch products
    const bool dense_pat = nnz > 8;
    const int nq = dense_pat ? nnz : (nnz <= 4 ? 4 : 8);   // ints per row; unused ones point at column dn
    std::vector<int> pat((size_t)nq * N, dn);
    std::uniform_int_distribution<int> sign(0, 1);
    if (o_.sjlt_algo == 0) {
      const int chunk = dn / nnz;
      std::uniform_int_distribution<int> shift(0, chunk - 1);
      for (long long k = 0; k < N; k++)
        for (int q = 0; q < nnz; q++) {
          const int c = shift(e) + chunk * q;
          pat[(size_t)k * nq + q] = sign(e) == 0 ? c : (c | (int)0x80000000);
        }
    } else {
      std::vector<int> cols(dn);
      for (int j = 0; j < dn; j++) cols[j] = j;
      for (long long k = 0; k < N; k++)
        for (int q = 0; q < nnz; q++) {
          std::uniform_int_distribution<int> pick(q, dn - 1);
          std::swap(cols[q], cols[pick(e)]);
          pat[(size_t)k * nq + q] = sign(e) == 0 ? cols[q] : (cols[q] | (int)0x80000000);
        }
    }
    if (dense_pat) {
      std::vector<double> buf((size_t)dn * N, 0.);
      host_parallel_for((size_t)N, [&](size_t k) {
        for (int q = 0; q < nnz; q++) {
          const int p = pat[k * nq + q];
          buf[(size_t)(p & 0x7fffffff) + k * dn] = p < 0 ? -1. : 1.;
        }
      });
      ck(hssk_memcpy2d_h2d(ctx_, Rt_ + r0, sizeof(double) * dcap_, buf.data(), sizeof(double) * dn, sizeof(double) * dn, N));
      ck(hssk_sync(ctx_));
      stats_.t_random += now() - t0;
      return;
    }
    int* dp = work_->ints((size_t)nq * N);
    ck(hssk_memcpy_h2d(ctx_, dp, pat.data(), (long long)sizeof(int) * nq * N));
    ck(hssk_sjlt_dense(ctx_, Rt_ + r0, dn, N, dcap_, dp, nnz));
    sj_pat_ = dp;
    sj_nnz_ = nnz;
  } else if (o_.random_engine == 2) {
    // device Philox: element (sample s, column c) is a pure function of (seed, s * N + c)
    if (o_.random_dist != 0) throw std::invalid_argument("philox engine implements the normal distribution only");
    ck(hssk_randn(ctx_, Rt_ + r0, dn, N, dcap_, r0, N, 0x5354524dull));
  } else {
    // reference-identical host stream: DenseMatrix::random fills the N x dn block column-major,
    // i.e. sample by sample (dense/DenseMatrix.cpp:172-181); the generator persists across rounds
    // (HSSMatrix.compress_stable.hpp:108-112).
    if (r0 == 0 || !rng_) rng_.reset(new HostRng());
    if (o_.random_engine == 0 && o_.random_dist == 0 && N > 0 && dn > 0) {
      // the default (minstd_rand + normal): the same stream bit for bit, generated on all host threads (LinearNormal.hpp);
      // the N x dn block is uploaded as it is drawn and transposed into the sample rows on the device
      std::vector<double> blk((size_t)dn * N);
      rng_->linnorm.fill(blk.data(), blk.size(), [](std::size_t n, const std::function<void(std::size_t)>& fn) { host_parallel_for(n, fn); });
      double* dT = tmp_->dbl((size_t)dn * N);
      ck(hssk_memcpy_h2d(ctx_, dT, blk.data(), (long long)(sizeof(double) * blk.size())));
      hssk_transpose_desc t{dT, Rt_ + r0, (int)N, dn, (int)N, dcap_};
      ck(hssk_transpose(ctx_, &t, 1));
    } else {
      std::minstd_rand* lin = &rng_->lin;
      std::mt19937* mer = &rng_->mer;
      auto& nd = rng_->nd;
      auto& ud = rng_->ud;
      std::vector<double> buf((size_t)dn * N);
      for (int s = 0; s < dn; s++)
        for (long long c = 0; c < N; c++) {
          double v;
          if (o_.random_engine == 0) v = o_.random_dist == 0 ? nd(*lin) : ud(*lin);
          else v = o_.random_dist == 0 ? nd(*mer) : ud(*mer);
          buf[s + (size_t)c * dn] = v;
        }
      ck(hssk_memcpy2d_h2d(ctx_, Rt_ + r0, sizeof(double) * dcap_, buf.data(), sizeof(double) * dn, sizeof(double) * dn, N));
    }
  }
  ck(hssk_sync(ctx_));
  stats_.t_random += now() - t0;
}

bool DeviceHSS::compress_attempt(Source& src, int dcap) {
  reset_compression();
  attempt_++;
  dcap_ = dcap;
  const size_t N = n_;
  // sample arrays; with several GPUs the column count is padded to world * cols_per_rank so that
  // every rank's shard is one contiguous, equally sized block (in-place all-gather)
  cols_per_rank_ = o_.world > 1 ? ((long long)N + o_.world - 1) / o_.world : (long long)N;
  const size_t Npad = o_.world > 1 ? (size_t)cols_per_rank_ * o_.world : N;
  Rt_ = work_->dbl((size_t)dcap * N);
  Srt_ = work_->dbl((size_t)dcap * Npad);
  Sct_ = work_->dbl((size_t)dcap * Npad);
  if (o_.algorithm == 2) { Srt0_ = work_->dbl((size_t)dcap * Npad); Sct0_ = work_->dbl((size_t)dcap * Npad); }
  stats_.rounds = 0;
  stats_.f_sketch = stats_.f_local = stats_.f_reduce = stats_.f_id = stats_.f_ortho = 0;
  const bool original = (o_.algorithm != 1);
  if (!original) {
    // compress_stable(Amult, Aelem, opts), HSSMatrix.compress_stable.hpp:100-163
    int d = o_.d0, dd = o_.dd;
    while (!is_compressed()) {
      int c = (d == o_.d0) ? 0 : d;
      int dnew = (d == o_.d0) ? d + dd : dd;
      if (c + dnew > dcap) return false;
      fill_random(c, dnew);
      if (stats_.rounds == 0 && src.extract_before_sample()) {   // the leaves' diagonal blocks: independent of the samples
        std::vector<int> leaves;
        for (auto& ids : own_by_height_) for (int id : ids) if (nodes_[id].leaf() && nodes_[id].lvl != 0) leaves.push_back(id);
        extract_blocks(src, leaves);
      }
      // (the sketch phase is timed on the device clock -- stopwatch 5, read at the end of compress() --: a host clock needs a
      //  synchronisation behind the products, and the leaf level's first launch then starts a sync return and 512 descriptors
      //  later, 80 us at N = 1e5; now it is enqueued while the products run)
      ck(hssk_watch_start(ctx_, 5));
      src.sample(*this, c, dnew);
      ck(hssk_watch_stop(ctx_, 5));
      stats_.f_sketch += src.sketch_flops(*this, dnew);   // per product: 2 N^2 d (SJLT streamed: 2 nnz per element)
      if (o_.verbose) std::cout << "# compressing with d+dd = " << d << "+" << dd << " (stable)" << std::endl;
      stats_.rounds++;
      for (size_t h = 0; h < own_by_height_.size(); h++) {
        // above the leaves: every inner level of the round in ONE launch where that applies (hss_compress_tree.cpp); else,
        // and after a launch that found a rank above its bound, level by level
        if (h == 1 && tree_pass(src, d, dd)) {
          for (size_t hh = 1; hh < own_by_height_.size(); hh++) factor_ahead_level(hh);
          break;
        }
        process_level(src, own_by_height_[h], d, dd, false);
        factor_ahead_level(h);   // (EngineOptions::factor_ahead: the settled level's ULV factorization, on its own stream)
      }
      if (dist_subtree_) {
        exchange_cut_compress(d + dd);
        for (auto& ids : top_by_height_) process_level(src, ids, d, dd, false);
      }
      stats_.d_final = d + dd;
      if (!is_compressed()) {
        d += dd;
        dd = std::min(dd, o_.max_rank - d);
        if (dd <= 0) break;  // cannot add samples: compression failed (is_compressed() stays false)
      }
    }
  } else {
    // compress_original, HSSMatrix.compress.hpp:100-165
    int d_old = 0, d = o_.d0 + o_.p;
    while (!is_compressed()) {
      if (d > dcap) return false;
      fill_random(d_old, d - d_old);
      ck(hssk_watch_start(ctx_, 5));
      src.sample(*this, d_old, d - d_old);
      ck(hssk_watch_stop(ctx_, 5));
      stats_.f_sketch += src.sketch_flops(*this, d - d_old);
      if (o_.verbose) std::cout << "# compressing with d = " << d - o_.p << " + " << o_.p << (o_.algorithm == 2 ? " (original, hard restart)" : " (original)") << std::endl;
      if (o_.algorithm == 2) {   // keep the new samples as drawn
        if (dist_subtree_ || o_.world > 1) throw std::invalid_argument("hard restart is a single-GPU option");
        std::vector<hssk_colgather_desc> cp;
        cp.push_back(hssk_colgather_desc{Srt_ + d_old, Srt0_ + d_old, nullptr, d - d_old, (int)N, dcap_, dcap_, 0});
        cp.push_back(hssk_colgather_desc{Sct_ + d_old, Sct0_ + d_old, nullptr, d - d_old, (int)N, dcap_, dcap_, 0});
        ck(hssk_gather_cols(ctx_, cp.data(), 2));
      }
      stats_.rounds++;
      for (auto& ids : own_by_height_) process_level(src, ids, d, d - d_old, true);
      if (dist_subtree_) {
        exchange_cut_compress(d);
        for (auto& ids : top_by_height_) process_level(src, ids, d, d - d_old, true);
      }
      stats_.d_final = d;
      if (!is_compressed()) {
        d_old = d;
        d = 2 * (d_old - o_.p) + o_.p;
        if (d_old >= 4 * n_ + o_.p + 64) break;
        if (o_.algorithm == 2 && d <= dcap) restart_nodes(d_old);
      }
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// one tree height of one compression round
//   stable:   d, dd as in compress_recursive_stable (samples [0,d+dd), new ones [d,d+dd))
//   original: d = total samples, dd = newly added ones
// ---------------------------------------------------------------------------------------------
void DeviceHSS::process_level(Source& src, const std::vector<int>& ids_all, int d, int dd, bool original) {
  const int dtot = original ? d : d + dd;
  const int dnew0 = original ? d - dd : d;  // first new sample row
  std::vector<int> ids;
  for (int id : ids_all) {
    Node& nd = nodes_[id];
    if (!nd.leaf() && !(nodes_[nd.c0].compressed() && nodes_[nd.c1].compressed())) continue;
    if (nd.lvl == 0 && nd.compressed()) continue;
    ids.push_back(id);
  }
  if (ids.empty()) return;
  // --- extraction of D / B01 / B10 for untouched nodes (compress_stable.hpp:171-182, 204-217)
  std::vector<int> fresh;
  std::vector<char> was_untouched(nodes_.size(), 0), was_compressed(nodes_.size(), 0);
  for (int id : ids) {
    Node& nd = nodes_[id];
    was_untouched[id] = nd.untouched();
    was_compressed[id] = nd.compressed();
    if (nd.untouched()) fresh.push_back(id);
  }
  extract_blocks(src, fresh);
  std::vector<int> work_ids, r0s, dns;
  for (int id : ids) {
    Node& nd = nodes_[id];
    if (nd.lvl == 0) { nd.Ustate = nd.Vstate = 2; continue; }
    if (!nd.panels) {
      if (nd.leaf()) {
        nd.mU = nd.mV = nd.m;
        nd.Srt = Srt_ + (size_t)nd.lo * dcap_;
        nd.Sct = Sct_ + (size_t)nd.lo * dcap_;
        nd.Rrt = nd.Rct = Rt_ + (size_t)nd.lo * dcap_;
      } else {
        nd.mU = nodes_[nd.c0].rU + nodes_[nd.c1].rU;
        nd.mV = nodes_[nd.c0].rV + nodes_[nd.c1].rV;
        nd.Srt = work_->dbl((size_t)dcap_ * std::max(nd.mU, 1));
        nd.Sct = work_->dbl((size_t)dcap_ * std::max(nd.mV, 1));
        nd.Rrt = work_->dbl((size_t)dcap_ * std::max(nd.mV, 1));
        nd.Rct = work_->dbl((size_t)dcap_ * std::max(nd.mU, 1));
      }
      nd.panels = true;
    }
    work_ids.push_back(id);
    r0s.push_back(was_untouched[id] ? 0 : dnew0);
    dns.push_back(was_untouched[id] ? dtot : dtot - dnew0);
  }
  if (work_ids.empty()) return;
  local_samples(work_ids, r0s, dns);

  // --- bases
  std::vector<int> id_nodes, id_which, ot_nodes, ot_which;
  for (int id : work_ids) {
    Node& nd = nodes_[id];
    if (was_compressed[id]) continue;
    for (int w = 0; w < 2; w++) {
      int st = w == 0 ? nd.Ustate : nd.Vstate;
      if (st == 2) continue;
      int rows = w == 0 ? nd.mU : nd.mV;
      if (original || dtot >= o_.max_rank || dtot >= rows) { id_nodes.push_back(id); id_which.push_back(w); }
      else { ot_nodes.push_back(id); ot_which.push_back(w); }
    }
  }
  if (!ot_nodes.empty()) {
    std::vector<char> resolved;
    ortho_test(ot_nodes, ot_which, d, dd, resolved);
    for (size_t i = 0; i < ot_nodes.size(); i++) {
      if (resolved[i]) { id_nodes.push_back(ot_nodes[i]); id_which.push_back(ot_which[i]); }
      else {
        Node& nd = nodes_[ot_nodes[i]];
        (ot_which[i] == 0 ? nd.Ustate : nd.Vstate) = 1;
      }
    }
  }
  // (the host-side bookkeeping of the ID -- index sets, permutations -- is finished behind the launches of the sample
  // reduction below, which only need what is on the device; the ORIGINAL algorithm inspects and resets ranks first)
  defer_book_ = !original;
  run_id(id_nodes, id_which, dtot);
  defer_book_ = false;
  if (original) {
    // compute_U_V_bases acceptance, HSSMatrix.compress.hpp:663-686
    for (int id : work_ids) {
      Node& nd = nodes_[id];
      if (was_compressed[id]) continue;
      bool ok = (dtot - o_.p >= o_.max_rank) || (nd.rU < dtot - o_.p && nd.rV < dtot - o_.p);
      if (!ok) { nd.Ustate = nd.Vstate = 1; nd.rU = nd.rV = 0; nd.Ir.clear(); nd.Ic.clear(); }
    }
  }
  // --- reduce (reduce_local_samples, HSSMatrix.compress.hpp:689-724)
  std::vector<int> rd_ids, rd_r0, rd_dn;
  for (int id : work_ids) {
    Node& nd = nodes_[id];
    if (!nd.compressed()) continue;
    if (!was_compressed[id]) {
      nd.RrtRed = work_->dbl((size_t)dcap_ * std::max(nd.rV, 1));
      nd.RctRed = work_->dbl((size_t)dcap_ * std::max(nd.rU, 1));
      rd_ids.push_back(id); rd_r0.push_back(0); rd_dn.push_back(dtot);
    } else {
      rd_ids.push_back(id); rd_r0.push_back(dnew0); rd_dn.push_back(dtot - dnew0);
    }
  }
  reduce_samples(rd_ids, rd_r0, rd_dn);
  finish_id_bookkeeping();
}

void DeviceHSS::extract_blocks(Source& src, const std::vector<int>& ids) {
  std::vector<ElemReq> reqs;
  for (int id : ids) {
    Node& nd = nodes_[id];
    if (nd.leaf()) {
      if (nd.D) continue;   // (taken out ahead of the sketch: compress_attempt)
      nd.D = persist_->dbl((size_t)nd.m * nd.m);
      reqs.push_back(ElemReq{nullptr, nullptr, nullptr, nullptr, nd.lo, nd.lo, nd.m, nd.m, nd.D, nd.m});
    } else {
      Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
      nd.B01 = persist_->dbl((size_t)std::max(a.rU, 1) * std::max(b.rV, 1));
      nd.B10 = persist_->dbl((size_t)std::max(b.rU, 1) * std::max(a.rV, 1));
      reqs.push_back(ElemReq{a.dIr, b.dIc, &a.Ir, &b.Ic, 0, 0, a.rU, b.rV, nd.B01, std::max(a.rU, 1)});
      reqs.push_back(ElemReq{b.dIr, a.dIc, &b.Ir, &a.Ic, 0, 0, b.rU, a.rV, nd.B10, std::max(b.rU, 1)});
    }
  }
  if (!reqs.empty()) src.extract(*this, reqs);
}

// compute_local_samples (HSSMatrix.compress.hpp:524-629) on sample rows [r0, r0+dn) of each node
void DeviceHSS::local_samples(const std::vector<int>& ids, const std::vector<int>& r0s, const std::vector<int>& dns) {
  std::vector<hssk_combine_desc> cb;
  std::vector<hssk_gemm_desc> mm;
  std::vector<hssk_leaf_update_desc> lu;   // fused Sr / Sc update of the leaves (both share the R panel)
  for (size_t k = 0; k < ids.size(); k++) {
    Node& nd = nodes_[ids[k]];
    const int r0 = r0s[k], dn = dns[k];
    if (dn <= 0) continue;
    if (nd.leaf()) {
      const int m = nd.m;
      // Sr_loc -= D Rr_loc  ->  Srt -= Rt D^T ;  Sc_loc -= D^T Rc_loc  ->  Sct -= Rt D
      if (dn <= 192 && dn % 2 == 0 && r0 % 2 == 0 && nd.Rrt == nd.Rct)
        lu.push_back(hssk_leaf_update_desc{nd.Rrt + r0, nd.D, nd.Srt + r0, nd.Sct + r0, dn, m, dcap_, m, dcap_});
      else {
        mm.push_back(hssk_gemm_desc{nd.Rrt + r0, nd.D, nd.Srt + r0, dn, m, m, dcap_, m, dcap_, 0, 1, -1.0, 1.0});
        mm.push_back(hssk_gemm_desc{nd.Rct + r0, nd.D, nd.Sct + r0, dn, m, m, dcap_, m, dcap_, 0, 0, -1.0, 1.0});
      }
      stats_.f_local += 4.0 * m * (double)m * dn;
    } else {
      Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
      // the children's skeleton rows (extract_rows, compress.hpp:563-566, 611-614) minus the coupling terms, one fused
      // gather + product per block:  Sr0 = Sr_a(Jr_a) - B01 Rr1 ; Sr1 = Sr_b(Jr_b) - B10 Rr0 ;
      //                              Sc0 = Sc_a(Jc_a) - B10^T Rc1 ; Sc1 = Sc_b(Jc_b) - B01^T Rc0       (all transposed)
      const int none = 0x7fffffff;
      const int l01 = std::max(a.rU, 1), l10 = std::max(b.rU, 1);
      cb.push_back(hssk_combine_desc{a.Srt + r0, nullptr, dcap_, none, a.permU, b.RrtRed + r0, nullptr, dcap_, none, nullptr,
                                     nd.B01, 1, l01, -1.0, nd.Srt + r0, dcap_, dn, a.rU, b.rV});
      cb.push_back(hssk_combine_desc{b.Srt + r0, nullptr, dcap_, none, b.permU, a.RrtRed + r0, nullptr, dcap_, none, nullptr,
                                     nd.B10, 1, l10, -1.0, nd.Srt + r0 + (size_t)a.rU * dcap_, dcap_, dn, b.rU, a.rV});
      cb.push_back(hssk_combine_desc{a.Sct + r0, nullptr, dcap_, none, a.permV, b.RctRed + r0, nullptr, dcap_, none, nullptr,
                                     nd.B10, l10, 1, -1.0, nd.Sct + r0, dcap_, dn, a.rV, b.rU});
      cb.push_back(hssk_combine_desc{b.Sct + r0, nullptr, dcap_, none, b.permV, a.RctRed + r0, nullptr, dcap_, none, nullptr,
                                     nd.B01, l01, 1, -1.0, nd.Sct + r0 + (size_t)a.rV * dcap_, dcap_, dn, b.rV, a.rU});
      stats_.f_local += 4.0 * ((double)a.rU * b.rV + (double)b.rU * a.rV) * dn;
    }
  }
  if (!cb.empty()) ck(hssk_gather_combine(ctx_, cb.data(), (int)cb.size()));
  if (!lu.empty()) {
    int rc = hssk_leaf_update_vbatched(ctx_, lu.data(), (int)lu.size());
    if (rc == 2) {  // layout not eligible for the fused kernel: two plain GEMMs per leaf
      for (auto& u : lu) {
        mm.push_back(hssk_gemm_desc{u.R, u.D, u.Sr, u.d, u.m, u.m, u.ldr, u.ldd, u.lds, 0, 1, -1.0, 1.0});
        mm.push_back(hssk_gemm_desc{u.R, u.D, u.Sc, u.d, u.m, u.m, u.ldr, u.ldd, u.lds, 0, 0, -1.0, 1.0});
      }
    } else ck(rc);
  }
  if (!mm.empty()) ck(hssk_gemm_vbatched(ctx_, mm.data(), (int)mm.size()));
}

// reduce_local_samples: Rr_loc <- V^H Rr_loc, Rc_loc <- U^H Rc_loc (HSSBasisID::applyC), transposed
void DeviceHSS::reduce_samples(const std::vector<int>& ids, const std::vector<int>& r0s, const std::vector<int>& dns) {
  if (ids.empty()) return;
  // Rr_red = Rr(Jc, :) + XV Rr(rest, :) (transposed: columns of Rrt), where Rr of an inner node is the stack of its
  // children's reduced samples -- read in place from the two children ([a | b] with the split at a's rank), one fused
  // gather + product per (node, side)
  std::vector<hssk_combine_desc> cb;
  const int none = 0x7fffffff;
  for (size_t k = 0; k < ids.size(); k++) {
    Node& nd = nodes_[ids[k]];
    const int r0 = r0s[k], dn = dns[k];
    if (dn <= 0) continue;
    const double *rr0, *rr1 = nullptr, *rc0, *rc1 = nullptr;
    int sr = none, sc = none;
    if (nd.leaf()) { rr0 = nd.Rrt + r0; rc0 = nd.Rct + r0; }
    else {
      Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
      rr0 = a.RrtRed + r0; rr1 = b.RrtRed + r0; sr = a.rV;
      rc0 = a.RctRed + r0; rc1 = b.RctRed + r0; sc = a.rU;
    }
    {
      const int m = nd.mV, r = nd.rV, K = (m > r && r > 0) ? m - r : 0;
      if (r > 0)
        cb.push_back(hssk_combine_desc{rr0, rr1, dcap_, sr, nd.permV, rr0, rr1, dcap_, sr, nd.permV + r, nd.XV, 1, r, 1.0,
                                       nd.RrtRed + r0, dcap_, dn, r, K});
      stats_.f_reduce += 2.0 * r * (double)K * dn;
    }
    {
      const int m = nd.mU, r = nd.rU, K = (m > r && r > 0) ? m - r : 0;
      if (r > 0)
        cb.push_back(hssk_combine_desc{rc0, rc1, dcap_, sc, nd.permU, rc0, rc1, dcap_, sc, nd.permU + r, nd.XU, 1, r, 1.0,
                                       nd.RctRed + r0, dcap_, dn, r, K});
      stats_.f_reduce += 2.0 * r * (double)K * dn;
    }
  }
  if (!cb.empty()) ck(hssk_gather_combine(ctx_, cb.data(), (int)cb.size()));
}

// ID of the listed (node, basis) pairs on all dtot samples; commits ranks, X, perm, index sets
void DeviceHSS::run_id(const std::vector<int>& ids, const std::vector<int>& which, int dtot) {
  if (ids.empty()) return;
  Arena& tmp = *tmp_;
  tmp.rewind();
  const size_t cnt = ids.size();
  std::vector<double*> Ws(cnt, nullptr);
  std::vector<const double*> srcs(cnt, nullptr);
  std::vector<int> ds(cnt, dtot);
  for (size_t k = 0; k < cnt; k++) {
    Node& nd = nodes_[ids[k]];
    const int m = which[k] == 0 ? nd.mU : nd.mV;
    const double* S = which[k] == 0 ? nd.Srt : nd.Sct;
    if (m == 0) continue;
    Ws[k] = tmp.dbl((size_t)dtot * m);
    srcs[k] = S;
  }
  // (the samples stay where they are: the ID reads them in place and writes its factors to the panel in tmp_)
  id_panels(ids, which, Ws, ds, &srcs, dcap_);
}

// Tall panels (d >> m, the kernel-matrix path: d = thousands of sampled columns): the pivoted QR of W (d x m) only
// depends on R of W = Q R, so W is first reduced to its m x m triangular factor by an unpivoted Householder TSQR
// -- row chunks factored independently (register-resident / blocked batched QR), the R factors stacked pairwise
// and re-factored until one is left -- and the ID then runs on that small panel in the register kernels.  Same
// pivots, ranks and X = R11^{-1} R12 as the direct QRCP up to rounding; backward stable (all Householder).
static bool tsqr_staircase() {   // (read per call: the tests compare both paths in one process)
  const char* e = std::getenv("STRUMPACK_AMD_TSQR_DENSE");
  return !(e && std::atoi(e));
}

void DeviceHSS::tsqr_reduce(const std::vector<int>& ids, const std::vector<int>& which, std::vector<double*>& Ws,
                            std::vector<int>& ds) {
  Arena& tmp = *tmp_;
  const size_t cnt = ids.size();
  struct Piece { double* p; int ld, rows; };
  std::vector<std::vector<Piece>> pieces(cnt);
  std::vector<int> ms(cnt, 0);
  bool any = false;
  std::vector<hssk_qr_desc> qr;
  for (size_t k = 0; k < cnt; k++) {
    const Node& nd = nodes_[ids[k]];
    const int m = which[k] == 0 ? nd.mU : nd.mV, d = ds[k];
    ms[k] = m;
    if (m <= 0 || !Ws[k] || d <= std::max(256, 2 * m)) continue;
    // chunk rows: register QR (<= 256 rows x 192 columns, <= 208 rows x 208 columns: the 16-lanes-per-column kernels of
    // hssk_qr.hip), the 512-row blocked path, or the tall blocked path for wide panels
    const int chunk = m <= 192 ? 256 : (m <= 208 ? 208 : (m <= 256 ? 512 : 2 * m));
    for (int r0 = 0; r0 < d; r0 += chunk) {
      const int cr = std::min(chunk, d - r0);
      double* wk = tmp.dbl((size_t)cr + m);
      qr.push_back(hssk_qr_desc{Ws[k] + r0, d, cr, m, nullptr, 0, 0, nullptr, wk, 0, 0., 0., 1});   // (only R is read again)
      pieces[k].push_back(Piece{Ws[k] + r0, d, std::min(cr, m)});
      {   // Householder QR of a cr x m chunk, R only: 2 cr m^2 - (2/3) m^3 (cr >= m), 2 m cr^2 - (2/3) cr^3 otherwise
        const double a_ = std::max(cr, m), b_ = std::min(cr, m);
        stats_.f_ortho += 2.0 * a_ * b_ * b_ - 2.0 / 3.0 * b_ * b_ * b_;
      }
    }
    any = true;
  }
  if (!any) return;
  ck(hssk_qr_vbatched(ctx_, qr.data(), (int)qr.size()));
  // pairs of full triangles are merged by hssk_tpqr_vbatched: in place over the first one, no stacking, one launch per tree
  // level (STRUMPACK_AMD_TSQR_PAIRS=0 or STRUMPACK_AMD_TSQR_DENSE=1: the stacked blocked QR below for everything)
  static const bool pairs_off = [] { const char* e = std::getenv("STRUMPACK_AMD_TSQR_PAIRS"); return e && e[0] == '0'; }();
  for (;;) {
    std::vector<hssk_triu_desc> cp;
    std::vector<hssk_tpqr_desc> tp;
    qr.clear();
    bool more = false;
    for (size_t k = 0; k < cnt; k++) {
      std::vector<Piece>& pc = pieces[k];
      if (pc.size() <= 1) continue;
      const int m = ms[k];
      // fan-in: as many triangles as fit the register QR (256 rows) for narrow panels; wider ones either pairwise
      // (390-row blocked QR per tree level) or all at once through the tall blocked path (STRUMPACK_AMD_TSQR_FANIN)
      static const int fan_env = std::getenv("STRUMPACK_AMD_TSQR_FANIN") ? std::atoi(std::getenv("STRUMPACK_AMD_TSQR_FANIN")) : 0;
      const size_t fan_wide = fan_env >= 2 ? (size_t)fan_env : 2;
      const size_t fan = m <= 128 ? std::max<size_t>(2, 256 / std::max(m, 1)) : fan_wide;
      std::vector<Piece> next;
      for (size_t i = 0; i < pc.size(); i += fan) {
        const size_t cntp = std::min(fan, pc.size() - i);
        if (cntp == 1) { next.push_back(pc[i]); continue; }
        int rows = 0;
        bool full = true;   // every piece a full m x m triangle
        for (size_t t = 0; t < cntp; t++) { rows += pc[i + t].rows; full = full && pc[i + t].rows == m; }
        if (cntp == 2 && full && m <= 224 && !pairs_off && tsqr_staircase()) {
          tp.push_back(hssk_tpqr_desc{pc[i].p, pc[i].ld, pc[i + 1].p, pc[i + 1].ld, m});
          next.push_back(pc[i]);
          stats_.f_ortho += 2.0 / 3.0 * (double)m * m * m;   // structured QR of two stacked triangles
          continue;
        }
        double* dst = tmp.dbl((size_t)rows * m);
        // full triangles are stacked with their rows interleaved (row r of piece t -> row cntp r + t): column j of the
        // stack is then zero from row cntp (j + 1) on, and the blocked QR only sweeps that staircase (hssk_qr_desc::stair)
        const bool stair = full && rows > 256 && tsqr_staircase();
        int r0 = 0;
        for (size_t t = 0; t < cntp; t++) {
          if (stair) cp.push_back(hssk_triu_desc{pc[i + t].p, dst + t, pc[i + t].rows, m, pc[i + t].ld, rows, (int)cntp});
          else cp.push_back(hssk_triu_desc{pc[i + t].p, dst + r0, pc[i + t].rows, m, pc[i + t].ld, rows, 1});
          r0 += pc[i + t].rows;
        }
        double* wk = tmp.dbl((size_t)rows + m);
        qr.push_back(hssk_qr_desc{dst, rows, rows, m, nullptr, 0, 0, nullptr, wk, stair ? (int)cntp : 0, 0., 0., 1});
        next.push_back(Piece{dst, rows, std::min(rows, m)});
        {   // (counted as the dense QR of the stack; the staircase sweep does about a third of it)
          const double a_ = std::max(rows, m), b_ = std::min(rows, m);
          stats_.f_ortho += (2.0 * a_ * b_ * b_ - 2.0 / 3.0 * b_ * b_ * b_) * (stair ? 1.0 / 3.0 : 1.0);
        }
      }
      pc.swap(next);
      more = more || pc.size() > 1;
    }
    if (cp.empty() && tp.empty()) break;
    if (!tp.empty()) ck(hssk_tpqr_vbatched(ctx_, tp.data(), (int)tp.size()));
    if (!cp.empty()) {
      ck(hssk_copy_triu(ctx_, cp.data(), (int)cp.size()));
      ck(hssk_qr_vbatched(ctx_, qr.data(), (int)qr.size()));
    }
    if (!more) break;
  }
  // clean m x m (or shorter) triangular panels for the ID
  std::vector<hssk_triu_desc> fin;
  for (size_t k = 0; k < cnt; k++) {
    if (pieces[k].empty()) continue;
    const Piece& pc = pieces[k][0];
    double* R = tmp.dbl((size_t)pc.rows * ms[k]);
    fin.push_back(hssk_triu_desc{pc.p, R, pc.rows, ms[k], pc.ld, pc.rows});
    Ws[k] = R;
    ds[k] = pc.rows;
  }
  if (!fin.empty()) ck(hssk_copy_triu(ctx_, fin.data(), (int)fin.size()));
}

// Row ID of the listed (node, basis) pairs from prepared panels W_k = S_k^T (ds[k] x m_k, contiguous, in tmp_):
// truncated QRCP + X = R11^{-1} R12 on the device, then the commit of rank, permutation, skeleton indices.
void DeviceHSS::id_panels(const std::vector<int>& ids, const std::vector<int>& which, const std::vector<double*>& Ws_in,
                          const std::vector<int>& ds_in, const std::vector<const double*>* srcs, int ldsrc) {
  Arena& tmp = *tmp_;
  const size_t cnt = ids.size();
  std::vector<double*> Ws(Ws_in);
  std::vector<int> ds(ds_in);
  if (srcs) {
    // panels that take the TSQR pre-reduction (more than 256 sample rows) are reduced in place: those need their copy
    std::vector<hssk_colgather_desc> cp;
    bool tall = false;
    for (size_t k = 0; k < cnt; k++) tall = tall || ds[k] > 256;
    if (tall) {
      for (size_t k = 0; k < cnt; k++) {
        const int m = which[k] == 0 ? nodes_[ids[k]].mU : nodes_[ids[k]].mV;
        if (Ws[k] && m) cp.push_back(hssk_colgather_desc{(*srcs)[k], Ws[k], nullptr, ds[k], m, ldsrc, ds[k], 0});
      }
      if (!cp.empty()) ck(hssk_gather_cols(ctx_, cp.data(), (int)cp.size()));
      srcs = nullptr;
    }
  }
  // ---- tall panels at loose tolerances: the ID from the Gram matrix W^T W (kernels/hssk_pchol.hip) -- the d-long part of the
  // work as a product on the matrix cores, then `rank` steps of a pivoted Cholesky factorization, instead of the TSQR's
  // Householder sweeps (36 of the 48 ms of the blocks + ID phase of a 1e5-point kernel matrix).  STRUMPACK_AMD_ID_GRAM=0: off.
  static const bool gram_on = [] { const char* e = std::getenv("STRUMPACK_AMD_ID_GRAM"); return !(e && e[0] == '0'); }();
  std::vector<char> gram(cnt, 0);
  std::vector<double*> gramR(cnt, nullptr);
  std::vector<int> gramLd(cnt, 0);
  std::vector<hssk_pchol_desc> pcd;
  std::vector<size_t> pck;
  const std::vector<hssk_keval_desc>* gen = id_gen_;   // panels not evaluated yet (kernel matrices)
  std::vector<char> evaluated(cnt, gen ? 0 : 1);
  auto evaluate = [&](const std::vector<size_t>& ks_) {
    std::vector<hssk_keval_desc> ev;
    for (size_t k : ks_)
      if (!evaluated[k] && (*gen)[k].out) { ev.push_back((*gen)[k]); evaluated[k] = 1; }
    if (!ev.empty()) ck(hssk_kernel_eval_vbatched(ctx_, &id_gen_spec_, ev.data(), (int)ev.size()));
  };
  if (gram_on && id_gram_ && !srcs) {
    std::vector<hssk_gram_desc> gd;
    std::vector<hssk_gramgen_desc> gg;
    std::vector<hssk_sum_desc> sd;
    size_t tiles_total = 0;
    for (size_t k = 0; k < cnt; k++) {
      const Node& nd = nodes_[ids[k]];
      const int m = which[k] == 0 ? nd.mU : nd.mV, d = ds[k];
      if (m <= 0 || m > hssk_pchol_id_max_m() || !Ws[k] || d <= std::max(256, 2 * m) || o_.rel_tol / nd.lvl < 1e-6) continue;
      gram[k] = 1;
      tiles_total++;   // (a workgroup per panel and row chunk: hssk_gram_vbatched)
    }
    for (size_t k = 0; k < cnt; k++) {
      if (!gram[k]) continue;
      const Node& nd = nodes_[ids[k]];
      const int m = which[k] == 0 ? nd.mU : nd.mV, d = ds[k];
      // K chunks: enough workgroups over the level to fill the chip (STRUMPACK_AMD_GRAM_WGS, default 768: more and shorter chunks measured slower), chunks of >= 256 rows
      static const int wgs = [] { const char* e = std::getenv("STRUMPACK_AMD_GRAM_WGS"); return e ? std::max(1, std::atoi(e)) : 768; }();
      const int want = (int)std::max<size_t>(1, ((size_t)wgs + tiles_total - 1) / tiles_total);
      const int chunks = std::max(1, std::min(want, (d + 255) / 256)), rows = ((d + chunks - 1) / chunks + 15) & ~15;
      int nch = (d + rows - 1) / rows;
      if (nch > 1 && d - (nch - 1) * rows < 16) nch--;   // (a last chunk of a few rows joins its predecessor: a chunk has at least two)
      double* G = tmp.dbl((size_t)m * m);
      double* P = nch > 1 ? tmp.dbl((size_t)nch * m * m) : G;
      const bool fused = gen && (*gen)[k].out == Ws[k] && hssk_gram_gen_supported(&id_gen_spec_, m);
      for (int c = 0; c < nch; c++) {
        const int r0 = c * rows, kr = c == nch - 1 ? d - r0 : rows;
        if (fused) {
          const hssk_keval_desc& kd = (*gen)[k];
          gg.push_back(hssk_gramgen_desc{kd.ri ? kd.ri + r0 : nullptr, kd.r0 + r0, kd.ci, kd.c0, kr, m, P + (size_t)c * m * m, m});
        } else gd.push_back(hssk_gram_desc{Ws[k] + r0, d, kr, m, P + (size_t)c * m * m, m});
      }
      if (!fused && gen) evaluate({k});
      if (nch > 1) sd.push_back(hssk_sum_desc{P, (long long)m * m, (long long)m * m, nch, G});
      const int cap = std::max(1, std::min(m, hssk_pchol_id_rank_cap(m)));
      gramR[k] = tmp.dbl((size_t)cap * m);
      gramLd[k] = cap;
      pcd.push_back(hssk_pchol_desc{G, m, m, o_.rel_tol / nd.lvl, o_.abs_tol / nd.lvl, o_.max_rank, nullptr, nullptr, gramR[k], cap});
      pck.push_back(k);
      stats_.f_ortho += (double)d * m * (m + 1.0);   // the triangle of W^T W
    }
    if (!gd.empty()) ck(hssk_gram_vbatched(ctx_, gd.data(), (int)gd.size()));
    if (!gg.empty()) ck(hssk_gram_gen_vbatched(ctx_, &id_gen_spec_, gg.data(), (int)gg.size()));
    if (!sd.empty()) ck(hssk_sum_partials(ctx_, sd.data(), (int)sd.size()));
  }
  {
    // (the TSQR takes what the Gram form does not)
    std::vector<double*> Wt(Ws);
    for (size_t k = 0; k < cnt; k++) if (gram[k]) Wt[k] = nullptr;
    if (gen) {   // (what the Gram form does not take is evaluated now)
      std::vector<size_t> rest;
      for (size_t k = 0; k < cnt; k++) if (!gram[k]) rest.push_back(k);
      evaluate(rest);
    }
    tsqr_reduce(ids, which, Wt, ds);
    for (size_t k = 0; k < cnt; k++) if (!gram[k]) Ws[k] = Wt[k];
  }
  std::vector<hssk_id_desc> idd;
  int id_dmax = 0, id_mmax = 0;
  std::vector<int*> perms(cnt, nullptr);
  size_t perm_total = 0;
  for (size_t k = 0; k < cnt; k++) perm_total += (which[k] == 0 ? nodes_[ids[k]].mU : nodes_[ids[k]].mV);
  // ranks and permutations of the level in ONE device block: one read-back (one host synchronisation) per level
  int* rank_block = persist_->ints(cnt + std::max<size_t>(perm_total, 1));
  int* perm_block = rank_block + cnt;
  size_t poff = 0;
  for (size_t k = 0; k < cnt; k++) {
    Node& nd = nodes_[ids[k]];
    const int m = which[k] == 0 ? nd.mU : nd.mV;
    perms[k] = perm_block + poff;
    poff += m;
    if (m == 0) continue;
    if (gram[k]) continue;   // (its descriptor is completed below)
    double* wk = tmp.dbl(3 * (size_t)m);
    // (defer_x: X = R11^{-1} R12 is computed behind the read-back of the ranks, straight into its final place)
    idd.push_back(hssk_id_desc{Ws[k], ds[k], ds[k], m, o_.rel_tol / nd.lvl, o_.abs_tol / nd.lvl, o_.max_rank, perms[k], rank_block + k, wk,
                               srcs ? (*srcs)[k] : nullptr, ldsrc, 1});
    id_dmax = std::max(id_dmax, ds[k]);
    id_mmax = std::max(id_mmax, m);
  }
  if (!idd.empty()) ck(hssk_id_vbatched(ctx_, idd.data(), (int)idd.size()));
  for (size_t q = 0; q < pcd.size(); q++) { pcd[q].perm = perms[pck[q]]; pcd[q].rank = rank_block + pck[q]; }
  if (!pcd.empty()) ck(hssk_pchol_id_vbatched(ctx_, pcd.data(), (int)pcd.size()));
  const int x_solved = hssk_id_solves_inline(id_dmax, id_mmax);
  std::vector<int> hall(cnt + std::max<size_t>(perm_total, 1));
  ck(hssk_memcpy_d2h(ctx_, hall.data(), rank_block, (long long)sizeof(int) * (cnt + perm_total)));
  {
    // Gram panels whose rank outgrew the kernel's rows (-1): the Householder path for those, and one more read-back
    std::vector<int> rid, rwh, rds;
    std::vector<double*> rW;
    std::vector<size_t> rk_;
    for (size_t k = 0; k < cnt; k++)
      if (gram[k] && hall[k] < 0) { rid.push_back(ids[k]); rwh.push_back(which[k]); rW.push_back(Ws[k]); rds.push_back(ds[k]); rk_.push_back(k); }
    if (!rid.empty()) {
      if (gen) evaluate(rk_);
      tsqr_reduce(rid, rwh, rW, rds);
      std::vector<hssk_id_desc> rdd;
      int dmx = 0, mmx = 0;
      for (size_t q = 0; q < rid.size(); q++) {
        const size_t k = rk_[q];
        const Node& nd = nodes_[ids[k]];
        const int m = which[k] == 0 ? nd.mU : nd.mV;
        double* wk = tmp.dbl(3 * (size_t)m);
        rdd.push_back(hssk_id_desc{rW[q], rds[q], rds[q], m, o_.rel_tol / nd.lvl, o_.abs_tol / nd.lvl, o_.max_rank, perms[k], rank_block + k, wk, nullptr, 0, 0});
        dmx = std::max(dmx, rds[q]); mmx = std::max(mmx, m);
        gram[k] = 2;   // (X solved by the ID kernel itself: defer_x = 0)
        Ws[k] = rW[q]; ds[k] = rds[q];
      }
      ck(hssk_id_vbatched(ctx_, rdd.data(), (int)rdd.size()));
      ck(hssk_memcpy_d2h(ctx_, hall.data(), rank_block, (long long)sizeof(int) * (cnt + perm_total)));
    }
  }
  // (the cooperative ID's workgroups poll each other with a bounded spin; a timeout must not pass as a rank)
  if (hssk_sweep_status(ctx_)) throw std::runtime_error(std::string("compress: interpolative decomposition: ") + hssk_last_error());
  const int* hranks = hall.data();
  const int* hperm = hall.data() + cnt;
  // commit, in the order that puts the device back to work first: (A) ranks -> final places of X -> the X solves are
  // launched; (B) the host-side bookkeeping (permutations, global skeleton indices: vectors per node) while they run;
  // then the one index upload of the level
  std::vector<hssk_xsolve_desc> xc;
  std::vector<size_t> idx_off(cnt), perm_off(cnt);
  size_t idx_total = 0;
  poff = 0;
  for (size_t k = 0; k < cnt; k++) {
    Node& nd = nodes_[ids[k]];
    const int w = which[k];
    const int m = w == 0 ? nd.mU : nd.mV;
    const int dtot = gram[k] == 1 ? gramLd[k] : ds[k];
    const int r = m ? hranks[k] : 0;
    perm_off[k] = poff;
    poff += m;
    idx_off[k] = idx_total;
    idx_total += r;
    double* X = persist_->dbl((size_t)std::max(r, 1) * std::max(m - r, 1));
    if (r > 0 && m > r) xc.push_back(hssk_xsolve_desc{gram[k] == 1 ? gramR[k] : Ws[k], dtot, r, m, X, r, gram[k] == 1 ? 0 : (gram[k] == 2 ? 1 : x_solved)});
    if (w == 0) { nd.rU = r; nd.XU = X; nd.permU = perms[k]; nd.Ustate = 2; }
    else { nd.rV = r; nd.XV = X; nd.permV = perms[k]; nd.Vstate = 2; }
    stats_.f_id += 2.0 * (4.0 * m * (double)dtot * r - 2.0 * (m + dtot) * (double)r * r + 4.0 * r * (double)r * r / 3.0 + (double)r * r * (m - r));
  }
  if (!xc.empty()) ck(hssk_id_xsolve_vbatched(ctx_, xc.data(), (int)xc.size()));
  book_.ids = ids; book_.which = which; book_.hall = std::move(hall);
  book_.idx_off = std::move(idx_off); book_.perm_off = std::move(perm_off);
  book_.cnt = cnt; book_.idx_total = idx_total; book_.active = true;
  if (!defer_book_) finish_id_bookkeeping();
  // (no synchronisation: everything that reuses the W panels in tmp_ is enqueued behind these launches on the same stream)
}

// Second half of id_panels' commit: permutations and global skeleton indices of the level on the host (vectors per node),
// the one index upload of the level.  Nothing on the device waits for it except the next level's block extraction.
void DeviceHSS::finish_id_bookkeeping() {
  if (!book_.active) return;
  book_.active = false;
  const std::vector<int>&ids = book_.ids, &which = book_.which;
  const size_t cnt = book_.cnt, idx_total = book_.idx_total;
  const int* hperm = book_.hall.data() + cnt;
  std::vector<int> idx_host(std::max<size_t>(idx_total, 1));   // all skeleton index sets of this level: one upload
  // (on this thread: waking the host pool costs ~100 us, as much as the widest level's bookkeeping itself -- measured)
  for (size_t k = 0; k < cnt; k++) {
    Node& nd = nodes_[ids[k]];
    const int w = which[k];
    const int m = w == 0 ? nd.mU : nd.mV;
    const int r = w == 0 ? nd.rU : nd.rV;
    std::vector<int> perm(hperm + book_.perm_off[k], hperm + book_.perm_off[k] + m);
    // global skeleton indices (compress_stable.hpp:299-306, 334-341)
    std::vector<int> I(r);
    if (nd.leaf()) for (int i = 0; i < r; i++) I[i] = nd.lo + perm[i];
    else {
      const std::vector<int>& ia = w == 0 ? nodes_[nd.c0].Ir : nodes_[nd.c0].Ic;
      const std::vector<int>& ib = w == 0 ? nodes_[nd.c1].Ir : nodes_[nd.c1].Ic;
      const int r0 = (int)ia.size();
      for (int i = 0; i < r; i++) I[i] = perm[i] < r0 ? ia[perm[i]] : ib[perm[i] - r0];
    }
    std::copy(I.begin(), I.end(), idx_host.begin() + book_.idx_off[k]);
    if (w == 0) { nd.hpermU = std::move(perm); nd.Ir = std::move(I); }
    else { nd.hpermV = std::move(perm); nd.Ic = std::move(I); }
  }
  int* idx_dev = persist_->ints(std::max<size_t>(idx_total, 1));
  if (idx_total) ck(hssk_upload_async(ctx_, idx_dev, idx_host.data(), (long long)sizeof(int) * idx_total));
  for (size_t k = 0; k < cnt; k++) {
    Node& nd = nodes_[ids[k]];
    (which[k] == 0 ? nd.dIr : nd.dIc) = idx_dev + book_.idx_off[k];
  }
}

// update_orthogonal_basis (HSSMatrix.compress_stable.hpp:390-442) for the listed (node, basis) pairs
void DeviceHSS::ortho_test(const std::vector<int>& ids, const std::vector<int>& which, int d, int dd,
                           std::vector<char>& resolved) {
  const size_t cnt = ids.size();
  resolved.assign(cnt, 0);
  Arena& tmp = *tmp_;
  tmp.rewind();
  std::vector<hssk_transpose_desc> tr;
  std::vector<hssk_colgather_desc> cp;
  std::vector<hssk_qr_desc> qr;
  double* rdiag = tmp.dbl(2 * cnt);
  std::vector<char> untouched(cnt);
  // The QR of the first d sample columns only has to deliver max / min |R_ii| (DenseMatrix::orthogonalize,
  // dense/DenseMatrix.cpp:721-744); its explicit Q is needed by the Gram-Schmidt step alone, i.e. for the nodes
  // the R-diagonal test leaves undecided -- it is formed for those (hssk_formq_vbatched) after the read-back.
  for (size_t k = 0; k < cnt; k++) {
    Node& nd = nodes_[ids[k]];
    const int w = which[k];
    const int m = w == 0 ? nd.mU : nd.mV;
    const double* S = w == 0 ? nd.Srt : nd.Sct;
    double* Q = w == 0 ? nd.Qr : nd.Qc;
    untouched[k] = (w == 0 ? nd.Ustate : nd.Vstate) == 0;
    int c2, n2;
    if (untouched[k]) { c2 = 0; n2 = std::min(d, m); }
    else { c2 = d - dd; n2 = std::min(dd, m - (d - dd)); }
    double* T = tmp.dbl((size_t)m * std::max(n2, 1));
    if (untouched[k]) tr.push_back(hssk_transpose_desc{S, T, n2, m, dcap_, m});
    else cp.push_back(hssk_colgather_desc{Q + (size_t)c2 * m, T, nullptr, m, n2, m, m, 0});
    double* wk = tmp.dbl((size_t)m + n2);
    // the R-diagonal test below only needs to know whether SOME |R_ii| falls under the tolerance: the factorisation may stop
    // at the first one that does (hssk_qr_desc.stop_rel; the 1 - 1e-12 keeps the device's product form on the safe side of the
    // host's quotient form).  A node the test leaves undecided has run the full factorisation, which formq then uses.
    {
      const double atol = o_.abs_tol / nd.lvl, rtol = o_.rel_tol / nd.lvl;
      const double r0 = w == 0 ? nd.Ur_max : nd.Vr_max;
      hssk_qr_desc q{T, m, m, n2, nullptr, m, 0, rdiag + 2 * k, wk, 0, 0., 0.};
      if (untouched[k]) { q.stop_rel = rtol * (1. - 1e-12); q.stop_abs = atol; }
      else q.stop_abs = std::max(atol, rtol * std::abs(r0) * (1. - 1e-12));
      qr.push_back(q);
    }
    stats_.f_ortho += 4.0 * m * (double)n2 * n2;
  }
  if (!cp.empty()) ck(hssk_gather_cols(ctx_, cp.data(), (int)cp.size()));
  if (!tr.empty()) ck(hssk_transpose(ctx_, tr.data(), (int)tr.size()));
  ck(hssk_qr_vbatched(ctx_, qr.data(), (int)qr.size()));
  std::vector<double> hr(2 * cnt);
  ck(hssk_memcpy_d2h(ctx_, hr.data(), rdiag, (long long)sizeof(double) * 2 * cnt));
  std::vector<size_t> pend;
  for (size_t k = 0; k < cnt; k++) {
    Node& nd = nodes_[ids[k]];
    const int w = which[k];
    double r_max = hr[2 * k], r_min = hr[2 * k + 1];
    double& r_max_0 = w == 0 ? nd.Ur_max : nd.Vr_max;
    if (untouched[k]) r_max_0 = r_max;
    const double atol = o_.abs_tol / nd.lvl, rtol = o_.rel_tol / nd.lvl;
    if (std::abs(r_min) < atol || std::abs(r_min / r_max_0) < rtol) resolved[k] = 1;
    else pend.push_back(k);
  }
  if (pend.empty()) return;
  // undecided nodes: Q12 block from the stored reflectors, and Q(:, d:d+dd) = S(:, d:d+dd)
  {
    std::vector<hssk_qr_desc> fq;
    tr.clear();
    for (size_t k : pend) {
      Node& nd = nodes_[ids[k]];
      const int w = which[k];
      const int m = w == 0 ? nd.mU : nd.mV;
      const double* S = w == 0 ? nd.Srt : nd.Sct;
      double*& Q = w == 0 ? nd.Qr : nd.Qc;
      if (!Q) Q = work_->dbl((size_t)m * dcap_);
      const int c2 = untouched[k] ? 0 : d - dd;
      hssk_qr_desc q = qr[k];
      q.Q = Q + (size_t)c2 * m; q.ldq = m; q.nq = q.cols; q.rdiag = nullptr;
      fq.push_back(q);
      tr.push_back(hssk_transpose_desc{S + d, Q + (size_t)d * m, dd, m, dcap_, m});
    }
    ck(hssk_formq_vbatched(ctx_, fq.data(), (int)fq.size()));
    ck(hssk_transpose(ctx_, tr.data(), (int)tr.size()));
  }
  // iterated classical Gram-Schmidt of the dd new columns against Q12, norms of the first p columns
  const int pc = std::min(dd, o_.p);
  double* nrm = tmp.dbl(2 * pend.size());
  std::vector<hssk_norm_desc> n0, n1;
  std::vector<hssk_gemm_desc> g1, g2;
  for (size_t i = 0; i < pend.size(); i++) {
    size_t k = pend[i];
    Node& nd = nodes_[ids[k]];
    const int w = which[k];
    const int m = w == 0 ? nd.mU : nd.mV;
    double* Q = w == 0 ? nd.Qr : nd.Qc;
    const int q12 = std::min(d, m);
    double* Q3 = Q + (size_t)d * m;
    double* P = tmp.dbl((size_t)q12 * dd);
    n0.push_back(hssk_norm_desc{Q3, m, pc, m, nrm + 2 * i});
    g1.push_back(hssk_gemm_desc{Q, Q3, P, q12, dd, m, m, m, q12, 1, 0, 1.0, 0.0});
    g2.push_back(hssk_gemm_desc{Q, P, Q3, m, dd, q12, m, q12, m, 0, 0, -1.0, 1.0});
    n1.push_back(hssk_norm_desc{Q3, m, pc, m, nrm + 2 * i + 1});
    stats_.f_ortho += 8.0 * m * (double)q12 * dd;
  }
  ck(hssk_sumsq_vbatched(ctx_, n0.data(), (int)n0.size()));
  for (int it = 0; it < 2; it++) {
    ck(hssk_gemm_vbatched(ctx_, g1.data(), (int)g1.size()));
    ck(hssk_gemm_vbatched(ctx_, g2.data(), (int)g2.size()));
  }
  ck(hssk_sumsq_vbatched(ctx_, n1.data(), (int)n1.size()));
  std::vector<double> hn(2 * pend.size());
  ck(hssk_memcpy_d2h(ctx_, hn.data(), nrm, (long long)sizeof(double) * hn.size()));
  for (size_t i = 0; i < pend.size(); i++) {
    size_t k = pend[i];
    Node& nd = nodes_[ids[k]];
    const double atol = o_.abs_tol / nd.lvl, rtol = o_.rel_tol / nd.lvl;
    double S3 = std::sqrt(hn[2 * i]), Q3 = std::sqrt(hn[2 * i + 1]);
    if (Q3 / std::sqrt(double(dd)) < atol || Q3 / S3 < rtol) resolved[k] = 1;
  }
}

}  // namespace HSS
}  // namespace strumpack
