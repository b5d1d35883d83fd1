// DeviceHSS: compression of a kernel matrix from point coordinates.
#include "hss_engine_internal.hpp"

namespace strumpack {
namespace HSS {

// ---------------------------------------------------------------------------------------------
// Kernel-matrix compression from point coordinates:
// HSSMatrix::compress_with_coordinates / compress_recursive_ann / compute_local_samples_ann /
// compute_U_V_bases_ann (HSS/HSSMatrix.compress_kernel.hpp:50-293), level-synchronous.
// No random sketch: the sample of a node is S = K(I, cols) with cols = the neighbours of the node's points
// that lie outside the node (leaf) resp. the union of the children's column sets outside the node (inner),
// I = the node's rows (leaf) resp. its children's skeleton rows; symmetric: V = U, B10 = B01^T.
// ---------------------------------------------------------------------------------------------
void DeviceHSS::compress_kernel(const KernelSpec& ks, const int* user_ann, int user_k) {
  OpGuard op_guard(op_mu_);
  double t0 = now();
  stats_ = PhaseStats();
  struct GramScope { bool& f; explicit GramScope(bool& b) : f(b) { f = true; } ~GramScope() { f = false; } } gram_scope(id_gram_);
  const int N = n_, dim = ks.d;
  if (dim <= 0 || !ks.X) throw std::invalid_argument("compress_kernel: no points");
  int k = std::min(N, std::max(1, user_ann ? user_k : ks.ann));
  for (;;) {
    reset_compression();
    stats_.rounds++;
    // points and neighbour lists
    double* dX = ks.dX ? const_cast<double*>(ks.dX) : work_->dbl((size_t)dim * N);
    if (!ks.dX) ck(hssk_memcpy_h2d(ctx_, dX, ks.X, (long long)sizeof(double) * dim * N));
    hssk_kernel_spec spec{dX, N, dim, ks.type, ks.p, ks.h, ks.lambda};
    double tk0 = now();
    // Column sets on the DEVICE (hssk_colsets: sorted unions through an LDS bitmap) when nothing on the host needs them: the
    // kernel function is one of the library's (a user-defined one is evaluated on the host, with host index lists), the
    // tree is not shared between processes (the cut nodes' sets travel through the host), the point set fits the bitmap.
    // The neighbour lists then stay on the device (25.6 MB at N = 1e5, k = 64) and only the SIZES of the sets come back,
    // one word per node and level; on the host the sets took 5-7 ms of the 102 ms step, most of it the index upload of
    // every level.  STRUMPACK_AMD_KERNEL_HOST_SETS=1: the host form.
    static const bool host_sets_env = [] { const char* e = std::getenv("STRUMPACK_AMD_KERNEL_HOST_SETS"); return e && e[0] == '1'; }();
    const bool dev_sets = !host_sets_env && !ks.eval && !dist_subtree_ && o_.world == 1 && (long long)N <= hssk_colsets_max_universe();
    int* dann = nullptr;
    std::vector<int> ann(dev_sets && !user_ann && !ks.neighbors ? 0 : (size_t)k * N);
    if (user_ann && k == user_k) std::copy(user_ann, user_ann + (size_t)k * N, ann.begin());
    else if (ks.neighbors) ks.neighbors(k, ann.data());
    else {
      // one process per GPU: neighbours of this rank's own points only (its subtree's leaves are all that read them)
      int q0 = 0, q1 = N;
      if (dist_subtree_) { const Node& c = nodes_[cut_nodes_[o_.rank]]; q0 = c.lo; q1 = c.lo + c.m; }
      dann = work_->ints((size_t)k * N);
      // (the search kernel on the device clock: bench.py --workload kernel reports it against the FP32 vector roof --
      //  3 flops per coordinate and candidate: subtract, multiply, add)
      ck(hssk_watch_start(ctx_, 7));
      ck(hssk_knn(ctx_, dX, dim, N, k, q0, q1, dann));
      ck(hssk_watch_stop(ctx_, 7));
      {
        int pairs = 0;
        const double ms = hssk_watch_read_ms(ctx_, 7, &pairs);
        stats_.sketch_kernel_ms += ms;
        stats_.sketch_kernel_flops += 3.0 * dim * (double)N * (double)(q1 - q0);
        stats_.sketch_launches += pairs;
      }
      if (!dev_sets) ck(hssk_memcpy_d2h(ctx_, ann.data() + (size_t)k * q0, dann + (size_t)k * q0, (long long)sizeof(int) * k * (q1 - q0)));
    }
    if ((user_ann && k == user_k) || ks.neighbors) {
      // lists that come from the caller index the bitmaps below: an id outside [0, N) must not reach them (negative = no neighbour)
      for (size_t q = 0; q < ann.size(); q++)
        if (ann[q] >= N) throw std::invalid_argument("compress_kernel: neighbour id " + std::to_string(ann[q]) + " is not a point (n = " + std::to_string(N) + ")");
    }
    if (dev_sets && !dann) {   // lists from the caller: to the device once
      dann = work_->ints((size_t)k * N);
      ck(hssk_memcpy_h2d(ctx_, dann, ann.data(), (long long)sizeof(int) * k * N));
    }
    stats_.t_random += now() - tk0;   // neighbour search (reported in the 'random' slot: it replaces the random sketch)
    std::vector<std::vector<int>> cols(nodes_.size());   // per node: sorted unique column ids outside the node (host form)
    std::vector<int*> dcols(dev_sets ? nodes_.size() : 0, nullptr);   // device form: the sets and their sizes
    std::vector<int> dcnt(dev_sets ? nodes_.size() : 0, 0);
    auto set_size = [&](int id) { return dev_sets ? dcnt[id] : (int)cols[id].size(); };
    bool failed = false;
    // sample panels evaluated inside the Gram products of their IDs (STRUMPACK_AMD_ID_GRAM_GEN=0: evaluated here, as before)
    static const bool gen_env = [] { const char* e = std::getenv("STRUMPACK_AMD_ID_GRAM_GEN"); return !(e && e[0] == '0'); }();
    const bool gen_panels = gen_env && !ks.eval && hssk_gram_gen_supported(&spec, 1);
    // Column sets on the device: one launch per level (a workgroup per node).  They depend on the children's SETS only, not on
    // their IDs, and a parent reads its children's counts on the device (hssk_colset_desc::n0_dev), its own room being the bound
    // min(children's bounds, points outside the node): all levels are launched back to back before the first block is evaluated
    // and every count comes back in ONE read (a read per level in the middle of the stream cost ~0.1 ms each).
    if (dev_sets) {
      const double tc0 = now();
      int* dcount = work_->ints(nodes_.size());
      std::vector<int> bound(nodes_.size(), 0);
      for (auto& ids : own_by_height_) {
        std::vector<hssk_colset_desc> cd;
        for (size_t q = 0; q < ids.size(); q++) {
          const int id = ids[q];
          const Node& nd = nodes_[id];
          if (nd.lvl == 0) continue;
          hssk_colset_desc c{};
          if (nd.leaf()) { c.src0 = dann + (size_t)nd.lo * k; c.n0 = nd.m * k; }
          else {
            c.src0 = dcols[nd.c0]; c.n0 = bound[nd.c0]; c.n0_dev = dcount + nd.c0;
            c.src1 = dcols[nd.c1]; c.n1 = bound[nd.c1]; c.n1_dev = dcount + nd.c1;
          }
          c.lo = nd.lo; c.hi = nd.lo + nd.m;
          bound[id] = (int)std::min<long long>((long long)c.n0 + c.n1, (long long)N - nd.m);
          c.out = dcols[id] = work_->ints((size_t)std::max(c.n0 + c.n1, 1));
          c.count = dcount + id;
          cd.push_back(c);
        }
        if (!cd.empty()) ck(hssk_colsets(ctx_, cd.data(), (int)cd.size(), N));
      }
      std::vector<int> hc(nodes_.size(), 0);
      ck(hssk_memcpy_d2h(ctx_, hc.data(), dcount, (long long)sizeof(int) * nodes_.size()));
      for (auto& ids : own_by_height_)
        for (int id : ids)
          if (nodes_[id].lvl > 0) dcnt[id] = hc[id];
      stats_.t_sketch += now() - tc0;
    }
    auto do_level = [&](const std::vector<int>& ids) {
      if (ids.empty() || failed) return;
      tmp_->rewind();
      double tl0 = now();
      // ---- column sets and row sets (host), one index upload per level
      std::vector<int> hidx;
      std::vector<size_t> roff(ids.size()), coff(ids.size());
      std::vector<std::vector<int>> rows(ids.size());
      // With the column sets on the device a panel's rows need no list from the host either: a leaf's are a range, an inner
      // node's the skeleton indices of its children, which the ID of their level left next to each other on the device
      // (DeviceHSS::finish_id_bookkeeping: one block per level, in the level's order -- siblings are neighbours in it).
      std::vector<const int*> drows(ids.size(), nullptr);
      std::vector<char> fast(ids.size(), 0);
      auto rows_of = [&](size_t q) {
        Node& nd = nodes_[ids[q]];
        std::vector<int>& cs = cols[ids[q]];
        const int lo = nd.lo, hi = nd.lo + nd.m;
        if (dev_sets) {
          if (nd.leaf()) { nd.mU = nd.mV = nd.m; fast[q] = 1; return; }
          const Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
          if (a.dIr && (b.rU == 0 || a.dIr + a.rU == b.dIr)) { nd.mU = nd.mV = a.rU + b.rU; drows[q] = a.dIr; fast[q] = 1; return; }
        }
        if (nd.leaf()) {
          nd.mU = nd.mV = nd.m;
          rows[q].resize(nd.m);
          for (int i = 0; i < nd.m; i++) rows[q][i] = lo + i;
          if (nd.lvl > 0 && !dev_sets) {
            // sorted, duplicate-free ids outside the node: marked in a bitmap over the point set and read back in order
            // (m k ~ 1e4 ids per leaf: cheaper than sorting them)
            std::vector<unsigned long long> bits(((size_t)N + 63) / 64, 0ULL);
            size_t marked = 0;
            for (int i = lo; i < hi; i++)
              for (int j = 0; j < k; j++) {
                const int g = ann[(size_t)i * k + j];
                if (g >= 0 && (g < lo || g >= hi)) {
                  unsigned long long& wd = bits[(size_t)g >> 6];
                  const unsigned long long b = 1ULL << (g & 63);
                  marked += !(wd & b);
                  wd |= b;
                }
              }
            cs.reserve(marked);
            for (size_t wi = 0; wi < bits.size(); wi++) {
              unsigned long long wd = bits[wi];
              while (wd) {
                cs.push_back((int)(wi * 64 + (size_t)__builtin_ctzll(wd)));
                wd &= wd - 1;
              }
            }
          }
        } else {
          Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
          nd.mU = nd.mV = a.rU + b.rU;
          rows[q] = a.Ir;
          rows[q].insert(rows[q].end(), b.Ir.begin(), b.Ir.end());
          if (nd.lvl > 0 && !dev_sets) {
            // union of the children's (sorted, duplicate-free) sets without the ids inside this node
            const std::vector<int>&ca = cols[nd.c0], &cb = cols[nd.c1];
            cs.reserve(ca.size() + cb.size());
            size_t i = 0, j = 0;
            auto keep = [&](int g) { if (g < lo || g >= hi) cs.push_back(g); };
            while (i < ca.size() && j < cb.size()) {
              if (ca[i] < cb[j]) keep(ca[i++]);
              else if (cb[j] < ca[i]) keep(cb[j++]);
              else { keep(ca[i]); i++; j++; }
            }
            while (i < ca.size()) keep(ca[i++]);
            while (j < cb.size()) keep(cb[j++]);
          }
        }
      };
      // the nodes of a level are independent: host threads build their row / column sets side by side (nothing to build with the
      // sets and rows on the device: not worth waking the threads)
      if (dev_sets) for (size_t q = 0; q < ids.size(); q++) rows_of(q);
      else host_parallel_for(ids.size(), rows_of);
      for (size_t q = 0; q < ids.size(); q++) {
        const std::vector<int>& cs = cols[ids[q]];
        roff[q] = hidx.size(); hidx.insert(hidx.end(), rows[q].begin(), rows[q].end());
        coff[q] = hidx.size(); hidx.insert(hidx.end(), cs.begin(), cs.end());
      }
      // the children's column sets are not needed above this level
        for (int id : ids) {
          if (nodes_[id].leaf()) continue;
          std::vector<int>().swap(cols[nodes_[id].c0]);
          std::vector<int>().swap(cols[nodes_[id].c1]);
        }
      int* didx = tmp_->ints(std::max<size_t>(hidx.size(), 1));
      if (!hidx.empty()) ck(hssk_memcpy_h2d(ctx_, didx, hidx.data(), (long long)sizeof(int) * hidx.size()));
      stats_.t_sketch += now() - tl0;   // host column-set construction (the 'sketch' slot of this path)
      // ---- D (leaves), B01 / B10 (inner nodes), sample panels W = K(cols, rows)  [= S^T]
      std::vector<hssk_keval_desc> ev;
      struct HostEv { const int* ri; const int* ci; };   // host copies of a request's index lists (null: the range r0 / c0 + a)
      std::vector<HostEv> hev;
      std::vector<hssk_transpose_desc> tr;
      std::vector<int> idn, which;
      std::vector<double*> Ws;
      std::vector<int> ds;
      std::vector<hssk_keval_desc> wgen;   // per ID panel: its description as a block of the kernel matrix (gen_panels)
      for (size_t q = 0; q < ids.size(); q++) {
        Node& nd = nodes_[ids[q]];
        if (nd.leaf()) {
          nd.D = persist_->dbl((size_t)nd.m * nd.m);
          ev.push_back(hssk_keval_desc{nullptr, nullptr, nd.D, nd.m, nd.m, nd.m, nd.lo, nd.lo});
          hev.push_back(HostEv{nullptr, nullptr});
        } else {
          Node &a = nodes_[nd.c0], &b = nodes_[nd.c1];
          nd.B01 = persist_->dbl((size_t)std::max(a.rU, 1) * std::max(b.rV, 1));
          nd.B10 = persist_->dbl((size_t)std::max(b.rU, 1) * std::max(a.rV, 1));
          if (a.rU > 0 && b.rV > 0) {
            ev.push_back(hssk_keval_desc{a.dIr, b.dIc, nd.B01, a.rU, b.rV, a.rU, 0, 0});
            hev.push_back(HostEv{a.Ir.data(), b.Ic.data()});
            tr.push_back(hssk_transpose_desc{nd.B01, nd.B10, a.rU, b.rV, a.rU, b.rU});
          }
        }
        if (nd.lvl == 0) { nd.Ustate = nd.Vstate = 2; continue; }
        const int m = nd.mU, d = set_size(ids[q]);
        idn.push_back(ids[q]); which.push_back(0); ds.push_back(d);
        double* W = (m > 0 && d > 0) ? tmp_->dbl((size_t)d * m) : nullptr;
        Ws.push_back(W);
        wgen.push_back(hssk_keval_desc{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0});
        if (W) {
          // (rows: the leaf's range, the children's device-resident skeleton indices, or the uploaded list)
          const hssk_keval_desc wd{dev_sets ? dcols[ids[q]] : didx + coff[q], fast[q] ? drows[q] : didx + roff[q], W, d, m, d, 0, fast[q] && !drows[q] ? nd.lo : 0};
          // (the sample panels of a library kernel are evaluated where they are used: id_panels, DeviceHSS::id_gen_)
          if (gen_panels) wgen.back() = wd;
          else {
            ev.push_back(wd);
            hev.push_back(HostEv{hidx.data() + coff[q], hidx.data() + roff[q]});   // (read by the host evaluation only: never with device sets)
          }
        }
      }
      if (!ev.empty() && ks.eval) {
        // user-defined kernel function: the blocks are evaluated on the host's threads, column by column, and uploaded
        std::vector<size_t> boff(ev.size() + 1, 0);
        for (size_t e = 0; e < ev.size(); e++) boff[e + 1] = boff[e] + (size_t)ev[e].nr * ev[e].nc;
        std::vector<double> hb(std::max<size_t>(boff.back(), 1));
        std::vector<std::pair<int, int>> cols_of;   // (request, column)
        for (size_t e = 0; e < ev.size(); e++)
          for (int c = 0; c < ev[e].nc; c++) cols_of.emplace_back((int)e, c);
        host_parallel_for(cols_of.size(), [&](size_t t) {
          const int e = cols_of[t].first, c = cols_of[t].second;
          const hssk_keval_desc& q = ev[e];
          const int gc = hev[e].ci ? hev[e].ci[c] : q.c0 + c;
          double* dst = hb.data() + boff[e] + (size_t)c * q.nr;
          for (int a = 0; a < q.nr; a++) dst[a] = ks.eval(hev[e].ri ? hev[e].ri[a] : q.r0 + a, gc);
        });
        for (size_t e = 0; e < ev.size(); e++)
          if (ev[e].nr > 0 && ev[e].nc > 0)
            ck(hssk_memcpy2d_h2d(ctx_, ev[e].out, sizeof(double) * ev[e].ldo, hb.data() + boff[e], sizeof(double) * ev[e].nr,
                                 sizeof(double) * ev[e].nr, ev[e].nc));
      } else if (!ev.empty()) ck(hssk_kernel_eval_vbatched(ctx_, &spec, ev.data(), (int)ev.size()));
      if (!tr.empty()) ck(hssk_transpose(ctx_, tr.data(), (int)tr.size()));
      if (idn.empty()) return;
      // nodes with an empty column set (d == 0) get rank 0 through a 1 x m zero panel
      for (size_t q = 0; q < idn.size(); q++)
        if (!Ws[q] && nodes_[idn[q]].mU > 0) {
          Ws[q] = tmp_->dbl(nodes_[idn[q]].mU);
          ck(hssk_memset_zero(ctx_, Ws[q], (long long)sizeof(double) * nodes_[idn[q]].mU));
          ds[q] = 1;
        }
      id_gen_ = gen_panels ? &wgen : nullptr;
      id_gen_spec_ = spec;
      try { id_panels(idn, which, Ws, ds); } catch (...) { id_gen_ = nullptr; throw; }
      id_gen_ = nullptr;
      // symmetric: V = U; acceptance test of compute_U_V_bases_ann (:262-272)
      for (size_t q = 0; q < idn.size(); q++) {
        Node& nd = nodes_[idn[q]];
        nd.rV = nd.rU; nd.XV = nd.XU; nd.permV = nd.permU; nd.hpermV = nd.hpermU; nd.Ic = nd.Ir; nd.dIc = nd.dIr; nd.Vstate = nd.Ustate;
        const int d = set_size(idn[q]);
        if (!(d >= nd.m || d >= o_.max_rank || nd.rU + o_.p < d)) failed = true;
      }
    };
    for (auto& ids : own_by_height_) do_level(ids);
    if (dist_subtree_) {
      failed = exchange_cut_kernel(cols, failed);
      for (auto& ids : top_by_height_) do_level(ids);
    }
    ck(hssk_sync(ctx_));
    if (!failed) break;
    if (k >= N) throw std::runtime_error("compress_kernel: the ID did not reach the required accuracy with all points as neighbours");
    k = std::min(2 * k, N);   // compress_with_coordinates: ann_number doubles until the tree compresses (:75)
    if (o_.verbose) std::cout << "# HSS kernel compression: increasing the neighbour count to " << k << std::endl;
  }
  if (dist_subtree_) exchange_node_table();
  free_compress_workspace();
  comm_arena_->reset();
  stats_.d_final = k;
  stats_.t_compress = now() - t0;
  stats_.t_tree = stats_.t_compress - stats_.t_sketch - stats_.t_random;
  if (o_.verbose)
    std::cout << "# HSS kernel compression: neighbours " << stats_.t_random << " s, column sets " << stats_.t_sketch << " s, blocks + ID "
              << stats_.t_tree << " s" << std::endl;
}

}  // namespace HSS
}  // namespace strumpack
