// DeviceHSS: the operand sources of the compression (dense in HBM, sharded, streamed from host memory, callbacks) and the
// entry points that select them.
#include "hss_engine_internal.hpp"

namespace strumpack {
namespace HSS {

double DeviceHSS::Source::sketch_flops(const DeviceHSS& H, int dn) const {
  return 2.0 * products(H) * (double)H.n_ * (double)H.n_ * (H.sj_pat_ ? H.sj_nnz_ : dn);
}

struct DeviceHSS::DenseDeviceSource : DeviceHSS::Source {
  const double* dA;
  long long lda;
  DenseDeviceSource(const double* a, long long l) : dA(a), lda(l) {}
  bool extract_before_sample() const override { return true; }
  bool device_elems(const DeviceHSS&, hssk_elem_src* e) const override { e->A = dA; e->lda = lda; e->use_gen = 0; return true; }
  int done_products_ = 2;     // what the last sample() executed (the route is chosen there)
  bool done_sparse_ = false;
  int products(const DeviceHSS&) const override { return done_products_; }
  double sketch_flops(const DeviceHSS& H, int dn) const override {
    return 2.0 * done_products_ * (double)H.n_ * (double)H.n_ * (done_sparse_ ? H.sj_nnz_ : dn);
  }
  void sample(DeviceHSS& H, int r0, int dn) override {
    const long long N = H.n_;
    // AFunctor::operator()(Rr,Rc,Sr,Sc), HSSExtra.hpp:236-239, in the transposed sample layout.
    // Multi-GPU: this rank computes the sample columns [j0, j1) only (rows j0:j1 of A for Sr,
    // columns j0:j1 of A for Sc).
    long long j0 = 0, j1 = N;
    if (H.dist_subtree_) {           // this rank's subtree range: nothing is exchanged here
      const Node& c = H.nodes_[H.cut_nodes_[H.o_.rank]];
      j0 = c.lo; j1 = c.lo + c.m;
    } else if (H.o_.world > 1) {     // fallback: equal column shards + all-gather of the samples
      j0 = std::min(N, H.cols_per_rank_ * H.o_.rank); j1 = std::min(N, j0 + H.cols_per_rank_);
    }
    const long long nloc = j1 - j0;
    // SJLT sketch: stream A once per product instead of a dense GEMM (blocks wider than the kernel's LDS tile, or
    // STRUMPACK_AMD_SJLT_DENSE=1, multiply with the dense form of the pattern)
    static const bool sj_dense = std::getenv("STRUMPACK_AMD_SJLT_DENSE") && std::atoi(std::getenv("STRUMPACK_AMD_SJLT_DENSE"));
    done_sparse_ = H.sj_pat_ && dn <= 1024 && !sj_dense;
    done_products_ = (!done_sparse_ && H.o_.symmetric) ? 1 : 2;
    if (nloc > 0 && done_sparse_) {
      for (int t = 0; t < 2; t++) {
        const double* Aop = t == 0 ? dA + j0 : dA + j0 * lda;
        double* St = (t == 0 ? H.Srt_ : H.Sct_) + r0 + j0 * H.dcap_;
        ck(hssk_sjlt_sketch(H.ctx_, t, nloc, N, Aop, lda, H.sj_pat_, H.sj_nnz_, dn, St, H.dcap_));
        ck(hssk_sync(H.ctx_));
        float ms = hssk_last_dgemm_ms(H.ctx_);
        if (ms > 0) {
          H.stats_.sketch_kernel_ms += ms; H.stats_.sketch_launches++;
          H.stats_.sketch_kernel_flops += hssk_last_dgemm_flops(H.ctx_);
          H.stats_.sketch_kernel_bytes += 8.0 * (double)nloc * (double)N;
        }
      }
    } else if (nloc > 0) {
      // (the products' kernel timings are set aside and read at the end of compress(): no synchronisation between the products
      //  nor behind them)
      ck(hssk_dgemm(H.ctx_, 1, dn, nloc, N, 1.0, H.Rt_ + r0, H.dcap_, dA + j0, lda, 0.0, H.Srt_ + r0 + j0 * H.dcap_, H.dcap_));
      ck(hssk_dgemm_timing_defer(H.ctx_));
      if (H.o_.symmetric) {   // A^T R = A R: the second product is a copy of the first
        hssk_colgather_desc cp{H.Srt_ + r0 + j0 * H.dcap_, H.Sct_ + r0 + j0 * H.dcap_, nullptr, dn, (int)nloc, H.dcap_, H.dcap_, 0};
        ck(hssk_gather_cols(H.ctx_, &cp, 1));
      } else {
        ck(hssk_dgemm(H.ctx_, 0, dn, nloc, N, 1.0, H.Rt_ + r0, H.dcap_, dA + j0 * lda, lda, 0.0, H.Sct_ + r0 + j0 * H.dcap_, H.dcap_));
        ck(hssk_dgemm_timing_defer(H.ctx_));
      }
    }
    if (H.o_.world > 1 && !H.dist_subtree_) {
      const long long bytes = (long long)sizeof(double) * H.dcap_ * H.cols_per_rank_;
      H.comm(H.Srt_, bytes);
      H.comm(H.Sct_, bytes);
    }
  }
  void extract(DeviceHSS& H, const std::vector<ElemReq>& reqs) override {
    std::vector<hssk_elem_desc> d;
    d.reserve(reqs.size());
    for (auto& r : reqs)
      if (r.m > 0 && r.n > 0) d.push_back(hssk_elem_desc{dA, lda, r.dI, r.dJ, r.i0, r.j0, r.dB, r.m, r.n, r.ldb, 0});
    if (!d.empty()) ck(hssk_gather_elems(H.ctx_, d.data(), (int)d.size()));
  }
};

// Sharded dense operand (one process per GPU, subtree ownership): this rank holds the columns [j0, j1) of its subtree
// (n x nloc) and, optionally, the rows [j0, j1) (nloc x n) -- never the whole matrix.
//   sketch:  Sc(j0:j1, :) = A(:, j0:j1)^H R                                   -- local
//            Sr(j0:j1, :) = A(j0:j1, :) R                                     -- local when the row block is given, else
//            Sr = sum_g A(:, cols_g) R(cols_g, :): every rank multiplies its column block with its rows of R and the
//            partial N x d products are summed to the owners of the rows (reduce-scatter over xGMI, SURVEY.md 8(e)(5):
//            the "reduce of off-diagonal contributions"; per-rank flops are the same 2 n nloc d either way)
//   elements: blocks inside the subtree come from the column block; the coupling blocks of the replicated top nodes
//            B01 = A(Ir_0, Ic_1) straddle the ranks: every rank fills in the rows (columns) it holds, zeros elsewhere,
//            and the partial blocks are summed over the ranks (a few r x r blocks per top level)
struct DeviceHSS::ShardedDenseSource : DeviceHSS::Source {
  const double* dRows;
  long long ldr;
  const double* dCols;
  long long ldc;
  ShardedDenseSource(const double* r, long long lr, const double* c, long long lc) : dRows(r), ldr(lr), dCols(c), ldc(lc) {}
  // (blocks inside the rank's subtree lie in its column block, addressed with global column indices)
  bool device_elems(const DeviceHSS& H, hssk_elem_src* e) const override {
    const Node& c = H.nodes_[H.o_.world == 1 ? 0 : H.cut_nodes_[H.o_.rank]];
    e->A = dCols - (long long)c.lo * ldc; e->lda = ldc; e->use_gen = 0;
    return true;
  }
  void sample(DeviceHSS& H, int r0, int dn) override {
    const bool single = H.o_.world == 1;   // one rank: its "shard" is the whole operand (same code path, no collective)
    if (!single && !H.dist_subtree_) throw std::invalid_argument("sharded operand: the tree cannot be cut into one subtree per rank (world must be a power of two and the tree complete down to that depth)");
    if (H.sj_pat_) throw std::invalid_argument("sharded operand: the SJLT sketch needs the replicated-operand interface");
    const long long N = H.n_;
    const Node& c = H.nodes_[single ? 0 : H.cut_nodes_[H.o_.rank]];
    const long long j0 = c.lo, nloc = c.m;
    auto timed = [&] { ck(hssk_dgemm_timing_defer(H.ctx_)); };   // (read at the end of compress())
    ck(hssk_dgemm(H.ctx_, 0, dn, nloc, N, 1.0, H.Rt_ + r0, H.dcap_, dCols, ldc, 0.0, H.Sct_ + r0 + j0 * H.dcap_, H.dcap_));
    timed();
    if (dRows) {
      ck(hssk_dgemm(H.ctx_, 1, dn, nloc, N, 1.0, H.Rt_ + r0, H.dcap_, dRows, ldr, 0.0, H.Srt_ + r0 + j0 * H.dcap_, H.dcap_));
      timed();
    } else {
      // P (dn x N) = R(cols, :)^T A(:, cols)^T : the contribution of this rank's columns to every row of Sr
      H.tmp_->rewind();
      double* P = H.tmp_->dbl((size_t)dn * N);
      double* mineP = H.tmp_->dbl((size_t)dn * nloc);
      ck(hssk_dgemm(H.ctx_, 1, dn, N, nloc, 1.0, H.Rt_ + r0 + j0 * H.dcap_, H.dcap_, dCols, ldc, 0.0, P, dn));
      timed();
      const double tc = now();
      std::vector<long long> offs(H.o_.world), counts(H.o_.world);
      for (int g = 0; g < H.o_.world; g++) {
        const Node& cg = H.nodes_[single ? 0 : H.cut_nodes_[g]];
        offs[g] = (long long)cg.lo * dn; counts[g] = (long long)cg.m * dn;
      }
      H.reduce_scatter_sum(P, offs, counts, mineP);
      hssk_colgather_desc cp{mineP, H.Srt_ + r0 + j0 * H.dcap_, nullptr, dn, (int)nloc, dn, H.dcap_, 0};
      ck(hssk_gather_cols(H.ctx_, &cp, 1));
      ck(hssk_sync(H.ctx_));
      H.stats_.t_comm += now() - tc;
    }
  }
  void extract(DeviceHSS& H, const std::vector<ElemReq>& reqs) override {
    const Node& c = H.nodes_[H.o_.world == 1 ? 0 : H.cut_nodes_[H.o_.rank]];
    const int j0 = c.lo, j1 = c.lo + c.m;
    auto inside = [&](const std::vector<int>* h, int i0, int cnt) {
      if (!h) return i0 >= j0 && i0 + cnt <= j1;
      for (int i = 0; i < cnt; i++) if ((*h)[i] < j0 || (*h)[i] >= j1) return false;
      return true;
    };
    std::vector<hssk_elem_desc> own, part, back;
    size_t tot = 0;
    for (auto& r : reqs) if (r.m > 0 && r.n > 0 && !(inside(r.hI, r.i0, r.m) && inside(r.hJ, r.j0, r.n))) tot += (size_t)r.m * r.n;
    double* stage = tot ? H.comm_arena_->dbl(tot) : nullptr;
    size_t off = 0;
    for (auto& r : reqs) {
      if (r.m <= 0 || r.n <= 0) continue;
      if (inside(r.hI, r.i0, r.m) && inside(r.hJ, r.j0, r.n)) {
        // column block addressed with global column indices: A'(i, j) = dCols[i + (j - j0) ldc]
        own.push_back(hssk_elem_desc{dCols - (long long)j0 * ldc, ldc, r.dI, r.dJ, r.i0, r.j0, r.dB, r.m, r.n, r.ldb, 0, 0, 0, 0, 0});
      } else {
        double* T = stage + off;
        off += (size_t)r.m * r.n;
        if (dRows) part.push_back(hssk_elem_desc{dRows - j0, ldr, r.dI, r.dJ, r.i0, r.j0, T, r.m, r.n, r.m, 0, j0, j1, 0, 0});
        else part.push_back(hssk_elem_desc{dCols - (long long)j0 * ldc, ldc, r.dI, r.dJ, r.i0, r.j0, T, r.m, r.n, r.m, 0, 0, 0, j0, j1});
        back.push_back(hssk_elem_desc{T, r.m, nullptr, nullptr, 0, 0, r.dB, r.m, r.n, r.ldb, 0, 0, 0, 0, 0});
      }
    }
    if (!own.empty()) ck(hssk_gather_elems(H.ctx_, own.data(), (int)own.size()));
    if (!part.empty()) {
      const double tc = now();
      ck(hssk_gather_elems(H.ctx_, part.data(), (int)part.size()));
      H.allreduce_sum(stage, (long long)tot);
      ck(hssk_gather_elems(H.ctx_, back.data(), (int)back.size()));
      H.stats_.t_comm += now() - tc;
    }
  }
};

// Host-resident operand, streamed: column blocks A(:, c0:c1) cross PCIe once per sampling round through two device
// buffers; the upload of block b+1 (copy stream, pinned bounce buffers filled by host threads) overlaps the two sketch
// GEMMs of block b (compute stream):
//   Sc(c0:c1, :)  = A(:, c0:c1)^H R            -- complete for these columns
//   Sr(:, :)     += A(:, c0:c1) R(c0:c1, :)    -- the block's contribution to every row
// The diagonal blocks and the coupling blocks are read from the host operand afterwards (contiguous 2-D copies, resp. a
// multi-threaded host gather of the few scattered entries + one upload).  At most 2 x n x nb doubles of A are in HBM.
// An operand of another scalar type (float, complex<float>, complex<double>: `dtype`, the reference's other instantiations,
// HSS/HSSMatrix.cpp:513-516) crosses the link in ITS format -- half / a quarter of the bytes of its double-precision real image --
// into two staging buffers; hssk_expand_image writes the image of a block (interleaved [re -im; im re] for complex scalars,
// HSSMatrixPromoted.hpp) into the one block buffer the GEMMs read.
struct DeviceHSS::HostBlockSource : DeviceHSS::Source {
  const void* hA;          // column-major host matrix (scalars of `dtype`), or null when `fill` evaluates the columns
  long long lda;           // in scalars
  int dtype = HSSK_DT_F64;
  const host_fill_t* fill;
  const host_elem_t* elem;
  double* dBuf[2] = {nullptr, nullptr};
  double* dNat[2] = {nullptr, nullptr};   // staging of the native blocks (dtype != HSSK_DT_F64)
  long long nb = 0;
  int gen = -1;   // compression attempt the buffers were carved in (a restart resets the work arena)
  // diagonal blocks of the leaves, copied out of the column blocks while they pass through the device (first sample of an
  // attempt): extract() then serves them from here instead of gathering them from host memory again
  std::vector<double*> dcache;   // by node id
  HostBlockSource(const void* a, long long l, const host_fill_t* f, const host_elem_t* e, int dt = HSSK_DT_F64)
      : hA(a), lda(l), dtype(dt), fill(f), elem(e) {}
  long long reals() const { return dtype == HSSK_DT_C32 || dtype == HSSK_DT_C64 ? 2 : 1; }   // image rows per scalar row
  size_t esize() const { return dtype == HSSK_DT_F32 ? 4 : dtype == HSSK_DT_C64 ? 16 : 8; }
  // entry (I, J) of the image
  double image_at(size_t I, size_t J) const {
    switch (dtype) {
      case HSSK_DT_F32: return (double)((const float*)hA)[I + J * (size_t)lda];
      case HSSK_DT_C32: {
        const float* z = (const float*)hA + 2 * (I / 2 + (J / 2) * (size_t)lda);
        return (I & 1) == (J & 1) ? (double)z[0] : ((I & 1) ? (double)z[1] : -(double)z[1]);
      }
      case HSSK_DT_C64: {
        const double* z = (const double*)hA + 2 * (I / 2 + (J / 2) * (size_t)lda);
        return (I & 1) == (J & 1) ? z[0] : ((I & 1) ? z[1] : -z[1]);
      }
      default: return ((const double*)hA)[I + J * (size_t)lda];
    }
  }
  void sample(DeviceHSS& H, int r0, int dn) override {
    if (H.o_.world > 1) throw std::invalid_argument("host-resident operands are single-GPU (use the device / sharded interfaces)");
    // (an SJLT sketching matrix is applied in its dense form here -- Rt_ holds it, DeviceHSS::fill_random: the streaming
    // SJLT kernels overwrite their output, the blocks of a streamed operand have to accumulate)
    const long long N = H.n_;
    const bool typed = hA && dtype != HSSK_DT_F64;
    const long long W = typed ? reals() : 1, ns = N / W;   // scalar rows
    if (typed && N % W) throw std::logic_error("image dimension of a complex operand must be even");
    const bool first = gen != H.attempt_;
    if (first) {
      gen = H.attempt_;
      // ~1.5 GB per buffer (STRUMPACK_AMD_HOST_BLOCK_MB, or _KB for small operands, to change), whole 64-column tiles of the
      // sketch GEMM
      long long bytes = 1536LL << 20;
      if (const char* e = std::getenv("STRUMPACK_AMD_HOST_BLOCK_MB")) bytes = std::max(1LL, std::atoll(e)) << 20;
      if (const char* e = std::getenv("STRUMPACK_AMD_HOST_BLOCK_KB")) bytes = std::max(1LL, std::atoll(e)) << 10;
      nb = std::max<long long>(64, bytes / (8 * std::max<long long>(N, 1)) / 64 * 64);
      nb = std::min(nb, (N + 63) / 64 * 64);
      dBuf[0] = H.work_->dbl((size_t)N * nb);
      if (!typed) dBuf[1] = H.work_->dbl((size_t)N * nb);
      else for (int k = 0; k < 2; k++) dNat[k] = H.work_->dbl(((size_t)ns * (size_t)(nb / W) * esize() + 7) / 8);
    }
    const long long nblk = (N + nb - 1) / nb;
    const bool capture = first;
    if (capture) {
      dcache.assign(H.nodes_.size(), nullptr);
      for (size_t id = 0; id < H.nodes_.size(); id++)
        if (H.nodes_[id].leaf() && H.nodes_[id].m > 0) dcache[id] = H.work_->dbl((size_t)H.nodes_[id].m * H.nodes_[id].m);
    }
    std::vector<double> tmp;   // columns evaluated by `fill` (packed into the pinned ring before the call returns)
    auto upload = [&](long long b) {
      const long long c0 = b * nb, c1 = std::min(N, c0 + nb);
      if (typed)   // (c0 and c1 are even: nb is a multiple of 64, N = 2 ns)
        ck(hssk_h2d_bytes_async(H.ctx_, dNat[b & 1], (long long)(ns * esize()), (const char*)hA + (size_t)(c0 / W) * lda * esize(),
                                (long long)(lda * esize()), (long long)(ns * esize()), (c1 - c0) / W));
      else if (hA) ck(hssk_h2d_block_async(H.ctx_, dBuf[b & 1], N, (const double*)hA + (size_t)c0 * lda, lda, N, c1 - c0));
      else {
        tmp.resize((size_t)N * (c1 - c0));
        (*fill)(c0, c1, tmp.data());
        ck(hssk_h2d_block_async(H.ctx_, dBuf[b & 1], N, tmp.data(), N, N, c1 - c0));
      }
    };
    upload(0);
    for (long long b = 0; b < nblk; b++) {
      const long long c0 = b * nb, c1 = std::min(N, c0 + nb);
      ck(hssk_copy_fence(H.ctx_));          // the GEMMs below wait for block b
      const double* Ab = typed ? dBuf[0] : dBuf[b & 1];
      if (typed) {
        // the image of block b (behind the GEMMs of block b - 1 in stream order); its staging buffer is free from here on
        ck(hssk_expand_image(H.ctx_, dBuf[0], N, dNat[b & 1], ns, ns, (c1 - c0) / W, dtype));
        ck(hssk_compute_mark(H.ctx_, (int)(b & 1)));
      }
      ck(hssk_dgemm(H.ctx_, 0, dn, c1 - c0, N, 1.0, H.Rt_ + r0, H.dcap_, Ab, N, 0.0, H.Sct_ + r0 + c0 * H.dcap_, H.dcap_));
      ck(hssk_dgemm(H.ctx_, 1, dn, N, c1 - c0, 1.0, H.Rt_ + r0 + c0 * H.dcap_, H.dcap_, Ab, N, b ? 1.0 : 0.0, H.Srt_ + r0, H.dcap_));
      if (capture) {
        // the columns of the leaves' diagonal blocks that lie in this column block
        std::vector<hssk_colgather_desc> dg;
        for (size_t id = 0; id < H.nodes_.size(); id++) {
          const Node& nd = H.nodes_[id];
          if (!dcache[id]) continue;
          const long long a = std::max<long long>(nd.lo, c0), e = std::min<long long>(nd.lo + nd.m, c1);
          if (a >= e) continue;
          dg.push_back(hssk_colgather_desc{Ab + nd.lo + (size_t)(a - c0) * N, dcache[id] + (size_t)(a - nd.lo) * nd.m, nullptr, nd.m, (int)(e - a),
                                           (int)N, nd.m, 0});
        }
        if (!dg.empty()) ck(hssk_gather_cols(H.ctx_, dg.data(), (int)dg.size()));
      }
      // block b + 1 overwrites the buffer the work of block b - 1 read -- and only that: the upload (whose packing blocks
      // this thread for most of its duration) is issued AFTER the GEMMs of block b, which then run under it, and it does
      // not wait for them.  (Issued before them, every block stalled the copy stream for the 2.4 ms of its GEMMs.)
      if (!typed) ck(hssk_compute_mark(H.ctx_, (int)(b & 1)));
      if (b + 1 < nblk) {
        ck(hssk_copy_wait(H.ctx_, (int)((b + 1) & 1)));
        upload(b + 1);
      }
    }
    ck(hssk_sync(H.ctx_));
  }
  void extract(DeviceHSS& H, const std::vector<ElemReq>& reqs) override {
    // every requested block is compact (ldb == m): gather on the host threads into one staging image, one upload each
    std::vector<size_t> off(reqs.size() + 1, 0);
    for (size_t k = 0; k < reqs.size(); k++) off[k + 1] = off[k] + (size_t)std::max(reqs[k].m, 0) * std::max(reqs[k].n, 0);
    // blocks served from the device-side cache of leaf diagonal blocks
    std::vector<const double*> hit(reqs.size(), nullptr);
    if (!dcache.empty()) {
      std::vector<std::pair<int, size_t>> by_lo;   // (lo, node) of the cached leaves, for the lookups below
      for (size_t id = 0; id < H.nodes_.size(); id++) if (dcache[id]) by_lo.push_back({H.nodes_[id].lo, id});
      std::sort(by_lo.begin(), by_lo.end());
      std::vector<hssk_colgather_desc> cp;
      for (size_t k = 0; k < reqs.size(); k++) {
        const ElemReq& r = reqs[k];
        if (r.hI || r.hJ || r.i0 != r.j0 || r.m != r.n || r.m <= 0 || gen != H.attempt_) continue;
        auto it = std::lower_bound(by_lo.begin(), by_lo.end(), std::make_pair(r.i0, size_t(0)));
        if (it == by_lo.end() || it->first != r.i0 || H.nodes_[it->second].m != r.m) continue;
        hit[k] = dcache[it->second];
        cp.push_back(hssk_colgather_desc{hit[k], r.dB, nullptr, r.m, r.n, r.m, r.ldb, 0});
      }
      if (!cp.empty()) ck(hssk_gather_cols(H.ctx_, cp.data(), (int)cp.size()));
    }
    for (size_t k = 0; k < reqs.size(); k++)
      if (hit[k]) off[k + 1] = off[k];   // (no staging space for served requests)
      else off[k + 1] = off[k] + (size_t)std::max(reqs[k].m, 0) * std::max(reqs[k].n, 0);
    std::unique_ptr<double[]> img_store(new double[std::max<size_t>(off.back(), 1)]);   // (not value-initialised: every element is written)
    struct { double* p; double* data() { return p; } } img{img_store.get()};
    host_parallel_for(reqs.size(), [&](size_t k) {
      const ElemReq& r = reqs[k];
      if (r.m <= 0 || r.n <= 0 || hit[k]) return;
      double* B = img.data() + off[k];
      if (hA && dtype != HSSK_DT_F64) {
        for (int j = 0; j < r.n; j++) {
          const size_t J = (size_t)(r.hJ ? (*r.hJ)[j] : r.j0 + j);
          for (int i = 0; i < r.m; i++) B[i + (size_t)j * r.m] = image_at((size_t)(r.hI ? (*r.hI)[i] : r.i0 + i), J);
        }
      } else if (hA) {
        for (int j = 0; j < r.n; j++) {
          const double* col = (const double*)hA + (size_t)(r.hJ ? (*r.hJ)[j] : r.j0 + j) * lda;
          if (r.hI) for (int i = 0; i < r.m; i++) B[i + (size_t)j * r.m] = col[(*r.hI)[i]];
          else std::memcpy(B + (size_t)j * r.m, col + r.i0, sizeof(double) * r.m);
        }
      } else {
        std::vector<int> I(r.m), J(r.n);
        for (int i = 0; i < r.m; i++) I[i] = r.hI ? (*r.hI)[i] : r.i0 + i;
        for (int j = 0; j < r.n; j++) J[j] = r.hJ ? (*r.hJ)[j] : r.j0 + j;
        (*elem)(r.m, I.data(), r.n, J.data(), B, r.m);
      }
    });
    for (size_t k = 0; k < reqs.size(); k++) {
      const ElemReq& r = reqs[k];
      if (r.m <= 0 || r.n <= 0 || hit[k]) continue;
      if (r.ldb == r.m) ck(hssk_upload_async(H.ctx_, r.dB, img.data() + off[k], (long long)(sizeof(double) * (off[k + 1] - off[k]))));
      else ck(hssk_memcpy2d_h2d(H.ctx_, r.dB, sizeof(double) * r.ldb, img.data() + off[k], sizeof(double) * r.m, sizeof(double) * r.m, r.n));
    }
  }
};

struct DeviceHSS::CallbackSource : DeviceHSS::Source {
  const host_mult_t* mult;
  const host_sample_t* usample = nullptr;   // user_defined_random: the callee fills the random block as well
  const host_elem_t& elem;
  CallbackSource(const host_mult_t& m, const host_elem_t& e) : mult(&m), elem(e) {}
  CallbackSource(const host_sample_t& s, const host_elem_t& e) : mult(nullptr), usample(&s), elem(e) {}
  void sample(DeviceHSS& H, int r0, int dn) override {
    if (H.o_.world > 1) throw std::invalid_argument("the host-callback interface is single-GPU");
    if (usample) {
      // compress_stable.hpp:126-141 with user_defined_random: Amult(Rr_new, Rc_new, Sr_new, Sc_new) fills all four blocks
      const int N = H.n_;
      if (N == 0 || dn == 0) return;
      std::vector<double> R((size_t)N * dn), Sr((size_t)N * dn), Sc((size_t)N * dn);
      (*usample)(N, dn, R.data(), Sr.data(), Sc.data());
      double* dT = H.tmp_->dbl((size_t)N * dn);
      const double* src[3] = {R.data(), Sr.data(), Sc.data()};
      double* dst[3] = {H.Rt_ + r0, H.Srt_ + r0, H.Sct_ + r0};
      for (int q = 0; q < 3; q++) {
        ck(hssk_memcpy_h2d(H.ctx_, dT, src[q], (long long)(sizeof(double) * (size_t)N * dn)));
        hssk_transpose_desc b{dT, dst[q], N, dn, N, H.dcap_};
        ck(hssk_transpose(H.ctx_, &b, 1));
        ck(hssk_sync(H.ctx_));   // dT is reused
      }
      return;
    }
    // The user multiplies column-major N x dn blocks (AFunctor / Amult of the reference, HSSExtra.hpp:231-239); the
    // samples live transposed on the device (dn x N rows of Rt / Srt / Sct).  The transposes run on the device; the host
    // sees contiguous blocks only.
    const int N = H.n_;
    if (N == 0 || dn == 0) return;
    double* dT = H.tmp_->dbl((size_t)N * dn);   // N x dn, column-major
    std::vector<double> R((size_t)N * dn), S((size_t)N * dn);
    hssk_transpose_desc t{H.Rt_ + r0, dT, dn, N, H.dcap_, N};
    ck(hssk_transpose(H.ctx_, &t, 1));
    ck(hssk_memcpy_d2h(H.ctx_, R.data(), dT, (long long)(sizeof(double) * R.size())));   // (synchronous)
    for (int pass = 0; pass < 2; pass++) {
      (*mult)(pass == 0 ? 'N' : 'C', N, dn, R.data(), N, S.data(), N);
      ck(hssk_memcpy_h2d(H.ctx_, dT, S.data(), (long long)(sizeof(double) * S.size())));
      hssk_transpose_desc b{dT, (pass == 0 ? H.Srt_ : H.Sct_) + r0, N, dn, N, H.dcap_};
      ck(hssk_transpose(H.ctx_, &b, 1));
      ck(hssk_sync(H.ctx_));   // dT is reused by the next pass
    }
  }
  void extract(DeviceHSS& H, const std::vector<ElemReq>& reqs) override {
    for (auto& r : reqs) {
      if (r.m <= 0 || r.n <= 0) continue;
      std::vector<int> I(r.m), J(r.n);
      for (int i = 0; i < r.m; i++) I[i] = r.hI ? (*r.hI)[i] : r.i0 + i;
      for (int j = 0; j < r.n; j++) J[j] = r.hJ ? (*r.hJ)[j] : r.j0 + j;
      std::vector<double> B((size_t)r.m * r.n);
      elem(r.m, I.data(), r.n, J.data(), B.data(), r.m);
      ck(hssk_memcpy2d_h2d(H.ctx_, r.dB, sizeof(double) * r.ldb, B.data(), sizeof(double) * r.m, sizeof(double) * r.m, r.n));
    }
  }
};

// Operand given by a formula (hssk_gen): nothing is stored, on one rank or many.  Sampling = the sketch GEMM whose second
// operand is evaluated inside the kernel (same ranges, stats and collectives as the dense operand resident in HBM, whose
// results it reproduces bit for bit); scattered entries = the formula at the requested indices.
struct DeviceHSS::GeneratorSource : DeviceHSS::Source {
  hssk_gen g;
  explicit GeneratorSource(const hssk_gen& g_) : g(g_) {}
  bool extract_before_sample() const override { return true; }
  bool device_elems(const DeviceHSS&, hssk_elem_src* e) const override { e->A = nullptr; e->lda = 0; e->gen = g; e->use_gen = 1; return true; }
  int products(const DeviceHSS& H) const override { return H.o_.symmetric ? 1 : 2; }
  void sample(DeviceHSS& H, int r0, int dn) override {
    if (H.sj_pat_) throw std::invalid_argument("generated operand: the SJLT sketch streams a stored matrix; use the Gaussian sketch");
    const long long N = H.n_;
    long long j0 = 0, j1 = N;
    if (H.dist_subtree_) {
      const Node& c = H.nodes_[H.cut_nodes_[H.o_.rank]];
      j0 = c.lo; j1 = c.lo + c.m;
    } else if (H.o_.world > 1) {
      j0 = std::min(N, H.cols_per_rank_ * H.o_.rank); j1 = std::min(N, j0 + H.cols_per_rank_);
    }
    const long long nloc = j1 - j0;
    auto timed = [&] { ck(hssk_dgemm_timing_defer(H.ctx_)); };   // (read at the end of compress())
    if (nloc > 0) {
      // Sr(j0:j1, :) = A(j0:j1, :) R  ->  op(G)(k, j) = G(j0 + j, k);   Sc(j0:j1, :) = A(:, j0:j1)^T R  ->  op(G)(k, j) = G(k, j0 + j)
      ck(hssk_sketch_gen(H.ctx_, &g, 1, dn, nloc, N, j0, 1.0, H.Rt_ + r0, H.dcap_, 0.0, H.Srt_ + r0 + j0 * H.dcap_, H.dcap_));
      timed();
      if (H.o_.symmetric) {   // A^T R = A R: the second product is a copy of the first
        hssk_colgather_desc cp{H.Srt_ + r0 + j0 * H.dcap_, H.Sct_ + r0 + j0 * H.dcap_, nullptr, dn, (int)nloc, H.dcap_, H.dcap_, 0};
        ck(hssk_gather_cols(H.ctx_, &cp, 1));
      } else {
        ck(hssk_sketch_gen(H.ctx_, &g, 0, dn, nloc, N, j0, 1.0, H.Rt_ + r0, H.dcap_, 0.0, H.Sct_ + r0 + j0 * H.dcap_, H.dcap_));
        timed();
      }
    }
    if (H.o_.world > 1 && !H.dist_subtree_) {
      const long long bytes = (long long)sizeof(double) * H.dcap_ * H.cols_per_rank_;
      H.comm(H.Srt_, bytes);
      H.comm(H.Sct_, bytes);
    }
  }
  void extract(DeviceHSS& H, const std::vector<ElemReq>& reqs) override {
    std::vector<hssk_elem_desc> d;
    d.reserve(reqs.size());
    for (auto& r : reqs)
      if (r.m > 0 && r.n > 0) d.push_back(hssk_elem_desc{nullptr, 0, r.dI, r.dJ, r.i0, r.j0, r.dB, r.m, r.n, r.ldb, 0});
    if (!d.empty()) ck(hssk_gen_elems(H.ctx_, &g, d.data(), (int)d.size()));
  }
};

// ---------------------------------------------------------------------------------------------
// compression driver
// ---------------------------------------------------------------------------------------------
void DeviceHSS::compress_generator(const hssk_gen& g) {
  GeneratorSource s(g);
  check_symmetry(s);
  compress(s);
}
// EngineOptions::symmetric == 2: the hint is checked on 512 x 512 scattered entries against their mirror images
void DeviceHSS::check_symmetry(Source& src) {
  if (o_.symmetric != 2 || n_ < 2) return;
  const int q = std::min(512, n_);
  std::vector<int> I(q), J(q);
  std::minstd_rand g(12345);
  for (int i = 0; i < q; i++) { I[i] = (int)(g() % (unsigned)n_); J[i] = (int)(g() % (unsigned)n_); }
  tmp_->rewind();
  int* dI = tmp_->ints(q);
  int* dJ = tmp_->ints(q);
  double* B1 = tmp_->dbl((size_t)q * q);
  double* B2 = tmp_->dbl((size_t)q * q);
  ck(hssk_memcpy_h2d(ctx_, dI, I.data(), (long long)sizeof(int) * q));
  ck(hssk_memcpy_h2d(ctx_, dJ, J.data(), (long long)sizeof(int) * q));
  std::vector<ElemReq> rq{ElemReq{dI, dJ, &I, &J, 0, 0, q, q, B1, q}, ElemReq{dJ, dI, &J, &I, 0, 0, q, q, B2, q}};
  src.extract(*this, rq);
  std::vector<double> h1((size_t)q * q), h2((size_t)q * q);
  ck(hssk_memcpy_d2h(ctx_, h1.data(), B1, (long long)sizeof(double) * q * q));
  ck(hssk_memcpy_d2h(ctx_, h2.data(), B2, (long long)sizeof(double) * q * q));
  double dmax = 0, amax = 0;
  for (int j = 0; j < q; j++)
    for (int i = 0; i < q; i++) {
      dmax = std::max(dmax, std::abs(h1[i + (size_t)j * q] - h2[j + (size_t)i * q]));
      amax = std::max(amax, std::abs(h1[i + (size_t)j * q]));
    }
  if (dmax > 1e-14 * std::max(amax, 1e-300)) throw std::invalid_argument("compress: the operand was declared symmetric and is not (sampled entries differ from their mirror images)");
}

void DeviceHSS::compress_dense_device(const double* dA, long long lda) {
  DenseDeviceSource s(dA, lda);
  check_symmetry(s);
  compress(s);
}
void DeviceHSS::compress_dense_host(const double* A, long long lda) {
  HostBlockSource s(A, lda, nullptr, nullptr);
  compress(s);
}
void DeviceHSS::compress_dense_host_typed(const void* A, long long lda, int dtype) {
  if (dtype != HSSK_DT_F32 && dtype != HSSK_DT_C32 && dtype != HSSK_DT_C64) throw std::invalid_argument("compress_dense_host_typed: unknown scalar type");
  HostBlockSource s(A, lda, nullptr, nullptr, dtype);
  compress(s);
}
void DeviceHSS::compress_host_blocks(const host_fill_t& fill, const host_elem_t& elem) {
  HostBlockSource s(nullptr, 0, &fill, &elem);
  compress(s);
}
void DeviceHSS::compress_dense_device_sharded(const double* dRows, long long ldr, const double* dCols, long long ldc) {
  if (!dCols) throw std::invalid_argument("sharded operand: the column block is required");
  ShardedDenseSource s(dRows, ldr, dCols, ldc);
  compress(s);
}
bool DeviceHSS::shard_range(int rank, int& lo, int& hi) const {
  if (o_.world == 1) { lo = 0; hi = n_; return true; }
  if (!dist_subtree_ || rank < 0 || rank >= (int)cut_nodes_.size()) return false;
  lo = nodes_[cut_nodes_[rank]].lo;
  hi = lo + nodes_[cut_nodes_[rank]].m;
  return true;
}
void DeviceHSS::compress_callbacks_user_random(const host_sample_t& sample, const host_elem_t& elem) {
  if (!o_.user_random) throw std::logic_error("compress_callbacks_user_random: the engine was not built with user_random");
  CallbackSource s(sample, elem);
  compress(s);
}
void DeviceHSS::compress_callbacks(const host_mult_t& mult, const host_elem_t& elem) {
  CallbackSource s(mult, elem);
  compress(s);
}

}  // namespace HSS
}  // namespace strumpack
