// DeviceHSS: sub-tree products and the Schur complement of the (0,0) block (HSSMatrix.Schur.hpp).
#include "hss_engine_internal.hpp"

namespace strumpack {
namespace HSS {

// ---------------------------------------------------------------------------------------------
// Sub-tree helpers and the Schur complement of the (0,0) block (HSSMatrix.Schur.hpp)
// ---------------------------------------------------------------------------------------------
int DeviceHSS::subtree_end(int sr) const {
  int id = sr;
  while (!nodes_[id].leaf()) id = nodes_[id].c1;
  return id + 1;
}

std::vector<std::vector<int>> DeviceHSS::sublists(const std::vector<std::vector<int>>& lists, int sr) const {
  const int end = subtree_end(sr);
  std::vector<std::vector<int>> out;
  for (auto& l : lists) {
    std::vector<int> f;
    for (int id : l) if (id >= sr && id < end) f.push_back(id);
    if (!f.empty()) out.push_back(std::move(f));
  }
  return out;
}

void DeviceHSS::mult_child(int c, char trans, int nrhs, const double* x, long long ldx, double* y, long long ldy,
                           bool on_device) {
  OpGuard op_guard(op_mu_);
  if (nodes_[0].leaf()) throw std::logic_error("mult_child: the root is a leaf");
  mult_sub(c == 0 ? nodes_[0].c0 : nodes_[0].c1, trans, nrhs, x, ldx, y, ldy, on_device, 0.0);
}

void DeviceHSS::mult_node(int node, char trans, int nrhs, const double* x, long long ldx, double* y, long long ldy,
                          bool on_device, double beta) {
  OpGuard op_guard(op_mu_);
  if (node < 0 || node >= (int)nodes_.size()) throw std::invalid_argument("mult_node: no such node");
  mult_sub(node, trans, nrhs, x, ldx, y, ldy, on_device, beta);
}

void DeviceHSS::basis_up(int sr, bool useU, const double* dA, long long lda, int c, double* dOut, int ldout, Arena& wk) {
  if (c <= 0) return;
  if (lda > 0x7fffffffLL) throw std::invalid_argument("basis_up: leading dimension too large");
  const int lo0 = nodes_[sr].lo, end = subtree_end(sr);
  auto rk = [&](const Node& nd) { return useU ? nd.rU : nd.rV; };
  auto rows = [&](const Node& nd) { return nd.leaf() ? nd.m : rk(nodes_[nd.c0]) + rk(nodes_[nd.c1]); };
  std::vector<double*> cat(nodes_.size(), nullptr);
  for (int id = sr; id < end; id++)
    if (!nodes_[id].leaf()) cat[id] = wk.dbl((size_t)std::max(rows(nodes_[id]), 1) * c);
  for (auto& ids : sublists(by_height_, sr)) {
    std::vector<hssk_rowgather_desc> g;
    std::vector<hssk_gemm_desc> mm;
    for (int id : ids) {
      const Node& nd = nodes_[id];
      const int m = rows(nd), r = rk(nd);
      if (r == 0) continue;
      const int* perm = useU ? nd.permU : nd.permV;
      const double* X = useU ? nd.XU : nd.XV;
      const double* src = nd.leaf() ? dA + (nd.lo - lo0) : cat[id];
      const int lds = nd.leaf() ? (int)lda : std::max(m, 1);
      double* dst = dOut;
      int ldd = ldout;
      if (id != sr) {
        const Node& pa = nodes_[nd.parent];
        dst = cat[nd.parent] + (id == pa.c0 ? 0 : rk(nodes_[pa.c0]));
        ldd = std::max(rows(pa), 1);
      }
      g.push_back(hssk_rowgather_desc{src, dst, perm, r, c, lds, ldd, 0, 0});
      if (m > r) {
        double* Tm = wk.dbl((size_t)(m - r) * c);
        g.push_back(hssk_rowgather_desc{src, Tm, perm + r, m - r, c, lds, m - r, 0, 0});
        mm.push_back(hssk_gemm_desc{X, Tm, dst, r, c, m - r, r, m - r, ldd, 0, 0, 1.0, 1.0});
      }
    }
    if (!g.empty()) ck(hssk_gather_rows(ctx_, g.data(), (int)g.size()));
    if (!mm.empty()) ck(hssk_gemm_vbatched(ctx_, mm.data(), (int)mm.size()));
  }
}

void DeviceHSS::basis_down(int sr, bool useU, const double* dIn, int ldin, int c, double* dOut, long long ldo, Arena& wk,
                           bool recurse) {
  if (c <= 0) return;
  if (ldo > 0x7fffffffLL) throw std::invalid_argument("basis_down: leading dimension too large");
  const int lo0 = nodes_[sr].lo;
  auto rk = [&](const Node& nd) { return useU ? nd.rU : nd.rV; };
  auto rows = [&](const Node& nd) { return nd.leaf() ? nd.m : rk(nodes_[nd.c0]) + rk(nodes_[nd.c1]); };
  std::vector<double*> t(nodes_.size(), nullptr);
  std::vector<std::vector<int>> lists;
  if (recurse) lists = sublists(by_depth_, sr);
  else lists.push_back(std::vector<int>{sr});
  for (auto& ids : lists) {
    std::vector<hssk_rowgather_desc> sc;
    std::vector<hssk_gemm_desc> mm, zero;
    for (int id : ids) {
      const Node& nd = nodes_[id];
      const int mo = rows(nd), r = rk(nd);
      if (mo == 0) continue;
      double* out;
      int ld;
      if (nd.leaf() || !recurse) {
        out = dOut + (recurse ? nd.lo - lo0 : 0);
        ld = (int)ldo;
      } else {
        out = t[id] = wk.dbl((size_t)mo * c);
        ld = mo;
      }
      const double* in = dIn;
      int ldi = ldin;
      if (id != sr) {
        const Node& pa = nodes_[nd.parent];
        in = t[nd.parent] + (id == pa.c0 ? 0 : rk(nodes_[pa.c0]));
        ldi = std::max(rows(pa), 1);
      }
      if (r == 0) {   // no basis: this block row of the product is zero (Schur.hpp:262, :269)
        zero.push_back(hssk_gemm_desc{out, out, out, mo, c, 0, ld, 1, ld, 0, 0, 0.0, 0.0});
        continue;
      }
      const int* perm = useU ? nd.permU : nd.permV;
      const double* X = useU ? nd.XU : nd.XV;
      // out(perm[:r]) = in ; out(perm[r:]) = X^T in      (HSSBasisID::apply)
      sc.push_back(hssk_rowgather_desc{in, out, perm, r, c, ldi, ld, 1, 0});
      if (mo > r) {
        double* E2 = wk.dbl((size_t)(mo - r) * c);
        mm.push_back(hssk_gemm_desc{X, in, E2, mo - r, c, r, r, ldi, mo - r, 1, 0, 1.0, 0.0});
        sc.push_back(hssk_rowgather_desc{E2, out, perm + r, mo - r, c, mo - r, ld, 1, 0});
      }
    }
    if (!zero.empty()) ck(hssk_gemm_vbatched(ctx_, zero.data(), (int)zero.size()));
    if (!mm.empty()) ck(hssk_gemm_vbatched(ctx_, mm.data(), (int)mm.size()));
    if (!sc.empty()) ck(hssk_gather_rows(ctx_, sc.data(), (int)sc.size()));
  }
}

DeviceHSS::SchurDims DeviceHSS::schur_dims() const {
  SchurDims d;
  if (nodes_[0].leaf()) return d;
  const Node &a = nodes_[nodes_[0].c0], &b = nodes_[nodes_[0].c1];
  d.n0 = a.m; d.n1 = b.m;
  d.rV0 = a.rV; d.rU0 = a.rU; d.rV1 = b.rV; d.rU1 = b.rU;
  d.mu0 = a.leaf() ? a.m : nodes_[a.c0].rU + nodes_[a.c1].rU;
  return d;
}

void DeviceHSS::schur_update(double* Theta, long long ldt, double* DUB01, long long ldd, double* Phi, long long ldp,
                             double* Vhat, long long ldv) {
  OpGuard op_guard(op_mu_);
  ensure_ready("Schur_update");
  if (nodes_[0].leaf()) return;    // Schur.hpp:42
  if (!partial_factored_) throw std::logic_error("Schur_update: partial_factor() has not been called");
  const Node& root = nodes_[0];
  const Node &a = nodes_[root.c0], &b = nodes_[root.c1];
  const SchurDims d = schur_dims();
  ck(hssk_sync(ctx_));
  schur_->rewind();
  Arena wk;
  auto L = [](int x) { return std::max(x, 1); };
  sDUB01_ = schur_->dbl((size_t)L(d.mu0) * L(d.rV1));
  sTheta_ = schur_->dbl((size_t)L(d.n1) * L(d.rV0));
  sPhi_ = schur_->dbl((size_t)L(d.n1) * L(d.mu0));
  sVtDUB01_ = schur_->dbl((size_t)L(d.rV0) * L(d.rV1));
  sW_ = schur_->dbl((size_t)L(d.rU1) * L(d.rV1));
  // DUB01 = D00^{-1} (U0 B01)                                         (Schur.hpp:46-48)
  basis_down(root.c0, true, root.B01, L(d.rU0), d.rV1, sDUB01_, L(d.mu0), wk, false);
  if (d.mu0 && d.rV1) {
    hssk_lusolve_desc ls{a.LU, a.piv, sDUB01_, d.mu0, d.rV1, d.mu0, L(d.mu0)};
    ck(hssk_getrs_vbatched(ctx_, &ls, 1));
  }
  // Theta = U1big B10 ; Phi = V1big DUB01^T                          (Schur.hpp:52-58)
  basis_down(root.c1, true, root.B10, L(d.rU1), d.rV0, sTheta_, L(d.n1), wk);
  double* Dt = wk.dbl((size_t)L(d.rV1) * L(d.mu0));
  if (d.mu0 && d.rV1) {
    hssk_transpose_desc tr{sDUB01_, Dt, d.mu0, d.rV1, L(d.mu0), L(d.rV1)};
    ck(hssk_transpose(ctx_, &tr, 1));
  }
  basis_down(root.c1, false, Dt, L(d.rV1), d.mu0, sPhi_, L(d.n1), wk);
  // small products reused by every Schur_product_*: Vhat^T DUB01 (rV0 x rV1) and W = B10 Vhat^T DUB01 (rU1 x rV1)
  std::vector<hssk_gemm_desc> g;
  g.push_back(hssk_gemm_desc{a.Vt0, sDUB01_, sVtDUB01_, d.rV0, d.rV1, d.mu0, L(d.mu0), L(d.mu0), L(d.rV0), 1, 0, 1.0, 0.0});
  ck(hssk_gemm_vbatched(ctx_, g.data(), 1));
  g[0] = hssk_gemm_desc{root.B10, sVtDUB01_, sW_, d.rU1, d.rV1, d.rV0, L(d.rU1), L(d.rV0), L(d.rU1), 0, 0, 1.0, 0.0};
  ck(hssk_gemm_vbatched(ctx_, g.data(), 1));
  ck(hssk_sync(ctx_));
  schur_ready_ = true;
  auto get = [&](double* h, long long ldh, const double* dsrc, int rows, int cols) {
    if (h && rows > 0 && cols > 0)
      ck(hssk_memcpy2d_d2h(ctx_, h, sizeof(double) * ldh, dsrc, sizeof(double) * rows, sizeof(double) * rows, cols));
  };
  get(Theta, ldt, sTheta_, d.n1, d.rV0);
  get(DUB01, ldd, sDUB01_, d.mu0, d.rV1);
  get(Phi, ldp, sPhi_, d.n1, d.mu0);
  get(Vhat, ldv, a.Vt0, d.mu0, d.rV0);
}

void DeviceHSS::schur_product_direct(int c, const double* R, long long ldr, double* Sr, long long ldsr, double* Sc,
                                     long long ldsc, bool on_device) {
  OpGuard op_guard(op_mu_);
  if (!schur_ready_) throw std::logic_error("Schur_product_direct: Schur_update() has not been called");
  if (c <= 0) return;
  const Node& root = nodes_[0];
  const Node& a = nodes_[root.c0];
  const SchurDims d = schur_dims();
  Arena wk;
  auto L = [](int x) { return std::max(x, 1); };
  const int n1 = d.n1;
  const double* dR = R;
  double *dSr = Sr, *dSc = Sc;
  long long lr = ldr, lsr = ldsr, lsc = ldsc;
  if (!on_device) {
    double* b = wk.dbl((size_t)n1 * c);
    ck(hssk_memcpy2d_h2d(ctx_, b, sizeof(double) * n1, R, sizeof(double) * ldr, sizeof(double) * n1, c));
    dR = b; dSr = wk.dbl((size_t)n1 * c); dSc = wk.dbl((size_t)n1 * c);
    lr = lsr = lsc = n1;
  }
  if (lsr > 0x7fffffffLL || lsc > 0x7fffffffLL) throw std::invalid_argument("Schur_product_direct: leading dimension too large");
  // Sr = H11 R, Sc = H11^T R; the basis products V1big^T R / U1big^T R are the forward halves of those applies
  mult_sub(root.c1, 'N', c, dR, lr, dSr, lsr, true, 0.0);
  mult_sub(root.c1, 'T', c, dR, lr, dSc, lsc, true, 0.0);
  double* V1tR = wk.dbl((size_t)L(d.rV1) * c);
  double* U1tR = wk.dbl((size_t)L(d.rU1) * c);
  basis_up(root.c1, false, dR, lr, c, V1tR, L(d.rV1), wk);
  basis_up(root.c1, true, dR, lr, c, U1tR, L(d.rU1), wk);
  // Sr -= Theta (Vhat^T DUB01) (V1big^T R) ;  Sc -= Phi Vhat B10^T (U1big^T R)        (Schur.hpp:60-71)
  double* t1 = wk.dbl((size_t)L(d.rV0) * c);
  double* t2 = wk.dbl((size_t)L(d.rV0) * c);
  double* t3 = wk.dbl((size_t)L(d.mu0) * c);
  std::vector<hssk_gemm_desc> g(2);
  g[0] = hssk_gemm_desc{sVtDUB01_, V1tR, t1, d.rV0, c, d.rV1, L(d.rV0), L(d.rV1), L(d.rV0), 0, 0, 1.0, 0.0};
  g[1] = hssk_gemm_desc{root.B10, U1tR, t2, d.rV0, c, d.rU1, L(d.rU1), L(d.rU1), L(d.rV0), 1, 0, 1.0, 0.0};
  ck(hssk_gemm_vbatched(ctx_, g.data(), 2));
  g[0] = hssk_gemm_desc{a.Vt0, t2, t3, d.mu0, c, d.rV0, L(d.mu0), L(d.rV0), L(d.mu0), 0, 0, 1.0, 0.0};
  ck(hssk_gemm_vbatched(ctx_, g.data(), 1));
  g[0] = hssk_gemm_desc{sTheta_, t1, dSr, n1, c, d.rV0, L(n1), L(d.rV0), (int)lsr, 0, 0, -1.0, 1.0};
  g[1] = hssk_gemm_desc{sPhi_, t3, dSc, n1, c, d.mu0, L(n1), L(d.mu0), (int)lsc, 0, 0, -1.0, 1.0};
  ck(hssk_gemm_vbatched(ctx_, g.data(), 2));
  if (!on_device) {
    ck(hssk_memcpy2d_d2h(ctx_, Sr, sizeof(double) * ldsr, dSr, sizeof(double) * n1, sizeof(double) * n1, c));
    ck(hssk_memcpy2d_d2h(ctx_, Sc, sizeof(double) * ldsc, dSc, sizeof(double) * n1, sizeof(double) * n1, c));
  }
  ck(hssk_sync(ctx_));
}

void DeviceHSS::schur_product_indirect(int c, const double* R0, long long ldr0, const double* R1, long long ldr1,
                                       const double* Sr1, long long ldsr1, const double* Sc1, long long ldsc1,
                                       double* Sr, long long ldsr, double* Sc, long long ldsc, bool on_device) {
  OpGuard op_guard(op_mu_);
  if (nodes_[0].leaf()) return;   // Schur.hpp:158
  if (!schur_ready_) throw std::logic_error("Schur_product_indirect: Schur_update() has not been called");
  if (c <= 0) return;
  const Node& root = nodes_[0];
  const SchurDims d = schur_dims();
  Arena wk;
  auto L = [](int x) { return std::max(x, 1); };
  const int n0 = d.n0, n1 = d.n1;
  const double *dR0 = R0, *dR1 = R1;
  double *dSr = Sr, *dSc = Sc;
  long long l0 = ldr0, l1 = ldr1, lsr = ldsr, lsc = ldsc;
  auto up = [&](const double* h, long long ldh, int rows) {
    double* b = wk.dbl((size_t)L(rows) * c);
    if (rows) ck(hssk_memcpy2d_h2d(ctx_, b, sizeof(double) * rows, h, sizeof(double) * ldh, sizeof(double) * rows, c));
    return b;
  };
  if (!on_device) {
    dR0 = up(R0, ldr0, n0); dR1 = up(R1, ldr1, n1);
    dSr = up(Sr1, ldsr1, n1); dSc = up(Sc1, ldsc1, n1);
    l0 = n0; l1 = lsr = lsc = n1;
  } else {
    // start from Sr1 / Sc1
    if (Sr != Sr1) { hssk_rowgather_desc cp{Sr1, Sr, nullptr, n1, c, (int)ldsr1, (int)ldsr, 0, 0}; ck(hssk_gather_rows(ctx_, &cp, 1)); }
    if (Sc != Sc1) { hssk_rowgather_desc cp{Sc1, Sc, nullptr, n1, c, (int)ldsc1, (int)ldsc, 0, 0}; ck(hssk_gather_rows(ctx_, &cp, 1)); }
  }
  double* V0tR0 = wk.dbl((size_t)L(d.rV0) * c);
  double* U0tR0 = wk.dbl((size_t)L(d.rU0) * c);
  double* V1tR1 = wk.dbl((size_t)L(d.rV1) * c);
  double* U1tR1 = wk.dbl((size_t)L(d.rU1) * c);
  basis_up(root.c0, false, dR0, l0, c, V0tR0, L(d.rV0), wk);
  basis_up(root.c0, true, dR0, l0, c, U0tR0, L(d.rU0), wk);
  basis_up(root.c1, false, dR1, l1, c, V1tR1, L(d.rV1), wk);
  basis_up(root.c1, true, dR1, l1, c, U1tR1, L(d.rU1), wk);
  // P = -(B10 V0big^T R0 + W V1big^T R1)  (rU1 x c) ; Q = -(B01^T U0big^T R0 + W^T U1big^T R1)  (rV1 x c)
  double* P = wk.dbl((size_t)L(d.rU1) * c);
  double* Q = wk.dbl((size_t)L(d.rV1) * c);
  std::vector<hssk_gemm_desc> g(2);
  g[0] = hssk_gemm_desc{root.B10, V0tR0, P, d.rU1, c, d.rV0, L(d.rU1), L(d.rV0), L(d.rU1), 0, 0, -1.0, 0.0};
  g[1] = hssk_gemm_desc{root.B01, U0tR0, Q, d.rV1, c, d.rU0, L(d.rU0), L(d.rU0), L(d.rV1), 1, 0, -1.0, 0.0};
  ck(hssk_gemm_vbatched(ctx_, g.data(), 2));
  g[0] = hssk_gemm_desc{sW_, V1tR1, P, d.rU1, c, d.rV1, L(d.rU1), L(d.rV1), L(d.rU1), 0, 0, -1.0, 1.0};
  g[1] = hssk_gemm_desc{sW_, U1tR1, Q, d.rV1, c, d.rU1, L(d.rU1), L(d.rU1), L(d.rV1), 1, 0, -1.0, 1.0};
  ck(hssk_gemm_vbatched(ctx_, g.data(), 2));
  // Sr = Sr1 + U1big P ; Sc = Sc1 + V1big Q                           (Schur.hpp:213-218)
  double* E = wk.dbl((size_t)L(n1) * c);
  basis_down(root.c1, true, P, L(d.rU1), c, E, L(n1), wk);
  { hssk_rowgather_desc ad{E, dSr, nullptr, n1, c, L(n1), (int)lsr, 0, 1}; ck(hssk_gather_rows(ctx_, &ad, 1)); }
  basis_down(root.c1, false, Q, L(d.rV1), c, E, L(n1), wk);
  { hssk_rowgather_desc ad{E, dSc, nullptr, n1, c, L(n1), (int)lsc, 0, 1}; ck(hssk_gather_rows(ctx_, &ad, 1)); }
  if (!on_device) {
    ck(hssk_memcpy2d_d2h(ctx_, Sr, sizeof(double) * ldsr, dSr, sizeof(double) * n1, sizeof(double) * n1, c));
    ck(hssk_memcpy2d_d2h(ctx_, Sc, sizeof(double) * ldsc, dSc, sizeof(double) * n1, sizeof(double) * n1, c));
  }
  ck(hssk_sync(ctx_));
}

}  // namespace HSS
}  // namespace strumpack
