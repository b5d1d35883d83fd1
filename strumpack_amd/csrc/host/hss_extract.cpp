// DeviceHSS: sub-block extraction by tree traversal (HSSMatrix::extract / extract_add, HSS/HSSMatrix.extract.hpp:36-104;
// what a sparse HSS front's extend-add calls on its children, sparse/fronts/FrontHSS.cpp extract_CB_sub_matrix).
#include "hss_engine_internal.hpp"

namespace strumpack {
namespace HSS {

// device-resident node table + inverse permutations of the bases, built on first use after a compression
void DeviceHSS::ensure_dev_tree() {
  if (dev_tree_) return;
  const size_t nn = nodes_.size();
  std::vector<hssk_tree_node> t(nn);
  size_t nperm = 0;
  for (auto& nd : nodes_) if (nd.lvl > 0) nperm += (size_t)nd.mU + nd.mV;
  std::vector<int> ip(std::max<size_t>(nperm, 1));
  int* dip = persist_->ints(std::max<size_t>(nperm, 1));
  size_t off = 0;
  ex_rmax_ = 1;
  ex_depth_ = 1;
  for (size_t i = 0; i < nn; i++) {
    const Node& nd = nodes_[i];
    hssk_tree_node& d = t[i];
    d = hssk_tree_node{};
    d.lo = nd.lo; d.m = nd.m; d.c0 = nd.c0; d.c1 = nd.c1;
    d.D = nd.D; d.B01 = nd.B01; d.B10 = nd.B10;
    ex_depth_ = std::max(ex_depth_, nd.lvl + 1);
    if (nd.lvl == 0) continue;
    if ((int)nd.hpermU.size() != nd.mU || (int)nd.hpermV.size() != nd.mV) throw std::logic_error("extract: the bases' permutations are not available on the host");
    d.rU = nd.rU; d.rV = nd.rV; d.mU = nd.mU; d.mV = nd.mV; d.XU = nd.XU; d.XV = nd.XV;
    ex_rmax_ = std::max(ex_rmax_, std::max(nd.rU, nd.rV));
    for (int k = 0; k < nd.mU; k++) ip[off + nd.hpermU[k]] = k;
    d.ipermU = dip + off; off += nd.mU;
    for (int k = 0; k < nd.mV; k++) ip[off + nd.hpermV[k]] = k;
    d.ipermV = dip + off; off += nd.mV;
  }
  if (nperm) ck(hssk_memcpy_h2d(ctx_, dip, ip.data(), (long long)(sizeof(int) * nperm)));
  hssk_tree_node* dt = (hssk_tree_node*)persist_->alloc(sizeof(hssk_tree_node) * nn);
  ck(hssk_memcpy_h2d(ctx_, dt, t.data(), (long long)(sizeof(hssk_tree_node) * nn)));
  dev_tree_ = dt;
}

bool DeviceHSS::extract_by_traversal_ok(int node, long long ni, long long nj) const {
  if (o_.world != 1 || !is_compressed()) return false;
  int rm = 1;
  for (int i = node, e = subtree_end(node); i < e; i++) rm = std::max(rm, std::max(nodes_[i].rU, nodes_[i].rV));
  if (rm > 256) return false;
  // one product with the whole matrix per column costs about N (leaf + 4 r) flops; the traversal r^2 per entry
  const double N = nodes_[node].m;
  return (double)ni * rm * rm <= 64.0 * std::max(N, 1.0) * 8.0 && ni * nj <= (1LL << 31);
}

void DeviceHSS::extract_blocks(int node, int nb, const int* rows, const int* roff, const int* cols, const int* coff, double* const* out,
                               const int* ldo, bool on_device, bool add) {
  OpGuard op_guard(op_mu_);
  ensure_ready("extract");
  if (nb <= 0) return;
  if (o_.world != 1) throw std::logic_error("extract: tree traversal needs a single-process matrix");
  if (node < 0 || node >= (int)nodes_.size()) throw std::invalid_argument("extract: no such node");
  ensure_dev_tree();
  if (ex_rmax_ > 256) throw std::invalid_argument("extract: ranks beyond 256 are not supported by the traversal");
  const int lo0 = nodes_[node].lo, mN = nodes_[node].m;
  const int nr = roff[nb], nc = coff[nb];
  std::vector<int> hr(std::max(nr, 1)), hc(std::max(nc, 1));
  for (int i = 0; i < nr; i++) { if (rows[i] < 0 || rows[i] >= mN) throw std::invalid_argument("extract: row index out of range"); hr[i] = rows[i] + lo0; }
  for (int j = 0; j < nc; j++) { if (cols[j] < 0 || cols[j] >= mN) throw std::invalid_argument("extract: column index out of range"); hc[j] = cols[j] + lo0; }
  std::vector<hssk_extract_block> hb(nb);
  std::vector<long long> po(nb + 1, 0);
  for (int b = 0; b < nb; b++) po[b + 1] = po[b] + (long long)(roff[b + 1] - roff[b]) * (coff[b + 1] - coff[b]);
  const long long npairs = po[nb];
  if (npairs == 0) return;
  Arena& tmp = *tmp_;
  tmp.rewind();
  double* dout = nullptr;   // host outputs: one device buffer for all blocks
  if (!on_device) dout = tmp.dbl((size_t)npairs);
  for (int b = 0; b < nb; b++) {
    const int ni = roff[b + 1] - roff[b], nj = coff[b + 1] - coff[b];
    hb[b] = hssk_extract_block{roff[b], ni, coff[b], nj, on_device ? out[b] : dout + po[b], on_device ? ldo[b] : std::max(ni, 1)};
  }
  int* dr = tmp.ints((size_t)std::max(nr, 1));
  int* dc = tmp.ints((size_t)std::max(nc, 1));
  auto* db = (hssk_extract_block*)tmp.alloc(sizeof(hssk_extract_block) * nb);
  auto* dpo = (long long*)tmp.alloc(sizeof(long long) * (nb + 1));
  double* work = tmp.dbl((size_t)(nr + nc) * ex_depth_ * ex_rmax_);
  if (nr) ck(hssk_memcpy_h2d(ctx_, dr, hr.data(), (long long)sizeof(int) * nr));
  if (nc) ck(hssk_memcpy_h2d(ctx_, dc, hc.data(), (long long)sizeof(int) * nc));
  ck(hssk_memcpy_h2d(ctx_, db, hb.data(), (long long)(sizeof(hssk_extract_block) * nb)));
  ck(hssk_memcpy_h2d(ctx_, dpo, po.data(), (long long)(sizeof(long long) * (nb + 1))));
  ck(hssk_hss_extract(ctx_, dev_tree_, node, ex_rmax_, ex_depth_, dr, nr, dc, nc, db, dpo, nb, npairs, (on_device && add) ? 1 : 0, work));
  if (!on_device) {
    std::vector<double> h((size_t)npairs);
    ck(hssk_memcpy_d2h(ctx_, h.data(), dout, (long long)(sizeof(double) * npairs)));
    for (int b = 0; b < nb; b++) {
      const int ni = roff[b + 1] - roff[b], nj = coff[b + 1] - coff[b];
      for (int j = 0; j < nj; j++)
        for (int i = 0; i < ni; i++) {
          double& o = out[b][(size_t)i + (size_t)j * ldo[b]];
          const double v = h[(size_t)po[b] + (size_t)i + (size_t)j * ni];
          o = add ? o + v : v;
        }
    }
  } else {
    ck(hssk_sync(ctx_));
  }
}

}  // namespace HSS
}  // namespace strumpack
