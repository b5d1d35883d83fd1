// DeviceHSS: serialization of the compressed representation.
#include "hss_engine_internal.hpp"

namespace strumpack {
namespace HSS {

// ---------------------------------------------------------------------------------------------
// save / load  (HSSMatrix::write / read, HSS/HSSMatrix.cpp:438-510).  Own, self-describing binary layout: the
// reference writes raw object images (sizeof(DenseMatrix) including its vtable and data pointers), which only the
// binary that wrote them can read back, so there is no common file format to follow.
//   "HSSAMD01" | int n, nnodes | per node (pre-order): 13 ints {lo, m, lvl, height, c0, c1, parent, Ustate, Vstate,
//   rU, rV, mU, mV}, then nine blocks, each as (int64 count, payload): D, B01, B10, XU, permU, Ir, XV, permV, Ic
//   (doubles / ints, little-endian, column-major; count 0 when the node does not have the block).
// ---------------------------------------------------------------------------------------------
namespace {
template <class T> void put(std::ostream& os, const T* p, size_t n) {
  const long long c = (long long)n;
  os.write((const char*)&c, sizeof(c));
  if (n) os.write((const char*)p, sizeof(T) * n);
}
// bytes left in the stream behind the read position (-1: not seekable, no bound available)
long long bytes_left(std::istream& is) {
  const std::istream::pos_type here = is.tellg();
  if (here == std::istream::pos_type(-1)) return -1;
  is.seekg(0, std::ios::end);
  const std::istream::pos_type end = is.tellg();
  is.seekg(here);
  if (!is || end == std::istream::pos_type(-1)) { is.clear(); is.seekg(here); return -1; }
  return (long long)(end - here);
}
// a block: (int64 count, payload).  The count is checked against what the stream still holds (and against `cap`, what the
// header allows for this block) BEFORE anything is allocated: a corrupt or foreign file must end as "corrupt HSS file",
// not as a multi-terabyte allocation
template <class T> std::vector<T> get(std::istream& is, long long cap = (1LL << 40)) {
  long long c = -1;
  is.read((char*)&c, sizeof(c));
  if (!is || c < 0 || c > cap) throw std::runtime_error("HSS file is truncated or corrupt");
  const long long left = bytes_left(is);
  if (left >= 0 && c > left / (long long)sizeof(T)) throw std::runtime_error("HSS file is truncated or corrupt");
  std::vector<T> v((size_t)c);
  if (c) is.read((char*)v.data(), sizeof(T) * (size_t)c);
  if (!is) throw std::runtime_error("HSS file is truncated");
  return v;
}
}  // namespace

void DeviceHSS::save(std::ostream& os) const {
  if (o_.world > 1) throw std::invalid_argument("write: not supported for a matrix sharded over several processes");
  ck(hssk_sync(ctx_));
  os.write("HSSAMD01", 8);
  const int hdr[2] = {n_, (int)nodes_.size()};
  os.write((const char*)hdr, sizeof(hdr));
  auto dump_d = [&](const double* d, size_t cnt) {
    std::vector<double> buf(d ? cnt : 0);
    if (!buf.empty()) ck(hssk_memcpy_d2h(ctx_, buf.data(), d, (long long)(sizeof(double) * buf.size())));
    put(os, buf.data(), buf.size());
  };
  auto dump_i = [&](const int* d, size_t cnt) {
    std::vector<int> buf(d ? cnt : 0);
    if (!buf.empty()) ck(hssk_memcpy_d2h(ctx_, buf.data(), d, (long long)(sizeof(int) * buf.size())));
    put(os, buf.data(), buf.size());
  };
  for (const Node& nd : nodes_) {
    const int f[13] = {nd.lo, nd.m, nd.lvl, nd.height, nd.c0, nd.c1, nd.parent, nd.Ustate, nd.Vstate, nd.rU, nd.rV, nd.mU, nd.mV};
    os.write((const char*)f, sizeof(f));
    const bool basis = nd.lvl > 0 && nd.compressed();
    dump_d(nd.leaf() ? nd.D : nullptr, (size_t)nd.m * nd.m);
    dump_d(nd.leaf() ? nullptr : nd.B01, nd.leaf() ? 0 : (size_t)nodes_[nd.c0].rU * nodes_[nd.c1].rV);
    dump_d(nd.leaf() ? nullptr : nd.B10, nd.leaf() ? 0 : (size_t)nodes_[nd.c1].rU * nodes_[nd.c0].rV);
    dump_d(basis ? nd.XU : nullptr, (size_t)nd.rU * std::max(nd.mU - nd.rU, 0));
    dump_i(basis ? nd.permU : nullptr, nd.mU);
    put(os, nd.Ir.data(), basis ? nd.Ir.size() : 0);
    dump_d(basis ? nd.XV : nullptr, (size_t)nd.rV * std::max(nd.mV - nd.rV, 0));
    dump_i(basis ? nd.permV : nullptr, nd.mV);
    put(os, nd.Ic.data(), basis ? nd.Ic.size() : 0);
  }
  if (!os) throw std::runtime_error("write: I/O error");
}

std::unique_ptr<DeviceHSS> DeviceHSS::load(std::istream& is, const EngineOptions& opts) {
  char magic[8];
  is.read(magic, 8);
  if (!is || std::memcmp(magic, "HSSAMD01", 8)) throw std::runtime_error("not an HSS matrix file of this library");
  int hdr[2] = {0, 0};
  is.read((char*)hdr, sizeof(hdr));
  const int n = hdr[0], nn = hdr[1];
  if (!is || n < 0 || nn < 1) throw std::runtime_error("corrupt HSS file header");
  struct Rec { int f[13]; std::vector<double> D, B01, B10, XU, XV; std::vector<int> pU, pV, Ir, Ic; };
  std::vector<Rec> recs(nn);
  for (auto& r : recs) {
    is.read((char*)r.f, sizeof(r.f));
    if (!is) throw std::runtime_error("HSS file is truncated");
    // every block is bounded by the header's matrix size: doubles by n^2 (a node's D, B, X), ints by n
    const long long cap_d = (long long)n * (long long)n, cap_i = n;
    r.D = get<double>(is, cap_d); r.B01 = get<double>(is, cap_d); r.B10 = get<double>(is, cap_d);
    r.XU = get<double>(is, cap_d); r.pU = get<int>(is, cap_i); r.Ir = get<int>(is, cap_i);
    r.XV = get<double>(is, cap_d); r.pV = get<int>(is, cap_i); r.Ic = get<int>(is, cap_i);
  }
  // cluster tree from the node table (children follow their parent in pre-order)
  std::function<structured::ClusterTree(int)> tree_of = [&](int i) {
    if (i < 0 || i >= nn) throw std::runtime_error("corrupt HSS file (tree)");
    structured::ClusterTree t(recs[i].f[1]);
    if (recs[i].f[4] >= 0) {
      if (recs[i].f[4] <= i || recs[i].f[5] <= i) throw std::runtime_error("corrupt HSS file (tree)");
      t.c.resize(2);
      t.c[0] = tree_of(recs[i].f[4]);
      t.c[1] = tree_of(recs[i].f[5]);
    }
    return t;
  };
  structured::ClusterTree tree = tree_of(0);
  if (tree.size != n) throw std::runtime_error("corrupt HSS file (size)");
  std::unique_ptr<DeviceHSS> H(new DeviceHSS(n, opts, &tree));
  if ((int)H->nodes_.size() != nn) throw std::runtime_error("corrupt HSS file (node count)");
  Arena& P = *H->persist_;
  auto up_d = [&](const std::vector<double>& v) -> double* {
    double* d = P.dbl(std::max<size_t>(v.size(), 1));
    if (!v.empty()) ck(hssk_memcpy_h2d(H->ctx_, d, v.data(), (long long)(sizeof(double) * v.size())));
    return d;
  };
  auto up_i = [&](const std::vector<int>& v) -> int* {
    int* d = P.ints(std::max<size_t>(v.size(), 1));
    if (!v.empty()) ck(hssk_memcpy_h2d(H->ctx_, d, v.data(), (long long)(sizeof(int) * v.size())));
    return d;
  };
  for (int i = 0; i < nn; i++) {
    Node& nd = H->nodes_[i];
    const Rec& r = recs[i];
    if (nd.lo != r.f[0] || nd.m != r.f[1] || nd.c0 != r.f[4] || nd.c1 != r.f[5]) throw std::runtime_error("corrupt HSS file (node table)");
    nd.Ustate = r.f[7]; nd.Vstate = r.f[8]; nd.rU = r.f[9]; nd.rV = r.f[10]; nd.mU = r.f[11]; nd.mV = r.f[12];
    // every block is checked against the node table before it reaches the device: a truncated-but-parseable or foreign
    // file must fail here, not as an out-of-bounds read in the first mult / factor
    if (nd.Ustate < 0 || nd.Ustate > 2 || nd.Vstate < 0 || nd.Vstate > 2) throw std::runtime_error("corrupt HSS file (node state)");
    if (nd.rU < 0 || nd.rV < 0 || nd.mU < 0 || nd.mV < 0 || nd.rU > nd.mU || nd.rV > nd.mV) throw std::runtime_error("corrupt HSS file (ranks)");
    const bool used = nd.compressed() || nd.lvl == 0;   // blocks exist once the node has been processed
    if (nd.leaf()) {
      if (r.D.size() != (used || !r.D.empty() ? (size_t)nd.m * nd.m : 0)) throw std::runtime_error("corrupt HSS file (leaf block size)");
      if (nd.lvl > 0 && nd.compressed() && (nd.mU != nd.m || nd.mV != nd.m)) throw std::runtime_error("corrupt HSS file (leaf basis rows)");
      if (!r.D.empty()) nd.D = up_d(r.D);
    } else {
      const Rec &a = recs[nd.c0], &b = recs[nd.c1];   // (rU, rV) of the children: f[9], f[10]
      if (used) {
        if (r.B01.size() != (size_t)a.f[9] * b.f[10] || r.B10.size() != (size_t)b.f[9] * a.f[10])
          throw std::runtime_error("corrupt HSS file (coupling block sizes)");
        if (nd.lvl > 0 && (nd.mU != a.f[9] + b.f[9] || nd.mV != a.f[10] + b.f[10])) throw std::runtime_error("corrupt HSS file (basis rows)");
      } else if (!r.B01.empty() || !r.B10.empty()) throw std::runtime_error("corrupt HSS file (coupling blocks of an untouched node)");
      nd.B01 = up_d(r.B01); nd.B10 = up_d(r.B10);
    }
    if (nd.lvl > 0 && nd.compressed()) {
      if ((int)r.pU.size() != nd.mU || (int)r.pV.size() != nd.mV || (int)r.Ir.size() != nd.rU || (int)r.Ic.size() != nd.rV ||
          r.XU.size() != (size_t)nd.rU * std::max(nd.mU - nd.rU, 0) || r.XV.size() != (size_t)nd.rV * std::max(nd.mV - nd.rV, 0))
        throw std::runtime_error("corrupt HSS file (basis sizes)");
      auto is_perm = [](const std::vector<int>& p) {
        std::vector<char> seen(p.size(), 0);
        for (int v : p) { if (v < 0 || v >= (int)p.size() || seen[v]) return false; seen[v] = 1; }
        return true;
      };
      if (!is_perm(r.pU) || !is_perm(r.pV)) throw std::runtime_error("corrupt HSS file (permutation)");
      for (int v : r.Ir) if (v < 0 || v >= n) throw std::runtime_error("corrupt HSS file (row index set)");
      for (int v : r.Ic) if (v < 0 || v >= n) throw std::runtime_error("corrupt HSS file (column index set)");
      nd.XU = up_d(r.XU); nd.permU = up_i(r.pU); nd.hpermU = r.pU; nd.Ir = r.Ir; nd.dIr = up_i(r.Ir);
      nd.XV = up_d(r.XV); nd.permV = up_i(r.pV); nd.hpermV = r.pV; nd.Ic = r.Ic; nd.dIc = up_i(r.Ic);
    }
  }
  ck(hssk_sync(H->ctx_));
  return H;
}

}  // namespace HSS
}  // namespace strumpack
