// Kernel<double>::fit_HSS / predict (reference: kernel/KernelRegression.hpp:56-123) and the C interface of
// include/kernel/Kernel.h (reference: kernel/Kernel.cpp:43-180).
#include "Kernel.hpp"

#include <chrono>
#include <cstdlib>

#include "HSSMatrix.hpp"
#include "NeighborSearch.hpp"
#include "hssk.h"
#include "kernel/Kernel.h"

namespace strumpack {
namespace kernel {

namespace {
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct FitInfo {
  long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
thread_local FitInfo last_fit;
thread_local std::vector<int> last_nodes;
void ckk(int rc) { if (rc) throw std::runtime_error(hssk_last_error()); }
}  // namespace

DenseMatrix<double> Kernel<double>::fit_HSS(std::vector<double>& labels, const HSS::HSSOptions<double>& opts) {
  if (labels.size() != n()) throw std::invalid_argument("fit_HSS: one label per training point expected");
  double t0 = now();
  if (opts.verbose()) std::cout << "# starting HSS compression..." << std::endl;
  HSS::HSSMatrix<double> H(*this, opts);
  // labels to the cluster order: new label i = old label perm[i] (lapmt, forward)
  {
    std::vector<double> old(labels);
    for (std::size_t i = 0; i < labels.size(); i++) labels[i] = old[perm_[i] - 1];
  }
  double t1 = now();
  if (opts.verbose()) {
    std::cout << "# HSS compression time = " << t1 - t0 << std::endl;
    if (H.is_compressed())
      std::cout << "# created HSS matrix of dimension " << H.rows() << " x " << H.cols() << " with " << H.levels() << " levels" << std::endl
                << "# compression succeeded!" << std::endl;
    else std::cout << "# compression failed!!!" << std::endl;
    std::cout << "# rank(H) = " << H.rank() << std::endl << "# HSS memory(H) = " << H.memory() / 1e6 << " MB " << std::endl << std::endl
              << "# factorization start" << std::endl;
  }
  H.factor();
  double t2 = now();
  if (opts.verbose()) std::cout << "# factorization time = " << t2 - t1 << std::endl << "# solution start..." << std::endl;
  DenseMatrix<double> weights(n(), 1, labels.data(), n());
  H.solve(weights);
  double t3 = now();
  if (opts.verbose()) std::cout << "# solve time = " << t3 - t2 << std::endl;
  last_fit.v[0] = H.is_compressed(); last_fit.v[1] = (long long)H.levels(); last_fit.v[2] = (long long)H.rank();
  last_fit.v[3] = (long long)H.memory(); last_fit.v[4] = H.engine()->stats().d_final;
  last_nodes.assign(6 * (size_t)H.engine()->num_nodes(), 0);
  H.engine()->node_info(last_nodes.data());
  last_fit.v[5] = (long long)((t1 - t0) * 1e6); last_fit.v[6] = (long long)((t2 - t1) * 1e6); last_fit.v[7] = (long long)((t3 - t2) * 1e6);
  return weights;
}

std::vector<double> Kernel<double>::predict(const DenseMatrix<double>& test, const DenseMatrix<double>& weights) const {
  if (test.rows() != d()) throw std::invalid_argument("predict: test points have the wrong dimension");
  if (weights.rows() != n()) throw std::invalid_argument("predict: one weight per training point expected");
  const int m = int(test.cols()), dim = int(d());
  std::vector<double> prediction(m, 0.);
  if (m == 0) return prediction;
  if (device_type() < 0) {
    // a user-defined kernel function (only its virtual evaluation is known): the sums on the host
    for (int c = 0; c < m; c++) {
      double s = 0.;
      for (std::size_t r = 0; r < n(); r++) s += weights(r, 0) * eval_kernel_function(data_.ptr(0, r), test.ptr(0, c));
      prediction[c] = s;
    }
    return prediction;
  }
  int dev = 0;
  if (const char* e = std::getenv("STRUMPACK_AMD_DEVICE")) dev = std::atoi(e);
  hssk_ctx* ctx = nullptr;
  ckk(hssk_ctx_create(&ctx, dev));
  void *dX = nullptr, *dT = nullptr, *dw = nullptr, *dp = nullptr;
  try {
    auto dalloc = [](long long bytes) { void* q = hssk_malloc(bytes); if (!q) throw std::runtime_error(hssk_last_error()); return q; };
    dX = dalloc((long long)sizeof(double) * dim * n());
    dT = dalloc((long long)sizeof(double) * dim * m);
    dw = dalloc((long long)sizeof(double) * n());
    dp = dalloc((long long)sizeof(double) * m);
    ckk(hssk_memcpy2d_h2d(ctx, dX, sizeof(double) * dim, data_.data(), sizeof(double) * data_.ld(), sizeof(double) * dim, (long long)n()));
    ckk(hssk_memcpy2d_h2d(ctx, dT, sizeof(double) * dim, test.data(), sizeof(double) * test.ld(), sizeof(double) * dim, m));
    ckk(hssk_memcpy_h2d(ctx, dw, weights.data(), (long long)sizeof(double) * n()));
    hssk_kernel_spec spec{(const double*)dX, (long long)n(), dim, device_type(), degree(), width(), 0.};
    ckk(hssk_kernel_predict(ctx, &spec, (const double*)dw, (const double*)dT, m, (double*)dp));
    ckk(hssk_memcpy_d2h(ctx, prediction.data(), dp, (long long)sizeof(double) * m));
    ckk(hssk_sync(ctx));
  } catch (...) {
    hssk_free(dX); hssk_free(dT); hssk_free(dw); hssk_free(dp);
    hssk_ctx_destroy(ctx);
    throw;
  }
  hssk_free(dX); hssk_free(dT); hssk_free(dw); hssk_free(dp);
  hssk_ctx_destroy(ctx);
  return prediction;
}

const long long* last_fit_info() { return last_fit.v; }
const std::vector<int>& last_fit_nodes() { return last_nodes; }

}  // namespace kernel
}  // namespace strumpack

// ---- C interface ---------------------------------------------------------------------------------------
using namespace strumpack;
namespace {
struct KernelRegression {
  std::unique_ptr<kernel::Kernel<double>> K;
  DenseMatrix<double> training, weights;
  long long info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::vector<int> nodes;   // node table of the last fit (6 ints per node)
};
void report(const std::exception& e) { std::cerr << "Operation failed: " << e.what() << std::endl; }
}  // namespace

extern "C" {

STRUMPACKKernel STRUMPACK_create_kernel_double(int n, int d, double* train, double h, double lambda, int p, int type) {
  try {
    auto kr = new KernelRegression();
    kr->training = DenseMatrix<double>(d, n, train, d);
    switch (type) {
      case 0: kr->K.reset(new kernel::GaussKernel<double>(kr->training, h, lambda)); break;
      case 1: kr->K.reset(new kernel::LaplaceKernel<double>(kr->training, h, lambda)); break;
      case 2: kr->K.reset(new kernel::ANOVAKernel<double>(kr->training, h, lambda, p)); break;
      default: std::cout << "ERROR: Kernel type not recognized!" << std::endl;
    }
    return kr;
  } catch (const std::exception& e) { report(e); return nullptr; }
}
void STRUMPACK_destroy_kernel_double(STRUMPACKKernel K) { delete static_cast<KernelRegression*>(K); }

void STRUMPACK_kernel_fit_HSS_double(STRUMPACKKernel K, double* labels, int argc, char* argv[]) {
  try {
    auto kr = static_cast<KernelRegression*>(K);
    if (!kr || !kr->K) throw std::invalid_argument("no kernel");
    std::vector<double> vl(labels, labels + kr->K->n());
    HSS::HSSOptions<double> opts;
    opts.set_verbose(false);
    opts.set_clustering_algorithm(ClusteringAlgorithm::COBBLE);   // kernel/Kernel.cpp:82
    opts.set_from_command_line(argc, argv);
    kr->weights = kr->K->fit_HSS(vl, opts);
    std::copy(kernel::last_fit_info(), kernel::last_fit_info() + 8, kr->info);
    kr->nodes = kernel::last_fit_nodes();
  } catch (const std::exception& e) { report(e); }
}
void STRUMPACK_kernel_predict_double(STRUMPACKKernel K, int m, double* test, double* prediction) {
  try {
    auto kr = static_cast<KernelRegression*>(K);
    if (!kr || !kr->K) throw std::invalid_argument("no kernel");
    DenseMatrix<double> t(kr->K->d(), m, test, kr->K->d());
    auto pred = kr->K->predict(t, kr->weights);
    std::copy(pred.begin(), pred.end(), prediction);
  } catch (const std::exception& e) { report(e); }
}
int SPX_approximate_neighbors(int n, int d, const double* data, int iterations, int k, int* ann, double* scores) {
  try {
    DenseMatrix<double> p(d, n, data, d);
    DenseMatrix<std::uint32_t> nb;
    DenseMatrix<double> sc;
    find_approximate_neighbors(p, iterations, k, nb, sc);
    for (size_t i = 0; i < (size_t)k * n; i++) { ann[i] = (int)nb.data()[i]; if (scores) scores[i] = sc.data()[i]; }
    return 0;
  } catch (const std::exception& e) { report(e); return 1; }
}
int SPX_kernel_set_neighbors(STRUMPACKKernel K, int k, const int* ann) {
  auto kr = static_cast<KernelRegression*>(K);
  if (!kr || !kr->K) return 1;
  kr->K->set_neighbors(ann, k);
  return 0;
}
int SPX_kernel_node_info(STRUMPACKKernel K, int* out, int cap) {
  auto kr = static_cast<KernelRegression*>(K);
  if (!kr) return -1;
  int c = (int)kr->nodes.size() / 6;
  std::copy(kr->nodes.begin(), kr->nodes.begin() + 6 * std::min(c, cap), out);
  return c;
}
int SPX_kernel_fit_info(STRUMPACKKernel K, long long* out) {
  auto kr = static_cast<KernelRegression*>(K);
  if (!kr) return 1;
  std::copy(kr->info, kr->info + 8, out);
  return 0;
}
int SPX_kernel_permutation(STRUMPACKKernel K, int* perm) {
  auto kr = static_cast<KernelRegression*>(K);
  if (!kr || !kr->K) return 1;
  std::copy(kr->K->permutation().begin(), kr->K->permutation().end(), perm);
  return 0;
}
int SPX_kernel_weights(STRUMPACKKernel K, double* w) {
  auto kr = static_cast<KernelRegression*>(K);
  if (!kr || kr->weights.rows() == 0) return 1;
  std::copy(kr->weights.data(), kr->weights.data() + kr->weights.rows(), w);
  return 0;
}
int SPX_clustering(int n, int d, double* data, int algo, int leaf_size, int* perm, int* leaf_sizes, int cap) {
  try {
    DenseMatrix<double> p(d, n, data, d);
    std::vector<int> pm;
    static const ClusteringAlgorithm algos[] = {ClusteringAlgorithm::NATURAL, ClusteringAlgorithm::TWO_MEANS, ClusteringAlgorithm::KD_TREE,
                                                ClusteringAlgorithm::PCA, ClusteringAlgorithm::COBBLE};
    if (algo < 0 || algo > 4) throw std::invalid_argument("clustering algorithm out of range");
    auto t = binary_tree_clustering(algos[algo], p, pm, leaf_size);
    std::copy(p.data(), p.data() + (size_t)d * n, data);
    std::copy(pm.begin(), pm.end(), perm);
    int c = 0;
    std::function<void(const structured::ClusterTree&)> walk = [&](const structured::ClusterTree& nd) {
      if (nd.c.empty()) { if (c < cap) leaf_sizes[c] = nd.size; c++; }
      else for (auto& ch : nd.c) walk(ch);
    };
    walk(t);
    return c;
  } catch (const std::exception& e) { report(e); return -1; }
}

int SPX_clustering_device(int n, int d, double* data, int algo, int leaf_size, int* perm, int* leaf_sizes, int cap, int* status) {
  try {
    DenseMatrix<double> p(d, n, data, d);
    std::vector<int> pm;
    static const ClusteringAlgorithm algos[] = {ClusteringAlgorithm::NATURAL, ClusteringAlgorithm::TWO_MEANS, ClusteringAlgorithm::KD_TREE,
                                                ClusteringAlgorithm::PCA, ClusteringAlgorithm::COBBLE};
    if (algo < 0 || algo > 4) throw std::invalid_argument("clustering algorithm out of range");
    int dev = 0;
    if (const char* e = std::getenv("STRUMPACK_AMD_DEVICE")) dev = std::atoi(e);
    structured::ClusterTree t(0);
    const int st = binary_tree_clustering_device(algos[algo], p, pm, (std::size_t)leaf_size, dev, t);
    if (status) *status = st;
    if (st) return 0;
    std::copy(p.data(), p.data() + (size_t)d * n, data);
    std::copy(pm.begin(), pm.end(), perm);
    int c = 0;
    std::function<void(const structured::ClusterTree&)> walk = [&](const structured::ClusterTree& nd) {
      if (nd.c.empty()) { if (c < cap) leaf_sizes[c] = nd.size; c++; }
      else for (auto& ch : nd.c) walk(ch);
    };
    walk(t);
    return c;
  } catch (const std::exception& e) { report(e); return -1; }
}

}  // extern "C"
