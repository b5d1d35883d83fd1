// A user-defined kernel function: a subclass of kernel::Kernel<double> that only overrides the virtual evaluation
// (kernel/Kernel.hpp:73-170 of the reference).  HSSMatrix(K, opts) must compress it -- its blocks are evaluated on the host and
// uploaded, the rest of the path is the device's -- and fit_HSS / predict must work.   usage: test_user_kernel <n> <d>
#include <cmath>
#include <iostream>
#include <random>
#include <vector>

#include "HSS/HSSMatrix.hpp"
#include "kernel/Kernel.hpp"
#include "kernel/KernelRegression.hpp"

using namespace strumpack;

// rational quadratic ("Cauchy") kernel: k(x, y) = 1 / (1 + |x - y|^2 / h^2)
class CauchyKernel : public kernel::Kernel<double> {
 public:
  CauchyKernel(DenseMatrix<double>& data, double h, double lambda) : Kernel<double>(data, lambda), h_(h) {}

 protected:
  double h_;
  double eval_kernel_function(const double* x, const double* y) const override {
    double s = 0.;
    for (std::size_t k = 0; k < this->d(); k++) s += (x[k] - y[k]) * (x[k] - y[k]);
    return 1. / (1. + s / (h_ * h_));
  }
};

int main(int argc, char* argv[]) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 600, d = argc > 2 ? std::atoi(argv[2]) : 3;
  std::mt19937 g(7);
  std::uniform_real_distribution<double> u(0., 1.);
  DenseMatrix<double> X(d, n), T(d, 50);
  for (int j = 0; j < n; j++) for (int i = 0; i < d; i++) X(i, j) = u(g);
  for (int j = 0; j < 50; j++) for (int i = 0; i < d; i++) T(i, j) = u(g);
  std::vector<double> labels(n);
  for (int j = 0; j < n; j++) labels[j] = X(0, j) > 0.5 ? 1. : -1.;
  const double h = 0.7, lambda = 2.0;
  HSS::HSSOptions<double> opts;
  opts.set_rel_tol(1e-6); opts.set_abs_tol(1e-10); opts.set_leaf_size(64);
  opts.set_clustering_algorithm(ClusteringAlgorithm::KD_TREE);
  opts.set_approximate_neighbors(64);
  {
    DenseMatrix<double> Xc(X);
    CauchyKernel K(Xc, h, lambda);
    HSS::HSSMatrix<double> H(K, opts);      // clusters (Xc is reordered), compresses
    if (!H.is_compressed()) { std::cout << "ERROR: compression failed" << std::endl; return 1; }
    auto Hd = H.dense();
    double num = 0, den = 0;
    for (int j = 0; j < n; j++)
      for (int i = 0; i < n; i++) {
        const double e = K.eval(i, j);
        num += (Hd(i, j) - e) * (Hd(i, j) - e); den += e * e;
      }
    std::cout << "# user-defined kernel: rank " << H.rank() << ", ||H - K||_F / ||K||_F = " << std::sqrt(num / den) << std::endl;
    if (std::sqrt(num / den) > 1e-4) { std::cout << "ERROR: compression error too large" << std::endl; return 1; }
  }
  {
    DenseMatrix<double> Xc(X);
    CauchyKernel K(Xc, h, lambda);
    auto w = K.fit_HSS(labels, opts);       // clusters, compresses, factors, solves
    // the weights solve (K + lambda I) w = y for the permuted labels: check through predict on the training points' own kernel
    double num = 0, den = 0;
    for (int i = 0; i < n; i++) {
      double s = 0.;
      for (int j = 0; j < n; j++) s += K.eval(i, j) * w(j, 0);
      num += (s - labels[i]) * (s - labels[i]); den += labels[i] * labels[i];
    }
    std::cout << "# ||K w - y|| / ||y|| = " << std::sqrt(num / den) << std::endl;
    if (std::sqrt(num / den) > 1e-3) { std::cout << "ERROR: fit_HSS residual" << std::endl; return 1; }
    auto pred = K.predict(T, w);
    for (int c = 0; c < 50; c++) {
      double s = 0.;
      for (int r = 0; r < n; r++) {
        double q = 0.;
        for (int k = 0; k < d; k++) q += (Xc(k, r) - T(k, c)) * (Xc(k, r) - T(k, c));
        s += w(r, 0) / (1. + q / (h * h));
      }
      if (std::abs(s - pred[c]) > 1e-10 * (1. + std::abs(s))) { std::cout << "ERROR: predict" << std::endl; return 1; }
    }
  }
  std::cout << "# exiting" << std::endl;
  return 0;
}
