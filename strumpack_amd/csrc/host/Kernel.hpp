// kernel::Kernel and its Gauss / Laplace / ANOVA subclasses (reference: kernel/Kernel.hpp:73-399,
// kernel/KernelRegression.hpp:56-123): a kernel matrix K(i, j) = k(x_i, x_j) + lambda [i == j] over the columns
// of a d x n point matrix, kernel ridge regression through an HSS approximation (fit_HSS) and prediction.
//
// The point set stays on the host in the caller's matrix (the reference keeps a reference to it and reorders
// it in place while clustering -- same here); entries needed by the HSS construction, the nearest-neighbour
// lists and the prediction sums are evaluated on the MI355X from a device copy (hssk_kernel_eval_vbatched,
// hssk_knn, hssk_kernel_predict).  eval() / operator() remain available on the host as the scalar API of the
// reference (single entries, small blocks); they are not on any compute path of this library.
#pragma once
#include <cmath>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "DenseMatrix.hpp"
#include "HSSOptions.hpp"

namespace strumpack {
namespace kernel {

enum class KernelType { DENSE, GAUSS, LAPLACE, ANOVA };

inline std::string get_name(KernelType k) {
  switch (k) {
    case KernelType::DENSE: return "dense";
    case KernelType::GAUSS: return "Gauss";
    case KernelType::LAPLACE: return "Laplace";
    case KernelType::ANOVA: return "ANOVA";
  }
  return "UNKNOWN";
}
inline KernelType kernel_type(const std::string& k) {
  if (k == "dense") return KernelType::DENSE;
  if (k == "Gauss") return KernelType::GAUSS;
  if (k == "Laplace") return KernelType::LAPLACE;
  if (k == "ANOVA") return KernelType::ANOVA;
  std::cerr << "ERROR: Kernel type not recogonized,  setting kernel type to Gauss." << std::endl;
  return KernelType::GAUSS;
}

template <typename scalar_t> class Kernel;

template <> class Kernel<double> {
  using scalar_t = double;
  using DenseM_t = DenseMatrix<double>;

 public:
  Kernel(DenseM_t& data, scalar_t lambda) : data_(data), lambda_(lambda) {}
  virtual ~Kernel() = default;

  std::size_t n() const { return data_.cols(); }
  std::size_t d() const { return data_.rows(); }

  virtual scalar_t eval(std::size_t i, std::size_t j) const {
    return eval_kernel_function(data_.ptr(0, i), data_.ptr(0, j)) + ((i == j) ? lambda_ : scalar_t(0.));
  }
  void operator()(const std::vector<std::size_t>& I, const std::vector<std::size_t>& J, DenseM_t& B) const {
    if (B.rows() != I.size() || B.cols() != J.size()) throw std::invalid_argument("Kernel::operator(): B has the wrong size");
    for (std::size_t j = 0; j < J.size(); j++)
      for (std::size_t i = 0; i < I.size(); i++) B(i, j) = eval(I[i], J[j]);
  }

  // kernel ridge regression: weights = (K + lambda I)^{-1} labels through an HSS approximation of K
  // (labels are permuted to the cluster order in place, as in the reference)
  DenseM_t fit_HSS(std::vector<scalar_t>& labels, const HSS::HSSOptions<scalar_t>& opts);
  // prediction[c] = sum_r weights(r) k(x_r, test_c)
  std::vector<scalar_t> predict(const DenseM_t& test, const DenseM_t& weights) const;

  const DenseM_t& data() const { return data_; }
  DenseM_t& data() { return data_; }
  std::vector<int>& permutation() { return perm_; }
  const std::vector<int>& permutation() const { return perm_; }
  // the clustering already reordered data() in place (binary_tree_clustering); nothing left to move
  virtual void permute() {}

  scalar_t lambda() const { return lambda_; }
  // extension (tests): neighbour lists (k x n, 0-based ids in cluster order) that replace the device search of the
  // first compression round
  void set_neighbors(const int* ann, int k) { user_ann_.assign(ann, ann + (size_t)k * n()); user_k_ = k; }
  const int* neighbors() const { return user_ann_.empty() ? nullptr : user_ann_.data(); }
  int neighbor_count() const { return user_k_; }
  // device evaluation parameters: 0 Gauss, 1 Laplace, 2 ANOVA, -1 = user-defined (host only)
  virtual int device_type() const { return -1; }
  virtual scalar_t width() const { return 1.; }
  virtual int degree() const { return 1; }

 protected:
  DenseM_t& data_;
  scalar_t lambda_;
  std::vector<int> perm_, user_ann_;
  int user_k_ = 0;
  virtual scalar_t eval_kernel_function(const scalar_t* x, const scalar_t* y) const = 0;
};

template <typename scalar_t> class GaussKernel;
template <> class GaussKernel<double> : public Kernel<double> {
 public:
  GaussKernel(DenseMatrix<double>& data, double h, double lambda) : Kernel<double>(data, lambda), h_(h) {}
  int device_type() const override { return 0; }
  double width() const override { return h_; }

 protected:
  double h_;
  double eval_kernel_function(const double* x, const double* y) const override {
    double s = 0.;
    for (std::size_t i = 0; i < d(); i++) { double t = x[i] - y[i]; s += t * t; }
    return std::exp(-s / (2. * h_ * h_));
  }
};

template <typename scalar_t> class LaplaceKernel;
template <> class LaplaceKernel<double> : public Kernel<double> {
 public:
  LaplaceKernel(DenseMatrix<double>& data, double h, double lambda) : Kernel<double>(data, lambda), h_(h) {}
  int device_type() const override { return 1; }
  double width() const override { return h_; }

 protected:
  double h_;
  double eval_kernel_function(const double* x, const double* y) const override {
    double s = 0.;
    for (std::size_t i = 0; i < d(); i++) s += std::abs(x[i] - y[i]);
    return std::exp(-s / h_);
  }
};

template <typename scalar_t> class ANOVAKernel;
template <> class ANOVAKernel<double> : public Kernel<double> {
 public:
  ANOVAKernel(DenseMatrix<double>& data, double h, double lambda, int p = 1) : Kernel<double>(data, lambda), h_(h), p_(p) {
    if (p < 1 || p > int(d())) throw std::invalid_argument("ANOVAKernel: degree must be in [1, d]");
  }
  int device_type() const override { return 2; }
  double width() const override { return h_; }
  int degree() const override { return p_; }

 protected:
  double h_;
  int p_;
  double eval_kernel_function(const double* x, const double* y) const override {
    std::vector<double> Kss(p_, 0.), Kpp(p_ + 1);
    for (std::size_t i = 0; i < d(); i++) {
      const double t = x[i] - y[i], tmp = std::exp(-(t * t) / (2. * h_ * h_));
      double pw = tmp;
      for (int j = 0; j < p_; j++) { Kss[j] += pw; pw *= tmp; }
    }
    Kpp[0] = 1.;
    for (int i = 1; i <= p_; i++) {
      double s = 0.;
      for (int q = 1; q <= i; q++) s += ((q & 1) ? 1. : -1.) * Kpp[i - q] * Kss[q - 1];
      Kpp[i] = s / i;
    }
    return Kpp[p_];
  }
};

template <typename scalar_t>
std::unique_ptr<Kernel<scalar_t>> create_kernel(KernelType k, DenseMatrix<scalar_t>& data, scalar_t h, scalar_t lambda, int p = 1) {
  switch (k) {
    case KernelType::LAPLACE: return std::unique_ptr<Kernel<scalar_t>>(new LaplaceKernel<scalar_t>(data, h, lambda));
    case KernelType::ANOVA: return std::unique_ptr<Kernel<scalar_t>>(new ANOVAKernel<scalar_t>(data, h, lambda, p));
    case KernelType::GAUSS:
    default: return std::unique_ptr<Kernel<scalar_t>>(new GaussKernel<scalar_t>(data, h, lambda));
  }
}

}  // namespace kernel
}  // namespace strumpack
