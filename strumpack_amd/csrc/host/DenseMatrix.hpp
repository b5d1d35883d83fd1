// strumpack::DenseMatrix<T> / DenseMatrixWrapper<T>: the host-side column-major containers callers of
// the reference compile against (reference: dense/DenseMatrix.hpp:139-146,1018).  Only the subset
// the HSS / structured API exchanges with the user is provided; all arithmetic of the hot path runs
// on the device (HIP kernels behind include/hssk.h), so these are plain storage + a few O(mn) helpers.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <iostream>
#include <random>
#include <string>
#include <vector>

namespace strumpack {

enum class Trans : char { N = 'N', C = 'C', T = 'T' };
inline Trans c2T(char op) {
  switch (op) {
    case 'n': case 'N': return Trans::N;
    case 't': case 'T': return Trans::T;
    case 'c': case 'C': return Trans::C;
    default: std::cerr << "ERROR: char " << op << " not recognized, should be one of n/N, t/T or c/C" << std::endl; return Trans::N;
  }
}

template <typename scalar_t> class DenseMatrix {
 public:
  DenseMatrix() {}
  DenseMatrix(std::size_t m, std::size_t n) : rows_(m), cols_(n), ld_(std::max<std::size_t>(1, m)), own_(true) {
    data_ = new scalar_t[ld_ * std::max<std::size_t>(1, n)]();
  }
  // copy of an m x n block with leading dimension ld
  DenseMatrix(std::size_t m, std::size_t n, const scalar_t* D, std::size_t ld) : DenseMatrix(m, n) {
    for (std::size_t j = 0; j < n; j++) std::memcpy(data_ + j * ld_, D + j * ld, sizeof(scalar_t) * m);
  }
  // copy of the sub-block D(i:i+m, j:j+n)
  DenseMatrix(std::size_t m, std::size_t n, const DenseMatrix<scalar_t>& D, std::size_t i, std::size_t j)
      : DenseMatrix(m, n, D.ptr(i, j), D.ld()) {}
  DenseMatrix(const DenseMatrix<scalar_t>& D) : DenseMatrix(D.rows(), D.cols(), D.data(), D.ld()) {}
  DenseMatrix(DenseMatrix<scalar_t>&& D) noexcept { steal(D); }
  virtual ~DenseMatrix() { if (own_) delete[] data_; }
  DenseMatrix<scalar_t>& operator=(const DenseMatrix<scalar_t>& D) {
    if (this == &D) return *this;
    if (own_) delete[] data_;
    rows_ = D.rows(); cols_ = D.cols(); ld_ = std::max<std::size_t>(1, rows_); own_ = true;
    data_ = new scalar_t[ld_ * std::max<std::size_t>(1, cols_)];
    for (std::size_t j = 0; j < cols_; j++) std::memcpy(data_ + j * ld_, D.ptr(0, j), sizeof(scalar_t) * rows_);
    return *this;
  }
  DenseMatrix<scalar_t>& operator=(DenseMatrix<scalar_t>&& D) noexcept {
    if (this != &D) { if (own_) delete[] data_; steal(D); }
    return *this;
  }

  std::size_t rows() const { return rows_; }
  std::size_t cols() const { return cols_; }
  int ld() const { return int(ld_); }
  const scalar_t* data() const { return data_; }
  scalar_t* data() { return data_; }
  const scalar_t& operator()(std::size_t i, std::size_t j) const { return data_[i + ld_ * j]; }
  scalar_t& operator()(std::size_t i, std::size_t j) { return data_[i + ld_ * j]; }
  const scalar_t* ptr(std::size_t i, std::size_t j) const { return data_ + i + ld_ * j; }
  scalar_t* ptr(std::size_t i, std::size_t j) { return data_ + i + ld_ * j; }

  void zero() { fill(scalar_t(0.)); }
  void fill(scalar_t v) { for (std::size_t j = 0; j < cols_; j++) for (std::size_t i = 0; i < rows_; i++) (*this)(i, j) = v; }
  void eye() { zero(); for (std::size_t i = 0; i < std::min(rows_, cols_); i++) (*this)(i, i) = scalar_t(1.); }
  // default generator of the reference: minstd_rand(0) + normal_distribution, column-major serial fill
  // (dense/DenseMatrix.cpp:183-190, misc/RandomWrapper.hpp:238-241)
  void random() {
    std::minstd_rand e(0);
    std::normal_distribution<double> d;
    for (std::size_t j = 0; j < cols_; j++) for (std::size_t i = 0; i < rows_; i++) (*this)(i, j) = scalar_t(d(e));
  }
  void clear() { if (own_) delete[] data_; data_ = nullptr; rows_ = cols_ = 0; ld_ = 1; own_ = true; }
  void copy(const DenseMatrix<scalar_t>& B, std::size_t i = 0, std::size_t j = 0) {
    for (std::size_t c = 0; c < cols_; c++) for (std::size_t r = 0; r < rows_; r++) (*this)(r, c) = B(i + r, j + c);
  }
  DenseMatrix<scalar_t>& scaled_add(scalar_t alpha, const DenseMatrix<scalar_t>& B) {
    for (std::size_t j = 0; j < cols_; j++) for (std::size_t i = 0; i < rows_; i++) (*this)(i, j) += alpha * B(i, j);
    return *this;
  }
  double normF() const {
    long double s = 0;
    for (std::size_t j = 0; j < cols_; j++) for (std::size_t i = 0; i < rows_; i++) s += (long double)(*this)(i, j) * (*this)(i, j);
    return std::sqrt((double)s);
  }
  double norm() const { return normF(); }
  std::size_t memory() const { return sizeof(scalar_t) * rows_ * cols_; }
  std::size_t nonzeros() const { return rows_ * cols_; }
  void print(const std::string& name = "A") const {
    std::cout << name << " = [  % " << rows_ << "x" << cols_ << ", ld=" << ld_ << std::endl;
    for (std::size_t i = 0; i < rows_; i++) { for (std::size_t j = 0; j < cols_; j++) std::cout << (*this)(i, j) << "  "; std::cout << std::endl; }
    std::cout << "];" << std::endl;
  }

 protected:
  void steal(DenseMatrix<scalar_t>& D) {
    data_ = D.data_; rows_ = D.rows_; cols_ = D.cols_; ld_ = D.ld_; own_ = D.own_;
    D.data_ = nullptr; D.rows_ = D.cols_ = 0; D.ld_ = 1; D.own_ = true;
  }
  scalar_t* data_ = nullptr;
  std::size_t rows_ = 0, cols_ = 0, ld_ = 1;
  bool own_ = true;
};

// non-owning view (reference: dense/DenseMatrix.hpp:1018)
template <typename scalar_t> class DenseMatrixWrapper : public DenseMatrix<scalar_t> {
 public:
  DenseMatrixWrapper() { this->own_ = false; }
  DenseMatrixWrapper(std::size_t m, std::size_t n, scalar_t* D, std::size_t ld) {
    this->data_ = D; this->rows_ = m; this->cols_ = n; this->ld_ = std::max<std::size_t>(1, ld); this->own_ = false;
  }
  DenseMatrixWrapper(std::size_t m, std::size_t n, DenseMatrix<scalar_t>& D, std::size_t i, std::size_t j)
      : DenseMatrixWrapper(m, n, D.ptr(i, j), D.ld()) {}
  DenseMatrixWrapper(const DenseMatrixWrapper<scalar_t>& o) : DenseMatrix<scalar_t>() {
    this->data_ = o.data_; this->rows_ = o.rows_; this->cols_ = o.cols_; this->ld_ = o.ld_; this->own_ = false;
  }
  DenseMatrixWrapper<scalar_t>& operator=(const DenseMatrixWrapper<scalar_t>& o) {
    this->data_ = o.data_; this->rows_ = o.rows_; this->cols_ = o.cols_; this->ld_ = o.ld_; this->own_ = false;
    return *this;
  }
};

template <typename scalar_t>
DenseMatrixWrapper<scalar_t> ConstDenseMatrixWrapper(std::size_t m, std::size_t n, const scalar_t* D, std::size_t ld) {
  return DenseMatrixWrapper<scalar_t>(m, n, const_cast<scalar_t*>(D), ld);
}

}  // namespace strumpack
