// strumpack::DenseMatrix<T> / DenseMatrixWrapper<T>: the host-side column-major containers callers of
// the reference compile against (reference: dense/DenseMatrix.hpp:139-146,1018).  Only the subset
// the HSS / structured API exchanges with the user is provided; all arithmetic of the hot path runs
// on the device (HIP kernels behind include/hssk.h), so these are plain storage + a few O(mn) helpers.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <complex>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <random>
#include <string>
#include <vector>
#if defined(_OPENMP)
#include <omp.h>   // (the reference's headers bring it in; its drivers call omp_get_max_threads())
#endif

namespace strumpack {

enum class Trans : char { N = 'N', C = 'C', T = 'T' };
enum class Side : char { L = 'L', R = 'R' };
enum class UpLo : char { U = 'U', L = 'L' };
enum class Diag : char { U = 'U', N = 'N' };
inline Trans c2T(char op) {
  switch (op) {
    case 'n': case 'N': return Trans::N;
    case 't': case 'T': return Trans::T;
    case 'c': case 'C': return Trans::C;
    default: std::cerr << "ERROR: char " << op << " not recognized, should be one of n/N, t/T or c/C" << std::endl; return Trans::N;
  }
}

template <typename T> inline T conj_of(const T& v) { return v; }
template <typename T> inline std::complex<T> conj_of(const std::complex<T>& v) { return std::conj(v); }

template <typename scalar_t> class DenseMatrix {
 public:
  DenseMatrix() {}
  DenseMatrix(std::size_t m, std::size_t n) : rows_(m), cols_(n), ld_(std::max<std::size_t>(1, m)), own_(true) {
    data_ = new scalar_t[ld_ * std::max<std::size_t>(1, n)]();
  }
  // copy of an m x n block with leading dimension ld
  // (element by element from a function, dense/DenseMatrix.hpp:131-133)
  DenseMatrix(std::size_t m, std::size_t n, const std::function<scalar_t(std::size_t, std::size_t)>& A) : DenseMatrix(m, n) {
    for (std::size_t j = 0; j < n; j++) for (std::size_t i = 0; i < m; i++) (*this)(i, j) = A(i, j);
  }
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
  // B = (*this)(I, J)  (dense/DenseMatrix.hpp:551)
  DenseMatrix<scalar_t> extract(const std::vector<std::size_t>& I, const std::vector<std::size_t>& J) const {
    DenseMatrix<scalar_t> B(I.size(), J.size());
    for (std::size_t j = 0; j < J.size(); j++)
      for (std::size_t i = 0; i < I.size(); i++) {
        assert(I[i] < rows_ && J[j] < cols_);
        B(i, j) = (*this)(I[i], J[j]);
      }
    return B;
  }
  DenseMatrix<scalar_t> extract_rows(const std::vector<std::size_t>& I) const {
    DenseMatrix<scalar_t> B(I.size(), cols_);
    for (std::size_t j = 0; j < cols_; j++)
      for (std::size_t i = 0; i < I.size(); i++) B(i, j) = (*this)(I[i], j);
    return B;
  }
  DenseMatrix<scalar_t> extract_cols(const std::vector<std::size_t>& J) const {
    DenseMatrix<scalar_t> B(rows_, J.size());
    for (std::size_t j = 0; j < J.size(); j++)
      for (std::size_t i = 0; i < rows_; i++) B(i, j) = (*this)(i, J[j]);
    return B;
  }
  DenseMatrix<scalar_t> transpose() const {
    DenseMatrix<scalar_t> T(cols_, rows_);
    for (std::size_t j = 0; j < cols_; j++)
      for (std::size_t i = 0; i < rows_; i++) T(j, i) = (*this)(i, j);
    return T;
  }
  // (real scalars: the plain transpose; complex: conjugated -- dense/DenseMatrix.hpp:431-436)
  DenseMatrix<scalar_t> conj_transpose() const {
    DenseMatrix<scalar_t> T(cols_, rows_);
    conj_transpose(T);
    return T;
  }
  void conj_transpose(DenseMatrix<scalar_t>& X) const {
    assert(X.rows() == cols_ && X.cols() == rows_);
    for (std::size_t j = 0; j < cols_; j++)
      for (std::size_t i = 0; i < rows_; i++) X(j, i) = conj_of((*this)(i, j));
  }
  // this = alpha this + B; rows scaled / divided by D (dense/DenseMatrix.cpp:463-541)
  DenseMatrix<scalar_t>& scale_and_add(scalar_t alpha, const DenseMatrix<scalar_t>& B, int = 0) {
    assert(B.rows() == rows_ && B.cols() == cols_);
    for (std::size_t j = 0; j < cols_; j++) for (std::size_t i = 0; i < rows_; i++) (*this)(i, j) = alpha * (*this)(i, j) + B(i, j);
    return *this;
  }
  DenseMatrix<scalar_t>& scale_rows(const std::vector<scalar_t>& D, int = 0) { assert(D.size() == rows_); return scale_rows(D.data()); }
  DenseMatrix<scalar_t>& scale_rows(const scalar_t* D, int = 0) {
    for (std::size_t j = 0; j < cols_; j++) for (std::size_t i = 0; i < rows_; i++) (*this)(i, j) *= D[i];
    return *this;
  }
  DenseMatrix<scalar_t>& div_rows(const std::vector<scalar_t>& D, int = 0) {
    for (std::size_t j = 0; j < cols_; j++) for (std::size_t i = 0; i < rows_; i++) (*this)(i, j) /= D[i];
    return *this;
  }
  // A(i, i) += sigma; counts of exact zeros / of entries that are not normal numbers (dense/DenseMatrix.cpp:813-860)
  void shift(scalar_t sigma) { for (std::size_t i = 0; i < std::min(rows_, cols_); i++) (*this)(i, i) += sigma; }
  std::size_t zeros() const {
    std::size_t nz = 0;
    for (std::size_t j = 0; j < cols_; j++) for (std::size_t i = 0; i < rows_; i++) nz += (*this)(i, j) == scalar_t(0.);
    return nz;
  }
  std::size_t subnormals() const {
    std::size_t ns = 0;
    for (std::size_t j = 0; j < cols_; j++)
      for (std::size_t i = 0; i < rows_; i++) ns += !std::isnormal(std::real((*this)(i, j))) && !std::isnormal(std::imag((*this)(i, j)));
    return ns;
  }
  // largest column sum / row sum of absolute values (lange '1' / 'I')
  auto norm1() const {
    decltype(std::abs(scalar_t())) nrm = 0;
    for (std::size_t j = 0; j < cols_; j++) { decltype(nrm) s = 0; for (std::size_t i = 0; i < rows_; i++) s += std::abs((*this)(i, j)); nrm = std::max(nrm, s); }
    return nrm;
  }
  auto normI() const {
    std::vector<decltype(std::abs(scalar_t()))> s(rows_, 0);
    for (std::size_t j = 0; j < cols_; j++) for (std::size_t i = 0; i < rows_; i++) s[i] += std::abs((*this)(i, j));
    decltype(std::abs(scalar_t())) nrm = 0;
    for (auto v : s) nrm = std::max(nrm, v);
    return nrm;
  }
  // row / column permutations of LAPACK's lapmr / lapmt (1-based P; forward: row P(i) moves to row i), dense/DenseMatrix.cpp:300-309
  void lapmr(const std::vector<int>& P, bool fwd) {
    if (!rows_ || !cols_) return;
    DenseMatrix<scalar_t> T(*this);
    for (std::size_t i = 0; i < rows_; i++)
      for (std::size_t j = 0; j < cols_; j++) {
        if (fwd) (*this)(i, j) = T(std::size_t(P[i] - 1), j);
        else (*this)(std::size_t(P[i] - 1), j) = T(i, j);
      }
  }
  void lapmt(const std::vector<int>& P, bool fwd) {
    if (!rows_ || !cols_) return;
    DenseMatrix<scalar_t> T(*this);
    for (std::size_t j = 0; j < cols_; j++)
      for (std::size_t i = 0; i < rows_; i++) {
        if (fwd) (*this)(i, j) = T(i, std::size_t(P[j] - 1));
        else (*this)(i, std::size_t(P[j] - 1)) = T(i, j);
      }
  }
  // LU with partial pivoting in place (getrf: 1-based pivots for laswp; returns the first zero pivot's column + 1, else 0) and
  // the solves with it (dense/DenseMatrix.hpp:702-790).  Host loops: the blocks a caller factors here are small -- the
  // engines factor theirs on the device.
  int LU(std::vector<int>& piv, int = 0) {
    assert(rows_ == cols_);
    const std::size_t n = rows_;
    piv.assign(n, 0);
    int info = 0;
    for (std::size_t k = 0; k < n; k++) {
      std::size_t p = k;
      for (std::size_t i = k + 1; i < n; i++) if (std::abs((*this)(i, k)) > std::abs((*this)(p, k))) p = i;
      piv[k] = int(p) + 1;
      if ((*this)(p, k) == scalar_t(0.)) { if (!info) info = int(k) + 1; continue; }
      if (p != k) for (std::size_t j = 0; j < n; j++) std::swap((*this)(k, j), (*this)(p, j));
      const scalar_t d = scalar_t(1.) / (*this)(k, k);
      for (std::size_t i = k + 1; i < n; i++) (*this)(i, k) *= d;
      for (std::size_t j = k + 1; j < n; j++) {
        const scalar_t u = (*this)(k, j);
        if (u != scalar_t(0.)) for (std::size_t i = k + 1; i < n; i++) (*this)(i, j) -= (*this)(i, k) * u;
      }
    }
    return info;
  }
  std::vector<int> LU(int = 0) {
    std::vector<int> piv;
    if (int info = LU(piv)) std::cerr << "ERROR: LU factorization failed with info=" << info << std::endl;
    return piv;
  }
  void solve_LU_in_place(DenseMatrix<scalar_t>& b, const std::vector<int>& piv, int = 0) const {
    assert(rows_ == cols_ && b.rows() == rows_ && piv.size() >= rows_);
    const std::size_t n = rows_;
    b.laswp(piv, true);
    for (std::size_t c = 0; c < b.cols(); c++) {
      for (std::size_t k = 0; k < n; k++) { const scalar_t v = b(k, c); if (v != scalar_t(0.)) for (std::size_t i = k + 1; i < n; i++) b(i, c) -= (*this)(i, k) * v; }
      for (std::size_t k = n; k-- > 0;) { b(k, c) /= (*this)(k, k); const scalar_t v = b(k, c); for (std::size_t i = 0; i < k; i++) b(i, c) -= (*this)(i, k) * v; }
    }
  }
  DenseMatrix<scalar_t> solve(const DenseMatrix<scalar_t>& b, const std::vector<int>& piv, int = 0) const {
    DenseMatrix<scalar_t> x(b);
    solve_LU_in_place(x, piv);
    return x;
  }
  // keeps the leading min(rows, m) x min(cols, n) block (dense/DenseMatrix.hpp: resize)
  void resize(std::size_t m, std::size_t n) {
    DenseMatrix<scalar_t> T(m, n);
    for (std::size_t j = 0; j < std::min(cols_, n); j++)
      for (std::size_t i = 0; i < std::min(rows_, m); i++) T(i, j) = (*this)(i, j);
    *this = std::move(T);
  }
  DenseMatrix<scalar_t>& scale(scalar_t alpha) {
    for (std::size_t j = 0; j < cols_; j++) for (std::size_t i = 0; i < rows_; i++) (*this)(i, j) *= alpha;
    return *this;
  }
  DenseMatrix<scalar_t>& add(const DenseMatrix<scalar_t>& B) { return scaled_add(scalar_t(1.), B); }
  DenseMatrix<scalar_t>& sub(const DenseMatrix<scalar_t>& B) { return scaled_add(scalar_t(-1.), B); }
  // the row interchanges of getrf (1-based LAPACK pivots), applied forwards or backwards (dense/DenseMatrix.cpp:287-297)
  void laswp(const std::vector<int>& P, bool fwd) {
    const long long n = (long long)P.size();
    for (long long q = 0; q < n; q++) {
      const long long k = fwd ? q : n - 1 - q;
      const std::size_t p = std::size_t(P[k] - 1);
      if (p != std::size_t(k))
        for (std::size_t j = 0; j < cols_; j++) std::swap((*this)(k, j), (*this)(p, j));
    }
  }
  void print_to_file(const std::string& name, const std::string& filename, int width = 8) const {
    std::ofstream f(filename);
    f.precision(width + 8);
    f << name << " = [" << std::endl;
    for (std::size_t i = 0; i < rows_; i++) { for (std::size_t j = 0; j < cols_; j++) f << (*this)(i, j) << "  "; f << std::endl; }
    f << "];" << std::endl;
  }
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

// ---- BLAS-shaped free functions on DenseMatrix (dense/DenseMatrix.hpp:1346-1400): C = alpha op(A) op(B) + beta C etc.
// Host loops (these are the caller-side helpers of tests and examples, not the hot path); a double-precision gemm above
// ~0.5 GFLOP is carried to the MI355X by the library (strumpack_amd_dense_gemm: batched MFMA GEMM of include/hssk.h).
extern "C" int strumpack_amd_dense_gemm(char ta, char tb, int m, int n, int k, double alpha, const double* A, int lda,
                                        const double* B, int ldb, double beta, double* C, int ldc);
namespace detail {
template <typename T> inline T conj_if(T v, bool) { return v; }
template <typename R> inline std::complex<R> conj_if(std::complex<R> v, bool c) { return c ? std::conj(v) : v; }
template <typename scalar_t> inline scalar_t opel(const DenseMatrix<scalar_t>& A, Trans t, std::size_t i, std::size_t j) {
  return t == Trans::N ? A(i, j) : conj_if(A(j, i), t == Trans::C);
}
template <typename scalar_t>
void gemm_host(Trans ta, Trans tb, scalar_t alpha, const DenseMatrix<scalar_t>& a, const DenseMatrix<scalar_t>& b, scalar_t beta,
               DenseMatrix<scalar_t>& c) {
  const std::size_t m = c.rows(), n = c.cols(), k = ta == Trans::N ? a.cols() : a.rows();
  for (std::size_t j = 0; j < n; j++) {
    for (std::size_t i = 0; i < m; i++) c(i, j) = beta == scalar_t(0.) ? scalar_t(0.) : beta * c(i, j);
    for (std::size_t l = 0; l < k; l++) {
      const scalar_t blj = alpha * opel(b, tb, l, j);
      if (blj == scalar_t(0.)) continue;
      if (ta == Trans::N) for (std::size_t i = 0; i < m; i++) c(i, j) += a(i, l) * blj;
      else for (std::size_t i = 0; i < m; i++) c(i, j) += opel(a, ta, i, l) * blj;
    }
  }
}
}  // namespace detail
template <typename scalar_t>
void gemm(Trans ta, Trans tb, scalar_t alpha, const DenseMatrix<scalar_t>& a, const DenseMatrix<scalar_t>& b, scalar_t beta,
          DenseMatrix<scalar_t>& c, int /*depth*/ = 0) {
  assert((ta == Trans::N ? a.rows() : a.cols()) == c.rows() && (tb == Trans::N ? b.cols() : b.rows()) == c.cols());
  detail::gemm_host(ta, tb, alpha, a, b, beta, c);
}
inline void gemm(Trans ta, Trans tb, double alpha, const DenseMatrix<double>& a, const DenseMatrix<double>& b, double beta,
                 DenseMatrix<double>& c, int /*depth*/ = 0) {
  const std::size_t m = c.rows(), n = c.cols(), k = ta == Trans::N ? a.cols() : a.rows();
  assert((ta == Trans::N ? a.rows() : a.cols()) == m && (tb == Trans::N ? b.cols() : b.rows()) == n);
  if (2.0 * double(m) * double(n) * double(k) >= 5e8 &&
      strumpack_amd_dense_gemm(ta == Trans::N ? 'N' : 'T', tb == Trans::N ? 'N' : 'T', int(m), int(n), int(k), alpha, a.data(), a.ld(),
                               b.data(), b.ld(), beta, c.data(), c.ld()) == 0)
    return;
  detail::gemm_host(ta, tb, alpha, a, b, beta, c);
}
template <typename scalar_t>
void gemv(Trans ta, scalar_t alpha, const DenseMatrix<scalar_t>& a, const DenseMatrix<scalar_t>& x, scalar_t beta,
          DenseMatrix<scalar_t>& y, int /*depth*/ = 0) {
  detail::gemm_host(ta, Trans::N, alpha, a, x, beta, y);
}
// y <- alpha op(a) x + beta y on strided vectors (dense/DenseMatrix.hpp: gemv with raw pointers)
template <typename scalar_t>
void gemv(Trans ta, scalar_t alpha, const DenseMatrix<scalar_t>& a, const scalar_t* x, int incx, scalar_t beta, scalar_t* y, int incy,
          int /*depth*/ = 0) {
  const std::size_t m = a.rows(), n = a.cols();
  const std::size_t ny = ta == Trans::N ? m : n, nx = ta == Trans::N ? n : m;
  for (std::size_t i = 0; i < ny; i++) y[i * incy] = beta == scalar_t(0.) ? scalar_t(0.) : beta * y[i * incy];
  if (ta == Trans::N) {
    for (std::size_t j = 0; j < nx; j++) { const scalar_t t = alpha * x[j * incx]; for (std::size_t i = 0; i < m; i++) y[i * incy] += a(i, j) * t; }
  } else {
    for (std::size_t j = 0; j < ny; j++) {
      scalar_t t = 0;
      for (std::size_t i = 0; i < m; i++) t += (ta == Trans::C ? conj_of(a(i, j)) : a(i, j)) * x[i * incx];
      y[j * incy] += alpha * t;
    }
  }
}
// b <- alpha op(a)^{-1} b (Side::L) or alpha b op(a)^{-1} (Side::R), a triangular (dense/DenseMatrix.cpp:1059-1085)
template <typename scalar_t>
void trsm(Side s, UpLo ul, Trans ta, Diag d, scalar_t alpha, const DenseMatrix<scalar_t>& a, DenseMatrix<scalar_t>& b, int /*depth*/ = 0) {
  const std::size_t n = a.rows();
  // effective triangle of op(a): transposing flips it
  const bool lower = (ul == UpLo::L) != (ta != Trans::N);
  auto A = [&](std::size_t i, std::size_t j) { return detail::opel(a, ta, i, j); };
  b.scale(alpha);
  if (s == Side::L) {
    for (std::size_t c = 0; c < b.cols(); c++) {
      if (lower)
        for (std::size_t i = 0; i < n; i++) {
          scalar_t v = b(i, c);
          for (std::size_t l = 0; l < i; l++) v -= A(i, l) * b(l, c);
          b(i, c) = d == Diag::U ? v : v / A(i, i);
        }
      else
        for (std::size_t q = n; q-- > 0;) {
          scalar_t v = b(q, c);
          for (std::size_t l = q + 1; l < n; l++) v -= A(q, l) * b(l, c);
          b(q, c) = d == Diag::U ? v : v / A(q, q);
        }
    }
  } else {   // x op(a) = b, row by row: x(r, j) = (b(r, j) - sum_l x(r, l) op(a)(l, j)) / op(a)(j, j)
    for (std::size_t r = 0; r < b.rows(); r++) {
      if (lower)   // x(r, j) depends on l > j
        for (std::size_t j = n; j-- > 0;) {
          scalar_t v = b(r, j);
          for (std::size_t l = j + 1; l < n; l++) v -= b(r, l) * A(l, j);
          b(r, j) = d == Diag::U ? v : v / A(j, j);
        }
      else
        for (std::size_t j = 0; j < n; j++) {
          scalar_t v = b(r, j);
          for (std::size_t l = 0; l < j; l++) v -= b(r, l) * A(l, j);
          b(r, j) = d == Diag::U ? v : v / A(j, j);
        }
    }
  }
}
template <typename scalar_t>
void trsv(UpLo ul, Trans ta, Diag d, const DenseMatrix<scalar_t>& a, DenseMatrix<scalar_t>& b, int depth = 0) {
  trsm(Side::L, ul, ta, d, scalar_t(1.), a, b, depth);
}
template <typename scalar_t> DenseMatrix<scalar_t> vconcat(const DenseMatrix<scalar_t>& a, const DenseMatrix<scalar_t>& b) {
  assert(a.cols() == b.cols());
  DenseMatrix<scalar_t> t(a.rows() + b.rows(), a.cols());
  for (std::size_t j = 0; j < a.cols(); j++) {
    for (std::size_t i = 0; i < a.rows(); i++) t(i, j) = a(i, j);
    for (std::size_t i = 0; i < b.rows(); i++) t(a.rows() + i, j) = b(i, j);
  }
  return t;
}
template <typename scalar_t> DenseMatrix<scalar_t> hconcat(const DenseMatrix<scalar_t>& a, const DenseMatrix<scalar_t>& b) {
  assert(a.rows() == b.rows());
  DenseMatrix<scalar_t> t(a.rows(), a.cols() + b.cols());
  for (std::size_t j = 0; j < a.cols(); j++) for (std::size_t i = 0; i < a.rows(); i++) t(i, j) = a(i, j);
  for (std::size_t j = 0; j < b.cols(); j++) for (std::size_t i = 0; i < a.rows(); i++) t(i, a.cols() + j) = b(i, j);
  return t;
}

}  // namespace strumpack
