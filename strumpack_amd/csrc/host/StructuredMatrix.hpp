// structured::StructuredMatrix<T> and its factories -- the format-agnostic facade callers of the
// reference compile against (reference: structured/StructuredMatrix.hpp:209-419 class, :462-869
// factories; dispatch structured/StructuredMatrix.cpp:54-76,203-273,637-651).  This build implements
// the Type::HSS branch (the hot path); every other type throws std::invalid_argument exactly like the
// reference does for formats it was not configured with.
#pragma once
#include <functional>
#include <memory>
#include <stdexcept>
#include <vector>

#include "ClusterTree.hpp"
#include "DenseMatrix.hpp"
#include "StructuredOptions.hpp"

namespace strumpack {
namespace structured {

template <typename scalar_t> using extract_t = std::function<scalar_t(std::size_t i, std::size_t j)>;
template <typename scalar_t> using extract_block_t =
    std::function<void(const std::vector<std::size_t>& I, const std::vector<std::size_t>& J, DenseMatrix<scalar_t>& B)>;
template <typename scalar_t> using mult_t =
    std::function<void(Trans op, const DenseMatrix<scalar_t>& R, DenseMatrix<scalar_t>& S)>;
using admissibility_t = DenseMatrix<bool>;

template <typename scalar_t> class StructuredMatrix {
 public:
  virtual ~StructuredMatrix() = default;
  virtual std::size_t rows() const = 0;
  virtual std::size_t cols() const = 0;
  virtual std::size_t memory() const = 0;
  virtual std::size_t nonzeros() const = 0;
  virtual std::size_t rank() const = 0;
  // y = op(A) x
  virtual void mult(Trans op, const DenseMatrix<scalar_t>& x, DenseMatrix<scalar_t>& y) const {
    throw std::invalid_argument("Operation mult not supported for this type.");
  }
  void mult(Trans op, int m, const scalar_t* x, int ldx, scalar_t* y, int ldy) const {
    int nx = (op == Trans::N) ? int(cols()) : int(rows());
    int ny = (op == Trans::N) ? int(rows()) : int(cols());
    auto X = ConstDenseMatrixWrapper<scalar_t>(nx, m, x, ldx);
    DenseMatrixWrapper<scalar_t> Y(ny, m, y, ldy);
    mult(op, X, Y);
  }
  virtual void factor() { throw std::invalid_argument("Operation factor not supported for this type."); }
  virtual void solve(DenseMatrix<scalar_t>& b) const {
    throw std::invalid_argument("Operation solve not supported for this type.");
  }
  virtual void solve(int nrhs, scalar_t* b, int ldb) const {
    DenseMatrixWrapper<scalar_t> B(rows(), nrhs, b, ldb);
    solve(B);
  }
  virtual void shift(scalar_t s) { throw std::invalid_argument("Operation shift not supported for this type."); }
  // 1-d block row distribution of the MPI formats (structured/StructuredMatrix.hpp:262-320): the formats built here are
  // sequential objects (one process per GPU shares a matrix through the C interface, not through this class), so the local
  // range accessors throw like the reference's defaults and dist() is the single block {0, rows()}
  virtual std::size_t local_rows() const { throw std::invalid_argument("1d block row distribution not supported for this format."); }
  virtual std::size_t begin_row() const { throw std::invalid_argument("1d block row distribution not supported for this format."); }
  virtual std::size_t end_row() const { throw std::invalid_argument("1d block row distribution not supported for this format."); }
  virtual const std::vector<int>& dist() const {
    dist_ = {0, int(rows())};
    return dist_;
  }
  virtual const std::vector<int>& rdist() const { return dist(); }
  virtual const std::vector<int>& cdist() const { return dist(); }

 private:
  mutable std::vector<int> dist_;
};

// (structured/StructuredMatrix.cpp:572-605: HSS and BLR do not support matrix-free compression -- std::invalid_argument, as in
//  the reference; the formats that do, HODLR / HODBF / BUTTERFLY / LR, are out of scope)
template <typename scalar_t>
std::unique_ptr<StructuredMatrix<scalar_t>> construct_matrix_free(int rows, int cols, const mult_t<scalar_t>& Amult,
                                                                  const StructuredOptions<scalar_t>& opts,
                                                                  const ClusterTree* row_tree = nullptr,
                                                                  const ClusterTree* col_tree = nullptr) {
  (void)rows; (void)cols; (void)Amult; (void)row_tree; (void)col_tree;
  switch (opts.type()) {
    case Type::HSS: throw std::invalid_argument("Type HSS does not support matrix-free compression.");
    case Type::BLR: throw std::invalid_argument("Type BLR does not support matrix-free compression.");
    default: throw std::invalid_argument("construct_matrix_free: this structured format is not part of this library (HSS and BLR are)");
  }
}

template <typename scalar_t>
std::unique_ptr<StructuredMatrix<scalar_t>> construct_from_dense(
    const DenseMatrix<scalar_t>& A, const StructuredOptions<scalar_t>& opts, const ClusterTree* row_tree = nullptr,
    const ClusterTree* col_tree = nullptr, const admissibility_t* adm = nullptr);

template <typename scalar_t>
std::unique_ptr<StructuredMatrix<scalar_t>> construct_from_dense(
    int rows, int cols, const scalar_t* A, int ldA, const StructuredOptions<scalar_t>& opts,
    const ClusterTree* row_tree = nullptr, const ClusterTree* col_tree = nullptr, const admissibility_t* adm = nullptr);

template <typename scalar_t>
std::unique_ptr<StructuredMatrix<scalar_t>> construct_from_elements(
    int rows, int cols, const extract_block_t<scalar_t>& A, const StructuredOptions<scalar_t>& opts,
    const ClusterTree* row_tree = nullptr, const ClusterTree* col_tree = nullptr, const admissibility_t* adm = nullptr,
    const DenseMatrix<scalar_t>* p = nullptr);

template <typename scalar_t>
std::unique_ptr<StructuredMatrix<scalar_t>> construct_from_elements(
    int rows, int cols, const extract_t<scalar_t>& A, const StructuredOptions<scalar_t>& opts,
    const ClusterTree* row_tree = nullptr, const ClusterTree* col_tree = nullptr, const admissibility_t* adm = nullptr,
    const DenseMatrix<scalar_t>* p = nullptr);

template <typename scalar_t>
std::unique_ptr<StructuredMatrix<scalar_t>> construct_and_factor_from_dense(
    const DenseMatrix<scalar_t>& A, const StructuredOptions<scalar_t>& opts, const ClusterTree* row_tree = nullptr,
    const ClusterTree* col_tree = nullptr, const admissibility_t* adm = nullptr);

template <typename scalar_t>
std::unique_ptr<StructuredMatrix<scalar_t>> construct_and_factor_from_elements(
    int rows, int cols, const extract_block_t<scalar_t>& A, const StructuredOptions<scalar_t>& opts,
    const ClusterTree* row_tree = nullptr, const ClusterTree* col_tree = nullptr, const admissibility_t* adm = nullptr,
    const DenseMatrix<scalar_t>* p = nullptr);

template <typename scalar_t>
std::unique_ptr<StructuredMatrix<scalar_t>> construct_partially_matrix_free(
    int rows, int cols, const mult_t<scalar_t>& Amult, const extract_block_t<scalar_t>& Aelem,
    const StructuredOptions<scalar_t>& opts, const ClusterTree* row_tree = nullptr, const ClusterTree* col_tree = nullptr);

template <typename scalar_t>
std::unique_ptr<StructuredMatrix<scalar_t>> construct_partially_matrix_free(
    int rows, int cols, const mult_t<scalar_t>& Amult, const extract_t<scalar_t>& Aelem,
    const StructuredOptions<scalar_t>& opts, const ClusterTree* row_tree = nullptr, const ClusterTree* col_tree = nullptr);

// extension: A already resident in HBM (column-major device pointer); nothing is copied
std::unique_ptr<StructuredMatrix<double>> construct_from_dense_device(
    int rows, int cols, const double* dA, long long ldA, const StructuredOptions<double>& opts,
    const ClusterTree* row_tree = nullptr);

}  // namespace structured
}  // namespace strumpack
