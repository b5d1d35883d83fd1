// Factories of the structured facade: HSS dispatch (reference structured/StructuredMatrix.cpp:54-76
// dense, :203-273 elements, :637-651 partially matrix-free).
#include "StructuredMatrix.hpp"

#include "BLRMatrix.hpp"
#include "HSSMatrix.hpp"
#include "HSSMatrixPromoted.hpp"

namespace strumpack {
namespace structured {

namespace {
void require_hss(Type t, int rows, int cols) {
  if (t != Type::HSS)
    throw std::invalid_argument("Structured type " + get_name(t) + " is not available in this build (HSS and dense BLR only).");
  if (rows != cols) throw std::invalid_argument("HSS compression only supported for square matrices.");
}
HSS::HSSMatrix<double>* new_hss(int n, const StructuredOptions<double>& opts, const ClusterTree* row_tree,
                                HSS::HSSOptions<double>& hss_opts) {
  hss_opts = HSS::HSSOptions<double>(opts);
  return row_tree ? new HSS::HSSMatrix<double>(*row_tree, hss_opts) : new HSS::HSSMatrix<double>(n, n, hss_opts);
}
}  // namespace

// BLR of a dense matrix (reference structured/StructuredMatrix.cpp:77-97, :408-430): tiles = the leaves of the cluster
// trees (default: bisection refined to leaf_size), every tile admissible unless an admissibility matrix says otherwise
std::unique_ptr<StructuredMatrix<double>> blr_from_dense(const DenseMatrix<double>& A, const StructuredOptions<double>& opts,
                                                         const ClusterTree* row_tree, const ClusterTree* col_tree,
                                                         const admissibility_t* adm, bool factor) {
  auto row_leafs = row_tree ? row_tree->leaf_sizes<std::size_t>() : ClusterTree(int(A.rows())).refine(opts.leaf_size()).leaf_sizes<std::size_t>();
  auto col_leafs = col_tree ? col_tree->leaf_sizes<std::size_t>() : ClusterTree(int(A.cols())).refine(opts.leaf_size()).leaf_sizes<std::size_t>();
  std::unique_ptr<BLR::BLRMatrix<double>> B(new BLR::BLRMatrix<double>(A.rows(), row_leafs, A.cols(), col_leafs));
  BLR::BLROptions<double> bo(opts);
  DenseMatrix<bool> ad(row_leafs.size(), col_leafs.size());
  if (adm) {
    if (adm->rows() != row_leafs.size() || adm->cols() != col_leafs.size()) throw std::invalid_argument("Admissibility matrix wrong size");
    ad = *adm;
  } else ad.fill(true);
  if (factor) B->compress_and_factor(A, ad, bo);
  else B->compress(A, ad, bo);
  return std::unique_ptr<StructuredMatrix<double>>(B.release());
}

template <>
std::unique_ptr<StructuredMatrix<double>> construct_from_dense(const DenseMatrix<double>& A, const StructuredOptions<double>& opts,
                                                               const ClusterTree* row_tree, const ClusterTree* col_tree, const admissibility_t* adm) {
  if (opts.type() == Type::BLR) return blr_from_dense(A, opts, row_tree, col_tree, adm, false);
  require_hss(opts.type(), int(A.rows()), int(A.cols()));
  HSS::HSSOptions<double> ho;
  std::unique_ptr<HSS::HSSMatrix<double>> H(new_hss(int(A.rows()), opts, row_tree, ho));
  H->compress(A, ho);
  return std::unique_ptr<StructuredMatrix<double>>(H.release());
}

template <>
std::unique_ptr<StructuredMatrix<double>> construct_from_dense(int rows, int cols, const double* A, int ldA,
                                                               const StructuredOptions<double>& opts, const ClusterTree* row_tree,
                                                               const ClusterTree* col_tree, const admissibility_t* adm) {
  auto M = ConstDenseMatrixWrapper<double>(rows, cols, A, ldA);
  return construct_from_dense<double>(M, opts, row_tree, col_tree, adm);
}

std::unique_ptr<StructuredMatrix<double>> construct_from_dense_device(int rows, int cols, const double* dA, long long ldA,
                                                                      const StructuredOptions<double>& opts, const ClusterTree* row_tree) {
  require_hss(opts.type(), rows, cols);
  HSS::HSSOptions<double> ho;
  std::unique_ptr<HSS::HSSMatrix<double>> H(new_hss(rows, opts, row_tree, ho));
  H->compress_device(dA, ldA, ho);
  return std::unique_ptr<StructuredMatrix<double>>(H.release());
}

template <>
std::unique_ptr<StructuredMatrix<double>> construct_from_elements(int rows, int cols, const extract_block_t<double>& A,
                                                                  const StructuredOptions<double>& opts, const ClusterTree* row_tree,
                                                                  const ClusterTree* col_tree, const admissibility_t* adm, const DenseMatrix<double>*) {
  if (opts.type() == Type::BLR) {
    // BLRMatrix::compress(Aelem, adm, opts) (StructuredMatrix.cpp:274-295, BLR/BLRMatrix.cpp:102-111) evaluates every tile in
    // full before it compresses it: here the tiles of a block column are evaluated on the host threads into the dense image
    // and the dense route takes over
    DenseMatrix<double> D(rows, cols);
    const int bs = std::max(1, opts.leaf_size());
    std::vector<std::size_t> I(rows);
    for (int i = 0; i < rows; i++) I[i] = std::size_t(i);
    for (int j0 = 0; j0 < cols; j0 += bs) {
      const int nb = std::min(bs, cols - j0);
      std::vector<std::size_t> J(nb);
      for (int j = 0; j < nb; j++) J[j] = std::size_t(j0 + j);
      DenseMatrix<double> B(rows, nb);
      A(I, J, B);
      for (int j = 0; j < nb; j++) for (int i = 0; i < rows; i++) D(i, j0 + j) = B(i, j);
    }
    return blr_from_dense(D, opts, row_tree, col_tree, adm, false);
  }
  require_hss(opts.type(), rows, cols);
  // The reference samples A on the fly in B x B tiles and never stores it (StructuredMatrix.cpp:214-262); here the column
  // blocks are evaluated tile by tile on the host threads and streamed through the device, uploads overlapped with the
  // sketch GEMMs (DeviceHSS::HostBlockSource) -- no N^2 image on either side.
  HSS::HSSOptions<double> ho;
  std::unique_ptr<HSS::HSSMatrix<double>> H(new_hss(rows, opts, row_tree, ho));
  H->compress_from_elements(A, ho);
  return std::unique_ptr<StructuredMatrix<double>>(H.release());
}

template <>
std::unique_ptr<StructuredMatrix<double>> construct_from_elements(int rows, int cols, const extract_t<double>& A,
                                                                  const StructuredOptions<double>& opts, const ClusterTree* row_tree,
                                                                  const ClusterTree* col_tree, const admissibility_t* adm,
                                                                  const DenseMatrix<double>* p) {
  extract_block_t<double> blk = [&A](const std::vector<std::size_t>& I, const std::vector<std::size_t>& J, DenseMatrix<double>& B) {
    for (std::size_t j = 0; j < J.size(); j++)
      for (std::size_t i = 0; i < I.size(); i++) B(i, j) = A(I[i], J[j]);
  };
  return construct_from_elements<double>(rows, cols, blk, opts, row_tree, col_tree, adm, p);
}

template <>
std::unique_ptr<StructuredMatrix<double>> construct_and_factor_from_dense(const DenseMatrix<double>& A, const StructuredOptions<double>& opts,
                                                                          const ClusterTree* row_tree, const ClusterTree* col_tree,
                                                                          const admissibility_t* adm) {
  if (opts.type() == Type::BLR) return blr_from_dense(A, opts, row_tree, col_tree, adm, true);
  auto S = construct_from_dense<double>(A, opts, row_tree, col_tree, adm);
  S->factor();
  return S;
}

template <>
std::unique_ptr<StructuredMatrix<double>> construct_and_factor_from_elements(int rows, int cols, const extract_block_t<double>& A,
                                                                             const StructuredOptions<double>& opts, const ClusterTree* row_tree,
                                                                             const ClusterTree* col_tree, const admissibility_t* adm,
                                                                             const DenseMatrix<double>* p) {
  auto S = construct_from_elements<double>(rows, cols, A, opts, row_tree, col_tree, adm, p);
  S->factor();
  return S;
}

template <>
std::unique_ptr<StructuredMatrix<double>> construct_partially_matrix_free(int rows, int cols, const mult_t<double>& Amult,
                                                                          const extract_block_t<double>& Aelem,
                                                                          const StructuredOptions<double>& opts, const ClusterTree* row_tree,
                                                                          const ClusterTree* col_tree) {
  // (StructuredMatrix.cpp:651-652: BLR has no use for the product and is built from the elements)
  if (opts.type() == Type::BLR) return construct_from_elements<double>(rows, cols, Aelem, opts, row_tree, col_tree, nullptr, nullptr);
  require_hss(opts.type(), rows, cols);
  HSS::HSSOptions<double> ho;
  std::unique_ptr<HSS::HSSMatrix<double>> H(new_hss(rows, opts, row_tree, ho));
  // adaptor of StructuredMatrix.cpp:643-647
  auto sample = [&Amult](DenseMatrix<double>& Rr, DenseMatrix<double>& Rc, DenseMatrix<double>& Sr, DenseMatrix<double>& Sc) {
    Amult(Trans::N, Rr, Sr);
    Amult(Trans::C, Rc, Sc);
  };
  H->compress(sample, Aelem, ho);
  return std::unique_ptr<StructuredMatrix<double>>(H.release());
}

template <>
std::unique_ptr<StructuredMatrix<double>> construct_partially_matrix_free(int rows, int cols, const mult_t<double>& Amult,
                                                                          const extract_t<double>& Aelem, const StructuredOptions<double>& opts,
                                                                          const ClusterTree* row_tree, const ClusterTree* col_tree) {
  extract_block_t<double> blk = [&Aelem](const std::vector<std::size_t>& I, const std::vector<std::size_t>& J, DenseMatrix<double>& B) {
    for (std::size_t j = 0; j < J.size(); j++)
      for (std::size_t i = 0; i < I.size(); i++) B(i, j) = Aelem(I[i], J[j]);
  };
  return construct_partially_matrix_free<double>(rows, cols, Amult, blk, opts, row_tree, col_tree);
}

// ---- float / complex instantiations (reference: explicit instantiations at the end of structured/StructuredMatrix.cpp);
// computed by the double-precision device engine through HSSMatrixPromoted.hpp
#define SPX_PROMOTED_FACTORIES(T)                                                                                          \
  template <>                                                                                                              \
  std::unique_ptr<StructuredMatrix<T>> construct_from_dense(const DenseMatrix<T>& A, const StructuredOptions<T>& opts,     \
                                                            const ClusterTree* row_tree, const ClusterTree*,               \
                                                            const admissibility_t*) {                                      \
    if (opts.type() != Type::HSS)                                                                                          \
      throw std::invalid_argument("Structured type " + get_name(opts.type()) + " is not available in this build (HSS hot path only)."); \
    if (A.rows() != A.cols()) throw std::invalid_argument("HSS compression only supported for square matrices.");         \
    HSS::HSSOptions<double> ho{to_double_options(opts)};                                                           \
    std::unique_ptr<HSS::HSSMatrix<T>> H(row_tree ? new HSS::HSSMatrix<T>(*row_tree, ho)                                   \
                                                  : new HSS::HSSMatrix<T>(A.rows(), A.cols(), ho));                        \
    H->compress(A, ho);                                                                                                    \
    return std::unique_ptr<StructuredMatrix<T>>(H.release());                                                              \
  }                                                                                                                        \
  template <>                                                                                                              \
  std::unique_ptr<StructuredMatrix<T>> construct_from_dense(int rows, int cols, const T* A, int ldA,                       \
                                                            const StructuredOptions<T>& opts, const ClusterTree* row_tree, \
                                                            const ClusterTree* col_tree, const admissibility_t* adm) {     \
    auto M = ConstDenseMatrixWrapper<T>(rows, cols, A, ldA);                                                               \
    return construct_from_dense<T>(M, opts, row_tree, col_tree, adm);                                                      \
  }                                                                                                                        \
  template <>                                                                                                              \
  std::unique_ptr<StructuredMatrix<T>> construct_from_elements(int rows, int cols, const extract_block_t<T>& A,            \
                                                               const StructuredOptions<T>& opts, const ClusterTree* row_tree, \
                                                               const ClusterTree*, const admissibility_t*,                \
                                                               const DenseMatrix<T>*) { \
    if (opts.type() != Type::HSS)                                                                                          \
      throw std::invalid_argument("Structured type " + get_name(opts.type()) + " is not available in this build (HSS hot path only)."); \
    if (rows != cols) throw std::invalid_argument("HSS compression only supported for square matrices.");                 \
    HSS::HSSOptions<double> ho{to_double_options(opts)};                                                           \
    std::unique_ptr<HSS::HSSMatrix<T>> H(row_tree ? new HSS::HSSMatrix<T>(*row_tree, ho) : new HSS::HSSMatrix<T>(rows, cols, ho)); \
    H->compress(A, ho);                                                                                                    \
    return std::unique_ptr<StructuredMatrix<T>>(H.release());                                                              \
  }                                                                                                                        \
  template <>                                                                                                              \
  std::unique_ptr<StructuredMatrix<T>> construct_from_elements(int rows, int cols, const extract_t<T>& A,                  \
                                                               const StructuredOptions<T>& opts, const ClusterTree* row_tree, \
                                                               const ClusterTree* col_tree, const admissibility_t* adm,   \
                                                               const DenseMatrix<T>* p) { \
    extract_block_t<T> blk = [&A](const std::vector<std::size_t>& I, const std::vector<std::size_t>& J, DenseMatrix<T>& B) { \
      for (std::size_t j = 0; j < J.size(); j++)                                                                           \
        for (std::size_t i = 0; i < I.size(); i++) B(i, j) = A(I[i], J[j]);                                                \
    };                                                                                                                     \
    return construct_from_elements<T>(rows, cols, blk, opts, row_tree, col_tree, adm, p);                                  \
  }
SPX_PROMOTED_FACTORIES(float)
SPX_PROMOTED_FACTORIES(std::complex<float>)
SPX_PROMOTED_FACTORIES(std::complex<double>)
#undef SPX_PROMOTED_FACTORIES

}  // namespace structured
}  // namespace strumpack
