// strumpack::BLR::BLRMatrix<double>: the reference's block low-rank matrix class (BLR/BLRMatrix.hpp:68-330) for the dense
// slice of SURVEY.md section 8(f2) -- compress / mult, compress_and_factor / solve -- on the MI355X (DeviceBLR).
#pragma once
#include <memory>
#include <vector>

#include "DenseMatrix.hpp"
#include "StructuredMatrix.hpp"
#include "StructuredOptions.hpp"
#include "blr_engine.hpp"

namespace strumpack {
namespace BLR {

// the subset of BLR/BLROptions.hpp:100-140 that the dense path reads (defaults as there: RRQR tiles, algorithm RL)
template <typename scalar_t> class BLROptions : public structured::StructuredOptions<scalar_t> {
 public:
  BLROptions() : structured::StructuredOptions<scalar_t>(structured::Type::BLR) {}
  BLROptions(const structured::StructuredOptions<scalar_t>& o) : structured::StructuredOptions<scalar_t>(o) {}
};

template <typename scalar_t> class BLRMatrix;

template <> class BLRMatrix<double> : public structured::StructuredMatrix<double> {
 public:
  using DenseM_t = DenseMatrix<double>;
  using Opts_t = BLROptions<double>;
  using adm_t = DenseMatrix<bool>;

  BLRMatrix(std::size_t m, const std::vector<std::size_t>& rowtiles, std::size_t n, const std::vector<std::size_t>& coltiles)
      : m_(m), n_(n), rt_(rowtiles.begin(), rowtiles.end()), ct_(coltiles.begin(), coltiles.end()) {}

  std::size_t rows() const override { return m_; }
  std::size_t cols() const override { return n_; }
  std::size_t memory() const override { return eng_ ? std::size_t(eng_->memory()) : 0; }
  std::size_t nonzeros() const override { return eng_ ? std::size_t(eng_->nonzeros()) : 0; }
  std::size_t rank() const override { return eng_ ? std::size_t(eng_->rank()) : 0; }
  std::size_t rowblocks() const { return rt_.size(); }
  std::size_t colblocks() const { return ct_.size(); }

  // BLRMatrix::compress (BLRMatrix.cpp:92-100)
  void compress(const DenseM_t& A, const adm_t& admissible, const Opts_t& opts) {
    make(opts);
    auto adm = flags(admissible);
    eng_->compress_host(A.data(), A.ld(), adm.data());
  }
  // BLRMatrix::compress_and_factor (BLRMatrix.cpp:114-243, algorithm RL)
  void compress_and_factor(const DenseM_t& A, const adm_t& admissible, const Opts_t& opts) {
    make(opts);
    auto adm = flags(admissible);
    eng_->compress_and_factor_host(A.data(), A.ld(), adm.data());
  }
  void mult(Trans op, const DenseM_t& x, DenseM_t& y) const override {
    need();
    eng_->mult(char(op), int(x.cols()), x.data(), x.ld(), y.data(), y.ld());
  }
  using structured::StructuredMatrix<double>::mult;
  void solve(DenseM_t& b) const override {
    need();
    eng_->solve(int(b.cols()), b.data(), b.ld());
  }
  using structured::StructuredMatrix<double>::solve;
  DenseM_t dense() const {
    need();
    DenseM_t A(m_, n_);
    eng_->dense(A.data(), A.ld());
    return A;
  }
  const DeviceBLR* engine() const { return eng_.get(); }

 private:
  void need() const { if (!eng_) throw std::logic_error("BLR matrix has not been compressed"); }
  void make(const Opts_t& o) {
    BLREngineOptions e;
    e.rel_tol = o.rel_tol(); e.abs_tol = o.abs_tol(); e.max_rank = o.max_rank(); e.verbose = o.verbose();
    if (const char* d = std::getenv("STRUMPACK_AMD_DEVICE")) e.device = std::atoi(d);
    eng_.reset(new DeviceBLR(int(m_), rt_, int(n_), ct_, e));
  }
  std::vector<char> flags(const adm_t& a) const {
    if (a.rows() != rt_.size() || a.cols() != ct_.size()) throw std::invalid_argument("Admissibility matrix wrong size");
    std::vector<char> f(rt_.size() * ct_.size());
    for (std::size_t j = 0; j < ct_.size(); j++)
      for (std::size_t i = 0; i < rt_.size(); i++) f[i + j * rt_.size()] = a(i, j) ? 1 : 0;
    return f;
  }
  std::size_t m_, n_;
  std::vector<int> rt_, ct_;
  std::unique_ptr<DeviceBLR> eng_;
};

}  // namespace BLR
}  // namespace strumpack
