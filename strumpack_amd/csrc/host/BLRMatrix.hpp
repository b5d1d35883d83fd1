// strumpack::BLR::BLRMatrix<double>: the reference's block low-rank matrix class (BLR/BLRMatrix.hpp:68-330) for the dense
// slice of SURVEY.md section 8(f2) -- compress / mult, compress_and_factor / solve -- on the MI355X (DeviceBLR).
#pragma once
#include <cassert>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "DenseMatrix.hpp"
#include "StructuredMatrix.hpp"
#include "StructuredOptions.hpp"
#include "blr_engine.hpp"

namespace strumpack {
namespace BLR {

// BLR/BLROptions.hpp:46-146: the options the dense and the frontal paths read.  Defaults as there: rel_tol 1e-4, abs_tol
// 1e-12, leaf_size 256, max_rank 5000, RRQR tiles, weak admissibility, factorization algorithm RL, compression kernel half.
enum class LowRankAlgorithm { RRQR, ACA, BACA };
enum class Admissibility { STRONG, WEAK };
enum class BLRFactorAlgorithm { COLWISE, RL, LL, COMB, STAR };
enum class CompressionKernel { HALF, FULL };
inline std::string get_name(LowRankAlgorithm a) { return a == LowRankAlgorithm::RRQR ? "RRQR" : (a == LowRankAlgorithm::ACA ? "ACA" : "BACA"); }
inline std::string get_name(Admissibility a) { return a == Admissibility::STRONG ? "strong" : "weak"; }
inline std::string get_name(BLRFactorAlgorithm a) {
  switch (a) {
    case BLRFactorAlgorithm::COLWISE: return "COLWISE"; case BLRFactorAlgorithm::RL: return "RL"; case BLRFactorAlgorithm::LL: return "LL";
    case BLRFactorAlgorithm::COMB: return "Comb"; case BLRFactorAlgorithm::STAR: return "Star";
  }
  return "unknown";
}
inline std::string get_name(CompressionKernel a) { return a == CompressionKernel::FULL ? "full" : "half"; }
template <typename scalar_t> class BLROptions : public structured::StructuredOptions<scalar_t> {
 public:
  BLROptions() : structured::StructuredOptions<scalar_t>(structured::Type::BLR) {
    this->set_rel_tol(1e-4);
    this->set_abs_tol(1e-12);
    this->set_leaf_size(256);
    this->set_max_rank(5000);
  }
  BLROptions(const structured::StructuredOptions<scalar_t>& o) : structured::StructuredOptions<scalar_t>(o) { this->set_type(structured::Type::BLR); }
  void set_low_rank_algorithm(LowRankAlgorithm a) { lr_algo_ = a; }
  void set_admissibility(Admissibility a) { adm_ = a; }
  void set_BLR_factor_algorithm(BLRFactorAlgorithm a) { blr_algo_ = a; }
  void set_compression_kernel(CompressionKernel a) { crn_krnl_ = a; }
  void set_BACA_blocksize(int B) { assert(B > 0); BACA_blocksize_ = B; }
  LowRankAlgorithm low_rank_algorithm() const { return lr_algo_; }
  Admissibility admissibility() const { return adm_; }
  BLRFactorAlgorithm BLR_factor_algorithm() const { return blr_algo_; }
  CompressionKernel compression_kernel() const { return crn_krnl_; }
  int BACA_blocksize() const { return BACA_blocksize_; }
  // Which of the variants run (a selection that does not is refused where it would be used, not silently replaced):
  //  * RRQR and ACA tiles; BACA is not built.
  //  * RL, and LL: the same dense ("always into full rank") Schur updates in left-looking order, every tile compressed at the
  //    same point (BLR/BLRMatrix.cpp:838-990) -- the reference's own RL and LL runs agree bit for bit on the test fronts.
  //  * COMB / STAR (LUAR, :991-1140) schedule the SAME factorization differently: the low-rank updates of a tile are
  //    accumulated and recompressed before they are subtracted (a cache optimisation of the CPU code; the compression kernel
  //    option picks the recompression).  Here they run the RL schedule, and SAY SO (a warning on stderr, once per process:
  //    check_supported) -- dense updates are what the matrix cores are fast
  //    at -- and the result sits inside the spread the reference's own variants show against each other (tile ranks up
  //    to 2 apart on ~3 % of the tiles of the test fronts, Schur complements equal to the tolerance): fixtures of the
  //    reference's STAR and COMB runs are checked with RL's tolerances (tests/blr_cases.py).  COLWISE (the fronts'
  //    memory-saving column-wise mode, construct_and_partial_factor_col) is not built.
  void check_supported() const {
    if (lr_algo_ != LowRankAlgorithm::RRQR && lr_algo_ != LowRankAlgorithm::ACA)
      throw std::invalid_argument("BLR: RRQR and ACA tile compression are available (BACA is not)");
    if (blr_algo_ == BLRFactorAlgorithm::COLWISE)
      throw std::invalid_argument("BLR: the COLWISE factorization mode is not available (RL, LL, Comb and Star are)");
    if (blr_algo_ == BLRFactorAlgorithm::COMB || blr_algo_ == BLRFactorAlgorithm::STAR) {
      // said out loud, once per process: the selection changes nothing here
      static bool told = false;
      if (!told) {
        told = true;
        std::cerr << "# WARNING: --blr_factor_algorithm " << (blr_algo_ == BLRFactorAlgorithm::COMB ? "Comb" : "Star")
                  << ": the accumulate-and-recompress (LUAR) schedule is not built for the GPU; the factorization runs with dense\n"
                  << "#          Schur updates (the RL / LL result, inside the reference's own spread between its variants).\n";
      }
    }
  }
  // --blr_* flags of the reference (BLR/BLROptions.cpp:78-197)
  void set_from_command_line(int argc, const char* const* argv) override {
    for (int i = 1; i < argc; i++) {
      std::string v;
      using structured::detail::match_flag;
      if (match_flag(argc, argv, i, "blr_rel_tol", v, true)) this->set_rel_tol(std::atof(v.c_str()));
      else if (match_flag(argc, argv, i, "blr_abs_tol", v, true)) this->set_abs_tol(std::atof(v.c_str()));
      else if (match_flag(argc, argv, i, "blr_leaf_size", v, true)) this->set_leaf_size(std::atoi(v.c_str()));
      else if (match_flag(argc, argv, i, "blr_max_rank", v, true)) this->set_max_rank(std::atoi(v.c_str()));
      else if (match_flag(argc, argv, i, "blr_low_rank_algorithm", v, true)) {
        if (v == "RRQR") set_low_rank_algorithm(LowRankAlgorithm::RRQR);
        else if (v == "ACA") set_low_rank_algorithm(LowRankAlgorithm::ACA);
        else if (v == "BACA") set_low_rank_algorithm(LowRankAlgorithm::BACA);
        else std::cerr << "# WARNING: low-rank algorithm not recognized, use 'RRQR', 'ACA' or 'BACA'." << std::endl;
      } else if (match_flag(argc, argv, i, "blr_admissibility", v, true)) {
        if (v == "weak") set_admissibility(Admissibility::WEAK);
        else if (v == "strong") set_admissibility(Admissibility::STRONG);
        else std::cerr << "# WARNING: admisibility not recognized, use 'weak' or 'strong'." << std::endl;
      } else if (match_flag(argc, argv, i, "blr_BACA_blocksize", v, true)) set_BACA_blocksize(std::atoi(v.c_str()));
      else if (match_flag(argc, argv, i, "blr_factor_algorithm", v, true)) {
        if (v == "COLWISE") set_BLR_factor_algorithm(BLRFactorAlgorithm::COLWISE);
        else if (v == "RL") set_BLR_factor_algorithm(BLRFactorAlgorithm::RL);
        else if (v == "LL") set_BLR_factor_algorithm(BLRFactorAlgorithm::LL);
        else if (v == "Comb") set_BLR_factor_algorithm(BLRFactorAlgorithm::COMB);
        else if (v == "Star") set_BLR_factor_algorithm(BLRFactorAlgorithm::STAR);
        else std::cerr << "# WARNING: BLR algorithm not recognized, use 'COLWISE', 'RL', 'LL', 'Comb' or 'Star'." << std::endl;
      } else if (match_flag(argc, argv, i, "blr_compression_kernel", v, true)) {
        if (v == "full") set_compression_kernel(CompressionKernel::FULL);
        else if (v == "half") set_compression_kernel(CompressionKernel::HALF);
        else std::cerr << "# WARNING: compression kernel not recognized, use 'full' or 'half'." << std::endl;
      } else if (match_flag(argc, argv, i, "blr_verbose", v, false)) this->set_verbose(true);
      else if (match_flag(argc, argv, i, "blr_quiet", v, false)) this->set_verbose(false);
    }
  }
  void describe_options() const override {
    std::cout << "# BLR Options:\n#   --blr_rel_tol real_t (default " << this->rel_tol() << ")\n#   --blr_abs_tol real_t (default " << this->abs_tol()
              << ")\n#   --blr_leaf_size int (default " << this->leaf_size() << ")\n#   --blr_max_rank int (default " << this->max_rank()
              << ")\n#   --blr_low_rank_algorithm (default " << get_name(lr_algo_) << ")\n#      should be [RRQR|ACA|BACA]  (BACA: not available)\n"
              << "#   --blr_admissibility (default " << get_name(adm_) << ")\n#      should be one of [weak|strong]\n"
              << "#   --blr_factor_algorithm (default " << get_name(blr_algo_) << ")\n#      should be [COLWISE|RL|LL|Comb|Star]  (COLWISE: not available)\n"
              << "#   --blr_compression_kernel (default " << get_name(crn_krnl_) << ")\n#      should be [full|half]\n"
              << "#   --blr_BACA_blocksize int (default " << BACA_blocksize() << ")\n#   --blr_verbose or -v (default " << this->verbose()
              << ")\n#   --blr_quiet or -q (default " << !this->verbose() << ")\n#   --help or -h\n" << std::endl;
  }

 private:
  LowRankAlgorithm lr_algo_ = LowRankAlgorithm::RRQR;
  Admissibility adm_ = Admissibility::WEAK;
  BLRFactorAlgorithm blr_algo_ = BLRFactorAlgorithm::RL;
  CompressionKernel crn_krnl_ = CompressionKernel::HALF;
  int BACA_blocksize_ = 4;
};

template <typename scalar_t> class BLRMatrix;

template <> class BLRMatrix<double> : public structured::StructuredMatrix<double> {
 public:
  using DenseM_t = DenseMatrix<double>;
  using Opts_t = BLROptions<double>;
  using adm_t = DenseMatrix<bool>;

  BLRMatrix() {}
  BLRMatrix(std::size_t m, const std::vector<std::size_t>& rowtiles, std::size_t n, const std::vector<std::size_t>& coltiles)
      : m_(m), n_(n), rt_(rowtiles.begin(), rowtiles.end()), ct_(coltiles.begin(), coltiles.end()) {}

  std::size_t rows() const override { return m_; }
  std::size_t cols() const override { return n_; }
  std::size_t memory() const override { return nonzeros() * sizeof(double); }
  std::size_t nonzeros() const override {
    if (!eng_) return 0;
    if (!part_) return std::size_t(eng_->nonzeros());
    long long nz[3];
    eng_->front_nonzeros(nz);
    return std::size_t(nz[part_ - 1]);
  }
  std::size_t rank() const override {
    if (!eng_) return 0;
    if (!part_) return std::size_t(eng_->rank());
    // largest rank among this block's tiles of the front
    const int nt = eng_->rowblocks(), ns = eng_->sep_blocks();
    std::vector<int> rk((std::size_t)nt * nt);
    eng_->tile_ranks(rk.data());
    int r = 0;
    for (int j = 0; j < nt; j++)
      for (int i = 0; i < nt; i++) {
        const int blk = (i < ns && j < ns) ? 1 : (i < ns ? 2 : (j < ns ? 3 : 0));
        if (blk == part_) r = std::max(r, rk[(std::size_t)i + (std::size_t)j * nt]);
      }
    return std::size_t(r);
  }

  // ---- frontal matrices: BLRMatrix::construct_and_partial_factor(A11, A12, A21, A22, B11, B12, B21, tiles1, tiles2,
  // admissible, opts) (BLR/BLRMatrix.hpp:186-194, BLR/BLRMatrix.cpp:740-1037; caller sparse/fronts/FrontBLR.cpp:419-432).
  // As in the reference A22 is overwritten with the Schur complement A22 - A21 A11^{-1} A12 and A11 / A12 / A21 are
  // released.  B11, B12, B21 come back as three views of ONE device-resident factorization (the elimination works on the
  // whole front as one array); they serve the calls the front's solve makes -- piv(), trsmLNU_gemm, gemm_trsmUNN
  // (FrontBLR.cpp:525-570) -- plus rows / cols / rank / nonzeros / memory.
  static void construct_and_partial_factor(DenseM_t& A11, DenseM_t& A12, DenseM_t& A21, DenseM_t& A22, BLRMatrix<double>& B11,
                                           BLRMatrix<double>& B12, BLRMatrix<double>& B21, const std::vector<std::size_t>& tiles1,
                                           const std::vector<std::size_t>& tiles2, const adm_t& admissible, const Opts_t& opts) {
    opts.check_supported();
    const std::size_t ds = A11.rows(), du = A12.cols();
    if (A11.cols() != ds || A12.rows() != ds || A21.rows() != du || A21.cols() != ds || (du && (A22.rows() != du || A22.cols() != du)))
      throw std::invalid_argument("construct_and_partial_factor: the four blocks do not form a square front");
    if (admissible.rows() != tiles1.size() || admissible.cols() != tiles1.size()) throw std::invalid_argument("Admissibility matrix wrong size");
    std::vector<int> tiles(tiles1.begin(), tiles1.end());
    tiles.insert(tiles.end(), tiles2.begin(), tiles2.end());
    BLREngineOptions e;
    e.rel_tol = opts.rel_tol(); e.abs_tol = opts.abs_tol(); e.max_rank = opts.max_rank(); e.verbose = opts.verbose();
    e.lr_algo = opts.low_rank_algorithm() == LowRankAlgorithm::ACA ? 1 : 0;
    if (const char* d = std::getenv("STRUMPACK_AMD_DEVICE")) e.device = std::atoi(d);
    std::shared_ptr<DeviceBLR> eng(new DeviceBLR(int(ds + du), tiles, int(ds + du), tiles, e));
    std::vector<char> adm(tiles1.size() * tiles1.size());
    for (std::size_t j = 0; j < tiles1.size(); j++)
      for (std::size_t i = 0; i < tiles1.size(); i++) adm[i + j * tiles1.size()] = admissible(i, j) ? 1 : 0;
    eng->partial_factor_host(int(tiles1.size()), A11.data(), A11.ld(), du ? A12.data() : nullptr, A12.ld(), du ? A21.data() : nullptr,
                             A21.ld(), du ? A22.data() : nullptr, A22.ld(), adm.data());
    if (du) eng->schur_host(A22.data(), A22.ld());
    B11 = BLRMatrix<double>(ds, tiles1, ds, tiles1);
    B12 = BLRMatrix<double>(ds, tiles1, du, tiles2);
    B21 = BLRMatrix<double>(du, tiles2, ds, tiles1);
    B11.eng_ = B12.eng_ = B21.eng_ = eng;
    B11.part_ = 1; B12.part_ = 2; B21.part_ = 3;
    // the row interchanges are applied inside the forward phase (trsmLNU_gemm); piv() is the identity in LAPACK's
    // convention, so that the reference's call sequence `bloc.laswp(F11.piv(), true); trsmLNU_gemm(...)` is unchanged
    B11.piv_.resize(ds);
    for (std::size_t i = 0; i < ds; i++) B11.piv_[i] = int(i) + 1;
    A11.clear(); A12.clear(); A21.clear();
  }
  const std::vector<int>& piv() const { return piv_; }
  // B1 <- L11^{-1} P B1,  B2 <- B2 - F2 B1   with F1 = B11, F2 = B21 of one front  (BLRMatrix.cpp:1552-1608)
  static void trsmLNU_gemm(const BLRMatrix<double>& F1, const BLRMatrix<double>& F2, DenseM_t& B1, DenseM_t& B2, int /*task_depth*/) {
    same_front(F1, 1, F2, 3);
    F1.eng_->front_forward(int(B1.cols()), B1.data(), B1.ld(), F2.rows() ? B2.data() : nullptr, B2.ld());
  }
  // B1 <- U11^{-1} (B1 - F2 B2)   with F1 = B11, F2 = B12 of one front  (BLRMatrix.cpp:1610-1665)
  static void gemm_trsmUNN(const BLRMatrix<double>& F1, const BLRMatrix<double>& F2, DenseM_t& B1, DenseM_t& B2, int /*task_depth*/) {
    same_front(F1, 1, F2, 2);
    F1.eng_->front_backward(int(B1.cols()), B1.data(), B1.ld(), F2.cols() ? B2.data() : nullptr, B2.ld());
  }
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
    if (part_ == 1) {   // B11 of a front: B11 \ b = the two solve phases with an empty update part
      const int du = eng_->upd_rows();
      DenseM_t z(du, b.cols());
      eng_->front_forward(int(b.cols()), b.data(), b.ld(), du ? z.data() : nullptr, z.ld());
      z.zero();
      eng_->front_backward(int(b.cols()), b.data(), b.ld(), du ? z.data() : nullptr, z.ld());
      return;
    }
    if (part_) throw std::logic_error("BLR solve: only the separator block B11 of a front can be solved with");
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
  // the tile partition (BLR/BLRMatrix.hpp: tilerows / tilecols / tileroff / tilecoff / maxtile*; rowblocks / colblocks above)
  std::size_t tilerows(std::size_t i) const { return std::size_t(rt_.at(i)); }
  std::size_t tilecols(std::size_t j) const { return std::size_t(ct_.at(j)); }
  std::size_t tileroff(std::size_t i) const { std::size_t o = 0; for (std::size_t k = 0; k < i; k++) o += std::size_t(rt_.at(k)); return o; }
  std::size_t tilecoff(std::size_t j) const { std::size_t o = 0; for (std::size_t k = 0; k < j; k++) o += std::size_t(ct_.at(k)); return o; }
  std::size_t maxtilerows() const { int m = 0; for (int t : rt_) m = std::max(m, t); return std::size_t(m); }
  std::size_t maxtilecols() const { int m = 0; for (int t : ct_) m = std::max(m, t); return std::size_t(m); }

 private:
  void need() const { if (!eng_) throw std::logic_error("BLR matrix has not been compressed"); }
  static void same_front(const BLRMatrix<double>& a, int pa, const BLRMatrix<double>& b, int pb) {
    if (!a.eng_ || a.eng_ != b.eng_ || a.part_ != pa || b.part_ != pb)
      throw std::invalid_argument("BLR front: the two blocks do not come from one construct_and_partial_factor call");
  }
  void make(const Opts_t& o) {
    BLREngineOptions e;
    e.rel_tol = o.rel_tol(); e.abs_tol = o.abs_tol(); e.max_rank = o.max_rank(); e.verbose = o.verbose();
    e.lr_algo = o.low_rank_algorithm() == LowRankAlgorithm::ACA ? 1 : 0;
    if (const char* d = std::getenv("STRUMPACK_AMD_DEVICE")) e.device = std::atoi(d);
    eng_.reset(new DeviceBLR(int(m_), rt_, int(n_), ct_, e));
    part_ = 0;
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
  std::shared_ptr<DeviceBLR> eng_;
  int part_ = 0;            // 0: a matrix of its own; 1 / 2 / 3: B11 / B12 / B21 of a front
  std::vector<int> piv_;
};

}  // namespace BLR
}  // namespace strumpack
