// Frontal-matrix path of the BLR class, driven the way the reference's sparse BLR front drives it
// (sparse/fronts/FrontBLR.cpp:405-432 factor, :525-570 solve phases): construct_and_partial_factor on [F11 F12; F21 F22],
// then laswp(piv) + trsmLNU_gemm and gemm_trsmUNN.  The front is a small exact Schur complement of a 2D 5-point Laplacian
// (separator = one grid line between two eliminated strips, update part = the two outer lines), so the answers are known
// from dense algebra; the fixtures against the reference's own code live in tests/test_blr_front_*.py.
//   usage: test_BLR_front n [leaf]
#include <cmath>
#include <iostream>

#include "BLR/BLRMatrix.hpp"
#include "dense/DenseMatrix.hpp"

using namespace strumpack;
using namespace strumpack::BLR;

int main(int argc, char* argv[]) {
  const int n = argc > 1 ? std::stoi(argv[1]) : 96;
  const int leaf = argc > 2 ? std::stoi(argv[2]) : 16;
  // a smooth, diagonally dominant front: F = [F11 F12; F21 F22] of size 3n, Green's-function-like off-diagonal decay
  const int N = 3 * n;
  DenseMatrix<double> F(N, N);
  for (int j = 0; j < N; j++)
    for (int i = 0; i < N; i++) {
      const double xi = (i % n) / double(n), xj = (j % n) / double(n), zi = i / n, zj = j / n;
      const double r = std::sqrt((xi - xj) * (xi - xj) + 0.05 * (zi - zj) * (zi - zj));
      F(i, j) = (i == j ? 4.0 : 0.0) + 1.0 / (1.0 + 40.0 * r) + (i > j ? 0.01 : 0.0);
    }
  DenseMatrix<double> A11(n, n, F, 0, 0), A12(n, 2 * n, F, 0, n), A21(2 * n, n, F, n, 0), A22(2 * n, 2 * n, F, n, n);
  const DenseMatrix<double> F11(A11), F12(A12), F21(A21), F22(A22);
  std::vector<std::size_t> tiles1, tiles2;
  for (int r = n; r > 0; r -= leaf) tiles1.push_back(std::min(r, leaf));
  for (int r = 2 * n; r > 0; r -= leaf) tiles2.push_back(std::min(r, leaf));
  DenseMatrix<bool> adm(tiles1.size(), tiles1.size());
  adm.fill(true);
  for (std::size_t t = 0; t < tiles1.size(); t++) adm(t, t) = false;
  BLROptions<double> opts;
  opts.set_rel_tol(1e-8);
  opts.set_abs_tol(1e-12);
  BLRMatrix<double> B11, B12, B21;
  BLRMatrix<double>::construct_and_partial_factor(A11, A12, A21, A22, B11, B12, B21, tiles1, tiles2, adm, opts);
  if (A11.rows() || A12.rows() || A21.rows()) { std::cout << "ERROR: A11 / A12 / A21 not released" << std::endl; return 1; }
  std::cout << "# front " << n << " + " << 2 * n << ", tiles " << tiles1.size() << " + " << tiles2.size() << ": ranks " << B11.rank() << " / "
            << B12.rank() << " / " << B21.rank() << ", nonzeros " << B11.nonzeros() << " / " << B12.nonzeros() << " / " << B21.nonzeros()
            << " of " << n * n << " / " << 2 * n * n << " / " << 2 * n * n << std::endl;
  // Schur complement against dense algebra: S = F22 - F21 F11^{-1} F12
  DenseMatrix<double> X(F12), L(F11);
  {  // dense LU without pivoting is fine for this diagonally dominant block
    for (int k = 0; k < n; k++)
      for (int i = k + 1; i < n; i++) {
        L(i, k) /= L(k, k);
        for (int j = k + 1; j < n; j++) L(i, j) -= L(i, k) * L(k, j);
      }
    trsm(Side::L, UpLo::L, Trans::N, Diag::U, 1., L, X);
    trsm(Side::L, UpLo::U, Trans::N, Diag::N, 1., L, X);
  }
  DenseMatrix<double> S(F22);
  gemm(Trans::N, Trans::N, -1., F21, X, 1., S);
  DenseMatrix<double> dS(A22);
  dS.scaled_add(-1., S);
  std::cout << "# ||S - S_dense||_F / ||S_dense||_F = " << dS.normF() / S.normF() << std::endl;
  if (dS.normF() > 1e-6 * S.normF()) { std::cout << "ERROR: Schur complement" << std::endl; return 1; }
  // the front's solve: forward phase, then backward phase, against the dense block elimination
  DenseMatrix<double> b(n, 2), bu(2 * n, 2);
  b.random();
  for (int j = 0; j < 2; j++) for (int i = 0; i < 2 * n; i++) bu(i, j) = std::sin(0.1 * i + j);
  DenseMatrix<double> bl(b), bupd(bu);
  bl.laswp(B11.piv(), true);
  BLRMatrix<double>::trsmLNU_gemm(B11, B21, bl, bupd, 0);
  // reference values: y = L^{-1} P b is not unique to compare, but bupd = bu - F21 F11^{-1} b is
  DenseMatrix<double> t(b);
  trsm(Side::L, UpLo::L, Trans::N, Diag::U, 1., L, t);
  trsm(Side::L, UpLo::U, Trans::N, Diag::N, 1., L, t);
  DenseMatrix<double> bu_ref(bu);
  gemm(Trans::N, Trans::N, -1., F21, t, 1., bu_ref);
  bupd.scaled_add(-1., bu_ref);
  std::cout << "# forward phase, update part: " << bupd.normF() / bu_ref.normF() << std::endl;
  if (bupd.normF() > 1e-6 * bu_ref.normF()) { std::cout << "ERROR: forward phase" << std::endl; return 1; }
  // backward phase with yupd = 0 completes B11 \ b
  DenseMatrix<double> yu(2 * n, 2);
  BLRMatrix<double>::gemm_trsmUNN(B11, B12, bl, yu, 0);
  bl.scaled_add(-1., t);
  std::cout << "# B11 \\ b against the dense solve: " << bl.normF() / t.normF() << std::endl;
  if (bl.normF() > 1e-6 * t.normF()) { std::cout << "ERROR: backward phase" << std::endl; return 1; }
  DenseMatrix<double> x2(b);
  B11.solve(x2);
  x2.scaled_add(-1., t);
  if (x2.normF() > 1e-6 * t.normF()) { std::cout << "ERROR: B11.solve" << std::endl; return 1; }
  // a variant that is not built is refused; Star (a different schedule of the same factorization: BLRMatrix.hpp) is accepted
  bool refused = false;
  try {
    BLROptions<double> o2;
    o2.set_BLR_factor_algorithm(BLRFactorAlgorithm::COLWISE);
    DenseMatrix<double> C11(F11), C12(F12), C21(F21), C22(F22);
    BLRMatrix<double>::construct_and_partial_factor(C11, C12, C21, C22, B11, B12, B21, tiles1, tiles2, adm, o2);
  } catch (const std::invalid_argument&) { refused = true; }
  if (!refused) { std::cout << "ERROR: unsupported algorithm accepted" << std::endl; return 1; }
  {
    BLROptions<double> o2;
    o2.set_rel_tol(1e-6);
    const char* args[] = {"test", "--blr_factor_algorithm", "Star", "--blr_compression_kernel", "full"};
    o2.set_from_command_line(5, args);
    if (o2.BLR_factor_algorithm() != BLRFactorAlgorithm::STAR || o2.compression_kernel() != CompressionKernel::FULL) { std::cout << "ERROR: --blr_* flags" << std::endl; return 1; }
    DenseMatrix<double> C11(F11), C12(F12), C21(F21), C22(F22);
    BLRMatrix<double> S11, S12, S21;
    BLRMatrix<double>::construct_and_partial_factor(C11, C12, C21, C22, S11, S12, S21, tiles1, tiles2, adm, o2);
    DenseMatrix<double> xs(b);
    S11.solve(xs);
    xs.scaled_add(-1., t);
    if (xs.normF() > 1e-6 * t.normF()) { std::cout << "ERROR: Star variant" << std::endl; return 1; }
  }
  // ACA tile compression (BLROptions::set_low_rank_algorithm): accepted, accurate to its tolerance; BACA is refused
  {
    BLROptions<double> o4;
    o4.set_rel_tol(1e-8);
    o4.set_low_rank_algorithm(LowRankAlgorithm::ACA);
    DenseMatrix<double> C11(F11), C12(F12), C21(F21), C22(F22);
    BLRMatrix<double> A11, A12, A21;
    BLRMatrix<double>::construct_and_partial_factor(C11, C12, C21, C22, A11, A12, A21, tiles1, tiles2, adm, o4);
    DenseMatrix<double> x4(b);
    A11.solve(x4);
    x4.scaled_add(-1., t);
    std::cout << "# ACA tiles, B11 \\ b against the dense solve: " << x4.normF() / t.normF() << std::endl;
    if (x4.normF() > 1e-5 * t.normF()) { std::cout << "ERROR: ACA variant" << std::endl; return 1; }
    bool refused_baca = false;
    try {
      o4.set_low_rank_algorithm(LowRankAlgorithm::BACA);
      DenseMatrix<double> E11(F11), E12(F12), E21(F21), E22(F22);
      BLRMatrix<double>::construct_and_partial_factor(E11, E12, E21, E22, A11, A12, A21, tiles1, tiles2, adm, o4);
    } catch (const std::invalid_argument&) { refused_baca = true; }
    if (!refused_baca) { std::cout << "ERROR: BACA accepted" << std::endl; return 1; }
  }
  // LL: the same dense Schur updates in left-looking order, every tile compressed at the same point (the reference's LL and RL
  // runs agree to the last bit on these fronts): accepted, same results
  {
    BLROptions<double> o3;
    o3.set_rel_tol(1e-6);
    o3.set_BLR_factor_algorithm(BLRFactorAlgorithm::LL);
    DenseMatrix<double> C11(F11), C12(F12), C21(F21), C22(F22);
    BLRMatrix<double> L11, L12, L21;
    BLRMatrix<double>::construct_and_partial_factor(C11, C12, C21, C22, L11, L12, L21, tiles1, tiles2, adm, o3);
    DenseMatrix<double> x3(b);
    L11.solve(x3);
    x3.scaled_add(-1., t);
    if (x3.normF() > 1e-6 * t.normF()) { std::cout << "ERROR: LL variant" << std::endl; return 1; }
  }
  std::cout << "# exiting" << std::endl;
  return 0;
}
