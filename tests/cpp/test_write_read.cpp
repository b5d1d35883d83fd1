// HSSMatrix::write / HSSMatrix::read round trip (reference API: HSS/HSSMatrix.hpp:499-509): the matrix read back
// applies, factors and solves like the one written.
#include <cmath>
#include <cstdio>
#include <iostream>

#include "HSSMatrix.hpp"

using namespace strumpack;

int main(int argc, char* argv[]) {
  int n = argc > 1 ? std::atoi(argv[1]) : 300;
  std::string fname = argc > 2 ? argv[2] : "/tmp/hss_write_read_test.bin";
  DenseMatrix<double> A(n, n);
  for (int j = 0; j < n; j++)
    for (int i = 0; i < n; i++) A(i, j) = (i == j) ? 1. : 1. / (1 + std::abs(i - j));
  HSS::HSSOptions<double> opts;
  opts.set_leaf_size(32);
  opts.set_rel_tol(1e-6);
  opts.set_d0(32);
  opts.set_dd(16);
  HSS::HSSMatrix<double> H(A, opts);
  if (!H.is_compressed()) { std::cout << "compression failed" << std::endl; return 1; }
  H.write(fname);
  auto G = HSS::HSSMatrix<double>::read(fname);
  std::remove(fname.c_str());
  if (G.rows() != H.rows() || G.rank() != H.rank() || G.levels() != H.levels() || G.memory() != H.memory() || !G.is_compressed()) {
    std::cout << "ERROR: header of the matrix read back differs" << std::endl;
    return 1;
  }
  DenseMatrix<double> X(n, 3);
  for (int j = 0; j < 3; j++)
    for (int i = 0; i < n; i++) X(i, j) = std::sin(0.37 * i + j);
  auto Y1 = H.apply(X), Y2 = G.apply(X);
  Y2.scaled_add(-1., Y1);
  std::cout << "# ||G*X - H*X||_F/||H*X||_F = " << Y2.normF() / Y1.normF() << std::endl;
  if (Y2.normF() != 0.) { std::cout << "ERROR: the matrix read back applies differently" << std::endl; return 1; }
  G.factor();
  DenseMatrix<double> B(Y1);
  G.solve(B);   // B = G \ (H X) == X
  B.scaled_add(-1., X);
  std::cout << "# ||G\\(H*X) - X||_F/||X||_F = " << B.normF() / X.normF() << std::endl;
  if (B.normF() / X.normF() > 1e-10) { std::cout << "ERROR: solve with the matrix read back" << std::endl; return 1; }
  std::cout << "# exiting" << std::endl;
  return 0;
}
