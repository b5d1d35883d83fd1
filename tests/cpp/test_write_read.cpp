// HSSMatrix::write / HSSMatrix::read round trip (reference API: HSS/HSSMatrix.hpp:499-509): the matrix read back
// applies, factors and solves like the one written.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <sstream>
#include <iterator>
#include <vector>

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
  // damaged files must be refused by read() -- never loaded and then read out of bounds on the device: a truncated
  // file, and files whose payload was altered in place (a block count, a permutation entry)
  {
    std::ifstream in(fname, std::ios::binary);
    std::vector<char> img((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    in.close();
    auto refused = [&](const std::vector<char>& bytes, const char* what) {
      const std::string f2 = fname + ".bad";
      { std::ofstream o(f2, std::ios::binary | std::ios::trunc); o.write(bytes.data(), (std::streamsize)bytes.size()); }
      bool thrown = false;
      try { auto B = HSS::HSSMatrix<double>::read(f2); } catch (const std::exception&) { thrown = true; }
      std::remove(f2.c_str());
      if (!thrown) std::cout << "ERROR: a corrupt file (" << what << ") was accepted" << std::endl;
      return thrown;
    };
    bool ok = true;
    ok = refused(std::vector<char>(img.begin(), img.begin() + img.size() / 2), "truncated") && ok;
    // first node record: 16 bytes header, 13 ints node table, then the int64 element count of the D block (0 for the
    // root of a multi-level tree), then B01's count: claim one element fewer
    {
      auto b = img;
      long long cnt;
      const size_t off = 16 + 13 * sizeof(int) + sizeof(long long);   // count of B01 of the root
      std::memcpy(&cnt, b.data() + off, sizeof(cnt));
      if (cnt > 1) {
        // drop the last double of the block and fix the count: still parseable, but the size no longer matches the ranks
        cnt -= 1;
        std::memcpy(b.data() + off, &cnt, sizeof(cnt));
        b.erase(b.begin() + off + sizeof(long long) + cnt * sizeof(double), b.begin() + off + sizeof(long long) + (cnt + 1) * sizeof(double));
        ok = refused(b, "coupling block size") && ok;
      }
    }
    // a permutation entry out of range: find the second node's permU block = after its D / B01 / B10 / XU blocks
    {
      auto b = img;
      size_t off = 16;
      auto skip_node = [&](size_t o, size_t* permU_off) {
        o += 13 * sizeof(int);
        for (int blk = 0; blk < 9; blk++) {
          long long cnt;
          std::memcpy(&cnt, b.data() + o, sizeof(cnt));
          if (blk == 4 && permU_off) *permU_off = cnt > 0 ? o + sizeof(long long) : 0;
          o += sizeof(long long) + (size_t)cnt * ((blk == 4 || blk == 5 || blk == 7 || blk == 8) ? sizeof(int) : sizeof(double));
        }
        return o;
      };
      off = skip_node(off, nullptr);
      size_t pu = 0;
      skip_node(off, &pu);
      if (pu) {
        const int badv = 1 << 28;
        std::memcpy(b.data() + pu, &badv, sizeof(int));
        ok = refused(b, "permutation entry") && ok;
      }
    }
    if (!ok) return 1;
  }
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
  // clone (HSSMatrix.hpp:186): an independent deep copy; reset (:313): back to uncompressed; draw (:492): the partition as gnuplot
  {
    auto C = H.clone();
    auto Y3 = C->apply(X);
    Y3.scaled_add(-1., Y1);
    if (Y3.normF() != 0. || C->rank() != H.rank() || C->memory() != H.memory()) { std::cout << "ERROR: clone differs" << std::endl; return 1; }
    C->reset();
    if (C->is_compressed() || !H.is_compressed()) { std::cout << "ERROR: reset" << std::endl; return 1; }
    std::ostringstream os;
    H.draw(os);
    const std::string d = os.str();
    std::size_t rects = 0;
    for (std::size_t p = d.find("set obj rect"); p != std::string::npos; p = d.find("set obj rect", p + 1)) rects++;
    // one rectangle per leaf and two per inner node
    std::size_t leaves = 0, inner = 0;
    std::function<void(const HSS::HSSMatrix<double>*)> count = [&](const HSS::HSSMatrix<double>* M) {
      if (M->leaf()) { leaves++; return; }
      inner++;
      count(M->child(0));
      count(M->child(1));
    };
    count(&H);
    if (rects != leaves + 2 * inner) { std::cout << "ERROR: draw wrote " << rects << " rectangles for " << leaves << " leaves" << std::endl; return 1; }
    H.set_openmp_task_depth(2);
    // factor_nonzeros (HSSMatrixBase.hpp:177): L, Vt0, W1, Q of the eliminated nodes + D of the root, from the node table
    {
      std::size_t want = 0;
      std::function<void(const HSS::HSSMatrix<double>*, bool)> cnt = [&](const HSS::HSSMatrix<double>* M, bool root) {
        if (root) { const std::size_t mu = M->leaf() ? M->rows() : M->child(0)->U_rank() + M->child(1)->U_rank(); want += mu * mu; }
        else if (M->U_rows() > M->U_rank()) {
          const std::size_t mm = M->U_rows(), r = M->U_rank(), q = mm - r;
          want += q * q + q * M->V_rank() + r * mm + mm * mm;
        }
        if (!M->leaf()) { cnt(M->child(0), false); cnt(M->child(1), false); }
      };
      auto Fc = H.clone();
      if (Fc->factor_nonzeros() != 0) { std::cout << "ERROR: factor_nonzeros before factor()" << std::endl; return 1; }
      Fc->factor();
      cnt(Fc.get(), true);
      if (Fc->factor_nonzeros() != want || want == 0) { std::cout << "ERROR: factor_nonzeros " << Fc->factor_nonzeros() << " vs " << want << std::endl; return 1; }
    }
    // StructuredMatrix interface defaults (structured/StructuredMatrix.hpp:262-320, StructuredMatrix.cpp:572-605)
    const structured::StructuredMatrix<double>& SM = H;
    if (SM.dist().size() != 2 || SM.dist()[1] != int(H.rows()) || SM.rdist() != SM.cdist()) { std::cout << "ERROR: dist()" << std::endl; return 1; }
    bool t1 = false, t2 = false;
    try { (void)SM.local_rows(); } catch (const std::invalid_argument&) { t1 = true; }
    try {
      structured::StructuredOptions<double> so;
      so.set_type(structured::Type::HSS);
      structured::mult_t<double> mf = [](Trans, const DenseMatrix<double>&, DenseMatrix<double>&) {};
      structured::construct_matrix_free<double>(10, 10, mf, so);
    } catch (const std::invalid_argument&) { t2 = true; }
    if (!t1 || !t2) { std::cout << "ERROR: StructuredMatrix defaults" << std::endl; return 1; }
    // DenseMatrix utilities of the reference's class (dense/DenseMatrix.hpp): LU / solve, norms, permutations
    {
      const int q = 7;
      DenseMatrix<double> M(q, q), xq(q, 2), bq(q, 2);
      M.random(); xq.random();
      M.shift(3.);
      gemm(Trans::N, Trans::N, 1., M, xq, 0., bq);
      DenseMatrix<double> F(M);
      auto pv = F.LU();
      auto sol = F.solve(bq, pv);
      sol.scaled_add(-1., xq);
      if (sol.normF() > 1e-12 * xq.normF()) { std::cout << "ERROR: DenseMatrix LU / solve" << std::endl; return 1; }
      auto Mt = M.conj_transpose();
      if (std::abs(M.norm1() - Mt.normI()) > 1e-14 * M.norm1() || M.zeros() != 0 || M.subnormals() != 0) { std::cout << "ERROR: DenseMatrix norms" << std::endl; return 1; }
      std::vector<int> P = {3, 1, 2, 7, 6, 5, 4};
      DenseMatrix<double> R1(M);
      R1.lapmr(P, true);
      for (int i = 0; i < q; i++) if (R1(i, 2) != M(P[i] - 1, 2)) { std::cout << "ERROR: lapmr" << std::endl; return 1; }
      R1.lapmr(P, false);
      R1.lapmt(P, true);
      R1.lapmt(P, false);
      R1.scaled_add(-1., M);
      if (R1.normF() != 0.) { std::cout << "ERROR: lapmr / lapmt round trip" << std::endl; return 1; }
      DenseMatrix<double> S1(M);
      S1.scale_and_add(2., M);
      S1.scaled_add(-3., M);
      if (S1.normF() > 1e-13 * M.normF()) { std::cout << "ERROR: scale_and_add" << std::endl; return 1; }
    }
    // delete_trailing_block (HSSMatrix.hpp:476): what is left no longer applies as a whole
    auto T = H.clone();
    T->delete_trailing_block();
    bool threw = false;
    try { T->apply(X); } catch (const std::logic_error&) { threw = true; }
    if (!threw) { std::cout << "ERROR: apply after delete_trailing_block" << std::endl; return 1; }
  }
  std::cout << "# exiting" << std::endl;
  return 0;
}
