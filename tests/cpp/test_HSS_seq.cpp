// C++ driver over the reference-shaped classes (strumpack::HSS::HSSMatrix<double>, HSSOptions,
// DenseMatrix): same command line, same checks and same pass/fail thresholds as the reference's
// test/test_HSS_seq.cpp (:38-39 tolerances, :143-152 compression error, :193-233 element / sub-block
// extraction, :235-250 ULV solve), written against this repository's headers.  Links either the product
// library (GPU tests) or the emulator build (CPU tests).
//   test_HSS_seq T|U|L n [--hss_leaf_size ...] [--hss_rel_tol ...] ...
#include <cmath>
#include <iostream>
#include <random>

#include "HSSMatrix.hpp"

using namespace strumpack;
using namespace strumpack::HSS;

#define ERROR_TOLERANCE 1e2
#define SOLVE_TOLERANCE 1e-12

int main(int argc, char* argv[]) {
  if (argc < 3) { std::cout << "usage: test_HSS_seq T|U|L n [HSS options]" << std::endl; return 2; }
  const char prob = argv[1][0];
  const int m = std::stoi(argv[2]);
  HSSOptions<double> hss_opts;
  hss_opts.set_verbose(false);
  hss_opts.set_from_command_line(argc, argv);

  DenseMatrix<double> A(m, m);
  if (prob == 'T' || prob == 'U') {
    for (int j = 0; j < m; j++)
      for (int i = 0; i < m; i++) {
        A(i, j) = (i == j) ? 1. : 1. / (1 + std::abs(i - j));
        if (prob == 'U' && i > j) A(i, j) = 0.;
      }
  } else {  // 'L': identity + (1/m) U V^T, U == V from the default generator
    const int k = std::max(1, int(0.3 * m));
    DenseMatrix<double> U(m, k);
    U.random();
    A.eye();
    for (int j = 0; j < m; j++)
      for (int i = 0; i < m; i++) {
        double s = 0;
        for (int l = 0; l < k; l++) s += U(i, l) * U(j, l);
        A(i, j) += s / m;
      }
  }
  std::cout << "# tol = " << hss_opts.rel_tol() << std::endl;
  HSSMatrix<double> H(A, hss_opts);
  if (!H.is_compressed()) { std::cout << "# compression failed!!!!!!!!" << std::endl; return 1; }
  std::cout << "# created H matrix of dimension " << H.rows() << " x " << H.cols() << " with " << H.levels()
            << " levels" << std::endl << "# compression succeeded!" << std::endl;
  std::cout << "# rank(H) = " << H.rank() << std::endl;
  std::cout << "# memory(H) = " << H.memory() / 1e6 << " MB, " << 100. * H.memory() / A.memory() << "% of dense" << std::endl;

  auto Hdense = H.dense();
  Hdense.scaled_add(-1., A);
  const double tol = ERROR_TOLERANCE * std::max(hss_opts.rel_tol(), hss_opts.abs_tol());
  std::cout << "# relative error = ||A-H*I||_F/||A||_F = " << Hdense.normF() / A.normF() << std::endl;
  if (Hdense.normF() / A.normF() > tol) { std::cout << "ERROR: compression error too big!!" << std::endl; return 1; }

  // transposed product against the dense matrix
  {
    DenseMatrix<double> B(m, 3), C(m, 3), Cc(m, 3);
    B.random();
    apply_HSS(Trans::C, H, B, 0., C);
    for (int j = 0; j < 3; j++)
      for (int i = 0; i < m; i++) { double s = 0; for (int l = 0; l < m; l++) s += A(l, i) * B(l, j); Cc(i, j) = s; }
    C.scaled_add(-1., Cc);
    std::cout << "# relative error = ||H'*B-A'*B||_F/||A'*B||_F = " << C.normF() / Cc.normF() << std::endl;
    if (C.normF() / Cc.normF() > tol) { std::cout << "ERROR: transposed product error too big!!" << std::endl; return 1; }
  }

  // C = H B + beta C with many right-hand sides (the leaves' D B + beta C runs next to the tree sweep, on a side stream):
  // against the same product column by column
  {
    const int nb = 20;
    DenseMatrix<double> B(m, nb), C(m, nb), C1(m, nb);
    B.random();
    C.random();
    C1.copy(C);
    apply_HSS(Trans::N, H, B, -0.5, C);
    for (int j = 0; j < nb; j++) {
      DenseMatrix<double> bj(m, 1), cj(m, 1);
      for (int i = 0; i < m; i++) { bj(i, 0) = B(i, j); cj(i, 0) = C1(i, j); }
      apply_HSS(Trans::N, H, bj, -0.5, cj);
      for (int i = 0; i < m; i++) C1(i, j) = cj(i, 0);
    }
    C.scaled_add(-1., C1);
    if (C.normF() > 1e-12 * C1.normF()) { std::cout << "ERROR: H B + beta C with 20 right-hand sides: " << C.normF() / C1.normF() << std::endl; return 1; }
  }

  std::default_random_engine gen;
  std::uniform_int_distribution<std::size_t> random_idx(0, m - 1);
  double ex_err = 0;
  const int iex = 5;
  for (int i = 0; i < iex; i++) { auto r = random_idx(gen), c = random_idx(gen); ex_err += std::abs(H.get(r, c) - A(r, c)); }
  std::cout << "# extracting individual elements, avg error = " << ex_err / iex << std::endl;
  if (ex_err / iex > tol) { std::cout << "ERROR: extraction error too big!!" << std::endl; return 1; }
  std::vector<std::size_t> I, J;
  for (int i = 0; i < 8; i++) I.push_back(random_idx(gen));
  for (int j = 0; j < 8; j++) J.push_back(random_idx(gen));
  auto sub = H.extract(I, J);
  DenseMatrix<double> sub_dense(8, 8);
  for (int j = 0; j < 8; j++) for (int i = 0; i < 8; i++) sub_dense(i, j) = A(I[i], J[j]);
  sub.scaled_add(-1., sub_dense);
  std::cout << "# sub-matrix extraction error = " << sub.normF() / sub_dense.normF() << std::endl;
  if (sub.normF() / sub_dense.normF() > tol) { std::cout << "ERROR: extraction error too big!!" << std::endl; return 1; }

  std::cout << "# computing ULV factorization of HSS matrix .." << std::endl;
  H.factor();
  std::cout << "# solving linear system .." << std::endl;
  DenseMatrix<double> B(m, 1);
  B.random();
  DenseMatrix<double> C(B);
  H.solve(C);
  auto Bcheck = H.apply(C);
  Bcheck.scaled_add(-1., B);
  std::cout << "# relative error = ||B-H*(H\\B)||_F/||B||_F = " << Bcheck.normF() / B.normF() << std::endl;
  if (Bcheck.normF() / B.normF() > SOLVE_TOLERANCE) { std::cout << "ERROR: ULV solve relative error too big!!" << std::endl; return 1; }

  // shift + re-factor (structured API semantics)
  H.shift(1.5);
  H.factor();
  DenseMatrix<double> X(B);
  H.solve(X);
  auto R = H.apply(X);
  R.scaled_add(-1., B);
  if (R.normF() / B.normF() > SOLVE_TOLERANCE) { std::cout << "ERROR: solve after shift failed" << std::endl; return 1; }
  {   // the same solve in its two halves (HSSMatrix.hpp:360-376)
    WorkSolve<double> w;
    DenseMatrix<double> X2(B.rows(), B.cols());
    H.forward_solve(w, B, false);
    H.backward_solve(w, X2);
    X2.scaled_add(-1., X);
    if (X2.normF() > 1e-13 * X.normF()) { std::cout << "ERROR: forward_solve + backward_solve differ from solve" << std::endl; return 1; }
  }

  if (!H.leaf()) {
    // test_HSS_seq.cpp:252-260, plus the check the reference leaves as a TODO: with H z = [0; y] the Schur
    // complement S of the (0,0) block satisfies S z1 = y (and S^T likewise through the transposed system)
    DenseMatrix<double> z(m, 1);
    z.random();
    H.partial_factor();
    std::cout << "# Computing Schur update .." << std::endl;
    DenseMatrix<double> Theta, Phi, DUB01;
    H.Schur_update(Theta, DUB01, Phi);
    const std::size_t n1 = Theta.rows(), n0 = m - n1;
    auto Vhat = H.Vhat();
    if (Vhat.rows() != DUB01.rows() || Vhat.cols() != Theta.cols() || Phi.cols() != DUB01.rows()) {
      std::cout << "ERROR: Schur update shapes are inconsistent" << std::endl;
      return 1;
    }
    DenseMatrix<double> z0(n0, 1), z1(n1, 1), Sr, Sc, TV;
    for (std::size_t i = 0; i < n0; i++) z0(i, 0) = z(i, 0);
    for (std::size_t i = 0; i < n1; i++) z1(i, 0) = z(n0 + i, 0);
    H.Schur_product_direct(Theta, DUB01, Phi, TV, z1, Sr, Sc);
    // samples of H and H^T at z -> samples of S at z1 (indirect), must agree with the direct product
    auto Hz = H.apply(z), Htz = H.applyC(z);
    DenseMatrix<double> Sr1(n1, 1), Sc1(n1, 1), Sr2, Sc2;
    for (std::size_t i = 0; i < n1; i++) { Sr1(i, 0) = Hz(n0 + i, 0); Sc1(i, 0) = Htz(n0 + i, 0); }
    H.Schur_product_indirect(DUB01, z0, z1, Sr1, Sc1, Sr2, Sc2);
    Sr2.scaled_add(-1., Sr);
    Sc2.scaled_add(-1., Sc);
    std::cout << "# Schur products, direct vs indirect = " << Sr2.normF() / Sr.normF() << " , " << Sc2.normF() / Sc.normF() << std::endl;
    if (Sr2.normF() > 1e-10 * Sr.normF() || Sc2.normF() > 1e-10 * Sc.normF()) { std::cout << "ERROR: Schur products disagree" << std::endl; return 1; }
    // the solve of a sparse front on top of the partial factorization (sparse/fronts/FrontHSS.cpp:446-501): forward half on
    // child(0) with partial = true, the update part through Theta and the dense Schur complement, Phi^* y_upd off the reduced
    // solution, backward half -- [x0; x1] then solves the compressed system H x = b
    {
      auto c0 = H.child(0);
      auto c1 = H.child(1);
      const int nb = 3;
      DenseMatrix<double> bf(m, nb), b0(n0, nb), b1(n1, nb);
      bf.random();
      for (int j = 0; j < nb; j++) {
        for (std::size_t i = 0; i < n0; i++) b0(i, j) = bf(i, j);
        for (std::size_t i = 0; i < n1; i++) b1(i, j) = bf(n0 + i, j);
      }
      WorkSolve<double> w;
      c0->forward_solve(w, b0, true);
      if (w.reduced_rhs.rows() != Theta.cols() || w.x.rows() != Phi.cols()) { std::cout << "ERROR: forward_solve shapes" << std::endl; return 1; }
      gemm(Trans::N, Trans::N, -1., Theta, w.reduced_rhs, 1., b1);                 // b_upd -= Theta reduced_rhs
      auto S = c1->dense();                                                        // S = H11 - Theta Vhat^* Phi^*
      DenseMatrix<double> VP(Vhat.cols(), Phi.rows());
      gemm(Trans::C, Trans::C, 1., Vhat, Phi, 0., VP);
      gemm(Trans::N, Trans::N, -1., Theta, VP, 1., S);
      auto pv = S.LU();
      auto x1 = S.solve(b1, pv);
      gemm(Trans::C, Trans::N, -1., Phi, x1, 1., w.x);                             // x_root -= Phi^* y_upd
      DenseMatrix<double> x0(n0, nb);
      c0->backward_solve(w, x0);
      DenseMatrix<double> xf(m, nb);
      for (int j = 0; j < nb; j++) {
        for (std::size_t i = 0; i < n0; i++) xf(i, j) = x0(i, j);
        for (std::size_t i = 0; i < n1; i++) xf(n0 + i, j) = x1(i, j);
      }
      auto rf = H.apply(xf);
      rf.scaled_add(-1., bf);
      std::cout << "# front solve on the partial factorization: ||H x - b||_F/||b||_F = " << rf.normF() / bf.normF() << std::endl;
      if (rf.normF() > 1e-9 * bf.normF()) { std::cout << "ERROR: forward_solve / backward_solve with a partial factorization" << std::endl; return 1; }
    }
    // child views (HSSMatrix.hpp:194-202): blocks of H through child(c), children of children, Vhat through the view
    {
      auto c0 = H.child(0);
      auto c1 = H.child(1);
      if (c0->rows() != n0 || c1->rows() != n1 || c0->levels() + 1 > H.levels() || c1->rank() > H.rank()) {
        std::cout << "ERROR: child views have the wrong shape" << std::endl;
        return 1;
      }
      if (c0->V_rank() != Theta.cols() || c0->ULV().Vhat().rows() != Vhat.rows()) { std::cout << "ERROR: child(0) ranks" << std::endl; return 1; }
      auto D11 = c1->dense();
      auto Hd = H.dense();
      double e11 = 0, n11 = 0;
      for (std::size_t j = 0; j < n1; j++)
        for (std::size_t i = 0; i < n1; i++) { const double df = D11(i, j) - Hd(n0 + i, n0 + j); e11 += df * df; n11 += Hd(n0 + i, n0 + j) * Hd(n0 + i, n0 + j); }
      std::cout << "# ||child(1)->dense() - H(1,1)||_F/||H(1,1)||_F = " << std::sqrt(e11 / n11) << std::endl;
      if (std::sqrt(e11 / n11) > 1e-13) { std::cout << "ERROR: child(1) is not the (1,1) block" << std::endl; return 1; }
      if (!c1->leaf()) {
        auto c10 = c1->child(0);
        DenseMatrix<double> x(c10->rows(), 2);
        x.random();
        auto y = c10->apply(x);
        double e = 0, nn = 0;
        for (std::size_t j = 0; j < 2; j++)
          for (std::size_t i = 0; i < c10->rows(); i++) {
            double s = 0;
            for (std::size_t k = 0; k < c10->rows(); k++) s += D11(i, k) * x(k, j);
            e += (y(i, j) - s) * (y(i, j) - s); nn += s * s;
          }
        if (std::sqrt(e / nn) > 1e-12) { std::cout << "ERROR: child(1)->child(0)->apply" << std::endl; return 1; }
        if (std::abs(c10->get(1 % c10->rows(), 0) - D11(1 % c10->rows(), 0)) > 1e-13) { std::cout << "ERROR: child get" << std::endl; return 1; }
      }
    }
    // a child factored and solved as a matrix of its own: child(1)->solve(b) is D11^{-1} b, with D11 the compressed (1,1) block
    {
      auto c1 = H.child(1);
      auto D11 = c1->dense();
      DenseMatrix<double> xs(n1, 20), bs(n1, 20);   // (20 columns: the many-right-hand-side forms of the sweeps, on a subtree)
      xs.random();
      gemm(Trans::N, Trans::N, 1., D11, xs, 0., bs);
      {
        auto ys = c1->apply(xs);
        ys.scaled_add(-1., bs);
        if (ys.normF() > 1e-12 * bs.normF()) { std::cout << "ERROR: child(1)->apply with 20 columns" << std::endl; return 1; }
      }
      c1->factor();
      c1->solve(bs);
      bs.scaled_add(-1., xs);
      std::cout << "# ||child(1)^{-1} (D11 x) - x||_F/||x||_F = " << bs.normF() / xs.normF() << std::endl;
      if (bs.normF() > 1e-9 * xs.normF()) { std::cout << "ERROR: child(1)->factor() / solve()" << std::endl; return 1; }
      if (!c1->leaf()) {   // a grandchild, and the whole matrix refuses to solve on the child's factors
        auto c10 = c1->child(0);
        auto D = c10->dense();
        DenseMatrix<double> x2(c10->rows(), 1), b2(c10->rows(), 1);
        x2.random();
        gemm(Trans::N, Trans::N, 1., D, x2, 0., b2);
        c10->factor();
        c10->solve(b2);
        b2.scaled_add(-1., x2);
        if (b2.normF() > 1e-9 * x2.normF()) { std::cout << "ERROR: child(1)->child(0)->factor() / solve()" << std::endl; return 1; }
        bool threw = false;
        try { DenseMatrix<double> t(n1, 1); c1->solve(t); } catch (const std::logic_error&) { threw = true; }
        if (!threw) { std::cout << "ERROR: child(1)->solve() on the factors of its child" << std::endl; return 1; }
      }
    }
    // a shift kills the factors of a child as it kills the whole matrix's: solving on them afterwards is refused
    {
      auto c1 = H.child(1);
      c1->factor();
      H.shift(0.25);
      bool threw = false;
      try { DenseMatrix<double> t(n1, 1); c1->solve(t); } catch (const std::logic_error&) { threw = true; }
      H.shift(-0.25);
      if (!threw) { std::cout << "ERROR: child(1)->solve() on factors from before a shift" << std::endl; return 1; }
    }
    // S^{-1} y is the lower part of H^{-1} [0; y]
    H.factor();
    DenseMatrix<double> rhs(m, 1);
    for (std::size_t i = 0; i < n1; i++) rhs(n0 + i, 0) = Sr(i, 0);
    H.solve(rhs);
    double num = 0, den = 0;
    for (std::size_t i = 0; i < n1; i++) { num += (rhs(n0 + i, 0) - z1(i, 0)) * (rhs(n0 + i, 0) - z1(i, 0)); den += z1(i, 0) * z1(i, 0); }
    std::cout << "# ||S^{-1} (S z1) - z1|| / ||z1|| = " << std::sqrt(num / den) << std::endl;
    if (std::sqrt(num / den) > 1e-9) { std::cout << "ERROR: Schur complement is not consistent with the ULV solve" << std::endl; return 1; }
  }
  // partially matrix-free construction: compress(Amult, Aelem, opts) (HSS/HSSMatrix.cpp:173-186) -- the user multiplies the
  // random blocks and evaluates elements; same sketching matrix, so the result is the matrix compressed from A itself
  if (m > 1 && m <= 600) {
    HSSMatrix<double>::mult_t Amult = [&](DenseMatrix<double>& Rr, DenseMatrix<double>& Rc, DenseMatrix<double>& Sr, DenseMatrix<double>& Sc) {
      const std::size_t d = Rr.cols();
      for (std::size_t c = 0; c < d; c++)
        for (int i = 0; i < m; i++) {
          double sr = 0, sc = 0;
          for (int l = 0; l < m; l++) { sr += A(i, l) * Rr(l, c); sc += A(l, i) * Rc(l, c); }
          Sr(i, c) = sr; Sc(i, c) = sc;
        }
    };
    HSSMatrix<double>::elem_t Aelem = [&](const std::vector<std::size_t>& I, const std::vector<std::size_t>& J, DenseMatrix<double>& B) {
      for (std::size_t j = 0; j < J.size(); j++)
        for (std::size_t i = 0; i < I.size(); i++) B(i, j) = A(I[i], J[j]);
    };
    HSSMatrix<double> Hf(m, m, hss_opts);
    Hf.compress(Amult, Aelem, hss_opts);
    if (!Hf.is_compressed()) { std::cout << "# matrix-free compression failed!!!!!!!!" << std::endl; return 1; }
    auto Hfd = Hf.dense();
    Hfd.scaled_add(-1., A);
    std::cout << "# matrix-free: rank " << Hf.rank() << ", relative error = " << Hfd.normF() / A.normF() << std::endl;
    if (Hfd.normF() / A.normF() > tol) { std::cout << "ERROR: matrix-free compression error too big!!" << std::endl; return 1; }
    if (std::abs(double(Hf.rank()) - double(H.rank())) > 1) { std::cout << "ERROR: matrix-free rank differs!!" << std::endl; return 1; }
    // user_defined_random (HSSOptions.hpp:289, compress_stable.hpp:110-141; the sparse HSS fronts' indirect sampling,
    // sparse/fronts/FrontHSS.cpp:383-385): the multiplication routine ALSO fills the random block.  Filled with the
    // reference's default stream, the compression retraces the one above (same ranks); SJLT would draw its own pattern
    if (hss_opts.compression_sketch() == CompressionSketch::GAUSSIAN) {
      auto uo = hss_opts;
      uo.set_user_defined_random(true);
      std::minstd_rand eng(0);
      std::normal_distribution<double> nd;
      int calls = 0;
      HSSMatrix<double>::mult_t Umult = [&](DenseMatrix<double>& Rr, DenseMatrix<double>& Rc, DenseMatrix<double>& Sr, DenseMatrix<double>& Sc) {
        calls++;
        for (std::size_t c = 0; c < Rr.cols(); c++)
          for (int i = 0; i < m; i++) Rr(i, c) = nd(eng);
        Rc.copy(Rr);
        Amult(Rr, Rc, Sr, Sc);
      };
      HSSMatrix<double> Hu(m, m, uo);
      Hu.compress(Umult, Aelem, uo);
      if (!Hu.is_compressed() || calls == 0) { std::cout << "# user-random compression failed!!!!!!!!" << std::endl; return 1; }
      auto Hud = Hu.dense();
      Hud.scaled_add(-1., A);
      std::cout << "# user-defined random: rank " << Hu.rank() << ", relative error = " << Hud.normF() / A.normF() << std::endl;
      if (Hud.normF() / A.normF() > tol) { std::cout << "ERROR: user-random compression error too big!!" << std::endl; return 1; }
      if (hss_opts.random_engine() == random::RandomEngine::LINEAR && Hu.rank() != Hf.rank()) { std::cout << "ERROR: user-random rank differs!!" << std::endl; return 1; }
      // distinct Rr / Rc are refused, not silently merged
      HSSMatrix<double>::mult_t Bad = [&](DenseMatrix<double>& Rr, DenseMatrix<double>& Rc, DenseMatrix<double>& Sr, DenseMatrix<double>& Sc) {
        Rr.fill(1.); Rc.fill(2.); Sr.zero(); Sc.zero();
      };
      bool refused = false;
      try { HSSMatrix<double> Hb(m, m, uo); Hb.compress(Bad, Aelem, uo); } catch (const std::invalid_argument&) { refused = true; }
      if (!refused) { std::cout << "ERROR: distinct Rr / Rc accepted" << std::endl; return 1; }
    }
  }

  std::cout << "# exiting" << std::endl;
  return 0;
}
