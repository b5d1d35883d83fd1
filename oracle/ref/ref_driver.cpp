// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or shipped with the product.
//
// A C-ABI shim over the *reference's own* CPU HSS classes (strumpack::HSS::HSSMatrix<double>,
// /root/reference/src/HSS/HSSMatrix.hpp:79) so that Python (ctypes) can (a) pin oracle/hss_oracle.py
// against the real reference, (b) generate the committed fixtures in tests/golden/, and (c) time the
// reference CPU path as bench.py's cpu_baseline (kind "reference").
// Compiled by oracle/ref/Makefile from the sources where they lie under /root/reference; the
// resulting library lives in oracle/_ref/ (git-ignored, travels to the GPU box as a built file).
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "HSS/HSSMatrix.hpp"
#include "StrumpackParameters.hpp"
#include "dense/DenseMatrix.hpp"
#include "misc/RandomWrapper.hpp"
#include "kernel/KernelRegression.hpp"
#include "clustering/Clustering.hpp"
#include "clustering/NeighborSearch.hpp"
#include "BLR/BLRMatrix.hpp"

using namespace strumpack;
using namespace strumpack::HSS;

namespace {
struct RefHSS {
  std::unique_ptr<HSSMatrix<double>> H;
  int n = 0;
  DenseMatrix<double> Theta, DUB01, Phi, TV;   // Schur_update results (+ what FrontHSS.cpp:396-407 derives)
};

HSSOptions<double> make_opts(double rel_tol, double abs_tol, int leaf, int d0, int dd, int p,
                             int max_rank, int algo) {
  HSSOptions<double> o;
  o.set_verbose(false);
  o.set_rel_tol(rel_tol);
  o.set_abs_tol(abs_tol);
  o.set_leaf_size(leaf);
  o.set_d0(d0);
  o.set_dd(dd);
  o.set_p(p);
  o.set_max_rank(max_rank);
  o.set_compression_algorithm(algo == 0 ? CompressionAlgorithm::ORIGINAL
                                         : CompressionAlgorithm::STABLE);
  return o;
}

double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

extern "C" {

// N(0,1) draws of the reference's default generator (minstd_rand(0) + normal_distribution).
void ref_randn(long long count, double* out) {
  auto g = random::make_default_random_generator<double>();
  for (long long i = 0; i < count; i++) out[i] = g->get();
}

// Test matrices of test/test_HSS_seq.cpp:69-105, column-major n x n into A:
//  'T' Toeplitz (:75-78), 'U' upper-triangular Toeplitz (:86-90),
//  'L' identity + (1/n) U V^T with U, V from DenseMatrix::random() (:92-105).
void ref_fill_test_matrix(char kind, int n, double* A) {
  if (kind == 'L') {
    DenseMatrix<double> Ad(n, n);
    Ad.eye();
    int k = std::max(1, int(0.3 * n));
    DenseMatrix<double> U(n, k), V(n, k);
    U.random();
    V.random();
    gemm(Trans::N, Trans::C, 1. / n, U, V, 1., Ad);
    for (int j = 0; j < n; j++)
      std::memcpy(A + (size_t)j * n, Ad.ptr(0, j), sizeof(double) * n);
    return;
  }
#pragma omp parallel for schedule(static)
  for (int j = 0; j < n; j++)
    for (int i = 0; i < n; i++) {
      double v = (i == j) ? 1. : 1. / (1 + std::abs(i - j));
      if (kind == 'U' && i > j) v = 0.;
      A[i + (size_t)j * n] = v;
    }
}

void* ref_hss_create(int n, const double* A, int lda, double rel_tol, double abs_tol, int leaf,
                     int d0, int dd, int p, int max_rank, int algo) {
  auto o = make_opts(rel_tol, abs_tol, leaf, d0, dd, p, max_rank, algo);
  DenseMatrix<double> Ad(n, n, A, lda);
  auto* r = new RefHSS;
  r->n = n;
  r->H.reset(new HSSMatrix<double>(Ad, o));
  return r;
}

// The Toeplitz matrix T(n) of test_HSS_seq.cpp:75-78 WITHOUT storing it: HSSMatrix(n, n, opts) + compress(Amult, Aelem, opts)
// (HSS/HSSMatrix.cpp:173-186; the route structured::construct_partially_matrix_free takes, structured/StructuredMatrix.cpp:
// 637-651) -- BASELINE configs[2] (N = 100000) needs 80 GB as a dense operand, this needs N x d buffers.  The products
// Sr = T Rr, Sc = T^T Rc are formed tile by tile (tiles generated on the fly, dgemm per tile), O(N^2 d) like the dense route;
// the random blocks come from the same default generator, so the compression is the one HSSMatrix(A, opts) performs.
void* ref_hss_create_toeplitz_matfree(int n, double rel_tol, double abs_tol, int leaf, int d0, int dd, int p, int max_rank, int algo) {
  auto o = make_opts(rel_tol, abs_tol, leaf, d0, dd, p, max_rank, algo);
  auto* r = new RefHSS;
  r->n = n;
  auto Amult = [n](DenseMatrix<double>& Rr, DenseMatrix<double>& Rc, DenseMatrix<double>& Sr, DenseMatrix<double>& Sc) {
    const int B = 1024, nt = (n + B - 1) / B, d = (int)Rr.cols();
    bool same = true;
    for (int j = 0; j < d && same; j++)
      for (int i = 0; i < n; i++) if (Rr(i, j) != Rc(i, j)) { same = false; break; }
    Sr.zero();
    if (!same) Sc.zero();
#pragma omp parallel
    {
      DenseMatrix<double> T(B, B);
#pragma omp for schedule(dynamic)
      for (int ti = 0; ti < nt; ti++) {
        const int i0 = ti * B, mi = std::min(B, n - i0);
        for (int tj = 0; tj < nt; tj++) {
          const int j0 = tj * B, nj = std::min(B, n - j0);
          for (int j = 0; j < nj; j++)
            for (int i = 0; i < mi; i++) {
              const int df = std::abs((i0 + i) - (j0 + j));
              T(i, j) = df ? 1. / (1 + df) : 1.;
            }
          DenseMatrixWrapper<double> Tw(mi, nj, T, 0, 0), Rj(nj, d, Rr, j0, 0), Si(mi, d, Sr, i0, 0);
          gemm(Trans::N, Trans::N, 1., Tw, Rj, 1., Si, params::task_recursion_cutoff_level);
          if (!same) {   // T is symmetric: T^T Rc = T Rc
            DenseMatrixWrapper<double> Rcj(nj, d, Rc, j0, 0), Sci(mi, d, Sc, i0, 0);
            gemm(Trans::N, Trans::N, 1., Tw, Rcj, 1., Sci, params::task_recursion_cutoff_level);
          }
        }
      }
    }
    if (same) Sc.copy(Sr);
  };
  auto Aelem = [](const std::vector<std::size_t>& I, const std::vector<std::size_t>& J, DenseMatrix<double>& Bm) {
    for (std::size_t j = 0; j < J.size(); j++)
      for (std::size_t i = 0; i < I.size(); i++) {
        const long long df = std::llabs((long long)I[i] - (long long)J[j]);
        Bm(i, j) = df ? 1. / (1 + df) : 1.;
      }
  };
  r->H.reset(new HSSMatrix<double>(n, n, o));
  r->H->compress(Amult, Aelem, o);
  return r;
}

// the SJLT sketch (--hss_compression_sketch SJLT, test/CMakeLists.txt:145-159): the reference seeds its pattern
// generator from the clock (HSSMatrix.sketch.hpp:266-270), so repeated calls give different matrices
void* ref_hss_create_sjlt(int n, const double* A, int lda, double rel_tol, double abs_tol, int leaf,
                          int d0, int dd, int algo, int perm, int nnz0, int nnz) {
  auto o = make_opts(rel_tol, abs_tol, leaf, d0, dd, 10, 50000, algo);
  o.set_compression_sketch(CompressionSketch::SJLT);
  o.set_SJLT_algo(perm ? SJLTAlgo::PERM : SJLTAlgo::CHUNK);
  o.set_nnz0(nnz0);
  o.set_nnz(nnz);
  DenseMatrix<double> Ad(n, n, A, lda);
  auto* r = new RefHSS;
  r->n = n;
  r->H.reset(new HSSMatrix<double>(Ad, o));
  return r;
}

void ref_hss_destroy(void* h) { delete static_cast<RefHSS*>(h); }
int ref_hss_is_compressed(void* h) { return static_cast<RefHSS*>(h)->H->is_compressed(); }
int ref_hss_levels(void* h) { return static_cast<RefHSS*>(h)->H->levels(); }
int ref_hss_rank(void* h) { return static_cast<RefHSS*>(h)->H->rank(); }
long long ref_hss_memory(void* h) { return static_cast<RefHSS*>(h)->H->memory(); }
long long ref_hss_nonzeros(void* h) { return static_cast<RefHSS*>(h)->H->nonzeros(); }

// Pre-order node table, 6 ints per node: row_offset, rows, U_rows, U_rank, V_rank, is_leaf.
// (parsed from the reference's own print_info, HSS/HSSMatrix.cpp:333-356). Returns node count.
int ref_hss_node_info(void* h, int* out, int cap_nodes) {
  std::ostringstream os;
  static_cast<RefHSS*>(h)->H->print_info(os, 0, 0);
  std::istringstream is(os.str());
  std::string line;
  int cnt = 0;
  while (std::getline(is, line)) {
    int rk, r0, r1, c0, c1, um, ur, vm, vr;
    char kind[32];
    if (sscanf(line.c_str(), "SEQ rank=%d b = [%d,%d x %d,%d]  U = %d x %d V = %d x %d %31s", &rk,
               &r0, &r1, &c0, &c1, &um, &ur, &vm, &vr, kind) == 10) {
      if (cnt < cap_nodes) {
        int* o = out + 6 * cnt;
        o[0] = r0; o[1] = r1 - r0; o[2] = um; o[3] = ur; o[4] = vr;
        o[5] = (std::strcmp(kind, "leaf") == 0);
      }
      cnt++;
    }
  }
  return cnt;
}

void ref_hss_dense(void* h, double* out, int ld) {
  auto D = static_cast<RefHSS*>(h)->H->dense();
  for (std::size_t j = 0; j < D.cols(); j++)
    std::memcpy(out + j * (size_t)ld, D.ptr(0, j), sizeof(double) * D.rows());
}

void ref_hss_mult(void* h, char trans, int nrhs, const double* B, int ldb, double* C, int ldc) {
  auto* r = static_cast<RefHSS*>(h);
  DenseMatrix<double> Bd(r->n, nrhs, B, ldb), Cd(r->n, nrhs);
  r->H->mult(trans == 'N' || trans == 'n' ? Trans::N : Trans::C, Bd, Cd);
  for (int j = 0; j < nrhs; j++)
    std::memcpy(C + j * (size_t)ldc, Cd.ptr(0, j), sizeof(double) * r->n);
}

void ref_hss_factor(void* h) { static_cast<RefHSS*>(h)->H->factor(); }

void ref_hss_solve(void* h, int nrhs, double* B, int ldb) {
  auto* r = static_cast<RefHSS*>(h);
  DenseMatrixWrapper<double> Bw(r->n, nrhs, B, ldb);
  r->H->solve(Bw);
}

void ref_hss_shift(void* h, double s) { static_cast<RefHSS*>(h)->H->shift(s); }

// partial_factor + Schur_update exactly as the sparse HSS front drives them (sparse/fronts/FrontHSS.cpp:391-407).
// dims: [0] n1 = rows of block 1, [1] Theta cols, [2] Phi cols, [3] DUB01 cols, [4] Vhat rows, [5] Vhat cols.
// Returns 0 for a leaf root (nothing done, test_HSS_seq.cpp:252).
int ref_hss_schur_update(void* h, int* dims) {
  auto* r = static_cast<RefHSS*>(h);
  for (int i = 0; i < 6; i++) dims[i] = 0;
  if (r->H->leaf()) return 0;
  r->H->partial_factor();
  r->H->Schur_update(r->Theta, r->DUB01, r->Phi);
  const DenseMatrix<double>& Vhat = r->H->child(0)->ULV().Vhat();
  if (r->Theta.cols() < r->Phi.cols()) {
    r->TV = DenseMatrix<double>(Vhat.cols(), r->Phi.rows());
    gemm(Trans::C, Trans::C, 1., Vhat, r->Phi, 0., r->TV);
  } else {
    r->TV = DenseMatrix<double>(r->Theta.rows(), Vhat.rows());
    gemm(Trans::N, Trans::C, 1., r->Theta, Vhat, 0., r->TV);
  }
  dims[0] = (int)r->H->child(1)->rows(); dims[1] = (int)r->Theta.cols(); dims[2] = (int)r->Phi.cols();
  dims[3] = (int)r->DUB01.cols(); dims[4] = (int)Vhat.rows(); dims[5] = (int)Vhat.cols();
  return 1;
}

// which: 0 Theta, 1 DUB01, 2 Phi, 3 Vhat; out is column-major with ld = the matrix's rows
void ref_hss_schur_get(void* h, int which, double* out) {
  auto* r = static_cast<RefHSS*>(h);
  const DenseMatrix<double>* M = which == 0 ? &r->Theta : which == 1 ? &r->DUB01 : which == 2 ? &r->Phi
                                                                       : &r->H->child(0)->ULV().Vhat();
  for (std::size_t j = 0; j < M->cols(); j++)
    for (std::size_t i = 0; i < M->rows(); i++) out[i + j * M->rows()] = (*M)(i, j);
}

// the low-rank update H10 H00^{-1} H01 of the Schur complement as a dense n1 x n1 matrix (FrontHSS.cpp:61-70)
void ref_hss_schur_update_dense(void* h, double* out) {
  auto* r = static_cast<RefHSS*>(h);
  const std::size_t n1 = r->H->child(1)->rows();
  DenseMatrix<double> U(n1, n1);
  U.zero();
  if (r->Theta.cols() < r->Phi.cols()) gemm(Trans::N, Trans::N, 1., r->Theta, r->TV, 0., U);
  else gemm(Trans::N, Trans::C, 1., r->TV, r->Phi, 0., U);
  for (std::size_t j = 0; j < n1; j++) std::memcpy(out + j * n1, U.ptr(0, j), sizeof(double) * n1);
}

// Sr = S R, Sc = S^T R with S = H11 - H10 H00^{-1} H01 (Schur_product_direct, as FrontHSS.cpp:218-219 calls it)
void ref_hss_schur_product_direct(void* h, int c, const double* R, double* Sr, double* Sc) {
  auto* r = static_cast<RefHSS*>(h);
  const std::size_t n1 = r->H->child(1)->rows();
  DenseMatrix<double> Rd(n1, c, R, n1), Srd(n1, c), Scd(n1, c);
  r->H->Schur_product_direct(r->Theta, r->DUB01, r->Phi, r->TV, Rd, Srd, Scd);
  for (int j = 0; j < c; j++) {
    std::memcpy(Sr + j * n1, Srd.ptr(0, j), sizeof(double) * n1);
    std::memcpy(Sc + j * n1, Scd.ptr(0, j), sizeof(double) * n1);
  }
}

// Reference flop counters (StrumpackParameters.hpp:78-97): out[0..8] =
// flops, update_sample, reduce_sample, ID, QR, ortho, random, ULV_factor, hss_solve.
void ref_flops(long long* out, int reset) {
  out[0] = params::flops; out[1] = params::update_sample_flops;
  out[2] = params::reduce_sample_flops; out[3] = params::ID_flops; out[4] = params::QR_flops;
  out[5] = params::ortho_flops; out[6] = params::random_flops;
  out[7] = params::ULV_factor_flops; out[8] = params::hss_solve_flops;
  if (reset) {
    params::flops = 0; params::update_sample_flops = 0; params::reduce_sample_flops = 0;
    params::ID_flops = 0; params::QR_flops = 0; params::ortho_flops = 0;
    params::random_flops = 0; params::ULV_factor_flops = 0; params::hss_solve_flops = 0;
  }
}

// CPU baseline: time compress / factor / solve / apply of the reference on Toeplitz T(n)
// (nrhs right-hand sides). times[0..3] seconds; stats: rank, levels, resid ||b-H(H\b)||/||b||.
// Returns 0 on success, 1 if compression failed.
int ref_hss_bench_toeplitz(int n, int leaf, double rel_tol, double abs_tol, int nrhs,
                           double* times, double* stats) {
  DenseMatrix<double> A(n, n);
  ref_fill_test_matrix('T', n, A.data());
  HSSOptions<double> o;
  o.set_verbose(false);
  o.set_leaf_size(leaf);
  o.set_rel_tol(rel_tol);
  o.set_abs_tol(abs_tol);
  double t0 = now();
  HSSMatrix<double> H(A, o);
  double t1 = now();
  if (!H.is_compressed()) return 1;
  H.factor();
  double t2 = now();
  DenseMatrix<double> B(n, nrhs);
  B.random();
  DenseMatrix<double> X(B);
  double t3 = now();
  H.solve(X);
  double t4 = now();
  auto C = H.apply(X);
  double t5 = now();
  C.scaled_add(-1., B);
  times[0] = t1 - t0; times[1] = t2 - t1; times[2] = t4 - t3; times[3] = t5 - t4;
  stats[0] = H.rank(); stats[1] = H.levels(); stats[2] = C.normF() / B.normF();
  stats[3] = H.memory();
  return 0;
}


// ---- BLR, dense slice: structured::construct_from_dense / construct_and_factor_from_dense with
// type BLR (structured/StructuredMatrix.cpp:77-97, 408-430).  Y = B X of the compressed matrix, Z = A^{-1} X through the
// BLR LU; stats: rank, memory (bytes), nonzeros of the compressed matrix.
int ref_blr_dense(int n, const double* A, int leaf, double rel_tol, double abs_tol, int nrhs, const double* X,
                  double* Y, double* Z, double* stats) {
  structured::StructuredOptions<double> o;
  o.set_type(structured::Type::BLR);
  o.set_leaf_size(leaf);
  o.set_rel_tol(rel_tol);
  o.set_abs_tol(abs_tol);
  o.set_verbose(false);
  DenseMatrix<double> Ad(n, n, A, n), Xd(n, nrhs, X, n), Yd(n, nrhs);
  auto B = structured::construct_from_dense(Ad, o);
  B->mult(Trans::N, Xd, Yd);
  std::memcpy(Y, Yd.data(), sizeof(double) * n * nrhs);
  stats[0] = B->rank(); stats[1] = B->memory(); stats[2] = B->nonzeros();
  auto F = structured::construct_and_factor_from_dense(Ad, o);
  DenseMatrix<double> Zd(Xd);
  F->solve(Zd);
  std::memcpy(Z, Zd.data(), sizeof(double) * n * nrhs);
  return 0;
}

// ---- BLR frontal matrix (BASELINE configs[4]): BLRMatrix<double>::construct_and_partial_factor
// (BLR/BLRMatrix.cpp:740-1037) on [F11 F12; F21 F22] exactly as sparse/fronts/FrontBLR.cpp:419-432 calls it (default
// options: RRQR tiles, algorithm RL), followed by the two solve phases of the front (FrontBLR.cpp:525-570).
//   F22: in = the assembled update block, out = the Schur complement.  adm: nt1 x nt1 column-major (null: weak
//   admissibility -- every off-diagonal tile).  ranks: (nt1 + nt2)^2 column-major over the tiles of the whole front,
//   rank of a low-rank tile, -1 dense, -2 the F22 part.  bsep / bupd: forward phase in place; ysep: backward phase in
//   place with yupd.  stats: [0] seconds, [1..3] nonzeros of B11 / B12 / B21, [4] largest rank of the three.
int ref_blr_front(int dsep, int dupd, const double* F11, const double* F12, const double* F21, double* F22, int nt1,
                  const int* tiles1, int nt2, const int* tiles2, const char* adm, double rel_tol, double abs_tol, int nrhs,
                  double* bsep, double* bupd, double* ysep, const double* yupd, int* ranks, double* stats) {
  using BLRM = BLR::BLRMatrix<double>;
  BLR::BLROptions<double> o;
  o.set_rel_tol(rel_tol);
  o.set_abs_tol(abs_tol);
  o.set_verbose(false);
  // REF_BLR_ALGO = LL / COMB / STAR: the reference's other factorization variants (default RL)
  if (const char* e = std::getenv("REF_BLR_ALGO")) {
    const std::string a(e);
    if (a == "LL") o.set_BLR_factor_algorithm(BLR::BLRFactorAlgorithm::LL);
    else if (a == "COMB") o.set_BLR_factor_algorithm(BLR::BLRFactorAlgorithm::COMB);
    else if (a == "STAR") o.set_BLR_factor_algorithm(BLR::BLRFactorAlgorithm::STAR);
  }
  // REF_BLR_LRA = ACA / BACA: the tile compression (default RRQR)
  if (const char* e = std::getenv("REF_BLR_LRA")) {
    const std::string a(e);
    if (a == "ACA") o.set_low_rank_algorithm(BLR::LowRankAlgorithm::ACA);
    else if (a == "BACA") o.set_low_rank_algorithm(BLR::LowRankAlgorithm::BACA);
  }
  std::vector<std::size_t> t1(tiles1, tiles1 + nt1), t2(tiles2, tiles2 + nt2);
  DenseMatrix<bool> A(nt1, nt1);
  for (int j = 0; j < nt1; j++)
    for (int i = 0; i < nt1; i++) A(i, j) = adm ? adm[i + (size_t)j * nt1] != 0 : (i != j);
  DenseMatrix<double> A11(dsep, dsep, F11, dsep), A12(dsep, dupd, F12, std::max(dsep, 1)), A21(dupd, dsep, F21, std::max(dupd, 1));
  DenseMatrixWrapper<double> A22(dupd, dupd, F22, std::max(dupd, 1));
  BLRM B11, B12, B21;
  double t0 = now();
  BLRM::construct_and_partial_factor(A11, A12, A21, A22, B11, B12, B21, t1, t2, A, o);
  stats[0] = now() - t0;
  stats[1] = B11.nonzeros(); stats[2] = B12.nonzeros(); stats[3] = B21.nonzeros();
  stats[4] = std::max(B11.rank(), std::max(B12.rank(), B21.rank()));
  const int nt = nt1 + nt2;
  for (int j = 0; j < nt; j++)
    for (int i = 0; i < nt; i++) {
      int r = -2;
      const BLR::BLRTile<double>* t = nullptr;
      if (i < nt1 && j < nt1) t = &B11.tile(i, j);
      else if (i < nt1) t = &B12.tile(i, j - nt1);
      else if (j < nt1) t = &B21.tile(i - nt1, j);
      if (t) r = t->is_low_rank() ? (int)t->rank() : -1;
      ranks[i + (size_t)j * nt] = r;
    }
  if (nrhs > 0) {
    DenseMatrixWrapper<double> bl(dsep, nrhs, bsep, std::max(dsep, 1)), bu(dupd, nrhs, bupd, std::max(dupd, 1));
    bl.laswp(B11.piv(), true);
    BLRM::trsmLNU_gemm(B11, B21, bl, bu, 0);
    DenseMatrixWrapper<double> yl(dsep, nrhs, ysep, std::max(dsep, 1));
    DenseMatrix<double> yu(dupd, nrhs, yupd, std::max(dupd, 1));
    BLRM::gemm_trsmUNN(B11, B12, yl, yu, 0);
  }
  return 0;
}

// ---- kernel-matrix front end: HSSMatrix(kernel::Kernel&, opts), HSS/HSSMatrix.cpp:88-106 ----
// data: d x n column-major (one point per column), copied; the reference reorders its copy while clustering.
struct RefKernelHSS {
  DenseMatrix<double> data;
  std::unique_ptr<kernel::Kernel<double>> K;
  std::unique_ptr<HSSMatrix<double>> H;
};
void* ref_kernel_hss_create(int n, int d, const double* data, int ktype, double h, double lambda, int p,
                            double rel_tol, double abs_tol, int leaf, int max_rank, int clustering,
                            int ann, int ann_iterations) {
  auto r = new RefKernelHSS();
  r->data = DenseMatrix<double>(d, n, data, d);
  r->K = kernel::create_kernel<double>(ktype == 1 ? kernel::KernelType::LAPLACE : (ktype == 2 ? kernel::KernelType::ANOVA : kernel::KernelType::GAUSS), r->data, h, lambda, p);
  HSSOptions<double> o;
  o.set_verbose(false);
  o.set_rel_tol(rel_tol); o.set_abs_tol(abs_tol); o.set_leaf_size(leaf); o.set_max_rank(max_rank);
  o.set_clustering_algorithm(clustering == 0 ? ClusteringAlgorithm::NATURAL : (clustering == 2 ? ClusteringAlgorithm::KD_TREE : ClusteringAlgorithm::TWO_MEANS));
  o.set_approximate_neighbors(ann);
  o.set_ann_iterations(ann_iterations);
  r->H.reset(new HSSMatrix<double>(*r->K, o));
  return r;
}
void ref_kernel_hss_destroy(void* h) { delete static_cast<RefKernelHSS*>(h); }
// the (re-ordered) data points the reference's kernel holds after construction, and its 1-based permutation
void ref_kernel_hss_data(void* h, double* data_out, int* perm_out) {
  auto r = static_cast<RefKernelHSS*>(h);
  std::memcpy(data_out, r->data.data(), sizeof(double) * r->data.rows() * r->data.cols());
  auto& pm = r->K->permutation();
  for (std::size_t i = 0; i < pm.size(); i++) perm_out[i] = pm[i];
}
int ref_kernel_hss_info(void* h, long long* out) {
  auto r = static_cast<RefKernelHSS*>(h);
  out[0] = r->H->is_compressed(); out[1] = r->H->levels(); out[2] = r->H->rank(); out[3] = r->H->memory();
  return 0;
}
int ref_kernel_hss_node_info(void* h, int* out, int cap_nodes) {
  RefHSS t; t.H = std::move(static_cast<RefKernelHSS*>(h)->H);
  int c = ref_hss_node_info(&t, out, cap_nodes);
  static_cast<RefKernelHSS*>(h)->H = std::move(t.H);
  return c;
}
void ref_kernel_hss_mult(void* h, int nrhs, const double* B, int ldb, double* C, int ldc) {
  auto r = static_cast<RefKernelHSS*>(h);
  int n = r->H->rows();
  DenseMatrixWrapper<double> b(n, nrhs, const_cast<double*>(B), ldb), c(n, nrhs, C, ldc);
  c.copy(r->H->apply(b));
}
void ref_kernel_hss_factor_solve(void* h, int nrhs, double* B, int ldb) {
  auto r = static_cast<RefKernelHSS*>(h);
  r->H->factor();
  DenseMatrixWrapper<double> b(r->H->rows(), nrhs, B, ldb);
  r->H->solve(b);
}
// exact kernel entries K(I, J) as the reference evaluates them on its (re-ordered) data
void ref_kernel_eval(void* h, int ni, const int* I, int nj, const int* J, double* out) {
  auto r = static_cast<RefKernelHSS*>(h);
  for (int j = 0; j < nj; j++)
    for (int i = 0; i < ni; i++) out[i + (size_t)j * ni] = r->K->eval(I[i], J[j]);
}
// Kernel::fit_HSS + Kernel::predict (kernel/KernelRegression.hpp:56-123), as examples/dense/KernelRegression.cpp drives them
void ref_kernel_regression(int n, int d, const double* train, const double* labels, int m, const double* test,
                           int ktype, double h, double lambda, int p, double rel_tol, double abs_tol, int leaf,
                           int clustering, int ann, double* weights_out, double* prediction_out, long long* info) {
  DenseMatrix<double> data(d, n, train, d);
  auto K = kernel::create_kernel<double>(ktype == 1 ? kernel::KernelType::LAPLACE : (ktype == 2 ? kernel::KernelType::ANOVA : kernel::KernelType::GAUSS), data, h, lambda, p);
  HSSOptions<double> o;
  o.set_verbose(false);
  o.set_rel_tol(rel_tol); o.set_abs_tol(abs_tol); o.set_leaf_size(leaf);
  o.set_clustering_algorithm(clustering == 0 ? ClusteringAlgorithm::NATURAL : (clustering == 2 ? ClusteringAlgorithm::KD_TREE : ClusteringAlgorithm::TWO_MEANS));
  o.set_approximate_neighbors(ann);
  std::vector<double> lab(labels, labels + n);
  double t0 = now();
  auto w = K->fit_HSS(lab, o);
  double t1 = now();
  DenseMatrix<double> tst(d, m, test, d);
  auto pred = K->predict(tst, w);
  double t2 = now();
  for (int i = 0; i < n; i++) weights_out[i] = w(i, 0);
  for (int i = 0; i < m; i++) prediction_out[i] = pred[i];
  info[0] = (long long)((t1 - t0) * 1e6); info[1] = (long long)((t2 - t1) * 1e6);
}
// binary_tree_clustering (clustering/Clustering.hpp:143-168): perm (1-based) and the leaf sizes of the tree, in order
int ref_clustering(int n, int d, double* data_inout, int algo, int leaf, int* perm_out, int* leaf_sizes, int cap) {
  DenseMatrix<double> p(d, n, data_inout, d);
  std::vector<int> perm;
  auto t = binary_tree_clustering(algo == 0 ? ClusteringAlgorithm::NATURAL : (algo == 2 ? ClusteringAlgorithm::KD_TREE : (algo == 4 ? ClusteringAlgorithm::COBBLE : (algo == 3 ? ClusteringAlgorithm::PCA : ClusteringAlgorithm::TWO_MEANS))), p, perm, leaf);
  std::memcpy(data_inout, p.data(), sizeof(double) * d * n);
  for (int i = 0; i < n; i++) perm_out[i] = perm[i];
  auto ls = t.template leaf_sizes<int>();
  int c = 0;
  for (auto s : ls) { if (c < cap) leaf_sizes[c] = s; c++; }
  return c;
}
// find_approximate_neighbors (clustering/NeighborSearch.cpp:324): ann is k x n (neighbour ids of point i in column i)
void ref_ann(int n, int d, const double* data, int iterations, int k, unsigned* ann_out, double* scores_out) {
  DenseMatrix<double> p(d, n, data, d);
  DenseMatrix<std::uint32_t> ann;
  DenseMatrix<double> scores;
  find_approximate_neighbors(p, iterations, k, ann, scores);
  for (int j = 0; j < n; j++)
    for (int i = 0; i < k; i++) { ann_out[i + (size_t)j * k] = ann(i, j); scores_out[i + (size_t)j * k] = scores(i, j); }
}

}  // extern "C"
