// C interface SP_d_struct_* (reference structured/StructuredMatrixC.cpp:61-119 handle + error
// convention, :83-821 entry points) and the SPX_* device-operand extensions.
#include "DevicePool.hpp"
#include <complex>
#include <cstdlib>
#include <iostream>

#include "BLRMatrix.hpp"
#include "HSSMatrix.hpp"
#include "Comm.hpp"
#include "Kernel.hpp"
#include "StructuredMatrix.hpp"
#include "structured/StructuredMatrix.h"

using namespace strumpack;
using namespace strumpack::structured;

namespace {
struct CStructMat {
  std::unique_ptr<StructuredMatrix<double>> S;
};
inline StructuredMatrix<double>* mat(const CSPStructMat S) { return static_cast<CStructMat*>(S)->S.get(); }
inline HSS::HSSMatrix<double>* hss(const CSPStructMat S) { return dynamic_cast<HSS::HSSMatrix<double>*>(mat(S)); }

template <typename T> struct CStructMatT {
  std::unique_ptr<StructuredMatrix<T>> S;
};
template <typename T> inline StructuredMatrix<T>* matT(const CSPStructMat S) { return static_cast<CStructMatT<T>*>(S)->S.get(); }
template <typename T> StructuredOptions<T> get_options_t(const CSPOptions* o) {
  StructuredOptions<T> opts;
  opts.set_type(Type(int(o->type)));
  opts.set_rel_tol(o->rel_tol);
  opts.set_abs_tol(o->abs_tol);
  opts.set_leaf_size(o->leaf_size);
  opts.set_max_rank(o->max_rank);
  opts.set_verbose(o->verbose != 0);
  return opts;
}
StructuredOptions<double> get_options(const CSPOptions* o) {
  StructuredOptions<double> opts;
  opts.set_type(Type(int(o->type)));
  opts.set_rel_tol(o->rel_tol);
  opts.set_abs_tol(o->abs_tol);
  opts.set_leaf_size(o->leaf_size);
  opts.set_max_rank(o->max_rank);
  opts.set_verbose(o->verbose != 0);
  return opts;
}
HSS::HSSOptions<double> get_hss_options(const CSPOptions* o, const SPXHSSOptions* h) {
  HSS::HSSOptions<double> ho(get_options(o));
  if (h) {
    ho.set_d0(h->d0); ho.set_dd(h->dd); ho.set_p(h->p);
    ho.set_compression_algorithm(h->compression_algorithm == 0 ? HSS::CompressionAlgorithm::ORIGINAL : (h->compression_algorithm == 2 ? HSS::CompressionAlgorithm::HARD_RESTART : HSS::CompressionAlgorithm::STABLE));
    ho.set_random_engine(h->random_engine == 0 ? random::RandomEngine::LINEAR : (h->random_engine == 1 ? random::RandomEngine::MERSENNE : random::RandomEngine::PHILOX));
    ho.set_random_distribution(h->random_distribution == 0 ? random::RandomDistribution::NORMAL : random::RandomDistribution::UNIFORM);
    ho.set_compression_sketch(h->compression_sketch == 1 ? HSS::CompressionSketch::SJLT : HSS::CompressionSketch::GAUSSIAN);
    ho.set_SJLT_algo(h->sjlt_algo == 1 ? HSS::SJLTAlgo::PERM : HSS::SJLTAlgo::CHUNK);
    if (h->nnz0 > 0) ho.set_nnz0(h->nnz0);
    if (h->nnz > 0) ho.set_nnz(h->nnz);
    ho.set_factor_ahead(h->factor_ahead != 0);
    ho.set_symmetric_operand(h->symmetric_operand);
  }
  return ho;
}
#define SP_TRY try {
#define SP_CATCH                                                      \
  }                                                                   \
  catch (std::exception & e) {                                        \
    std::cerr << "Operation failed: " << e.what() << std::endl;       \
    return 1;                                                         \
  }                                                                   \
  return 0;
}  // namespace

extern "C" {

void SP_d_struct_default_options(CSPOptions* o) {
  StructuredOptions<double> d;
  o->type = SP_STRUCTURED_TYPE(int(d.type()));
  o->rel_tol = d.rel_tol(); o->abs_tol = d.abs_tol(); o->leaf_size = d.leaf_size();
  o->max_rank = d.max_rank(); o->verbose = d.verbose();
}
void SP_d_struct_destroy(CSPStructMat* S) {
  delete static_cast<CStructMat*>(*S);
  *S = NULL;
}
int SP_d_struct_rows(const CSPStructMat S) { return int(mat(S)->rows()); }
int SP_d_struct_cols(const CSPStructMat S) { return int(mat(S)->cols()); }
long long int SP_d_struct_memory(const CSPStructMat S) { return (long long)mat(S)->memory(); }
long long int SP_d_struct_nonzeros(const CSPStructMat S) { return (long long)mat(S)->nonzeros(); }
int SP_d_struct_rank(const CSPStructMat S) { return int(mat(S)->rank()); }

int SP_d_struct_from_dense(CSPStructMat* S, int rows, int cols, const double* A, int ldA, const CSPOptions* opts) {
  SP_TRY
  std::unique_ptr<CStructMat> s(new CStructMat);
  s->S = construct_from_dense<double>(rows, cols, A, ldA, get_options(opts));
  *S = s.release();
  SP_CATCH
}
int SP_d_struct_from_elements(CSPStructMat* S, int rows, int cols, double A(int i, int j), const CSPOptions* opts) {
  SP_TRY
  std::unique_ptr<CStructMat> s(new CStructMat);
  extract_t<double> f = [A](std::size_t i, std::size_t j) { return A(int(i), int(j)); };
  s->S = construct_from_elements<double>(rows, cols, f, get_options(opts));
  *S = s.release();
  SP_CATCH
}
int SP_d_struct_mult(const CSPStructMat S, char trans, int m, const double* B, int ldB, double* C, int ldC) {
  SP_TRY
  mat(S)->mult(c2T(trans), m, B, ldB, C, ldC);
  SP_CATCH
}
int SP_d_struct_factor(CSPStructMat S) {
  SP_TRY
  mat(S)->factor();
  SP_CATCH
}
int SP_d_struct_solve(const CSPStructMat S, int nrhs, double* B, int ldB) {
  SP_TRY
  mat(S)->solve(nrhs, B, ldB);
  SP_CATCH
}
int SP_d_struct_shift(CSPStructMat S, double s) {
  SP_TRY
  mat(S)->shift(s);
  SP_CATCH
}

// ---- single precision and complex entry points (reference StructuredMatrix.h:103-602, StructuredMatrixC.cpp:83-821):
// same conventions as SP_d_*; carried by the double-precision device engine (HSSMatrixPromoted.hpp)
#define SPX_C_API(P, T, CT, TOCPP, REAL)                                                                             \
  void SP_##P##_struct_default_options(CSPOptions* o) {                                                              \
    StructuredOptions<T> d;                                                                                          \
    o->type = SP_STRUCTURED_TYPE(int(d.type()));                                                                     \
    o->rel_tol = d.rel_tol(); o->abs_tol = d.abs_tol(); o->leaf_size = d.leaf_size();                                \
    o->max_rank = d.max_rank(); o->verbose = d.verbose();                                                            \
  }                                                                                                                  \
  void SP_##P##_struct_destroy(CSPStructMat* S) { delete static_cast<CStructMatT<T>*>(*S); *S = NULL; }              \
  int SP_##P##_struct_rows(const CSPStructMat S) { return int(matT<T>(S)->rows()); }                                 \
  int SP_##P##_struct_cols(const CSPStructMat S) { return int(matT<T>(S)->cols()); }                                 \
  long long int SP_##P##_struct_memory(const CSPStructMat S) { return (long long)matT<T>(S)->memory(); }             \
  long long int SP_##P##_struct_nonzeros(const CSPStructMat S) { return (long long)matT<T>(S)->nonzeros(); }         \
  int SP_##P##_struct_rank(const CSPStructMat S) { return int(matT<T>(S)->rank()); }                                 \
  int SP_##P##_struct_from_dense(CSPStructMat* S, int rows, int cols, const CT* A, int ldA, const CSPOptions* opts) { \
    SP_TRY                                                                                                           \
    std::unique_ptr<CStructMatT<T>> s(new CStructMatT<T>);                                                           \
    s->S = construct_from_dense<T>(rows, cols, reinterpret_cast<const T*>(A), ldA, get_options_t<T>(opts));          \
    *S = s.release();                                                                                                \
    SP_CATCH                                                                                                         \
  }                                                                                                                  \
  int SP_##P##_struct_from_elements(CSPStructMat* S, int rows, int cols, CT A(int i, int j), const CSPOptions* opts) { \
    SP_TRY                                                                                                           \
    std::unique_ptr<CStructMatT<T>> s(new CStructMatT<T>);                                                           \
    extract_t<T> f = [A](std::size_t i, std::size_t j) { const CT v = A(int(i), int(j)); return TOCPP(v); };         \
    s->S = construct_from_elements<T>(rows, cols, f, get_options_t<T>(opts));                                        \
    *S = s.release();                                                                                                \
    SP_CATCH                                                                                                         \
  }                                                                                                                  \
  int SP_##P##_struct_mult(const CSPStructMat S, char trans, int m, const CT* B, int ldB, CT* C, int ldC) {          \
    SP_TRY                                                                                                           \
    matT<T>(S)->mult(c2T(trans), m, reinterpret_cast<const T*>(B), ldB, reinterpret_cast<T*>(C), ldC);               \
    SP_CATCH                                                                                                         \
  }                                                                                                                  \
  int SP_##P##_struct_factor(CSPStructMat S) {                                                                       \
    SP_TRY                                                                                                           \
    matT<T>(S)->factor();                                                                                            \
    SP_CATCH                                                                                                         \
  }                                                                                                                  \
  int SP_##P##_struct_solve(const CSPStructMat S, int nrhs, CT* B, int ldB) {                                        \
    SP_TRY                                                                                                           \
    matT<T>(S)->solve(nrhs, reinterpret_cast<T*>(B), ldB);                                                           \
    SP_CATCH                                                                                                         \
  }                                                                                                                  \
  int SP_##P##_struct_shift(CSPStructMat S, CT s) {                                                                  \
    SP_TRY                                                                                                           \
    matT<T>(S)->shift(TOCPP(s));                                                                                     \
    SP_CATCH                                                                                                         \
  }
#define SPX_ID(v) (v)
#define SPX_CF(v) std::complex<float>(__real__(v), __imag__(v))
#define SPX_CD(v) std::complex<double>(__real__(v), __imag__(v))
SPX_C_API(s, float, float, SPX_ID, float)
SPX_C_API(c, std::complex<float>, float _Complex, SPX_CF, float)
SPX_C_API(z, std::complex<double>, double _Complex, SPX_CD, double)
#undef SPX_C_API

// ---- extensions ------------------------------------------------------------------------------
// structured::construct_and_factor_from_dense (structured/StructuredMatrix.hpp:536-553) -- the route to a factored BLR
// matrix (its LU is computed while the tiles are compressed, BLRMatrix::compress_and_factor); HSS: construct + factor
int SPX_d_struct_from_dense_and_factor(CSPStructMat* S, int rows, int cols, const double* A, int ldA, const CSPOptions* opts) {
  SP_TRY
  std::unique_ptr<CStructMat> s(new CStructMat);
  auto M = ConstDenseMatrixWrapper<double>(rows, cols, A, ldA);
  s->S = construct_and_factor_from_dense<double>(M, get_options(opts));
  *S = s.release();
  SP_CATCH
}
void SPX_d_struct_default_hss_options(SPXHSSOptions* h) {
  HSS::HSSOptions<double> d;
  h->d0 = d.d0(); h->dd = d.dd(); h->p = d.p();
  h->compression_algorithm = 1; h->random_engine = 0; h->random_distribution = 0;
  h->compression_sketch = 0; h->sjlt_algo = 0; h->nnz0 = d.nnz0(); h->nnz = d.nnz();
  h->factor_ahead = 0;
  h->symmetric_operand = 0;
}
int SPX_d_struct_from_dense_hss(CSPStructMat* S, int rows, int cols, const double* A, int ldA, const CSPOptions* opts,
                                const SPXHSSOptions* h) {
  SP_TRY
  if (opts->type != SP_TYPE_HSS) throw std::invalid_argument("SPX_d_struct_from_dense_hss requires type SP_TYPE_HSS");
  if (rows != cols) throw std::invalid_argument("HSS compression only supported for square matrices.");
  auto ho = get_hss_options(opts, h);
  std::unique_ptr<HSS::HSSMatrix<double>> H(new HSS::HSSMatrix<double>(rows, cols, ho));
  auto M = ConstDenseMatrixWrapper<double>(rows, cols, A, ldA);
  H->compress(M, ho);
  std::unique_ptr<CStructMat> s(new CStructMat);
  s->S.reset(H.release());
  *S = s.release();
  SP_CATCH
}
int SPX_d_struct_from_dense_device(CSPStructMat* S, int rows, int cols, const double* dA, long long ldA,
                                   const CSPOptions* opts, const SPXHSSOptions* h) {
  SP_TRY
  if (opts->type != SP_TYPE_HSS) throw std::invalid_argument("SPX_d_struct_from_dense_device requires type SP_TYPE_HSS");
  if (rows != cols) throw std::invalid_argument("HSS compression only supported for square matrices.");
  auto ho = get_hss_options(opts, h);
  std::unique_ptr<HSS::HSSMatrix<double>> H(new HSS::HSSMatrix<double>(rows, cols, ho));
  H->compress_device(dA, ldA, ho);
  std::unique_ptr<CStructMat> s(new CStructMat);
  s->S.reset(H.release());
  *S = s.release();
  SP_CATCH
}
int SPX_d_struct_from_dense_device_sharded(CSPStructMat* S, int rows, int cols, const double* dA, long long ldA,
                                           const CSPOptions* opts, const SPXHSSOptions* h, int world, int rank,
                                           SPXAllGatherFn exchange, void* user) {
  SP_TRY
  if (opts->type != SP_TYPE_HSS) throw std::invalid_argument("sharded construction requires type SP_TYPE_HSS");
  if (rows != cols) throw std::invalid_argument("HSS compression only supported for square matrices.");
  if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("invalid world/rank");
  auto ho = get_hss_options(opts, h);
  std::unique_ptr<HSS::HSSMatrix<double>> H(new HSS::HSSMatrix<double>(rows, cols, ho));
  H->compress_device_sharded(dA, ldA, ho, world, rank, exchange, user);
  std::unique_ptr<CStructMat> s(new CStructMat);
  s->S.reset(H.release());
  *S = s.release();
  SP_CATCH
}
// HSS approximation of the kernel matrix over `points` (d x n): clusters (reordering points in place, perm 1-based),
// builds the tree, compresses from the coordinates (HSSMatrix(kernel::Kernel&, opts), HSS/HSSMatrix.cpp:88-106)
static int from_kernel_group(CSPStructMat* S, int n, int d, double* points, int ktype, double h, double lambda, int p,
                             const CSPOptions* opts, int clustering, int neighbors, int* perm, const HSS::CommSpec& pg) {
  SP_TRY
  if (opts->type != SP_TYPE_HSS) throw std::invalid_argument("kernel construction requires type SP_TYPE_HSS");
  if (pg.world < 1 || pg.rank < 0 || pg.rank >= pg.world) throw std::invalid_argument("invalid world/rank");
  static const ClusteringAlgorithm algos[] = {ClusteringAlgorithm::NATURAL, ClusteringAlgorithm::TWO_MEANS, ClusteringAlgorithm::KD_TREE,
                                              ClusteringAlgorithm::PCA, ClusteringAlgorithm::COBBLE};
  if (clustering < 0 || clustering > 4) throw std::invalid_argument("clustering algorithm out of range");
  auto ho = get_hss_options(opts, nullptr);
  ho.set_clustering_algorithm(algos[clustering]);
  if (neighbors > 0) ho.set_approximate_neighbors(neighbors);
  DenseMatrix<double> X(d, n, points, d);
  std::unique_ptr<kernel::Kernel<double>> K;
  switch (ktype) {
    case 0: K.reset(new kernel::GaussKernel<double>(X, h, lambda)); break;
    case 1: K.reset(new kernel::LaplaceKernel<double>(X, h, lambda)); break;
    case 2: K.reset(new kernel::ANOVAKernel<double>(X, h, lambda, p)); break;
    default: throw std::invalid_argument("kernel type must be 0 (Gauss), 1 (Laplace) or 2 (ANOVA)");
  }
  std::unique_ptr<HSS::HSSMatrix<double>> H(new HSS::HSSMatrix<double>(*K, ho, pg));
  std::copy(X.data(), X.data() + (size_t)d * n, points);
  if (perm) std::copy(K->permutation().begin(), K->permutation().end(), perm);
  std::unique_ptr<CStructMat> s(new CStructMat);
  s->S.reset(H.release());
  *S = s.release();
  SP_CATCH
}
static HSS::CommSpec callback_group(int world, int rank, SPXAllGatherFn fn, void* user) {
  HSS::CommSpec pg;
  pg.world = world; pg.rank = rank; pg.allgather = fn; pg.user = user; pg.native = false;
  return pg;
}
static HSS::CommSpec native_group(SPXComm comm) {
  if (!comm) throw std::invalid_argument("null communicator");
  auto* c = (comm::RcclComm*)comm;
  HSS::CommSpec pg;
  pg.world = c->world(); pg.rank = c->rank(); pg.user = c; pg.native = true;
  return pg;
}
int SPX_d_struct_from_kernel_sharded(CSPStructMat* S, int n, int d, double* points, int ktype, double h, double lambda, int p,
                                     const CSPOptions* opts, int clustering, int neighbors, int* perm, int world, int rank,
                                     SPXAllGatherFn allgather, void* user) {
  return from_kernel_group(S, n, d, points, ktype, h, lambda, p, opts, clustering, neighbors, perm, callback_group(world, rank, allgather, user));
}
int SPX_d_struct_from_kernel_comm(CSPStructMat* S, int n, int d, double* points, int ktype, double h, double lambda, int p,
                                  const CSPOptions* opts, int clustering, int neighbors, int* perm, SPXComm comm) {
  SP_TRY
  return from_kernel_group(S, n, d, points, ktype, h, lambda, p, opts, clustering, neighbors, perm, native_group(comm));
  SP_CATCH
}
// ---- native process group (Comm.hpp)
int SPX_comm_unique_id(char id[128]) {
  SP_TRY
  comm::RcclComm::unique_id(id);
  SP_CATCH
}
int SPX_comm_create(SPXComm* out, int world, int rank, const char id[128]) {
  SP_TRY
  *out = new comm::RcclComm(world, rank, id);
  SP_CATCH
}
void SPX_comm_destroy(SPXComm* c) {
  if (c && *c) { delete (comm::RcclComm*)*c; *c = nullptr; }
}
// exercises the three collectives the engine uses on small device buffers and checks the results on every rank
int SPX_comm_selftest(SPXComm comm) {
  SP_TRY
  auto* c = (comm::RcclComm*)comm;
  if (!c) throw std::invalid_argument("null communicator");
  const int G = c->world(), me = c->rank(), L = 1000;
  hssk_ctx* ctx = nullptr;
  int dev = 0;
  if (const char* e = std::getenv("STRUMPACK_AMD_DEVICE")) dev = std::atoi(e);   // (the device the engines of this process use)
  if (hssk_ctx_create(&ctx, dev)) throw std::runtime_error(hssk_last_error());
  struct Guard { hssk_ctx* c; std::vector<void*> p; ~Guard() { for (void* q : p) hssk_free(q); hssk_ctx_destroy(c); } } g{ctx, {}};
  auto dmal = [&](size_t n) { void* p = hssk_malloc((long long)(sizeof(double) * n)); if (!p) throw std::runtime_error("selftest: allocation failed"); g.p.push_back(p); return (double*)p; };
  void* stream = hssk_ctx_stream(ctx);
  // all-gather: block r holds r + 1 everywhere
  std::vector<double> h((size_t)L * G, 0.), out((size_t)L * G);
  for (int i = 0; i < L; i++) h[(size_t)L * me + i] = me + 1.;
  double* d = dmal((size_t)L * G);
  if (hssk_memcpy_h2d(ctx, d, h.data(), (long long)(sizeof(double) * L * G))) throw std::runtime_error(hssk_last_error());
  c->allgather(d, (long long)sizeof(double) * L, stream);
  if (hssk_memcpy_d2h(ctx, out.data(), d, (long long)(sizeof(double) * L * G))) throw std::runtime_error(hssk_last_error());
  for (int r = 0; r < G; r++) for (int i = 0; i < L; i++) if (out[(size_t)L * r + i] != r + 1.) throw std::runtime_error("selftest: all-gather mismatch");
  // all-reduce: every rank contributes rank + 1 -> G (G + 1) / 2
  for (auto& v : h) v = me + 1.;
  if (hssk_memcpy_h2d(ctx, d, h.data(), (long long)(sizeof(double) * L * G))) throw std::runtime_error(hssk_last_error());
  c->allreduce_sum(d, (long long)L * G, stream);
  if (hssk_memcpy_d2h(ctx, out.data(), d, (long long)(sizeof(double) * L * G))) throw std::runtime_error(hssk_last_error());
  for (auto v : out) if (v != G * (G + 1) / 2.) throw std::runtime_error("selftest: all-reduce mismatch");
  // reduce-scatter with per-rank counts: rank r receives r + 1 entries of value sum_g (g + 1) * (offset + i)
  std::vector<long long> offs(G), cnts(G);
  long long tot = 0;
  for (int r = 0; r < G; r++) { offs[r] = tot; cnts[r] = r + 1; tot += r + 1; }
  std::vector<double> hs(tot), hr(cnts[me]);
  for (long long i = 0; i < tot; i++) hs[i] = (me + 1.) * (double)i;
  double* ds = dmal(tot);
  double* dr = dmal(cnts[me]);
  if (hssk_memcpy_h2d(ctx, ds, hs.data(), (long long)(sizeof(double) * tot))) throw std::runtime_error(hssk_last_error());
  c->reduce_scatter_sum(ds, offs.data(), cnts.data(), dr, stream);
  if (hssk_memcpy_d2h(ctx, hr.data(), dr, (long long)(sizeof(double) * cnts[me]))) throw std::runtime_error(hssk_last_error());
  for (long long i = 0; i < cnts[me]; i++) if (hr[i] != G * (G + 1) / 2. * (double)(offs[me] + i)) throw std::runtime_error("selftest: reduce-scatter mismatch");
  SP_CATCH
}
int SPX_comm_size(const SPXComm c) { return c ? ((comm::RcclComm*)c)->world() : 1; }
int SPX_comm_rank(const SPXComm c) { return c ? ((comm::RcclComm*)c)->rank() : 0; }
int SPX_struct_shard_range(int n, const CSPOptions* opts, int world, int rank, int* lo, int* hi) {
  SP_TRY
  if (world < 1 || rank < 0 || rank >= world || (world & (world - 1))) throw std::invalid_argument("shard_range: world must be a power of two, 0 <= rank < world");
  int c = 0;
  while ((1 << c) < world) c++;
  // the bisection of HSSMatrix(m, n, opts) (HSS/HSSMatrix.cpp:60-70): children of size m/2 and m - m/2 while m > leaf
  int l = 0, m = n;
  for (int lev = 0; lev < c; lev++) {
    if (m <= opts->leaf_size) throw std::invalid_argument("shard_range: the tree is shallower than log2(world) levels");
    const int m0 = m / 2;
    if ((rank >> (c - 1 - lev)) & 1) { l += m0; m -= m0; } else m = m0;
  }
  *lo = l; *hi = l + m;
  SP_CATCH
}
static int from_blocks(CSPStructMat* S, int rows, int cols, const double* dArows, long long ldr, const double* dAcols,
                       long long ldc, const CSPOptions* opts, const SPXHSSOptions* h, const HSS::CommSpec& pg) {
  SP_TRY
  if (opts->type != SP_TYPE_HSS) throw std::invalid_argument("sharded construction requires type SP_TYPE_HSS");
  if (rows != cols) throw std::invalid_argument("HSS compression only supported for square matrices.");
  auto ho = get_hss_options(opts, h);
  std::unique_ptr<HSS::HSSMatrix<double>> H(new HSS::HSSMatrix<double>(rows, cols, ho));
  H->compress_device_blocks(dArows, ldr, dAcols, ldc, ho, pg);
  std::unique_ptr<CStructMat> s(new CStructMat);
  s->S.reset(H.release());
  *S = s.release();
  SP_CATCH
}
int SPX_d_struct_from_blocks_device(CSPStructMat* S, int rows, int cols, const double* dArows, long long ldr,
                                    const double* dAcols, long long ldc, const CSPOptions* opts, const SPXHSSOptions* h,
                                    SPXComm comm) {
  SP_TRY
  return from_blocks(S, rows, cols, dArows, ldr, dAcols, ldc, opts, h, native_group(comm));
  SP_CATCH
}
int SPX_d_struct_from_blocks_device_cb(CSPStructMat* S, int rows, int cols, const double* dArows, long long ldr,
                                       const double* dAcols, long long ldc, const CSPOptions* opts, const SPXHSSOptions* h,
                                       int world, int rank, SPXAllGatherFn allgather, void* user) {
  if (world < 1 || rank < 0 || rank >= world) return 1;
  return from_blocks(S, rows, cols, dArows, ldr, dAcols, ldc, opts, h, callback_group(world, rank, allgather, user));
}
int SPX_d_struct_from_dense_device_comm(CSPStructMat* S, int rows, int cols, const double* dA, long long ldA,
                                        const CSPOptions* opts, const SPXHSSOptions* h, SPXComm comm) {
  SP_TRY
  if (opts->type != SP_TYPE_HSS) throw std::invalid_argument("sharded construction requires type SP_TYPE_HSS");
  if (rows != cols) throw std::invalid_argument("HSS compression only supported for square matrices.");
  auto ho = get_hss_options(opts, h);
  std::unique_ptr<HSS::HSSMatrix<double>> H(new HSS::HSSMatrix<double>(rows, cols, ho));
  H->compress_device_sharded(dA, ldA, ho, native_group(comm));
  std::unique_ptr<CStructMat> s(new CStructMat);
  s->S.reset(H.release());
  *S = s.release();
  SP_CATCH
}
// the matrix is a formula of the library's (hssk_gen kinds): never stored, on one rank or, with a communicator, on many
static int from_generator(CSPStructMat* S, int n, int kind, const CSPOptions* opts, const SPXHSSOptions* h, const HSS::CommSpec* pg) {
  SP_TRY
  if (opts->type != SP_TYPE_HSS) throw std::invalid_argument("SPX_d_struct_from_generator requires type SP_TYPE_HSS");
  if (kind != HSSK_GEN_TOEPLITZ && kind != HSSK_GEN_TOEPLITZ_UPPER) throw std::invalid_argument("SPX_d_struct_from_generator: unknown generator kind");
  auto ho = get_hss_options(opts, h);
  std::unique_ptr<HSS::HSSMatrix<double>> H(new HSS::HSSMatrix<double>(n, n, ho));
  if (pg) H->compress_generator(kind, ho, *pg);
  else H->compress_generator(kind, ho);
  std::unique_ptr<CStructMat> s(new CStructMat);
  s->S.reset(H.release());
  *S = s.release();
  SP_CATCH
}
int SPX_d_struct_from_generator(CSPStructMat* S, int n, int kind, const CSPOptions* opts, const SPXHSSOptions* h) {
  return from_generator(S, n, kind, opts, h, nullptr);
}
int SPX_d_struct_from_generator_comm(CSPStructMat* S, int n, int kind, const CSPOptions* opts, const SPXHSSOptions* h, SPXComm comm) {
  const HSS::CommSpec pg = native_group(comm);
  return from_generator(S, n, kind, opts, h, &pg);
}
int SPX_d_struct_from_generator_sharded(CSPStructMat* S, int n, int kind, const CSPOptions* opts, const SPXHSSOptions* h,
                                        int world, int rank, SPXAllGatherFn allgather, void* user) {
  const HSS::CommSpec pg = callback_group(world, rank, allgather, user);
  return from_generator(S, n, kind, opts, h, &pg);
}
int SPX_d_struct_from_kernel(CSPStructMat* S, int n, int d, double* points, int ktype, double h, double lambda, int p,
                             const CSPOptions* opts, int clustering, int neighbors, int* perm) {
  return SPX_d_struct_from_kernel_sharded(S, n, d, points, ktype, h, lambda, p, opts, clustering, neighbors, perm, 1, 0, nullptr, nullptr);
}
int SPX_d_struct_mult_device(const CSPStructMat S, char trans, int m, const double* dB, long long ldB, double* dC, long long ldC) {
  SP_TRY
  if (!hss(S)) throw std::invalid_argument("not an HSS matrix");
  hss(S)->mult_device(c2T(trans), m, dB, ldB, dC, ldC);
  SP_CATCH
}
int SPX_d_struct_solve_device(const CSPStructMat S, int nrhs, double* dB, long long ldB) {
  SP_TRY
  if (!hss(S)) throw std::invalid_argument("not an HSS matrix");
  hss(S)->solve_device(nrhs, dB, ldB);
  SP_CATCH
}
int SPX_d_struct_levels(const CSPStructMat S) { return hss(S) ? int(hss(S)->levels()) : 0; }
int SPX_d_struct_is_compressed(const CSPStructMat S) { return hss(S) ? int(hss(S)->is_compressed()) : 0; }
int SPX_d_struct_num_nodes(const CSPStructMat S) { return hss(S) ? hss(S)->engine()->num_nodes() : 0; }
int SPX_d_struct_node_info(const CSPStructMat S, int* out) {
  SP_TRY
  if (!hss(S)) throw std::invalid_argument("not an HSS matrix");
  hss(S)->engine()->node_info(out);
  SP_CATCH
}
int SPX_d_struct_stats(const CSPStructMat S, double* o) {
  SP_TRY
  if (!hss(S)) throw std::invalid_argument("not an HSS matrix");
  const HSS::PhaseStats& st = hss(S)->engine()->stats();
  o[0] = st.t_compress; o[1] = st.t_sketch; o[2] = st.t_random; o[3] = st.t_tree; o[4] = st.t_factor;
  o[5] = st.t_solve; o[6] = st.t_mult; o[7] = st.sketch_kernel_ms; o[8] = st.sketch_launches; o[9] = st.rounds;
  o[10] = st.d_final; o[11] = st.f_sketch; o[12] = st.f_local; o[13] = st.f_reduce; o[14] = st.f_id;
  o[15] = st.f_ortho; o[16] = st.f_ulv; o[17] = st.f_solve; o[18] = (double)hss(S)->engine()->factor_memory();
  o[19] = st.sketch_kernel_flops;
  o[20] = st.sketch_kernel_bytes;
  o[21] = st.b_solve; o[22] = st.b_mult;
  o[23] = st.t_comm;
  SP_CATCH
}
// ---- Schur complement of the (0,0) block (HSS only; HSSMatrix.Schur.hpp)
int SPX_d_struct_partial_factor(CSPStructMat S) {
  SP_TRY
  if (!hss(S)) throw std::invalid_argument("not an HSS matrix");
  hss(S)->partial_factor();
  SP_CATCH
}
int SPX_d_struct_schur_dims(const CSPStructMat S, int* out) {
  SP_TRY
  if (!hss(S)) throw std::invalid_argument("not an HSS matrix");
  const auto d = hss(S)->engine()->schur_dims();
  out[0] = d.n0; out[1] = d.n1; out[2] = d.rV0; out[3] = d.mu0; out[4] = d.rV1; out[5] = d.rU0; out[6] = d.rU1;
  SP_CATCH
}
int SPX_d_struct_schur_update(CSPStructMat S, double* Theta, int ldT, double* DUB01, int ldD, double* Phi, int ldP,
                              double* Vhat, int ldV) {
  SP_TRY
  if (!hss(S)) throw std::invalid_argument("not an HSS matrix");
  hss(S)->engine()->schur_update(Theta, ldT, DUB01, ldD, Phi, ldP, Vhat, ldV);
  SP_CATCH
}
int SPX_d_struct_schur_product_direct(const CSPStructMat S, int c, const double* R, long long ldR, double* Sr,
                                      long long ldSr, double* Sc, long long ldSc, int on_device) {
  SP_TRY
  if (!hss(S)) throw std::invalid_argument("not an HSS matrix");
  hss(S)->engine()->schur_product_direct(c, R, ldR, Sr, ldSr, Sc, ldSc, on_device != 0);
  SP_CATCH
}
int SPX_d_struct_schur_product_indirect(const CSPStructMat S, int c, const double* R0, long long ldR0, const double* R1,
                                        long long ldR1, const double* Sr1, long long ldSr1, const double* Sc1,
                                        long long ldSc1, double* Sr, long long ldSr, double* Sc, long long ldSc,
                                        int on_device) {
  SP_TRY
  if (!hss(S)) throw std::invalid_argument("not an HSS matrix");
  hss(S)->engine()->schur_product_indirect(c, R0, ldR0, R1, ldR1, Sr1, ldSr1, Sc1, ldSc1, Sr, ldSr, Sc, ldSc, on_device != 0);
  SP_CATCH
}
int SPX_d_struct_mult_child(const CSPStructMat S, int child, char trans, int m, const double* B, long long ldB, double* C,
                            long long ldC, int on_device) {
  SP_TRY
  if (!hss(S)) throw std::invalid_argument("not an HSS matrix");
  hss(S)->engine()->mult_child(child, trans, m, B, ldB, C, ldC, on_device != 0);
  SP_CATCH
}
int SPX_d_struct_extract_blocks(const CSPStructMat S, int nb, const int* rows, const int* roff, const int* cols, const int* coff,
                                double* const* out, const int* ldo, int add, int on_device) {
  SP_TRY
  if (!hss(S)) throw std::invalid_argument("not an HSS matrix");
  hss(S)->engine()->extract_blocks(0, nb, rows, roff, cols, coff, out, ldo, on_device != 0, add != 0);
  SP_CATCH
}

// ---- BLR frontal matrix (BLRMatrix::construct_and_partial_factor, BLR/BLRMatrix.cpp:740-1037) -------------------------
extern "C++" {
namespace {
bool g_blr_time_phases = false;
int g_blr_low_rank_algorithm = 0;   // SPX_blr_low_rank_algorithm: what the BLR fronts made through this interface compress their tiles with
std::unique_ptr<BLR::DeviceBLR> make_front(int dsep, int dupd, int nt1, const int* t1, int nt2, const int* t2, const CSPOptions* opts) {
  if (dsep < 0 || dupd < 0 || nt1 < 0 || nt2 < 0 || (nt1 && !t1) || (nt2 && !t2)) throw std::invalid_argument("BLR front: bad dimensions");
  std::vector<int> tiles(t1, t1 + nt1);
  tiles.insert(tiles.end(), t2, t2 + nt2);
  long long s1 = 0, s2 = 0;
  for (int i = 0; i < nt1; i++) { if (t1[i] < 0) throw std::invalid_argument("BLR front: negative tile size"); s1 += t1[i]; }
  for (int i = 0; i < nt2; i++) { if (t2[i] < 0) throw std::invalid_argument("BLR front: negative tile size"); s2 += t2[i]; }
  if (s1 != dsep || s2 != dupd) throw std::invalid_argument("BLR front: the tile sizes do not add up to dsep / dupd");
  BLR::BLREngineOptions e;
  BLR::BLROptions<double> d;
  e.rel_tol = opts ? opts->rel_tol : d.rel_tol();
  e.abs_tol = opts ? opts->abs_tol : d.abs_tol();
  e.max_rank = opts ? opts->max_rank : d.max_rank();
  e.lr_algo = g_blr_low_rank_algorithm;
  e.verbose = opts && opts->verbose;
  if (const char* dv = std::getenv("STRUMPACK_AMD_DEVICE")) e.device = std::atoi(dv);
  std::unique_ptr<BLR::DeviceBLR> f(new BLR::DeviceBLR(dsep + dupd, tiles, dsep + dupd, tiles, e));
  f->time_phases = g_blr_time_phases;
  return f;
}
inline BLR::DeviceBLR* front(const SPXBLRFront F) {
  if (!F) throw std::invalid_argument("BLR front: null handle");
  return static_cast<BLR::DeviceBLR*>(F);
}
}  // namespace
}  // extern "C++"
void SPX_d_blr_front_time_phases(int on) { g_blr_time_phases = on != 0; }
int SPX_blr_low_rank_algorithm(int algo) {
  if (algo != 0 && algo != 1) return 1;   // 0 RRQR, 1 ACA (BLR::LowRankAlgorithm; BACA is not available)
  g_blr_low_rank_algorithm = algo;
  return 0;
}
int SPX_d_blr_front_factor(SPXBLRFront* F, int dsep, int dupd, const double* F11, int ld11, const double* F12, int ld12,
                           const double* F21, int ld21, double* F22, int ld22, int ntiles1, const int* tiles1, int ntiles2,
                           const int* tiles2, const char* admissible, const CSPOptions* opts) {
  SP_TRY
  auto f = make_front(dsep, dupd, ntiles1, tiles1, ntiles2, tiles2, opts);
  f->partial_factor_host(ntiles1, F11, ld11, F12, ld12, F21, ld21, F22, ld22, admissible);
  if (F22 && dupd > 0) f->schur_host(F22, ld22);
  *F = f.release();
  SP_CATCH
}
int SPX_d_blr_front_factor_device(SPXBLRFront* F, int dsep, int dupd, const double* dF11, long long ld11, const double* dF12,
                                  long long ld12, const double* dF21, long long ld21, const double* dF22, long long ld22,
                                  int ntiles1, const int* tiles1, int ntiles2, const int* tiles2, const char* admissible,
                                  const CSPOptions* opts) {
  SP_TRY
  auto f = make_front(dsep, dupd, ntiles1, tiles1, ntiles2, tiles2, opts);
  f->partial_factor_device(ntiles1, dF11, ld11, dF12, ld12, dF21, ld21, dF22, ld22, admissible);
  *F = f.release();
  SP_CATCH
}
int SPX_d_blr_front_forward(const SPXBLRFront F, int nrhs, double* bsep, int ldb, double* bupd, int ldu) {
  SP_TRY
  front(F)->front_forward(nrhs, bsep, ldb, bupd, ldu);
  SP_CATCH
}
int SPX_d_blr_front_backward(const SPXBLRFront F, int nrhs, double* ysep, int ldy, const double* yupd, int ldu) {
  SP_TRY
  front(F)->front_backward(nrhs, ysep, ldy, yupd, ldu);
  SP_CATCH
}
int SPX_d_blr_front_schur(const SPXBLRFront F, double* F22, int ld22) {
  SP_TRY
  front(F)->schur_host(F22, ld22);
  SP_CATCH
}
const double* SPX_d_blr_front_schur_device(const SPXBLRFront F, long long* ld) {
  if (!F) return nullptr;
  auto* f = static_cast<BLR::DeviceBLR*>(F);
  if (ld) *ld = f->schur_ld();
  return f->upd_rows() > 0 ? f->schur_device() : nullptr;
}
int SPX_d_blr_front_tile_ranks(const SPXBLRFront F, int* out) {
  SP_TRY
  front(F)->tile_ranks(out);
  SP_CATCH
}
int SPX_d_blr_front_stats(const SPXBLRFront F, double* out) {
  SP_TRY
  auto* f = front(F);
  long long nz[3];
  f->front_nonzeros(nz);
  out[0] = f->t_factor;
  for (int p = 0; p < 4; p++) out[1 + p] = f->phase_ms[p];
  out[5] = f->f_schur; out[6] = f->f_total;
  out[7] = (double)nz[0]; out[8] = (double)nz[1]; out[9] = (double)nz[2];
  out[10] = f->rank(); out[11] = f->schur_launches; out[12] = f->b_schur;
  SP_CATCH
}
void SPX_d_blr_front_destroy(SPXBLRFront* F) {
  if (!F) return;
  delete static_cast<BLR::DeviceBLR*>(*F);
  *F = nullptr;
}

void* SPX_d_struct_hssk_ctx(const CSPStructMat S) { return hss(S) ? (void*)hss(S)->engine()->ctx() : nullptr; }

// compression rounds whose inner tree levels ran as one launch / that fell back after a rank above the launch's bound
long long SPX_tree_pass_launches(void) { return strumpack::HSS::tree_pass_launches(); }
long long SPX_tree_pass_fallbacks(void) { return strumpack::HSS::tree_pass_fallbacks(); }

// The process-wide cache of device chunks (DevicePool.hpp): what it holds is free memory only this library can see
long long SPX_device_pool_cached_bytes(void) { return (long long)strumpack::DevicePool::get().cached(); }
long long SPX_device_pool_limit_bytes(void) { return (long long)strumpack::DevicePool::get().limit(); }
void SPX_device_pool_trim(void) { strumpack::DevicePool::get().trim(); }
void SPX_device_pool_set_limit_gb(double gb) {
  strumpack::DevicePool::get().set_limit(gb <= 0 ? 0 : (size_t)(gb * 1073741824.0));
}

}  // extern "C"
