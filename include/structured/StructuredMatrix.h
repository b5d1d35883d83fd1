/* structured/StructuredMatrix.h -- the C interface of the rank-structured dense solver, as bound by
 * C / Fortran / Python users of pghysels/STRUMPACK (reference: src/structured/StructuredMatrix.h:46-602,
 * implementation src/structured/StructuredMatrixC.cpp:83-821).  Double-precision real entry points
 * (SP_d_struct_*), same names, argument order, ownership and 0/1 return convention.
 *
 * This build implements type SP_TYPE_HSS (the MI355X HSS engine) and, for dense operands, SP_TYPE_BLR (block low-rank:
 * from_dense + mult, and compress-and-factor + solve); the other types return 1 with "Operation failed: ..." on stderr,
 * exactly like a reference build configured without them.
 * Caller buffers are HOST memory (reference rule, doc/doxygen/pages/GPU_support.txt:24-26); the
 * SPX_* entry points at the end are extensions for operands already resident in HBM.
 */
#ifndef STRUCTURED_MATRIX_H
#define STRUCTURED_MATRIX_H

typedef enum {
  SP_TYPE_HSS = 0,
  SP_TYPE_BLR,
  SP_TYPE_HODLR,
  SP_TYPE_HODBF,
  SP_TYPE_BUTTERFLY,
  SP_TYPE_LR,
  SP_TYPE_LOSSY,
  SP_TYPE_LOSSLESS
} SP_STRUCTURED_TYPE; /* reference StructuredMatrix.h:46-55 */

typedef struct CSPOptions {
  SP_STRUCTURED_TYPE type;
  double rel_tol;
  double abs_tol;
  int leaf_size;
  int max_rank;
  int verbose;
} CSPOptions; /* reference :68-75 */

typedef void* CSPStructMat; /* reference :85 */

#ifdef __cplusplus
extern "C" {
#endif

/* reference :109 */
void SP_d_struct_default_options(CSPOptions* opts);
/* reference :137 -- frees the matrix and sets *S to NULL */
void SP_d_struct_destroy(CSPStructMat* S);
/* reference :163, :185, :211, :239, :265 */
int SP_d_struct_rows(const CSPStructMat S);
int SP_d_struct_cols(const CSPStructMat S);
long long int SP_d_struct_memory(const CSPStructMat S);
long long int SP_d_struct_nonzeros(const CSPStructMat S);
int SP_d_struct_rank(const CSPStructMat S);
/* reference :313 -- A is rows x cols column-major with leading dimension ldA, borrowed for the call */
int SP_d_struct_from_dense(CSPStructMat* S, int rows, int cols, const double* A, int ldA, const CSPOptions* opts);
/* reference :357 -- A(i,j) element callback */
int SP_d_struct_from_elements(CSPStructMat* S, int rows, int cols, double A(int i, int j), const CSPOptions* opts);
/* reference :408 -- C = op(S) B, trans in {'N','T','C'}, m columns */
int SP_d_struct_mult(const CSPStructMat S, char trans, int m, const double* B, int ldB, double* C, int ldC);
/* reference :474 */
int SP_d_struct_factor(CSPStructMat S);
/* reference :525 -- B <- S^{-1} B, in place */
int SP_d_struct_solve(const CSPStructMat S, int nrhs, double* B, int ldB);
/* reference :580 -- S <- S + s I (factor again afterwards) */
int SP_d_struct_shift(CSPStructMat S, double s);

/* ---- single precision and complex variants (reference :103-602): same conventions as SP_d_*; the handle types are
 * distinct per precision (a matrix made by SP_z_struct_from_dense must only go to SP_z_struct_* routines).  These
 * instantiations are carried by the double-precision MI355X engine: float is promoted to double, a complex matrix is
 * compressed through its interleaved real image (see csrc/host/HSSMatrixPromoted.hpp), so a complex caller's vectors are
 * used in place.  SP_?_struct_rank reports ranks in the caller's scalar type. */
#define SPX_DECLARE_C_API(P, CT)                                                                                        \
  void SP_##P##_struct_default_options(CSPOptions* opts);                                                               \
  void SP_##P##_struct_destroy(CSPStructMat* S);                                                                        \
  int SP_##P##_struct_rows(const CSPStructMat S);                                                                       \
  int SP_##P##_struct_cols(const CSPStructMat S);                                                                       \
  long long int SP_##P##_struct_memory(const CSPStructMat S);                                                           \
  long long int SP_##P##_struct_nonzeros(const CSPStructMat S);                                                         \
  int SP_##P##_struct_rank(const CSPStructMat S);                                                                       \
  int SP_##P##_struct_from_dense(CSPStructMat* S, int rows, int cols, const CT* A, int ldA, const CSPOptions* opts);    \
  int SP_##P##_struct_from_elements(CSPStructMat* S, int rows, int cols, CT A(int i, int j), const CSPOptions* opts);   \
  int SP_##P##_struct_mult(const CSPStructMat S, char trans, int m, const CT* B, int ldB, CT* C, int ldC);              \
  int SP_##P##_struct_factor(CSPStructMat S);                                                                           \
  int SP_##P##_struct_solve(const CSPStructMat S, int nrhs, CT* B, int ldB);                                            \
  int SP_##P##_struct_shift(CSPStructMat S, CT s);
SPX_DECLARE_C_API(s, float)
SPX_DECLARE_C_API(c, float _Complex)
SPX_DECLARE_C_API(z, double _Complex)
#undef SPX_DECLARE_C_API

/* ---- extensions (not in the reference): HSS knobs and device-resident operands ---------------- */
/* structured::construct_and_factor_from_dense (reference structured/StructuredMatrix.hpp:536-553, no C binding there):
 * compress and factor in one call -- for SP_TYPE_BLR the LU factorization is computed while the tiles are compressed
 * (BLRMatrix::compress_and_factor) and SP_d_struct_solve applies it; SP_TYPE_HSS: construct + factor. */
int SPX_d_struct_from_dense_and_factor(CSPStructMat* S, int rows, int cols, const double* A, int ldA, const CSPOptions* opts);
/* HSSOptions beyond CSPOptions (HSS/HSSOptions.hpp:465-490); call between default_options and from_* */
typedef struct SPXHSSOptions {
  int d0, dd, p;
  int compression_algorithm; /* 0 original, 1 stable */
  int random_engine;         /* 0 minstd_rand (reference default), 1 mt19937, 2 philox (device) */
  int random_distribution;   /* 0 normal, 1 uniform */
  /* sketching matrix (HSS/HSSOptions.hpp:110-133): 0 Gaussian, 1 SJLT (nnz entries +-1 per row); SJLT placement
   * 0 chunk / 1 perm; nonzeros per row in the first d0 + dd columns (nnz0) and in every further dd columns (nnz) */
  int compression_sketch, sjlt_algo, nnz0, nnz;
  /* the caller will factor: the ULV factorization of each tree level is enqueued on a second stream as soon as the
   * compression has settled that level; SP_d_struct_factor then only waits for it (0: factor when told to) */
  int factor_ahead;
  /* the operand is symmetric (0 no claim; 1 trusted; 2 checked on a sample of entries first): with the one random matrix of
   * both products A^T R = A R, so the second sketch GEMM is a copy of the first (device-resident and generated operands) */
  int symmetric_operand;
} SPXHSSOptions;
void SPX_d_struct_default_hss_options(SPXHSSOptions* h);
/* like SP_d_struct_from_dense, with explicit HSS options (h may be NULL) */
int SPX_d_struct_from_dense_hss(CSPStructMat* S, int rows, int cols, const double* A, int ldA,
                                const CSPOptions* opts, const SPXHSSOptions* h);
/* A is a DEVICE pointer (column-major, ldA); nothing is copied, A is borrowed for the call */
int SPX_d_struct_from_dense_device(CSPStructMat* S, int rows, int cols, const double* dA, long long ldA,
                                   const CSPOptions* opts, const SPXHSSOptions* h);
/* one process per GPU.  With 2^c ranks rank g owns the g-th subtree at depth c (its sketch columns,
 * compression, ULV factors and sweeps: no communication); the top of the tree is replicated after small
 * all-gathers of the cut nodes' reduced blocks.  allgather(user, dbuf, bytes_per_rank) must perform an
 * IN-PLACE all-gather of the DEVICE buffer dbuf of world * bytes_per_rank bytes (rank r contributes the
 * block at offset r * bytes_per_rank) -- RCCL over xGMI in production (strumpack_amd/dist.py), gloo in
 * the CPU tests.  The same hook is used by factor / solve / mult of the returned matrix, so it must
 * stay valid for the matrix' lifetime; every rank must call the same operations in the same order. */
typedef void (*SPXAllGatherFn)(void* user, void* dbuf, long long bytes_per_rank);
int SPX_d_struct_from_dense_device_sharded(CSPStructMat* S, int rows, int cols, const double* dA, long long ldA,
                                           const CSPOptions* opts, const SPXHSSOptions* h, int world, int rank,
                                           SPXAllGatherFn allgather, void* user);
/* ---- native process group: RCCL over xGMI, bound inside the library (no Python, no MPI needed) --------------------------
 * Rank 0 calls SPX_comm_unique_id and hands the 128 bytes to the other ranks by any means (MPI_Bcast, a file, a socket);
 * then every rank calls SPX_comm_create (collective; binds to the calling thread's current HIP device).  The engine issues
 * its collectives -- in-place all-gathers of the cut nodes' reduced blocks, sums of the replicated top nodes' coupling
 * blocks and, for a column-sharded operand, the reduction of the off-diagonal sample contributions -- on its own HIP
 * stream, in order with its kernels.  The communicator must outlive every matrix built with it; all ranks must call the
 * same operations in the same order. */
typedef void* SPXComm;
int SPX_comm_unique_id(char id[128]);
int SPX_comm_create(SPXComm* comm, int world, int rank, const char id[128]);
void SPX_comm_destroy(SPXComm* comm);
/* runs the engine's three collectives (in-place all-gather, all-reduce, reduce-scatter with per-rank counts) on small
 * device buffers and checks the results; collective; 0 = ok */
int SPX_comm_selftest(SPXComm comm);
int SPX_comm_size(const SPXComm comm);
int SPX_comm_rank(const SPXComm comm);
/* rows == columns [lo, hi) of an n x n matrix that `rank` of `world` owns (its subtree of the bisection tree for this leaf
 * size, structured/ClusterTree.hpp:104-114): the part of the operand that rank has to hold.  Returns non-zero if the tree
 * cannot be cut into one subtree per rank (world not a power of two, or n too small). */
int SPX_struct_shard_range(int n, const CSPOptions* opts, int world, int rank, int* lo, int* hi);
/* replicated operand (every rank passes the whole dA), native communicator */
int SPX_d_struct_from_dense_device_comm(CSPStructMat* S, int rows, int cols, const double* dA, long long ldA,
                                        const CSPOptions* opts, const SPXHSSOptions* h, SPXComm comm);
/* The matrix is one of the library's FORMULAS (kind: HSSK_GEN_TOEPLITZ = 1, 1/(1+|i-j|), the reference's test matrix;
 * HSSK_GEN_TOEPLITZ_UPPER = 2, its upper triangle): it is never stored -- the sketch kernel evaluates the tiles of the
 * operand it multiplies (the reference's blocked sampler over an element routine, structured/StructuredMatrix.cpp:214-262,
 * moved into the GEMM), scattered entries come from the formula.  Bit for bit the matrix SPX_d_struct_from_dense_device
 * builds from the stored operand.  _comm / _sharded: one process per GPU, no operand shards at all. */
int SPX_d_struct_from_generator(CSPStructMat* S, int n, int kind, const CSPOptions* opts, const SPXHSSOptions* h);
int SPX_d_struct_from_generator_comm(CSPStructMat* S, int n, int kind, const CSPOptions* opts, const SPXHSSOptions* h, SPXComm comm);
int SPX_d_struct_from_generator_sharded(CSPStructMat* S, int n, int kind, const CSPOptions* opts, const SPXHSSOptions* h,
                                        int world, int rank, SPXAllGatherFn allgather, void* user);
/* SHARDED operand: this rank passes only its blocks (DEVICE pointers, column-major): dArows = A(lo:hi, :) ((hi-lo) x cols,
 * ldr) and dAcols = A(:, lo:hi) (rows x (hi-lo), ldc).  dArows may be NULL -- a column-sharded operator: the contributions
 * A(:, cols_g) R(cols_g, :) of all ranks to Sr are then summed to the owners of the rows (reduce-scatter).  No rank ever
 * holds the full matrix. */
int SPX_d_struct_from_blocks_device(CSPStructMat* S, int rows, int cols, const double* dArows, long long ldr,
                                    const double* dAcols, long long ldc, const CSPOptions* opts, const SPXHSSOptions* h,
                                    SPXComm comm);
/* the same over the all-gather callback (gloo in the CPU tests; the reductions are emulated with all-gathers) */
int SPX_d_struct_from_blocks_device_cb(CSPStructMat* S, int rows, int cols, const double* dArows, long long ldr,
                                       const double* dAcols, long long ldc, const CSPOptions* opts, const SPXHSSOptions* h,
                                       int world, int rank, SPXAllGatherFn allgather, void* user);
/* HSS approximation of a kernel matrix K(i, j) = k(x_i, x_j) + lambda [i == j] over n points in R^d (points: d x n,
 * one point per column, HOST; reordered in place by the clustering, perm (n ints, may be NULL) receives the 1-based
 * permutation: new point i = old point perm[i]).  ktype 0 Gauss, 1 Laplace, 2 ANOVA (degree p); clustering 0 natural,
 * 1 2means, 2 kdtree, 3 pca, 4 cobble; neighbors = initial neighbour count (<= 0: default 64).  Reference:
 * HSSMatrix(kernel::Kernel&, opts) (HSS/HSSMatrix.cpp:88-106) -- no random sketch, the samples are kernel columns
 * chosen from nearest neighbours (HSS/HSSMatrix.compress_kernel.hpp).  The _sharded form is the one-process-per-GPU
 * variant (same conventions as SPX_d_struct_from_dense_device_sharded; every rank passes the same points). */
int SPX_d_struct_from_kernel(CSPStructMat* S, int n, int d, double* points, int ktype, double h, double lambda, int p,
                             const CSPOptions* opts, int clustering, int neighbors, int* perm);
int SPX_d_struct_from_kernel_sharded(CSPStructMat* S, int n, int d, double* points, int ktype, double h, double lambda, int p,
                                     const CSPOptions* opts, int clustering, int neighbors, int* perm, int world, int rank,
                                     SPXAllGatherFn allgather, void* user);
int SPX_d_struct_from_kernel_comm(CSPStructMat* S, int n, int d, double* points, int ktype, double h, double lambda, int p,
                                  const CSPOptions* opts, int clustering, int neighbors, int* perm, SPXComm comm);
int SPX_d_struct_mult_device(const CSPStructMat S, char trans, int m, const double* dB, long long ldB,
                             double* dC, long long ldC);
int SPX_d_struct_solve_device(const CSPStructMat S, int nrhs, double* dB, long long ldB);
int SPX_d_struct_levels(const CSPStructMat S);
int SPX_d_struct_is_compressed(const CSPStructMat S);
int SPX_d_struct_num_nodes(const CSPStructMat S);
/* pre-order node table, 6 ints per node: row_offset, rows, U_rows, U_rank, V_rank, is_leaf */
int SPX_d_struct_node_info(const CSPStructMat S, int* out);
/* phase timings (s) and the algorithmic flop model; out has 24 doubles:
 * [0] t_compress [1] t_sketch [2] t_random [3] t_tree [4] t_factor [5] t_solve [6] t_mult
 * [7] sketch_kernel_ms [8] sketch_launches [9] rounds [10] d_final
 * [11] f_sketch [12] f_local [13] f_reduce [14] f_id [15] f_ortho [16] f_ulv [17] f_solve
 * [18] factor_memory_bytes [19] sketch_kernel_flops (algorithmic flops of the launches timed in [7])
 * [20] sketch_kernel_bytes (SJLT sketch: algorithmic HBM bytes of those launches) */
int SPX_d_struct_stats(const CSPStructMat S, double* out);
/* ---- Schur complement of the (0,0) block of an HSS matrix: S = H11 - H10 H00^{-1} H01 -- what the reference's sparse
 * HSS fronts call on HSSMatrix<T> (HSS/HSSMatrix.hpp:330 partial_factor, :456 Schur_update, :459 Schur_product_direct,
 * :465 Schur_product_indirect; HSS/HSSMatrix.Schur.hpp; use: sparse/fronts/FrontHSS.cpp:391-407, :164, :218).
 * dims: [0] n0 [1] n1 [2] rV0 [3] mu0 [4] rV1 [5] rU0 [6] rU1;  Theta n1 x rV0, DUB01 mu0 x rV1, Phi n1 x mu0,
 * Vhat mu0 x rV0 (host outputs, any may be NULL):  S = H11 - Theta Vhat^T Phi^T.  The factors stay in HBM for the
 * products: Sr = S R, Sc = S^T R (direct);  Sr = Sr1 - H10 R0 - (H11 - S) R1, Sc = Sc1 - H01^T R0 - (H11 - S)^T R1
 * (indirect).  on_device: R / S* are device pointers. */
int SPX_d_struct_partial_factor(CSPStructMat S);
int SPX_d_struct_schur_dims(const CSPStructMat S, int* dims);
int SPX_d_struct_schur_update(CSPStructMat S, double* Theta, int ldT, double* DUB01, int ldD, double* Phi, int ldP,
                              double* Vhat, int ldV);
int SPX_d_struct_schur_product_direct(const CSPStructMat S, int c, const double* R, long long ldR, double* Sr,
                                      long long ldSr, double* Sc, long long ldSc, int on_device);
int SPX_d_struct_schur_product_indirect(const CSPStructMat S, int c, const double* R0, long long ldR0, const double* R1,
                                        long long ldR1, const double* Sr1, long long ldSr1, const double* Sc1,
                                        long long ldSc1, double* Sr, long long ldSr, double* Sc, long long ldSc,
                                        int on_device);
/* C = op(H_cc) B for the diagonal block of child c (HSSMatrix::child(c)->apply, HSS/HSSMatrix.hpp:194-202) */
int SPX_d_struct_mult_child(const CSPStructMat S, int child, char trans, int m, const double* B, long long ldB, double* C,
                            long long ldC, int on_device);
/* ---- sub-block extraction (HSS::HSSMatrix<T>::extract / extract_add, HSS/HSSMatrix.hpp:418-434, HSS/HSSMatrix.extract.hpp:36-104):
 * nb requests in one call, by tree traversal on the device.  rows / cols: the requests' 0-based index lists concatenated,
 * roff / coff: their prefix sums (nb + 1 entries); block b = H(rows[roff[b] .. roff[b+1]), cols[coff[b] .. coff[b+1])) goes to
 * out[b] (column-major, leading dimension ldo[b]); add != 0: is added to it.  on_device: out[b] are device pointers. */
int SPX_d_struct_extract_blocks(const CSPStructMat S, int nb, const int* rows, const int* roff, const int* cols, const int* coff,
                                double* const* out, const int* ldo, int add, int on_device);
/* ---- BLR frontal matrix: partial factorization of F = [F11 F12; F21 F22] -- what the reference's sparse BLR fronts call,
 * BLR::BLRMatrix<T>::construct_and_partial_factor(A11, A12, A21, A22, B11, B12, B21, tiles1, tiles2, admissible, opts)
 * (BLR/BLRMatrix.hpp:186-194, BLR/BLRMatrix.cpp:740-1037, algorithm RL = its default; batched GPU precedent
 * BLR/BLRMatrix.GPU.cpp:71-262; caller sparse/fronts/FrontBLR.cpp:329-432).  F11 is dsep x dsep with row / column clusters
 * tiles1 (ntiles1 sizes summing to dsep), F12 dsep x dupd and F21 dupd x dsep with clusters tiles2 on the update side.
 * Per block step: LU of the diagonal tile, truncated-RRQR compression of the block row of [F11 F12] and of the block column
 * of [F11; F21] (tolerances / max_rank from opts; F11 tiles per `admissible`, ntiles1 x ntiles1 column-major chars, NULL =
 * every off-diagonal tile; F12 / F21 tiles always), triangular solves on the factors, Schur update of everything that
 * trails -- including F22 <- F22 - F21 F11^{-1} F12, which stays dense.  Operands are column-major.
 *   _factor:        HOST operands; F22 (may be NULL: taken as zero) is overwritten with the Schur complement.
 *   _factor_device: DEVICE operands, borrowed for the call and left untouched; the Schur complement stays in HBM
 *                   (SPX_d_blr_front_schur_device / SPX_d_blr_front_schur).
 * Solve phases of a front (FrontBLR.cpp:525-570, BLRMatrix::trsmLNU_gemm / gemm_trsmUNN):
 *   _forward:  bsep <- L11^{-1} P bsep,  bupd <- bupd - B21 bsep;      _backward:  ysep <- U11^{-1} (ysep - B12 yupd).
 * _tile_ranks: (ntiles1 + ntiles2)^2 ints, column-major over the tiles of the whole front: rank of a U V^T tile, -1 dense.
 * _stats: 16 doubles: [0] seconds of the factorization, [1..4] device ms of LU / compression / triangular solves / Schur
 *   GEMMs (only with SPX_d_blr_front_time_phases(1) set before the call, else 0), [5] flops of the Schur GEMMs, [6] all
 *   flops, [7..9] stored scalars of B11 / B12 / B21, [10] largest tile rank, [11] launches of the Schur GEMM phase, [12] algorithmic bytes of the Schur GEMMs (operands once,
 *   the updated block read and written), [13..15] reserved. */
typedef void* SPXBLRFront;
int SPX_d_blr_front_factor(SPXBLRFront* F, int dsep, int dupd, const double* F11, int ld11, const double* F12, int ld12,
                           const double* F21, int ld21, double* F22, int ld22, int ntiles1, const int* tiles1, int ntiles2,
                           const int* tiles2, const char* admissible, const CSPOptions* opts);
int SPX_d_blr_front_factor_device(SPXBLRFront* F, int dsep, int dupd, const double* dF11, long long ld11, const double* dF12,
                                  long long ld12, const double* dF21, long long ld21, const double* dF22, long long ld22,
                                  int ntiles1, const int* tiles1, int ntiles2, const int* tiles2, const char* admissible,
                                  const CSPOptions* opts);
void SPX_d_blr_front_time_phases(int on);
/* tile compression of the BLR fronts made through this interface afterwards (the reference's BLROptions::set_low_rank_algorithm,
 * which its C options struct does not carry): 0 RRQR (truncated pivoted QR, the default), 1 ACA; returns non-zero for others */
int SPX_blr_low_rank_algorithm(int algo);
int SPX_d_blr_front_forward(const SPXBLRFront F, int nrhs, double* bsep, int ldb, double* bupd, int ldu);
int SPX_d_blr_front_backward(const SPXBLRFront F, int nrhs, double* ysep, int ldy, const double* yupd, int ldu);
int SPX_d_blr_front_schur(const SPXBLRFront F, double* F22, int ld22);
const double* SPX_d_blr_front_schur_device(const SPXBLRFront F, long long* ld);
int SPX_d_blr_front_tile_ranks(const SPXBLRFront F, int* out);
int SPX_d_blr_front_stats(const SPXBLRFront F, double* out);
void SPX_d_blr_front_destroy(SPXBLRFront* F);
/* the hssk kernel context of the matrix (include/hssk.h), for callers that share its stream */
void* SPX_d_struct_hssk_ctx(const CSPStructMat S);
/* Diagnostics (process-wide counters): compression rounds whose inner tree levels ran as ONE launch (kernels/hssk_tree.hip; the
 * default wherever the operand's entries can be read on the device; STRUMPACK_AMD_TREE_LAUNCH=0 switches it off), and how many of
 * those found a rank above the launch's speculated bound and redid the inner levels one by one. */
long long SPX_tree_pass_launches(void);
long long SPX_tree_pass_fallbacks(void);
/* The library keeps released device chunks in a process-wide cache for reuse (no hipMalloc / hipFree page-table work in solver
 * loops); the cache is invisible to the other allocators of the process (torch, RCCL).  Default cap: a fifth of the device's memory
 * (environment STRUMPACK_AMD_POOL_GB overrides).  _trim returns everything cached to the device now; _set_limit_gb changes the cap
 * (and trims down to it; 0 = cache nothing). */
long long SPX_device_pool_cached_bytes(void);
long long SPX_device_pool_limit_bytes(void);
void SPX_device_pool_trim(void);
void SPX_device_pool_set_limit_gb(double gb);

#ifdef __cplusplus
}
#endif
#endif /* STRUCTURED_MATRIX_H */
