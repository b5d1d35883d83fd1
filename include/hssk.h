/* hssk.h -- thin C-ABI over the hand-written gfx950 (CDNA4) HIP kernels of the HSS engine.
 *
 * This layer is NOT in the reference (pghysels/STRUMPACK has no GPU path for HSS,
 * doc/doxygen/pages/GPU_support.txt:4-6); it is the boundary "host C++ -> HIP" that SURVEY.md
 * section 8(b) defines.  Each entry point is the device equivalent of a dense routine the
 * reference's HSS code calls on the CPU; the reference call sites are cited per function.
 *
 * Conventions: every matrix pointer is a DEVICE pointer to column-major doubles; descriptor arrays
 * are HOST arrays (the library stages them to the device); one call == one variable-size batched
 * launch (one HSS tree level).  All functions return 0 on success, non-zero on error
 * (hssk_last_error() gives the message) and enqueue work on the context's stream without
 * synchronising unless stated.  Plain C types only.
 */
#ifndef HSSK_H
#define HSSK_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hssk_ctx hssk_ctx;

/* ---- context ------------------------------------------------------------------------------ */
int hssk_ctx_create(hssk_ctx** ctx, int device);
void hssk_ctx_destroy(hssk_ctx* ctx);
void* hssk_ctx_stream(hssk_ctx* ctx); /* hipStream_t */
int hssk_sync(hssk_ctx* ctx);
const char* hssk_last_error(void);
/* ---- sweep plans ------------------------------------------------------------------------------
 * Between hssk_plan_begin and hssk_plan_end every batched call on this context (from the calling thread) is executed
 * AND recorded: its kernel launches with their arguments, its descriptor arrays in storage owned by the plan.
 * hssk_plan_replay re-issues the launches on the context's stream: one launch per step, no descriptor building, no
 * staging -- for sweeps that are repeated on the same buffers (solves with a factored matrix, mat-vecs in a Krylov
 * loop).  Every pointer the recorded calls used must still be valid at replay.  Not for calls that synchronise or copy
 * to the host. */
typedef struct hssk_plan hssk_plan;
int hssk_plan_begin(hssk_ctx* ctx, hssk_plan** plan);
int hssk_plan_end(hssk_ctx* ctx);
int hssk_plan_replay(hssk_ctx* ctx, hssk_plan* plan);
void hssk_plan_destroy(hssk_plan* plan);
int hssk_plan_size(const hssk_plan* plan);

/* device memory helpers (hipMalloc/hipFree/hipMemcpy wrappers so FFI users need no HIP binding) */
void* hssk_malloc(long long bytes);
void hssk_free(void* dptr);
long long hssk_device_total_bytes(void); /* HBM of the calling thread's current device; 0 if unknown */
int hssk_memcpy_h2d(hssk_ctx* ctx, void* dst, const void* src, long long bytes);
int hssk_memcpy_d2h(hssk_ctx* ctx, void* dst, const void* src, long long bytes); /* synchronises */
/* host -> device without synchronising: src is copied into the context's pinned staging ring first, so the caller's
 * buffer may be released on return; ordered on the context's stream like a kernel launch */
int hssk_upload_async(hssk_ctx* ctx, void* dst, const void* src, long long bytes);
int hssk_memcpy_d2d(hssk_ctx* ctx, void* dst, const void* src, long long bytes); /* async */
/* strided copies of `height` columns of `width` bytes (pitches in bytes); both synchronise */
int hssk_memcpy2d_h2d(hssk_ctx* ctx, void* dst, long long dpitch, const void* src, long long spitch,
                      long long width, long long height);
int hssk_memcpy2d_d2h(hssk_ctx* ctx, void* dst, long long dpitch, const void* src, long long spitch,
                      long long width, long long height);
int hssk_memset_zero(hssk_ctx* ctx, void* dst, long long bytes); /* async */
/* Pipelined upload of a column-major block of doubles (rows x cols, host leading dimension lds, device leading dimension
 * ldd) on the context's COPY stream, so that it overlaps the kernels of the compute stream: pageable host memory goes
 * through a ring of pinned bounce buffers filled by host threads (the call returns when the last piece is queued), pinned
 * host memory is read by the DMA engines in place.  hssk_copy_fence: work enqueued on the compute stream afterwards waits
 * for the uploads issued so far; hssk_compute_fence: uploads issued afterwards wait for the compute work enqueued so far
 * (before a device buffer is overwritten).  Serves the streaming sampler of a host-resident operand
 * (structured/StructuredMatrix.cpp:214-262 never stores A either). */
int hssk_h2d_block_async(hssk_ctx* ctx, double* dst, long long ldd, const double* src, long long lds, long long rows,
                         long long cols);
/* The same for a block of `cols` columns of `width` bytes each (pitches in bytes): operands of the other scalar types, which
 * cross the link in their own format (hssk_expand_image). */
int hssk_h2d_bytes_async(hssk_ctx* ctx, void* dst, long long dpitch, const void* src, long long spitch, long long width,
                         long long cols);
int hssk_copy_fence(hssk_ctx* ctx);
int hssk_compute_fence(hssk_ctx* ctx);
/* The same dependency in two halves, for double-buffered streams: hssk_compute_mark(slot) remembers the compute work enqueued
 * so far under slot (0 / 1); hssk_copy_wait(slot) makes the uploads issued afterwards wait for exactly that work (and not
 * for compute work enqueued since) -- a no-op for a slot never marked. */
int hssk_compute_mark(hssk_ctx* ctx, int slot);
int hssk_copy_wait(hssk_ctx* ctx, int slot);
/* 1 if ptr is device memory of the current process (hipPointerGetAttributes) */
int hssk_is_device_pointer(const void* ptr);
/* Two contexts = two streams of the same device: the work enqueued on `waiter` from now on starts after everything enqueued so
 * far on `on` has finished (an event recorded on `on`'s stream, waited for by `waiter`'s).  Nothing blocks on the host.  This is how
 * a BLR block step factors its diagonal tile next to the compression of its block row and column. */
/* Batched adaptive cross approximation A (m x n, lda) ~ U V^T of dense tiles, one workgroup per tile
 * (adaptive_cross_approximation, dense/ACA.cpp:41-118: the tile compression of BLR matrices under --blr_low_rank_algorithm ACA).
 * U (m x rank, ldu) and V (n x rank, ldv) need room for min(m, n, max_rank) columns; row0 = the first row (the reference
 * draws it from a default-seeded std::mt19937); m, n <= 2048. */
typedef struct hssk_aca_desc {
  const double* A;
  int lda, m, n;
  double rtol, atol;
  int max_rank, row0;
  double* U;
  int ldu;
  double* V;
  int ldv;
  int* rank; /* out (device) */
} hssk_aca_desc;
int hssk_aca_vbatched(hssk_ctx* ctx, const hssk_aca_desc* descs, int count);
/* Side stream of a context: launches issued between hssk_side_begin and hssk_side_end go to a second stream that first
 * waits for everything issued on the main stream so far; hssk_side_join makes the main stream wait for them.  For work
 * that does not depend on what the main stream does meanwhile (the leaves' D x of a mat-vec next to the tree sweep).
 * Recorded plans (hssk_plan_*) replay the same routing. */
int hssk_side_begin(hssk_ctx* ctx);
int hssk_side_end(hssk_ctx* ctx);
int hssk_side_join(hssk_ctx* ctx);
int hssk_stream_wait(hssk_ctx* waiter, hssk_ctx* on);
/* Device-clock stopwatches on the compute stream: hssk_watch_start / _stop bracket the launches in between with HIP events
 * (any number of start / stop pairs per stopwatch); hssk_watch_read_ms synchronises, returns the summed duration of
 * stopwatch `id` (0 .. 7) in ms, writes the number of pairs to *pairs (may be NULL) and clears it.  This is how bench.py
 * times one kind of kernel inside a timed region without a host synchronisation per launch. */
int hssk_watch_start(hssk_ctx* ctx, int id);
int hssk_watch_stop(hssk_ctx* ctx, int id);
double hssk_watch_read_ms(hssk_ctx* ctx, int id, int* pairs);
/* duration (HIP events on the launch stream, ms) and algorithmic flops (2 m cols k) of the MAIN kernel launch of
 * the last hssk_dgemm on this context -- the launch whose grid fills whole rounds of the 512 workgroup slots;
 * the short tail / edge launches and the reduce pass are outside the bracket. */
float hssk_last_dgemm_ms(hssk_ctx* ctx);
double hssk_last_dgemm_flops(hssk_ctx* ctx);
/* the same without a synchronisation behind every product: _defer sets the last launch's bracket aside (hssk_last_dgemm_ms
 * then has nothing to report until the next product), _collect synchronises once and returns the summed duration (ms),
 * algorithmic flops and number of the brackets set aside since its last call */
int hssk_dgemm_timing_defer(hssk_ctx* ctx);
int hssk_dgemm_timing_collect(hssk_ctx* ctx, double* ms, double* flops, int* launches);
/* profiling aid: per-workgroup records of that main launch, 4 long long each {start, end (100 MHz ticks), hardware id
 * (XCC_ID << 32 | HW_ID), column tile}; returns the number of records copied (<= max_wgs).  Synchronises. */
long long hssk_last_dgemm_trace(hssk_ctx* ctx, long long* out, long long max_wgs);
/* effective shader clock (GHz) seen by workgroup 0 of the last hssk_dgemm main launch (s_memtime /
 * s_memrealtime); 0 if unavailable.  Synchronises. */
double hssk_last_dgemm_clock_ghz(hssk_ctx* ctx);

/* ---- generators --------------------------------------------------------------------------- */
/* Test matrices of test/test_HSS_seq.cpp:69-91 generated in HBM: kind 'T' Toeplitz
 * A(i,j) = i==j ? 1 : 1/(1+|i-j|), 'U' its upper triangle.  A is n x n, leading dimension lda. */
int hssk_fill_toeplitz(hssk_ctx* ctx, double* A, int n, long long lda, char kind);
/* the rows x cols block of that matrix whose first entry is global (i0, j0): one rank's shard of the operand */
int hssk_fill_toeplitz_block(hssk_ctx* ctx, double* A, int rows, int cols, long long lda, int i0, int j0, char kind);
/* N(0,1) samples, counter-based (Philox4x32-10 + Box-Muller): element (r,c) of the rows x cols
 * panel (leading dimension ld) is a pure function of (seed, (row0+r)*stride + c).  Device
 * replacement for DenseMatrix::random (dense/DenseMatrix.cpp:172-181) in performance runs. */
int hssk_randn(hssk_ctx* ctx, double* P, int rows, long long cols, long long ld, int row0,
               long long stride, unsigned long long seed);

/* ---- large GEMM: the sketch  S^T = R^T op(A)   (HSS/HSSExtra.hpp:236-239) ------------------- */
/* C(m x n) = alpha * A(m x k) * op(B) + beta * C;  op(B) = B (k x n) or B^T (B is n x k).
 * Tuned for m <= 512 (the sample count d), n and k huge (the matrix dimension). */
int hssk_dgemm(hssk_ctx* ctx, int transB, int m, long long n, long long k, double alpha,
               const double* A, long long lda, const double* B, long long ldb, double beta,
               double* C, long long ldc);

/* ---- operands given by a formula: the matrix is never stored ---------------------------------
 * (the reference's counterpart: the blocked sampler that evaluates tiles of an element routine on the fly,
 * structured/StructuredMatrix.cpp:214-262; here the tiles are evaluated INSIDE the sketch kernel, straight into the LDS
 * image the matrix cores read, so the sketch of a generated matrix moves no N x N bytes at all.)
 * kind HSSK_GEN_TOEPLITZ: G(i,j) = 1/(1+|i-j|) (test/test_HSS_seq.cpp:75-78); HSSK_GEN_TOEPLITZ_UPPER: its upper triangle
 * (:86-90) -- bit for bit the entries hssk_fill_toeplitz writes. */
#define HSSK_GEN_TOEPLITZ 1
#define HSSK_GEN_TOEPLITZ_UPPER 2
typedef struct hssk_gen {
  int kind, reserved;
  double p[4]; /* parameters of later kinds */
} hssk_gen;
/* hssk_dgemm with op(B)(kk, j) = transG ? G(j0 + j, kk) : G(kk, j0 + j), kk in [0, k), j in [0, n).  Shapes the eight-wave
 * kernel takes (m a multiple of its row block -- 64 / 128 / 192 --, k a multiple of 16, n >= 128, even lda: what the engine's
 * sketch passes) run the same tiles, K-split and summation order as hssk_dgemm on the stored matrix: bitwise the results of the
 * dense route.  Other shapes (ragged k, odd m, narrow outputs) are evaluated in blocks of at most 1024 columns that are
 * multiplied as stored operands, each with the K-split of its own width: equal to the dense route up to rounding only. */
int hssk_sketch_gen(hssk_ctx* ctx, const hssk_gen* g, int transG, int m, long long n, long long k, long long j0, double alpha,
                    const double* A, long long lda, double beta, double* C, long long ldc);
/* A(il, jl) = trans ? G(j0 + jl, i0 + il) : G(i0 + il, j0 + jl) for a rows x cols block (leading dimension lda) */
int hssk_gen_fill(hssk_ctx* ctx, const hssk_gen* g, double* A, long long rows, long long cols, long long lda, long long i0,
                  long long j0, int trans);

/* ---- variable-size batched GEMM on FP64 MFMA ------------------------------------------------ */
/* C_i(m x n) = alpha * op(A_i) * op(B_i) + beta * C_i  -- every small gemm() of
 * HSS/HSSMatrix.{compress,factor,solve,apply}.hpp (dense/DenseMatrix.cpp:935-1023). */
typedef struct hssk_gemm_desc {
  const double* A;
  const double* B;
  double* C;
  int m, n, k;
  int lda, ldb, ldc;
  int transA, transB; /* 0 = N, 1 = T */
  double alpha, beta;
} hssk_gemm_desc;
int hssk_gemm_vbatched(hssk_ctx* ctx, const hssk_gemm_desc* descs, int count);

/* Fused leaf sample update (compute_local_samples, leaf branch: HSS/HSSMatrix.compress.hpp:541-545 and
 * :591-594):  Sr -= R D^T  and  Sc -= R D  in one pass -- both products share the R panel (d x m,
 * transposed sample layout) and the m x m leaf block D.  Requires d even, d <= 192, ldr / lds even and
 * 16-byte aligned panels (the engine's layout); returns 2 (nothing done) otherwise. */
typedef struct hssk_leaf_update_desc {
  const double* R;  /* d x m */
  const double* D;  /* m x m */
  double* Sr;       /* d x m */
  double* Sc;       /* d x m */
  int d, m, ldr, ldd, lds;
} hssk_leaf_update_desc;
int hssk_leaf_update_vbatched(hssk_ctx* ctx, const hssk_leaf_update_desc* descs, int count);

/* ---- kernel matrices ---------------------------------------------------------- */
/* A kernel matrix K(i, j) = k(x_i, x_j) + lambda [i == j] over n points x_i in R^d stored one point per column
 * (d x n, column-major, device).  type 0: Gauss exp(-|x-y|_2^2 / (2 h^2)), 1: Laplace exp(-|x-y|_1 / h),
 * 2: ANOVA of degree p <= 8   (kernel/Kernel.hpp:333-399; eval :122-125). */
typedef struct hssk_kernel_spec {
  const double* X;
  long long n;
  int d, type, p;
  double h, lambda;
} hssk_kernel_spec;
/* out(a, b) = K(rows[a], cols[b]); rows = ri[0..nr) (device ints) or the range r0 + a when ri == NULL, same for
 * the columns  (Kernel::operator()(I, J, B), kernel/Kernel.hpp:138-147). */
typedef struct hssk_keval_desc {
  const int* ri;
  const int* ci;
  double* out;
  int nr, nc, ldo, r0, c0;
} hssk_keval_desc;
int hssk_kernel_eval_vbatched(hssk_ctx* ctx, const hssk_kernel_spec* spec, const hssk_keval_desc* descs, int count);
/* exact k nearest neighbours (Euclidean, the point itself excluded) of the points q0 <= i < q1 among all n points:
 * out_idx is k x n (device ints, neighbours of point i in column i, unordered, -1 where n - 1 < k; only the columns
 * of the query range are written -- one process per GPU searches for its own points).  Serves the neighbour lists of
 * HSSMatrix::compress_with_coordinates (HSS/HSSMatrix.compress_kernel.hpp:58-66).  d <= 64. */
int hssk_knn(hssk_ctx* ctx, const double* X, int d, int n, int k, int q0, int q1, int* out_idx);
/* Column sets of the kernel-matrix compression (the sorted, duplicate-free ids a node samples its rows on:
 * HSS/HSSMatrix.compress_kernel.hpp:108-131 for a leaf -- the neighbours of its points outside the leaf --, :159-183 for an
 * inner node -- the union of its children's sets without the ids inside the node).  out = sorted unique ids of
 * src0[0:n0] and src1[0:n1] (device ints; src1 may be NULL) that are >= 0 and outside [lo, hi); *count = their number
 * (out has room for n0 + n1).  universe = number of points (all ids are below it): a workgroup marks a bitmap of that many
 * bits in its LDS and reads it back in order; returns HSSK_UNSUPPORTED when the bitmap does not fit the LDS.
 * n0_dev / n1_dev (device ints, may be NULL): when given, the lengths of src0 / src1 are read from there -- the counts an earlier
 * launch wrote -- and n0 / n1 are only their upper bounds (out has room for n0 + n1): the levels of a tree are then launched
 * back to back without reading a count back in between. */
typedef struct hssk_colset_desc {
  const int* src0;
  const int* src1;
  int n0, n1, lo, hi;
  int* out;
  int* count;
  const int* n0_dev;
  const int* n1_dev;
} hssk_colset_desc;
int hssk_colsets(hssk_ctx* ctx, const hssk_colset_desc* descs, int count, int universe);
long long hssk_colsets_max_universe(void);   /* largest universe hssk_colsets takes on this device */
/* Binary-tree clustering of a point set by median splits, on the device (binary_tree_clustering, clustering/Clustering.hpp:143-168,
 * for the partitioners whose tree does not depend on the data: algo 4 = cobble, clustering/CobblePartitioning.cpp:36-78; algo 2 =
 * kd, clustering/KDTree.cpp:36-95).  X (d x n, a point per column) and perm (n ints, out, 0-based: new column i is old column
 * perm[i]) are DEVICE arrays, rearranged in place into cluster order; clusters of >= cluster_size points are halved (n / 2 | n - n / 2).
 * One launch per tree level.  *status (host, out): 0 = the arrangement is the reference's; non-zero = ties at a median / at the
 * farthest point or a long displacement chain were met -- X and perm are then NOT to be used and the caller takes the host form
 * (host/Clustering.hpp).  Returns 2 for an algorithm / dimension this form does not take.  Synchronises. */
int hssk_cluster_median(hssk_ctx* ctx, double* X, int d, int n, int algo, int cluster_size, int* perm, int* status);
/* pred[c] = sum_r w[r] k(x_r, t_c), c < m; T is d x m (device)   (Kernel::predict, kernel/KernelRegression.hpp:112-123) */
int hssk_kernel_predict(hssk_ctx* ctx, const hssk_kernel_spec* spec, const double* w, const double* T, int m, double* pred);

/* ---- gathers / scatters ----------------------------------------------------------------------- */
/* dst(:, j) = src(:, idx[j]) (idx == NULL: identity) -- DenseMatrix::extract_rows in the
 * transposed sample layout (dense/DenseMatrix.cpp:323-333), laswp (:287-297). */
typedef struct hssk_colgather_desc {
  const double* src;
  double* dst;
  const int* idx; /* device, ncols entries, or NULL */
  int rows, ncols, lds, ldd;
  int scatter; /* 1: dst(:, idx[j]) = src(:, j) */
} hssk_colgather_desc;
int hssk_gather_cols(hssk_ctx* ctx, const hssk_colgather_desc* descs, int count);
/* Real double-precision image of a column-major block (rows x cols scalars, leading dimension lds scalars) of another scalar
 * type, already on the device (the float / complex instantiations of the reference, HSS/HSSMatrix.cpp:513-516, are carried by
 * the double-precision engine):
 *   HSSK_DT_F32: dst(i, j) = (double) src(i, j)                                          -- dst is rows x cols;
 *   HSSK_DT_C32 / HSSK_DT_C64: dst(2i + a, 2j + b) = [re -im; im re](a, b) of src(i, j)     -- dst is 2 rows x 2 cols.
 * ldd in doubles.  Compute stream. */
enum { HSSK_DT_F64 = 0, HSSK_DT_F32 = 1, HSSK_DT_C32 = 2, HSSK_DT_C64 = 3 };
int hssk_expand_image(hssk_ctx* ctx, double* dst, long long ldd, const void* src, long long lds, long long rows, long long cols,
                      int dtype);
/* dst(i, :) = src(idx[i], :)  (scatter: dst(idx[i], :) = src(i, :)); accumulate: += */
typedef struct hssk_rowgather_desc {
  const double* src;
  double* dst;
  const int* idx; /* device, nrows entries, or NULL */
  int nrows, cols, lds, ldd;
  int scatter, accumulate;
} hssk_rowgather_desc;
int hssk_gather_rows(hssk_ctx* ctx, const hssk_rowgather_desc* descs, int count);
/* B(i,j) = A(I[i], J[j]) -- AFunctor::operator()(I,J,B) (HSS/HSSExtra.hpp:240-248).
 * I == NULL / J == NULL: contiguous range starting at i0 / j0.  transpose: B(j,i) = ... */
typedef struct hssk_elem_desc {
  const double* A;
  long long lda;
  const int* I; /* device */
  const int* J; /* device */
  int i0, j0;
  double* B;
  int m, n, ldb, transpose;
  /* ownership windows of a sharded operand (0, 0 = none): entries whose global row lies outside [rlo, rhi) or whose
   * global column lies outside [clo, chi) are written as 0 and A is not read there -- every process of a multi-GPU run
   * fills in the part of a coupling block it holds, the parts are summed over the ranks */
  int rlo, rhi, clo, chi;
} hssk_elem_desc;
int hssk_gather_elems(hssk_ctx* ctx, const hssk_elem_desc* descs, int count);
/* B(i,j) = G(I[i], J[j]): hssk_gather_elems with the matrix replaced by the formula (descs[].A / lda are ignored) */
int hssk_gen_elems(hssk_ctx* ctx, const hssk_gen* g, const hssk_elem_desc* descs, int count);
/* out[0:count) = sum over g < nslab of slabs[g * stride + (0:count)]  (partial blocks of several ranks after an all-gather) */
int hssk_sum_slabs(hssk_ctx* ctx, const double* slabs, long long count, long long stride, int nslab, double* out);
/* dst = src^T : rows x cols (src, lds) -> cols x rows (dst, ldd) */
typedef struct hssk_transpose_desc {
  const double* src;
  double* dst;
  int rows, cols, lds, ldd;
} hssk_transpose_desc;
int hssk_transpose(hssk_ctx* ctx, const hssk_transpose_desc* descs, int count);

/* dst = upper trapezoid of src (rows x cols), zeros below the diagonal: stacks the R factors of a TSQR tree */
typedef struct hssk_triu_desc {
  const double* src;
  double* dst;
  int rows, cols, lds, ldd;
  int dstride; /* row i of src goes to row i * dstride of dst (0 or 1: contiguous): interleaves stacked triangles */
} hssk_triu_desc;
int hssk_copy_triu(hssk_ctx* ctx, const hssk_triu_desc* descs, int count);

/* ---- sparse Johnson-Lindenstrauss sketch ---------------------------------------------------------------
 * The reference's --hss_compression_sketch SJLT (HSS/HSSMatrix.sketch.hpp; compress_stable.hpp:39-97): the sketching
 * matrix R (K x dn) has nnz <= 8 entries +-1 per row.  pat (DEVICE ints, NQ = (nnz <= 4 ? 4 : 8) per row):
 * pat[k * NQ + q] = column index of the q-th nonzero of row k, sign bit set for -1; unused entries = dn.
 * hssk_sjlt_dense: Rt (dn x K, ld) = R^T as a dense block (SJLTMatrix::SJLT_to_dense, sketch.hpp:573-596).
 * hssk_sjlt_sketch: St (dn x n_out, lds) = (op(A) R)^T with op(A) n_out x K: transA = 0: A(i, k) = A[i + k lda]
 * (matrix_times_SJLT, sketch.hpp:611-721), transA = 1: A(k, j) = A[k + j lda] (matrixT_times_SJLT, :723-809).
 * One pass over A (8 bytes per element), NQ LDS adds per element; dn <= 1024.  The launch is bracketed by the
 * events behind hssk_last_dgemm_ms / _flops (2 nnz flops per element). */
int hssk_sjlt_dense(hssk_ctx* ctx, double* Rt, int dn, long long K, long long ld, const int* pat, int nnz);
int hssk_sjlt_sketch(hssk_ctx* ctx, int transA, long long n_out, long long K, const double* A, long long lda,
                     const int* pat, int nnz, int dn, double* St, long long lds);

/* ---- column gather + small product ----------------------------------------------------------------- */
/* out(i, j) = G(i, g_j) + alpha sum_{k < K} M(i, m_k) C(j, k),  i < rows, j < J.
 * G(:, c) is column c of [G0 | G1] (G0 has gsplit columns; G0 == NULL: no gathered part), g_j = gidx ? gidx[j] : j;
 * M(:, c) likewise of [M0 | M1] with msplit, m_k = midx ? midx[k] : k; C(j, k) = C[j csj + k csk].  All operands
 * column-major with the rows contiguous; out must not overlap G or M.  Serves compute_local_samples / reduce_local_samples of
 * the inner levels (HSS/HSSMatrix.compress.hpp:524-629, 689-724) as one launch each. */
typedef struct hssk_combine_desc {
  const double *G0, *G1;
  int ldg, gsplit;
  const int* gidx; /* device */
  const double *M0, *M1;
  int ldm, msplit;
  const int* midx; /* device */
  const double* C;
  int csj, csk;
  double alpha;
  double* out;
  int ldo, rows, J, K;
} hssk_combine_desc;
int hssk_gather_combine(hssk_ctx* ctx, const hssk_combine_desc* descs, int count);

/* ---- first step of the ULV elimination of a node (HSS/HSSMatrix.factor.hpp:109-118) ------------------------- */
/* W1 (r x m, ldw) = (P^T D)(0:r, :) and W0t (m x (m - r), ldt) = (P^T D)(r:, :)^T - W1^T X, with P^T the row permutation
 * perm (device, m ints: row k of P^T D is row perm[k] of D) and X (r x (m - r), ldx) of the node's row ID.  m <= 256 (return
 * code 2 beyond: the caller composes the step from hssk_gather_elems and hssk_gemm_vbatched). */
typedef struct hssk_ulvsplit_desc {
  const double* D;
  int ldd, m, r;
  const int* perm;
  const double* X;
  int ldx;
  double* W1;
  int ldw;
  double* W0t;
  int ldt;
} hssk_ulvsplit_desc;
int hssk_ulv_split(hssk_ctx* ctx, const hssk_ulvsplit_desc* descs, int count);
/* The whole ULV step of an INNER node as one workgroup (kernels/hssk_ulv_node.hip; HSSMatrix.factor.hpp:65-137): children a, b
 * with U ranks ra, rb (m = ra + rb rows) and V ranks rva, rvb.
 *   D-hat (Dh, m x m, ld m: the diagonal blocks -- the children's Dt -- are in place): Dh(0:ra, ra:) = B01 Vt1b^T,
 *     Dh(ra:, 0:ra) = B10 Vt1a^T      (B01 ra x rvb, B10 rb x rva, Vt1a ra x rva, Vt1b rb x rvb, each ld = max(rows, 1))
 *   V-hat (Vh, m x rv, ld m) = [Vt1a Vd(0:rva, :); Vt1b Vd(rva:, :)]   (Vd (rva + rvb) x rv, ld rva + rvb; NULL: skipped)
 * eliminate != 0 (r < m): with the row ID (perm, X: r x (m - r), ld max(r, 1))
 *   W1 = (P^T Dh)(0:r, :) (ld max(r, 1)), Rlq (m x (m - r), ld m) = QR of W0^T = (P^T Dh)(r:, :)^T - W1^T X with tau (m doubles),
 *   Qt (m x m, ld m) the explicit Q~, Vt0T = Vh^T Q~(:, 0:q) (rv x q, ld rv), Vt1 = Q~(:, q:)^T Vh (r x rv, ld max(r, 1)),
 *   Dt = W1 Q~(:, q:) (r x r, ld ldt), WQ = W1 Q~(:, 0:q) (r x q, ld max(r, 1)),  q = m - r.
 * eliminate == 0: the assembly only (the root, factored by hssk_getrf_vbatched).  Returns 2 (nothing issued) unless
 * hssk_ulv_node_fits() holds for every node -- the caller then takes the batched steps. */
typedef struct hssk_ulvnode_desc {
  const double *B01, *B10, *Vt1a, *Vt1b, *Vd;
  double *Dh, *Vh;
  int ra, rb, rva, rvb;
  int eliminate;
  const int* perm;
  const double* X;
  double *W1, *Rlq, *Qt, *tau, *Vt0T, *Vt1, *Dt, *WQ;
  int m, r, rv, ldt;
} hssk_ulvnode_desc;
int hssk_ulv_node_vbatched(hssk_ctx* ctx, const hssk_ulvnode_desc* descs, int count);
/* 1 if a node of these dimensions fits the fused step (its blocks next to each other in the LDS) */
int hssk_ulv_node_fits(int m, int r, int rv, int ra, int rb, int rva, int rvb);
long long hssk_ulv_node_launches(void);   /* launches so far (process-wide; tests) */

/* ---- batched interpolative decomposition ------------------------------------------------------- */
/* Truncated column-pivoted Householder QR of W (d x m, column-major, overwritten), i.e. the row ID
 * of the m x d sample block:  DenseMatrix::ID_row -> ID_column_GEQP3 -> geqp3tol + trsm
 * (dense/DenseMatrix.cpp:746-790, dense/lapack/dgeqp3tol.f:203-232).  Stops at the first c with
 * |R_cc|/|R_00| <= rtol or |R_cc| <= atol; rank = min(c, max_rank).
 * Outputs: perm[0..m) (0-based: pivoted column k is original column perm[k]); *rank; and
 * X = R11^{-1} R12 (rank x (m-rank)) stored in W(0:rank, rank:m); the rest of W is left unspecified. */
typedef struct hssk_id_desc {
  double* W;
  int ldw, d, m;
  double rtol, atol;
  int max_rank;
  int* perm;    /* device, m ints */
  int* rank;    /* device, 1 int */
  double* work; /* device, 3*m doubles */
  const double* src; /* NULL: the panel is in W.  Otherwise the panel is read from src (d x m, leading dimension lds, left
                      * untouched) and W (ldw >= d) only receives the outputs -- saves the caller a copy of the samples */
  int lds;
  int defer_x; /* non-zero: the caller finishes with hssk_id_xsolve_vbatched once it has read the ranks (X then goes
                * straight to its final place, and the solve is off the path to the ranks).  W(0:rank, 0:m) holds
                * [R11 R12] on return, or already X behind R11 when hssk_id_solves_inline() says so for the batch. */
} hssk_id_desc;
int hssk_id_vbatched(hssk_ctx* ctx, const hssk_id_desc* descs, int count);
/* batches this process factored with several workgroups per panel (id_group_kernel: panels of 129..256 rows and up to 512
 * columns beyond the single-workgroup register kernels); tests assert that the path was taken */
long long hssk_id_group_launches(void);
/* 1: a batch with these largest dimensions is factored by kernels that always leave X in W (defer_x has no effect) */
int hssk_id_solves_inline(int dmax, int mmax);
/* X (rank x (m - rank), leading dimension ldx) = R11^{-1} R12 from the factored panel W (solved == 0), or a plain copy of
 * W(0:rank, rank:m) (solved != 0: the panel already holds X). */
typedef struct hssk_xsolve_desc {
  const double* W;
  int ldw, rank, m;
  double* X;
  int ldx, solved;
} hssk_xsolve_desc;
int hssk_id_xsolve_vbatched(hssk_ctx* ctx, const hssk_xsolve_desc* descs, int count);
/* The same ID for TALL panels W (d x m, d >> m), from the Gram matrix G = W^T W (m x m, both triangles; the caller forms it on the
 * matrix cores: hssk_gemm_vbatched over K chunks + hssk_sum_partials): a diagonally pivoted Cholesky factorization of G is the
 * column-pivoted QR of W -- same pivots, same |R_kk|, same stopping rule (dgeqp3tol.f:225-232) -- in `rank` steps of O(k m).
 * G resolves singular values down to ~1e-8 of the largest: for tolerances >= 1e-6 only (the caller's choice).
 * Outputs: perm[0..m) as hssk_id_desc (skeleton columns in pivot order, then the rest in index order); *rank, or -1 when the
 * rank reached min(ldr, hssk_pchol_id_rank_cap(m)) rows without meeting the tolerance (the caller then takes the QR path);
 * R(0:rank, 0:m) = [R11 R12] in pivoted column order (leading dimension ldr), for hssk_id_xsolve_vbatched.  m <= 256. */
typedef struct hssk_pchol_desc {
  const double* G;
  int ldg, m;
  double rtol, atol;
  int max_rank;
  int* perm; /* device, m ints */
  int* rank; /* device, 1 int */
  double* R;
  int ldr;
} hssk_pchol_desc;
int hssk_pchol_id_vbatched(hssk_ctx* ctx, const hssk_pchol_desc* descs, int count);
int hssk_pchol_id_max_m(void);
int hssk_pchol_id_rank_cap(int m); /* rows of R the kernel can hold for a panel of m columns */
/* out[0..n) = P[0..n) + P[stride .. stride + n) + ... (count terms, added in this order): the K-split partial products of a
 * long inner dimension, summed deterministically */
typedef struct hssk_sum_desc {
  const double* P;
  long long stride, n;
  int count;
  double* out;
} hssk_sum_desc;
int hssk_sum_partials(hssk_ctx* ctx, const hssk_sum_desc* descs, int count);
/* G (m x m, leading dimension ldg, both triangles written) = W^T W for tall panels W (rows x m, leading dimension ldw, rows >= 2):
 * 128 x 128 blocks on and above the diagonal on the FP64 matrix cores, mirrored below it.  The K-split of a long panel is the
 * caller's: one descriptor per row chunk into its own G, then hssk_sum_partials. */
typedef struct hssk_gram_desc {
  const double* W;
  int ldw, rows, m;
  double* G;
  int ldg;
} hssk_gram_desc;
int hssk_gram_vbatched(hssk_ctx* ctx, const hssk_gram_desc* descs, int count);
/* The same for panels that are blocks of a kernel matrix, W(k, c) = K(x_row(k), x_col(c)) with rows = ri[0..rows) (device ints) or the
 * range r0 + k, columns = ci[0..m) or c0 + c (as hssk_keval_desc, no diagonal shift: the two point sets are disjoint): the
 * entries are evaluated while they are staged, the panel itself never exists.  Gauss / Laplace kernels, point dimension <= 16,
 * m <= 256 (hssk_gram_gen_supported); returns 2 otherwise. */
typedef struct hssk_gramgen_desc {
  const int* ri;
  int r0;
  const int* ci;
  int c0;
  int rows, m;
  double* G;
  int ldg;
} hssk_gramgen_desc;
int hssk_gram_gen_vbatched(hssk_ctx* ctx, const hssk_kernel_spec* spec, const hssk_gramgen_desc* descs, int count);
int hssk_gram_gen_supported(const hssk_kernel_spec* spec, int mmax);

/* ---- the inner levels of a compression round as ONE launch (kernels/hssk_tree.hip) ------------------------------------
 * compress_recursive_stable above the leaves (HSS/HSSMatrix.compress_stable.hpp:165-348, HSS/HSSMatrix.compress.hpp:555-629,
 * 689-724): coupling blocks, local samples, both interpolative decompositions, reduced samples and skeleton indices of every
 * inner node, one workgroup per (node, basis), children before parents, ranks never leaving the device.
 * nodes: DEVICE array, one record per node of the (sub)tree, leaves included.  Leaves (and any node already compressed) are
 * inputs: S / perm / Rred / I / r as the level-synchronous calls left them, flag = {1, 1}, status = 0.  Inner nodes are
 * outputs: c0 / c1 / lvl and the storage pointers set, r = m = flag = status = 0; storage sized for ranks <= rcap:
 *   S[s]: d x 2 rcap (leading dimension lds), Rred[s]: d x rcap (lds), perm[s]: 2 rcap ints, I[s]: rcap ints,
 *   X[s]: rcap x 2 rcap doubles (written r x (m - r) with leading dimension r), B01 / B10: rcap^2 doubles (written with the
 *   children's ranks as leading dimensions, as hssk_gather_elems would), W[s]: (2 rcap)^2 doubles of workspace.
 * s = 0: the row basis U (from the row samples Srt), s = 1: the column basis V.  Rred[0] = V^T Rr (rV columns), Rred[1] =
 * U^T Rc (rU columns).  lvl = depth of the node (the tolerances of its decompositions are rtol / lvl, atol / lvl); a node
 * with lvl == 0 is the root: coupling blocks only.
 * order (HOST, count entries): (node << 1) | s in dispatch order -- every entry after both sides of both its children; the
 * root once, with s = 0.  res (DEVICE, 4 ints per node; written for every entry of `order`): rU, rV, status of side 0, status of
 * side 1 (1: a rank above rcap or more rows than samples somewhere below -- nothing of the node is valid; the caller takes the
 * level-synchronous calls).
 * rcap in {32, 48, 64}, d <= 256; returns 2 otherwise.  hssk_sweep_status() reports a workgroup that gave up waiting. */
typedef struct hssk_tnode {
  int c0, c1, lvl, reserved;
  double* S[2];
  int* perm[2];
  double* Rred[2];
  int* I[2];
  double* X[2];
  double *B01, *B10;
  double* W[2];
  int r[2], m[2], flag[2];
  int status, pad;
} hssk_tnode;
typedef struct hssk_elem_src {   /* where scattered entries of the operand come from: A(i, j) = A[i + j lda], or the formula */
  const double* A;
  long long lda;
  hssk_gen gen;
  int use_gen;
} hssk_elem_src;
int hssk_tree_inner(hssk_ctx* ctx, hssk_tnode* nodes, const int* order, int count, int d, int lds, int rcap, double rtol,
                    double atol, int max_rank, const hssk_elem_src* src, int* res);
int hssk_tree_rcap_max(void);

/* ---- batched Householder QR ------------------------------------------------------------------- */
/* A (rows x cols, overwritten) = Q R.  nq > 0: the first nq columns of Q are written to Q
 * (rows x nq, ldq).  rdiag (device, 2 doubles, may be NULL) receives max|R_ii|, min|R_ii| over
 * i < min(rows, cols).  Serves DenseMatrix::orthogonalize (geqrf+orgqr, dense/DenseMatrix.cpp:721-744)
 * and DenseMatrix::LQ (gelqf+orglq of W0 == QR of W0^T, :693-719). */
typedef struct hssk_qr_desc {
  double* A;
  int lda, rows, cols;
  double* Q;
  int ldq, nq;
  double* rdiag;
  double* work; /* device, rows + cols doubles */
  /* stair > 0 (needs nq == 0, more than 256 rows): column j of A is zero at and below row stair * (j + 1) -- a stack
   * of `stair` upper-triangular factors with their rows interleaved (TSQR tree).  The blocked factorisation then only
   * touches the rows [j0, stair * (j0 + panel)) of each panel step; R is the same as for the dense sweep. */
  int stair;
  /* early exit for callers that only want the R-diagonal TEST of the stable stopping criterion (rdiag set, nq == 0): the
   * factorisation stops at the first step k with |R_kk| < stop_abs or |R_kk| < stop_rel * max_{i<=k} |R_ii|, and rdiag then
   * holds that prefix maximum and |R_kk|.  Since the maximum over all of the diagonal can only be larger, "min / max below the
   * tolerance" is decided exactly as by the full factorisation (which runs when no step qualifies); A and work are left
   * incomplete after an early exit.  0 / 0: never.  Honoured by the register kernels, ignored (full sweep) elsewhere. */
  double stop_rel, stop_abs;
  /* r_only != 0 (nq == 0): only R is wanted -- the register kernels then write back the upper triangle alone (the reflectors
   * and the taus are dropped: half the stores of a TSQR chunk); ignored elsewhere. */
  int r_only;
} hssk_qr_desc;
int hssk_qr_vbatched(hssk_ctx* ctx, const hssk_qr_desc* descs, int count);
/* R1 (m x m, upper triangle, updated in place) <- R of the QR factorisation of [triu(R1); triu(R2)]: the merge step of a
 * TSQR tree.  Only the upper triangles are read (whatever else the arrays hold -- the reflectors of an earlier factorisation --
 * is ignored) and only the upper triangle of R1 is written.  m <= 224 (return code 2 beyond). */
typedef struct hssk_tpqr_desc {
  double* R1;
  int ld1;
  const double* R2;
  int ld2, m;
} hssk_tpqr_desc;
int hssk_tpqr_vbatched(hssk_ctx* ctx, const hssk_tpqr_desc* descs, int count);
/* Q(:, 0:nq) only, from panels factored by an earlier hssk_qr_vbatched call with the same A (reflectors + R)
 * and work (taus); rdiag is not touched.  Lets the rank-adequacy test form Q only for the nodes whose
 * R-diagonal test did not already settle (compress_stable.hpp:405-417). */
int hssk_formq_vbatched(hssk_ctx* ctx, const hssk_qr_desc* descs, int count);

/* ---- batched triangular solve / LU ------------------------------------------------------------- */
/* B <- op(T)^{-1} B, T (n x n) triangular, B (n x nrhs)  (trsm Side::L, dense/DenseMatrix.cpp:1059-1085) */
typedef struct hssk_trsm_desc {
  const double* T;
  double* B;
  int n, nrhs, ldt, ldb;
  int lower, transT, unit;
  const double* Tinv;   /* optional (may be NULL): the inverted 64 x 64 diagonal blocks of T as hssk_trtri_diag_vbatched leaves them
                         * (mode 2 for a unit lower T, mode 1 for an upper T), for callers that solve with the same triangle
                         * more than once -- the blocked forms (n >= 128) then skip the inversion */
} hssk_trsm_desc;
int hssk_trsm_vbatched(hssk_ctx* ctx, const hssk_trsm_desc* descs, int count);
/* In-place LU with partial pivoting (DenseMatrix::LU, getrf, dense/DenseMatrix.cpp:564-589);
 * piv: device, n ints, 0-based row interchanges; info: device int (0 ok, >0 zero pivot). */
typedef struct hssk_lu_desc {
  double* A;
  int n, lda;
  int* piv;
  int* info;
} hssk_lu_desc;
int hssk_getrf_vbatched(hssk_ctx* ctx, const hssk_lu_desc* descs, int count);
/* B <- A^{-1} B with the factors above (DenseMatrix::solve, getrs, :624-640) */
typedef struct hssk_lusolve_desc {
  const double* LU;
  const int* piv;
  double* B;
  int n, nrhs, lda, ldb;
} hssk_lusolve_desc;
int hssk_getrs_vbatched(hssk_ctx* ctx, const hssk_lusolve_desc* descs, int count);
/* B <- P B only: the row interchanges of getrf (0-based piv, n of them) applied to B (n x nrhs) in order  (DenseMatrix::laswp,
 * dense/DenseMatrix.cpp:287-297); LU is not read */
int hssk_laswp_vbatched(hssk_ctx* ctx, const hssk_lusolve_desc* descs, int count);

/* ---- block substitution with the factors of a BLR front, one right-hand side, ONE launch (kernels/hssk_blr_sweep.hip) --------
 * BLRMatrix::solve (BLR/BLRMatrix.hpp:118-122), FrontBLR::fwd_solve_phase2 / bwd_solve_phase1 (sparse/fronts/FrontBLR.cpp:525-570).
 * rows[w] (HOST): block row w of the sweep in dispatch order; it owns X[off, off + m), subtracts its terms
 * acc -= U (V^T X[src_off, src_off + n)) and -- LU != NULL -- solves with its diagonal tile: mode 0: x = L^{-1} P acc (piv: the
 * tile's 0-based interchanges; Tinv: hssk_trtri_diag_vbatched mode 2), mode 1: x = U^{-1} acc (Tinv: mode 1).  A term's source is
 * either an input (src_flag < 0) or the piece of block row src_flag < w of the same launch, waited for.  X: device, one column;
 * flags: device, nrows ints (cleared by the call).  Tiles of up to 512 rows; returns 2 beyond.  hssk_sweep_status() reports a
 * workgroup that gave up waiting. */
typedef struct hssk_blr_term {
  const double* U;   /* m x r, leading dimension m (the row's) */
  const double* V;   /* n x r, leading dimension n */
  int r, n, src_off, src_flag;
} hssk_blr_term;
typedef struct hssk_blr_row {
  int first_term, nterms, off, m;
  const double* LU;
  int lda, mode;
  const int* piv;
  const double* Tinv;
} hssk_blr_row;
int hssk_blr_sweep(hssk_ctx* ctx, const hssk_blr_row* rows, int nrows, const hssk_blr_term* terms, int nterms, double* X, int* flags);
/* The same with the two tables resident on the device (the factors do not change between solves: the caller uploads them once).
 * hssk_blr_sweep_check: the checks of hssk_blr_sweep on host tables, nothing issued (0 fine, 2 beyond the kernel, 1 malformed);
 * hssk_blr_sweep_resident: d_rows / d_terms = device copies of tables that passed it. */
int hssk_blr_sweep_check(const hssk_blr_row* rows, int nrows, const hssk_blr_term* terms, int nterms);
int hssk_blr_sweep_resident(hssk_ctx* ctx, const hssk_blr_row* d_rows, int nrows, const hssk_blr_term* d_terms, double* X, int* flags);

/* ---- single-launch tree sweeps (few right-hand sides) ---------------------------------------------------------
 * The forward / backward ULV sweeps (HSS/HSSMatrix.solve.hpp:69-238) and the mat-vec up / down sweeps
 * (HSS/HSSMatrix.apply.hpp:55-220) of a whole (sub)tree as ONE launch each: one workgroup per node, ordered so that a
 * node only depends on lower-indexed descriptors (children before parents going up, parents before children going
 * down); wait* name those descriptors (-1: none / produced by an earlier launch) and are checked for that order.  The
 * vectors handed from node to node (ft1 / z / x going up and down the solve, tmp1 / tmp2 of the mat-vec) live in buffers
 * that hssk_sweep_arm fills with a sentinel beforehand; a consumer polls the words it needs until they are written.  See
 * kernels/hssk_sweep.hip.  Right-hand sides are processed in groups of four (blockIdx.y); node dimensions <= 256,
 * otherwise the calls return 2 and do nothing (the caller issues the batched calls instead).
 * hssk_sweep_status: non-zero if a workgroup of an earlier sweep gave up waiting (checked after a synchronisation). */
int hssk_sweep_arm(hssk_ctx* ctx, double* handoff, long long count);
typedef struct hssk_sweep_fwd_desc {
  const double* fsrc;   /* m x nrhs right-hand side rows of the node (leaf: rows of b; inner: [ft1_0; ft1_1]) */
  const double *B01, *B10, *zc; /* inner nodes: f(0:rU0) -= B01 zc(rV0:), f(rU0:) -= B10 zc(0:rV0); leaves: NULL */
  const int* permU;
  const double* XU;     /* r x (m - r) */
  const double* Rlq;    /* m x (m - r): R~ = L^T in the upper triangle */
  const double* Tinv;   /* ceil((m-r)/64) blocks of 64 x 64: inverses of the diagonal blocks of R~^T (hssk_trtri_diag_vbatched) */
  const double* WQ;     /* r x (m - r): W1 Q~(:, 0:m-r) */
  const double* Vt0T;   /* rv x (m - r): Vt0^T, Vt0 = Q~(:, 0:m-r)^T Vhat */
  const int* permV;     /* inner nodes only */
  const double* XV;
  double *ft1, *y, *z;  /* out: r x nrhs (ld ldp), (m-r) x nrhs (ld m-r), rv x nrhs (ld ldz) */
  /* the root of the factorization (LU != NULL): x = LU^{-1} f with f assembled as above (m rows), written to xroot
   * (ld ldxr); TinvL / TinvU = inverted diagonal blocks of L and U (hssk_trtri_diag_vbatched modes 2 / 1) */
  const double *LU, *TinvL, *TinvU;
  const int* piv;
  double* xroot;
  int ldf, rU0, rU1, rV0, rV1, ldz_in, m, r, mv, rv, ldp, ldz, ldxr;
  int wait0, wait1;
  /* chain block of an inner node (optional; hssk_sweep_chain_ok says which shapes the kernels take): the node's whole step as
   * ONE matrix, [ft1; z] = G [f; zc], G (r + rv) x (m + mv), leading dimension ldg -- what the parent waits for is then one
   * pass behind the children's vectors instead of five dependent ones; y (not on the chain) follows through the blocks above.
   * The caller obtains G by running this very sweep on the columns of an identity (DeviceHSS::chain_blocks). */
  const double* G;
  int ldg;
} hssk_sweep_fwd_desc;
/* 1 if the vector forms of the forward sweep (nrhs <= 4) use a chain block of this shape */
int hssk_sweep_chain_ok(int m, int r, int mv, int rv);
/* forward sweeps launched in the chain-block form so far (process-wide; tests) */
long long hssk_sweep_chain_launches(void);
int hssk_ulv_fwd_sweep(hssk_ctx* ctx, const hssk_sweep_fwd_desc* descs, int count, int nrhs);
/* out (m x nrhs, ldo) = Qt(:, 0:m-r) y + Qt(:, m-r:) xpart */
typedef struct hssk_sweep_bwd_desc {
  const double *Qt, *y, *xpart;
  double* out;
  int m, r, ldx, ldo;
  int wait0;
} hssk_sweep_bwd_desc;
int hssk_ulv_bwd_sweep(hssk_ctx* ctx, const hssk_sweep_bwd_desc* descs, int count, int nrhs);
/* up: dst (r x nrhs, ldd) = src(perm[0:r], :) + X src(perm[r:], :)   (HSSBasisID::applyC; X is r x (m - r)) */
typedef struct hssk_apply_up_desc {
  const double* src;
  const int* perm;
  const double* X;
  double* dst;
  int m, r, lds, ldd;
  int inner; /* src holds the children's results (handed over inside the launch or by an earlier one); 0: rows of x */
  int wait0, wait1;
} hssk_apply_up_desc;
/* down: leaf (D != NULL): out = op(D) x + beta out + U tmp2;  inner: out = [B01 t1_1; B10 t1_0] (trans: [B10^T t1_1;
 * B01^T t1_0]) + U tmp2, with U tmp2 = scatter through perm of [tmp2; X^T tmp2] (HSSBasisID::apply; X is ro x (mo - ro));
 * tmp2 == NULL for the root of the sweep.  wait0 = the parent's down descriptor, wait1 / wait2 = the children's up
 * descriptors (inner nodes); indices count the up descriptors first, then the down descriptors. */
typedef struct hssk_apply_down_desc {
  const double* tmp2;
  const int* perm;
  const double* X;
  const double *D, *x;
  const double *B01, *B10, *t1;
  double* out;
  double beta;
  int ld2, mo, ro, m, ldx, trans, ri_a, ri_b, ro_a, ro_b, ldt1, ldo;
  int wait0, wait1, wait2;
  int acc; /* matrix-core form only: a leaf whose op(D) x + beta out is already in `out` (D == NULL, no t1): out += U tmp2 */
} hssk_apply_down_desc;
int hssk_apply_sweep(hssk_ctx* ctx, const hssk_apply_up_desc* ups, int nup, const hssk_apply_down_desc* downs, int ndown,
                     int nrhs);
int hssk_sweep_status(hssk_ctx* ctx);
/* While set, the three sweeps return 2 (nothing issued) for operands their matrix-core form (kernels/hssk_sweep_mma.h: at
 * least hssk_sweep_mma_min_nrhs() right-hand sides, node vectors that fit the LDS) does not take, instead of running the
 * vector form: callers with a better alternative for that case (batched launches per level) ask first. */
int hssk_sweep_require_mma(hssk_ctx* ctx, int on);
int hssk_sweep_mma_min_nrhs(void);
/* number of sweeps this process issued in the many-right-hand-side matrix-core form (kernels/hssk_sweep_mma.h) */
long long hssk_sweep_mma_launches(void);
/* Tinv (ceil(n/64) blocks of 64 x 64, leading dimension 64) = transposed inverses of the 64 x 64 diagonal blocks of the
 * triangular R (n x n, ldr), zero padded; see `mode`. */
typedef struct hssk_trtri_desc {
  const double* R;
  double* Tinv;
  int n, ldr;
  int mode; /* 0: R upper, blocks hold (R(b,b)^{-1})^T;  1: R upper, plain inverses;  2: unit lower triangle of R, plain inverses */
} hssk_trtri_desc;
int hssk_trtri_diag_vbatched(hssk_ctx* ctx, const hssk_trtri_desc* descs, int count);

/* ---- sub-block extraction by tree traversal (HSSMatrix::extract / extract_add, HSS/HSSMatrix.extract.hpp:36-104) -------------
 * nodes: DEVICE array, the matrix's nodes in pre-order (the sub-tree of `root` is what is extracted from; row / column
 * indices are global: lo of a node is its first row).  iperm* are the inverses of the interpolative bases' permutations
 * (row q of U = P [I; E] is row iperm[q] of [I; E]).  rows / cols: all requested indices, concatenated over the requests
 * (device); block b takes rows [ri0, ri0 + ni) and cols [cj0, cj0 + nj) of those lists and writes (accumulate: adds to) its
 * ni x nj result at out (device, leading dimension ldo).  pair_off: prefix sums of ni * nj (nblocks + 1 entries, device).
 * work: (nrows + ncols) * maxdepth * rmax doubles (device); rmax >= every rank, maxdepth > the depth of every leaf. */
typedef struct hssk_tree_node {
  int lo, m, c0, c1;
  int rU, rV, mU, mV;
  const double *XU, *XV;      /* r x (m - r), leading dimension r */
  const int *ipermU, *ipermV; /* mU / mV ints */
  const double *B01, *B10, *D;
} hssk_tree_node;
typedef struct hssk_extract_block {
  int ri0, ni, cj0, nj;
  double* out;
  int ldo;
} hssk_extract_block;
int hssk_hss_extract(hssk_ctx* ctx, const hssk_tree_node* nodes, int root, int rmax, int maxdepth, const int* rows, int nrows,
                     const int* cols, int ncols, const hssk_extract_block* blocks, const long long* pair_off, int nblocks,
                     long long npairs, int accumulate, double* work);

/* ---- small utilities --------------------------------------------------------------------------- */
/* out[j] = sum_i P(i,j)^2 over the rows x cols panel: Frobenius norms for the stopping test
 * (HSS/HSSMatrix.compress_stable.hpp:418,438) */
typedef struct hssk_norm_desc {
  const double* P;
  int rows, cols, ld;
  double* out; /* device, 1 double: sum of squares */
} hssk_norm_desc;
int hssk_sumsq_vbatched(hssk_ctx* ctx, const hssk_norm_desc* descs, int count);
/* out(perm[k], j) = k < r ? (k == j) : X(j, k - r): dense form of the interpolative basis P [I; E]
 * with E = X^T (HSSBasisID::dense, HSS/HSSBasisID.hpp:146-153); out is m x r */
typedef struct hssk_basis_desc {
  const double* X; /* r x (m - r), ldx */
  const int* perm; /* device, m ints */
  double* out;
  int m, r, ldx, ldo;
} hssk_basis_desc;
int hssk_basis_dense(hssk_ctx* ctx, const hssk_basis_desc* descs, int count);
/* A(i,i) += sigma for i < n  (DenseMatrix::shift; HSS/HSSMatrix.cpp:359-365) */
typedef struct hssk_shift_desc {
  double* A;
  int n, lda;
} hssk_shift_desc;
int hssk_shift_diag(hssk_ctx* ctx, const hssk_shift_desc* descs, int count, double sigma);
/* the same for the real 2n x 2n image [re -im; im re] (entries interleaved) of a complex matrix: adds the image of
 * (re + i im) I, i.e. A(k,k) += re, A(2j, 2j+1) -= im, A(2j+1, 2j) += im; n (the real dimension) must be even */
int hssk_shift_diag_cplx(hssk_ctx* ctx, const hssk_shift_desc* descs, int count, double re, double im);
/* peak-rate probe: runs a dependent-free v_mfma_f64_16x16x4_f64 loop on every CU and returns the
 * measured TFLOP/s (used by bench.py to confirm the FP64 matrix roof on the box) */
double hssk_mfma_f64_peak_tflops(hssk_ctx* ctx, int iters);
/* detailed probe: waves_per_simd in {1,2,4,8}, zero_data != 0 feeds zeros (DVFS check).
 * out[0] TFLOP/s, out[1] shader cycles per MFMA per wave, out[2] effective shader clock in GHz */
int hssk_mfma_f64_probe(hssk_ctx* ctx, int iters, int waves_per_simd, int zero_data, double* out);

#ifdef __cplusplus
}
#endif
#endif /* HSSK_H */
