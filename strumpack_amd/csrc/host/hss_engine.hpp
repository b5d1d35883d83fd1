// DeviceHSS: the MI355X-resident HSS matrix and the level-synchronous orchestration of the hot
// path -- randomized compression, hierarchical apply, ULV factorization and solve -- over the
// hand-written HIP kernels of include/hssk.h (one variable-size batched launch per step and tree
// height).  strumpack::HSS::HSSMatrix<double> (HSSMatrix.hpp) is the reference-shaped facade on top.
//
// Data layout in HBM
//  * the tree is flattened on the host (pre-order node table, lists by height / depth); every node
//    owns device blocks carved from bump arenas: D (leaf, m x m), B01 / B10 (coupling), the
//    interpolative bases as X = R11^{-1} R12 (rank x (rows-rank), E = X^T) plus a 0-based row
//    permutation, and the ULV factors (Q~ = Q^T, R~ = L^T, W1, Vt0; root LU);
//  * the random samples are kept TRANSPOSED: Rt, Srt, Sct are (dcap x N) column-major, so the
//    samples of node [lo, lo+m) form one contiguous d x m panel, the operand layout that the sketch
//    GEMM, the per-node GEMMs and the column-pivoted QR of the ID all read contiguously.
#pragma once
#include <functional>
#include <iosfwd>
#include <map>
#include <tuple>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "ClusterTree.hpp"
#include "hssk.h"

namespace strumpack {
namespace HSS {

struct EngineOptions {
  double rel_tol = 1e-2, abs_tol = 1e-8;
  int leaf_size = 512, max_rank = 50000, d0 = 128, dd = 64, p = 10;
  int algorithm = 1;      // 0 original, 1 stable, 2 hard restart (original's acceptance rule; a failed round restarts every node)
  int random_engine = 0;  // 0 minstd_rand, 1 mt19937 (host, reference-identical), 2 philox (device)
  int random_dist = 0;    // 0 normal, 1 uniform
  // sketching matrix: 0 Gaussian (the distribution above), 1 SJLT = nnz entries +-1 per row (HSSMatrix.sketch.hpp);
  // sjlt_algo 0 chunk / 1 perm; nnz0 nonzeros per row in the first d0 + dd columns, nnz in every further dd
  int sketch = 0, sjlt_algo = 0, nnz0 = 4, nnz = 4;
  // the caller's multiplication routine also FILLS the random block (HSSOptions::user_defined_random,
  // HSSMatrix.compress_stable.hpp:110-141; sparse/fronts/FrontHSS.cpp:383-385): nothing is drawn here
  bool user_random = false;
  // The caller is going to factor: the ULV factorization of every tree level is enqueued -- on a stream of its own -- as soon
  // as the compression has settled that level, so the latency chain of the factorization's levels runs beside the chain of the
  // compression's upper levels instead of after it (extension; the reference factors when told to and not before).  factor()
  // then only waits.  Costs a caller who never factors the memory and the (overlapped) device time of the factors.
  bool factor_ahead = false;
  // The operand is SYMMETRIC (a hint of the caller's): with the one random matrix both products use
  // (compress_stable.hpp:77,139: Rc = Rr), Sc = A^T R is then Sr = A R and the second sketch GEMM is a copy.  1: trusted;
  // 2: checked first on a sample of 512 x 512 scattered entries against their mirror images (an error if they differ).
  // A wrong hint with 1 gives a wrong column basis -- a matrix that does not approximate the operand.
  int symmetric = 0;
  bool verbose = false;
  int device = 0;
  // multi-GPU (one process per GPU): `allgather` is an in-place all-gather of a DEVICE buffer of
  // world * bytes_per_rank bytes (rank r contributes the block at offset r * bytes_per_rank) --
  // RCCL over xGMI in production (strumpack_amd/dist.py), gloo in the CPU tests
  int world = 1, rank = 0;
  void (*allgather)(void* user, void* dbuf, long long bytes_per_rank) = nullptr;
  void* comm_user = nullptr;
  // stream-ordered collectives (native RCCL, Comm.hpp): issued on the engine's own stream, no host synchronisation.
  // When allgather_stream is null the host-synchronous `allgather` callback above is used and the two reductions are
  // emulated with it (all-gather of the partial blocks + a local sum).
  void (*allgather_stream)(void* user, void* dbuf, long long bytes_per_rank, void* stream) = nullptr;
  void (*allreduce_stream)(void* user, double* buf, long long count, void* stream) = nullptr;
  void (*reduce_scatter_stream)(void* user, const double* send, const long long* offs, const long long* counts, double* recv,
                                void* stream) = nullptr;
};

// process group of a multi-GPU matrix: either the native RCCL communicator (Comm.hpp; collectives on the engine's stream)
// or a host-synchronous all-gather callback (gloo in the CPU tests, torch.distributed in dist.py)
struct CommSpec {
  int world = 1, rank = 0;
  void (*allgather)(void*, void*, long long) = nullptr;
  void* user = nullptr;
  bool native = false;   // user is a comm::RcclComm*
  void apply(EngineOptions& e) const;
};

// host callbacks of the matrix-free / element interfaces (column-major host buffers)
using host_mult_t = std::function<void(char trans, int n, int nrhs, const double* R, int ldr, double* S, int lds)>;
// user_random: one call per sampling round; the callee fills R (n x nrhs, the random block: Rr == Rc) and both products
using host_sample_t = std::function<void(int n, int nrhs, double* R, double* Sr, double* Sc)>;
using host_elem_t = std::function<void(int m, const int* I, int n, const int* J, double* B, int ldb)>;
class DeviceHSS;
// compression rounds whose inner levels ran as one launch (hssk_tree_inner) / that fell back to the level-synchronous path after
// it reported a rank above its bound -- process-wide counters, for the tests
long long tree_pass_launches();
long long tree_pass_fallbacks();

struct PhaseStats {
  double t_compress = 0, t_sketch = 0, t_random = 0, t_tree = 0, t_factor = 0, t_solve = 0, t_mult = 0;
  double t_comm = 0;   // host-side time spent in the collectives of the compression (multi-GPU)
  double sketch_kernel_ms = 0;  // sum of HIP-event durations of the sketch GEMM main launches
  double sketch_kernel_flops = 0;  // algorithmic flops of those launches
  double sketch_kernel_bytes = 0;  // SJLT sketch: algorithmic HBM bytes of those launches (8 per element of A read)
  int sketch_launches = 0, rounds = 0, d_final = 0;
  double t_mark = 0;   // host clock at the end of compress() (STRUMPACK_AMD_TRACE_HOST)
  // algorithmic flop model
  double f_sketch = 0, f_local = 0, f_reduce = 0, f_id = 0, f_ortho = 0, f_ulv = 0, f_solve = 0;
  // algorithmic HBM bytes of one solve / one mat-vec: every block the sweep reads, once (vectors not included)
  double b_solve = 0, b_mult = 0;
};

class Arena;
struct HostRng;

class Arena;
// state between the forward and the backward half of a split solve (HSS::WorkSolve): device vectors of every node of the
// subtree in an arena of their own
struct SolveWork {
  std::shared_ptr<Arena> arena;
  std::vector<double*> f, y, zc, xb;
  double* db = nullptr;
  int nrhs = 0, sr = -1;
  bool valid = false, partial = false;
  // host buffers of the call in progress
  double* xroot = nullptr;
  double* reduced = nullptr;
  long long ldx = 0, ldr = 0;
};

class DeviceHSS {
 public:
  DeviceHSS(int n, const EngineOptions& opts, const structured::ClusterTree* tree = nullptr);
  ~DeviceHSS();
  DeviceHSS(const DeviceHSS&) = delete;
  DeviceHSS& operator=(const DeviceHSS&) = delete;

  // ---- construction (compression) ----
  void compress_dense_device(const double* dA, long long lda);       // A resident in HBM
  // A in host memory: streamed through the device in column blocks, uploads overlapped with the sketch GEMMs; the full
  // matrix is never resident in HBM (the reference's element sampler never stores A either, StructuredMatrix.cpp:214-262)
  void compress_dense_host(const double* A, long long lda);
  // A holds scalars of another type (HSSK_DT_F32 / _C32 / _C64, lda in scalars); this matrix is its real image (n() = rows for
  // float, 2 x rows for complex scalars): the operand is uploaded in its own format and expanded on the device
  void compress_dense_host_typed(const void* A, long long lda, int dtype);
  // the same for an operand given by element evaluation: fill(c0, c1, dst) writes A(:, c0:c1) (n x (c1-c0), ld n, host),
  // elem evaluates scattered blocks; both may be called concurrently from several host threads
  using host_fill_t = std::function<void(long long c0, long long c1, double* dst)>;
  void compress_host_blocks(const host_fill_t& fill, const host_elem_t& elem);
  // multi-GPU, sharded operand: this rank holds the rows and / or the columns [lo, hi) of its subtree (shard_range()):
  // dRows = A(lo:hi, :) ((hi-lo) x n, ldr) or null, dCols = A(:, lo:hi) (n x (hi-lo), ldc).  Without the row block
  // the operator is column-sharded: Sr = sum_g A(:, cols_g) R(cols_g, :) is reduced over the ranks.
  void compress_dense_device_sharded(const double* dRows, long long ldr, const double* dCols, long long ldc);
  // A given by one of the library's formulas (hssk_gen, include/hssk.h): never stored, on any number of ranks -- the sketch
  // kernel evaluates its tiles of the operand itself, scattered entries come from the formula (the reference's blocked
  // sampler over an element routine, structured/StructuredMatrix.cpp:214-262, taken to its end)
  void compress_generator(const hssk_gen& g);
  // rows / columns [lo, hi) owned by `rank` (its subtree below the cut); false if the tree cannot be cut for this world size
  bool shard_range(int rank, int& lo, int& hi) const;
  void compress_callbacks(const host_mult_t& mult, const host_elem_t& elem);  // matrix-free
  void compress_callbacks_user_random(const host_sample_t& sample, const host_elem_t& elem);
  // kernel matrix over points X (host, d x n, already in tree order); user_ann (k x n, optional) replaces the
  // device nearest-neighbour search of the first round (tests pin the compression against the reference's lists)
  struct KernelSpec {
    const double* X = nullptr;
    const double* dX = nullptr;   // the same points already on the engine's device (the clustering left them there): no upload
    int d = 0, type = 0, p = 1, ann = 64;
    double h = 1., lambda = 0.;
    // optional host neighbour search: fills ann (k x n, ids in cluster order, -1 = none) for the given k; replaces
    // hssk_knn in every round (the reference's randomized search, see NeighborSearch.hpp)
    std::function<void(int k, int* ann)> neighbors;
    // optional host evaluation of an entry (a user-defined Kernel subclass: kernel/Kernel.hpp:73-170, virtual eval): when set,
    // every block the compression needs is evaluated on the host's threads and uploaded instead of hssk_kernel_eval_vbatched
    // (neighbour lists still come from the point coordinates).  Must be callable concurrently.
    std::function<double(int i, int j)> eval;
  };
  void compress_kernel(const KernelSpec& ks, const int* user_ann = nullptr, int user_k = 0);

  // ---- serialization of the compressed representation (not the ULV factors): HSSMatrix::write / read ----
  void save(std::ostream& os) const;
  static std::unique_ptr<DeviceHSS> load(std::istream& is, const EngineOptions& opts);

  // ---- operations; x/b/y are column-major, host or device (on_device) ----
  void mult(char trans, int nrhs, const double* x, long long ldx, double* y, long long ldy,
            bool on_device, double beta = 0.0);
  void factor();
  void solve(int nrhs, double* b, long long ldb, bool on_device);
  void shift(double sigma);
  // back to the uncompressed state (tree kept): HSSMatrix::reset
  void reset() { OpGuard g(op_mu_); reset_compression(); }
  // the matrix is the real image [re -im; im re] (interleaved) of a complex one: adds the image of (re + i im) I
  void shift_cplx(double re, double im);

  // ---- Schur complement of the (0,0) block, H11 - H10 H00^{-1} H01 (HSSMatrix.Schur.hpp, factor.hpp:43-49);
  //      single-process matrices with a non-leaf root.  After partial_factor() + schur_update() the factors
  //      Theta = U1big B10 (n1 x rV0), DUB01 = D00^{-1} U0 B01 (mu0 x rV1), Phi = V1big DUB01^T (n1 x mu0) and
  //      Vhat (mu0 x rV0) stay resident in HBM:  S = H11 - Theta Vhat^T Phi^T.
  struct SchurDims { int n0 = 0, n1 = 0, rV0 = 0, rU0 = 0, rV1 = 0, rU1 = 0, mu0 = 0; };
  void partial_factor();
  // A node's diagonal block as an HSS matrix of its own (HSSMatrix::child(c)->factor() / ->solve(b), HSSMatrix.hpp:194-202):
  // ULV factors of the subtree with the node as root.  They share the node storage with the factors of the whole matrix (as
  // the reference's per-node ULV_ members do): factoring a child replaces them, and the other way round.
  void factor_node(int node) { factor_sub(node, false); }
  void solve_node(int node, int nrhs, double* b, long long ldb, bool on_device) { solve_sub(node, nrhs, b, ldb, on_device); }
  bool node_is_factored(int node) const { return node == 0 ? factored_ : sub_factored_ == node; }
  // the two halves of a solve (HSSMatrix::forward_solve / backward_solve): hss_solve.cpp
  void forward_solve_node(int node, struct SolveWork& w, int nrhs, const double* b, long long ldb, bool partial, double* xroot,
                          long long ldx, double* reduced, long long ldr);
  void backward_solve_node(int node, struct SolveWork& w, const double* xroot, long long ldx, double* x, long long ldxo);
  bool is_partially_factored() const { return partial_factored_; }
  SchurDims schur_dims() const;
  // computes the factors on the device; every non-null HOST pointer receives a copy (column-major)
  void schur_update(double* Theta, long long ldt, double* DUB01, long long ldd, double* Phi, long long ldp,
                    double* Vhat, long long ldv);
  // Sr = S R, Sc = S^T R  (R, Sr, Sc: n1 x c)                      (Schur_product_direct, Schur.hpp:73-143)
  void schur_product_direct(int c, const double* R, long long ldr, double* Sr, long long ldsr, double* Sc,
                            long long ldsc, bool on_device);
  // Sr = Sr1 - H10 R0 - (H11 - S) R1, Sc = Sc1 - H01^T R0 - (H11 - S)^T R1   (Schur_product_indirect, :145-221):
  // turns samples of the whole matrix into samples of the Schur complement
  void schur_product_indirect(int c, const double* R0, long long ldr0, const double* R1, long long ldr1,
                              const double* Sr1, long long ldsr1, const double* Sc1, long long ldsc1, double* Sr,
                              long long ldsr, double* Sc, long long ldsc, bool on_device);
  // rows of the (c)-th child's sub-matrix applied to x: y = op(H_cc) x  (child(c)->apply)
  void mult_child(int c, char trans, int nrhs, const double* x, long long ldx, double* y, long long ldy, bool on_device);
  // the same for the sub-matrix rooted at any node of the pre-order table (HSSMatrix::child(c)->child(c')...)
  void mult_node(int node, char trans, int nrhs, const double* x, long long ldx, double* y, long long ldy, bool on_device,
                 double beta = 0.0);
  // H(I_b, J_b) for a batch of requests by tree traversal (HSSMatrix::extract / extract_add, HSS/HSSMatrix.extract.hpp:36-104):
  // rows / cols = the requests' index lists concatenated (local to the sub-matrix rooted at `node`), roff / coff their prefix
  // sums (nb + 1 entries); out[b] (leading dimension ldo[b]) receives -- add: is incremented by -- block b; host or device
  // pointers.  One pair of launches for the whole batch, O(r^2 (|I| + |J|) log N + r^2 |I| |J|) per request.
  void extract_blocks(int node, int nb, const int* rows, const int* roff, const int* cols, const int* coff, double* const* out,
                      const int* ldo, bool on_device, bool add);
  // whether a request of ni x nj entries is cheaper by traversal than by nj products with the matrix
  bool extract_by_traversal_ok(int node, long long ni, long long nj) const;
  int node_end(int node) const { return subtree_end(node); }   // pre-order ids of the sub-tree: [node, node_end)
  int rank(int node) const;
  long long memory(int node) const;
  long long nonzeros(int node) const;

  // ---- introspection ----
  int rows() const { return n_; }
  bool is_compressed() const;
  bool is_factored() const { return factored_; }
  int levels() const;
  int rank() const;
  static constexpr long long kRefNodeBytes = 592;   // sizeof(HSS::HSSMatrix<double>) of the reference (see memory(int))
  long long memory() const;    // bytes of the compressed representation (+ kRefNodeBytes per node, as the reference counts)
  long long nonzeros() const;  // stored scalars + permutation entries
  long long factor_memory() const;
  int num_nodes() const { return (int)nodes_.size(); }
  // pre-order node table, 6 ints per node: row_offset, rows, U_rows, U_rank, V_rank, is_leaf
  void node_info(int* out) const;
  const PhaseStats& stats() const { return stats_; }
  hssk_ctx* ctx() const { return ctx_; }
  // Operations on ONE matrix from several host threads take turns (the reference's mult / solve are const and safe to call
  // concurrently, HSSMatrixBase.hpp:273 and the per-call work objects of apply / solve; here every operation uses the
  // matrix's stream and work arenas).  Different matrices run concurrently.
  using OpGuard = std::lock_guard<std::recursive_mutex>;
  const EngineOptions& options() const { return o_; }
  void set_options(const EngineOptions& o) { o_ = o; }

  struct Node {
    int lo = 0, m = 0, lvl = 0, height = 0, c0 = -1, c1 = -1, parent = -1;
    int Ustate = 0, Vstate = 0;  // 0 untouched, 1 partially compressed, 2 compressed
    int rU = 0, rV = 0, mU = 0, mV = 0;
    // compressed representation (device)
    double *D = nullptr, *B01 = nullptr, *B10 = nullptr, *XU = nullptr, *XV = nullptr;
    int *permU = nullptr, *permV = nullptr, *dIr = nullptr, *dIc = nullptr;
    std::vector<int> hpermU, hpermV, Ir, Ic;
    // compression workspace (device, transposed sample panels)
    double *Srt = nullptr, *Sct = nullptr, *Rrt = nullptr, *Rct = nullptr, *RrtRed = nullptr, *RctRed = nullptr;
    double *Qr = nullptr, *Qc = nullptr;  // orthogonal bases of the stable stopping test (mU x dcap)
    double Ur_max = 0, Vr_max = 0;
    bool panels = false;
    // ULV factors (device)
    double *Qt = nullptr, *Rlq = nullptr, *W1 = nullptr, *Vt0 = nullptr, *Dt = nullptr, *Vt1 = nullptr;
    // derived factors read by the single-launch solve sweeps: WQ = W1 Q~(:, 0:m-r), Vt0T = Vt0^T, inverted 64 x 64 diagonal blocks of
    // R~^T (non-root) resp. of the root's L and U
    double *WQ = nullptr, *Tinv = nullptr, *TinvU = nullptr, *Vt0T = nullptr;
    // chain block of the single-vector forward sweep: [ft1; z] = Gc [f; zc], (rU + rV) x (mU + mV) (DeviceHSS::chain_blocks)
    double* Gc = nullptr;
    double* LU = nullptr;
    int* piv = nullptr;
    bool leaf() const { return c0 < 0; }
    bool compressed() const { return Ustate == 2 && Vstate == 2; }
    bool untouched() const { return Ustate == 0 && Vstate == 0; }
  };
  const std::vector<Node>& nodes() const { return nodes_; }

 private:
  mutable std::recursive_mutex op_mu_;
  struct Source;
  struct DenseDeviceSource;
  struct HostBlockSource;
  struct ShardedDenseSource;
  struct CallbackSource;
  struct GeneratorSource;

  void check_symmetry(Source& src);
  void build_tree(const structured::ClusterTree* tree);
  void compress(Source& src);
  bool compress_attempt(Source& src, int dcap);
  void reset_compression();
  void restart_nodes(int d_have);
  void fill_random(int r0, int dn);
  void process_level(Source& src, const std::vector<int>& ids, int d, int dd, bool original);
  bool tree_pass(Source& src, int d, int dd);   // all inner levels of a round in one launch (hss_compress_tree.cpp)
  void extract_blocks(Source& src, const std::vector<int>& ids);
  void local_samples(const std::vector<int>& ids, const std::vector<int>& r0, const std::vector<int>& dn);
  void reduce_samples(const std::vector<int>& ids, const std::vector<int>& r0, const std::vector<int>& dn);
  void run_id(const std::vector<int>& ids, const std::vector<int>& which, int dtot);
  // host-side half of an ID commit, finished behind the next launches (see id_panels / finish_id_bookkeeping)
  struct PendingBook {
    std::vector<int> ids, which, hall;
    std::vector<size_t> idx_off, perm_off;
    size_t cnt = 0, idx_total = 0;
    bool active = false;
  } book_;
  bool defer_book_ = false;
  // the row ID of tall panels may be taken from their Gram matrices (id_panels): set by the kernel-matrix compression only --
  // its panels are thousands of rows tall and its tolerances loose; the sketch-based compression keeps the Householder forms
  bool id_gram_ = false;
  // kernel matrices: the panels id_panels receives have NOT been evaluated -- entry k describes panel k as a block of the kernel
  // matrix (out = the panel's place): the Gram form evaluates the entries while it multiplies (hssk_gram_gen_vbatched), every
  // other route evaluates its panels first
  const std::vector<hssk_keval_desc>* id_gen_ = nullptr;
  hssk_kernel_spec id_gen_spec_{};
  void finish_id_bookkeeping();
  void id_panels(const std::vector<int>& ids, const std::vector<int>& which, const std::vector<double*>& Ws,
                 const std::vector<int>& ds, const std::vector<const double*>* srcs = nullptr, int ldsrc = 0);
  void tsqr_reduce(const std::vector<int>& ids, const std::vector<int>& which, std::vector<double*>& Ws, std::vector<int>& ds);
  void ortho_test(const std::vector<int>& ids, const std::vector<int>& which, int d, int dd,
                  std::vector<char>& resolved);
  void free_compress_workspace();
  void ensure_ready(const char* what) const;
  // ---- sub-tree variants (node sr as the root of its own HSS matrix)
  int subtree_end(int sr) const;   // pre-order ids [sr, end)
  std::vector<std::vector<int>> sublists(const std::vector<std::vector<int>>& lists, int sr) const;
  void mult_sub(int sr, char trans, int nrhs, const double* x, long long ldx, double* y, long long ldy,
                bool on_device, double beta);
  void factor_sub(int sr, bool partial);
  void solve_sub(int sr, int nrhs, double* b, long long ldb, bool on_device, int phase = 0, struct SolveWork* ws = nullptr);
  // out (rank(sr) x c) = Ubig^T A or Vbig^T A  (apply_UtVt_big, Schur.hpp:223-252)
  void basis_up(int sr, bool useU, const double* dA, long long lda, int c, double* dOut, int ldout, Arena& wk);
  // out (rows(sr) x c) = Ubig in or Vbig in (apply_UV_big, Schur.hpp:254-323); !recurse: sr's own basis only
  void basis_down(int sr, bool useU, const double* dIn, int ldin, int c, double* dOut, long long ldo, Arena& wk,
                  bool recurse = true);
  // sweep plans: an apply / solve repeated on the same device buffers is recorded once (hssk_plan_*) and replayed
  struct PlanKey {
    int op; char trans; int nrhs; const void* x; void* y; long long ldx, ldy; double beta;
    bool operator<(const PlanKey& o) const {
      return std::tie(op, trans, nrhs, x, y, ldx, ldy, beta) < std::tie(o.op, o.trans, o.nrhs, o.x, o.y, o.ldx, o.ldy, o.beta);
    }
  };
  struct PlanEntry { int seen = 0; hssk_plan* plan = nullptr; };
  std::map<PlanKey, PlanEntry> plans_;
  std::unique_ptr<Arena> plan_arena_;   // work vectors of the recorded sweeps (never rewound while plans live)
  // chain blocks of the inner nodes (hssk_sweep_fwd_desc::G): built when a single-vector solve is about to be recorded -- a
  // caller that solves once never pays for them --, dropped with the factors
  std::unique_ptr<Arena> chain_arena_;
  bool chain_built_ = false;
  void chain_blocks();
  void drop_plans();
  bool plans_enabled() const;
  // ---- multi-GPU: subtree ownership below the cut level, replicated top
  bool mine(int id) const { return owner_[id] < 0 || owner_[id] == o_.rank; }
  void setup_ownership();
  void comm(void* dbuf, long long bytes_per_rank);
  void allreduce_sum(double* dbuf, long long count);
  void reduce_scatter_sum(const double* send, const std::vector<long long>& offs, const std::vector<long long>& counts, double* recv);
  void allgather_ints(std::vector<int>& v, int per_rank);
  void exchange_cut_compress(int dtot);
  bool exchange_cut_kernel(std::vector<std::vector<int>>& cols, bool failed);
  void exchange_node_table();
  void exchange_cut_factor();
  void allgather_rows(double* dx, long long ldx, int nrhs);

  int n_;
  EngineOptions o_;
  hssk_ctx* ctx_ = nullptr;
  std::vector<Node> nodes_;
  std::vector<std::vector<int>> by_height_, by_depth_;
  // nodes of this rank's subtree (depth >= cut) and the replicated top (depth < cut); without
  // subtree distribution own_* hold every node and top_* are empty
  std::vector<std::vector<int>> own_by_height_, top_by_height_, own_by_depth_, top_by_depth_;
  std::vector<int> owner_, cut_nodes_;
  bool dist_subtree_ = false;
  std::unique_ptr<Arena> comm_arena_;
  std::unique_ptr<Arena> persist_, work_, fact_, tmp_;
  // global transposed sample arrays (dcap x N)
  double *Rt_ = nullptr, *Srt_ = nullptr, *Sct_ = nullptr;
  double *Srt0_ = nullptr, *Sct0_ = nullptr;   // hard restart: the samples as drawn (the tree levels update Srt_ / Sct_ in place)
  int dcap_ = 0;
  int attempt_ = 0;   // compression attempts so far (sources re-carve their work buffers after a restart)
  // cut-exchange buffers of the distributed compression (exchange_cut_compress), reused across the rounds of an attempt
  double* cut_buf_ = nullptr;
  int* cut_idx_ = nullptr;
  size_t cut_buf_cap_ = 0, cut_idx_cap_ = 0;
  int cut_gen_ = -1;
  const int* sj_pat_ = nullptr;   // SJLT pattern of the sample block filled last (device, nnz x N)
  int sj_nnz_ = 0;
  long long cols_per_rank_ = 0;  // sketch column shard (multi-GPU)
  int* d_ranks_ = nullptr;
  // a factorization in progress: the state its levels share (hss_factor.cpp).  `ahead` = started by the compression
  // (EngineOptions::factor_ahead): its launches are on fctx_'s stream, `done` heights of own_by_height_ are enqueued.
  struct FactorRun {
    bool active = false, ahead = false, partial = false;
    int sr = 0;
    hssk_ctx* cx = nullptr;
    std::vector<double*> Dh, Vh, Vd;
    std::vector<hssk_trtri_desc> ti;
    std::vector<char> is_cut;
    size_t done = 0;
  } frun_;
  hssk_ctx* fctx_ = nullptr;   // second context (stream, descriptor rings) of a factorization that runs ahead
  void factor_begin(int sr, bool partial, hssk_ctx* cx);
  void factor_prep(const std::vector<int>& ids);     // dense column bases of these nodes (they depend on the compression only)
  void factor_level(const std::vector<int>& ids);
  void factor_ahead_level(size_t h);                  // compression hook: height h has been processed
  void factor_cancel();                               // waits for a run ahead and forgets it
  bool factored_ = false, partial_factored_ = false, schur_ready_ = false;
  // every way the ULV factors die (shift, recompression, restart, reset): the whole matrix's, a child's (factor_node) and
  // the partial factorization with its Schur factors go together -- a later solve / solve_node / Schur_update must refuse
  void invalidate_factors() { factor_cancel(); factored_ = partial_factored_ = schur_ready_ = false; sub_factored_ = -1; }
  int sub_factored_ = -1;   // node whose subtree was ULV-factored as a matrix of its own (factor_node), -1: none
  std::unique_ptr<Arena> schur_;
  // device-resident node table of the extraction kernels (persist arena; rebuilt after a compression)
  void ensure_dev_tree();
  const hssk_tree_node* dev_tree_ = nullptr;
  int ex_rmax_ = 1, ex_depth_ = 1;
  double *sTheta_ = nullptr, *sPhi_ = nullptr, *sDUB01_ = nullptr, *sVtDUB01_ = nullptr, *sW_ = nullptr;
  PhaseStats stats_;
  std::shared_ptr<HostRng> rng_;
};

}  // namespace HSS
}  // namespace strumpack
