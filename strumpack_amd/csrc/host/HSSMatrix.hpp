// HSS::HSSMatrix<double>: the reference's public HSS class (HSS/HSSMatrix.hpp:79-711) on top of the
// device engine (hss_engine.hpp).  construct / compress / mult / apply / factor / solve / shift and
// the introspection calls keep the reference's names, argument meaning and error behaviour; the
// arithmetic runs in hand-written HIP kernels on the MI355X.
#pragma once
#include <memory>

#include "HSSOptions.hpp"
#include "StructuredMatrix.hpp"
#include "hss_engine.hpp"

namespace strumpack {
namespace kernel {
template <typename scalar_t> class Kernel;
}
namespace HSS {

template <typename scalar_t> class HSSMatrix;

// HSS::WorkSolve (HSS/HSSExtra.hpp): what travels from forward_solve to backward_solve.  x is the solution in the reduced
// unknowns of the root of the call and reduced_rhs (partial solves) = Vhat^* x + V^* [z_0; z_1]: the two members a sparse front
// reads and updates between the halves (sparse/fronts/FrontHSS.cpp:458-493); the vectors of the nodes below stay on the device.
template <typename scalar_t> class WorkSolve;
template <> class WorkSolve<double> {
 public:
  DenseMatrix<double> x, reduced_rhs;

 private:
  friend class HSSMatrix<double>;
  mutable SolveWork state_;
};

template <> class HSSMatrix<double> : public structured::StructuredMatrix<double> {
  using scalar_t = double;
  using DenseM_t = DenseMatrix<scalar_t>;
  using DenseMW_t = DenseMatrixWrapper<scalar_t>;
  using opts_t = HSSOptions<scalar_t>;

 public:
  // sampling callback: Sr = A Rr, Sc = A^H Rc ; element callback: B = A(I, J)  (HSSMatrix.hpp:83-89)
  using mult_t = std::function<void(DenseM_t& Rr, DenseM_t& Rc, DenseM_t& Sr, DenseM_t& Sc)>;
  using elem_t = std::function<void(const std::vector<std::size_t>& I, const std::vector<std::size_t>& J, DenseM_t& B)>;

  HSSMatrix() {}
  // HSSMatrix::child(c) (HSS/HSSMatrix.hpp:194-202): "a child of an HSS matrix is itself an HSS matrix".  Here a child is
  // an HSSMatrix that VIEWS a node of its parent's device-resident tree: introspection, mult / apply / applyC, dense,
  // extract / get, print_info and child() of the view work on the diagonal block of that node; compressing, factoring or
  // solving with a child on its own is not offered (logic_error) -- the ULV factors belong to the whole tree.  Views stay
  // valid as long as the parent is neither destroyed nor re-compressed.
  HSSMatrix(DeviceHSS* parent_engine, int node);
  // compress the dense matrix A (HSS/HSSMatrix.cpp:50-54)
  HSSMatrix(const DenseM_t& A, const opts_t& opts) : HSSMatrix(A.rows(), A.cols(), opts) { compress(A, opts); }
  // uncompressed m x n HSS matrix with the bisection tree (HSSMatrix.cpp:56-70)
  HSSMatrix(std::size_t m, std::size_t n, const opts_t& opts);
  // tree given by a cluster tree (HSSMatrix.cpp:72-86)
  HSSMatrix(const structured::ClusterTree& t, const opts_t& opts);
  // kernel matrix: clusters the points (reordering K's data), builds the tree and compresses from the point
  // coordinates without a random sketch (HSS/HSSMatrix.cpp:88-106, HSSMatrix.compress_kernel.hpp)
  HSSMatrix(kernel::Kernel<double>& K, const opts_t& opts);
  // extension: one process per GPU (every rank holds all points; subtree ownership as in compress_device_sharded)
  HSSMatrix(kernel::Kernel<double>& K, const opts_t& opts, int world, int rank, void (*allgather)(void*, void*, long long), void* user);
  ~HSSMatrix() override;

  void compress(const DenseM_t& A, const opts_t& opts);
  // extension (HSSMatrixPromoted.hpp): this matrix is the real image of a host matrix of another scalar type (dtype =
  // HSSK_DT_F32 / _C32 / _C64, lda in scalars)
  void compress_image(const void* A, std::size_t lda, int dtype, const opts_t& opts);
  void compress(const mult_t& Amult, const elem_t& Aelem, const opts_t& opts);
  // extension: operand given by element evaluation only -- the columns are evaluated in blocks on the host threads and
  // streamed through the device (the reference's tile sampler, structured/StructuredMatrix.cpp:214-262)
  void compress_from_elements(const elem_t& Aelem, const opts_t& opts);
  void compress(const kernel::Kernel<double>& K, const opts_t& opts);
  // extension (tests): the neighbour lists of the first round are given (k x n, 0-based, column i = point i)
  void compress_with_neighbors(const kernel::Kernel<double>& K, const opts_t& opts, const int* ann, int k);
  const double* device_points_ = nullptr;   // set while the kernel constructor compresses: the points the device clustering left on the device
  // extension: A resident in HBM
  void compress_device(const double* dA, long long lda, const opts_t& opts);
  // extension: one process per GPU (subtree ownership below the cut level, replicated top);
  // `allgather` is an in-place all-gather of a device buffer (RCCL)
  void compress_device_sharded(const double* dA, long long lda, const opts_t& opts, int world, int rank,
                               void (*allgather)(void*, void*, long long), void* user);
  // the same with a process group (native RCCL communicator or callback), replicated operand
  void compress_device_sharded(const double* dA, long long lda, const opts_t& opts, const CommSpec& pg);
  // sharded operand: this rank's row block A(lo:hi, :) (may be null: column-sharded operator) and column block
  // A(:, lo:hi), [lo, hi) = shard_range(pg.rank)
  void compress_device_blocks(const double* dRows, long long ldr, const double* dCols, long long ldc, const opts_t& opts,
                              const CommSpec& pg);
  // extension: the matrix is one of the library's formulas (hssk_gen kinds, include/hssk.h) -- never stored; with a process
  // group every rank evaluates what its subtree needs (no operand shards, no exchange beyond the cut nodes')
  void compress_generator(int kind, const opts_t& opts);
  void compress_generator(int kind, const opts_t& opts, const CommSpec& pg);
  HSSMatrix(kernel::Kernel<double>& K, const opts_t& opts, const CommSpec& pg);

  std::size_t rows() const override { return rows_; }
  std::size_t cols() const override { return cols_; }
  std::size_t memory() const override;
  std::size_t nonzeros() const override;
  std::size_t rank() const override;
  std::size_t levels() const;
  bool is_compressed() const;
  bool leaf() const;

  void mult(Trans op, const DenseM_t& x, DenseM_t& y) const override;
  using structured::StructuredMatrix<double>::mult;
  DenseM_t apply(const DenseM_t& b) const;
  DenseM_t applyC(const DenseM_t& b) const;
  void factor() override;
  void solve(DenseM_t& b) const override;
  using structured::StructuredMatrix<double>::solve;
  void shift(scalar_t sigma) override;
  DenseM_t dense() const;
  // H(I, J) and H(i, j) (reference: HSSMatrix.extract.hpp:36-104): tree traversal on the device
  DenseM_t extract(const std::vector<std::size_t>& I, const std::vector<std::size_t>& J) const;
  scalar_t get(std::size_t i, std::size_t j) const;
  // H(I, J) added into B (HSSMatrix.hpp:430-434, extract_add)
  void extract_add(const std::vector<std::size_t>& I, const std::vector<std::size_t>& J, DenseM_t& B) const;
  // extension: many requests in one pair of kernel launches, B[b] = H(I[b], J[b]) (add: B[b] += ...)
  void extract_blocks(const std::vector<std::vector<std::size_t>>& I, const std::vector<std::vector<std::size_t>>& J,
                      std::vector<DenseM_t>& B, bool add = false) const;
  // ---- Schur complement of the (0,0) block, as the sparse HSS fronts use it (sparse/fronts/FrontHSS.cpp:391-407):
  //   partial_factor(): ULV of child(0) only (HSSMatrix.hpp:330, factor.hpp:43-49)
  //   Schur_update(Theta, DUB01, Phi): Theta = U1big B10, DUB01 = D00^{-1} U0 B01, Phi = V1big DUB01^H, so that
  //       H11 - H10 H00^{-1} H01 = H11 - Theta Vhat^H Phi^H        (HSSMatrix.hpp:456, Schur.hpp:40-59)
  //   Vhat(): child(0)->ULV().Vhat() (HSSExtra.hpp:191)
  //   Schur_product_direct(...): Sr = S R, Sc = S^H R              (HSSMatrix.hpp:459, Schur.hpp:73-143)
  //   Schur_product_indirect(...): samples of H -> samples of S    (HSSMatrix.hpp:465, Schur.hpp:145-221)
  // The factors stay resident in HBM after Schur_update(); the matrices passed back into Schur_product_* must be the
  // ones it returned (checked by shape) and are not uploaded again.  DUB01 / Phi / Vhat are expressed in this
  // library's reduced unknowns of block 0, so only products such as Vhat^H DUB01 or Theta Vhat^H Phi^H compare
  // with the reference's.
  // the two halves of solve(b) (HSSMatrix.hpp:360-376).  partial: after the parent's partial_factor(), on child(0) -- the root
  // of the call keeps its column basis and w.reduced_rhs is formed for the front's update part; w.x may be changed between
  // the two calls (the front subtracts Phi^* y_upd).
  // scalars of the ULV factors as the reference counts them (HSSMatrixBase.cpp:73-78, HSSExtra.hpp:183-186: L, Vt0, W1, Q of
  // every eliminated node, D of the root): 0 before factor()
  std::size_t factor_nonzeros() const;
  void forward_solve(WorkSolve<double>& w, const DenseM_t& b, bool partial) const;
  void backward_solve(WorkSolve<double>& w, DenseM_t& x) const;
  void partial_factor();
  void Schur_update(DenseM_t& Theta, DenseM_t& DUB01, DenseM_t& Phi) const;
  DenseM_t Vhat() const;
  void Schur_product_direct(const DenseM_t& Theta, const DenseM_t& DUB01, const DenseM_t& Phi,
                            const DenseM_t& ThetaVhatC_or_VhatCPhiC, const DenseM_t& R, DenseM_t& Sr, DenseM_t& Sc) const;
  void Schur_product_indirect(const DenseM_t& DUB01, const DenseM_t& R0, const DenseM_t& R1, const DenseM_t& Sr1,
                              const DenseM_t& Sc1, DenseM_t& Sr, DenseM_t& Sc) const;
  // y = op(H_cc) x with H_cc the diagonal block of child c (child(c)->apply / applyC, HSSMatrix.hpp:194-202)
  DenseM_t apply_child(int c, Trans op, const DenseM_t& x) const;
  // child c (0 or 1; the matrix must not be a leaf) as an HSS matrix that views this one's tree (see the view constructor)
  const HSSMatrix<double>* child(int c) const;
  HSSMatrix<double>* child(int c);
  // the root node's own bases (HSSMatrixBase.hpp:346-349): zero for a root, the node's ranks for a child
  std::size_t U_rank() const;
  std::size_t V_rank() const;
  std::size_t U_rows() const;
  std::size_t V_rows() const;
  std::pair<std::size_t, std::size_t> dims() const { return {rows(), cols()}; }
  // ULV().Vhat() (HSSExtra.hpp:191): on child(0) after the parent's partial_factor() (sparse/fronts/FrontHSS.cpp:395)
  struct Factors {
    const HSSMatrix<double>* self;
    DenseM_t Vhat() const;
  };
  Factors ULV() const { return Factors{this}; }
  bool is_view() const { return veng_ != nullptr; }
  int node() const { return vnode_; }
  void print_info(std::ostream& out = std::cout, std::size_t roff = 0, std::size_t coff = 0) const;
  // gnuplot rectangles of the HSS partition, off-diagonal blocks coloured by rank (HSS/HSSMatrix.cpp:367-404); the free
  // draw(H, name) below writes the script plot<name>.gnuplot as the reference's does (:407-417)
  void draw(std::ostream& of, std::size_t rlo = 0, std::size_t clo = 0) const;
  // deep copy of the compressed representation (HSSMatrix.hpp:186; the ULV factors are not copied: factor() the clone)
  std::unique_ptr<HSSMatrix<double>> clone() const;
  // back to the uncompressed state, keeping the tree (HSSMatrix.hpp:313)
  void reset();
  // OpenMP task depth of the reference's recursive algorithms (HSSMatrixBase.hpp:273): no meaning for the level-synchronous
  // device engine; accepted so that callers compile
  void set_openmp_task_depth(int) {}
  // frees the (1,1) block once the Schur complement has been formed (HSSMatrix.hpp:470; sparse/fronts/FrontHSS.cpp:411).  The
  // engine keeps its blocks in arenas that are released with the matrix, so this only marks the block as gone: products with
  // the whole matrix are refused afterwards, the Schur factors stay usable.
  void delete_trailing_block() { trailing_deleted_ = true; }
  bool trailing_block_deleted() const { return trailing_deleted_; }
  // binary file with the compressed representation (tree, D, B, bases; not the ULV factors) and back
  // (HSSMatrix.cpp:438-510; the file layout is this library's own, see hss_io.cpp)
  void write(const std::string& fname) const;
  static HSSMatrix<double> read(const std::string& fname);
  HSSMatrix(HSSMatrix<double>&&) = default;
  HSSMatrix<double>& operator=(HSSMatrix<double>&&) = default;

  // device-resident operands (extension)
  void mult_device(Trans op, int nrhs, const double* dx, long long ldx, double* dy, long long ldy, double beta = 0.) const;
  void solve_device(int nrhs, double* db, long long ldb) const;

  DeviceHSS* engine() const { return eng_ ? eng_.get() : veng_; }
  static EngineOptions engine_options(const opts_t& opts);

 private:
  void make_engine(const opts_t& opts, const structured::ClusterTree* t);
  std::size_t rows_ = 0, cols_ = 0;
  std::unique_ptr<structured::ClusterTree> tree_;
  mutable std::unique_ptr<DeviceHSS> eng_;
  DeviceHSS* veng_ = nullptr;   // view: the parent's engine ...
  int vnode_ = 0;               // ... and the node this matrix is rooted at (0: the whole tree)
  bool trailing_deleted_ = false;
  void owner(const char* what) const { if (veng_) throw std::logic_error(std::string(what) + ": not offered on a child view (the operation belongs to the whole HSS tree)"); }
  mutable std::unique_ptr<HSSMatrix<double>> ch_[2];
};

// writes plot<name>.gnuplot (HSS/HSSMatrix.hpp:706, HSSMatrix.cpp:407-417)
void draw(const HSSMatrix<double>& H, const std::string& name);
// y = op(H) x + beta y   (HSS/HSSMatrix.hpp:720, HSSMatrix.cpp:419-435)
void apply_HSS(Trans op, const HSSMatrix<double>& A, const DenseMatrix<double>& B, double beta, DenseMatrix<double>& C);

}  // namespace HSS
}  // namespace strumpack
