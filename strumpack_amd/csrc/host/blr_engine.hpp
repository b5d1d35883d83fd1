// DeviceBLR: block low-rank (BLR) compression, mat-vec, LU factorization and solve of a dense matrix on the MI355X, over
// the batched kernels of include/hssk.h (the dense slice of the reference's BLR path -- the
// sparse multifrontal driver around it stays out of scope).
//
// Reference behaviour followed (all under /root/reference/src/BLR):
//   BLRMatrix.cpp:88-112, 563-570   compress: one tile per pair of clusters, admissible tiles as U V^T if
//                                   rank (m + n) <= m n, dense otherwise
//   LRTile.cpp / dense/DenseMatrix.cpp:low_rank   truncated RRQR (geqp3tol) with rel_tol / abs_tol / max_rank
//   BLRMatrix.cpp:114-243           compress_and_factor, default algorithm RL: LU of the diagonal tile, compression of the
//                                   block row / column, triangular solves on the factors, Schur updates into full rank
//   BLRMatrix.hpp:118-122, BLRMatrix.cpp:1767-  solve (laswp + two block triangular solves), mult
//   BLRMatrix.GPU.cpp:71-262, BLRBatch.hpp:46-168  the reference's batched GPU formulation of the same steps (precedent
//                                   for executing a block row / column of a step as variable-size batched kernels)
//
// Layout in HBM: the operand (or its running Schur complement) as one column-major n x n array; every off-diagonal tile
// as a pair U (m x r), V (n x r) carved from a bump arena -- a tile kept dense is stored as U = the tile, V = I, so that
// every step below is one code path; diagonal tiles in place in the array (their LU factors after factorization).
// The truncated RRQR of a tile is the batched register-resident QRCP of the HSS engine (hssk_id_vbatched): the column ID
// T ~ T(:, J) [I X] P^T spans the same subspace as Q R P^T with the same stopping rule, so U = T(:, J), V = P [I; X^T].
#pragma once
#include <memory>
#include <mutex>
#include <vector>

#include "hssk.h"

namespace strumpack {
namespace BLR {

struct BLREngineOptions {
  double rel_tol = 1e-4, abs_tol = 1e-10;
  int max_rank = 5000;
  int device = 0;
  bool verbose = false;
  int lr_algo = 0;   // tile compression: 0 truncated pivoted QR (RRQR), 1 adaptive cross approximation (ACA)
};

class Arena2;

class DeviceBLR {
 public:
  DeviceBLR(int m, const std::vector<int>& rowtiles, int n, const std::vector<int>& coltiles, const BLREngineOptions& o);
  ~DeviceBLR();
  DeviceBLR(const DeviceBLR&) = delete;
  DeviceBLR& operator=(const DeviceBLR&) = delete;

  // adm: rowblocks x colblocks, column-major, non-zero = compress the tile (null: every tile)
  void compress_host(const double* A, long long lda, const char* adm);
  void compress_device(const double* dA, long long lda, const char* adm);
  // square matrix, same clusters for rows and columns: LU with compression (RL); diagonal tiles are always dense
  void compress_and_factor_host(const double* A, long long lda, const char* adm);
  void compress_and_factor_device(const double* dA, long long lda, const char* adm);

  // ---- frontal matrix [F11 F12; F21 F22] of a multifrontal factorization (BLRMatrix::construct_and_partial_factor,
  // BLRMatrix.cpp:740-1037, algorithm RL; batched precedent BLRMatrix.GPU.cpp:71-262; caller sparse/fronts/FrontBLR.cpp:329-432).
  // The matrix of this object is the whole front: clusters = the separator's tiles followed by the update tiles; the first
  // `sep_blocks` block steps are eliminated -- LU of the diagonal tile, compression of the block row of [F11 F12] and of the
  // block column of [F11; F21] (F11 tiles per `adm11`, sep_blocks x sep_blocks column-major, null = all but the diagonal;
  // F12 / F21 tiles always), triangular solves on the factors, Schur update of the trailing part of F11, F12, F21 AND of
  // F22 <- F22 - F21 F11^{-1} F12, which stays a dense array.  Any of F12 / F21 / F22 may be null when the front has no update part.
  void partial_factor_host(int sep_blocks, const double* F11, long long ld11, const double* F12, long long ld12,
                           const double* F21, long long ld21, const double* F22, long long ld22, const char* adm11);
  void partial_factor_device(int sep_blocks, const double* F11, long long ld11, const double* F12, long long ld12,
                             const double* F21, long long ld21, const double* F22, long long ld22, const char* adm11);
  int sep_rows() const { return roff_[nsteps_]; }
  int upd_rows() const { return m_ - roff_[nsteps_]; }
  int sep_blocks() const { return nsteps_; }
  // the Schur complement F22 - F21 F11^{-1} F12 (upd_rows() square): copy to the host / in place in HBM (column-major, schur_ld())
  void schur_host(double* F22, long long ld) const;
  const double* schur_device() const { return dA_ ? blk(nsteps_, nsteps_) : nullptr; }
  long long schur_ld() const { return ld_; }
  // the two solve phases of a front (FrontBLR.cpp:525-570; BLRMatrix::trsmLNU_gemm / gemm_trsmUNN, BLRMatrix.cpp:1552-1665):
  //   forward:  bsep <- L11^{-1} P bsep;  bupd <- bupd - B21 bsep          backward:  ysep <- U11^{-1} (ysep - B12 yupd)
  // host vectors; bupd / yupd may be null when the front has no update part
  void front_forward(int nrhs, double* bsep, long long ldb, double* bupd, long long ldu) const;
  void front_backward(int nrhs, double* ysep, long long ldy, const double* yupd, long long ldu) const;
  // ranks of all tiles, rowblocks() x colblocks() column-major: >= 0 rank of a U V^T tile, -1 dense (diagonal, not
  // admissible, rank does not pay, or the untouched F22 part)
  void tile_ranks(int* out) const;
  // stored scalars of B11 / B12 / B21 of a front (the reference's F11blr_ / F12blr_ / F21blr_ .nonzeros())
  void front_nonzeros(long long out[3]) const;

  void mult(char trans, int nrhs, const double* x, long long ldx, double* y, long long ldy) const;   // host vectors
  void solve(int nrhs, double* b, long long ldb) const;                                             // host, in place
  void dense(double* A, long long lda) const;   // host image of the compressed (not factored) matrix

  int rows() const { return m_; }
  int cols() const { return n_; }
  int rowblocks() const { return (int)roff_.size() - 1; }
  int colblocks() const { return (int)coff_.size() - 1; }
  bool factored() const { return factored_; }
  int rank() const;             // largest rank of a compressed tile
  long long memory() const;     // bytes of the representation
  long long nonzeros() const;   // stored scalars
  double t_compress = 0, t_factor = 0;
  // phases of the last factorization on the device clock (ms) and its algorithmic flops: [0] LU of the diagonal tiles,
  // [1] tile compression, [2] triangular solves, [3] Schur-update GEMMs; f_schur = flops of [3]
  double phase_ms[4] = {0, 0, 0, 0};
  double f_schur = 0, f_total = 0;
  double b_schur = 0;   // algorithmic bytes of the Schur-update GEMMs (operands once, the updated block read and written)
  int schur_launches = 0;
  bool time_phases = false;

 private:
  struct Tile {
    int r = -1;          // rank of the U V^T form; -1: tile lives dense in the array only (diagonal / not yet processed)
    bool lowrank = false;   // false with r >= 0: kept dense, stored as U = tile, V = I
    double *U = nullptr, *V = nullptr;
  };
  Tile& tile(int i, int j) { return tiles_[(size_t)i + (size_t)j * rowblocks()]; }
  const Tile& tile(int i, int j) const { return tiles_[(size_t)i + (size_t)j * rowblocks()]; }
  int tm(int i) const { return roff_[i + 1] - roff_[i]; }
  int tn(int j) const { return coff_[j + 1] - coff_[j]; }
  double* blk(int i, int j) const { return dA_ + roff_[i] + (size_t)coff_[j] * ld_; }
  void load(const double* A, long long lda, bool on_device);
  void alloc_array();
  void free_array();
  void put_block(int r0, int c0, int rows, int cols, const double* src, long long lds, bool on_device);
  void partial_factor(int sep_blocks, const double* F11, long long ld11, const double* F12, long long ld12, const double* F21,
                      long long ld21, const double* F22, long long ld22, const char* adm11, bool on_device);
  bool sweep(double* X, bool backward) const;   // one right-hand side: a substitution as ONE launch (hssk_blr_sweep)
  void fwd(double* X, int nrhs, double* t, int Rmax) const;
  void bwd(double* X, int nrhs, double* t, int Rmax) const;
  int rmax() const;
  // truncated RRQR of the listed tiles of the array (batched), results into tiles_
  void compress_tiles(const std::vector<std::pair<int, int>>& ij, const char* adm);
  void factor_rl(const char* adm, int nsteps);

  mutable std::recursive_mutex op_mu_;
  int m_, n_;
  std::vector<int> roff_, coff_;
  BLREngineOptions o_;
  hssk_ctx* ctx_ = nullptr;
  hssk_ctx* ctx2_ = nullptr;   // second stream: the diagonal tile's LU runs next to the compression of its block row / column
  double* dA_ = nullptr;   // n x n working array / diagonal tiles (a chunk of the process-wide pool)
  size_t dA_bytes_ = 0;
  long long ld_ = 0;
  int* dpiv_ = nullptr;    // pivots of the diagonal tiles (0-based, local), rows() ints
  std::vector<const double*> invL_, invU_;   // per block step: inverted 64 x 64 diagonal blocks of the tile's L and U 
  std::vector<Tile> tiles_;
  std::unique_ptr<Arena2> store_, tmp_, blk_;   // blk_: products kept for a block of steps (deferred Schur updates, factor_rl)
  // row / term tables of the single-launch substitutions (forward, backward), resident in store_ from the first solve on
  struct SweepTables { bool built = false, ok = false; int nrows = 0; hssk_blr_row* rows = nullptr; hssk_blr_term* terms = nullptr; };
  mutable SweepTables sweep_tab_[2];
  bool compressed_ = false, factored_ = false;
  int nsteps_ = 0;   // eliminated block steps (== rowblocks() for a full factorization)
};

}  // namespace BLR
}  // namespace strumpack
