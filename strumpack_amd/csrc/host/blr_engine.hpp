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
  // truncated RRQR of the listed tiles of the array (batched), results into tiles_
  void compress_tiles(const std::vector<std::pair<int, int>>& ij, const char* adm);
  void factor_rl(const char* adm);

  mutable std::recursive_mutex op_mu_;
  int m_, n_;
  std::vector<int> roff_, coff_;
  BLREngineOptions o_;
  hssk_ctx* ctx_ = nullptr;
  double* dA_ = nullptr;   // n x n working array / diagonal tiles
  long long ld_ = 0;
  int* dpiv_ = nullptr;    // pivots of the diagonal tiles (0-based, local), rows() ints
  std::vector<Tile> tiles_;
  std::unique_ptr<Arena2> store_, tmp_;
  bool compressed_ = false, factored_ = false;
};

}  // namespace BLR
}  // namespace strumpack
