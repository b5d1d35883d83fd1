// Sub-block extraction H(I, J) of a compressed HSS matrix by tree traversal, batched over many requests.
//
// Reference behaviour restated: HSSMatrix::extract / extract_add (HSS/HSSMatrix.extract.hpp:36-104: extract_fwd collects
// V^H restricted to the requested columns going up the tree, extract_bwd expands through the coupling blocks and U going
// down) -- what a sparse HSS front calls for every block of its parent's assembly, O(r^2 (|I| + |J|) log N) per request
// instead of |J| products with the whole matrix.  Restructured for the device: for an index i the row of the nested basis
// Ubig_t(i, :) at an ancestor t follows from the row one level below by one small product with t's interpolative basis
// U_t = P [I; E] (rows of the child inside U_t); so
//   1. extract_chain_kernel: one wave per requested row (column) index walks leaf -> root and leaves that index's basis row
//      at EVERY depth in a work array (the transfer through U = P [I; E] is a gather for the identity rows and a contiguous
//      column of X = E^T for the others);
//   2. extract_pair_kernel: one thread per requested entry (i, j) finds the lowest common ancestor by descending from the
//      root, and forms u_i^T B v_j with the two basis rows one level below it (B01 or B10), or reads D when both indices
//      fall into one leaf.
// Requests of any size and number are one pair of launches; the matrix is only read.
#include "hssk_device.h"
#include "hssk_internal.h"

#include <algorithm>

namespace {

constexpr int EX_DEPTH = 48;   // deepest tree walked (2^48 leaves: any matrix)
constexpr int EX_RL = 4;       // basis rows up to 64 EX_RL entries

// path root -> leaf of index x: node ids by depth into `path`, returns the leaf's depth
__device__ __forceinline__ int ex_descend(const hssk_tree_node* __restrict__ nodes, int root, int x, int* path) {
  int t = root, d = 0;
  path[0] = t;
  while (nodes[t].c0 >= 0 && d + 1 < EX_DEPTH) {
    const int c1 = nodes[t].c1;
    t = x < nodes[c1].lo ? nodes[t].c0 : c1;
    path[++d] = t;
  }
  return d;
}

// One wave per index.  work[(idx * maxdepth + depth) * rmax + k] = k-th entry of the basis row at that depth (depth >= 1:
// the root carries no basis).  rows first (bases U), then columns (bases V).
__global__ __launch_bounds__(256) void extract_chain_kernel(const hssk_tree_node* __restrict__ nodes, int root, int rmax, int maxdepth,
                                                            const int* __restrict__ rows, int nrows, const int* __restrict__ cols, int ncols,
                                                            double* __restrict__ work) {
  const int lane = threadIdx.x & 63;
  const int idx = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (idx >= nrows + ncols) return;
  const bool isrow = idx < nrows;
  const int x = isrow ? rows[idx] : cols[idx - nrows];
  int path[EX_DEPTH];
  const int dl = ex_descend(nodes, root, x, path);
  double u[EX_RL];   // entry k = lane + 64 t of the current basis row
#pragma unroll
  for (int t = 0; t < EX_RL; t++) u[t] = 0.;
  int rc = 0;        // its length
  for (int d = dl; d >= 1; d--) {
    const hssk_tree_node nd = nodes[path[d]];
    const int r = isrow ? nd.rU : nd.rV, m = isrow ? nd.mU : nd.mV;
    const double* X = isrow ? nd.XU : nd.XV;
    const int* iperm = isrow ? nd.ipermU : nd.ipermV;
    double un[EX_RL];
#pragma unroll
    for (int t = 0; t < EX_RL; t++) un[t] = 0.;
    if (d == dl) {
      // leaf: row (x - lo) of P [I; E]
      const int k = iperm[x - nd.lo];
#pragma unroll
      for (int t = 0; t < EX_RL; t++) {
        const int kk = lane + 64 * t;
        if (kk < r) un[t] = k < r ? (kk == k ? 1. : 0.) : hssk_gload(X, (size_t)kk + (size_t)(k - r) * r);
      }
    } else {
      // inner: u_t = sum_q u_child[q] * row (off + q) of P [I; E], off = 0 for the first child, rank of the first child otherwise
      const hssk_tree_node c0n = nodes[nd.c0];
      const int off = path[d + 1] == nd.c0 ? 0 : (isrow ? c0n.rU : c0n.rV);
      for (int q = 0; q < rc; q++) {
        const double uq = hssk_bcast_lane(q < 64 ? u[0] : (q < 128 ? u[1] : (q < 192 ? u[2] : u[3])), q & 63);
        const int k = iperm[off + q];
#pragma unroll
        for (int t = 0; t < EX_RL; t++) {
          const int kk = lane + 64 * t;
          if (kk < r) un[t] += k < r ? (kk == k ? uq : 0.) : uq * hssk_gload(X, (size_t)kk + (size_t)(k - r) * r);
        }
      }
      (void)m;
    }
#pragma unroll
    for (int t = 0; t < EX_RL; t++) {
      u[t] = un[t];
      const int kk = lane + 64 * t;
      if (kk < r) work[((size_t)idx * maxdepth + d) * rmax + kk] = un[t];
    }
    rc = r;
  }
}

// One thread per requested entry.
__global__ __launch_bounds__(256) void extract_pair_kernel(const hssk_tree_node* __restrict__ nodes, int root, int rmax, int maxdepth,
                                                           const int* __restrict__ rows, int nrows, const int* __restrict__ cols,
                                                           const hssk_extract_block* __restrict__ blocks, const long long* __restrict__ pair_off,
                                                           int nblocks, long long npairs, const double* __restrict__ work, int accumulate) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= npairs) return;
  // block of this entry: binary search over the prefix sums
  int lo = 0, hi = nblocks - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (pair_off[mid] <= e) lo = mid; else hi = mid - 1;
  }
  const hssk_extract_block b = blocks[lo];
  const long long le = e - pair_off[lo];
  const int ii = (int)(le % b.ni), jj = (int)(le / b.ni);
  const int ri = b.ri0 + ii, cj = nrows + b.cj0 + jj;
  const int i = rows[ri], j = cols[b.cj0 + jj];
  // lowest common ancestor
  int t = root, d = 0;
  double val;
  for (;;) {
    const hssk_tree_node nd = nodes[t];
    if (nd.c0 < 0) { val = hssk_gload(nd.D, (size_t)(i - nd.lo) + (size_t)(j - nd.lo) * nd.m); break; }
    const int mid = nodes[nd.c1].lo;
    const bool iL = i < mid, jL = j < mid;
    if (iL == jL) { t = iL ? nd.c0 : nd.c1; d++; continue; }
    // i and j part here: B01 (i left, j right: rU(c0) x rV(c1)) or B10 (rU(c1) x rV(c0))
    const hssk_tree_node a = nodes[nd.c0], c = nodes[nd.c1];
    const double* B = iL ? nd.B01 : nd.B10;
    const int ru = iL ? a.rU : c.rU, rv = iL ? c.rV : a.rV;
    const double* u = work + ((size_t)ri * maxdepth + (d + 1)) * rmax;
    const double* v = work + ((size_t)cj * maxdepth + (d + 1)) * rmax;
    double s = 0.;
    for (int q = 0; q < rv; q++) {
      double tq = 0.;
      for (int k = 0; k < ru; k++) tq += u[k] * hssk_gload(B, (size_t)k + (size_t)q * ru);
      s += tq * v[q];
    }
    val = s;
    break;
  }
  double* o = b.out + (size_t)ii + (size_t)jj * b.ldo;
  *o = accumulate ? *o + val : val;
}

}  // namespace

extern "C" int hssk_hss_extract(hssk_ctx* ctx, const hssk_tree_node* nodes, int root, int rmax, int maxdepth, const int* rows, int nrows,
                                const int* cols, int ncols, const hssk_extract_block* blocks, const long long* pair_off, int nblocks,
                                long long npairs, int accumulate, double* work) {
  HSSK_API_BEGIN
  if (nblocks <= 0 || npairs <= 0) return 0;
  if (rmax > 64 * EX_RL) HSSK_UNSUPPORTED("ranks beyond 256");
  if (maxdepth > EX_DEPTH) HSSK_UNSUPPORTED("tree deeper than 48 levels");
  const int nidx = nrows + ncols;
  HSSK_LAUNCH(extract_chain_kernel, dim3((unsigned)((nidx + 3) / 4)), dim3(256), 0, ctx->stream, nodes, root, rmax, maxdepth, rows, nrows, cols, ncols, work);
  HSSK_LAUNCH(extract_pair_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, ctx->stream, nodes, root, rmax, maxdepth, rows, nrows, cols,
              blocks, pair_off, nblocks, npairs, work, accumulate);
  hssk_rt::check_launch();
  HSSK_API_END
}
