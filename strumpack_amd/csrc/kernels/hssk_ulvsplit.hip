// hssk_ulv_split: the first step of a node's ULV elimination in ONE launch (HSSMatrix.factor.hpp:109-118).  With the row
// ID U = P [I; X^T] of the node and its m x m block D:
//     W1    = (P^T D)(0:r, :)                         (r x m)
//     W0^T  = (P^T D)(r:, :)^T - W1^T X               (m x (m - r): the panel whose QR is the node's LQ)
// As two row gathers (one of them transposing) and a batched GEMM this read D twice along its rows -- one element per
// cache line -- and took 0.35 ms at the leaf level of N = 1e5.  Here a workgroup takes 32 columns of D, reads them once,
// coalesced, into LDS (stored in permuted row order), and produces the 32 columns of W1 and the 32 ROWS of W0^T they
// determine: the row gathers disappear, the product is a short loop on LDS operands.
#include "hssk_device.h"
#include "hssk_internal.h"

#include <algorithm>
#include <vector>

namespace {

constexpr int US_T = 256;
constexpr int US_C = 32;       // columns of D per workgroup
constexpr int US_MMAX = 256;   // largest block
constexpr int US_XJ = 8;       // the X buffer holds US_XJ columns at full rank (r = m), proportionally more at lower ranks
struct UsWork { int prob, cblock; };

// MMAX: largest block of the batch the LDS arrays are sized for (208 rows: two workgroups per CU)
template <int MMAX>
__global__ __launch_bounds__(US_T) void ulv_split_kernel(const hssk_ulvsplit_desc* __restrict__ descs, const UsWork* __restrict__ work) {
  HSSK_SHARED double s_D[MMAX * (US_C + 1)];   // [k][c]: row k of P^T D (the tile is stored in PERMUTED row order), odd stride
  HSSK_SHARED double s_X[US_XJ * MMAX];        // X(:, j0 : j0 + xj), column jj at s_X + jj * r
  HSSK_SHARED int s_inv[MMAX];
  const UsWork w = work[blockIdx.x];
  const hssk_ulvsplit_desc p = descs[w.prob];
  const int tid = threadIdx.x, m = p.m, r = p.r, q = m - r;
  const int c0 = w.cblock * US_C, nc = min(US_C, m - c0);
  for (int k = tid; k < m; k += US_T) s_inv[p.perm[k]] = k;
  __syncthreads();
  // coalesced along the rows of D, stored at the row's position in P^T D
  // (eight loads in flight per thread: a load per iteration would wait out a memory round trip each)
  for (int e0 = tid; e0 < m * nc; e0 += 8 * US_T) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int e = e0 + u * US_T;
      v[u] = e < m * nc ? hssk_gload(p.D, (size_t)(e % m) + (size_t)(c0 + e / m) * p.ldd) : 0.;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int e = e0 + u * US_T;
      if (e < m * nc) s_D[s_inv[e % m] * (US_C + 1) + e / m] = v[u];
    }
  }
  __syncthreads();
  // W1(k, c0 + c) = (P^T D)(k, c0 + c)
  for (int e = tid; e < r * nc; e += US_T) {
    const int k = e % r, c = e / r;
    hssk_gstore(p.W1, (size_t)k + (size_t)(c0 + c) * p.ldw, s_D[k * (US_C + 1) + c]);
  }
  // W0^T(c0 + c, j) = (P^T D)(r + j, c0 + c) - sum_k (P^T D)(k, c0 + c) X(k, j):  thread = (c, one of 8 column groups);
  // X goes through LDS US_XJ columns at a time (read as broadcasts: the 32 threads of a group share their column)
  const int c = tid % US_C, jg = tid / US_C;
  const int xj = r > 0 ? max(US_T / US_C, (US_XJ * MMAX) / r) : q;   // as many columns as the buffer holds at this rank
  for (int j0 = 0; j0 < q; j0 += xj) {
    const int nj = min(xj, q - j0);
    __syncthreads();
    for (int e = tid; e < r * nj; e += US_T) s_X[(e / r) * r + (e % r)] = hssk_gload(p.X, (size_t)(e % r) + (size_t)(j0 + e / r) * p.ldx);
    __syncthreads();
    if (c < nc)
      for (int jj = jg; jj < nj; jj += US_T / US_C) {
        const double* x = s_X + jj * r;
        const double* dc = s_D + c;
        double s0 = dc[(r + j0 + jj) * (US_C + 1)], s1 = 0., s2 = 0., s3 = 0.;
        int k = 0;
        for (; k + 3 < r; k += 4) {
          s0 -= dc[k * (US_C + 1)] * x[k];
          s1 -= dc[(k + 1) * (US_C + 1)] * x[k + 1];
          s2 -= dc[(k + 2) * (US_C + 1)] * x[k + 2];
          s3 -= dc[(k + 3) * (US_C + 1)] * x[k + 3];
        }
        for (; k < r; k++) s0 -= dc[k * (US_C + 1)] * x[k];
        hssk_gstore(p.W0t, (size_t)(c0 + c) + (size_t)(j0 + jj) * p.ldt, (s0 + s1) + (s2 + s3));
      }
  }
}

}  // namespace

extern "C" int hssk_ulv_split(hssk_ctx* ctx, const hssk_ulvsplit_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  std::vector<UsWork> work;
  for (int p = 0; p < count; p++) {
    const hssk_ulvsplit_desc& d = descs[p];
    if (d.m <= 0) continue;
    if (d.m > US_MMAX || d.r < 0 || d.r > d.m) HSSK_UNSUPPORTED("block beyond the LDS tile");   // (larger blocks: the caller's gathers + GEMM)
    for (int cb = 0; cb * US_C < d.m; cb++) work.push_back(UsWork{p, cb});
  }
  if (work.empty()) return 0;
  auto* dd = (const hssk_ulvsplit_desc*)ctx->stage(descs, sizeof(*descs) * count);
  auto* dw = (const UsWork*)ctx->stage(work.data(), sizeof(UsWork) * work.size());
  int mmax = 0;
  for (int p = 0; p < count; p++) mmax = std::max(mmax, descs[p].m);
  if (mmax <= 128) HSSK_LAUNCH(ulv_split_kernel<128>, dim3((unsigned)work.size()), dim3(US_T), 0, ctx->stream, dd, dw);
  else if (mmax <= 208) HSSK_LAUNCH(ulv_split_kernel<208>, dim3((unsigned)work.size()), dim3(US_T), 0, ctx->stream, dd, dw);
  else HSSK_LAUNCH(ulv_split_kernel<US_MMAX>, dim3((unsigned)work.size()), dim3(US_T), 0, ctx->stream, dd, dw);
  hssk_rt::check_launch();
  HSSK_API_END
}
