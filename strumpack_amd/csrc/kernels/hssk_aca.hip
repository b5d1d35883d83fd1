// Batched adaptive cross approximation of dense tiles: A (m x n) ~ U V^T, one workgroup per tile.
//
// Reference: adaptive_cross_approximation (dense/ACA.cpp:41-118), the tile compression of BLR matrices under
// --blr_low_rank_algorithm ACA (BLR/LRTile.cpp:66-74).  Same sequence of decisions: the first row is the reference's
// (its default-seeded std::mt19937 draw on [0, m), computed by the caller), each step takes the residual of the current
// row, its largest entry among the columns not yet used (first index on ties, as std::max_element), the residual of that
// column scaled by the pivot, the running estimate ||U V^T||_F^2 += 2 sum_l (u_l . u)(v_l . v) + |u|^2 |v|^2, and stops
// when |u| |v| < rtol ||U V^T||_F or < atol; the next row is the largest entry of the new column among the rows not yet used.
//
// A step touches one row and one column of the tile and the rank columns of U and V found so far (L2-resident: a tile
// and its factors are < 1 MB); its reductions are a dozen barrier-separated stages, so the kernel is latency-bound like the
// pivoted QR it replaces, at rank steps of ~(rank + 2)(m + n) flops instead of m n.
#include "hssk_device.h"
#include "hssk_internal.h"

#include <algorithm>
#include <cfloat>
#include <vector>

namespace {

constexpr int ACA_T = 256;
constexpr int ACA_MAXD = 2048;   // largest tile dimension (selected-row / -column flags live in LDS)

struct ArgMax { double v; int i; };
// the largest v, smallest index among equals (std::max_element); all threads of the workgroup call, all get the result
__device__ __forceinline__ ArgMax block_argmax(double v, int i, double* s_v, int* s_i) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  hssk_wave_argmax(v, i);
  if (lane == 0) { s_v[wave] = v; s_i[wave] = i; }
  __syncthreads();
  ArgMax r{s_v[0], s_i[0]};
#pragma unroll
  for (int w = 1; w < ACA_T / 64; w++)
    if (s_v[w] > r.v || (s_v[w] == r.v && s_i[w] < r.i)) { r.v = s_v[w]; r.i = s_i[w]; }
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(ACA_T) void aca_kernel(const hssk_aca_desc* __restrict__ descs) {
  HSSK_SHARED unsigned char s_rsel[ACA_MAXD];
  HSSK_SHARED unsigned char s_csel[ACA_MAXD];
  HSSK_SHARED double s_du[ACA_MAXD / 4];
  HSSK_SHARED double s_dv[ACA_MAXD / 4];
  HSSK_SHARED double s_v[ACA_T / 64];
  HSSK_SHARED int s_i[ACA_T / 64];
  HSSK_SHARED double s_scal[2];
  const hssk_aca_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = p.m, n = p.n;
  const int rmax = min(min(m, n), min(p.max_rank, ACA_MAXD / 4));
  for (int e = tid; e < m; e += ACA_T) s_rsel[e] = 0;
  for (int e = tid; e < n; e += ACA_T) s_csel[e] = 0;
  __syncthreads();
  int row = p.row0, rank = 0;
  double approx2 = 0.;   // ||U V^T||_F^2 so far
  while (rank < rmax) {
    if (tid == 0) s_rsel[row] = 1;
    // ---- residual of row `row`: v_j = A(row, j) - sum_l U(row, l) V(j, l); the largest |v_j| among the unused columns
    double bv = -2.;
    int bi = 0x7fffffff;
    for (int j = tid; j < n; j += ACA_T) {
      double v = hssk_gload(p.A, (size_t)row + (size_t)j * p.lda);
      for (int l = 0; l < rank; l++) v -= hssk_gload(p.U, (size_t)row + (size_t)l * p.ldu) * hssk_gload(p.V, (size_t)j + (size_t)l * p.ldv);
      hssk_gstore(p.V, (size_t)j + (size_t)rank * p.ldv, v);
      const double a = s_csel[j] ? -1. : fabs(v);
      if (a > bv) { bv = a; bi = j; }
    }
    __syncthreads();   // (the flag of `row`, the stores of v)
    const ArgMax cm = block_argmax(bv, bi, s_v, s_i);
    const int col = cm.i;
    const double piv = hssk_gload(p.V, (size_t)col + (size_t)rank * p.ldv);
    __syncthreads();   // (every thread holds the pivot before its owner rescales it)
    if (fabs(piv) < DBL_MIN) break;
    if (tid == 0) s_csel[col] = 1;
    const double ipiv = 1. / piv;
    for (int j = tid; j < n; j += ACA_T) hssk_gstore(p.V, (size_t)j + (size_t)rank * p.ldv, hssk_gload(p.V, (size_t)j + (size_t)rank * p.ldv) * ipiv);
    // ---- residual of column `col`: u_i = A(i, col) - sum_l U(i, l) V(col, l)
    for (int i = tid; i < m; i += ACA_T) {
      double u = hssk_gload(p.A, (size_t)i + (size_t)col * p.lda);
      for (int l = 0; l < rank; l++) u -= hssk_gload(p.U, (size_t)i + (size_t)l * p.ldu) * hssk_gload(p.V, (size_t)col + (size_t)l * p.ldv);
      hssk_gstore(p.U, (size_t)i + (size_t)rank * p.ldu, u);
    }
    __syncthreads();
    // ---- du_l = u_l . u, dv_l = v_l . v for l <= rank (a wave per l)
    for (int l = wave; l <= rank; l += ACA_T / 64) {
      double su = 0., sv = 0.;
      for (int i = lane; i < m; i += 64) su += hssk_gload(p.U, (size_t)i + (size_t)l * p.ldu) * hssk_gload(p.U, (size_t)i + (size_t)rank * p.ldu);
      for (int j = lane; j < n; j += 64) sv += hssk_gload(p.V, (size_t)j + (size_t)l * p.ldv) * hssk_gload(p.V, (size_t)j + (size_t)rank * p.ldv);
      su = hssk_wave_sum(su);
      sv = hssk_wave_sum(sv);
      if (lane == 0) { s_du[l] = su; s_dv[l] = sv; }
    }
    __syncthreads();
    if (tid == 0) {
      double cross = 0.;
      for (int l = 0; l < rank; l++) cross += s_du[l] * s_dv[l];
      s_scal[0] = approx2 + 2. * cross + s_du[rank] * s_dv[rank];
      s_scal[1] = s_du[rank] * s_dv[rank];
    }
    __syncthreads();
    approx2 = s_scal[0];
    const double nrm_uv = sqrt(s_scal[1]);
    rank++;
    if (nrm_uv < sqrt(approx2) * p.rtol || nrm_uv < p.atol) break;
    // ---- next row: the largest |u_i| among the unused rows
    bv = -2.;
    bi = 0x7fffffff;
    for (int i = tid; i < m; i += ACA_T) {
      const double a = s_rsel[i] ? -1. : fabs(hssk_gload(p.U, (size_t)i + (size_t)(rank - 1) * p.ldu));
      if (a > bv) { bv = a; bi = i; }
    }
    const ArgMax rm = block_argmax(bv, bi, s_v, s_i);
    row = rm.i;
  }
  if (tid == 0) *p.rank = rank;
}

}  // namespace

extern "C" int hssk_aca_vbatched(hssk_ctx* ctx, const hssk_aca_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  for (int i = 0; i < count; i++) {
    const hssk_aca_desc& d = descs[i];
    if (d.m <= 0 || d.n <= 0 || d.m > ACA_MAXD || d.n > ACA_MAXD) HSSK_UNSUPPORTED("ACA: tile beyond 2048 rows / columns (or empty)");
    if (d.row0 < 0 || d.row0 >= d.m || d.ldu < d.m || d.ldv < d.n) throw std::invalid_argument("hssk_aca_vbatched: bad descriptor");
  }
  auto* dd = (const hssk_aca_desc*)ctx->stage(descs, sizeof(*descs) * count);
  HSSK_LAUNCH(aca_kernel, dim3((unsigned)count), dim3(ACA_T), 0, ctx->stream, dd);
  hssk_rt::check_launch();
  HSSK_API_END
}
