// Batched triangular solves and the (small) root LU of the ULV factorization.
//
// Reference call sites: trsm(Side::L, UpLo::L, ...) with the LQ factor L in the forward solve
// (HSS/HSSMatrix.solve.hpp:161-162; dense/DenseMatrix.cpp:1059-1085); DenseMatrix::LU / solve =
// getrf / getrs at the root (HSS/HSSMatrix.factor.hpp:104-106, solve.hpp:133-135;
// dense/DenseMatrix.cpp:564-640).
//
// trsm: one workgroup per node; each wave owns right-hand-side columns (strided by the wave count);
// the column being solved lives in LDS; per row step either a shuffle-reduced dot product with a
// contiguous column of T (transposed forms) or an axpy with a contiguous column of T (plain forms),
// so T is always read along its contiguous dimension.  Bound: latency (n dependent steps).
// getrf: one workgroup per matrix, right-looking with partial pivoting, columns contiguous.
#include "hssk_device.h"
#include "hssk_internal.h"

#include <algorithm>
#include <vector>

namespace {

constexpr int TR_THREADS = 256;
constexpr int TR_WAVES = TR_THREADS / 64;

// one workgroup per (problem, group of TR_WAVES right-hand sides): the groups of a problem run side by side
struct TrWork {
  int prob, group;
};
__global__ __launch_bounds__(TR_THREADS) void trsm_kernel(const hssk_trsm_desc* __restrict__ descs, const TrWork* __restrict__ work) {
  HSSK_DYN_SHARED(double, xs_all);
  const TrWork wk = work[blockIdx.x];
  const hssk_trsm_desc p = descs[wk.prob];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = p.n, ldt = p.ldt;
  const double* __restrict__ T = p.T;
  double* xs = xs_all + (size_t)wave * n;
  // effective orientation: forward substitution when (lower, N) or (upper, T)
  const bool forward = (p.lower != 0) == (p.transT == 0);
  {
    const int c0 = wk.group * TR_WAVES;
    const int c = c0 + wave;
    const bool valid = c < p.nrhs;
    double* b = p.B + (size_t)(valid ? c : 0) * p.ldb;
    for (int i = lane; i < n; i += 64) xs[i] = valid ? b[i] : 0.;
    __syncthreads();
    for (int step = 0; step < n; step++) {
      const int i = forward ? step : n - 1 - step;
      if (p.transT) {
        // x_i = (x_i - sum_{l solved} T(l, i) x_l) / T(i, i): column i of T is contiguous
        const double* tc = T + (size_t)i * ldt;
        double s = 0.;
        if (forward) for (int l = lane; l < i; l += 64) s += tc[l] * xs[l];
        else for (int l = i + 1 + lane; l < n; l += 64) s += tc[l] * xs[l];
        const double xi = xs[i];
        s = hssk_wave_sum(s);
        if (lane == 0) xs[i] = p.unit ? (xi - s) : (xi - s) / tc[i];
      } else {
        // x_i final; x_r -= T(r, i) x_i for unsolved r: column i of T is contiguous
        const double* tc = T + (size_t)i * ldt;
        const double xi = p.unit ? xs[i] : xs[i] / tc[i];
        __syncthreads();  // every lane has read xs[i] before lane 0 stores the scaled value
        if (lane == 0) xs[i] = xi;
        if (forward) for (int r = i + 1 + lane; r < n; r += 64) xs[r] -= tc[r] * xi;
        else for (int r = lane; r < i; r += 64) xs[r] -= tc[r] * xi;
      }
      __syncthreads();
    }
    if (valid) for (int i = lane; i < n; i += 64) b[i] = xs[i];
    __syncthreads();
  }
}

constexpr int LU_THREADS = 256;

// One workgroup per matrix, partial pivoting (first arg max), right-looking.  W / ld: the array the steps work on -- the
// matrix itself, or its image in LDS (getrf_lds_kernel: n <= LU_LDS_N; the six barrier-separated stages of a step then
// wait on LDS instead of L2: the 82 x 82 root of N = 1e5 took 167 us in global memory).
__device__ __forceinline__ int getrf_body(double* __restrict__ A, int ld, int n, int* __restrict__ piv, double* s_val, int* s_idx, int* s_piv) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int info = 0;
  for (int k = 0; k < n; k++) {
    // pivot: first arg max_{i >= k} |A(i, k)|
    double bv = -1.;
    int bi = 0x7fffffff;
    for (int i = k + tid; i < n; i += LU_THREADS) {
      double v = fabs(A[i + (size_t)k * ld]);
      if (v > bv) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      double ov = hssk_shfl_xor(bv, o);
      int oi = hssk_shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { s_val[wave] = bv; s_idx[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      double v = s_val[0];
      int ix = s_idx[0];
      for (int w = 1; w < LU_THREADS / 64; w++)
        if (s_val[w] > v || (s_val[w] == v && s_idx[w] < ix)) { v = s_val[w]; ix = s_idx[w]; }
      *s_piv = ix;
      piv[k] = ix;
    }
    __syncthreads();
    const int pv = *s_piv;
    if (pv != k)
      for (int j = tid; j < n; j += LU_THREADS) {
        double a = A[k + (size_t)j * ld], b = A[pv + (size_t)j * ld];
        A[k + (size_t)j * ld] = b;
        A[pv + (size_t)j * ld] = a;
      }
    __syncthreads();
    const double akk = A[k + (size_t)k * ld];
    if (akk == 0.) {
      if (!info) info = k + 1;
      __syncthreads();
      continue;
    }
    const double inv = 1. / akk;
    __syncthreads();  // everyone holds akk before the column is scaled
    for (int i = k + 1 + tid; i < n; i += LU_THREADS) A[i + (size_t)k * ld] *= inv;
    __syncthreads();
    // trailing update: one column per wave, lanes along the (contiguous) column
    const double* lk = A + (size_t)k * ld;
    for (int j = k + 1 + wave; j < n; j += LU_THREADS / 64) {
      double* col = A + (size_t)j * ld;
      const double ukj = col[k];
      for (int i = k + 1 + lane; i < n; i += 64) col[i] -= lk[i] * ukj;
    }
    __syncthreads();
  }
  return info;
}
__global__ __launch_bounds__(LU_THREADS) void getrf_kernel(const hssk_lu_desc* __restrict__ descs) {
  HSSK_SHARED double s_val[LU_THREADS / 64];
  HSSK_SHARED int s_idx[LU_THREADS / 64];
  HSSK_SHARED int s_piv;
  const hssk_lu_desc p = descs[blockIdx.x];
  const int info = getrf_body(p.A, p.lda, p.n, p.piv, s_val, s_idx, &s_piv);
  if (threadIdx.x == 0) *p.info = info;
}
constexpr int LU_LDS_N = 128;
__global__ __launch_bounds__(LU_THREADS) void getrf_lds_kernel(const hssk_lu_desc* __restrict__ descs) {
  HSSK_SHARED double s_A[LU_LDS_N * (LU_LDS_N + 1)];
  HSSK_SHARED double s_val[LU_THREADS / 64];
  HSSK_SHARED int s_idx[LU_THREADS / 64];
  HSSK_SHARED int s_piv;
  const hssk_lu_desc p = descs[blockIdx.x];
  const int n = p.n, lds = n | 1, tid = threadIdx.x;
  for (int e = tid; e < n * n; e += LU_THREADS) s_A[(e % n) + (e / n) * lds] = p.A[(e % n) + (size_t)(e / n) * p.lda];
  __syncthreads();
  const int info = getrf_body(s_A, lds, n, p.piv, s_val, s_idx, &s_piv);
  __syncthreads();
  for (int e = tid; e < n * n; e += LU_THREADS) p.A[(e % n) + (size_t)(e / n) * p.lda] = s_A[(e % n) + (e / n) * lds];
  if (tid == 0) *p.info = info;
}

// ---- blocked LU for large matrices (root of an HSS matrix with ranks in the hundreds): right-looking, LUB columns
// per panel; the panel is factored by one workgroup (same pivoting rule as getrf_kernel, restricted to the panel's
// columns), its row interchanges are applied to the other columns by lu_swap_kernel, U12 = L11^{-1} A12 is a batched
// unit-lower trsm and A22 -= L21 U12 a batched MFMA GEMM.
constexpr int LUB = 64;
__global__ __launch_bounds__(LU_THREADS) void getrf_panel_kernel(const hssk_lu_desc* __restrict__ descs, int j0) {
  HSSK_SHARED double s_val[LU_THREADS / 64];
  HSSK_SHARED int s_idx[LU_THREADS / 64];
  HSSK_SHARED int s_piv;
  const hssk_lu_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = p.n, ld = p.lda;
  if (j0 >= n) return;
  const int jend = min(n, j0 + LUB);
  double* __restrict__ A = p.A;
  int info = j0 == 0 ? 0 : *p.info;
  for (int k = j0; k < jend; k++) {
    double bv = -1.;
    int bi = 0x7fffffff;
    for (int i = k + tid; i < n; i += LU_THREADS) {
      const double v = fabs(A[i + (size_t)k * ld]);
      if (v > bv) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = hssk_shfl_xor(bv, o);
      const int oi = hssk_shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { s_val[wave] = bv; s_idx[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      double v = s_val[0];
      int ix = s_idx[0];
      for (int w = 1; w < LU_THREADS / 64; w++)
        if (s_val[w] > v || (s_val[w] == v && s_idx[w] < ix)) { v = s_val[w]; ix = s_idx[w]; }
      s_piv = ix;
      p.piv[k] = ix;
    }
    __syncthreads();
    const int pv = s_piv;
    if (pv != k)
      for (int j = j0 + tid; j < jend; j += LU_THREADS) {
        const double a = A[k + (size_t)j * ld], b = A[pv + (size_t)j * ld];
        A[k + (size_t)j * ld] = b;
        A[pv + (size_t)j * ld] = a;
      }
    __syncthreads();
    const double akk = A[k + (size_t)k * ld];
    if (akk == 0.) {
      if (!info) info = k + 1;
      __syncthreads();
      continue;
    }
    const double inv = 1. / akk;
    __syncthreads();
    for (int i = k + 1 + tid; i < n; i += LU_THREADS) A[i + (size_t)k * ld] *= inv;
    __syncthreads();
    const double* lk = A + (size_t)k * ld;
    for (int j = k + 1 + wave; j < jend; j += LU_THREADS / 64) {
      double* col = A + (size_t)j * ld;
      const double ukj = col[k];
      for (int i = k + 1 + lane; i < n; i += 64) col[i] -= lk[i] * ukj;
    }
    __syncthreads();
  }
  if (tid == 0) *p.info = info;
}

// the row interchanges piv[j0 .. j0+LUB) applied to the columns outside the panel, one thread per column
__global__ void lu_swap_kernel(const hssk_lu_desc* __restrict__ descs, int j0) {
  const hssk_lu_desc p = descs[blockIdx.y];
  const int n = p.n, jend = min(n, j0 + LUB);
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n - (jend - j0)) return;
  if (c >= j0) c += jend - j0;   // skip the panel's own columns
  double* col = p.A + (size_t)c * p.lda;
  for (int k = j0; k < jend; k++) {
    const int pv = p.piv[k];
    if (pv != k) { const double t = col[k]; col[k] = col[pv]; col[pv] = t; }
  }
}

// B <- P B (row interchanges of getrf applied in order), one thread per right-hand side
__global__ void laswp_kernel(const hssk_lusolve_desc* __restrict__ descs) {
  const hssk_lusolve_desc p = descs[blockIdx.x];
  for (int c = threadIdx.x; c < p.nrhs; c += blockDim.x) {
    double* b = p.B + (size_t)c * p.ldb;
    for (int k = 0; k < p.n; k++) {
      int pv = p.piv[k];
      if (pv != k) { double t = b[k]; b[k] = b[pv]; b[pv] = t; }
    }
  }
}

// the same with the right-hand sides spread over blockIdx.y (wide blocks: a whole block row of a BLR factorization)
__global__ void laswp_wide_kernel(const hssk_lusolve_desc* __restrict__ descs) {
  const hssk_lusolve_desc p = descs[blockIdx.x];
  for (int c = blockIdx.y * blockDim.x + threadIdx.x; c < p.nrhs; c += gridDim.y * blockDim.x) {
    double* b = p.B + (size_t)c * p.ldb;
    for (int k = 0; k < p.n; k++) {
      int pv = p.piv[k];
      if (pv != k) { double t = b[k]; b[k] = b[pv]; b[pv] = t; }
    }
  }
}

}  // namespace

extern "C" {

int hssk_laswp_vbatched(hssk_ctx* ctx, const hssk_lusolve_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  int cmax = 0;
  for (int i = 0; i < count; i++) cmax = std::max(cmax, descs[i].nrhs);
  if (cmax <= 0) return 0;
  auto* dd = (const hssk_lusolve_desc*)ctx->stage(descs, sizeof(*descs) * count);
  HSSK_LAUNCH(laswp_wide_kernel, dim3((unsigned)count, (unsigned)std::min(256, (cmax + 63) / 64)), dim3(64), 0, ctx->stream, dd);
  hssk_rt::check_launch();
  HSSK_API_END
}

int hssk_trsm_vbatched(hssk_ctx* ctx, const hssk_trsm_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  int nmax = 0;
  for (int i = 0; i < count; i++) nmax = std::max(nmax, descs[i].n);
  if (nmax == 0) return 0;
  size_t shmem = sizeof(double) * (size_t)nmax * TR_WAVES;
  if (shmem > 150 * 1024) throw std::runtime_error("hssk_trsm_vbatched: triangular block too large for LDS");
  std::vector<TrWork> work;
  for (int i = 0; i < count; i++)
    if (descs[i].n > 0)
      for (int g = 0; g * TR_WAVES < descs[i].nrhs; g++) work.push_back(TrWork{i, g});
  if (work.empty()) return 0;
  auto* dd = (const hssk_trsm_desc*)ctx->stage(descs, sizeof(*descs) * count);
  auto* dw = (const TrWork*)ctx->stage(work.data(), sizeof(TrWork) * work.size());
  HSSK_LAUNCH(trsm_kernel, dim3((unsigned)work.size()), dim3(TR_THREADS), shmem, ctx->stream, dd, dw);
  hssk_rt::check_launch();
  HSSK_API_END
}

int hssk_getrf_vbatched(hssk_ctx* ctx, const hssk_lu_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  auto* dd = (const hssk_lu_desc*)ctx->stage(descs, sizeof(*descs) * count);
  int nmax = 0;
  for (int i = 0; i < count; i++) nmax = std::max(nmax, descs[i].n);
  if (nmax <= 384) {
    if (nmax <= LU_LDS_N) HSSK_LAUNCH(getrf_lds_kernel, dim3((unsigned)count), dim3(LU_THREADS), 0, ctx->stream, dd);
    else HSSK_LAUNCH(getrf_kernel, dim3((unsigned)count), dim3(LU_THREADS), 0, ctx->stream, dd);
  } else {
    std::vector<hssk_trsm_desc> tr;
    std::vector<hssk_gemm_desc> gm;
    for (int j0 = 0; j0 < nmax; j0 += LUB) {
      HSSK_LAUNCH(getrf_panel_kernel, dim3((unsigned)count), dim3(LU_THREADS), 0, ctx->stream, dd, j0);
      HSSK_LAUNCH(lu_swap_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)count), dim3(256), 0, ctx->stream, dd, j0);
      tr.clear(); gm.clear();
      for (int i = 0; i < count; i++) {
        const hssk_lu_desc& d = descs[i];
        const int jend = std::min(d.n, j0 + LUB), nb = jend - j0, rest = d.n - jend;
        if (nb <= 0 || rest <= 0) continue;
        double* A11 = d.A + j0 + (size_t)j0 * d.lda;
        double* A12 = d.A + j0 + (size_t)jend * d.lda;
        double* A21 = d.A + jend + (size_t)j0 * d.lda;
        double* A22 = d.A + jend + (size_t)jend * d.lda;
        tr.push_back(hssk_trsm_desc{A11, A12, nb, rest, d.lda, d.lda, 1, 0, 1});
        gm.push_back(hssk_gemm_desc{A21, A12, A22, rest, rest, nb, d.lda, d.lda, d.lda, 0, 0, -1.0, 1.0});
      }
      if (!tr.empty() && hssk_trsm_vbatched(ctx, tr.data(), (int)tr.size())) return 1;
      if (!gm.empty() && hssk_gemm_vbatched(ctx, gm.data(), (int)gm.size())) return 1;
    }
  }
  hssk_rt::check_launch();
  HSSK_API_END
}

int hssk_getrs_vbatched(hssk_ctx* ctx, const hssk_lusolve_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  auto* dd = (const hssk_lusolve_desc*)ctx->stage(descs, sizeof(*descs) * count);
  HSSK_LAUNCH(laswp_kernel, dim3((unsigned)count), dim3(64), 0, ctx->stream, dd);
  std::vector<hssk_trsm_desc> lo(count), up(count);
  for (int i = 0; i < count; i++) {
    lo[i] = hssk_trsm_desc{descs[i].LU, descs[i].B, descs[i].n, descs[i].nrhs, descs[i].lda, descs[i].ldb, 1, 0, 1};
    up[i] = hssk_trsm_desc{descs[i].LU, descs[i].B, descs[i].n, descs[i].nrhs, descs[i].lda, descs[i].ldb, 0, 0, 0};
  }
  if (hssk_trsm_vbatched(ctx, lo.data(), count)) return 1;
  if (hssk_trsm_vbatched(ctx, up.data(), count)) return 1;
  hssk_rt::check_launch();
  HSSK_API_END
}

}  // extern "C"
