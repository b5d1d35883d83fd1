// hssk_blr_sweep: the block forward / backward substitution with the factors of a (partially) factored BLR front
// (BLRMatrix::solve, BLR/BLRMatrix.hpp:118-122; the fronts' solve phases FrontBLR::fwd_solve_phase2 / bwd_solve_phase1,
// sparse/fronts/FrontBLR.cpp:525-570 -> trsmLNU_gemm / gemm_trsmUNN, BLR/BLRMatrix.cpp:1552-1665) for ONE right-hand side
// as ONE launch.
//
// Step by step -- x_i <- L_ii^{-1} P_i x_i, then x_k -= U_ki (V_ki^T x_i) for every block row k below -- the substitution is
// three launches per block step (row interchange + triangular solve, the V^T x products, the U t products) on operands of
// a few hundred numbers: 48 us per step of pure launch latency, 27 ms for the two sweeps over the 256 block steps of the
// 200 x 200 root front of the 200^3 problem.  Here a block ROW is a workgroup: it gathers the contributions of the block rows
// it depends on -- only tiles of non-zero rank are dependencies -- as they become final, solves with its diagonal tile and
// publishes its piece.  Workgroups are ordered so that a row depends on lower-indexed ones only (the in-order dispatch
// argument of hssk_sweep.hip); finished pieces are written with device-coherent stores and announced by a flag.
//   forward:  row k, terms (k, i), i < k ascending:   acc -= U_ki (V_ki^T x_i);  then x_k = L_kk^{-1} P_k acc   (k < steps)
//   backward: row i, terms (i, j), j > i:             acc -= U_ij (V_ij^T x_j);  then x_i = U_ii^{-1} acc
// The triangular solves run on 64-row blocks with the inverted diagonal blocks of hssk_trtri_diag_vbatched (modes 2 / 1).
#include "hssk_device.h"
#include "hssk_internal.h"

namespace {

constexpr int BS_T = 256;
constexpr int BS_MMAX = 512;          // rows of a tile
constexpr long BS_SPIN_LIMIT = 1L << 23;

__global__ __launch_bounds__(BS_T) void blr_sweep_kernel(const hssk_blr_row* __restrict__ rows, const hssk_blr_term* __restrict__ terms,
                                                        double* __restrict__ X, int* __restrict__ flags, int* __restrict__ err) {
  HSSK_SHARED double s_acc[BS_MMAX];
  HSSK_SHARED double s_x[BS_MMAX];
  HSSK_SHARED double s_t[BS_MMAX];
  HSSK_SHARED double s_y[64];     // the block being solved (zero beyond its rows)
  HSSK_SHARED double s_v[64];
  HSSK_SHARED int s_piv[BS_MMAX];
  HSSK_SHARED int s_ok;
  const hssk_blr_row rw = rows[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = rw.m;
  for (int i = tid; i < m; i += BS_T) s_acc[i] = hssk_gload(X, (size_t)rw.off + i);
  if (tid == 0) s_ok = 1;
  __syncthreads();
  for (int q = 0; q < rw.nterms; q++) {
    const hssk_blr_term tm = terms[rw.first_term + q];
    if (tm.src_flag >= 0) {
      if (tid == 0) {
        long spins = 0;
        while (hssk_flag_load(flags + tm.src_flag) == 0) {
          hssk_pause();
          if (++spins > BS_SPIN_LIMIT) { hssk_flag_raise(err); s_ok = 0; break; }
        }
      }
      __syncthreads();
      for (int i = tid; i < tm.n; i += BS_T) s_x[i] = hssk_cload(X, (size_t)tm.src_off + i);
    } else {
      for (int i = tid; i < tm.n; i += BS_T) s_x[i] = hssk_gload(X, (size_t)tm.src_off + i);
    }
    __syncthreads();
    // t = V^T x: a wave per column of V, the lane's (up to eight) elements loaded together from clamped addresses
    for (int j = wave; j < tm.r; j += BS_T / 64) {
      const double* v = tm.V + (size_t)j * tm.n;
      double e[BS_MMAX / 64];
#pragma unroll
      for (int u = 0; u < BS_MMAX / 64; u++) e[u] = hssk_gload(v, (size_t)min(lane + 64 * u, tm.n - 1));
      double s = 0.;
#pragma unroll
      for (int u = 0; u < BS_MMAX / 64; u++) s += lane + 64 * u < tm.n ? e[u] * s_x[lane + 64 * u] : 0.;
      s = hssk_wave_sum(s);
      if (lane == 0) s_t[j] = s;
    }
    __syncthreads();
    // acc -= U t: a thread per row, eight columns of U in flight
    for (int i0 = 0; i0 < m; i0 += BS_T) {
      const int i = min(i0 + tid, m - 1);
      double s = 0.;
      for (int j0 = 0; j0 < tm.r; j0 += 8) {
        double e[8];
#pragma unroll
        for (int u = 0; u < 8; u++) e[u] = hssk_gload(tm.U, (size_t)i + (size_t)min(j0 + u, tm.r - 1) * m);
#pragma unroll
        for (int u = 0; u < 8; u++) s += j0 + u < tm.r ? e[u] * s_t[j0 + u] : 0.;
      }
      if (i0 + tid < m) s_acc[i0 + tid] -= s;
    }
    __syncthreads();
  }
  if (rw.LU) {
    const int nb = (m + 63) / 64;
    // y = Tinv_b v: the inverted diagonal block is zero-padded to 64 x 64 -- a fixed-length product whose loads the compiler
    // keeps in flight; rest -= T(rest, b) y likewise with the columns beyond the block clamped and their factors zero
    auto block_solve = [&](int b) {
      const int b0 = 64 * b, bn = min(64, m - b0);
      if (tid < 64) s_v[tid] = tid < bn ? s_acc[b0 + tid] : 0.;
      __syncthreads();
      if (tid < 64) {
        const double* Ti = rw.Tinv + (size_t)b * 4096;
        double s = 0.;
#pragma unroll 16
        for (int c = 0; c < 64; c++) s += hssk_gload(Ti, (size_t)tid + (size_t)c * 64) * s_v[c];
        s_y[tid] = tid < bn ? s : 0.;
      }
      __syncthreads();
      if (tid < bn) s_acc[b0 + tid] = s_y[tid];
      return bn;
    };
    auto rest_update = [&](int b, int bn, int lo, int hi) {   // rows [lo, hi) -= T(rows, block b) y
      const int b0 = 64 * b;
      for (int i0 = lo; i0 < hi; i0 += BS_T) {
        const int i = min(i0 + tid, hi - 1);
        double s = 0.;
#pragma unroll 16
        for (int c = 0; c < 64; c++) s += hssk_gload(rw.LU, (size_t)i + (size_t)(b0 + min(c, bn - 1)) * rw.lda) * s_y[c];
        if (i0 + tid < hi) s_acc[i0 + tid] -= s;
      }
      __syncthreads();
    };
    if (rw.mode == 0) {
      // x <- P x (the interchanges of getrf, in order; the pivots staged in the LDS first: read one by one from global memory
      // they were m dependent round trips), then the unit lower triangle, block rows downwards
      for (int k = tid; k < m; k += BS_T) s_piv[k] = rw.piv[k];
      __syncthreads();
      if (tid == 0)
        for (int k = 0; k < m; k++) {
          const int pk = s_piv[k];
          if (pk != k) { const double a = s_acc[k]; s_acc[k] = s_acc[pk]; s_acc[pk] = a; }
        }
      __syncthreads();
      for (int b = 0; b < nb; b++) {
        const int bn = block_solve(b);
        if (64 * b + 64 < m) rest_update(b, bn, 64 * b + 64, m);
        else __syncthreads();
      }
    } else {
      for (int b = nb - 1; b >= 0; b--) {   // the upper triangle, block rows upwards
        const int bn = block_solve(b);
        if (b > 0) rest_update(b, bn, 0, 64 * b);
        else __syncthreads();
      }
    }
  }
  for (int i = tid; i < m; i += BS_T) hssk_cstore(X, (size_t)rw.off + i, s_acc[i]);
  hssk_drain_stores();
  __syncthreads();
  if (tid == 0) hssk_flag_store(flags + blockIdx.x, s_ok ? 1 : 2);
}

}  // namespace

extern "C" int hssk_blr_sweep_check(const hssk_blr_row* rows, int nrows, const hssk_blr_term* terms, int nterms) {
  HSSK_API_BEGIN
  for (int i = 0; i < nrows; i++) {
    if (rows[i].m > BS_MMAX) HSSK_UNSUPPORTED("tile beyond 512 rows");
    if (rows[i].LU && !rows[i].Tinv) HSSK_UNSUPPORTED("diagonal tile without its inverted 64 x 64 blocks");
    if (rows[i].first_term < 0 || rows[i].nterms < 0 || rows[i].first_term + rows[i].nterms > nterms) throw std::invalid_argument("hssk_blr_sweep: term range outside the table");
    for (int q = 0; q < rows[i].nterms; q++) {
      const hssk_blr_term& t = terms[rows[i].first_term + q];
      if (t.n > BS_MMAX || t.r > BS_MMAX) HSSK_UNSUPPORTED("tile beyond 512 rows / columns");
      if (t.src_flag >= i) throw std::invalid_argument("hssk_blr_sweep: a block row may only depend on rows in front of it");
    }
  }
  HSSK_API_END
}

extern "C" int hssk_blr_sweep_resident(hssk_ctx* ctx, const hssk_blr_row* d_rows, int nrows, const hssk_blr_term* d_terms, double* X,
                                       int* flags) {
  HSSK_API_BEGIN
  if (nrows <= 0) return 0;
  if (!ctx->h_sweep_err) { ctx->h_sweep_err = (int*)hssk_rt::pinned_malloc(64); *ctx->h_sweep_err = 0; }
  hssk_rt::memset_async(flags, 0, sizeof(int) * (size_t)nrows, ctx->stream);
  HSSK_LAUNCH(blr_sweep_kernel, dim3((unsigned)nrows), dim3(BS_T), 0, ctx->stream, d_rows, d_terms, X, flags, ctx->h_sweep_err);
  hssk_rt::check_launch();
  HSSK_API_END
}

extern "C" int hssk_blr_sweep(hssk_ctx* ctx, const hssk_blr_row* rows, int nrows, const hssk_blr_term* terms, int nterms, double* X,
                              int* flags) {
  if (nrows <= 0) return 0;
  if (const int rc = hssk_blr_sweep_check(rows, nrows, terms, nterms)) return rc;
  HSSK_API_BEGIN
  // (tables beyond the staging ring: the caller walks the block steps, as for any operand this launch does not take)
  if (sizeof(*rows) * (size_t)nrows + sizeof(*terms) * (size_t)std::max(nterms, 1) + 1024 > ctx->ring_bytes) HSSK_UNSUPPORTED("descriptor tables beyond the staging ring");
  auto* dr = (const hssk_blr_row*)ctx->stage(rows, sizeof(*rows) * (size_t)nrows);
  auto* dt = (const hssk_blr_term*)ctx->stage(nterms > 0 ? terms : (const hssk_blr_term*)rows, nterms > 0 ? sizeof(*terms) * (size_t)nterms : 8);
  return hssk_blr_sweep_resident(ctx, dr, nrows, dt, X, flags);
  HSSK_API_END
}
