// Fused ULV solve sweeps for few right-hand sides: ONE launch per tree level instead of seven (forward) / three
// (backward) batched BLAS-like launches.  The sweeps of a solve are ~100 dependent small launches over ~10 levels and
// are bound by launch-to-launch latency, not by the 0.5 GB of factors they read; here one workgroup does everything a
// node needs at its level with the vectors in LDS and the node's blocks streamed once.
//
// Reference: HSSMatrix::solve_fwd / solve_bwd (HSS/HSSMatrix.solve.hpp:69-238).  Forward, per non-root node with
// m rows and rank r (bases as X = R11^{-1} R12 + permutation, LQ of W0 stored as QR of W0^T: R~ in the upper triangle
// of Rlq, explicit Q~ = Q^T):
//   f   = rhs rows of the node (leaf) or [ft1_0; ft1_1] - [B01 z_1; B10 z_0] (inner)            solve.hpp:88-99
//   ft1 = f(perm[0:r]),  y = f(perm[r:]) - X^T ft1,  y <- R~^{-T} y                               :153-163
//   ft1 -= W1 (Q~(:, 0:m-r) y)                                                                     :100-131
//   z   = V^H [z_0; z_1] + Vt0^T y   (leaf: Vt0^T y)                                              :164-192
// Backward, per child c of a node:  x_c = Q~_c(:, 0:mc-rc) y_c + Q~_c(:, mc-rc:) x(part of the parent)   :199-238
#include "hssk_device.h"
#include "hssk_internal.h"

#include <algorithm>
#include <vector>

namespace {

constexpr int UL_T = 256;      // threads per workgroup
constexpr int UL_NR = 4;       // right-hand sides handled at once
constexpr int UL_TB = 32;      // block size of the substitution y <- R~^{-T} y
constexpr int UL_MAX = 256;    // largest node dimension (rows of a basis) the LDS vectors / register prefetch are sized for

__global__ __launch_bounds__(UL_T) void ulv_fwd_kernel(const hssk_ulv_fwd_desc* __restrict__ descs, int nrhs) {
  HSSK_SHARED double s_f[UL_MAX * UL_NR];    // f, later t = Q~(:, :m-r) y
  HSSK_SHARED double s_y[UL_MAX * UL_NR];
  HSSK_SHARED double s_a[UL_MAX * UL_NR];    // ft1, later the stacked children z
  HSSK_SHARED double s_p[4 * UL_NR * 64];    // per-wave partial sums of the split GEMVs
  HSSK_SHARED double s_T[UL_TB * UL_TB];      // diagonal block of R~ of the blocked substitution
  const hssk_ulv_fwd_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = p.m, r = p.r, q = m - r;
  // ---- f
  for (int e = tid; e < m * nrhs; e += UL_T) {
    const int i = e % m, c = e / m;
    s_f[i + c * UL_MAX] = hssk_gload(p.fsrc, i + (size_t)c * p.ldf);
  }
  __syncthreads();
  if (p.B01) {   // inner node: f(0:rU0) -= B01 zc(rV0:), f(rU0:) -= B10 zc(0:rV0)
    for (int i = tid; i < m; i += UL_T) {
      const bool top = i < p.rU0;
      const double* B = top ? p.B01 : p.B10;
      const int ii = top ? i : i - p.rU0, ldb = max(top ? p.rU0 : p.rU1, 1), kn = top ? p.rV1 : p.rV0, zo = top ? p.rV0 : 0;
      double acc[UL_NR] = {0., 0., 0., 0.};
#pragma unroll 8
      for (int k = 0; k < kn; k++) {
        const double b = hssk_gload(B, ii + (size_t)k * ldb);
        for (int c = 0; c < nrhs; c++) acc[c] += b * hssk_gload(p.zc, zo + k + (size_t)c * p.ldz_in);
      }
      for (int c = 0; c < nrhs; c++) s_f[i + c * UL_MAX] -= acc[c];
    }
    __syncthreads();
  }
  // ---- ft1 = f(perm[0:r]), y0 = f(perm[r:])
  for (int e = tid; e < m * nrhs; e += UL_T) {
    const int k = e % m, c = e / m;
    const double v = s_f[p.permU[k] + c * UL_MAX];
    if (k < r) s_a[k + c * UL_MAX] = v;
    else s_y[(k - r) + c * UL_MAX] = v;
  }
  __syncthreads();
  // ---- y -= X^T ft1   (X is r x q, column k contiguous): one wave per output row
  if (r > 0 && q > 0)
    for (int k = wave; k < q; k += 4) {
      double acc[UL_NR] = {0., 0., 0., 0.};
      for (int j = lane; j < r; j += 64) {
        const double x = hssk_gload(p.XU, j + (size_t)k * r);
        for (int c = 0; c < nrhs; c++) acc[c] += x * s_a[j + c * UL_MAX];
      }
      for (int c = 0; c < nrhs; c++) {
        const double v = hssk_wave_sum(acc[c]);
        if (lane == 0) s_y[k + c * UL_MAX] -= v;
      }
    }
  __syncthreads();
  // ---- y <- R~^{-T} y, blocked: the dependent chain only ever touches a UL_TB x UL_TB diagonal block that all threads
  // loaded into LDS at once; the coupling to the remaining rows is a parallel update with UL_TB independent loads per
  // thread.  (An unblocked chain pays a global-memory round trip per step: it was 40 % of the sweep.)
  if (q > 0) {
    for (int b0 = 0; b0 < q; b0 += UL_TB) {
      const int nb = min(UL_TB, q - b0);
      for (int e = tid; e < nb * nb; e += UL_T) {
        const int i = e % nb, j = e / nb;
        s_T[i + j * UL_TB] = i <= j ? hssk_gload(p.Rlq, (b0 + i) + (size_t)(b0 + j) * p.m) : 0.;
      }
      __syncthreads();
      if (wave == 0)
        for (int i = 0; i < nb; i++) {
          double acc[UL_NR] = {0., 0., 0., 0.};
          if (lane < i) {
            const double t = s_T[lane + i * UL_TB];
            for (int c = 0; c < nrhs; c++) acc[c] = t * s_y[b0 + lane + c * UL_MAX];
          }
          const double dia = s_T[i + i * UL_TB];
          for (int c = 0; c < nrhs; c++) {
            const double yi = s_y[b0 + i + c * UL_MAX];   // read by every lane before it is rewritten
            const double v = hssk_wave_sum(acc[c]);
            s_y[b0 + i + c * UL_MAX] = (yi - v) / dia;    // every lane stores the same value: no lane can run ahead
          }
        }
      __syncthreads();
      // rows below the block: y(k) -= R~(b0:b0+nb, k)^T y(b0:b0+nb)   (column k of R~, rows b0.. contiguous)
      for (int k = b0 + nb + tid; k < q; k += UL_T) {
        double acc[UL_NR] = {0., 0., 0., 0.};
        const size_t cb = (size_t)k * p.m + b0;
#pragma unroll 8
        for (int l = 0; l < nb; l++) {
          const double t = hssk_gload(p.Rlq, cb + l);
          for (int c = 0; c < nrhs; c++) acc[c] += t * s_y[b0 + l + c * UL_MAX];
        }
        for (int c = 0; c < nrhs; c++) s_y[k + c * UL_MAX] -= acc[c];
      }
      __syncthreads();
    }
    for (int e = tid; e < q * nrhs; e += UL_T) hssk_gstore(p.y, (e % q) + (size_t)(e / q) * q, s_y[(e % q) + (e / q) * UL_MAX]);
  }
  // ---- ft1 -= W1 (Q~(:, 0:q) y):  t (m) = Q~(:, :q) y  (row index contiguous),  then W1 (r x m, row index contiguous)
  if (r > 0) {
    if (q > 0) {
      // t = Q~(:, 0:q) y : lanes along the rows (contiguous), the four waves split the columns -> 4 x more loads in
      // flight; partial sums meet in LDS (s_p), then W1 t the same way
      for (int i0 = 0; i0 < m; i0 += 64) {
        const int i = i0 + lane;
        double acc[UL_NR] = {0., 0., 0., 0.};
        if (i < m) {
#pragma unroll 8
          for (int j = wave; j < q; j += 4) {
            const double t = hssk_gload(p.Qt, i + (size_t)j * m);
            for (int c = 0; c < nrhs; c++) acc[c] += t * s_y[j + c * UL_MAX];
          }
        }
        for (int c = 0; c < nrhs; c++) s_p[(wave * UL_NR + c) * 64 + lane] = acc[c];
        __syncthreads();
        if (wave == 0 && i < m)
          for (int c = 0; c < nrhs; c++)
            s_f[i + c * UL_MAX] = s_p[c * 64 + lane] + s_p[(UL_NR + c) * 64 + lane] + s_p[(2 * UL_NR + c) * 64 + lane] + s_p[(3 * UL_NR + c) * 64 + lane];
        __syncthreads();
      }
      for (int k0 = 0; k0 < r; k0 += 64) {
        const int k = k0 + lane;
        double acc[UL_NR] = {0., 0., 0., 0.};
        if (k < r) {
#pragma unroll 8
          for (int j = wave; j < m; j += 4) {
            const double w = hssk_gload(p.W1, k + (size_t)j * r);
            for (int c = 0; c < nrhs; c++) acc[c] += w * s_f[j + c * UL_MAX];
          }
        }
        for (int c = 0; c < nrhs; c++) s_p[(wave * UL_NR + c) * 64 + lane] = acc[c];
        __syncthreads();
        if (wave == 0 && k < r)
          for (int c = 0; c < nrhs; c++)
            s_a[k + c * UL_MAX] -= s_p[c * 64 + lane] + s_p[(UL_NR + c) * 64 + lane] + s_p[(2 * UL_NR + c) * 64 + lane] + s_p[(3 * UL_NR + c) * 64 + lane];
        __syncthreads();
      }
    }
    for (int e = tid; e < r * nrhs; e += UL_T) hssk_gstore(p.ft1, (e % r) + (size_t)(e / r) * p.ldp, s_a[(e % r) + (e / r) * UL_MAX]);
  }
  __syncthreads();
  // ---- z = V^H [z0; z1] + Vt0^T y
  const int rv = p.rv, mv = p.mv;
  if (rv > 0) {
    if (p.permV) {   // inner: stacked children z in zc (mv rows)
      for (int e = tid; e < mv * nrhs; e += UL_T) s_a[(e % mv) + (e / mv) * UL_MAX] = hssk_gload(p.zc, (e % mv) + (size_t)(e / mv) * p.ldz_in);
      __syncthreads();
    }
    for (int k = wave; k < rv; k += 4) {
      double acc[UL_NR] = {0., 0., 0., 0.};
      if (q > 0)   // Vt0 is q x rv, column k contiguous
        for (int j = lane; j < q; j += 64) {
          const double v = hssk_gload(p.Vt0, j + (size_t)k * q);
          for (int c = 0; c < nrhs; c++) acc[c] += v * s_y[j + c * UL_MAX];
        }
      if (p.permV)   // + zc(permV[k]) + sum_j XV(k, j) zc(permV[rv + j])   (XV is rv x (mv - rv))
        for (int j = lane; j < mv - rv; j += 64) {
          const double x = hssk_gload(p.XV, k + (size_t)j * rv);
          const int src = p.permV[rv + j];
          for (int c = 0; c < nrhs; c++) acc[c] += x * s_a[src + c * UL_MAX];
        }
      for (int c = 0; c < nrhs; c++) {
        double v = hssk_wave_sum(acc[c]);
        if (p.permV) v += s_a[p.permV[k] + c * UL_MAX];
        if (lane == 0) hssk_gstore(p.z, k + (size_t)c * p.ldz, v);
      }
    }
  }
}

// x_c = Q~(:, 0:q) y + Q~(:, q:) xpart ; mc == rc (nothing eliminated): x_c = xpart
__global__ __launch_bounds__(UL_T) void ulv_bwd_kernel(const hssk_ulv_bwd_desc* __restrict__ descs, int nrhs) {
  HSSK_SHARED double s_v[UL_MAX * UL_NR];   // [y; xpart]
  const hssk_ulv_bwd_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x;
  const int m = p.m, r = p.r, q = m - r;
  for (int e = tid; e < m * nrhs; e += UL_T) {
    const int i = e % m, c = e / m;
    s_v[i + c * UL_MAX] = i < q ? hssk_gload(p.y, i + (size_t)c * q) : hssk_gload(p.xpart, (i - q) + (size_t)c * p.ldx);
  }
  __syncthreads();
  for (int i = tid; i < m; i += UL_T) {
    double acc[UL_NR] = {0., 0., 0., 0.};
    if (q > 0) {
#pragma unroll 8
      for (int j = 0; j < m; j++) {
        const double t = hssk_gload(p.Qt, i + (size_t)j * m);
        for (int c = 0; c < nrhs; c++) acc[c] += t * s_v[j + c * UL_MAX];
      }
    } else {
      for (int c = 0; c < nrhs; c++) acc[c] = s_v[i + c * UL_MAX];
    }
    for (int c = 0; c < nrhs; c++) hssk_gstore(p.out, i + (size_t)c * p.ldo, acc[c]);
  }
}

}  // namespace

extern "C" int hssk_ulv_fwd_level(hssk_ctx* ctx, const hssk_ulv_fwd_desc* descs, int count, int nrhs) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  if (nrhs < 1 || nrhs > UL_NR) return 2;
  for (int i = 0; i < count; i++)
    if (descs[i].m > UL_MAX || descs[i].mv > UL_MAX || descs[i].m < 0) return 2;   // caller uses the unfused path
  auto* dd = (const hssk_ulv_fwd_desc*)ctx->stage(descs, sizeof(*descs) * count);
  HSSK_LAUNCH(ulv_fwd_kernel, dim3((unsigned)count), dim3(UL_T), 0, ctx->stream, dd, nrhs);
  hssk_rt::check_launch();
  HSSK_API_END
}

extern "C" int hssk_ulv_bwd_level(hssk_ctx* ctx, const hssk_ulv_bwd_desc* descs, int count, int nrhs) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  if (nrhs < 1 || nrhs > UL_NR) return 2;
  for (int i = 0; i < count; i++)
    if (descs[i].m > UL_MAX) return 2;
  auto* dd = (const hssk_ulv_bwd_desc*)ctx->stage(descs, sizeof(*descs) * count);
  HSSK_LAUNCH(ulv_bwd_kernel, dim3((unsigned)count), dim3(UL_T), 0, ctx->stream, dd, nrhs);
  hssk_rt::check_launch();
  HSSK_API_END
}
