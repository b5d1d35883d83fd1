// Sparse Johnson-Lindenstrauss sketch (the reference's --hss_compression_sketch SJLT, HSS/HSSMatrix.sketch.hpp):
// the sketching matrix R (K x dn) has nnz entries +-1 per ROW, so  S = op(A) R  costs 2 nnz flops per element of A
// instead of 2 dn -- at nnz = 4 the sketch is bound by streaming A from HBM once (8 bytes per element), not by MFMA.
//
// R is kept as a pattern  pat[q * K + k] = column | (negative ? 1 << 31 : 0)  (q < nnz), K contiguous, so that the
// lanes of a wave that walk k read it coalesced.  Samples use the engine's transposed layout: St is dn x n_out.
//
//   sjlt_t_kernel  (Sc = A^T R):  St(:, j) = sum_k A(k, j) R(k, :).  Lanes walk k down CT columns of A (coalesced, every
//     byte of A read once); a lane's row pattern is loaded once and reused for the CT columns; the sums land in an
//     LDS tile acc[CT][dn] through ds_add_f64 (lanes of one wave hit different columns c of the same row j).
//   sjlt_n_kernel  (Sr = A R):    St(:, i) = sum_k A(i, k) R(k, :).  Lanes walk RT consecutive rows i of 64 / RT
//     columns k at a time (RT * 8 byte segments of a column of A); the pattern of a column is uniform over its lanes;
//     acc[dn][RT + 1] in LDS, again ds_add_f64 because the waves of the workgroup take different k.
// Both kernels: one pass over their part of A, no re-reads; LDS work is nnz atomic adds per 8 bytes loaded.
#include "hssk_device.h"
#include "hssk_internal.h"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>

namespace {

constexpr int SJ_NNZ_MAX = 8;
constexpr int SJ_T = 256;     // threads per workgroup
constexpr int SJ_U = 8;       // independent loads in flight per lane

// dense form of the pattern, Rt (dn x K, ld): column k gets its nnz entries, the rest zeros (SJLT_to_dense)
__global__ void sjlt_dense_kernel(double* __restrict__ Rt, int dn, long long K, long long ld, const int* __restrict__ pat, int nnz) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long long k = (long long)blockIdx.x * 4 + wave; k < K; k += (long long)gridDim.x * 4) {
    double* col = Rt + k * ld;
    for (int r = lane; r < dn; r += 64) {
      double v = 0.;
      for (int q = 0; q < nnz; q++) {
        const int p = pat[(long long)q * K + k];
        if ((p & 0x7fffffff) == r) v = p < 0 ? -1. : 1.;
      }
      col[r] = v;
    }
  }
}

// St(0:dn, j) = sum_k A(k, j) R(k, :) for CT columns j per workgroup
template <int CT>
__global__ void __launch_bounds__(SJ_T) sjlt_t_kernel(const double* __restrict__ A, long long lda, long long K, long long n_out,
                                                       const int* __restrict__ pat, int nnz, int dn, double* __restrict__ St,
                                                       long long lds) {
  HSSK_DYN_SHARED(double, acc);   // CT x dn
  const long long j0 = (long long)blockIdx.x * CT;
  const int nj = (int)std::min<long long>(CT, n_out - j0);
  for (int i = threadIdx.x; i < CT * dn; i += SJ_T) acc[i] = 0.;
  __syncthreads();
  for (long long kb = (long long)threadIdx.x; kb < K; kb += SJ_T) {
    int p[SJ_NNZ_MAX];
#pragma unroll
    for (int q = 0; q < SJ_NNZ_MAX; q++) p[q] = q < nnz ? pat[(long long)q * K + kb] : 0;
    const double* a = A + kb + j0 * lda;
    for (int jb = 0; jb < nj; jb += SJ_U) {
      double v[SJ_U];
#pragma unroll
      for (int u = 0; u < SJ_U; u++) v[u] = jb + u < nj ? a[(long long)(jb + u) * lda] : 0.;
#pragma unroll
      for (int u = 0; u < SJ_U; u++) {
        if (jb + u < nj) {
          double* row = acc + (jb + u) * dn;
#pragma unroll
          for (int q = 0; q < SJ_NNZ_MAX; q++)
            if (q < nnz) hssk_lds_add(row + (p[q] & 0x7fffffff), p[q] < 0 ? -v[u] : v[u]);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nj * dn; i += SJ_T) {
    const int jj = i / dn, c = i - jj * dn;
    St[c + (j0 + jj) * lds] = acc[i];
  }
}

// St(0:dn, i) = sum_k A(i, k) R(k, :) for RT rows i per workgroup; a wave covers G = 64 / RT columns k per load
template <int RT>
__global__ void __launch_bounds__(SJ_T) sjlt_n_kernel(const double* __restrict__ A, long long lda, long long K, long long n_out,
                                                       const int* __restrict__ pat, int nnz, int dn, double* __restrict__ St,
                                                       long long lds) {
  HSSK_DYN_SHARED(double, acc);   // dn x (RT + 1)
  constexpr int G = SJ_T / RT;    // columns k covered by one workgroup-wide load
  constexpr int LD = RT + 1;
  const long long i0 = (long long)blockIdx.x * RT;
  const int r = threadIdx.x % RT, g = threadIdx.x / RT;
  const bool rok = i0 + r < n_out;
  for (int i = threadIdx.x; i < dn * LD; i += SJ_T) acc[i] = 0.;
  __syncthreads();
  const double* a = A + i0 + r;
  for (long long k0 = g; k0 < K; k0 += (long long)G * SJ_U) {
    double v[SJ_U];
#pragma unroll
    for (int u = 0; u < SJ_U; u++) {
      const long long k = k0 + (long long)u * G;
      v[u] = (rok && k < K) ? a[k * lda] : 0.;
    }
#pragma unroll
    for (int u = 0; u < SJ_U; u++) {
      const long long k = k0 + (long long)u * G;
      if (k < K) {
#pragma unroll
        for (int q = 0; q < SJ_NNZ_MAX; q++)
          if (q < nnz) {
            const int p = pat[(long long)q * K + k];
            hssk_lds_add(acc + (p & 0x7fffffff) * LD + r, p < 0 ? -v[u] : v[u]);
          }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < RT * dn; i += SJ_T) {
    const int rr = i / dn, c = i - rr * dn;
    if (i0 + rr < n_out) St[c + (i0 + rr) * lds] = acc[c * LD + rr];
  }
}

constexpr int SJ_LDS_DOUBLES = 8000;   // < 64 KB (the default dynamic LDS limit): two workgroups per CU

}  // namespace

int hssk_sjlt_dense(hssk_ctx* ctx, double* Rt, int dn, long long K, long long ld, const int* pat, int nnz) {
  HSSK_API_BEGIN
  if (dn <= 0 || K <= 0) return 0;
  if (nnz < 0 || nnz > SJ_NNZ_MAX) throw std::invalid_argument("hssk_sjlt_dense: nnz out of range (max 8)");
  unsigned blocks = (unsigned)std::min<long long>((K + 3) / 4, 16384);
  HSSK_LAUNCH(sjlt_dense_kernel, dim3(blocks), dim3(256), 0, ctx->stream, Rt, dn, K, ld, pat, nnz);
  hssk_rt::check_launch();
  HSSK_API_END
}

int hssk_sjlt_sketch(hssk_ctx* ctx, int transA, long long n_out, long long K, const double* A, long long lda, const int* pat,
                     int nnz, int dn, double* St, long long lds) {
  HSSK_API_BEGIN
  if (n_out <= 0 || dn <= 0) return 0;
  if (nnz < 0 || nnz > SJ_NNZ_MAX) throw std::invalid_argument("hssk_sjlt_sketch: nnz out of range (max 8)");
  if (dn > 1024) throw std::invalid_argument("hssk_sjlt_sketch: more than 1024 sketch columns in one call");
  // tuning overrides (tools/sjlt_only.py): columns per workgroup of the A^T R kernel, rows per workgroup of the A R kernel
  auto env_int = [](const char* name) { const char* v = std::getenv(name); return v ? std::atoi(v) : 0; };
  const int ct_req = env_int("HSSK_SJLT_CT"), rt_req = env_int("HSSK_SJLT_RT");
  hssk_rt::event_record(ctx->ev0, ctx->stream);
  if (transA) {
#define SJ_LAUNCH_T(CT)                                                                                                   \
  HSSK_LAUNCH(sjlt_t_kernel<CT>, dim3((unsigned)((n_out + CT - 1) / CT)), dim3(SJ_T), sizeof(double) * CT * dn, ctx->stream, \
              A, lda, K, n_out, pat, nnz, dn, St, lds)
    int ct = 16 * dn <= SJ_LDS_DOUBLES ? 16 : (8 * dn <= SJ_LDS_DOUBLES ? 8 : 4);
    if ((ct_req == 4 || ct_req == 8 || ct_req == 16 || ct_req == 32) && ct_req * dn <= SJ_LDS_DOUBLES) ct = ct_req;
    if (ct == 32) SJ_LAUNCH_T(32);
    else if (ct == 16) SJ_LAUNCH_T(16);
    else if (ct == 8) SJ_LAUNCH_T(8);
    else SJ_LAUNCH_T(4);
#undef SJ_LAUNCH_T
  } else {
#define SJ_LAUNCH_N(RT)                                                                                                   \
  HSSK_LAUNCH(sjlt_n_kernel<RT>, dim3((unsigned)((n_out + RT - 1) / RT)), dim3(SJ_T), sizeof(double) * (RT + 1) * dn,      \
              ctx->stream, A, lda, K, n_out, pat, nnz, dn, St, lds)
    int rt = 65 * dn <= SJ_LDS_DOUBLES ? 64 : (33 * dn <= SJ_LDS_DOUBLES ? 32 : (17 * dn <= SJ_LDS_DOUBLES ? 16 : (9 * dn <= SJ_LDS_DOUBLES ? 8 : 4)));
    if ((rt_req == 4 || rt_req == 8 || rt_req == 16 || rt_req == 32 || rt_req == 64) && (rt_req + 1) * dn <= SJ_LDS_DOUBLES) rt = rt_req;
    if (rt == 64) SJ_LAUNCH_N(64);
    else if (rt == 32) SJ_LAUNCH_N(32);
    else if (rt == 16) SJ_LAUNCH_N(16);
    else if (rt == 8) SJ_LAUNCH_N(8);
    else SJ_LAUNCH_N(4);
#undef SJ_LAUNCH_N
  }
  hssk_rt::event_record(ctx->ev1, ctx->stream);
  ctx->dgemm_timed_flops = 2.0 * (double)n_out * (double)K * nnz;
  hssk_rt::check_launch();
  HSSK_API_END
}
