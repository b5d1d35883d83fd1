// Sparse Johnson-Lindenstrauss sketch (the reference's --hss_compression_sketch SJLT, HSS/HSSMatrix.sketch.hpp):
// the sketching matrix R (K x dn) has nnz entries +-1 per ROW, so  S = op(A) R  costs 2 nnz flops per element of A
// instead of 2 dn -- at nnz = 4 the sketch is bound by streaming A from HBM once (8 bytes per element), not by MFMA.
//
// R is kept as a pattern of NQ = 4 (nnz <= 4) or 8 ints per row, pat[k * NQ + q] = column | (negative ? 1 << 31 : 0),
// unused entries = column dn (a scratch row of the accumulators).  Samples use the engine's transposed layout:
// St is dn x n_out.  Every kernel reads its part of A exactly once and sums into an LDS tile with ds_add_f64.
//
//   sjlt_n_kernel  (Sr = A R):   St(:, i) = sum_k A(i, k) R(k, :).  A workgroup owns 64 consecutive rows i; each of
//     its NW = 16 waves streams its own columns k (512-byte coalesced loads, U = 16 in flight per lane).  k is uniform over the
//     wave, so the pattern arrives through scalar loads and the NQ updates of acc[c][lane] (row stride 65 doubles) are
//     conflict-free.  The waves of a workgroup take different k, hence the atomic add.
//   sjlt_t_kernel  (Sc = A^T R): St(:, j) = sum_k A(k, j) R(k, :).  A workgroup owns 64 consecutive columns j, lane = j:
//     every lane loads 128 contiguous bytes (16 k) of its own column -- full cache lines, and the NW = 8 waves of the
//     workgroup take adjacent 16-row blocks so a column is read in 1 KB runs.  Again k is wave-uniform:
//     scalar pattern loads, conflict-free acc[c][lane].
//   sjlt_n_small_kernel / sjlt_t_small_kernel: the same sums with accumulator tiles that fit 64 KB (RT rows / CT
//     columns per workgroup) for sketch blocks too wide for the kernels above (dn > 300).
#include "hssk_device.h"
#include "hssk_internal.h"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>

namespace {

constexpr int SJ_T = 256;     // threads per workgroup of the small kernels
constexpr int SJ_U = 8;       // independent loads in flight per lane (small kernels)
constexpr int SJ_LD = 65;     // accumulator row stride of the 64-wide tiles (doubles): conflict-free transposed read

// MODE (diagnostics, tools/sjlt_only.py): 0 = product path, 1 = plain read-modify-write instead of ds_add_f64 (races:
// wrong sums, same instruction count), 2 = no LDS updates (streaming rate of the access pattern alone)
template <int MODE> __device__ __forceinline__ void sj_acc(double* p, double v, double& sink) {
  if (MODE == 0) hssk_lds_add(p, v);
  else if (MODE == 1) *p += v;
  else sink += v;
}

// dense form of the pattern, Rt (dn x K, ld): column k gets its entries, the rest zeros (SJLT_to_dense)
__global__ void sjlt_dense_kernel(double* __restrict__ Rt, int dn, long long K, long long ld, const int* __restrict__ pat, int nq) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long long k = (long long)blockIdx.x * 4 + wave; k < K; k += (long long)gridDim.x * 4) {
    double* col = Rt + k * ld;
    for (int r = lane; r < dn; r += 64) {
      double v = 0.;
      for (int q = 0; q < nq; q++) {
        const int p = pat[k * nq + q];
        if ((p & 0x7fffffff) == r) v = p < 0 ? -1. : 1.;
      }
      col[r] = v;
    }
  }
}

// ---- Sr = A R, 64 rows per workgroup, NW waves ---------------------------------------------------------------------
template <int NQ, int NW, int U, int MODE>
__global__ void __launch_bounds__(NW * 64) sjlt_n_kernel(const double* __restrict__ A, long long lda, long long K, long long n_out,
                                                          const int* __restrict__ pat, int dn, double* __restrict__ St, long long lds) {
  HSSK_DYN_SHARED(double, acc);   // (dn + 1) x SJ_LD
  const int lane = threadIdx.x & 63, wave = hssk_uniform(threadIdx.x >> 6);
  const long long i0 = (long long)blockIdx.x * 64;
  const bool rok = i0 + lane < n_out;
  for (int i = threadIdx.x; i < (dn + 1) * SJ_LD; i += NW * 64) acc[i] = 0.;
  __syncthreads();
  const double* a = A + i0 + (rok ? lane : 0);   // (rows past the end read row i0 and are not stored)
  double* my = acc + lane;
  double sink = 0.;
  // main loop: this wave takes columns k0 + u NW (u < U); the next block's loads are issued before the current block's
  // LDS updates, the patterns of the U columns come through scalar loads
  const long long step = (long long)NW * U, kfull = K - (long long)(U - 1) * NW;
  long long k0 = wave;
  double v[U], w[U];
  auto fetch = [&](double (&x)[U], long long kb) {
#pragma unroll
    for (int u = 0; u < U; u++) x[u] = a[(kb + (long long)u * NW) * lda];
  };
  auto update = [&](const double (&x)[U], long long kb) {
    int p[U][NQ];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int q = 0; q < NQ; q++) p[u][q] = pat[(kb + (long long)u * NW) * NQ + q];
    HSSK_COMPILER_FENCE();   // all pattern loads are issued before the first LDS update
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int q = 0; q < NQ; q++) sj_acc<MODE>(my + (p[u][q] & 0x7fffffff) * SJ_LD, p[u][q] < 0 ? -x[u] : x[u], sink);
  };
  if (k0 < kfull) {
    fetch(v, k0);
    for (;;) {   // two blocks per trip so that the prefetched registers are used in place
      long long kn = k0 + step;
      bool hn = kn < kfull;
      fetch(w, hn ? kn : k0);   // (unconditional: the wait counts in update() stay exact; the last one re-reads k0)
      update(v, k0);
      k0 = kn;
      if (!hn) break;
      kn = k0 + step;
      hn = kn < kfull;
      fetch(v, hn ? kn : k0);
      update(w, k0);
      k0 = kn;
      if (!hn) break;
    }
  }
  for (long long k = k0; k < K; k += NW) {   // tail
    const double x = a[k * lda];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int pq = pat[k * NQ + q];
      sj_acc<MODE>(my + (pq & 0x7fffffff) * SJ_LD, pq < 0 ? -x : x, sink);
    }
  }
  if (MODE == 2) my[0] = sink;
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * dn; i += NW * 64) {
    const int rr = i / dn, c = i - rr * dn;
    if (i0 + rr < n_out) St[c + (i0 + rr) * lds] = acc[c * SJ_LD + rr];
  }
}

// ---- Sc = A^T R, 64 columns per workgroup (lane = column), NW waves ----------------------------------------------
// KB = rows per lane and block: 8 (one 64-byte run) or 16 (one 128-byte run)
template <int NQ, int NW, int KB, int MODE>
__global__ void __launch_bounds__(NW * 64) sjlt_t_kernel(const double* __restrict__ A, long long lda, long long K, long long n_out,
                                                          const int* __restrict__ pat, int dn, double* __restrict__ St, long long lds,
                                                          int aligned) {
  HSSK_DYN_SHARED(double, acc);   // (dn + 1) x SJ_LD
  const int lane = threadIdx.x & 63, wave = hssk_uniform(threadIdx.x >> 6);
  const long long j0 = (long long)blockIdx.x * 64;
  const bool cok = j0 + lane < n_out;
  for (int i = threadIdx.x; i < (dn + 1) * SJ_LD; i += NW * 64) acc[i] = 0.;
  __syncthreads();
  const double* a = A + (j0 + (cok ? lane : 0)) * lda;   // (columns past the end read column j0 and are not stored)
  double* my = acc + lane;
  double sink = 0.;
  const long long step = (long long)NW * KB, kfull = aligned ? K - (KB - 1) : 0;   // full blocks use 16-byte loads
  long long k0 = (long long)wave * KB;
  hssk_d2 v[KB / 2], w[KB / 2];
  auto fetch = [&](hssk_d2 (&x)[KB / 2], long long kb) {
#pragma unroll
    for (int t = 0; t < KB / 2; t++) x[t] = *reinterpret_cast<const hssk_d2*>(a + kb + 2 * t);
  };
  auto update = [&](const hssk_d2 (&x)[KB / 2], long long kb) {
#pragma unroll
    for (int h = 0; h < KB / 8; h++) {   // 8 rows at a time: their patterns fit the scalar registers
      int p[8][NQ];
#pragma unroll
      for (int t = 0; t < 8; t++)
#pragma unroll
        for (int q = 0; q < NQ; q++) p[t][q] = pat[(kb + 8 * h + t) * NQ + q];
      HSSK_COMPILER_FENCE();
#pragma unroll
      for (int t = 0; t < 8; t++) {
        const double y = x[4 * h + (t >> 1)][t & 1];
#pragma unroll
        for (int q = 0; q < NQ; q++) sj_acc<MODE>(my + (p[t][q] & 0x7fffffff) * SJ_LD, p[t][q] < 0 ? -y : y, sink);
      }
    }
  };
  if (k0 < kfull) {
    fetch(v, k0);
    for (;;) {
      long long kn = k0 + step;
      bool hn = kn < kfull;
      fetch(w, hn ? kn : k0);   // (unconditional: the wait counts in update() stay exact; the last one re-reads k0)
      update(v, k0);
      k0 = kn;
      if (!hn) break;
      kn = k0 + step;
      hn = kn < kfull;
      fetch(v, hn ? kn : k0);
      update(w, k0);
      k0 = kn;
      if (!hn) break;
    }
  }
  for (; k0 < K; k0 += step)   // tail (and the whole range when the columns are not 16-byte aligned)
    for (int t = 0; t < KB && k0 + t < K; t++) {
      const double x = a[k0 + t];
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const int pq = pat[(k0 + t) * NQ + q];
        sj_acc<MODE>(my + (pq & 0x7fffffff) * SJ_LD, pq < 0 ? -x : x, sink);
      }
    }
  if (MODE == 2) my[0] = sink;
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * dn; i += NW * 64) {
    const int jj = i / dn, c = i - jj * dn;
    if (j0 + jj < n_out) St[c + (j0 + jj) * lds] = acc[c * SJ_LD + jj];
  }
}

// ---- fallbacks for wide blocks: accumulator tiles below 64 KB -----------------------------------------------------
// St(0:dn, j) = sum_k A(k, j) R(k, :) for CT columns j per workgroup, lanes walk k (pattern in registers)
template <int NQ, int CT>
__global__ void __launch_bounds__(SJ_T) sjlt_t_small_kernel(const double* __restrict__ A, long long lda, long long K, long long n_out,
                                                             const int* __restrict__ pat, int dn, double* __restrict__ St, long long lds) {
  HSSK_DYN_SHARED(double, acc);   // CT x (dn + 1)
  const int ldc = dn + 1;
  const long long j0 = (long long)blockIdx.x * CT;
  const int nj = (int)std::min<long long>(CT, n_out - j0);
  for (int i = threadIdx.x; i < CT * ldc; i += SJ_T) acc[i] = 0.;
  __syncthreads();
  for (long long kb = (long long)threadIdx.x; kb < K; kb += SJ_T) {
    int p[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) p[q] = pat[kb * NQ + q];
    const double* a = A + kb + j0 * lda;
    for (int jb = 0; jb < nj; jb += SJ_U) {
      double v[SJ_U];
#pragma unroll
      for (int u = 0; u < SJ_U; u++) v[u] = jb + u < nj ? a[(long long)(jb + u) * lda] : 0.;
#pragma unroll
      for (int u = 0; u < SJ_U; u++) {
        if (jb + u < nj) {
          double* row = acc + (jb + u) * ldc;
#pragma unroll
          for (int q = 0; q < NQ; q++) hssk_lds_add(row + (p[q] & 0x7fffffff), p[q] < 0 ? -v[u] : v[u]);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nj * dn; i += SJ_T) {
    const int jj = i / dn, c = i - jj * dn;
    St[c + (j0 + jj) * lds] = acc[jj * ldc + c];
  }
}

// St(0:dn, i) = sum_k A(i, k) R(k, :) for RT rows i per workgroup; a workgroup-wide load covers 256 / RT columns k
template <int NQ, int RT>
__global__ void __launch_bounds__(SJ_T) sjlt_n_small_kernel(const double* __restrict__ A, long long lda, long long K, long long n_out,
                                                             const int* __restrict__ pat, int dn, double* __restrict__ St, long long lds) {
  HSSK_DYN_SHARED(double, acc);   // (dn + 1) x (RT + 1)
  constexpr int G = SJ_T / RT;
  constexpr int LD = RT + 1;
  const long long i0 = (long long)blockIdx.x * RT;
  const int r = threadIdx.x % RT, g = threadIdx.x / RT;
  const bool rok = i0 + r < n_out;
  for (int i = threadIdx.x; i < (dn + 1) * LD; i += SJ_T) acc[i] = 0.;
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
        for (int q = 0; q < NQ; q++) {
          const int p = pat[k * NQ + q];
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

constexpr int SJ_LDS_DOUBLES = 8000;          // < 64 KB, the default dynamic LDS limit (small kernels)
constexpr size_t SJ_BIG_LDS_BYTES = 150 * 1024;  // of the CU's 160 KB (64-wide kernels, one workgroup per CU)

int env_int(const char* name, int dflt) { const char* v = std::getenv(name); return v ? std::atoi(v) : dflt; }

template <int NQ> void launch_sketch(hssk_ctx* ctx, int transA, long long n_out, long long K, const double* A, long long lda,
                                     const int* pat, int dn, double* St, long long lds) {
  const size_t big = sizeof(double) * (size_t)(dn + 1) * SJ_LD;
  const unsigned g64 = (unsigned)((n_out + 63) / 64);
  // tuning / diagnostic overrides (tools/sjlt_only.py)
  const int variant = env_int(transA ? "HSSK_SJLT_TV" : "HSSK_SJLT_NV", 0), mode = env_int("HSSK_SJLT_MODE", 0);
#define SJ_BIG(kernel, threads, ...)                                                                  \
  do {                                                                                                \
    hssk_rt::allow_dynamic_lds(kernel, big);                                                          \
    HSSK_LAUNCH(kernel, dim3(g64), dim3(threads), big, ctx->stream, A, lda, K, n_out, pat, dn, St, lds, ##__VA_ARGS__); \
  } while (0)
  if (big <= SJ_BIG_LDS_BYTES && variant >= 0) {
    // measured at N = 1e5, dn = 192, nnz = 4 (tools/sjlt_only.py, profiles/r01_sjlt_variants.txt): A R 14.5 ms with
    // 16 waves x 16 loads in flight (14.8 with x 8), A^T R 16.5 ms with 8 waves x 128-byte runs (17.5 with 16 waves,
    // 21.9 with 64-byte runs)
    constexpr int UN = NQ == 4 ? 16 : 8;   // U x NQ pattern words live in scalar registers
    if (!transA) {
      if (mode == 1) SJ_BIG((sjlt_n_kernel<NQ, 16, UN, 1>), 1024);
      else if (mode == 2) SJ_BIG((sjlt_n_kernel<NQ, 16, UN, 2>), 1024);
      else if (variant == 1) SJ_BIG((sjlt_n_kernel<NQ, 16, 8, 0>), 1024);
      else if (variant == 2) SJ_BIG((sjlt_n_kernel<NQ, 8, UN, 0>), 512);
      else SJ_BIG((sjlt_n_kernel<NQ, 16, UN, 0>), 1024);
    } else {
      const int aligned = (lda % 2 == 0) && ((reinterpret_cast<size_t>(A) & 15) == 0);
      if (mode == 1) SJ_BIG((sjlt_t_kernel<NQ, 8, 16, 1>), 512, aligned);
      else if (mode == 2) SJ_BIG((sjlt_t_kernel<NQ, 8, 16, 2>), 512, aligned);
      else if (variant == 1) SJ_BIG((sjlt_t_kernel<NQ, 16, 16, 0>), 1024, aligned);
      else if (variant == 2) SJ_BIG((sjlt_t_kernel<NQ, 16, 8, 0>), 1024, aligned);
      else SJ_BIG((sjlt_t_kernel<NQ, 8, 16, 0>), 512, aligned);
    }
    return;
  }
#undef SJ_BIG
  if (transA) {
#define SJ_LAUNCH_T(CT)                                                                                                        \
  HSSK_LAUNCH((sjlt_t_small_kernel<NQ, CT>), dim3((unsigned)((n_out + CT - 1) / CT)), dim3(SJ_T), sizeof(double) * CT * (dn + 1), \
              ctx->stream, A, lda, K, n_out, pat, dn, St, lds)
    if (16 * (dn + 1) <= SJ_LDS_DOUBLES) SJ_LAUNCH_T(16);
    else if (8 * (dn + 1) <= SJ_LDS_DOUBLES) SJ_LAUNCH_T(8);
    else SJ_LAUNCH_T(4);
#undef SJ_LAUNCH_T
  } else {
#define SJ_LAUNCH_N(RT)                                                                                                        \
  HSSK_LAUNCH((sjlt_n_small_kernel<NQ, RT>), dim3((unsigned)((n_out + RT - 1) / RT)), dim3(SJ_T), sizeof(double) * (RT + 1) * (dn + 1), \
              ctx->stream, A, lda, K, n_out, pat, dn, St, lds)
    if (33 * (dn + 1) <= SJ_LDS_DOUBLES) SJ_LAUNCH_N(32);
    else if (17 * (dn + 1) <= SJ_LDS_DOUBLES) SJ_LAUNCH_N(16);
    else if (9 * (dn + 1) <= SJ_LDS_DOUBLES) SJ_LAUNCH_N(8);
    else SJ_LAUNCH_N(4);
#undef SJ_LAUNCH_N
  }
}

}  // namespace

int hssk_sjlt_dense(hssk_ctx* ctx, double* Rt, int dn, long long K, long long ld, const int* pat, int nnz) {
  HSSK_API_BEGIN
  if (dn <= 0 || K <= 0) return 0;
  if (nnz < 0 || nnz > 8) throw std::invalid_argument("hssk_sjlt_dense: nnz out of range (max 8)");
  unsigned blocks = (unsigned)std::min<long long>((K + 3) / 4, 16384);
  HSSK_LAUNCH(sjlt_dense_kernel, dim3(blocks), dim3(256), 0, ctx->stream, Rt, dn, K, ld, pat, nnz <= 4 ? 4 : 8);
  hssk_rt::check_launch();
  HSSK_API_END
}

int hssk_sjlt_sketch(hssk_ctx* ctx, int transA, long long n_out, long long K, const double* A, long long lda, const int* pat,
                     int nnz, int dn, double* St, long long lds) {
  HSSK_API_BEGIN
  if (n_out <= 0 || dn <= 0) return 0;
  if (nnz < 0 || nnz > 8) throw std::invalid_argument("hssk_sjlt_sketch: nnz out of range (max 8)");
  if (dn > 1024) throw std::invalid_argument("hssk_sjlt_sketch: more than 1024 sketch columns in one call");
  hssk_rt::event_record(ctx->ev0, ctx->stream);
  if (nnz <= 4) launch_sketch<4>(ctx, transA, n_out, K, A, lda, pat, dn, St, lds);
  else launch_sketch<8>(ctx, transA, n_out, K, A, lda, pat, dn, St, lds);
  hssk_rt::event_record(ctx->ev1, ctx->stream);
  ctx->dgemm_timed = true;
  ctx->dgemm_timed_flops = 2.0 * (double)n_out * (double)K * nnz;
  hssk_rt::check_launch();
  HSSK_API_END
}
