// The sketch GEMM  C(m x n) = alpha A(m x k) op(B) + beta C  with m = number of random samples
// (<= a few hundred) and n, k = the matrix dimension (1e5): 98% of all flops of HSS compression
// (reference: AFunctor, HSS/HSSExtra.hpp:236-239, Sr = A R and Sc = A^H R; here in the transposed
// sample layout S^T = R^T op(A), so every operand panel is read contiguously).
//
// Bound: FP64 MFMA (v_mfma_f64_16x16x4_f64; gfx950 FP64 matrix peak 78.6 TFLOP/s).  Arithmetic
// intensity per HBM byte of B is m/4 flop/B (48 at m = 192), far above the 10 flop/B ridge.
//
// Tiling: workgroup = 256 threads (4 wave64 as 2x2), output tile BM x 64 with BM in {64,128,192}
// chosen so one workgroup covers all m sample rows when m <= 192 (B, the N x N matrix, is then
// streamed from HBM exactly once; the A panel -- R^T, 24 KB per k-stage -- is shared by all
// workgroups through L2).  K advances 16 per stage through double-buffered LDS (As[k][i],
// Bs[k][j], row stride +16 doubles so the two 16-lane halves of a ds_read_b64 hit different bank
// halves) with the next stage prefetched into registers while the MFMAs of the current one issue.
// The K range is split over gridDim.z workgroups (deterministic: partial tiles go to scratch and a
// second kernel reduces them in fixed order) so that the workgroup count fills 256 CUs evenly.
#include "hssk_device.h"
#include "hssk_internal.h"

#include <algorithm>

namespace {

constexpr int BN = 64, BK = 16;

template <int BM, bool TRANSB>
__global__ __launch_bounds__(256) void dgemm_kernel(int m, long long n, long long k, const double* __restrict__ A,
                                                    long long lda, const double* __restrict__ B, long long ldb,
                                                    double* __restrict__ P, long long ldp, long long pstride,
                                                    long long kchunk) {
  constexpr int LDA_S = BM + 16, LDB_S = BN + 16;
  constexpr int WM = BM / 2;       // rows per wave
  constexpr int MT = WM / 16;      // MFMA tiles per wave along M
  constexpr int NT = 2;            // 32 columns per wave
  constexpr int A_PER_THREAD = BM * BK / 256;
  constexpr int B_PER_THREAD = BN * BK / 256;
  HSSK_SHARED double As[2 * BK * LDA_S];
  HSSK_SHARED double Bs[2 * BK * LDB_S];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const long long j0 = (long long)blockIdx.x * BN;
  const int i0 = blockIdx.y * BM;
  const long long kbeg = (long long)blockIdx.z * kchunk;
  const long long kend = kbeg + kchunk < k ? kbeg + kchunk : k;
  const int wm = (wave & 1) * WM, wn = (wave >> 1) * 32;

  hssk_d4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; a++)
#pragma unroll
    for (int b = 0; b < NT; b++) acc[a][b] = hssk_d4{0., 0., 0., 0.};

  double ra[A_PER_THREAD], rb[B_PER_THREAD];

  auto load_tiles = [&](long long k0) {
    // A tile: (i, kk) contiguous along i
#pragma unroll
    for (int r = 0; r < A_PER_THREAD; r++) {
      int e = tid + 256 * r;
      int i = e % BM, kk = e / BM;
      long long gk = k0 + kk;
      int gi = i0 + i;
      ra[r] = (gi < m && gk < kend) ? A[gi + gk * lda] : 0.;
    }
#pragma unroll
    for (int r = 0; r < B_PER_THREAD; r++) {
      int e = tid + 256 * r;
      if (TRANSB) {  // op(B)(k, j) = B(j, k): contiguous along j
        int j = e % BN, kk = e / BN;
        long long gk = k0 + kk, gj = j0 + j;
        rb[r] = (gj < n && gk < kend) ? B[gj + gk * ldb] : 0.;
      } else {       // op(B)(k, j) = B(k, j): contiguous along k
        int kk = e % BK, j = e / BK;
        long long gk = k0 + kk, gj = j0 + j;
        rb[r] = (gj < n && gk < kend) ? B[gk + gj * ldb] : 0.;
      }
    }
  };
  auto store_tiles = [&](int buf) {
    double* as = As + buf * BK * LDA_S;
    double* bs = Bs + buf * BK * LDB_S;
#pragma unroll
    for (int r = 0; r < A_PER_THREAD; r++) {
      int e = tid + 256 * r;
      as[(e / BM) * LDA_S + (e % BM)] = ra[r];
    }
#pragma unroll
    for (int r = 0; r < B_PER_THREAD; r++) {
      int e = tid + 256 * r;
      if (TRANSB) bs[(e / BN) * LDB_S + (e % BN)] = rb[r];
      else bs[(e % BK) * LDB_S + (e / BK)] = rb[r];
    }
  };

  if (kbeg < kend) {
    load_tiles(kbeg);
    store_tiles(0);
  }
  __syncthreads();
  int buf = 0;
  for (long long k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) load_tiles(k0 + BK);
    const double* as = As + buf * BK * LDA_S;
    const double* bs = Bs + buf * BK * LDB_S;
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      double af[MT], bf[NT];
#pragma unroll
      for (int a = 0; a < MT; a++) af[a] = as[(ks + l4) * LDA_S + wm + a * 16 + l15];
#pragma unroll
      for (int b = 0; b < NT; b++) bf[b] = bs[(ks + l4) * LDB_S + wn + b * 16 + l15];
#pragma unroll
      for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < NT; b++)  // swapped operands: lane holds C[i = l15][j = l4 + 4r]
          acc[a][b] = hssk_mfma_f64_16x16x4(bf[b], af[a], acc[a][b]);
    }
    if (more) store_tiles(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // partial tile -> P (slice blockIdx.z), plain stores; the reduce kernel applies alpha/beta
  double* Pz = P + (long long)blockIdx.z * pstride;
#pragma unroll
  for (int a = 0; a < MT; a++)
#pragma unroll
    for (int b = 0; b < NT; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int gi = i0 + wm + a * 16 + l15;
        long long gj = j0 + wn + b * 16 + l4 + 4 * r;
        if (gi < m && gj < n) Pz[gi + gj * ldp] = acc[a][b][r];
      }
}

// C = alpha * sum_z P_z + beta * C   (fixed summation order -> deterministic)
__global__ void dgemm_reduce_kernel(int m, long long n, const double* __restrict__ P, long long ldp,
                                    long long pstride, int nz, double alpha, double beta,
                                    double* __restrict__ C, long long ldc) {
  long long total = (long long)m * n;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    int i = (int)(e % m);
    long long j = e / m;
    double s = 0.;
    for (int z = 0; z < nz; z++) s += P[i + j * ldp + z * pstride];
    double* c = C + i + j * ldc;
    double v = alpha * s;
    if (beta != 0.) v += beta * (*c);
    *c = v;
  }
}

template <int BM>
void launch_dgemm(hssk_ctx* ctx, int transB, dim3 grid, int m, long long n, long long k, const double* A,
                  long long lda, const double* B, long long ldb, double* P, long long ldp, long long pstride,
                  long long kchunk) {
  if (transB)
    HSSK_LAUNCH((dgemm_kernel<BM, true>), grid, dim3(256), 0, ctx->stream, m, n, k, A, lda, B, ldb, P, ldp, pstride, kchunk);
  else
    HSSK_LAUNCH((dgemm_kernel<BM, false>), grid, dim3(256), 0, ctx->stream, m, n, k, A, lda, B, ldb, P, ldp, pstride, kchunk);
}

}  // namespace

extern "C" int hssk_dgemm(hssk_ctx* ctx, int transB, int m, long long n, long long k, double alpha,
                          const double* A, long long lda, const double* B, long long ldb, double beta,
                          double* C, long long ldc) {
  HSSK_API_BEGIN
  if (m <= 0 || n <= 0) return 0;
  int BM = m > 128 ? 192 : (m > 64 ? 128 : 64);
  unsigned gm = (unsigned)((m + BM - 1) / BM);
  unsigned gn = (unsigned)((n + BN - 1) / BN);
  // split K so that the grid is a near-multiple of the 512 resident workgroup slots (256 CUs x 2)
  const long long slots = 512;
  long long tiles = (long long)gm * gn;
  long long ksteps = (k + BK - 1) / BK;
  int best = 1;
  double best_eff = 0.;
  for (int s = 1; s <= 16; s++) {
    if (s > 1 && ksteps / s < 64) break;  // keep chunks long enough to amortise the epilogue
    long long wgs = tiles * s;
    long long rounds = (wgs + slots - 1) / slots;
    double eff = (double)wgs / (double)(rounds * slots);
    if (eff > best_eff + 0.02) { best_eff = eff; best = s; }
  }
  if (k <= 0) best = 1;
  long long kchunk = ((ksteps + best - 1) / best) * BK;
  if (kchunk <= 0) kchunk = BK;
  int nz = (int)((k + kchunk - 1) / kchunk);
  if (nz < 1) nz = 1;
  long long ldp = m;
  long long pstride = ldp * n;
  double* P = ctx->scratch(sizeof(double) * (size_t)pstride * nz);
  dim3 grid(gn, gm, (unsigned)nz);
  hssk_rt::event_record(ctx->ev0, ctx->stream);
  if (BM == 192) launch_dgemm<192>(ctx, transB, grid, m, n, k, A, lda, B, ldb, P, ldp, pstride, kchunk);
  else if (BM == 128) launch_dgemm<128>(ctx, transB, grid, m, n, k, A, lda, B, ldb, P, ldp, pstride, kchunk);
  else launch_dgemm<64>(ctx, transB, grid, m, n, k, A, lda, B, ldb, P, ldp, pstride, kchunk);
  hssk_rt::event_record(ctx->ev1, ctx->stream);
  ctx->dgemm_timed = true;
  long long total = (long long)m * n;
  unsigned rb = (unsigned)std::min<long long>((total + 255) / 256, 4096);
  HSSK_LAUNCH(dgemm_reduce_kernel, dim3(rb), dim3(256), 0, ctx->stream, m, n, (const double*)P, ldp, pstride, nz, alpha, beta, C, ldc);
  hssk_rt::check_launch();
  HSSK_API_END
}
