// getrf_quad_kernel: LU with partial pivoting (DenseMatrix::LU -> getrf, dense/DenseMatrix.cpp:564-589) of ONE matrix of up
// to 192 rows by ONE workgroup with the WHOLE matrix in registers -- the diagonal tiles of a BLR front whose tiles are
// smaller than the 256-row default (the 156-row tiles of the 200^3 problem's separators), the root blocks of HSS matrices.
//
// Why: the LU of a diagonal tile is a serial chain in the middle of every BLR block step.  The panel kernel
// (getrf_wg2_kernel, hssk_trsm_lu.hip) keeps a 32-column panel in registers and pays, per panel, a write-back, the row
// interchanges of the other columns through global memory and a trailing update that streams the rest of the tile through
// the CU: 0.42 ms for a 156 x 156 tile = 2.7 us per elimination step, 107 of the 192 ms of the 200 x 200 root front.
// A ROW is spread over two (eight waves, 256 registers per lane) or four (sixteen waves, 128) adjacent lanes, column j in lane
// j mod 2 (4), register j / 2 (4).  An elimination step is an arg max over the lanes that own column k (DPP in
// the wave, sixteen LDS words across the waves), the pivot row through the LDS (its quad writes it, every quad reads its
// own quarter as broadcasts), the multiplier handed round the quad by one DPP move and (n - k) / 4 fmas per lane on
// registers: two barriers, nothing through memory.  Rows never move: a quad keeps the POSITION of its row in the interchanged
// order (dgetf2's swap of rows k and p exchanges two positions), ties of the arg max go to the smaller position -- the pivots
// of the swapping kernels -- and the rows are written back at their positions.  (A first form with a whole row per lane --
// four waves, 512 registers each -- does not exist on this register file: 256 of them are accumulation registers.)
#include "hssk_device.h"
#include "hssk_internal.h"

namespace {

// the multiplier of a row, from the lane that owns column k to the LPR lanes of the row (kq is a constant of the unrolled step)
template <int LPR>
__device__ __forceinline__ double lq_bcast(double l, int kq) {
  if (LPR == 2) return kq == 0 ? hssk_pair_bcast<0>(l) : hssk_pair_bcast<1>(l);
  return kq == 0 ? hssk_quad_bcast<0>(l) : (kq == 1 ? hssk_quad_bcast<1>(l) : (kq == 2 ? hssk_quad_bcast<2>(l) : hssk_quad_bcast<3>(l)));
}

// LPR lanes per row (2 or 4), NCL columns per lane (a multiple of 8): the matrix has up to LPR NCL columns and up to 256 rows.
// The elimination steps are taken in blocks of 8 registers (8 LPR steps, unrolled: every register index static); after a
// block the live registers move down by 8, so that the NEXT block is the same code: the step loop is a loop, ~3000
// instructions that stay in the instruction cache.  (Fully unrolled, the 160 steps of a 156-row tile are 100-250 KB of code
// executed once: 2.4 us per step whatever the lanes per row -- instruction fetch, not arithmetic.)  A finished column -- the
// multipliers below the pivot, the U entries above -- leaves the registers as it is finished: to a scratch copy of the tile in
// the order of the THREADS' rows; the rows go to their final positions in one pass at the end.
template <int LPR, int NCL>
__global__ __launch_bounds__(256 * LPR) HSSK_WAVES_PER_SIMD(LPR) void getrf_quad_kernel(const hssk_lu_desc* __restrict__ descs, double* __restrict__ scratch, long long sstride) {
  constexpr int T = 256 * LPR, NWV = T / 64, NB = NCL / 8;
  static_assert(NCL % 8 == 0, "blocks of 8 registers");
  HSSK_SHARED double s_row[2][LPR][NCL];
  HSSK_SHARED double s_val[2][NWV];
  HSSK_SHARED int s_idx[2][NWV];
  HSSK_SHARED int s_pos[256];
  const hssk_lu_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = tid / LPR, q = tid % LPR;
  const int n = p.n, lda = p.lda;
  double* __restrict__ A = p.A;
  double* __restrict__ W = scratch + (size_t)blockIdx.x * (size_t)sstride;   // n x n, leading dimension n, rows in thread order
  const bool has = row < n;
  double a[NCL];
#pragma clang loop unroll(full)
  for (int i = 0; i < NCL; i++) a[i] = (has && q + LPR * i < n) ? hssk_gload(A, (size_t)row + (size_t)(q + LPR * i) * lda) : 0.;
  int pos = row, mypiv = 0, info = 0;
  for (int kb = 0; kb < NB; kb++) {
    const int nlive = NCL - 8 * kb;   // registers that still hold columns (a multiple of 8)
    if (LPR * 8 * kb >= n) break;
#pragma clang loop unroll(full)
    for (int sreg = 0; sreg < 8; sreg++) {
#pragma clang loop unroll(full)
      for (int kq = 0; kq < LPR; kq++) {
        const int k = LPR * (8 * kb + sreg) + kq;
        if (k < n) {
          const int pb = (LPR * sreg + kq) & 1;
          // ---- pivot: first arg max of |A(:, k)| over the positions >= k (the lanes that own column k)
          const bool cand = has && q == kq && pos >= k;
          double bv = cand ? fabs(a[sreg]) : -1.;
          int bi = cand ? pos : 0x7fffffff;
          hssk_wave_argmax(bv, bi);
          if (lane == 0) { s_val[pb][wave] = bv; s_idx[pb][wave] = bi; }
          __syncthreads();
          double v = s_val[pb][lane & (NWV - 1)];
          int pp = s_idx[pb][lane & (NWV - 1)];
          hssk_row_argmax(v, pp);
          // ---- the rows at positions k and pp exchange positions; the pivot row goes out through the LDS
          const bool ispiv = has && pos == pp;
          if (has && pos == k) pos = pp;
          else if (ispiv) pos = k;
          if (ispiv) {
#pragma clang loop unroll(full)
            for (int b = 0; b < NB; b++) {
              if (8 * b < nlive) {
#pragma clang loop unroll(full)
                for (int i = 8 * b; i < 8 * b + 8; i++)
                  if (i >= sreg) s_row[pb][q][i] = a[i];
              }
            }
          }
          if (tid == LPR * k) mypiv = pp;
          __syncthreads();
          const double akk = s_row[pb][kq][sreg];
          if (akk == 0. && !info) info = k + 1;
          // the multiplier of the row: computed by the lane that owns column k, handed to the other lanes of the row; the
          // finished entry of column k (multiplier or U entry) leaves for the scratch copy
          double l = 0.;
          if (q == kq && has) {
            if (pos > k && akk != 0.) { l = a[sreg] * (1. / akk); a[sreg] = l; }
            hssk_gstore(W, (size_t)row + (size_t)k * n, a[sreg]);
          }
          l = lq_bcast<LPR>(l, kq);
          if (q > kq) a[sreg] -= l * s_row[pb][q][sreg];
#pragma clang loop unroll(full)
          for (int b = 0; b < NB; b++) {
            if (8 * b < nlive) {
#pragma clang loop unroll(full)
              for (int i = 8 * b; i < 8 * b + 8; i++)
                if (i > sreg) a[i] -= l * s_row[pb][q][i];
            }
          }
        }
      }
    }
    // the live registers move down by 8
#pragma clang loop unroll(full)
    for (int b = 0; b + 1 < NB; b++) {
      if (8 * (b + 1) < nlive) {
#pragma clang loop unroll(full)
        for (int i = 8 * b; i < 8 * b + 8; i++) a[i] = a[i + 8];
      }
    }
  }
  if (has && q == 0) { p.piv[row] = mypiv; s_pos[row] = pos; }
  __syncthreads();
  // ---- rows to their final positions
  for (int e = tid; e < n * n; e += T) {
    const int r = e % n, j = e / n;
    hssk_gstore(A, (size_t)s_pos[r] + (size_t)j * lda, hssk_gload(W, (size_t)e));
  }
  if (tid == 0) *p.info = info;
}

}  // namespace

// launches the register kernel for a staged batch (dd: device descriptors) whose largest matrix has nmax rows; false when
// nmax is outside its reach (the caller takes its other kernels)
extern "C" bool hssk_getrf_row_launch(hssk_ctx* ctx, const hssk_lu_desc* dd, int count, int nmax) {
  static const bool off = [] { const char* e = std::getenv("HSSK_LU_NO_ROW"); return e && e[0] == '1'; }();   // (A/B)
  if (off || nmax > 192 || nmax <= 0) return false;
  const dim3 grid((unsigned)count);
  const long long sstride = (long long)nmax * nmax;
  double* scratch = ctx->aux(sizeof(double) * (size_t)sstride * (size_t)count);
  // two lanes per row (eight waves, two per SIMD, up to 96 columns = 192 registers per lane): what every lane of a wave executes
  // whether it owns the step's column or not -- the arg max, the reciprocal, the position bookkeeping -- is issued by half as
  // many waves per SIMD as with four lanes per row (the small matrices, where registers are no concern, keep four)
  if (nmax <= 96) HSSK_LAUNCH((getrf_quad_kernel<4, 24>), grid, dim3(1024), 0, ctx->stream, dd, scratch, sstride);
  else if (nmax <= 128) HSSK_LAUNCH((getrf_quad_kernel<2, 64>), grid, dim3(512), 0, ctx->stream, dd, scratch, sstride);
  else if (nmax <= 160) HSSK_LAUNCH((getrf_quad_kernel<2, 80>), grid, dim3(512), 0, ctx->stream, dd, scratch, sstride);
  else HSSK_LAUNCH((getrf_quad_kernel<2, 96>), grid, dim3(512), 0, ctx->stream, dd, scratch, sstride);
  return true;
}
