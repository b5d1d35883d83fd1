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
#include <cstdlib>
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
  int info = 0, mypiv = 0;
  for (int k = 0; k < n; k++) {
    // pivot: first arg max_{i >= k} |A(i, k)|
    double bv = -1.;
    int bi = 0x7fffffff;
    for (int i = k + tid; i < n; i += LU_THREADS) {
      double v = fabs(A[i + (size_t)k * ld]);
      if (v > bv) { bv = v; bi = i; }
    }
    hssk_wave_argmax(bv, bi);
    if (lane == 0) { s_val[wave] = bv; s_idx[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      double v = s_val[0];
      int ix = s_idx[0];
      for (int w = 1; w < LU_THREADS / 64; w++)
        if (s_val[w] > v || (s_val[w] == v && s_idx[w] < ix)) { v = s_val[w]; ix = s_idx[w]; }
      *s_piv = ix;
    }
    __syncthreads();
    const int pv = *s_piv;
    // (the pivots go to global memory a workgroup's worth at a time: thread k % LU_THREADS keeps step k's)
    if (tid == k % LU_THREADS) mypiv = pv;
    if ((k + 1) % LU_THREADS == 0 || k == n - 1) {
      const int kb = k - k % LU_THREADS;
      if (kb + tid <= k) piv[kb + tid] = mypiv;
    }
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

// ---- blocked LU of one matrix by ONE workgroup, 128 < n <= 512 (diagonal tiles of a BLR factorization, HSS roots with
// ranks in the hundreds).  getrf_kernel above runs every elimination step on the matrix in global memory with six
// barriers and one column per wave at a time: 4.6 ms for a 256 x 256 tile -- the longest kernel of a BLR block step.
// Here the current 32-column panel (rows j0 .. n) lives in LDS: a step is an LDS arg max, an LDS row swap and an LDS
// rank-1 update of at most 31 columns (three barriers, ~1 us); the row interchange of the columns OUTSIDE the panel is
// issued by one thread per column while the panel is updated (the same thread owns a column in every step, so its
// exchanges stay ordered).  Column k of the panel is kept unscaled until the panel is done (no read / write hazard inside
// a step; 1 / pivot in s_inv).  The trailing matrix is then updated from the LDS panel, 64 columns at a time:
// U12 = L11^{-1} A12 with a column per thread in registers, A22 -= L21 U12 with rows along the lanes (coalesced).
// Same pivoting rule as getrf_kernel (first arg max), same results up to the order of the updates.
constexpr int LUW_T = 512;
constexpr int LUW_NB = 32;
constexpr int LUW_NMAX = 512;
constexpr int LUW_CH = 64;   // trailing columns per pass
__global__ __launch_bounds__(LUW_T) void getrf_wg_kernel(const hssk_lu_desc* __restrict__ descs) {
  HSSK_DYN_SHARED(double, s_dyn);
  HSSK_SHARED double s_val[LUW_T / 64];
  HSSK_SHARED int s_idx[LUW_T / 64];
  HSSK_SHARED double s_inv[LUW_NB];
  HSSK_SHARED int s_src[LUW_NMAX];        // row that ends up in position r after the panel's interchanges
  HSSK_SHARED int s_aff[2 * LUW_NB];      // the positions that change
  HSSK_SHARED int s_naff;
  const hssk_lu_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = p.n, lda = p.lda;
  double* __restrict__ A = p.A;
  int info = 0;
  for (int j0 = 0; j0 < n; j0 += LUW_NB) {
    const int nb = min(LUW_NB, n - j0), mp = n - j0, LP = mp | 1, jend = j0 + nb;
    double* s_P = s_dyn;                              // mp x nb panel, leading dimension LP
    double* s_U = s_dyn + (size_t)LP * LUW_NB;        // nb x LUW_CH block of U12, leading dimension LUW_NB + 1
    for (int j = 0; j < nb; j++)
      for (int i = tid; i < mp; i += LUW_T) s_P[i + j * LP] = hssk_gload(A, (size_t)(j0 + i) + (size_t)(j0 + j) * lda);
    for (int e = tid; e < mp; e += LUW_T) s_src[e] = e;
    __syncthreads();
    for (int k = 0; k < nb; k++) {
      // ---- pivot: first arg max_{i >= k} |P(i, k)|  (one row per thread: mp <= 512)
      double bv = -1.;
      int bi = 0x7fffffff;
      if (k + tid < mp) { bv = fabs(s_P[(k + tid) + k * LP]); bi = k + tid; }
      hssk_wave_argmax(bv, bi);
      if (lane == 0) { s_val[wave] = bv; s_idx[wave] = bi; }
      __syncthreads();
      double v = s_val[0];
      int pv = s_idx[0];
      for (int w = 1; w < LUW_T / 64; w++)
        if (s_val[w] > v || (s_val[w] == v && s_idx[w] < pv)) { v = s_val[w]; pv = s_idx[w]; }
      if (tid == 0) {
        p.piv[j0 + k] = j0 + pv;
        const int t = s_src[k]; s_src[k] = s_src[pv]; s_src[pv] = t;   // (the columns outside the panel follow once per panel)
      }
      // ---- row interchange inside the panel
      if (pv != k && tid < nb) { const double a = s_P[k + tid * LP]; s_P[k + tid * LP] = s_P[pv + tid * LP]; s_P[pv + tid * LP] = a; }
      __syncthreads();
      const double akk = s_P[k + k * LP];
      if (akk == 0.) {
        if (!info) info = j0 + k + 1;
        if (tid == 0) s_inv[k] = 0.;
      } else {
        const double inv = 1. / akk;
        if (tid == 0) s_inv[k] = inv;
        // ---- rank-1 update of the panel's columns k+1 .. nb (column k itself stays unscaled for now)
        // (a row per thread -- mp <= 512 --, the columns in a loop: no index divisions, the pivot row read as broadcasts)
        const int i = k + 1 + tid;
        if (i < mp) {
          const double li = s_P[i + k * LP] * inv;
#pragma unroll 4
          for (int j = k + 1; j < nb; j++) s_P[i + j * LP] -= li * s_P[k + j * LP];
        }
      }
      __syncthreads();
    }
    // ---- L = P(:, k) / pivot below the diagonal; panel back to global memory
    for (int j = 0; j < nb; j++)
      for (int i = tid; i < mp; i += LUW_T) {
        double val = s_P[i + j * LP];
        if (i > j) { val *= s_inv[j]; s_P[i + j * LP] = val; }
        hssk_gstore(A, (size_t)(j0 + i) + (size_t)(j0 + j) * lda, val);
      }
    // ---- the panel's row interchanges on the columns outside it, all at once: the positions that changed (at most 2 nb)
    // are read -- every thread its share of (position, column) pairs, into registers -- and, behind a barrier, written.
    // (Exchanging two rows per elimination step in global memory cost two dependent round trips per step: the barrier's
    // fence waits for them -- 5 us per step against 1 us of LDS work.)
    if (tid == 0) {
      int na = 0;
      for (int r = 0; r < mp && na < 2 * LUW_NB; r++)
        if (s_src[r] != r) s_aff[na++] = r;
      s_naff = na;
    }
    __syncthreads();
    {
      const int na = s_naff, nco = n - nb;
      constexpr int PER = (2 * LUW_NB * (LUW_NMAX - 1) + LUW_T - 1) / LUW_T;   // pairs per thread at most
      double val[PER];
      // (consecutive threads take consecutive affected rows of one column: the rows lie within the panel's row range)
#pragma unroll
      for (int q = 0; q < PER; q++) {
        const int e = tid + q * LUW_T;
        val[q] = 0.;
        if (e < na * nco) {
          const int a = e % na, c = e / na, col = c < j0 ? c : c + nb;
          val[q] = hssk_gload(A, (size_t)(j0 + s_src[s_aff[a]]) + (size_t)col * lda);
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < PER; q++) {
        const int e = tid + q * LUW_T;
        if (e < na * nco) {
          const int a = e % na, c = e / na, col = c < j0 ? c : c + nb;
          hssk_gstore(A, (size_t)(j0 + s_aff[a]) + (size_t)col * lda, val[q]);
        }
      }
    }
    __syncthreads();
    // ---- trailing matrix, LUW_CH columns at a time
    constexpr int LU_ = LUW_NB + 1;
    for (int c0 = jend; c0 < n; c0 += LUW_CH) {
      const int nc = min(LUW_CH, n - c0);
      for (int e = tid; e < nb * nc; e += LUW_T) s_U[(e % nb) + (e / nb) * LU_] = hssk_gload(A, (size_t)(j0 + e % nb) + (size_t)(c0 + e / nb) * lda);
      __syncthreads();
      if (tid < nc) {   // U12(:, c) = L11^{-1} A12(:, c): the column in registers, L11 read as broadcasts
        double u[LUW_NB];
#pragma unroll
        for (int i = 0; i < LUW_NB; i++) u[i] = i < nb ? s_U[i + tid * LU_] : 0.;
#pragma unroll
        for (int l = 0; l < LUW_NB; l++) {
          if (l < nb) {
#pragma unroll
            for (int i = l + 1; i < LUW_NB; i++)
              if (i < nb) u[i] -= s_P[i + l * LP] * u[l];
          }
        }
#pragma unroll
        for (int i = 0; i < LUW_NB; i++)
          if (i < nb) { s_U[i + tid * LU_] = u[i]; hssk_gstore(A, (size_t)(j0 + i) + (size_t)(c0 + tid) * lda, u[i]); }
      }
      __syncthreads();
      // A22(:, c0 .. c0+nc) -= L21 U12: rows along the lanes, a thread takes 16 columns of its row
      const int rws = mp - nb;
      constexpr int RG = LUW_T / (LUW_CH / 16);   // rows per pass
      for (int ip = 0; ip < rws; ip += RG) {
        const int i = ip + tid % RG, cg = (tid / RG) * 16;
        if (i < rws && cg < nc) {
          double acc[16];
#pragma unroll
          for (int c = 0; c < 16; c++) acc[c] = cg + c < nc ? hssk_gload(A, (size_t)(jend + i) + (size_t)(c0 + cg + c) * lda) : 0.;
          for (int l = 0; l < nb; l++) {
            const double lil = s_P[(nb + i) + l * LP];
#pragma unroll
            for (int c = 0; c < 16; c++) acc[c] -= lil * s_U[l + (cg + c) * LU_];
          }
#pragma unroll
          for (int c = 0; c < 16; c++)
            if (cg + c < nc) hssk_gstore(A, (size_t)(jend + i) + (size_t)(c0 + cg + c) * lda, acc[c]);
        }
      }
      __syncthreads();
    }
  }
  if (tid == 0) *p.info = info;
}

// ---- second form of the one-workgroup LU (the diagonal tile of a BLR block step is ONE matrix: its LU is a serial chain on
// one CU while the rest of the chip compresses tiles -- 1.48 ms for a 256 x 256 tile with getrf_wg_kernel: 0.61 ms in the 256
// elimination steps (2.4 us each: three barriers around LDS read-modify-writes), 0.68 ms in the trailing updates (17 LDS reads
// per 16 fmas), 0.21 ms in the panel's write-back and row interchanges (a serial scan of the permutation by one lane)).
//  * a thread holds ITS ROW of the 32-column panel in registers for the panel's 32 steps; rows never move: a thread keeps the
//    POSITION of its row in the interchanged order (dgetf2's swap of rows k and p exchanges two positions), the arg max breaks
//    ties by position -- the same pivots as the swapping kernel --, the pivot row is broadcast through 32 LDS words and the
//    rank-1 update is 31 - k fmas on registers: two barriers per step;
//  * the rows go back to global memory (and into the LDS, for the trailing update) at their positions; the threads whose
//    position changed -- at most 64 -- list themselves with an LDS counter, and the interchange of the other columns is the
//    gather / barrier / scatter of getrf_wg_kernel over that list;
//  * U12 = L11^{-1} A12 with a column per thread, for all trailing columns at once when n <= 256 (LDS: 66 KB panel + 59 KB
//    U12), 64 at a time above; A22 -= L21 U12 on the matrix cores: a wave owns 16-row blocks of A22, the lanes hold the
//    TRANSPOSED 16 x 16 accumulator (D = (-U12^T) L21^T: a lane's four entries lie in one row of A22 per register and the 16
//    lanes of a quarter-wave in 16 consecutive rows: 128-byte segments of the column-major tile).
// Same pivoting rule and results as getrf_wg_kernel up to the order of the updates.  HSSK_LU_WG_V1=1 selects the first form.
constexpr int LUW_CHW_SMALL = 224;   // trailing columns per pass, n <= 256 (all of them)
template <int T>
__global__ __launch_bounds__(T) void getrf_wg2_kernel(const hssk_lu_desc* __restrict__ descs, int chw) {
  HSSK_DYN_SHARED(double, s_dyn);
  HSSK_SHARED double s_val[T / 64];
  HSSK_SHARED int s_idx[T / 64];
  HSSK_SHARED double s_row[LUW_NB];
  HSSK_SHARED int s_affd[2 * LUW_NB];   // positions that receive another row ...
  HSSK_SHARED int s_affs[2 * LUW_NB];   // ... and the position that row had before the panel
  HSSK_SHARED int s_naff;
  constexpr int NW = T / 64, LU_ = LUW_NB + 1;
  const hssk_lu_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = p.n, lda = p.lda;
  double* __restrict__ A = p.A;
  int info = 0;
  for (int j0 = 0; j0 < n; j0 += LUW_NB) {
    const int nb = min(LUW_NB, n - j0), mp = n - j0, LP = mp | 1, jend = j0 + nb;
    double* s_P = s_dyn;                              // mp x nb panel in position order, leading dimension LP
    double* s_U = s_dyn + (size_t)LP * LUW_NB;        // nb x chw block of U12, leading dimension LUW_NB + 1
    const bool has = tid < mp;
    double a[LUW_NB];
#pragma unroll
    for (int j = 0; j < LUW_NB; j++) a[j] = (has && j < nb) ? hssk_gload(A, (size_t)(j0 + tid) + (size_t)(j0 + j) * lda) : 0.;
    int pos = tid;
    int mypiv = 0;   // pivot of step tid of this panel: to global memory once per panel (a store per step in front of the
                     // step's barrier made the next step wait out its round trip)
    if (tid == 0) s_naff = 0;
#pragma unroll
    for (int k = 0; k < LUW_NB; k++) {
      if (k < nb) {
        // ---- pivot: first arg max over the positions >= k
        double bv = (has && pos >= k) ? fabs(a[k]) : -1.;
        int bi = (has && pos >= k) ? pos : 0x7fffffff;
        hssk_wave_argmax(bv, bi);
        if (lane == 0) { s_val[wave] = bv; s_idx[wave] = bi; }
        __syncthreads();
        double v = s_val[0];
        int pp = s_idx[0];
#pragma unroll
        for (int w = 1; w < NW; w++)
          if (s_val[w] > v || (s_val[w] == v && s_idx[w] < pp)) { v = s_val[w]; pp = s_idx[w]; }
        // ---- the rows at positions k and pp exchange positions; the pivot row goes out through the LDS
        const bool ispiv = has && pos == pp;
        if (has && pos == k) pos = pp;
        else if (ispiv) pos = k;
        if (ispiv) {
#pragma unroll
          for (int j = k; j < LUW_NB; j++) s_row[j] = a[j];
        }
        if (tid == k) mypiv = j0 + pp;
        __syncthreads();
        const double akk = s_row[k];
        if (akk == 0.) {
          if (!info) info = j0 + k + 1;
        } else if (has && pos > k) {
          const double l = a[k] * (1. / akk);
          a[k] = l;
#pragma unroll
          for (int j = k + 1; j < LUW_NB; j++) a[j] -= l * s_row[j];
        }
      }
    }
    if (tid < nb) p.piv[j0 + tid] = mypiv;
    // ---- rows to their positions: LDS (trailing update) and global memory; who moved
    if (has) {
#pragma unroll
      for (int j = 0; j < LUW_NB; j++)
        if (j < nb) {
          s_P[pos + j * LP] = a[j];
          hssk_gstore(A, (size_t)(j0 + pos) + (size_t)(j0 + j) * lda, a[j]);
        }
      if (pos != tid) {
        const int slot = hssk_lds_inc(&s_naff);
        s_affd[slot] = pos;
        s_affs[slot] = tid;
      }
    }
    __syncthreads();
    // ---- the panel's row interchanges on the columns outside it: gather into registers, barrier, scatter
    {
      const int na = s_naff, nco = n - nb;
      constexpr int NMAXT = T == 256 ? 256 : LUW_NMAX;
      constexpr int PER = (2 * LUW_NB * (NMAXT - 1) + T - 1) / T;
      double val[PER];
#pragma unroll
      for (int q = 0; q < PER; q++) {
        const int e = tid + q * T;
        val[q] = 0.;
        if (e < na * nco) {
          const int ai = e % na, c = e / na, col = c < j0 ? c : c + nb;
          val[q] = hssk_gload(A, (size_t)(j0 + s_affs[ai]) + (size_t)col * lda);
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < PER; q++) {
        const int e = tid + q * T;
        if (e < na * nco) {
          const int ai = e % na, c = e / na, col = c < j0 ? c : c + nb;
          hssk_gstore(A, (size_t)(j0 + s_affd[ai]) + (size_t)col * lda, val[q]);
        }
      }
    }
    __syncthreads();
    // ---- trailing matrix, chw columns at a time
    const int rws = mp - nb;
    for (int c0 = jend; c0 < n; c0 += chw) {
      const int nc = min(chw, n - c0);
      for (int e = tid; e < nb * nc; e += T) s_U[(e % nb) + (e / nb) * LU_] = hssk_gload(A, (size_t)(j0 + e % nb) + (size_t)(c0 + e / nb) * lda);
      __syncthreads();
      for (int c = tid; c < nc; c += T) {   // U12(:, c) = L11^{-1} A12(:, c): the column in registers, L11 read as broadcasts
        double u[LUW_NB];
#pragma unroll
        for (int i = 0; i < LUW_NB; i++) u[i] = i < nb ? s_U[i + c * LU_] : 0.;
#pragma unroll
        for (int l = 0; l < LUW_NB; l++) {
          if (l < nb) {
#pragma unroll
            for (int i = l + 1; i < LUW_NB; i++)
              if (i < nb) u[i] -= s_P[i + l * LP] * u[l];
          }
        }
#pragma unroll
        for (int i = 0; i < LUW_NB; i++)
          if (i < nb) { s_U[i + c * LU_] = u[i]; hssk_gstore(A, (size_t)(j0 + i) + (size_t)(c0 + c) * lda, u[i]); }
      }
      __syncthreads();
      // A22(:, c0 .. c0+nc) -= L21 U12 on the matrix cores (transposed accumulator, see above).  The tiles of a wave in one
      // sequence, the NEXT tile's entries of A22 loaded (unconditionally, from clamped addresses: masked at the store) while
      // the current one is multiplied: with the loads under their own conditions right in front of the products every tile
      // waited out an L2 round trip -- 25 to 50 tiles per wave and panel.
      if (rws > 0) {
        const int y = lane & 15, kq = lane >> 4;
        const int nrt = (rws + 15) / 16, nct = (nc + 15) / 16;
        const int ntw = wave < nrt ? ((nrt - wave + NW - 1) / NW) * nct : 0;
        auto load_tile = [&](int t, hssk_d4& v) {
          const int ti = wave + NW * (t / nct), tj = t % nct;
          const int rowc = min(ti * 16 + y, rws - 1);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int colc = min(tj * 16 + kq + 4 * r, nc - 1);
            v[r] = hssk_gload(A, (size_t)(jend + rowc) + (size_t)(c0 + colc) * lda);
          }
        };
        hssk_d4 nxt = {0., 0., 0., 0.};
        if (ntw) load_tile(0, nxt);
        // (the LDS operands beyond the panel's columns: read from a clamped index and multiplied by zero -- a select on the
        // read value puts every read under a branch of its own, with a full wait behind it)
        double bl[LUW_NB / 4], mk[LUW_NB / 4];
        int kc[LUW_NB / 4];
#pragma unroll
        for (int kk = 0; kk < LUW_NB / 4; kk++) { mk[kk] = kq + 4 * kk < nb ? 1. : 0.; kc[kk] = min(kq + 4 * kk, nb - 1); }
        for (int t = 0; t < ntw; t++) {
          const int ti = wave + NW * (t / nct), tj = t % nct;
          const int row = ti * 16 + y, rowc = min(row, rws - 1);
          if (tj == 0) {
#pragma unroll
            for (int kk = 0; kk < LUW_NB / 4; kk++) bl[kk] = s_P[(nb + rowc) + kc[kk] * LP] * mk[kk];
          }
          hssk_d4 acc = nxt;
          load_tile(min(t + 1, ntw - 1), nxt);
          const int xc = min(tj * 16 + y, nc - 1);
          double au[LUW_NB / 4];
#pragma unroll
          for (int kk = 0; kk < LUW_NB / 4; kk++) au[kk] = -s_U[kc[kk] + xc * LU_];   // (bl carries the mask)
#pragma unroll
          for (int kk = 0; kk < LUW_NB / 4; kk++) acc = hssk_mfma_f64_16x16x4(au[kk], bl[kk], acc);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int col = tj * 16 + kq + 4 * r;
            if (row < rws && col < nc) hssk_gstore(A, (size_t)(jend + row) + (size_t)(c0 + col) * lda, acc[r]);
          }
        }
      }
      __syncthreads();
    }
  }
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
    hssk_wave_argmax(bv, bi);
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

// up to 1024 rows: a thread per row replays the interchanges ON ITS ROW INDEX (n steps of two compares on a register, the pivots
// read as LDS broadcasts: no memory in the dependent chain) and the rows then move in one gather / barrier / scatter per
// group of eight columns.  The kernels above walk n dependent exchanges in global memory per column: 27 us for a 256-row
// tile whatever the number of columns -- a quarter of a front's forward step with one right-hand side.
constexpr int LP_T = 256, LP_NMAX = 1024, LP_CC = 8, LP_COLS = 64;
__global__ __launch_bounds__(LP_T) void laswp_perm_kernel(const hssk_lusolve_desc* __restrict__ descs) {
  HSSK_SHARED int s_piv[LP_NMAX];
  const hssk_lusolve_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x, n = p.n;
  const int c_lo = blockIdx.y * LP_COLS, c_hi = min(p.nrhs, c_lo + LP_COLS);
  if (c_lo >= c_hi) return;
  for (int e = tid; e < n; e += LP_T) s_piv[e] = p.piv[e];
  __syncthreads();
  constexpr int RPT = LP_NMAX / LP_T;   // rows per thread
  int dst[RPT];
#pragma unroll
  for (int q = 0; q < RPT; q++) dst[q] = tid + q * LP_T;
  for (int k = 0; k < n; k++) {
    const int pv = s_piv[k];
#pragma unroll
    for (int q = 0; q < RPT; q++) dst[q] = dst[q] == k ? pv : (dst[q] == pv ? k : dst[q]);
  }
  for (int c0 = c_lo; c0 < c_hi; c0 += LP_CC) {
    double val[RPT][LP_CC];
#pragma unroll
    for (int q = 0; q < RPT; q++)
#pragma unroll
      for (int cc = 0; cc < LP_CC; cc++) {
        const int r = tid + q * LP_T, c = c0 + cc;
        val[q][cc] = (r < n && c < c_hi) ? hssk_gload(p.B, (size_t)r + (size_t)c * p.ldb) : 0.;
      }
    __syncthreads();   // (all rows of these columns are read before any of them is overwritten)
#pragma unroll
    for (int q = 0; q < RPT; q++)
#pragma unroll
      for (int cc = 0; cc < LP_CC; cc++) {
        const int r = tid + q * LP_T, c = c0 + cc;
        if (r < n && c < c_hi) hssk_gstore(p.B, (size_t)dst[q] + (size_t)c * p.ldb, val[q][cc]);
      }
  }
}

}  // namespace

extern "C" {

int hssk_laswp_vbatched(hssk_ctx* ctx, const hssk_lusolve_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  int cmax = 0, nmax = 0;
  for (int i = 0; i < count; i++) { cmax = std::max(cmax, descs[i].nrhs); nmax = std::max(nmax, descs[i].n); }
  if (cmax <= 0) return 0;
  auto* dd = (const hssk_lusolve_desc*)ctx->stage(descs, sizeof(*descs) * count);
  static const bool no_perm = [] { const char* e = std::getenv("HSSK_LASWP_NO_PERM"); return e && e[0] == '1'; }();   // (A/B)
  if (nmax <= LP_NMAX && !no_perm) {
    HSSK_LAUNCH(laswp_perm_kernel, dim3((unsigned)count, (unsigned)((cmax + LP_COLS - 1) / LP_COLS)), dim3(LP_T), 0, ctx->stream, dd);
    hssk_rt::check_launch();
    return 0;
  }
  HSSK_LAUNCH(laswp_wide_kernel, dim3((unsigned)count, (unsigned)std::min(256, (cmax + 63) / 64)), dim3(64), 0, ctx->stream, dd);
  hssk_rt::check_launch();
  HSSK_API_END
}

// ---- the blocked solve of trsm_blocked below as ONE launch (triangles of up to 512 rows): a workgroup takes 16 right-hand
// sides, keeps their n x 16 block in the LDS for the whole solve and walks the ceil(n / 64) block steps itself -- X_b =
// inv(op(T)_bb) B_b (the inverted diagonal blocks of hssk_trtri_diag_vbatched), then the remaining block rows -= op(T)(rest, b)
// X_b, both on the matrix cores with the right-hand sides as the 16 columns of the tile.  The multi-launch form spends three
// launches per block step (copy of B_b, two batched GEMMs): 13 launches of 5 - 15 us for a 256 x 256 tile whatever the number
// of right-hand sides -- 0.43 ms of a BLR block step's 2.6 ms, and 0.11 ms per tile of a front's forward / backward solve
// with one right-hand side.  T is read from L2 (512 KB at most, shared by the workgroups of a launch).
struct TfWork {
  int prob, group;
  const double* inv;   // the triangle's inverted diagonal blocks (the call's aux buffer, or the caller's)
};
constexpr int TF_T = 256, TF_NB = 64, TF_NMAX = 512;
__global__ __launch_bounds__(TF_T) void trsm_fused_kernel(const hssk_trsm_desc* __restrict__ descs, const TfWork* __restrict__ work) {
  HSSK_DYN_SHARED(double, s_x);   // (round-up of n to 64) x 16 right-hand sides, leading dimension ldx
  const TfWork wk = work[blockIdx.x];
  const hssk_trsm_desc p = descs[wk.prob];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int y = lane & 15, kq = lane >> 4;
  const int n = p.n, ldt = p.ldt, nblk = (n + TF_NB - 1) / TF_NB, npad = nblk * TF_NB, ldx = npad + 1;
  const double* __restrict__ T = p.T;
  const int c0 = wk.group * 16, nc = min(16, p.nrhs - c0);
  for (int e = tid; e < npad * 16; e += TF_T) {
    const int i = e % npad, c = e / npad;
    s_x[i + c * ldx] = (i < n && c < nc) ? hssk_gload(p.B, (size_t)i + (size_t)(c0 + c) * p.ldb) : 0.;
  }
  __syncthreads();
  const bool fwd = p.lower || p.transT;   // the effective triangle of op(T): lower (blocks first to last) or upper (last to first)
  for (int s = 0; s < nblk; s++) {
    const int b = fwd ? s : nblk - 1 - s, b0 = b * TF_NB;
    const double* __restrict__ Ti = wk.inv + (size_t)b * TF_NB * TF_NB;   // plain inverse of T_bb, zero padded, ld 64
    // ---- X_b = inv(op(T)_bb) B_b: a wave per 16 rows of the block
    {
      double av[TF_NB / 4];
#pragma unroll
      for (int kk = 0; kk < TF_NB / 4; kk++) {
        const int i = wave * 16 + y, k = kq + 4 * kk;
        av[kk] = hssk_gload(Ti, p.transT ? (size_t)k + (size_t)i * TF_NB : (size_t)i + (size_t)k * TF_NB);
      }
      hssk_d4 acc = {0., 0., 0., 0.};
#pragma unroll
      for (int kk = 0; kk < TF_NB / 4; kk++) acc = hssk_mfma_f64_16x16x4(av[kk], s_x[(b0 + kq + 4 * kk) + y * ldx], acc);
      __syncthreads();   // (every wave has read all of B_b)
#pragma unroll
      for (int r = 0; r < 4; r++) s_x[(b0 + wave * 16 + kq + 4 * r) + y * ldx] = acc[r];
      __syncthreads();
    }
    // ---- the block rows still to come -= op(T)(rows, b) X_b: 16-row tiles over the waves
    const int r_lo = fwd ? b0 + TF_NB : 0, r_hi = fwd ? n : b0;   // rows [r_lo, r_hi)
    const int ntile = r_hi > r_lo ? (r_hi - r_lo + 15) / 16 : 0;
    for (int t = wave; t < ntile; t += TF_T / 64) {
      const int row = r_lo + t * 16 + y, rowc = min(row, n - 1);
      double av[TF_NB / 4];
#pragma unroll
      for (int kk = 0; kk < TF_NB / 4; kk++) {
        const int k = b0 + kq + 4 * kk, kc = min(k, n - 1);
        // op(T)(row, k): lower / upper, not transposed: T(row, k); transposed upper: T(k, row)
        const double v = hssk_gload(T, p.transT ? (size_t)kc + (size_t)rowc * ldt : (size_t)rowc + (size_t)kc * ldt);
        av[kk] = (k < n && row < n) ? -v : 0.;
      }
      hssk_d4 acc;
#pragma unroll
      for (int r = 0; r < 4; r++) acc[r] = s_x[(r_lo + t * 16 + kq + 4 * r) + y * ldx];
#pragma unroll
      for (int kk = 0; kk < TF_NB / 4; kk++) acc = hssk_mfma_f64_16x16x4(av[kk], s_x[(b0 + kq + 4 * kk) + y * ldx], acc);
#pragma unroll
      for (int r = 0; r < 4; r++) s_x[(r_lo + t * 16 + kq + 4 * r) + y * ldx] = acc[r];
    }
    __syncthreads();
  }
  for (int e = tid; e < n * nc; e += TF_T) {
    const int i = e % n, c = e / n;
    hssk_gstore(p.B, (size_t)i + (size_t)(c0 + c) * p.ldb, s_x[i + c * ldx]);
  }
}

// Large triangles with many right-hand sides (the triangular solves of a BLR block step: a 256 x 256 diagonal tile against
// the few hundred columns of a block row's U factors, or against every V factor of a block column): the substitution kernel
// below walks n dependent steps per group of four right-hand sides -- 0.29 ms per call for n = 256.  Blocked instead, as in
// the ULV sweeps: the 64 x 64 diagonal blocks are inverted once (hssk_trtri_diag_vbatched) and the solve becomes
// ceil(n / 64) block steps of two batched MFMA GEMMs over ALL right-hand sides -- X_b = inv(T_bb) B_b, then the remaining
// block rows -= T(rest, b) X_b.  Forms taken: unit lower, upper, transposed upper (the ones the factorizations use);
// everything else stays with the substitution kernel.
static bool trsm_blocked(hssk_ctx* ctx, const hssk_trsm_desc* descs, int count, std::vector<char>& done) {
  constexpr int NB = 64;
  struct Tri { const double* T; int n, ldt, lower, transT, unit; size_t off; const double* given; const double* inv; };
  std::vector<Tri> tris;            // distinct triangles (a block column's V factors all meet the same diagonal tile)
  std::vector<int> tri_of(count, -1);
  size_t inv_doubles = 0, tmp_doubles = 0;
  for (int i = 0; i < count; i++) {
    const hssk_trsm_desc& d = descs[i];
    const bool form = (d.lower && !d.transT && d.unit) || (!d.lower && !d.unit);
    if (!form || d.n < 2 * NB || d.nrhs < 1) continue;   // (a 128-step substitution costs more than the block's inverse, whatever the number of right-hand sides)
    int t = -1;
    for (size_t q = 0; q < tris.size(); q++)
      if (tris[q].T == d.T && tris[q].n == d.n && tris[q].ldt == d.ldt && tris[q].lower == d.lower && tris[q].transT == d.transT && tris[q].unit == d.unit && tris[q].given == d.Tinv) { t = (int)q; break; }
    if (t < 0) {
      t = (int)tris.size();
      tris.push_back(Tri{d.T, d.n, d.ldt, d.lower, d.transT, d.unit, inv_doubles, d.Tinv, nullptr});
      if (!d.Tinv) inv_doubles += (size_t)((d.n + NB - 1) / NB) * NB * NB;
    }
    tri_of[i] = t;
    tmp_doubles += (size_t)NB * d.nrhs;
  }
  if (tris.empty()) return false;
  // While a plan is being recorded the launches below keep these addresses for every replay, and ctx->aux() frees and
  // reallocates when a later call needs more (more right-hand sides): the recorded plan gets storage of its own instead.
  double* aux;
  if (ctx->recording) {
    char* shadow = nullptr;
    aux = (double*)ctx->recording->alloc(sizeof(double) * (inv_doubles + tmp_doubles), &shadow);
  } else {
    aux = ctx->aux(sizeof(double) * (inv_doubles + tmp_doubles));
  }
  double* tmp0 = aux + inv_doubles;
  std::vector<hssk_trtri_desc> ti;
  for (auto& t : tris) {
    t.inv = t.given ? t.given : aux + t.off;
    if (!t.given) ti.push_back(hssk_trtri_desc{t.T, aux + t.off, t.n, t.ldt, t.lower ? 2 : 1});
  }
  if (!ti.empty() && hssk_trtri_diag_vbatched(ctx, ti.data(), (int)ti.size())) throw std::runtime_error(hssk_last_error());
  int nblk_max = 0, n_max = 0;
  for (auto& t : tris) { nblk_max = std::max(nblk_max, (t.n + NB - 1) / NB); n_max = std::max(n_max, t.n); }
  static const bool no_fused = [] { const char* e = std::getenv("HSSK_TRSM_NO_FUSED"); return e && e[0] == '1'; }();   // (A/B: one launch per block step and stage)
  // (the one-launch form keeps 16 right-hand sides of up to 512 rows in the LDS: 66 KB -- on a device whose workgroups get
  // less, the block steps below take over)
  static const size_t lds_cap = hssk_rt::max_lds_per_workgroup();
  if (n_max <= TF_NMAX && !no_fused && sizeof(double) * 16 * (size_t)(nblk_max * NB + 1) <= lds_cap) {
    // all block steps in one launch: a workgroup per 16 right-hand sides
    std::vector<hssk_trsm_desc> sel;
    std::vector<TfWork> work;
    for (int i = 0; i < count; i++) {
      if (tri_of[i] < 0) continue;
      for (int g = 0; g * 16 < descs[i].nrhs; g++) work.push_back(TfWork{(int)sel.size(), g, tris[tri_of[i]].inv});
      sel.push_back(descs[i]);
    }
    auto* dd = (const hssk_trsm_desc*)ctx->stage(sel.data(), sizeof(hssk_trsm_desc) * sel.size());
    auto* dw = (const TfWork*)ctx->stage(work.data(), sizeof(TfWork) * work.size());
    const size_t shmem = sizeof(double) * 16 * (size_t)(nblk_max * NB + 1);
    hssk_rt::allow_dynamic_lds(trsm_fused_kernel, shmem);
    HSSK_LAUNCH(trsm_fused_kernel, dim3((unsigned)work.size()), dim3(TF_T), shmem, ctx->stream, dd, dw);
    for (int i = 0; i < count; i++) done[i] = tri_of[i] >= 0;
    return true;
  }
  std::vector<hssk_rowgather_desc> cp;
  std::vector<hssk_gemm_desc> g1, g2;
  for (int s = 0; s < nblk_max; s++) {
    cp.clear(); g1.clear(); g2.clear();
    size_t toff = 0;
    for (int i = 0; i < count; i++) {
      if (tri_of[i] < 0) continue;
      const hssk_trsm_desc& d = descs[i];
      const Tri& t = tris[tri_of[i]];
      double* tmp = tmp0 + toff;
      toff += (size_t)NB * d.nrhs;
      const int nblk = (d.n + NB - 1) / NB;
      if (s >= nblk) continue;
      // the effective triangle of op(T): lower (forward, blocks first to last) or upper (backward, last to first)
      const bool fwd = d.lower || d.transT;
      const int b = fwd ? s : nblk - 1 - s, b0 = b * NB, nb = std::min(NB, d.n - b0), end = b0 + nb;
      const double* Ti = t.inv + (size_t)b * NB * NB;
      double* Bb = d.B + b0;
      cp.push_back(hssk_rowgather_desc{Bb, tmp, nullptr, nb, d.nrhs, d.ldb, nb, 0, 0});
      // X_b = inv(op(T)_bb) B_b: the blocks hold plain inverses of T_bb; a transposed solve multiplies by their transposes
      g1.push_back(hssk_gemm_desc{Ti, tmp, Bb, nb, d.nrhs, nb, NB, nb, d.ldb, d.transT ? 1 : 0, 0, 1.0, 0.0});
      if (fwd && end < d.n) {
        // rows below -= op(T)(end:n, b0:end) X_b:  lower T: T(end:n, b0:end);  transposed upper: T(b0:end, end:n)^T
        if (d.lower) g2.push_back(hssk_gemm_desc{d.T + end + (size_t)b0 * d.ldt, Bb, d.B + end, d.n - end, d.nrhs, nb, d.ldt, d.ldb, d.ldb, 0, 0, -1.0, 1.0});
        else g2.push_back(hssk_gemm_desc{d.T + b0 + (size_t)end * d.ldt, Bb, d.B + end, d.n - end, d.nrhs, nb, d.ldt, d.ldb, d.ldb, 1, 0, -1.0, 1.0});
      } else if (!fwd && b0 > 0) {
        // rows above -= T(0:b0, b0:end) X_b
        g2.push_back(hssk_gemm_desc{d.T + (size_t)b0 * d.ldt, Bb, d.B, b0, d.nrhs, nb, d.ldt, d.ldb, d.ldb, 0, 0, -1.0, 1.0});
      }
    }
    if (!cp.empty() && hssk_gather_rows(ctx, cp.data(), (int)cp.size())) throw std::runtime_error(hssk_last_error());
    if (!g1.empty() && hssk_gemm_vbatched(ctx, g1.data(), (int)g1.size())) throw std::runtime_error(hssk_last_error());
    if (!g2.empty() && hssk_gemm_vbatched(ctx, g2.data(), (int)g2.size())) throw std::runtime_error(hssk_last_error());
  }
  for (int i = 0; i < count; i++) done[i] = tri_of[i] >= 0;
  return true;
}

int hssk_trsm_vbatched(hssk_ctx* ctx, const hssk_trsm_desc* descs_in, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  static const bool no_blocked = [] { const char* e = std::getenv("HSSK_TRSM_NO_BLOCKED"); return e && e[0] == '1'; }();   // (A/B)
  std::vector<char> done(count, 0);
  std::vector<hssk_trsm_desc> rest;
  const hssk_trsm_desc* descs = descs_in;
  if (!no_blocked && trsm_blocked(ctx, descs_in, count, done)) {
    for (int i = 0; i < count; i++) if (!done[i]) rest.push_back(descs_in[i]);
    if (rest.empty()) { hssk_rt::check_launch(); return 0; }
    descs = rest.data();
    count = (int)rest.size();
  }
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

bool hssk_getrf_row_launch(hssk_ctx* ctx, const hssk_lu_desc* dd, int count, int nmax);   // hssk_lu_row.hip
int hssk_getrf_vbatched(hssk_ctx* ctx, const hssk_lu_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  auto* dd = (const hssk_lu_desc*)ctx->stage(descs, sizeof(*descs) * count);
  int nmax = 0;
  for (int i = 0; i < count; i++) nmax = std::max(nmax, descs[i].n);
  static const bool no_wg = [] { const char* e = std::getenv("HSSK_LU_NO_WG"); return e && e[0] == '1'; }();   // (A/B: the first kernel)
  static const bool wg_v1 = [] { const char* e = std::getenv("HSSK_LU_WG_V1"); return e && e[0] == '1'; }();   // (A/B: the first one-workgroup form)
  // the one-workgroup kernels hold a 32-column panel and a block of U12 in up to 148 KB of LDS: where a workgroup cannot get
  // that much (gfx90a / gfx942: 64 KB) the global-memory kernels below serve every size
  static const size_t lds_cap = hssk_rt::max_lds_per_workgroup();
  const bool wg_fits = sizeof(double) * ((size_t)(nmax | 1) * LUW_NB + (size_t)(LUW_NB + 1) * (nmax <= 256 ? LUW_CHW_SMALL : LUW_CH)) <= lds_cap;
  if (hssk_getrf_row_launch(ctx, dd, count, nmax)) {
    // (up to 192 rows: the whole matrix in the registers of one workgroup, hssk_lu_row.hip)
  } else if (nmax <= LU_LDS_N) {
    HSSK_LAUNCH(getrf_lds_kernel, dim3((unsigned)count), dim3(LU_THREADS), 0, ctx->stream, dd);
  } else if (!wg_fits && nmax <= 384) {
    HSSK_LAUNCH(getrf_kernel, dim3((unsigned)count), dim3(LU_THREADS), 0, ctx->stream, dd);
  } else if (nmax <= LUW_NMAX && wg_fits && !no_wg && !wg_v1) {
    const int chw = nmax <= 256 ? LUW_CHW_SMALL : LUW_CH;
    const size_t shmem = sizeof(double) * ((size_t)(nmax | 1) * LUW_NB + (size_t)(LUW_NB + 1) * chw);
    if (nmax <= 256) {
      hssk_rt::allow_dynamic_lds(getrf_wg2_kernel<256>, shmem);
      HSSK_LAUNCH(getrf_wg2_kernel<256>, dim3((unsigned)count), dim3(256), shmem, ctx->stream, dd, chw);
    } else {
      hssk_rt::allow_dynamic_lds(getrf_wg2_kernel<512>, shmem);
      HSSK_LAUNCH(getrf_wg2_kernel<512>, dim3((unsigned)count), dim3(512), shmem, ctx->stream, dd, chw);
    }
  } else if (nmax <= LUW_NMAX && wg_fits && !no_wg) {
    const size_t shmem = sizeof(double) * ((size_t)(nmax | 1) * LUW_NB + (size_t)(LUW_NB + 1) * LUW_CH);
    hssk_rt::allow_dynamic_lds(getrf_wg_kernel, shmem);
    HSSK_LAUNCH(getrf_wg_kernel, dim3((unsigned)count), dim3(LUW_T), shmem, ctx->stream, dd);
  } else if (nmax <= 384) {
    HSSK_LAUNCH(getrf_kernel, dim3((unsigned)count), dim3(LU_THREADS), 0, ctx->stream, dd);
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
