// Batched interpolative decomposition: tolerance-truncated column-pivoted Householder QR + the
// triangular solve R11^{-1} R12, one workgroup per HSS node.
//
// Reference behaviour restated: DenseMatrix::ID_row -> ID_column_GEQP3 (dense/DenseMatrix.cpp:746-790)
// -> geqp3tol (dense/lapack/dgeqp3tol.f: LAPACK dgeqp3 that stops at the first diagonal entry with
// |R_cc|/|R_00| <= rtol or |R_cc| <= atol, :203-232; column norms are supplied exact,
// dense/BLASLAPACKWrapper.hpp:579-591) -> trsm.  The HSS engine keeps the samples transposed
// (d x m, one contiguous length-d column per candidate row), which is exactly the operand QRCP wants,
// so the reference's two explicit transposes disappear.
//
// Parallelisation (wave64): pivot search = per-thread scan + wave shuffle arg-max + 1 LDS hop;
// Householder vector = one wave, shuffle reductions; trailing update = one column per wave, the
// 64 lanes stride the column (coalesced), dot product by shuffle reduction; partial column norms
// are down-dated with LAPACK's dlaqp2 safeguard (recompute when cancellation is detected).
// Only rank+1 Householder steps are executed.  Bound: L2/LDS latency (Level-2 BLAS on a panel that
// lives in L2); flops are negligible.
#include "hssk_device.h"
#include "hssk_backsub.h"
#include "hssk_internal.h"
#include "hssk_id_reg.h"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>
#include <atomic>
#include <thread>
#include <vector>

namespace {

constexpr int ID_THREADS = 512;
constexpr int ID_WAVES = ID_THREADS / 64;

__global__ __launch_bounds__(ID_THREADS) void id_kernel(const hssk_id_desc* __restrict__ descs) {
  HSSK_SHARED double s_val[ID_WAVES];
  HSSK_SHARED int s_idx[ID_WAVES];
  HSSK_SHARED double s_tau, s_r00;
  HSSK_SHARED int s_piv, s_stop;

  const hssk_id_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = p.d, m = p.m, ld = p.ldw;
  double* __restrict__ W = p.W;
  double* vn1 = p.work;
  double* vn2 = p.work + m;
  // (steps beyond max_rank cannot change the result: the rank is cut there anyway -- a BLR tile stops at the rank that no longer pays)
  const int kmax = min(d < m ? d : m, p.max_rank > 0 ? p.max_rank : 0);
  const double tol3z = 1.4901161193847656e-08;  // sqrt(eps)

  for (int j = tid; j < m; j += ID_THREADS) p.perm[j] = j;
  for (int j = wave; j < m; j += ID_WAVES) {
    double s = 0.;
    for (int i = lane; i < d; i += 64) { double v = W[i + (size_t)j * ld]; s += v * v; }
    s = hssk_wave_sum(s);
    if (lane == 0) { vn1[j] = sqrt(s); vn2[j] = sqrt(s); }
  }
  if (tid == 0) { s_stop = 0; s_r00 = 0.; }
  __syncthreads();

  int rank = kmax;
  for (int k = 0; k < kmax; k++) {
    // ---- 1. pivot = first arg max_{j >= k} vn1[j]
    double bv = -1.;
    int bi = 0x7fffffff;
    for (int j = k + tid; j < m; j += ID_THREADS) {
      double v = vn1[j];
      if (v > bv) { bv = v; bi = j; }
    }
    hssk_wave_argmax(bv, bi);
    if (lane == 0) { s_val[wave] = bv; s_idx[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      double v = s_val[0];
      int ix = s_idx[0];
      for (int w = 1; w < ID_WAVES; w++)
        if (s_val[w] > v || (s_val[w] == v && s_idx[w] < ix)) { v = s_val[w]; ix = s_idx[w]; }
      s_piv = ix;
      if (ix != k) {
        int t = p.perm[k]; p.perm[k] = p.perm[ix]; p.perm[ix] = t;
        vn1[ix] = vn1[k]; vn2[ix] = vn2[k];  // (entries k are dead after this step)
      }
    }
    __syncthreads();
    const int pv = s_piv;
    // ---- 2. swap columns k <-> pv
    if (pv != k)
      for (int i = tid; i < d; i += ID_THREADS) {
        double a = W[i + (size_t)k * ld], b = W[i + (size_t)pv * ld];
        W[i + (size_t)k * ld] = b;
        W[i + (size_t)pv * ld] = a;
      }
    __syncthreads();
    // ---- 3. Householder reflector of W[k:d, k] (dlarfg)
    if (wave == 0) {
      double* col = W + (size_t)k * ld;
      double s = 0.;
      for (int i = k + 1 + lane; i < d; i += 64) { double v = col[i]; s += v * v; }
      const double alpha = col[k];  // read by every lane BEFORE the collective (lane 0 overwrites it)
      s = hssk_wave_sum(s);
      double tau = 0., beta = alpha;
      if (s != 0.) {
        double nrm = sqrt(alpha * alpha + s);
        beta = alpha >= 0. ? -nrm : nrm;
        tau = (beta - alpha) / beta;
        double scal = 1. / (alpha - beta);
        for (int i = k + 1 + lane; i < d; i += 64) col[i] *= scal;
      }
      if (lane == 0) {
        col[k] = beta;
        s_tau = tau;
        double ab = fabs(beta);
        if (k == 0) s_r00 = ab;
        double r00 = (k == 0) ? ab : s_r00;
        // dgeqp3tol.f:225-232 (0/0 is NaN -> false, then the absolute test decides)
        if ((r00 != 0. && ab / r00 <= p.rtol) || ab <= p.atol) s_stop = 1;
      }
    }
    __syncthreads();
    if (s_stop) { rank = k; break; }
    const double tau = s_tau;
    // ---- 4. apply H = I - tau v v^T (v = [1; W[k+1:d,k]]) to columns j > k, 5. down-date norms
    {
      const double* v = W + (size_t)k * ld;
      for (int j = k + 1 + wave; j < m; j += ID_WAVES) {
        double* col = W + (size_t)j * ld;
        double s = 0.;
        for (int i = k + 1 + lane; i < d; i += 64) s += v[i] * col[i];
        // scalars shared by the wave are read before the collective; lane 0 rewrites them after it
        const double ckj = col[k];
        const double n1 = vn1[j], n2 = vn2[j];
        s = hssk_wave_sum(s);
        double dot = tau * (ckj + s);
        for (int i = k + 1 + lane; i < d; i += 64) col[i] -= dot * v[i];
        double newk = ckj - dot;
        if (lane == 0) col[k] = newk;
        // dlaqp2 norm down-date
        int recompute = 0;
        double newn1 = n1;
        if (n1 != 0.) {
          double t = fabs(newk) / n1;
          t = (1. + t) * (1. - t);
          t = t > 0. ? t : 0.;
          double q = n1 / n2;
          double t2 = t * q * q;
          if (t2 <= tol3z) recompute = 1;
          else newn1 = n1 * sqrt(t);
        }
        if (recompute) {  // wave-uniform: every lane evaluated the same scalars
          double s2 = 0.;
          for (int i = k + 1 + lane; i < d; i += 64) { double x = col[i]; s2 += x * x; }
          s2 = hssk_wave_sum(s2);
          newn1 = sqrt(s2);
          if (lane == 0) vn2[j] = newn1;
        }
        if (lane == 0) vn1[j] = newn1;
      }
    }
    __syncthreads();
  }
  if (rank > p.max_rank) rank = p.max_rank;
  // ---- X = R11^{-1} R12 in place: one column per thread, back substitution
  for (int j = rank + tid; j < m; j += ID_THREADS) {
    double* x = W + (size_t)j * ld;
    for (int i = rank - 1; i >= 0; i--) {
      double s = x[i];
      for (int l = i + 1; l < rank; l++) s -= W[i + (size_t)l * ld] * x[l];
      x[i] = s / W[i + (size_t)i * ld];
    }
  }
  if (tid == 0) *p.rank = rank;
}

// ------------------------------------------------------------------------------------------------
// Streaming variant for panels beyond the register kernels (more than 224 columns or 256 rows: leaf size 512 -- 192 x 391
// sample panels --, 256 x 256 BLR tiles): the panel stays in global memory (L2 / MALL resident: 0.3 - 1 MB), one 16-wave
// workgroup per panel.  Same truncated QRCP as id_kernel, organised for memory-level parallelism: id_kernel walks the
// trailing columns one at a time per wave -- load, reduce, load again, store: two dependent round trips per column, 50 us
// per Householder step on a 256 x 256 tile --; here a wave takes G columns at once, all their elements (its lane's RL rows
// of each) in flight together and kept in registers between the dot product and the update (one read and one write of the
// trailing panel per step), the G reductions interleaved on the DPP network, the Householder vector held in registers for
// the whole step.  Pivot search, column swap and reflector are one wave's work between two barriers.  The triangular
// solve X = R11^{-1} R12 is a launch of its own (id_xsolve_all_kernel: 64 registers per thread next to this kernel's
// tile would spill).
// Capacity: d <= 64 RL (RL = 4 or 8), m <= IDS_MMAX.
// ------------------------------------------------------------------------------------------------
constexpr int IDS_T = 1024;
constexpr int IDS_W = IDS_T / 64;
constexpr int IDS_MMAX = 2048;
template <int RL>
__global__ __launch_bounds__(IDS_T) void id_stream_kernel(const hssk_id_desc* __restrict__ descs) {
  constexpr int G = 32 / RL;   // columns a wave keeps in flight (32 doubles of panel per lane)
  HSSK_SHARED double s_v[64 * RL];
  HSSK_SHARED double s_vn1[IDS_MMAX];
  HSSK_SHARED double s_vn2[IDS_MMAX];
  HSSK_SHARED double s_tau, s_r00;
  HSSK_SHARED int s_stop;

  const hssk_id_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = p.d, m = p.m, ld = p.ldw;
  double* __restrict__ W = p.W;
  // (steps beyond max_rank cannot change the result: the rank is cut there anyway -- a BLR tile stops at the rank that no longer pays)
  const int kmax = min(d < m ? d : m, p.max_rank > 0 ? p.max_rank : 0);
  const double tol3z = 1.4901161193847656e-08;  // sqrt(eps)

  for (int j = tid; j < m; j += IDS_T) p.perm[j] = j;
  // exact column norms (BLASLAPACKWrapper.hpp:579-591), G columns per wave at a time
  for (int j0 = wave * G; j0 < m; j0 += IDS_W * G) {
    double a[G][RL];
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
      for (int t = 0; t < RL; t++) {
        const int i = lane + 64 * t;
        a[g][t] = (j0 + g < m && i < d) ? hssk_gload(W, i + (size_t)(j0 + g) * ld) : 0.;
      }
#pragma unroll
    for (int g = 0; g < G; g++) {
      double sq = 0.;
#pragma unroll
      for (int t = 0; t < RL; t++) sq += a[g][t] * a[g][t];
      sq = hssk_wave_sum(sq);
      if (lane == 0 && j0 + g < m) { s_vn1[j0 + g] = sqrt(sq); s_vn2[j0 + g] = sqrt(sq); }
    }
  }
  if (tid == 0) { s_stop = 0; s_r00 = 0.; }
  __syncthreads();

  int rank = kmax;
  for (int k = 0; k < kmax; k++) {
    const int tk = k >> 6, lk = k & 63;
    if (wave == 0) {
      // ---- 1. pivot = first arg max_{j >= k} vn1[j]
      double bv = -1.;
      int bi = 0x7fffffff;
      for (int j = k + lane; j < m; j += 64) {
        const double v = s_vn1[j];
        if (v > bv) { bv = v; bi = j; }
      }
      hssk_wave_argmax(bv, bi);
      const int pv = bi;
      // ---- 2. swap columns k <-> pv (the new column k stays in registers for the reflector)
      double ck[RL], cp[RL];
#pragma unroll
      for (int t = 0; t < RL; t++) {
        const int i = lane + 64 * t;
        cp[t] = i < d ? hssk_gload(W, i + (size_t)pv * ld) : 0.;
        ck[t] = (pv != k && i < d) ? hssk_gload(W, i + (size_t)k * ld) : 0.;
      }
      if (pv != k) {
#pragma unroll
        for (int t = 0; t < RL; t++) {
          const int i = lane + 64 * t;
          if (i < d) hssk_gstore(W, i + (size_t)pv * ld, ck[t]);
        }
        if (lane == 0) {
          const int tp = p.perm[k]; p.perm[k] = p.perm[pv]; p.perm[pv] = tp;
          s_vn1[pv] = s_vn1[k]; s_vn2[pv] = s_vn2[k];   // (entries k are dead after this step)
        }
      }
      // ---- 3. Householder reflector of the pivot column's rows k .. d (dlarfg)
      double sq = 0., alpha_l = 0.;
#pragma unroll
      for (int t = 0; t < RL; t++) {
        const int i = lane + 64 * t;
        if (i > k) sq += cp[t] * cp[t];
        if (i == k) alpha_l = cp[t];
      }
      sq = hssk_wave_sum(sq);
      const double alpha = hssk_bcast_lane(alpha_l, lk);
      double tau = 0., beta = alpha, scal = 0.;
      if (sq != 0.) {
        const double nrm = sqrt(alpha * alpha + sq);
        beta = alpha >= 0. ? -nrm : nrm;
        tau = (beta - alpha) / beta;
        scal = 1. / (alpha - beta);
      }
#pragma unroll
      for (int t = 0; t < RL; t++) {
        const int i = lane + 64 * t;
        const double vi = i < k ? 0. : (i == k ? 1. : cp[t] * scal);   // (sq == 0: the rows below are zero already)
        s_v[i] = i < d ? vi : 0.;
        // column k of the panel: R(0:k, k) as it was, R(k, k) = beta, the reflector below (not read again)
        if (i < d) hssk_gstore(W, i + (size_t)k * ld, i < k ? cp[t] : (i == k ? beta : vi));
      }
      if (lane == 0) {
        s_tau = tau;
        const double ab = fabs(beta);
        if (k == 0) s_r00 = ab;
        const double r00 = (k == 0) ? ab : s_r00;
        // dgeqp3tol.f:225-232 (0/0 is NaN -> false, then the absolute test decides)
        if ((r00 != 0. && ab / r00 <= p.rtol) || ab <= p.atol) s_stop = 1;
      }
    }
    __syncthreads();
    if (s_stop) { rank = k; break; }
    const double tau = s_tau;
    // ---- 4. apply H = I - tau v v^T to the columns j > k (rows k .. d), 5. down-date their norms
    double vr[RL];
#pragma unroll
    for (int t = 0; t < RL; t++) vr[t] = s_v[lane + 64 * t];
    for (int j0 = k + 1 + wave * G; j0 < m; j0 += IDS_W * G) {
      double a[G][RL];
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int t = 0; t < RL; t++) {
          const int i = lane + 64 * t;
          a[g][t] = (t >= tk && j0 + g < m && i >= k && i < d) ? hssk_gload(W, i + (size_t)(j0 + g) * ld) : 0.;
        }
      double dot[G];
#pragma unroll
      for (int g = 0; g < G; g++) {
        dot[g] = 0.;
#pragma unroll
        for (int t = 0; t < RL; t++) dot[g] += vr[t] * a[g][t];
      }
#pragma unroll
      for (int g = 0; g < G; g++) dot[g] = tau * hssk_wave_sum(dot[g]);
#pragma unroll
      for (int g = 0; g < G; g++) {
        double sq = 0., newk_l = 0.;
#pragma unroll
        for (int t = 0; t < RL; t++) {
          const int i = lane + 64 * t;
          a[g][t] -= dot[g] * vr[t];
          if (t >= tk && j0 + g < m && i >= k && i < d) hssk_gstore(W, i + (size_t)(j0 + g) * ld, a[g][t]);
          if (i > k) sq += a[g][t] * a[g][t];
          if (i == k) newk_l = a[g][t];
        }
        if (j0 + g < m) {   // (wave-uniform)
          const int j = j0 + g;
          // (scalars shared by the wave are read BEFORE the collective; lane 0 rewrites them after it)
          const double n1 = s_vn1[j], n2 = s_vn2[j];
          const double newk = hssk_bcast_lane(newk_l, lk);
          // dlaqp2 norm down-date
          int recompute = 0;
          double newn1 = n1;
          if (n1 != 0.) {
            double tt = fabs(newk) / n1;
            tt = (1. + tt) * (1. - tt);
            tt = tt > 0. ? tt : 0.;
            const double q = n1 / n2;
            const double t2 = tt * q * q;
            if (t2 <= tol3z) recompute = 1;
            else newn1 = n1 * sqrt(tt);
          }
          if (recompute) {  // wave-uniform: every lane evaluated the same scalars; the updated rows are still in registers
            newn1 = sqrt(hssk_wave_sum(sq));
            if (lane == 0) s_vn2[j] = newn1;
          }
          if (lane == 0) s_vn1[j] = newn1;
        }
      }
    }
    __syncthreads();
  }
  if (rank > p.max_rank) rank = p.max_rank;
  if (tid == 0) *p.rank = rank;
}

// X = R11^{-1} R12 in place for the panels of a register-kernel launch whose rank came out <= 64 (W holds the pivoted,
// factored panel: R11 = W(0:rank, 0:rank), R12 the columns behind it; larger ranks were finished by id_reg_kernel
// itself).  One workgroup per panel, one thread per column of R12 with the column in registers; R11 is staged in LDS and
// read as broadcasts; the back substitution is unrolled over 8-row blocks (blocks at or above the rank are skipped by
// uniform branches), so every register index is static and a column costs rank^2 / 2 fmas with no cross-lane step.
// A launch of its own: the factorization kernel has no registers to spare next to its tile.  (The first version solved
// inside id_reg_kernel, a column per wave and a row per lane, and paid a 64-lane reduction per row: 185 us of the
// 390 us of a 192 x 195 leaf panel at rank 36.)
constexpr int XS_T = 256;
// body shared by the two front ends below: X (ldx) = R11^{-1} R12 of the factored panel W (ldw)
__device__ __forceinline__ void xsolve_body(const double* W, int ld, int rank, int m, double* X, int ldx,   // (X may lie over R12)
                                            double* s_R, double* s_rd) {
  constexpr int LR = HSSK_BACKSUB_LD;
  const int tid = threadIdx.x;
  if (rank <= 0 || rank >= m) return;
  if (rank > 64) {
    // large ranks (BLR tiles next to the diagonal, kernel matrices): block rows of 64 from the bottom.  A thread owns a
    // column of X; per block row it subtracts the rows already solved -- the 64 x 64 pieces of R11 staged in LDS and read
    // as broadcasts, the solved entries read back from the thread's own column -- and finishes with the register back
    // substitution on the diagonal block.  (The first version ran the whole substitution from global memory, a scalar
    // load per multiply: a rank-127 tile cost more than its factorization.)
    const int nblk = (rank + 63) / 64;
    for (int jp = 0; jp < m - rank; jp += XS_T) {   // (uniform trip count: the stagings below contain barriers)
      const int j = rank + jp + tid;
      const bool act = j < m;
      const double* bcol = W + (size_t)(act ? j : rank) * ld;
      double* xcol = X + (size_t)(act ? j - rank : 0) * ldx;
      for (int b = nblk - 1; b >= 0; b--) {
        const int i0 = 64 * b, nb = min(64, rank - i0);
        double x[64];
#pragma unroll
        for (int i = 0; i < 64; i++) x[i] = (act && i < nb) ? bcol[i0 + i] : 0.;
        for (int lc = i0 + 64; lc < rank; lc += 64) {
          const int nl = min(64, rank - lc);
          __syncthreads();
          for (int e = tid; e < 64 * 64; e += XS_T) {
            const int i = e & 63, l = e >> 6;
            s_R[i + l * LR] = (i < nb && l < nl) ? W[(i0 + i) + (size_t)(lc + l) * ld] : 0.;
          }
          __syncthreads();
          for (int l = 0; l < nl; l++) {
            const double xl = act ? xcol[lc + l] : 0.;
#pragma unroll
            for (int i = 0; i < 64; i++) x[i] -= s_R[i + l * LR] * xl;
          }
        }
        __syncthreads();
        for (int e = tid; e < 64 * 64; e += XS_T) {
          const int i = e & 63, l = e >> 6;
          s_R[i + l * LR] = (i < l && l < nb) ? W[(i0 + i) + (size_t)(i0 + l) * ld] : 0.;
        }
        if (tid < 64) s_rd[tid] = tid < nb ? 1. / W[(i0 + tid) + (size_t)(i0 + tid) * ld] : 0.;
        __syncthreads();
        hssk_backsub64(x, s_R, s_rd, nb);
        if (act) {
#pragma unroll
          for (int i = 0; i < 64; i++)
            if (i < nb) xcol[i0 + i] = x[i];
        }
      }
    }
    return;
  }
  for (int e = tid; e < 64 * 64; e += XS_T) {
    const int i = e & 63, l = e >> 6;
    s_R[i + l * LR] = (i < l && l < rank) ? W[i + (size_t)l * ld] : 0.;
  }
  if (tid < 64) s_rd[tid] = tid < rank ? 1. / W[tid + (size_t)tid * ld] : 0.;
  __syncthreads();
  for (int j = rank + tid; j < m; j += XS_T) {
    const double* bcol = W + (size_t)j * ld;
    double* xcol = X + (size_t)(j - rank) * ldx;
    double x[64];
#pragma unroll
    for (int b = 0; b < 8; b++) {
      if (8 * b < rank) {
#pragma unroll
        for (int i = 8 * b; i < 8 * b + 8; i++) x[i] = i < rank ? bcol[i] : 0.;
      } else {
#pragma unroll
        for (int i = 8 * b; i < 8 * b + 8; i++) x[i] = 0.;
      }
    }
    hssk_backsub64(x, s_R, s_rd, rank);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      if (8 * b < rank) {
#pragma unroll
        for (int i = 8 * b; i < 8 * b + 8; i++)
          if (i < rank) xcol[i] = x[i];
      }
    }
  }
}
// in place, right behind a register-kernel launch: rank from device memory, X over R12 (ranks above 64 were finished by
// id_reg_kernel itself; panels with defer_x set are left to hssk_id_xsolve_vbatched)
__global__ __launch_bounds__(XS_T) HSSK_WAVES_PER_SIMD(2) void id_xsolve_kernel(const hssk_id_desc* __restrict__ descs) {
  HSSK_SHARED double s_R[64 * HSSK_BACKSUB_LD];
  HSSK_SHARED double s_rd[64];
  const hssk_id_desc p = descs[blockIdx.x];
  const int rank = *p.rank;
  if (rank > 64 || p.defer_x) return;
  xsolve_body(p.W, p.ldw, rank, p.m, p.W + (size_t)rank * p.ldw, p.ldw, s_R, s_rd);
}
// behind id_stream_kernel: every rank, always in place (hssk_id_solves_inline() == 1 for these shapes)
__global__ __launch_bounds__(XS_T) HSSK_WAVES_PER_SIMD(2) void id_xsolve_all_kernel(const hssk_id_desc* __restrict__ descs) {
  HSSK_SHARED double s_R[64 * HSSK_BACKSUB_LD];
  HSSK_SHARED double s_rd[64];
  const hssk_id_desc p = descs[blockIdx.x];
  xsolve_body(p.W, p.ldw, *p.rank, p.m, p.W + (size_t)(*p.rank) * p.ldw, p.ldw, s_R, s_rd);
}
// deferred: rank and destination from the descriptor (hssk_id_xsolve_vbatched)
__global__ __launch_bounds__(XS_T) HSSK_WAVES_PER_SIMD(2) void id_xsolve_to_kernel(const hssk_xsolve_desc* __restrict__ descs) {
  HSSK_SHARED double s_R[64 * HSSK_BACKSUB_LD];
  HSSK_SHARED double s_rd[64];
  const hssk_xsolve_desc p = descs[blockIdx.x];
  if (p.solved) {
    for (int e = threadIdx.x; e < p.rank * (p.m - p.rank); e += XS_T) {
      const int i = e % p.rank, j = e / p.rank;
      p.X[i + (size_t)j * p.ldx] = p.W[i + (size_t)(p.rank + j) * p.ldw];
    }
    return;
  }
  xsolve_body(p.W, p.ldw, p.rank, p.m, p.X, p.ldx, s_R, s_rd);
}

// (id_reg_body: the register-resident factorization of one panel by one workgroup, hssk_id_reg.h)
template <int RT, int CT, int NW>
__global__ __launch_bounds__(NW * 64) HSSK_WAVES_PER_SIMD(NW / 4) void id_reg_kernel(const hssk_id_desc* __restrict__ descs) {
  const hssk_id_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x;
  const int m = p.m, ld = p.ldw;
  const int rank = id_reg_body<RT, CT, NW>(p);
  // ---- X = R11^{-1} R12 in place.  rank <= 64: left to id_xsolve_kernel (the next launch).  Larger ranks: one column
  // per thread from global memory.
  double* __restrict__ W = p.W;
  if (rank > 64 && !p.defer_x) {
    for (int j = rank + tid; j < m; j += NW * 64) {
      double* x = W + (size_t)j * ld;
      for (int i = rank - 1; i >= 0; i--) {
        double s = x[i];
        for (int l = i + 1; l < rank; l++) s -= W[i + (size_t)l * ld] * x[l];
        x[i] = s / W[i + (size_t)i * ld];
      }
    }
  }
  if (tid == 0) *p.rank = rank;
}

static thread_local bool g_all_deferred = false;   // (set by hssk_id_vbatched for the launch helpers below)
// ------------------------------------------------------------------------------------------------
// id_group_kernel: the register kernel above for panels that do not fit the registers of ONE workgroup (256 x 256 BLR tiles,
// the 192 x 391 sample panels of leaf size 512): H = 2 or 4 workgroups of 16 waves share a panel, each keeps its slice of
// the columns in registers for the whole factorization (the streaming kernel reads and writes the trailing panel through L2
// every step: 19 us per step on a 256 x 256 tile).  Per Householder step the workgroups exchange two small messages through
// device memory: their best pivot candidates (value, column), then -- from the owner of the pivot column -- the reflector,
// tau, |R_kk| and the stopping flag.  Every word of a message is written once with a coherent store and polled by its reader
// until it is no longer the sentinel the buffer was armed with (hssk_sweep_arm; one memory round trip per message, ~0.65 us
// across XCDs); step k has words of its own, so nothing is ever reset.  The workgroups of a panel are adjacent in the launch:
// in-order dispatch leaves at most the group at the dispatch frontier incomplete while all earlier groups are resident and
// finish, so a waiting workgroup is never starved; a bounded spin count turns a violation into an error code.
// Same decisions as id_reg_kernel (squared norms, first arg max with the smaller column index on ties, dlaqp2 down-date with
// its cancellation guard, the dgeqp3tol stopping rule), same outputs ([R11 R12] in the first `rank` rows at the pivoted column
// positions, perm, rank).
// ------------------------------------------------------------------------------------------------
constexpr unsigned long long IDG_SENTINEL = 0x7FF8DEADBEEF5EEDull;   // (the pattern hssk_sweep_arm writes)
constexpr long IDG_SPIN_LIMIT = 1L << 17;   // (~30 ms of polling: a partner that is merely not resident yet arrives within microseconds)
__device__ __forceinline__ double idg_take(const double* p, size_t off, int* err) {
  double v = hssk_cload(p, off);
  long spins = 0;
  while (hssk_bits(v) == IDG_SENTINEL) {
    hssk_pause();
    if (++spins > IDG_SPIN_LIMIT) { hssk_flag_raise(err); return 0.; }
    v = hssk_cload(p, off);
  }
  return v;
}
template <int RT, int H> constexpr int idg_step_words() { return 2 * H + 16 * RT + 3; }
// a[rk] of a register array, rk a wave-uniform run-time value: a one-hot combination (the weights are scalar selects, one
// fma per entry).  A chain of compare-selects is folded by the compiler into ONE load at a computed address, which moves the
// whole array from registers to scratch memory.
template <int RT> __device__ __forceinline__ double idg_pick(const double (&a)[RT], int rk) {
  double v = 0.;
#pragma unroll
  for (int r = 0; r < RT; r++) v = fma(a[r], r == rk ? 1. : 0., v);
  return v;
}

template <int RT, int CT, int NW, int H>
__global__ __launch_bounds__(NW * 64) HSSK_WAVES_PER_SIMD(NW / 4) void id_group_kernel(const hssk_id_desc* __restrict__ descs, double* __restrict__ xch,
                                                                              int kcap, int* err) {
  static_assert(NW == 8 || NW == 16, "one pivot candidate per lane of a 16-lane row");
  constexpr int NC = NW * 4, HC = NC * CT, NT = NW * 64;   // HC: columns a workgroup holds
  constexpr int SW = idg_step_words<RT, H>();
  HSSK_SHARED double s_v[16 * RT];
  HSSK_SHARED double s_vn1[HC];
  HSSK_SHARED double s_vn2[HC];
  HSSK_SHARED double s_val[NW];
  HSSK_SHARED int s_idx[NW];
  HSSK_SHARED double s_oval[H];
  HSSK_SHARED double s_oidx[H];
  HSSK_SHARED double s_msg[3];   // tau, stop, |R_kk|
  HSSK_SHARED double s_r00;
  HSSK_SHARED int s_perm[H * HC];
  HSSK_SHARED int s_pos[H * HC];
  const int panel = blockIdx.x / H, h = blockIdx.x % H;
  const hssk_id_desc p = descs[panel];
  double* xw = xch + (size_t)panel * kcap * SW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, sub = lane >> 4, grp = wave * 4 + sub;
  const int d = p.d, m = p.m, ld = p.ldw;
  const int kmax = min(min(d < m ? d : m, p.max_rank > 0 ? p.max_rank : 0), kcap);
  const double tol3z = 1.4901161193847656e-08;  // sqrt(eps)
  const double* __restrict__ in = p.src ? p.src : p.W;
  const int ldin = p.src ? p.lds : ld;
  const int c0 = h * HC;   // first column of this workgroup's slice
  double a[CT][RT];
#pragma unroll
  for (int c = 0; c < CT; c++) {
    const int lc = grp + NC * c, col = c0 + lc;
    double s = 0.;
#pragma unroll
    for (int r = 0; r < RT; r++) {
      const int row = l16 + 16 * r;
      a[c][r] = (row < d && col < m) ? hssk_gload(in, row + (size_t)col * ldin) : 0.;
      s += a[c][r] * a[c][r];
    }
    s = hssk_row_sum(s);
    if (l16 == 0) { s_vn1[lc] = s; s_vn2[lc] = s; }
  }
  if (tid == 0) s_r00 = 0.;
  unsigned used = 0;
  __syncthreads();

  int rank = kmax;
  // (one copy of the step for every 16-row register block `rk` -- the register kernel's form -- costs 2800 instructions per
  //  block and, at 256 rows, 350 bytes of spills per lane: 11.4 us per step.  Here rk is a run-time value: the entries of row
  //  k are picked out of the lane's RT registers of a column with compare-selects.)
  {
    for (int k = 0; k < kmax; k++) {
      const int rk = k >> 4, lk = k & 15;
      double* xs = xw + (size_t)k * SW;   // this step's words: [H values][H columns][16 RT reflector rows][tau, stop, |R_kk|]
      // ---- 1. pivot: this workgroup's first arg max over its unused columns ...
      {
        double bv = -1.;
        int bi = 0x7fffffff;
#pragma unroll
        for (int c = 0; c < CT; c++) {
          const int lc = grp + NC * c, col = c0 + lc;
          if (col < m && !((used >> c) & 1u)) {
            const double v = s_vn1[lc];
            if (v > bv) { bv = v; bi = col; }
          }
        }
        double wv = hssk_bcast_lane(bv, 0);
        int wi = hssk_bcast_lane_i(bi, 0);
#pragma unroll
        for (int q = 1; q < 4; q++) {
          const double v = hssk_bcast_lane(bv, 16 * q);
          const int ix = hssk_bcast_lane_i(bi, 16 * q);
          if (v > wv || (v == wv && ix < wi)) { wv = v; wi = ix; }
        }
        if (lane == 0) { s_val[wave] = wv; s_idx[wave] = wi; }
      }
      __syncthreads();
      double gv = s_val[lane & (NW - 1)];
      int pcol = s_idx[lane & (NW - 1)];
      hssk_row_argmax(gv, pcol);
      // ... published by two lanes of wave 0, the others' taken by wave 1: lane q < H the value of workgroup q, lane H + q its
      // column.  (Store and poll must not share a wave: the compiler is free to run the polling lanes of a divergent wave to
      // completion before the storing ones, and two workgroups doing that wait for each other forever.)
      if (tid < 2) hssk_cstore(xs, (size_t)(tid * H + h), tid ? (double)pcol : gv);
      if (wave == 1 && lane < 2 * H) {
        const int q = lane % H;
        const bool isidx = lane >= H;
        const double got = q == h ? (isidx ? (double)pcol : gv) : idg_take(xs, (size_t)lane, err);
        if (isidx) s_oidx[q] = got; else s_oval[q] = got;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < H; q++) {
        const double v = s_oval[q];
        const int ix = (int)s_oidx[q];
        if (q != h && (v > gv || (v == gv && ix < pcol))) { gv = v; pcol = ix; }
      }
      const int owner = pcol / HC, lp = pcol - owner * HC;
      const int pg = lp % NC, cp = lp / NC, wp = pg >> 2, sp = pg & 3;
      // ---- 2. reflector from the pivot column (dlarfg): its owner computes and publishes it
      if (h == owner) {
        if (wave == wp) {
          // (the pivot column's slot cp is a run-time value: its entries are picked out of the lane's slots with compare-selects,
          //  the new column is put back the same way -- one copy of the row loop instead of one per slot)
          const bool own = sub == sp;
          // (the pivot column is picked out of the slots twice -- for its norm, then for the reflector -- rather than kept in RT
          //  more registers between the two passes)
          const auto pick = [&](int r) {
            double v = 0.;
#pragma unroll
            for (int c = 0; c < CT; c++) v = fma(a[c][r], c == cp ? 1. : 0., v);
            return v;
          };
          double s = 0., arow = 0.;
#pragma unroll
          for (int r = 0; r < RT; r++) {
            const bool below = r > rk || (r == rk && l16 > lk);
            const double v = pick(r);
            s += below ? v * v : 0.;
            arow = fma(v, r == rk ? 1. : 0., arow);
          }
          const double alpha = hssk_shfl(arow, (lane & 48) | lk);
          s = hssk_row_sum(s);
          double tau = 0., beta = alpha, scal = 1.;
          if (s != 0.) {
            double nrm = sqrt(alpha * alpha + s);
            beta = alpha >= 0. ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            scal = 1. / (alpha - beta);
          }
#pragma unroll
          for (int r = 0; r < RT; r++) {
            const bool below = r > rk || (r == rk && l16 > lk);
            const bool diag = r == rk && l16 == lk;
            const double v = pick(r);
            const double scaled = v * scal;
            const double vrow = below ? scaled : (diag ? 1. : 0.);
            const double anew = below ? scaled : (diag ? beta : v);
            if (own) {
              s_v[l16 + 16 * r] = vrow;
              hssk_cstore(xs, (size_t)(2 * H + l16 + 16 * r), vrow);
            }
#pragma unroll
            for (int c = 0; c < CT; c++) a[c][r] = (own && c == cp) ? anew : a[c][r];
          }
          if (own) {
            used |= 1u << cp;
            if (l16 == 0) {
              const double ab = fabs(beta);
              const double r00 = (k == 0) ? ab : s_r00;
              // dgeqp3tol.f:225-232 (0/0 is NaN -> false, then the absolute test decides)
              const double stop = ((r00 != 0. && ab / r00 <= p.rtol) || ab <= p.atol) ? 1. : 0.;
              s_msg[0] = tau; s_msg[1] = stop; s_msg[2] = ab;
              hssk_cstore(xs, (size_t)(2 * H + 16 * RT), tau);
              hssk_cstore(xs, (size_t)(2 * H + 16 * RT + 1), stop);
              hssk_cstore(xs, (size_t)(2 * H + 16 * RT + 2), ab);
            }
          }
        }
      } else {
        // the other workgroups take it word by word
        if (tid < 16 * RT) s_v[tid] = idg_take(xs, (size_t)(2 * H + tid), err);
        else if (tid < 16 * RT + 3) s_msg[tid - 16 * RT] = idg_take(xs, (size_t)(2 * H + tid), err);
      }
      __syncthreads();
      if (tid == 0) {
        if (k == 0) s_r00 = s_msg[2];
        s_perm[k] = pcol;
      }
      if (s_msg[1] != 0.) { rank = k; __syncthreads(); break; }
      const double tau = s_msg[0];
      // ---- 3. apply H to this workgroup's unused columns and down-date their norms (dlaqp2)
      // all dot products of the wave's slots first, their row sums stage by stage (independent chains in flight).  The reflector
      // is read from the LDS in both passes (a compiler fence between them): 2 RT registers less across the row sums, which at
      // RT = 16 are the difference between a register tile and spills
      double dot[CT];
      {
        double d0[CT], d1[CT];
#pragma unroll
        for (int c = 0; c < CT; c++) d0[c] = d1[c] = 0.;
#pragma unroll
        for (int r = 0; r + 1 < RT; r += 2) {
          const double v0 = s_v[l16 + 16 * r], v1 = s_v[l16 + 16 * (r + 1)];
#pragma unroll
          for (int c = 0; c < CT; c++) { d0[c] += v0 * a[c][r]; d1[c] += v1 * a[c][r + 1]; }
        }
        if (RT & 1) {
          const double v0 = s_v[l16 + 16 * (RT - 1)];
#pragma unroll
          for (int c = 0; c < CT; c++) d0[c] += v0 * a[c][RT - 1];
        }
#pragma unroll
        for (int c = 0; c < CT; c++) dot[c] = d0[c] + d1[c];
      }
      HSSK_COMPILER_FENCE();
      hssk_row_sum_n(dot);
      double newk[CT];
      {
        double f[CT];
#pragma unroll
        for (int c = 0; c < CT; c++) {
          const int col = c0 + grp + NC * c;
          f[c] = (col < m && !((used >> c) & 1u)) ? dot[c] * tau : 0.;
        }
#pragma unroll
        for (int r = 0; r < RT; r++) {
          const double v = s_v[l16 + 16 * r];
#pragma unroll
          for (int c = 0; c < CT; c++) a[c][r] -= f[c] * v;
        }
#pragma unroll
        for (int c = 0; c < CT; c++) newk[c] = hssk_shfl(idg_pick<RT>(a[c], rk), (lane & 48) | lk);  // R(k, col)
      }
#pragma unroll
      for (int c = 0; c < CT; c++) {
        const int lc = grp + NC * c, col = c0 + lc;
        const bool act = col < m && !((used >> c) & 1u);
        double n1 = 0., n2 = 0., newn1 = 0.;
        int recompute = 0;
        if (act) {
          n1 = s_vn1[lc]; n2 = s_vn2[lc];
          newn1 = n1 - newk[c] * newk[c];
          newn1 = newn1 > 0. ? newn1 : 0.;
          recompute = (n1 != 0.) && (newn1 <= tol3z * n2);
        }
        if (hssk_any(recompute)) {
          double s2 = 0.;
#pragma unroll
          for (int r = 0; r < RT; r++)
            if (r > rk || (r == rk && l16 > lk)) s2 += a[c][r] * a[c][r];
          s2 = hssk_row_sum(s2);
          if (recompute) {
            newn1 = s2;
            if (l16 == 0) s_vn2[lc] = newn1;
          }
        }
        if (act && l16 == 0) s_vn1[lc] = newn1;
      }
      __syncthreads();   // (the step's LDS words are rewritten by the next one)
    }
  }
  // ---- pivoted column positions: skeleton columns first (pivot order), then the rest in index order (every workgroup works
  // the whole table out: the pivots of all steps are known to all)
  __syncthreads();
  for (int j = tid; j < H * HC; j += NT) s_pos[j] = -1;
  __syncthreads();
  for (int j = tid; j < rank; j += NT) s_pos[s_perm[j]] = j;
  __syncthreads();
  {
    // a column that was never a pivot goes behind the skeleton, in index order: rank + (number of such columns before it);
    // m <= H HC: up to H columns per thread, the new positions committed after everybody has counted
    int mine[H];
#pragma unroll
    for (int q = 0; q < H; q++) {
      const int col = tid + q * NT;
      mine[q] = -1;
      if (col < m && s_pos[col] < 0) {
        int c = 0;
        for (int e = 0; e < col; e++) c += s_pos[e] < 0;
        mine[q] = rank + c;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < H; q++)
      if (mine[q] >= 0) s_pos[tid + q * NT] = mine[q];
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < CT; c++) {
    const int col = c0 + grp + NC * c;
    if (col < m) {
      const int pos = s_pos[col];
#pragma unroll
      for (int r = 0; r < RT; r++) {
        const int row = l16 + 16 * r;
        if (16 * r < rank && row < d) hssk_gstore(p.W, row + (size_t)pos * ld, a[c][r]);
      }
      if (l16 == 0) p.perm[pos] = col;
    }
  }
  if (h == 0 && tid == 0) *p.rank = rank;
}

template <int RT, int CT, int NW>
void launch_id_reg(hssk_ctx* ctx, const hssk_id_desc* dd, int count) {
  HSSK_LAUNCH((id_reg_kernel<RT, CT, NW>), dim3((unsigned)count), dim3(NW * 64), 0, ctx->stream, dd);
  if (!g_all_deferred) HSSK_LAUNCH(id_xsolve_kernel, dim3((unsigned)count), dim3(XS_T), 0, ctx->stream, dd);
}
// panels of d <= 16 RT sample rows and up to 64 / 96 / 128 / 160 / 192 / 224 columns (a step costs per column slot); 8-wave workgroups (two waves per SIMD: 256 VGPRs
// for the register tile and the unrolled step loop; 16 waves with half the slots issue the same number of instructions
// per SIMD and step)
template <int RT>
bool launch_id_reg_ct(hssk_ctx* ctx, const hssk_id_desc* dd, int count, int mmax) {
  if (mmax <= 64) launch_id_reg<RT, 2, 8>(ctx, dd, count);
  else if (mmax <= 96) launch_id_reg<RT, 3, 8>(ctx, dd, count);
  else if (mmax <= 128) launch_id_reg<RT, 4, 8>(ctx, dd, count);
  else if (mmax <= 160) launch_id_reg<RT, 5, 8>(ctx, dd, count);   // (the 156-row tiles of the 200^3 problem's BLR fronts)
  else if (mmax <= 192) launch_id_reg<RT, 6, 8>(ctx, dd, count);
  else if (mmax <= 224) launch_id_reg<RT, 7, 8>(ctx, dd, count);
  else return false;
  return true;
}



// ------------------------------------------------------------------------------------------------
// Wide variant for a few LARGE panels (kernel matrices with ranks in the hundreds: m up to a few thousand): the
// same Level-2 truncated QRCP, but every Householder step of the whole batch is two launches -- idw_pivot_kernel
// (one workgroup per matrix: pivot search, column swap, reflector, stopping test) and idw_update_kernel (MANY
// workgroups per matrix: each wave applies the reflector to one trailing column and down-dates its norm) -- so the
// trailing update, which is all the work, runs on the whole chip instead of one CU per matrix.  The step index is a
// kernel argument; per-matrix state (stop flag, rank, tau, |R_00|) lives in device memory; X = R11^{-1} R12 is one
// batched triangular solve at the end.
// ------------------------------------------------------------------------------------------------
struct IdwState {
  int stop, rank;
  double tau, r00;
};
constexpr int IDW_COLS = 16;   // trailing columns per workgroup of the update kernel (4 waves x 4)

__global__ __launch_bounds__(256) void idw_init_kernel(const hssk_id_desc* __restrict__ descs, IdwState* __restrict__ st) {
  const hssk_id_desc p = descs[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (blockIdx.x == 0 && tid == 0) {
    const int kmax = p.d < p.m ? p.d : p.m;
    st[blockIdx.y] = IdwState{kmax == 0 ? 1 : 0, kmax, 0., 0.};
  }
  for (int j = blockIdx.x * 4 + wave; j < p.m; j += gridDim.x * 4) {
    double s = 0.;
    for (int i = lane; i < p.d; i += 64) { const double v = hssk_gload(p.W, i + (size_t)j * p.ldw); s += v * v; }
    s = hssk_wave_sum(s);
    if (lane == 0) { p.work[j] = sqrt(s); p.work[p.m + j] = sqrt(s); p.perm[j] = j; }
  }
}

__global__ __launch_bounds__(256) void idw_pivot_kernel(const hssk_id_desc* __restrict__ descs, IdwState* __restrict__ st, int k) {
  HSSK_SHARED double s_val[4];
  HSSK_SHARED int s_idx[4];
  HSSK_SHARED int s_piv;
  const hssk_id_desc p = descs[blockIdx.x];
  IdwState* me = st + blockIdx.x;
  const int kmax = p.d < p.m ? p.d : p.m;
  if (me->stop || k >= kmax) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = p.d, m = p.m, ld = p.ldw;
  double* vn1 = p.work;
  double* vn2 = p.work + m;
  double bv = -1.;
  int bi = 0x7fffffff;
  for (int j = k + tid; j < m; j += 256) {
    const double v = vn1[j];
    if (v > bv) { bv = v; bi = j; }
  }
  hssk_wave_argmax(bv, bi);
  if (lane == 0) { s_val[wave] = bv; s_idx[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
    double v = s_val[0];
    int ix = s_idx[0];
    for (int w = 1; w < 4; w++)
      if (s_val[w] > v || (s_val[w] == v && s_idx[w] < ix)) { v = s_val[w]; ix = s_idx[w]; }
    s_piv = ix;
    if (ix != k) {
      const int t = p.perm[k]; p.perm[k] = p.perm[ix]; p.perm[ix] = t;
      vn1[ix] = vn1[k]; vn2[ix] = vn2[k];
    }
  }
  __syncthreads();
  const int pv = s_piv;
  double* ck = p.W + (size_t)k * ld;
  if (pv != k) {
    double* cp = p.W + (size_t)pv * ld;
    for (int i = tid; i < d; i += 256) { const double a = ck[i], b = cp[i]; ck[i] = b; cp[i] = a; }
  }
  __syncthreads();
  // reflector of W[k:d, k]
  HSSK_SHARED double s_part[4];
  double s = 0.;
  for (int i = k + 1 + tid; i < d; i += 256) { const double v = ck[i]; s += v * v; }
  const double alpha = ck[k];   // read by everybody before thread 0 overwrites it
  s = hssk_wave_sum(s);
  if (lane == 0) s_part[wave] = s;
  __syncthreads();
  s = s_part[0] + s_part[1] + s_part[2] + s_part[3];
  double tau = 0., beta = alpha, scal = 1.;
  if (s != 0.) {
    const double nrm = sqrt(alpha * alpha + s);
    beta = alpha >= 0. ? -nrm : nrm;
    tau = (beta - alpha) / beta;
    scal = 1. / (alpha - beta);
    for (int i = k + 1 + tid; i < d; i += 256) ck[i] *= scal;
  }
  if (tid == 0) {
    ck[k] = beta;
    const double ab = fabs(beta);
    if (k == 0) me->r00 = ab;
    const double r00 = k == 0 ? ab : me->r00;
    me->tau = tau;
    if ((r00 != 0. && ab / r00 <= p.rtol) || ab <= p.atol) { me->stop = 1; me->rank = k; }
  }
}

__global__ __launch_bounds__(256) void idw_update_kernel(const hssk_id_desc* __restrict__ descs, const IdwState* __restrict__ st, int k) {
  const hssk_id_desc p = descs[blockIdx.y];
  const IdwState me = st[blockIdx.y];
  const int kmax = p.d < p.m ? p.d : p.m;
  if (me.stop || k >= kmax) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = p.d, m = p.m, ld = p.ldw;
  const double tol3z = 1.4901161193847656e-08;  // sqrt(eps)
  const double tau = me.tau;
  double* vn1 = p.work;
  double* vn2 = p.work + m;
  const double* v = p.W + (size_t)k * ld;
  const int j0 = k + 1 + blockIdx.x * IDW_COLS;
  for (int j = j0 + wave; j < j0 + IDW_COLS && j < m; j += 4) {
    double* col = p.W + (size_t)j * ld;
    double s = 0.;
    for (int i = k + 1 + lane; i < d; i += 64) s += v[i] * col[i];
    const double ckj = col[k];
    const double n1 = vn1[j], n2 = vn2[j];
    s = hssk_wave_sum(s);
    const double dot = tau * (ckj + s);
    if (tau != 0.) for (int i = k + 1 + lane; i < d; i += 64) col[i] -= dot * v[i];
    const double newk = ckj - dot;
    if (lane == 0) col[k] = newk;
    int recompute = 0;
    double newn1 = n1;
    if (n1 != 0.) {
      double t = fabs(newk) / n1;
      t = (1. + t) * (1. - t);
      t = t > 0. ? t : 0.;
      const double q = n1 / n2;
      if (t * q * q <= tol3z) recompute = 1;
      else newn1 = n1 * sqrt(t);
    }
    if (recompute) {
      double s2 = 0.;
      for (int i = k + 1 + lane; i < d; i += 64) { const double x = col[i]; s2 += x * x; }
      s2 = hssk_wave_sum(s2);
      newn1 = sqrt(s2);
      if (lane == 0) vn2[j] = newn1;
    }
    if (lane == 0) vn1[j] = newn1;
  }
}

void id_wide(hssk_ctx* ctx, const hssk_id_desc* descs, const hssk_id_desc* dd, int count) {
  int kmax = 0, mmax = 0;
  for (int i = 0; i < count; i++) {
    kmax = std::max(kmax, std::min(std::min(descs[i].d, descs[i].m), std::max(descs[i].max_rank, 0) + 1));
    mmax = std::max(mmax, descs[i].m);
  }
  IdwState* st = (IdwState*)ctx->scratch(sizeof(IdwState) * (size_t)count + 64);
  HSSK_LAUNCH(idw_init_kernel, dim3((unsigned)std::max(1, std::min(64, (mmax + 3) / 4)), (unsigned)count), dim3(256), 0, ctx->stream, dd, st);
  std::vector<IdwState> hs(count);
  const unsigned chunks = (unsigned)((mmax + IDW_COLS - 1) / IDW_COLS);
  for (int k = 0; k < kmax; k++) {
    HSSK_LAUNCH(idw_pivot_kernel, dim3((unsigned)count), dim3(256), 0, ctx->stream, dd, st, k);
    const unsigned ch = (unsigned)std::max(1, (mmax - (k + 1) + IDW_COLS - 1) / IDW_COLS);
    HSSK_LAUNCH(idw_update_kernel, dim3(std::min(ch, chunks), (unsigned)count), dim3(256), 0, ctx->stream, dd, st, k);
    if ((k & 63) == 63) {   // every 64 steps: has every matrix met its stopping test?
      hssk_rt::d2h(hs.data(), st, sizeof(IdwState) * count, ctx->stream);
      hssk_rt::sync(ctx->stream);
      bool all = true;
      for (auto& h : hs) all = all && h.stop;
      if (all) break;
    }
  }
  hssk_rt::d2h(hs.data(), st, sizeof(IdwState) * count, ctx->stream);
  hssk_rt::sync(ctx->stream);
  std::vector<hssk_trsm_desc> tr;
  std::vector<int> ranks(count);
  for (int i = 0; i < count; i++) {
    int r = std::min(hs[i].rank, descs[i].max_rank);
    r = std::max(r, 0);
    ranks[i] = r;
    if (r > 0 && descs[i].m > r)
      tr.push_back(hssk_trsm_desc{descs[i].W, descs[i].W + (size_t)r * descs[i].ldw, r, descs[i].m - r, descs[i].ldw, descs[i].ldw, 0, 0, 0});
  }
  // ranks to the device (one int each)
  for (int i = 0; i < count; i++) hssk_rt::h2d(descs[i].rank, &ranks[i], sizeof(int), ctx->stream);
  hssk_rt::sync(ctx->stream);
  if (!tr.empty() && hssk_trsm_vbatched(ctx, tr.data(), (int)tr.size())) throw std::runtime_error(hssk_last_error());
}

}  // namespace

static std::atomic<long long> g_group_launches{0};
extern "C" long long hssk_id_group_launches(void) { return g_group_launches; }

extern "C" int hssk_id_vbatched(hssk_ctx* ctx, const hssk_id_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  int dmax = 0, mmax = 0;
  for (int i = 0; i < count; i++) { dmax = std::max(dmax, descs[i].d); mmax = std::max(mmax, descs[i].m); }
  auto* dd = (const hssk_id_desc*)ctx->stage(descs, sizeof(*descs) * count);
  bool done = false;
  g_all_deferred = true;
  for (int i = 0; i < count; i++) g_all_deferred = g_all_deferred && descs[i].defer_x != 0;
  if (dmax <= 64) done = launch_id_reg_ct<4>(ctx, dd, count, mmax);
  else if (dmax <= 128) done = launch_id_reg_ct<8>(ctx, dd, count, mmax);
  else if (dmax <= 192) done = launch_id_reg_ct<12>(ctx, dd, count, mmax);
  else if (dmax <= 256) done = launch_id_reg_ct<16>(ctx, dd, count, mmax);
  if (!done) {
    // panels of up to 256 rows and 512 columns: two or four workgroups per panel, its columns in their registers (id_group_kernel)
    static const bool no_group = [] { const char* e = std::getenv("HSSK_ID_NO_GROUP"); return e && e[0] == '1'; }();
    bool group_ok = !no_group && dmax > 128 && dmax <= 256 && mmax <= 512 && count >= 1;
    // (the workgroups of a panel poll each other: as many as share a panel must be able to run together -- any GPU; the
    // test emulator with its pool of host threads answers for itself)
    group_ok = group_ok && hssk_rt::coresident_workgroups() >= 4;
    if (group_ok) {
      const int RTv = dmax <= 192 ? 12 : 16;
      // 256 x 256 tiles: three workgroups of 96 columns hold the tile without spills (5.4 us per step against 10 with two of
      // 128) -- as long as all of them are resident at once (one 8-wave workgroup of 240 registers per CU)
      static const bool no_h3 = [] { const char* e = std::getenv("HSSK_ID_GROUP_NO_H3"); return e && e[0] == '1'; }();
      static const int cus = hssk_rt::cu_count();
      const int H = mmax <= 256 ? ((RTv == 16 && count * 3 <= cus && !no_h3) ? 3 : 2) : 4;
      int kcap = 1;
      for (int i = 0; i < count; i++) kcap = std::max(kcap, std::min(std::min(descs[i].d, descs[i].m), std::max(descs[i].max_rank, 0)));
      const int sw = 2 * H + 16 * RTv + 3;
      const size_t words = (size_t)count * kcap * sw;
      double* xch = ctx->aux(sizeof(double) * words);
      { int rc = hssk_sweep_arm(ctx, xch, (long long)words); if (rc) return rc; }
      if (!ctx->h_sweep_err) { ctx->h_sweep_err = (int*)hssk_rt::pinned_malloc(64); *ctx->h_sweep_err = 0; }
      int* err = ctx->h_sweep_err;
      // (eight waves per workgroup, 128 columns each: the tile takes 128 of a lane's 256 registers; sixteen waves with half the
      //  columns each spill -- 232 bytes per lane at 256 rows)
      const dim3 grid((unsigned)count * H), block(512);
      g_group_launches++;
      if (RTv == 16 && H == 3) HSSK_LAUNCH((id_group_kernel<16, 3, 8, 3>), grid, block, 0, ctx->stream, dd, xch, kcap, err);
      else if (RTv == 16 && H == 2) HSSK_LAUNCH((id_group_kernel<16, 4, 8, 2>), grid, block, 0, ctx->stream, dd, xch, kcap, err);
      else if (RTv == 16) HSSK_LAUNCH((id_group_kernel<16, 4, 8, 4>), grid, block, 0, ctx->stream, dd, xch, kcap, err);
      else if (H == 2) HSSK_LAUNCH((id_group_kernel<12, 4, 8, 2>), grid, block, 0, ctx->stream, dd, xch, kcap, err);
      else HSSK_LAUNCH((id_group_kernel<12, 4, 8, 4>), grid, block, 0, ctx->stream, dd, xch, kcap, err);
      HSSK_LAUNCH(id_xsolve_all_kernel, dim3((unsigned)count), dim3(XS_T), 0, ctx->stream, dd);
      hssk_rt::check_launch();
      return 0;
    }
    // the in-place kernels want the panel in W
    std::vector<hssk_colgather_desc> cp;
    for (int i = 0; i < count; i++)
      if (descs[i].src) cp.push_back(hssk_colgather_desc{descs[i].src, descs[i].W, nullptr, descs[i].d, descs[i].m, descs[i].lds, descs[i].ldw, 0});
    if (!cp.empty()) { int rc = hssk_gather_cols(ctx, cp.data(), (int)cp.size()); if (rc) return rc; }
    // few large panels: spread every Householder step over the chip; many small ones: one workgroup each
    static const bool force_wide = [] { const char* e = std::getenv("HSSK_ID_WIDE"); return e && e[0] == '1'; }();
    static const bool no_stream = [] { const char* e = std::getenv("HSSK_ID_NO_STREAM"); return e && e[0] == '1'; }();   // (A/B: the first global-memory kernel)
    if (force_wide || (count <= 128 && (long long)dmax * mmax >= 256LL * 512)) id_wide(ctx, descs, dd, count);
    else if (dmax <= 512 && mmax <= IDS_MMAX && !no_stream) {
      if (dmax <= 256) HSSK_LAUNCH(id_stream_kernel<4>, dim3((unsigned)count), dim3(IDS_T), 0, ctx->stream, dd);
      else HSSK_LAUNCH(id_stream_kernel<8>, dim3((unsigned)count), dim3(IDS_T), 0, ctx->stream, dd);
      HSSK_LAUNCH(id_xsolve_all_kernel, dim3((unsigned)count), dim3(XS_T), 0, ctx->stream, dd);
    } else HSSK_LAUNCH(id_kernel, dim3((unsigned)count), dim3(ID_THREADS), 0, ctx->stream, dd);
  }
  hssk_rt::check_launch();
  HSSK_API_END
}

extern "C" int hssk_id_solves_inline(int dmax, int mmax) { return (dmax <= 256 && mmax <= 224) ? 0 : 1; }

extern "C" int hssk_id_xsolve_vbatched(hssk_ctx* ctx, const hssk_xsolve_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  auto* dd = (const hssk_xsolve_desc*)ctx->stage(descs, sizeof(*descs) * count);
  HSSK_LAUNCH(id_xsolve_to_kernel, dim3((unsigned)count), dim3(XS_T), 0, ctx->stream, dd);
  hssk_rt::check_launch();
  HSSK_API_END
}
