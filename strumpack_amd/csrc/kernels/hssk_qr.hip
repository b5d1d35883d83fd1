// Batched Householder QR with optional explicit Q, one workgroup per HSS node.
//
// Serves two reference routines: DenseMatrix::orthogonalize (geqrf + orgqr, returns max/min |R_ii|;
// dense/DenseMatrix.cpp:721-744 -- the rank-adequacy test of the stable compression,
// HSS/HSSMatrix.compress_stable.hpp:390-442) and DenseMatrix::LQ (gelqf + orglq with the full
// m x m Q; dense/DenseMatrix.cpp:693-719 -- the ULV factorization, HSS/HSSMatrix.factor.hpp:122):
// the LQ of W0 is computed as the QR of W0^T, which the engine forms directly.
//
// qr_kernel (global memory, Level-2) now serves as the PANEL factorisation of the tall blocked path (32 columns at a
// time, see qr_blocked); whole matrices go through the register kernels or the blocked paths.
// Same wave64 mapping as the ID kernel: the reflector is built by one wave with shuffle reductions,
// the trailing update gives one column to each wave, lanes stride the (contiguous) column.
// Q is accumulated backwards (dorg2r order) in a separate rows x nq buffer.
// Bound: L2 latency/bandwidth (Level-2 BLAS on an L2-resident panel).
#include "hssk_backsub.h"
#include "hssk_device.h"
#include "hssk_internal.h"
#include "hssk_qr_reg.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {

constexpr int QR_THREADS = 512;
constexpr int QR_WAVES = QR_THREADS / 64;

__global__ __launch_bounds__(QR_THREADS) void qr_kernel(const hssk_qr_desc* __restrict__ descs, int qonly) {
  HSSK_SHARED double s_tau;
  const hssk_qr_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rows = p.rows, cols = p.cols, ld = p.lda;
  double* __restrict__ A = p.A;
  double* taus = p.work;  // kmax entries
  const int kmax = rows < cols ? rows : cols;
  double rmax = 0., rmin = 0.;

  for (int k = 0; k < (qonly ? 0 : kmax); k++) {
    if (wave == 0) {
      double* col = A + (size_t)k * ld;
      double s = 0.;
      for (int i = k + 1 + lane; i < rows; i += 64) { double v = col[i]; s += v * v; }
      const double alpha = col[k];
      s = hssk_wave_sum(s);
      double tau = 0., beta = alpha;
      if (s != 0.) {
        double nrm = sqrt(alpha * alpha + s);
        beta = alpha >= 0. ? -nrm : nrm;
        tau = (beta - alpha) / beta;
        double scal = 1. / (alpha - beta);
        for (int i = k + 1 + lane; i < rows; i += 64) col[i] *= scal;
      }
      double ab = fabs(beta);
      if (k == 0) { rmax = ab; rmin = ab; }
      else { rmax = ab > rmax ? ab : rmax; rmin = ab < rmin ? ab : rmin; }
      if (lane == 0) { col[k] = beta; taus[k] = tau; s_tau = tau; }
    }
    __syncthreads();
    const double tau = s_tau;
    const double* v = A + (size_t)k * ld;
    if (tau != 0.)
      for (int j = k + 1 + wave; j < cols; j += QR_WAVES) {
        double* col = A + (size_t)j * ld;
        double s = 0.;
        for (int i = k + 1 + lane; i < rows; i += 64) s += v[i] * col[i];
        const double ckj = col[k];
        s = hssk_wave_sum(s);
        double dot = tau * (ckj + s);
        for (int i = k + 1 + lane; i < rows; i += 64) col[i] -= dot * v[i];
        if (lane == 0) col[k] = ckj - dot;
      }
    __syncthreads();
  }
  if (!qonly && p.rdiag && tid == 0) { p.rdiag[0] = rmax; p.rdiag[1] = rmin; }

  // ---- explicit Q(:, 0:nq) = H_0 H_1 ... H_{kmax-1} I(:, 0:nq)   (dorg2r)
  const int nq = p.nq;
  if (nq > 0) {
    double* __restrict__ Q = p.Q;
    const int ldq = p.ldq;
    for (int j = wave; j < nq; j += QR_WAVES)
      for (int i = lane; i < rows; i += 64) Q[i + (size_t)j * ldq] = (i == j) ? 1. : 0.;
    __syncthreads();
    for (int k = kmax - 1; k >= 0; k--) {
      const double tau = taus[k];
      const double* v = A + (size_t)k * ld;
      // columns j < k of Q(k:rows, :) are still zero at this point: start at j = k
      if (tau != 0.)
        for (int j = k + wave; j < nq; j += QR_WAVES) {
          double* col = Q + (size_t)j * ldq;
          double s = 0.;
          for (int i = k + 1 + lane; i < rows; i += 64) s += v[i] * col[i];
          const double ckj = col[k];
          s = hssk_wave_sum(s);
          double dot = tau * (ckj + s);
          for (int i = k + 1 + lane; i < rows; i += 64) col[i] -= dot * v[i];
          if (lane == 0) col[k] = ckj - dot;
        }
      __syncthreads();
    }
  }
}

// (the register-resident QR / Q formation bodies: hssk_qr_reg.h)
template <int RT, int CT, int NW>
__global__ __launch_bounds__(NW * 64) HSSK_WAVES_PER_SIMD(NW / 4) void qr_reg_kernel(const hssk_qr_desc* __restrict__ descs) {
  const hssk_qr_desc p = descs[blockIdx.x];
  qr_reg_body<RT, CT, NW>(p);
}
template <int RT, int CT, int NW>
__global__ __launch_bounds__(NW * 64) HSSK_WAVES_PER_SIMD(NW / 4) void formq_reg_kernel(const hssk_qr_desc* __restrict__ descs,
                                                                                       const QBlock* __restrict__ work) {
  const QBlock w = work[blockIdx.x];
  const hssk_qr_desc p = descs[w.prob];
  formq_reg_body<RT, CT, NW>(p, w.block);
}

// ------------------------------------------------------------------------------------------------
// QR of a PAIR of upper triangular factors stacked on top of each other, [R1; R2] = Q R (both m x m), the merge step of the
// TSQR tree (DeviceHSS::tsqr_reduce).  The structure does most of the work: the reflector of step k couples row k of R1
// with column k of R2 (rows 0..k) and nothing else, so
//   * R2 lives in the registers of the workgroup in the layout of qr_reg_kernel (one column per 16-lane row; its zeros
//     below the diagonal make every row mask unnecessary),
//   * R1 is never loaded as a matrix: step k reads its row k (one element per column, fetched one step ahead) and writes
//     it back as row k of R -- R1 is updated in place, R2 is only read,
//   * a step is one reflector generation on the owner group, one LDS broadcast, one barrier and the update of the columns
//     behind k: (2/3) m^3 flops per pair where the dense sweep of the 2m x m stack spends (10/3) m^3.
// One launch per tree level instead of a triangle copy and four launches per 32-column panel step of the blocked QR
// (the kernel-matrix workload at N = 1e5: ~33 ms of its 140 in those).  Capacity: m <= 16 RT rows, m <= 4 NW CT columns.
// ------------------------------------------------------------------------------------------------
template <int RT, int CT, int NW>
__global__ __launch_bounds__(NW * 64) HSSK_WAVES_PER_SIMD(NW / 4) void tpqr_reg_kernel(const hssk_tpqr_desc* __restrict__ descs) {
  constexpr int NC = NW * 4;
  // column slot c holds columns < NC (c + 1) of an upper triangular matrix: only the first RC(c) row registers can be non-zero
  // (they are the only ones allocated: every loop below is unrolled with these bounds)
#define RC(c) ((NC * ((c) + 1) + 15) / 16 < RT ? (NC * ((c) + 1) + 15) / 16 : RT)
  HSSK_SHARED double s_v[2 * 16 * RT];
  HSSK_SHARED double s_tau[2];
  HSSK_SHARED double s_t[2 * NC * CT];   // rows k, k + 1 of R1 (double-buffered: row k + 1 is staged during step k)
  const hssk_tpqr_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, sub = lane >> 4, grp = wave * 4 + sub;
  const int m = p.m;
  double* __restrict__ R1 = p.R1;
  double a[CT][RT];
#pragma unroll
  for (int c = 0; c < CT; c++)
#pragma unroll
    for (int r = 0; r < RC(c); r++) {
      const int row = l16 + 16 * r, col = grp + NC * c;
      a[c][r] = (row <= col && col < m) ? p.R2[row + (size_t)col * p.ld2] : 0.;
    }
  // R1(k, k) is fetched one step ahead into a register (the owner needs it first thing); row k as a whole is staged in LDS
  // during step k - 1 (thread j loads R1(k, j)) and read behind the barrier of step k.  No step writes a row but its own.
  double alpha_next = m > 0 ? hssk_gload(R1, 0) : 0.;
  if (tid < m) s_t[tid] = hssk_gload(R1, (size_t)tid * p.ld1);
#pragma clang loop unroll(full)
  for (int kc = 0; kc < CT; kc++) {
    const int ng = min(NC, m - NC * kc);   // steps whose column sits in slot kc
    for (int g = 0; g < ng; g++) {
      const int k = NC * kc + g, kw = g >> 2, ks = g & 3;
      double* sv = s_v + (k & 1) * 16 * RT;
      const double alpha = alpha_next;
      const double* st = s_t + (k & 1) * NC * CT;
      // row k + 1, for the next step.  UNCONDITIONAL loads from clamped addresses (entries at or left of the diagonal are never
      // read back): under conditions of their own the compiler cannot count them, and the owner wave's wait for R1(k, k) --
      // fetched a step ago -- became a wait for everything in flight, the row just asked for included: an L2 round trip per step
      const int kn = min(k + 1, m - 1);
      const double trow = hssk_gload(R1, (size_t)kn + (size_t)min(tid, m - 1) * p.ld1);
      alpha_next = hssk_gload(R1, (size_t)kn + (size_t)kn * p.ld1);
      if (wave == kw) {
        // reflector from [R1(k, k); R2(0:k, k)]
        double s = 0.;
#pragma unroll
        for (int r = 0; r < RC(kc); r++) s += a[kc][r] * a[kc][r];
        s = hssk_row_sum(s);
        double tau = 0., beta = alpha, scal = 1.;
        if (s != 0.) {
          const double nrm = sqrt(alpha * alpha + s);
          beta = alpha >= 0. ? -nrm : nrm;
          tau = (beta - alpha) / beta;
          scal = 1. / (alpha - beta);
        }
        if (sub == ks) {
#pragma unroll
          for (int r = 0; r < RC(kc); r++) {
            a[kc][r] *= scal;
            sv[l16 + 16 * r] = a[kc][r];
          }
          if (l16 == 0) {
            s_tau[k & 1] = tau;
            hssk_gstore(R1, (size_t)k + (size_t)k * p.ld1, beta);
          }
        }
      }
      __syncthreads();
      const double tau = s_tau[k & 1];
      if (tau != 0.) {
        double vr[RT];   // (the reflector of a column of slot kc is zero beyond its first RC(kc) row registers)
#pragma unroll
        for (int r = 0; r < RC(kc); r++) vr[r] = sv[l16 + 16 * r];
        double dot[CT];
#pragma unroll
        for (int c = 0; c < CT; c++) {
          double d0 = 0., d1 = 0.;
          if (c >= kc) {
#pragma unroll
            for (int r = 0; r + 1 < RC(kc); r += 2) { d0 += vr[r] * a[c][r]; d1 += vr[r + 1] * a[c][r + 1]; }
            if (RC(kc) & 1) d0 += vr[RC(kc) - 1] * a[c][RC(kc) - 1];
          }
          dot[c] = d0 + d1;
        }
        hssk_row_sum_n(dot);
#pragma unroll
        for (int c = kc; c < CT; c++) {
          const int col = grp + NC * c;
          if (col > k && col < m) {
            const double tc = st[col];
            const double f = (dot[c] + tc) * tau;   // w = R1(k, col) + v . R2(:, col)
#pragma unroll
            for (int r = 0; r < RC(kc); r++) a[c][r] -= f * vr[r];
            if (l16 == 0) hssk_gstore(R1, (size_t)k + (size_t)col * p.ld1, tc - f);
          }
        }
      }
      if (tid < m) s_t[((k + 1) & 1) * NC * CT + tid] = trow;   // (read behind the next step's barrier)
    }
  }
}
#undef RC

template <int RT, int CT, int NW>
void launch_qr_reg(hssk_ctx* ctx, const hssk_qr_desc* dd, int count) {
  HSSK_LAUNCH((qr_reg_kernel<RT, CT, NW>), dim3((unsigned)count), dim3(NW * 64), 0, ctx->stream, dd);
}
template <int RT, int CT, int NW>
void launch_formq_reg(hssk_ctx* ctx, const hssk_qr_desc* dd, const hssk_qr_desc* descs, int count) {
  std::vector<QBlock> blocks;
  for (int i = 0; i < count; i++)
    for (int b = 0; b * 4 * NW * CT < descs[i].nq; b++) blocks.push_back(QBlock{i, b});
  if (blocks.empty()) return;
  auto* dw = (const QBlock*)ctx->stage(blocks.data(), sizeof(QBlock) * blocks.size());
  HSSK_LAUNCH((formq_reg_kernel<RT, CT, NW>), dim3((unsigned)blocks.size()), dim3(NW * 64), 0, ctx->stream, dd, dw);
}
// Q of panels with up to 256 rows
void formq_reg(hssk_ctx* ctx, const hssk_qr_desc* dd, const hssk_qr_desc* descs, int count, int rmax) {
  // four column slots per group: four independent reflector-application chains in flight per wave
  if (rmax <= 64) launch_formq_reg<4, 2, 8>(ctx, dd, descs, count);
  else if (rmax <= 128) launch_formq_reg<8, 4, 8>(ctx, dd, descs, count);
  else if (rmax <= 208) launch_formq_reg<13, 4, 8>(ctx, dd, descs, count);
  else launch_formq_reg<16, 4, 8>(ctx, dd, descs, count);
}

// ------------------------------------------------------------------------------------------------
// Blocked path for panels beyond the register kernels (leaf size 512: W0^T is ~390 x 350, its Q 390 x 390).
// Level-synchronous over the whole batch, QB columns at a time:
//   1. qr_reg_kernel<32, 1, 8> factors the (rows - j0) x QB panel of every matrix in registers,
//   2. larft_kernel builds the compact-WY pair of the panel, V (unit lower trapezoidal, explicit) and
//      VT = V T  (T from the dlarft recurrence on V^T V, all in LDS), and folds the panel's max/min |R_ii|,
//   3. two batched MFMA GEMMs apply H^T = I - V (VT)^T to the trailing columns: W = VT^T A2, A2 -= V W.
// Q is then formed by the same two GEMMs per panel in reverse order (Q <- Q - VT (V^T Q)).
// This turns the Level-2, latency-bound sweep of qr_kernel into Level-3 work on the matrix cores.
// ------------------------------------------------------------------------------------------------
constexpr int QB = 32;
constexpr int QB_MAXROWS = 512;

struct QPanel {
  const double* A;   // factored panel (rows x nb, reflectors below the diagonal)
  const double* tau;
  double* Vc;        // out: explicit V
  double* VT;        // out: V T
  double* rd_panel;  // max/min |R_ii| of this panel (written by the panel kernel)
  double* rdiag;     // running max/min of the whole matrix (may be null)
  int lda, rows, nb, ldv, first;
};

// MAXR: largest panel height of the batch the LDS image of V is sized for (256 rows: two workgroups per CU).
// Three stages, each measured on 224-row panels (stamps; before -> after): G = V^T V 32 -> ~10 us (four partial sums per
// product: a single chain waits out an LDS round trip per term); T 23 -> ~2 us -- T^{-1} is KNOWN, the strict upper
// triangle of V^T V with 1 / tau on the diagonal, so column j of T is one register back substitution per lane
// (hssk_backsub64; the dlarft recurrence ran 32 dependent steps on one wave); V T 14 -> ~5 us (four products per pass).
// NT threads: the LDS image of a 512-row panel leaves one workgroup per CU, whose four waves spent 50 us per panel waiting
// on LDS round trips (leaf 512: 12 panels per factorization); more waves share the same loops (from 1024 threads on two per product
// of G, each half of the rows: a sum of two terms into a zeroed slot does not depend on the order).
template <int MAXR, int NT>
__global__ __launch_bounds__(NT) void larft_kernel(const QPanel* __restrict__ descs) {
  constexpr int PARTS = NT >= 1024 ? 2 : 1;
  constexpr int LDV = MAXR + 1;   // odd: lanes that walk along columns hit distinct LDS banks
  constexpr int LR = HSSK_BACKSUB_LD;
  HSSK_SHARED double s_V[QB * LDV];
  HSSK_SHARED double s_G[QB * LR];   // strict upper triangle of V^T V (order nb <= 32: hssk_backsub64 reads no further)
  HSSK_SHARED double s_rd[64];
  HSSK_SHARED double s_T[QB * QB];
  const QPanel p = descs[blockIdx.x];
  const int tid = threadIdx.x;
  const int rows = p.rows, nb = p.nb;
  if (tid == 0 && p.rdiag) {
    const double pm = nb ? p.rd_panel[0] : 0., pn = nb ? p.rd_panel[1] : 0.;
    if (p.first) { p.rdiag[0] = pm; p.rdiag[1] = pn; }
    else if (nb) {
      if (pm > p.rdiag[0]) p.rdiag[0] = pm;
      if (pn < p.rdiag[1]) p.rdiag[1] = pn;
    }
  }
  if (nb == 0) return;
  for (int e = tid; e < rows * nb; e += NT) {
    const int i = e % rows, j = e / rows;
    const double v = i < j ? 0. : (i == j ? 1. : hssk_gload(p.A, i + (size_t)j * p.lda));
    s_V[j * LDV + i] = v;
    hssk_gstore(p.Vc, i + (size_t)j * p.ldv, v);
  }
  for (int e = tid; e < QB * LR; e += NT) s_G[e] = 0.;
  // (tau_i == 0: H_i = I, row and column i of T are zero -- a zero "reciprocal" does that in the back substitution)
  if (tid < 64) s_rd[tid] = tid < nb ? p.tau[tid] : 0.;
  __syncthreads();
  // G(a, b) = v_a . v_b for a < b  (v_b is zero above row b): pair e of the nb (nb - 1) / 2, four partial sums
  for (int e = tid; e < PARTS * (nb * (nb - 1) / 2); e += NT) {
    int b = 1, rem = e / PARTS;
    const int part = e % PARTS;
    while (rem >= b) { rem -= b; b++; }   // pair = b (b - 1) / 2 + a
    const int a2 = rem;
    const double* va = s_V + a2 * LDV;
    const double* vb = s_V + b * LDV;
    double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
    const int mid = PARTS == 1 ? rows : b + (((rows - b) / 2) & ~3), lo = part ? mid : b, hi = part || PARTS == 1 ? rows : mid;
    int i = lo;
    for (; i + 3 < hi; i += 4) {
      s0 += va[i] * vb[i];
      s1 += va[i + 1] * vb[i + 1];
      s2 += va[i + 2] * vb[i + 2];
      s3 += va[i + 3] * vb[i + 3];
    }
    for (; i < hi; i++) s0 += va[i] * vb[i];
    if (PARTS == 1) s_G[a2 + b * LR] = (s0 + s1) + (s2 + s3);
    else hssk_lds_add(&s_G[a2 + b * LR], (s0 + s1) + (s2 + s3));
  }
  __syncthreads();
  // T = (striu(G) + diag(1 / tau))^{-1}: lane j of the first wave solves for column j
  if (tid < 64) {
    double x[64];
#pragma unroll
    for (int i = 0; i < 64; i++) x[i] = (i == tid && tid < nb) ? 1. : 0.;
    hssk_backsub64(x, s_G, s_rd, nb);
    if (tid < QB) {
#pragma unroll
      for (int i = 0; i < QB; i++) s_T[i + tid * QB] = x[i];
    }
  }
  __syncthreads();
  // VT = V T: entry (i, j) = sum_{c <= j} V(i, c) T(c, j); four columns j per pass and thread
  for (int e = tid; e < rows * (QB / 4); e += NT) {
    const int i = e % rows, j0 = 4 * (e / rows);
    if (j0 >= nb) continue;
    double s[4] = {0., 0., 0., 0.};
    for (int c = 0; c < min(nb, j0 + 4); c++) {
      const double v = s_V[c * LDV + i];
#pragma unroll
      for (int u = 0; u < 4; u++) s[u] += v * s_T[c + (j0 + u) * QB];   // (T is upper triangular: zero for c > j)
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (j0 + u < nb) hssk_gstore(p.VT, i + (size_t)(j0 + u) * p.ldv, s[u]);
  }
}

// ---- tall panels (rows > QB_MAXROWS): V does not fit LDS, so the compact-WY pair is assembled from batched GEMMs:
// make_v_kernel writes the explicit V (and folds the panel's max / min |R_ii|), G = V^T V and VT = V T are GEMMs,
// larft_small_kernel runs the dlarft recurrence on G
__global__ __launch_bounds__(256) void make_v_kernel(const QPanel* __restrict__ descs) {
  const QPanel p = descs[blockIdx.x];
  const int rows = p.rows, nb = p.nb;
  if (blockIdx.y == 0 && threadIdx.x == 0 && p.rdiag) {
    const double pm = nb ? p.rd_panel[0] : 0., pn = nb ? p.rd_panel[1] : 0.;
    if (p.first) { p.rdiag[0] = pm; p.rdiag[1] = pn; }
    else if (nb) {
      if (pm > p.rdiag[0]) p.rdiag[0] = pm;
      if (pn < p.rdiag[1]) p.rdiag[1] = pn;
    }
  }
  for (size_t e = (size_t)blockIdx.y * blockDim.x + threadIdx.x; e < (size_t)rows * nb; e += (size_t)gridDim.y * blockDim.x) {
    const int i = (int)(e % rows), j = (int)(e / rows);
    hssk_gstore(p.Vc, i + (size_t)j * p.ldv, i < j ? 0. : (i == j ? 1. : hssk_gload(p.A, i + (size_t)j * p.lda)));
  }
}
struct TDesc {
  const double* G;   // nb x nb (ld QB): V^T V, upper part used
  const double* tau;
  double* T;         // nb x nb (ld QB)
  int nb;
};
__global__ __launch_bounds__(64) void larft_small_kernel(const TDesc* __restrict__ descs) {
  HSSK_SHARED double s_T[QB * QB];
  const TDesc p = descs[blockIdx.x];
  const int lane = threadIdx.x, nb = p.nb;
  for (int e = lane; e < QB * QB; e += 64) s_T[e] = 0.;
  __syncthreads();
  if (lane < nb) {
    for (int i = 0; i < nb; i++) {
      const double ti = p.tau[i];
      if (lane < i) {
        double s = 0.;
        for (int c = lane; c < i; c++) s += s_T[lane + c * QB] * p.G[c + i * QB];
        s_T[lane + i * QB] = -ti * s;
      } else if (lane == i) s_T[lane + i * QB] = ti;
    }
  }
  __syncthreads();
  for (int e = lane; e < QB * QB; e += 64) p.T[e] = s_T[e];
}

struct EyeDesc {
  double* Q;
  int ldq, rows, nq;
};
__global__ void eye_kernel(const EyeDesc* __restrict__ descs) {
  const EyeDesc p = descs[blockIdx.x];
  for (size_t e = (size_t)blockIdx.y * blockDim.x + threadIdx.x; e < (size_t)p.rows * p.nq; e += (size_t)gridDim.y * blockDim.x) {
    const int i = (int)(e % p.rows), j = (int)(e / p.rows);
    hssk_gstore(p.Q, i + (size_t)j * p.ldq, i == j ? 1. : 0.);
  }
}

// ---- C <- C - V (VT^T C): a compact-WY block reflector (V, VT = V T: rr x nb, nb <= 32) applied to rr x nc columns, FUSED.
// As two batched GEMMs (W = VT^T C, then C -= V W) the columns cross HBM three times and every product of a 472 x 512 leaf
// block is a launch of thin tiles (leaf size 512: 12 panels x 2 launches for the factorization and as many for Q, 80 Gflop
// in 4.3 ms; factor phase 7.63 ms).  Here a workgroup takes NC columns of C into the LDS (wy_coff below), forms its 32 x NC
// block of W from there -- 16 x 16 tiles, the K range of a tile cut over the waves left (NC = 16, four waves: two tiles x
// two halves), VT fragments straight from global memory (L2: the pair is shared by the workgroups of the panel), a batch
// ahead --, leaves the partial W's in the LDS and applies V with the accumulators loaded from the LDS copy of C: C is read
// once and written once.  NC = 16 keeps two workgroups on a CU for 512 rows (79 KB each).
// SEVERAL reflector blocks in one pass (WyDesc::np pairs, applied in order).  The panels of a GROUP (wy_group() of them)
// are applied right away only to the columns of their own group; the columns beyond receive the whole group in one pass,
// the block staying in the LDS between the pairs (pair q acts on the rows from roff_q on).  Forming Q walks the groups
// backwards the same way (a pair meets exact zeros in the columns its panel has not reached: W = 0 there).
// Measured, leaf 512 at N = 1e5 (196 leaves, 12 panels each; factor phase of bench.py --leaf 512, gpurun_out/r04j .. r04p):
//   two batched GEMMs per panel                                                   7.63 ms
//   fused, loads under branches (vmcnt(0) after every prefetch)                   8.17
//   unconditional loads                                                           6.02   (wy launches 3.3 ms per step)
//   + groups of four panels (a quarter of the passes over C), T factors on 512 threads, tiled transposed gathers   5.49
//   + LDS reads in batches, two accumulators, row tiles side by side              5.57   (no change: not bound there)
//   + VT in 16-byte loads, a 128-byte line per column and 16-row step             5.14   (wy launches 2.8 ms per step)
// A workgroup-pair of a CU now spends ~16 us per reflector block where its 1024 MFMAs need 6.8: 0.42 of the matrix pipe
// inside the kernel; the rest is phase changes (two barriers and the reduction of W per block) with two workgroups per CU.
constexpr int WY_LDW = 34, WY_MAXP = 4;
struct WyPair {
  const double* V;    // rows x nb
  const double* VT;   // rows x nb   (V T)
  int roff, nb, rows, pad_;   // the pair acts on rows [roff, roff + rows) of the block
};
struct WyDesc {
  WyPair pr[WY_MAXP];
  double* C;          // rr x nc
  int ldv, ldc, rr, nc, np;
};
struct WyWork { int prob, cblock; };
// LDS copy of the column block: column j at wy_coff(j) = j ldr + 8 (j / 4) + j % 4 with ldr a multiple of 32 -- the 32
// lanes of an operand read (16 columns x 2 four-row groups: row 16 S + 4 l4 + u) then fall on 32 distinct banks
__host__ __device__ inline int wy_ldr(int rr) { return (((rr + 15) & ~15) + 1 + 31) & ~31; }   // (room for one zero row past the padded block)
__host__ __device__ inline int wy_coff(int j, int ldr) { return j * ldr + 8 * (j >> 2) + (j & 3); }
template <int NC, int T> __global__ __launch_bounds__(T, T / 128) void wy_apply_kernel(const WyDesc* __restrict__ descs, const WyWork* __restrict__ work) {
  HSSK_DYN_SHARED(double, wy_lds);
  constexpr int NW = T / 64, TILES = 2 * (NC / 16), KP = NW / TILES;   // W tiles (QB / 16 = 2 row tiles) and the K parts of each
  constexpr int KS = KP > 2 ? 2 : KP;                                  // partial W's left in the LDS (four parts fold into two first)
  constexpr int AHS = T >= 512 ? 2 : 4, TC = 4, NU = QB / 4;   // (16-row steps of VT in flight per wave: the register budget of four waves per SIMD)
  const WyWork w = work[blockIdx.x];
  const WyDesc* pd = descs + w.prob;
  double* const Cg = pd->C;
  const int ldv = pd->ldv, ldc = pd->ldc, rr = pd->rr, np = pd->np;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int j0 = w.cblock * NC, ncb = min(NC, pd->nc - j0);
  const int rpad = (rr + 15) & ~15, ldr = wy_ldr(rr);
  double* Cs = wy_lds;                                 // column j from wy_coff(j, ldr) on; rows [rr, rpad] zero
  double* Ws = wy_lds + wy_coff(NC - 1, ldr) + ldr;    // [KS][NC][WY_LDW]: part q of W(i, j) at Ws[(q * NC + j) * WY_LDW + i]
  const int tile = wave % TILES, q = wave / TILES, a = tile & 1, b = tile >> 1;   // this wave's (tile, K part) of W; tile = (row tile a of nb, column tile b)
  for (int pi = 0; pi < np; pi++) {
    const WyPair pr = pd->pr[pi];
    const int nb = pr.nb, r0 = pr.roff, rrp = pr.rows, rpadp = (rrp + 15) & ~15, ntile = rpadp / 16;   // (roff: a multiple of 16)
    // Every global load of a pair is issued before its first product: a workgroup is a chain of latencies otherwise (C, the VT
    // fragments batch by batch, V tile by tile) with 3 us of MFMA work in between.  And no load of V / VT sits under a branch
    // of its own or feeds a select: the compiler turns `cond ? load : 0` into a load under a branch and then waits for ALL
    // loads in flight at the next use -- vmcnt(0) after every prefetch, 16 us per workgroup.  What a clamped load brings in
    // meets a zero on the other side instead: steps beyond the wave's rows read the zero row of the LDS copy, rows of W
    // beyond nb are stored as zeros.
    // (1) first VT fragments.  The contraction runs in 16-row steps: lane (column i = l15, group l4) takes rows 16 S + 4 l4
    // .. + 3 of its column as two 16-byte loads -- a 128-byte line per column and step, where a row per lane and product
    // (rows 4 s + l4) asked the L1 for 16 lines per product, each a quarter used: the kernel was bound THERE (a pass over C
    // less per panel and twice the MFMA rate changed nothing).
    const int vi = 16 * a + l15;                               // column of VT this lane feeds
    const double* vt = pr.VT + (size_t)min(vi, nb - 1) * ldv;
    const bool vok = vi < nb;                                  // (rows of W beyond nb: computed from column nb - 1, stored as zeros)
    const int nfull = rrp >> 4;                                // whole 16-row steps; the rest (< 16 rows) row by row behind them
    const int perS = nfull ? ((nfull + KP - 1) / KP + AHS - 1) / AHS * AHS : 0;
    const int S_lo = q * perS;
    auto load_vt4 = [&](int S, hssk_d2 (&o)[2]) {
      const size_t off = (size_t)(16 * min(S, nfull - 1) + 4 * l4);
      o[0] = hssk_gload2u(vt, off);
      o[1] = hssk_gload2u(vt, off + 2);
    };
    hssk_d2 vq[AHS][2];
    if (nfull) {
#pragma unroll
      for (int x = 0; x < AHS; x++) load_vt4(S_lo + x, vq[x]);
    }
    // (2) first pair: C block -> LDS (zeros beyond the block: they feed the MFMAs); the NC loads of a pass are in flight together
    if (pi == 0) {
      for (int i0 = 0; i0 < rpad; i0 += T) {
        const int i = i0 + tid;
        const double* src = Cg + min(i, rr - 1) + (size_t)j0 * ldc;
        double v[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) v[c] = hssk_gload(src, (size_t)min(c, ncb - 1) * ldc);
        if (i < rpad) {
#pragma unroll
          for (int c = 0; c < NC; c++) Cs[wy_coff(c, ldr) + i] = (i < rr && c < ncb) ? v[c] : 0.;
        }
      }
      if (tid < NC) Cs[wy_coff(tid, ldr) + rpad] = 0.;
    }
    // (3) the V values of this wave's first row tiles of the second product (they arrive under the first one)
    auto load_v = [&](int t, double (&av)[NU]) {
      const int gi = 16 * t + l15;
      const double* vrow = pr.V + min(gi, rrp - 1);
#pragma unroll
      for (int u = 0; u < NU; u++) {
        const int k = 4 * u + l4;
        av[u] = -hssk_gload(vrow, (size_t)min(k, nb - 1) * ldv);   // (columns beyond nb meet zero rows of W, rows beyond the pair's are not stored)
      }
    };
    double avs[TC][NU];
#pragma unroll
    for (int x = 0; x < TC / 2; x++) load_v(wave + x * NW, avs[x]);   // (the other half once the VT registers are free)
    __syncthreads();   // (the block is in the LDS: staged, or left there by the pair before)
    if (pi == np - 1 && pi > 0 && r0 > 0) {   // rows above the last pair's: final since the pairs before, written out from the LDS
      for (int c = 0; c < ncb; c++)
        for (int i = tid; i < r0; i += T) hssk_gstore(Cg, (size_t)i + (size_t)(j0 + c) * ldc, Cs[wy_coff(c, ldr) + i]);
    }
    // ---- W = VT^T C
    {
      // (a batch of LDS reads, then its products on two accumulators in turn)
      hssk_d4 acc = {0., 0., 0., 0.}, acc1 = {0., 0., 0., 0.};
      const double* cs = Cs + wy_coff(16 * b + l15, ldr) + r0;
      const int zrow = rpad - r0;
      for (int S0 = S_lo; S0 < S_lo + perS; S0 += AHS) {
        double cv[AHS][4];
        hssk_d2 vb[AHS][2];
#pragma unroll
        for (int x = 0; x < AHS; x++) {
          const int S = S0 + x, k = 16 * S + 4 * l4;
#pragma unroll
          for (int u = 0; u < 4; u++) cv[x][u] = cs[S < nfull ? k + u : zrow];
          vb[x][0] = vq[x][0]; vb[x][1] = vq[x][1];
        }
#pragma unroll
        for (int x = 0; x < AHS; x++) load_vt4(S0 + AHS + x, vq[x]);
#pragma unroll
        for (int x = 0; x < AHS; x++) {
          acc = hssk_mfma_f64_16x16x4(cv[x][0], vb[x][0][0], acc);   // swapped: lane holds W[i = l15][j = l4 + 4 r]
          acc1 = hssk_mfma_f64_16x16x4(cv[x][1], vb[x][0][1], acc1);
          acc = hssk_mfma_f64_16x16x4(cv[x][2], vb[x][1][0], acc);
          acc1 = hssk_mfma_f64_16x16x4(cv[x][3], vb[x][1][1], acc1);
        }
      }
      if ((rrp & 15) && q == 0) {   // the rows behind the last whole step: a row per lane and product, clamped (the wait for these loads is the phase's last)
        double tv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) tv[u] = hssk_gload(vt, (size_t)min(16 * nfull + 4 * u + l4, rrp - 1));
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int k = 16 * nfull + 4 * u + l4;
          acc = hssk_mfma_f64_16x16x4(cs[k < rrp ? k : zrow], tv[u], acc);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; r++) acc[r] += acc1[r];
      if (!vok) acc = hssk_d4{0., 0., 0., 0.};
#pragma unroll
      for (int x = TC / 2; x < TC; x++) load_v(wave + x * NW, avs[x]);
      if (KP > KS) {   // parts 2, 3 first; parts 0, 1 add them to their own (same lanes, same addresses: no barrier in between)
        if (q >= KS) {
#pragma unroll
          for (int r = 0; r < 4; r++) Ws[((q - KS) * NC + 16 * b + l4 + 4 * r) * WY_LDW + 16 * a + l15] = acc[r];
        }
        __syncthreads();
        if (q < KS) {
#pragma unroll
          for (int r = 0; r < 4; r++) acc[r] += Ws[(q * NC + 16 * b + l4 + 4 * r) * WY_LDW + 16 * a + l15];
        }
      }
      if (q < KS) {
#pragma unroll
        for (int r = 0; r < 4; r++) Ws[(q * NC + 16 * b + l4 + 4 * r) * WY_LDW + 16 * a + l15] = acc[r];
      }
    }
    __syncthreads();
    // ---- C -= V W: 16-row tiles of the block's rows, dealt to the waves; all column tiles of a row tile together.  The
    // fragments of W (parts summed) stay in registers over the wave's tiles.  The last pair's result goes to global memory,
    // the others' back into the LDS.
    double wf[NC / 16][NU];
#pragma unroll
    for (int bb = 0; bb < NC / 16; bb++)
#pragma unroll
      for (int u = 0; u < NU; u++) {
        double wv = Ws[(16 * bb + l15) * WY_LDW + 4 * u + l4];
#pragma unroll
        for (int qq = 1; qq < KS; qq++) wv += Ws[(qq * NC + 16 * bb + l15) * WY_LDW + 4 * u + l4];
        wf[bb][u] = wv;
      }
    const bool last = pi == np - 1;
    for (int t0 = wave; t0 < ntile; t0 += TC * NW) {
      if (t0 != wave) {
#pragma unroll
        for (int x = 0; x < TC; x++) load_v(t0 + x * NW, avs[x]);
      }
      // (the chunk's tiles side by side: consecutive products are independent)
      hssk_d4 c[TC][NC / 16];
#pragma unroll
      for (int x = 0; x < TC; x++) {
        const int gi = r0 + min(16 * (t0 + x * NW) + l15, rpadp - 1);
#pragma unroll
        for (int bb = 0; bb < NC / 16; bb++)
#pragma unroll
          for (int r = 0; r < 4; r++) c[x][bb][r] = Cs[wy_coff(16 * bb + l4 + 4 * r, ldr) + gi];
      }
#pragma unroll
      for (int u = 0; u < NU; u++)
#pragma unroll
        for (int x = 0; x < TC; x++)
#pragma unroll
          for (int bb = 0; bb < NC / 16; bb++) c[x][bb] = hssk_mfma_f64_16x16x4(wf[bb][u], avs[x][u], c[x][bb]);   // lane holds C[i = l15][j = l4 + 4 r]
#pragma unroll
      for (int x = 0; x < TC; x++) {
        const int t = t0 + x * NW, gi = r0 + 16 * t + l15;
        if (t < ntile && 16 * t + l15 < rrp) {
          if (last) {
#pragma unroll
            for (int bb = 0; bb < NC / 16; bb++)
#pragma unroll
              for (int r = 0; r < 4; r++) {
                const int j = 16 * bb + l4 + 4 * r;
                if (j < ncb) hssk_gstore(Cg, (size_t)gi + (size_t)(j0 + j) * ldc, c[x][bb][r]);
              }
          } else {
#pragma unroll
            for (int bb = 0; bb < NC / 16; bb++)
#pragma unroll
              for (int r = 0; r < 4; r++) Cs[wy_coff(16 * bb + l4 + 4 * r, ldr) + gi] = c[x][bb][r];
          }
        }
      }
    }
  }
}
template <int NC, int T> inline size_t wy_lds_bytes(int rr) {
  constexpr int KP = (T / 64) / (2 * (NC / 16));
  const int ldr = wy_ldr(rr);
  return sizeof(double) * ((size_t)wy_coff(NC - 1, ldr) + ldr + (size_t)(KP > 2 ? 2 : KP) * NC * WY_LDW);
}
template <int NC, int T> bool wy_apply_nc(hssk_ctx* ctx, const std::vector<WyDesc>& d, int rmax) {
  static const size_t lds_cap = hssk_rt::max_lds_per_workgroup();
  const size_t shm = wy_lds_bytes<NC, T>(rmax);
  if (shm > lds_cap) return false;
  std::vector<WyWork> wk;
  for (size_t i = 0; i < d.size(); i++)
    for (int c = 0; c * NC < d[i].nc; c++) wk.push_back(WyWork{(int)i, c});
  if (wk.empty()) return true;
  auto* dd = (const WyDesc*)ctx->stage(d.data(), sizeof(WyDesc) * d.size());
  auto* dw = (const WyWork*)ctx->stage(wk.data(), sizeof(WyWork) * wk.size());
  hssk_rt::allow_dynamic_lds(wy_apply_kernel<NC, T>, shm);
  HSSK_LAUNCH((wy_apply_kernel<NC, T>), dim3((unsigned)wk.size()), dim3(T), shm, ctx->stream, dd, dw);
  return true;
}
inline int wy_mode() {
  static const int mode = [] { const char* e = std::getenv("HSSK_QR_WY"); return e ? std::atoi(e) : 16; }();   // 0: off (two batched GEMMs per panel)
  return mode;
}
// panels of up to `rows` rows can take the fused application (the LDS holds the column block)
bool wy_usable(int rows) {
  static const size_t lds_cap = hssk_rt::max_lds_per_workgroup();
  switch (wy_mode()) {
    case 0: return false;
    default: return wy_lds_bytes<16, 256>(rows) <= lds_cap;
  }
}
// panels per group (QB_GROUP above): the columns beyond a group see its panels in ONE pass (HSSK_QR_GROUP = 1: every panel its own)
inline int wy_group() {
  static const int g = [] { const char* e = std::getenv("HSSK_QR_GROUP"); return std::max(1, std::min(WY_MAXP, e ? std::atoi(e) : WY_MAXP)); }();
  return g;
}
// the fused application for a batch of (pairs, C); false if some block does not fit (callers ask wy_usable first)
bool wy_apply(hssk_ctx* ctx, const std::vector<WyDesc>& d) {
  if (d.empty()) return true;
  const int mode = wy_mode();
  if (mode == 0) return false;
  int rmax = 0;
  for (auto& x : d) {
    if (x.np < 1 || x.np > WY_MAXP || x.rr <= 0) return false;
    for (int i = 0; i < x.np; i++)
      if (x.pr[i].nb > QB || x.pr[i].nb <= 0 || x.pr[i].roff < 0 || x.pr[i].roff % 16 || x.pr[i].rows <= 0 || x.pr[i].roff + x.pr[i].rows > x.rr) return false;
    rmax = std::max(rmax, x.rr);
  }
  // (32 columns per workgroup, and eight waves on 16, were measured slower in every form: factor phase at leaf 512 6.79 and
  // 6.13 against 6.02 ms in the first, 7.06 against 5.57 in a later one -- both spill at the register budget of their occupancy)
  return wy_apply_nc<16, 256>(ctx, d, rmax);
}

void gemm_batch(hssk_ctx* ctx, std::vector<hssk_gemm_desc>& g) {
  if (g.empty()) return;
  if (hssk_gemm_vbatched(ctx, g.data(), (int)g.size())) throw std::runtime_error(hssk_last_error());
}

// factor == false: A already holds the factored panels and work the taus (a previous factor call); only the
// compact-WY pairs are rebuilt and Q is formed
void qr_blocked(hssk_ctx* ctx, const hssk_qr_desc* descs, int count, bool factor) {
  std::vector<size_t> offV(count), offW(count), offR(count);
  size_t tot = 0;
  int pmax = 0;
  bool anyq = false, tall = false;
  for (int i = 0; i < count; i++) {
    const hssk_qr_desc& d = descs[i];
    const size_t kmax = (size_t)std::min(d.rows, d.cols);
    offV[i] = tot; tot += 2 * (size_t)d.rows * kmax;
    offW[i] = tot; tot += (size_t)QB * std::max(std::max(d.cols, d.nq), 1);
    offR[i] = tot; tot += 2 + 2 * QB * QB;   // panel max/min, then G and T of the tall path
    pmax = std::max(pmax, (int)((kmax + QB - 1) / QB));
    anyq = anyq || d.nq > 0;
    tall = tall || d.rows > QB_MAXROWS;
  }
  double* ws = ctx->scratch(sizeof(double) * tot);
  std::vector<TDesc> td;
  std::vector<hssk_gemm_desc> gG, gVT;
  std::vector<hssk_qr_desc> pd;
  std::vector<QPanel> lp;
  std::vector<hssk_gemm_desc> g1, g2;
  std::vector<WyDesc> wy;
  int rows_max = 0;
  for (int i = 0; i < count; i++) rows_max = std::max(rows_max, descs[i].rows);
  const bool fused = wy_usable(rows_max);
  const int GP = fused ? wy_group() : 1;
  auto one_pair = [](const double* V, const double* VT, int nb, int rows, double* C, int ldv, int ldc, int nc) {
    WyDesc w{};
    w.pr[0] = WyPair{V, VT, 0, nb, rows, 0};
    w.C = C; w.ldv = ldv; w.ldc = ldc; w.rr = rows; w.nc = nc; w.np = 1;
    return w;
  };
  // (Q only, from panels factored before: the T factors of all panels depend on nothing -- one launch for all of them)
  const bool all_larft = !factor && !tall;
  for (int p = 0; p < std::max(pmax, 1); p++) {
    pd.clear(); g1.clear(); g2.clear(); td.clear(); gG.clear(); gVT.clear(); wy.clear();
    if (!all_larft || p == 0) lp.clear();
    const int j0 = p * QB, g0 = (p / GP) * GP;
    for (int i = 0; i < count; i++) {
      const hssk_qr_desc& d = descs[i];
      const int kmax = std::min(d.rows, d.cols);
      if (j0 >= kmax && !(factor && p == 0 && d.rdiag)) continue;
      const int nb = std::max(0, std::min(QB, kmax - j0));
      // staircase input (interleaved stack of triangles): the panel's columns are zero from row stair * (j0 + nb) on
      const int rend = (d.stair > 0 && d.nq == 0) ? std::min<long long>(d.rows, (long long)d.stair * (j0 + nb)) : d.rows;
      const int rr = rend - j0;
      double* Ap = d.A + j0 + (size_t)j0 * d.lda;
      double* Vc = ws + offV[i] + j0 + (size_t)j0 * d.rows;
      double* VT = Vc + (size_t)d.rows * kmax;
      double* W = ws + offW[i];
      double* rdp = ws + offR[i];
      lp.push_back(QPanel{Ap, d.work + j0, Vc, VT, rdp, factor ? d.rdiag : nullptr, d.lda, rr, nb, d.rows, p == 0});
      if (tall && nb > 0) {
        double* G = rdp + 2;
        double* T = G + QB * QB;
        gG.push_back(hssk_gemm_desc{Vc, Vc, G, nb, nb, rr, d.rows, d.rows, QB, 1, 0, 1.0, 0.0});
        td.push_back(TDesc{G, d.work + j0, T, nb});
        gVT.push_back(hssk_gemm_desc{Vc, T, VT, rr, nb, nb, d.rows, QB, d.rows, 0, 0, 1.0, 0.0});
      }
      if (nb == 0 || !factor) continue;
      pd.push_back(hssk_qr_desc{Ap, d.lda, rr, nb, nullptr, 0, 0, rdp, d.work + j0});
      // the columns of the panel's own group: this panel alone, now
      const int je = GP > 1 ? (int)std::min<long long>(d.cols, (long long)(g0 + GP) * QB) : d.cols;   // first column beyond the group
      const int nt = je - (j0 + nb);
      if (nt > 0) {
        double* A2 = Ap + (size_t)nb * d.lda;
        if (fused) wy.push_back(one_pair(Vc, VT, nb, rr, A2, d.rows, d.lda, nt));
        else {
          g1.push_back(hssk_gemm_desc{VT, A2, W, nb, nt, rr, d.rows, d.lda, QB, 1, 0, 1.0, 0.0});
          g2.push_back(hssk_gemm_desc{Vc, W, A2, rr, nt, nb, d.rows, QB, d.lda, 0, 0, -1.0, 1.0});
        }
      }
      // the columns beyond, once the group (or the matrix) has its last panel: all panels of the group in one pass
      if (GP > 1 && d.cols > je && (p == g0 + GP - 1 || p == (kmax - 1) / QB)) {
        WyDesc w{};
        const int jg = g0 * QB;
        for (int q = g0; q <= p; q++) {
          const int jq = q * QB, nbq = std::min(QB, kmax - jq);
          const int rendq = (d.stair > 0 && d.nq == 0) ? std::min<long long>(d.rows, (long long)d.stair * (jq + nbq)) : d.rows;
          const double* Vq = ws + offV[i] + jq + (size_t)jq * d.rows;
          w.pr[w.np++] = WyPair{Vq, Vq + (size_t)d.rows * kmax, jq - jg, nbq, rendq - jq, 0};
        }
        w.C = d.A + jg + (size_t)je * d.lda; w.ldv = d.rows; w.ldc = d.lda; w.rr = rend - jg; w.nc = d.cols - je;
        wy.push_back(w);
      }
    }
    if (!pd.empty()) {
      auto* dp = (const hssk_qr_desc*)ctx->stage(pd.data(), sizeof(hssk_qr_desc) * pd.size());
      if (tall) HSSK_LAUNCH(qr_kernel, dim3((unsigned)pd.size()), dim3(QR_THREADS), 0, ctx->stream, dp, 0);   // Level-2 on a 32-column panel
      else HSSK_LAUNCH((qr_reg_kernel<32, 1, 8>), dim3((unsigned)pd.size()), dim3(512), 0, ctx->stream, dp);
    }
    if (!lp.empty() && (!all_larft || p == std::max(pmax, 1) - 1)) {
      auto* dl = (const QPanel*)ctx->stage(lp.data(), sizeof(QPanel) * lp.size());
      if (tall) {
        HSSK_LAUNCH(make_v_kernel, dim3((unsigned)lp.size(), 8), dim3(256), 0, ctx->stream, dl);
        gemm_batch(ctx, gG);
        if (!td.empty()) {
          auto* dt = (const TDesc*)ctx->stage(td.data(), sizeof(TDesc) * td.size());
          HSSK_LAUNCH(larft_small_kernel, dim3((unsigned)td.size()), dim3(64), 0, ctx->stream, dt);
        }
        gemm_batch(ctx, gVT);
      } else {
        int lrmax = 0;
        for (auto& q : lp) lrmax = std::max(lrmax, q.rows);
        if (lrmax <= 256) HSSK_LAUNCH((larft_kernel<256, 256>), dim3((unsigned)lp.size()), dim3(256), 0, ctx->stream, dl);
        else HSSK_LAUNCH((larft_kernel<QB_MAXROWS, 512>), dim3((unsigned)lp.size()), dim3(512), 0, ctx->stream, dl);   // (1024 threads would leave the back substitution 128 registers: it keeps 64 doubles)
      }
    }
    if (fused) {
      if (!wy_apply(ctx, wy)) throw std::runtime_error("hssk_qr: fused block reflector refused a block it had accepted");
    } else {
      gemm_batch(ctx, g1);
      gemm_batch(ctx, g2);
    }
  }
  if (!anyq) return;
  std::vector<EyeDesc> ey;
  for (int i = 0; i < count; i++)
    if (descs[i].nq > 0 && descs[i].rows > 0) ey.push_back(EyeDesc{descs[i].Q, descs[i].ldq, descs[i].rows, descs[i].nq});
  if (ey.empty()) return;
  auto* de = (const EyeDesc*)ctx->stage(ey.data(), sizeof(EyeDesc) * ey.size());
  HSSK_LAUNCH(eye_kernel, dim3((unsigned)ey.size(), 16), dim3(256), 0, ctx->stream, de);
  // Q = H_0 ... H_k I, the groups backwards; inside a group the panels from the last to the first over the block
  // Q(j_g:, j_g:) -- a panel finds exact zeros in its rows of the columns it has not reached yet (W = 0 there)
  for (int g = (pmax - 1) / GP; g >= 0; g--) {
    wy.clear();
    const int g0 = g * GP, jg = g0 * QB;
    for (int ps = std::min(pmax, g0 + GP) - 1; ps >= g0; ps--) {   // (unfused: a pair of products per panel)
      g1.clear(); g2.clear();
      const int j0 = ps * QB;
      for (int i = 0; i < count && !fused; i++) {
        const hssk_qr_desc& d = descs[i];
        const int kmax = std::min(d.rows, d.cols);
        if (j0 >= kmax || d.nq <= j0) continue;
        const int nb = std::min(QB, kmax - j0), rr = d.rows - j0, cq = d.nq - j0;
        double* Vc = ws + offV[i] + j0 + (size_t)j0 * d.rows;
        double* VT = Vc + (size_t)d.rows * kmax;
        double* W = ws + offW[i];
        double* Qb = d.Q + j0 + (size_t)j0 * d.ldq;
        g1.push_back(hssk_gemm_desc{Vc, Qb, W, nb, cq, rr, d.rows, d.ldq, QB, 1, 0, 1.0, 0.0});
        g2.push_back(hssk_gemm_desc{VT, W, Qb, rr, cq, nb, d.rows, QB, d.ldq, 0, 0, -1.0, 1.0});
      }
      gemm_batch(ctx, g1);
      gemm_batch(ctx, g2);
    }
    if (!fused) continue;
    for (int i = 0; i < count; i++) {
      const hssk_qr_desc& d = descs[i];
      const int kmax = std::min(d.rows, d.cols);
      if (jg >= kmax || d.nq <= jg) continue;
      WyDesc w{};
      for (int q = std::min(pmax, g0 + GP) - 1; q >= g0; q--) {
        const int jq = q * QB;
        if (jq >= kmax || d.nq <= jq) continue;
        const double* Vq = ws + offV[i] + jq + (size_t)jq * d.rows;
        w.pr[w.np++] = WyPair{Vq + (size_t)d.rows * kmax, Vq, jq - jg, std::min(QB, kmax - jq), d.rows - jq, 0};   // Q <- Q - (V T)(V^T Q): the roles of the pair swapped
      }
      w.C = d.Q + jg + (size_t)jg * d.ldq; w.ldv = d.rows; w.ldc = d.ldq; w.rr = d.rows - jg; w.nc = d.nq - jg;
      wy.push_back(w);
    }
    if (!wy_apply(ctx, wy)) throw std::runtime_error("hssk_qr: fused block reflector refused a block it had accepted");
  }
}

bool force_blocked() {
  static const bool f = [] { const char* e = std::getenv("HSSK_QR_BLOCKED"); return e && e[0] == '1'; }();
  return f;
}

}  // namespace

extern "C" int hssk_qr_vbatched(hssk_ctx* ctx, const hssk_qr_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  int rmax = 0, cmax = 0, qmax = 0;
  for (int i = 0; i < count; i++) {
    rmax = std::max(rmax, descs[i].rows);
    cmax = std::max(cmax, descs[i].cols);
    qmax = std::max(qmax, descs[i].nq);
  }
  // register-resident kernels when the largest panel of the batch fits (rows <= 16 RT, cols <= 4 NW CT; the register
  // tiles of the widest variants, 13 x 7 and 16 x 6 doubles per lane, need the 256 VGPRs of an 8-wave workgroup)
  // (HSSK_QR_BLOCKED_ROWS: batches whose tallest panel has at least that many rows, and 96 columns, take the blocked path too)
  static const int blk_rows = [] { const char* e = std::getenv("HSSK_QR_BLOCKED_ROWS"); return e ? std::atoi(e) : 1 << 30; }();
  if (force_blocked() || rmax > 256 || cmax > 224 || (rmax > 208 && cmax > 192) || (rmax >= blk_rows && cmax >= 96 && wy_usable(rmax))) {
    qr_blocked(ctx, descs, count, true);
    hssk_rt::check_launch();
    return 0;
  }
  auto* dd = (const hssk_qr_desc*)ctx->stage(descs, sizeof(*descs) * count);
  if (rmax <= 64 && cmax <= 64) launch_qr_reg<4, 1, 16>(ctx, dd, count);
  else if (rmax <= 128 && cmax <= 128) launch_qr_reg<8, 2, 16>(ctx, dd, count);
  else if (cmax <= 128 && rmax <= 208) launch_qr_reg<13, 4, 8>(ctx, dd, count);
  else if (cmax <= 128) launch_qr_reg<16, 4, 8>(ctx, dd, count);
  else if (rmax > 208) launch_qr_reg<16, 6, 8>(ctx, dd, count);   // <= 256 rows x 192 columns
  else if (cmax <= 160) launch_qr_reg<13, 5, 8>(ctx, dd, count);   // (a step costs per column slot: the 195 x ~160 ULV panels of N = 1e5 need five or
  else if (cmax <= 192) launch_qr_reg<13, 6, 8>(ctx, dd, count);   //  six, not seven)
  else launch_qr_reg<13, 7, 8>(ctx, dd, count);
  // Q is then formed by a second, barrier-free launch over blocks of 64 columns.  HSSK_QR_FORMQ_WY=1: for the taller panels
  // from the compact-WY pairs of 32 reflectors each on the matrix cores instead (qr_blocked with the factorization skipped:
  // the T factors of all panels in one launch, then the fused block reflector over the groups of panels) -- measured SLOWER
  // on the 512 panels 196 x 155 of the ULV leaves at N = 1e5 (T factors 0.28 ms + two launches 0.57 ms against 0.70 ms of
  // formq_reg; factor phase 2.81 against 2.55 ms, gpurun_out/r04q), as is the whole blocked path from 128 rows on
  // (HSSK_QR_BLOCKED_ROWS=128: factor 3.37, tree 4.53 against 2.85 ms): off.
  static const bool q_wy = [] { const char* e = std::getenv("HSSK_QR_FORMQ_WY"); return e && e[0] == '1'; }();
  if (qmax > 0) {
    if (q_wy && rmax >= 128 && cmax >= 64 && wy_usable(rmax)) qr_blocked(ctx, descs, count, false);
    else formq_reg(ctx, dd, descs, count, rmax);
  }
  hssk_rt::check_launch();
  HSSK_API_END
}

// Q only, from panels factored by an earlier hssk_qr_vbatched call (A = reflectors + R, work = taus)
extern "C" int hssk_formq_vbatched(hssk_ctx* ctx, const hssk_qr_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  int rmax = 0, cmax = 0, qmax = 0;
  for (int i = 0; i < count; i++) {
    rmax = std::max(rmax, descs[i].rows);
    cmax = std::max(cmax, descs[i].cols);
    qmax = std::max(qmax, descs[i].nq);
  }
  if (qmax <= 0) return 0;
  if (force_blocked() || rmax > 256) {
    qr_blocked(ctx, descs, count, false);
  } else {
    static const bool q_wy = [] { const char* e = std::getenv("HSSK_QR_FORMQ_WY"); return e && e[0] == '1'; }();
    if (q_wy && rmax >= 128 && cmax >= 64 && wy_usable(rmax)) qr_blocked(ctx, descs, count, false);
    else {
      auto* dd = (const hssk_qr_desc*)ctx->stage(descs, sizeof(*descs) * count);
      formq_reg(ctx, dd, descs, count, rmax);   // rmax <= 256 (taller panels took the blocked path above)
    }
  }
  hssk_rt::check_launch();
  HSSK_API_END
}

extern "C" int hssk_tpqr_vbatched(hssk_ctx* ctx, const hssk_tpqr_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  int mmax = 0;
  for (int i = 0; i < count; i++) {
    if (descs[i].m < 0 || descs[i].ld1 < descs[i].m || descs[i].ld2 < descs[i].m) HSSK_UNSUPPORTED("bad triangle descriptor");
    mmax = std::max(mmax, descs[i].m);
  }
  if (mmax == 0) return 0;
  if (mmax > 224) HSSK_UNSUPPORTED("triangles beyond 224 columns");   // (beyond the register tile: the caller stacks the triangles and calls hssk_qr_vbatched)
  auto* dd = (const hssk_tpqr_desc*)ctx->stage(descs, sizeof(*descs) * count);
  if (mmax <= 64) HSSK_LAUNCH((tpqr_reg_kernel<4, 2, 8>), dim3((unsigned)count), dim3(512), 0, ctx->stream, dd);
  else if (mmax <= 128) HSSK_LAUNCH((tpqr_reg_kernel<8, 4, 8>), dim3((unsigned)count), dim3(512), 0, ctx->stream, dd);
  else if (mmax <= 208) HSSK_LAUNCH((tpqr_reg_kernel<13, 7, 8>), dim3((unsigned)count), dim3(512), 0, ctx->stream, dd);
  else HSSK_LAUNCH((tpqr_reg_kernel<14, 7, 8>), dim3((unsigned)count), dim3(512), 0, ctx->stream, dd);
  hssk_rt::check_launch();
  HSSK_API_END
}
