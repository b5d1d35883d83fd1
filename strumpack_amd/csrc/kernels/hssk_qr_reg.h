// Register-resident batched Householder QR and Q formation as workgroup-wide device bodies: the kernels of hssk_qr.hip
// (hssk_qr_vbatched / hssk_formq_vbatched) and the fused per-node ULV step (hssk_ulv_node.hip) instantiate them.
#pragma once
#include "hssk_device.h"
#include "hssk_internal.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Register-resident variant: the whole panel lives in the VGPRs of one workgroup of NW wave64, ONE COLUMN PER 16-LANE
// DPP ROW: column j belongs to row-group g = j % NC (wave g / 4, lanes 16 (g % 4) .. +15), slot j / NC, NC = 4 NW; row i
// sits in lane i % 16 of the group, register i / 16, so a lane holds a[CT][RT] doubles.  A Householder step costs one LDS
// broadcast of the reflector (double-buffered: one barrier per step) and, per owned column, RT fmas + four in-row DPP
// steps (hssk_row_sum: every lane of the group ends up with the dot product, no readlane) + RT fmas, four columns per
// wave at a time -- no global or L2 traffic inside the factorization, which is what bounds the global-memory kernel
// above.  (The first version spread a column over all 64 lanes: 6 DPP stages + a readlane per column.)  The step loop is
// unrolled over the slot and the register index of the diagonal row, so every register access is static.
// Capacity: rows <= 16 RT, cols <= 4 NW CT.
// ------------------------------------------------------------------------------------------------
template <int RT, int CT, int NW>
__device__ __forceinline__ void qr_reg_body(const hssk_qr_desc& p) {
  constexpr int NC = NW * 4, SPC = NC / 16;
  HSSK_SHARED double s_v[2 * 16 * RT];
  HSSK_SHARED double s_tau[NC * CT];
  HSSK_SHARED double s_rd[2];
  HSSK_SHARED int s_stop;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, sub = lane >> 4, grp = wave * 4 + sub;
  const int rows = p.rows, cols = p.cols;
  const int kmax = rows < cols ? rows : cols;
  const bool r_only = p.r_only != 0 && p.nq == 0;
  const bool may_stop = p.rdiag && p.nq == 0 && (p.stop_rel > 0. || p.stop_abs > 0.);
  if (tid == 0) s_stop = 0;
  bool done = false;
  double a[CT][RT];
#pragma unroll
  for (int c = 0; c < CT; c++)
#pragma unroll
    for (int r = 0; r < RT; r++) {
      const int row = l16 + 16 * r, col = grp + NC * c;
      a[c][r] = (row < rows && col < cols) ? p.A[row + (size_t)col * p.lda] : 0.;
    }
  // ---- factorization.  Step k = 16 rk + lk is owned by group g = k % NC of slot kc = k / NC
#pragma clang loop unroll(full)
  for (int kc = 0; kc < CT; kc++) {
#pragma clang loop unroll(full)
    for (int rq = 0; rq < SPC; rq++) {
      const int rk = kc * SPC + rq < RT ? kc * SPC + rq : RT - 1;   // (static; slots beyond the last row register run no step)
      const int nlk = (kc * SPC + rq < RT && !done) ? min(16, kmax - 16 * rk) : 0;
      for (int lk = 0; lk < nlk; lk++) {
        const int k = 16 * rk + lk;
        const int g = rq * 16 + lk, kw = g >> 2, ks = g & 3;
        double* sv = s_v + (k & 1) * 16 * RT;
        if (wave == kw) {
          const bool own = sub == ks;
          double s = 0.;
#pragma unroll
          for (int r = 0; r < RT; r++) {
            const int row = l16 + 16 * r;
            if (r > rk || (r == rk && l16 > lk)) s += a[kc][r] * a[kc][r];   // (static in r: rows beyond `rows` hold zeros)
          }
          const double alpha = hssk_shfl(a[kc][rk], (lane & 48) | lk);
          s = hssk_row_sum(s);
          double tau = 0., beta = alpha, scal = 1.;
          if (s != 0.) {
            double nrm = sqrt(alpha * alpha + s);
            beta = alpha >= 0. ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            scal = 1. / (alpha - beta);
          }
          if (own) {
#pragma unroll
            for (int r = 0; r < RT; r++) {
              const int row = l16 + 16 * r;
              if (r > rk || (r == rk && l16 > lk)) a[kc][r] *= scal;
              sv[row] = r < rk ? 0. : (r > rk ? a[kc][r] : (l16 > lk ? a[kc][r] : (l16 == lk ? 1. : 0.)));
              if (r == rk && l16 == lk) a[kc][r] = beta;
            }
            if (l16 == 0) {
              s_tau[k] = tau;
              const double ab = fabs(beta);
              if (k == 0) { s_rd[0] = ab; s_rd[1] = ab; }
              else { if (ab > s_rd[0]) s_rd[0] = ab; if (ab < s_rd[1]) s_rd[1] = ab; }
              // (the R-diagonal test is settled: see hssk_qr_desc.stop_rel)
              if (may_stop && (ab < p.stop_abs || ab < p.stop_rel * s_rd[0])) s_stop = 1;
            }
          }
        }
        __syncthreads();
        if (may_stop && s_stop) { done = true; break; }
        const double tau = s_tau[k];
        if (tau != 0.) {
          // The reflector is zero above row k: the row registers below rk (static in the unrolled step loop) take no part --
          // neither in the dot products (their terms are exact zeros: the sums keep their values bit for bit, the even / odd
          // pairing of the partial sums included) nor in the update.  On the square-ish panels (a 208 x 195 TSQR chunk, the
          // 195 x 170 panels of the ULV factorization) that is half the arithmetic of the factorization.
          double vr[RT];
#pragma unroll
          for (int r = 0; r < RT; r++) vr[r] = r >= rk ? sv[l16 + 16 * r] : 0.;
          // all dot products of the wave's slots first, their row sums stage by stage (independent chains in flight)
          double dot[CT];
#pragma unroll
          for (int c = 0; c < CT; c++) {
            double d0 = 0., d1 = 0.;
            if (c >= kc) {
#pragma unroll
              for (int r = 0; r < RT; r++) {
                if (r >= rk) {
                  if ((r & 1) == 0) d0 += vr[r] * a[c][r];
                  else d1 += vr[r] * a[c][r];
                }
              }
            }
            dot[c] = d0 + d1;
          }
          hssk_row_sum_n(dot);
#pragma unroll
          for (int c = kc; c < CT; c++) {
            const int col = grp + NC * c;
            const double f = (col > k && col < cols) ? dot[c] * tau : 0.;
#pragma unroll
            for (int r = 0; r < RT; r++)
              if (r >= rk) a[c][r] -= f * vr[r];
          }
        }
      }
    }
  }
  // factored panel (R + reflectors) back to A, taus to the work array
#pragma unroll
  for (int c = 0; c < CT; c++)
#pragma unroll
    for (int r = 0; r < RT; r++) {
      const int row = l16 + 16 * r, col = grp + NC * c;
      if (row < rows && col < cols && (!r_only || row <= col)) p.A[row + (size_t)col * p.lda] = a[c][r];
    }
  __syncthreads();
  if (!r_only)
    for (int k = tid; k < kmax; k += NW * 64) p.work[k] = s_tau[k];
  if (p.rdiag && tid == 0) { p.rdiag[0] = kmax ? s_rd[0] : 0.; p.rdiag[1] = kmax ? s_rd[1] : 0.; }
}

// Q(:, j0 : j0 + 4 NW CT) = H_0 ... H_{kmax-1} I(:, same columns): every 16-lane group owns CT columns in registers (same
// layout as above) and applies the reflectors, which the workgroup stages through LDS in chunks (see the kernel).  H_k
// leaves column j untouched for k > j, so the sweep starts at the last column of the block.
struct QBlock {
  int prob, block;
};
constexpr int FQ_KC = 16;   // reflectors staged in LDS at a time (formq_reg_kernel)
template <int RT, int CT, int NW>
__device__ __forceinline__ void formq_reg_body(const hssk_qr_desc& p, int block) {
  constexpr int NC = NW * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, grp = wave * 4 + (lane >> 4);
  const int rows = p.rows, cols = p.cols, nq = p.nq;
  const int kmax = rows < cols ? rows : cols;
  const int j0 = block * NC * CT;
  const double* __restrict__ A = p.A;
  const double* __restrict__ taus = p.work;
  double a[CT][RT];
#pragma unroll
  for (int c = 0; c < CT; c++)
#pragma unroll
    for (int r = 0; r < RT; r++) a[c][r] = (l16 + 16 * r == j0 + grp + NC * c) ? 1. : 0.;
  int kstart = j0 + NC * CT - 1;
  if (kstart > kmax - 1) kstart = kmax - 1;
  // The reflectors come through LDS, FQ_KC at a time: the workgroup loads a chunk once, coalesced, one chunk ahead of its
  // use (registers -> LDS behind a barrier per chunk), and every lane reads its rows of the current reflector from there.
  // (The first version had each of the 4 NW lane groups fetch every reflector from global memory itself, one step ahead:
  // 4 NW times the traffic through the vector L1 and a step that could not be shorter than one L2 / HBM round trip --
  // 1.75 us per reflector on the 195 x 195 leaf blocks of N = 1e5, 1.1 ms for the level.)
  constexpr int VL = 16 * RT;                       // padded reflector length
  constexpr int NL = (FQ_KC * VL + NW * 64 - 1) / (NW * 64);   // chunk elements per thread
  HSSK_SHARED double s_v[2 * FQ_KC * VL];
  HSSK_SHARED double s_tau[2 * FQ_KC];
  const int tid = threadIdx.x;
  const int nchunk = kstart >= 0 ? kstart / FQ_KC + 1 : 0;   // chunk ch covers k = kstart - ch FQ_KC - kk, kk < FQ_KC
  double stage[NL];
  double stage_tau = 0.;
  auto fetch = [&](int ch) {
#pragma unroll
    for (int u = 0; u < NL; u++) {
      const int e = tid + u * NW * 64, kk = e / VL, row = e % VL;
      const int k = kstart - ch * FQ_KC - kk;
      double v = 0.;
      if (e < FQ_KC * VL && k >= 0) v = (row > k && row < rows) ? hssk_gload(A, (size_t)row + (size_t)k * p.lda) : (row == k ? 1. : 0.);
      stage[u] = v;
    }
    if (tid < FQ_KC) { const int k = kstart - ch * FQ_KC - tid; stage_tau = k >= 0 ? taus[k] : 0.; }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NL; u++) {
      const int e = tid + u * NW * 64;
      if (e < FQ_KC * VL) s_v[buf * FQ_KC * VL + e] = stage[u];
    }
    if (tid < FQ_KC) s_tau[buf * FQ_KC + tid] = stage_tau;
  };
  if (nchunk > 0) { fetch(0); commit(0); }
  __syncthreads();
  for (int ch = 0; ch < nchunk; ch++) {
    const int buf = ch & 1;
    if (ch + 1 < nchunk) fetch(ch + 1);
    for (int kk = 0; kk < FQ_KC; kk++) {
      const int k = kstart - ch * FQ_KC - kk;
      if (k < 0) break;
      const double tau = s_tau[buf * FQ_KC + kk];
      if (tau == 0.) continue;
      const double* sv = s_v + (buf * FQ_KC + kk) * VL;
      double vr[RT];
#pragma unroll
      for (int r = 0; r < RT; r++) vr[r] = sv[l16 + 16 * r];
      double dot[CT];
#pragma unroll
      for (int c = 0; c < CT; c++) {
        double d0 = 0., d1 = 0.;
#pragma unroll
        for (int r = 0; r + 1 < RT; r += 2) { d0 += vr[r] * a[c][r]; d1 += vr[r + 1] * a[c][r + 1]; }
        if (RT & 1) d0 += vr[RT - 1] * a[c][RT - 1];
        dot[c] = d0 + d1;
      }
      hssk_row_sum_n(dot);
#pragma unroll
      for (int c = 0; c < CT; c++) {
        const int col = j0 + grp + NC * c;
        const double f = (col >= k && col < nq) ? dot[c] * tau : 0.;
#pragma unroll
        for (int r = 0; r < RT; r++) a[c][r] -= f * vr[r];
      }
    }
    if (ch + 1 < nchunk) commit(buf ^ 1);   // (the other buffer was last read in the previous chunk, before its closing barrier)
    __syncthreads();
  }
#pragma unroll
  for (int c = 0; c < CT; c++)
#pragma unroll
    for (int r = 0; r < RT; r++) {
      const int row = l16 + 16 * r, col = j0 + grp + NC * c;
      if (row < rows && col < nq) p.Q[row + (size_t)col * p.ldq] = a[c][r];
    }
}


}  // namespace
