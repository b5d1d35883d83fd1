// Register-resident truncated column-pivoted QR of one sample panel by ONE workgroup: the body shared by id_reg_kernel
// (hssk_id.hip: one launch per tree level) and the single-launch inner-tree pass (hssk_tree.hip).
#pragma once
#include "hssk.h"
#include "hssk_device.h"

// ------------------------------------------------------------------------------------------------
// Register-resident variant (same layout as qr_reg_kernel in hssk_qr.hip): the d x m sample panel lives in the VGPRs of
// one workgroup, ONE COLUMN PER 16-LANE DPP ROW -- column j belongs to row-group g = j % NC (wave g / 4, lanes 16 (g % 4)
// .. +15), slot j / NC, with NC = 4 NW groups per workgroup; row i sits in lane i % 16 of the group, register i / 16.
// A column dot product is then RT fmas and four in-row DPP steps (hssk_row_sum), with every lane of the group holding the
// result: no cross-row stages, no readlane, and a wave works on four columns at once.  (The first version spread a
// column over all 64 lanes: 6 DPP stages + a readlane per column, ~3x the instructions per Householder step.)
// Columns never move: pivoting only records the order (s_perm) and a per-lane "used" mask; the panel is written back
// once, in pivoted order, for the triangular solve.  The step loop is unrolled over the register index of the pivot row
// (k = 16 rk + lk), so R(k, j) is a statically indexed register.  Per Householder step: one LDS hop for the pivot search,
// one LDS broadcast of the reflector (both double-buffered, two barriers), and per owned column 2 RT fmas, the row sum,
// one in-group shuffle for R(k, j) and the dlaqp2 norm down-date.
// Capacity: d <= 16 RT, m <= 4 NW CT.
// ------------------------------------------------------------------------------------------------
// Returns the rank (uniform over the workgroup); W holds [R11 R12] in its first `rank` rows at the pivoted column positions,
// perm the pivoted order.  *p.rank is NOT written (the callers do, after whatever they finish first).  Ends with a barrier.
template <int RT, int CT, int NW>
__device__ __forceinline__ int id_reg_body(const hssk_id_desc& p) {
  constexpr int NC = NW * 4;
  HSSK_SHARED double s_v[2 * 16 * RT];
  HSSK_SHARED double s_vn1[NC * CT];
  HSSK_SHARED double s_vn2[NC * CT];
  HSSK_SHARED double s_val[2 * NW];
  HSSK_SHARED int s_idx[2 * NW];
  HSSK_SHARED double s_tau[2];
  HSSK_SHARED double s_r00;
  HSSK_SHARED int s_stop;
  HSSK_SHARED int s_perm[NC * CT];
  HSSK_SHARED int s_pos[NC * CT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, sub = lane >> 4, grp = wave * 4 + sub;
  const int d = p.d, m = p.m, ld = p.ldw;
  const int kmax = d < m ? d : m;
  const double tol3z = 1.4901161193847656e-08;  // sqrt(eps)
  const double* __restrict__ in = p.src ? p.src : p.W;
  const int ldin = p.src ? p.lds : ld;
  double a[CT][RT];
#pragma unroll
  for (int c = 0; c < CT; c++) {
    const int col = grp + NC * c;
    double s = 0.;
#pragma unroll
    for (int r = 0; r < RT; r++) {
      const int row = l16 + 16 * r;
      a[c][r] = (row < d && col < m) ? in[row + (size_t)col * ldin] : 0.;
      s += a[c][r] * a[c][r];
    }
    s = hssk_row_sum(s);
    // squared partial norms: dlaqp2's down-date  vn1 *= sqrt(1 - (R_kj/vn1)^2)  is  N1 -= R_kj^2 and its
    // cancellation guard  (1-(R/vn1)^2) (vn1/vn2)^2 <= sqrt(eps)  is  N1_new <= sqrt(eps) N2: no div / sqrt
    if (l16 == 0 && col < m) { s_vn1[col] = s; s_vn2[col] = s; }
  }
  if (tid == 0) { s_stop = 0; s_r00 = 0.; }
  for (int j = tid; j < NC * CT; j += NW * 64) s_pos[j] = -1;
  unsigned used = 0;  // bit c: this lane's column of slot c has been chosen as a pivot (uniform over the 16 lanes of a group)
  __syncthreads();

  int rank = kmax;
  bool done = false;
#pragma clang loop unroll(full)
  for (int rk = 0; rk < RT; rk++) {
    const int nlk = done ? 0 : min(16, kmax - 16 * rk);
    for (int lk = 0; lk < nlk; lk++) {
      const int k = rk * 16 + lk;
      const int pb = k & 1;
      double* sv = s_v + pb * 16 * RT;
      // ---- 1. pivot: first arg max over the unused columns
      {
        double bv = -1.;
        int bi = 0x7fffffff;
#pragma unroll
        for (int c = 0; c < CT; c++) {
          const int col = grp + NC * c;
          if (col < m && !((used >> c) & 1u)) {
            const double v = s_vn1[col];
            if (v > bv) { bv = v; bi = col; }  // the columns of a group are visited in increasing order
          }
        }
        // best of the wave's four groups (bv / bi are uniform over the 16 lanes of a group)
        double wv = hssk_bcast_lane(bv, 0);
        int wi = hssk_bcast_lane_i(bi, 0);
#pragma unroll
        for (int q = 1; q < 4; q++) {
          const double v = hssk_bcast_lane(bv, 16 * q);
          const int ix = hssk_bcast_lane_i(bi, 16 * q);
          if (v > wv || (v == wv && ix < wi)) { wv = v; wi = ix; }
        }
        if (lane == 0) { s_val[pb * NW + wave] = wv; s_idx[pb * NW + wave] = wi; }
      }
      __syncthreads();
      // best of the NW wave candidates: every 16-lane row loads them (lane l takes candidate l mod NW) and reduces on the
      // DPP network -- four exchange steps instead of a serial scan of NW LDS values in every lane
      static_assert(NW == 8 || NW == 16, "one candidate per lane of a 16-lane row");
      double gv = s_val[pb * NW + (lane & (NW - 1))];
      int pcol = s_idx[pb * NW + (lane & (NW - 1))];
      hssk_row_argmax(gv, pcol);
      const int pg = pcol % NC, cp = pcol / NC, wp = pg >> 2, sp = pg & 3;
      // ---- 2. reflector from the pivot column (dlarfg): the owner wave computes, the owner group commits
      // (static loop over the slots: the pivot column is used in place, no register copy)
      if (wave == wp) {
        const bool own = sub == sp;
#pragma unroll
        for (int c = 0; c < CT; c++)
          if (c == cp) {
            double s = 0.;
            // (row registers before the pivot row's are above it, those behind it below; only register rk needs a lane test.
            // Rows beyond d hold zeros.)
#pragma unroll
            for (int r = 0; r < RT; r++) {
              if (r > rk) s += a[c][r] * a[c][r];
              else if (r == rk && l16 > lk) s += a[c][r] * a[c][r];
            }
            const double alpha = hssk_shfl(a[c][rk], (lane & 48) | lk);
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
                if (r < rk) sv[row] = 0.;
                else if (r > rk) { a[c][r] *= scal; sv[row] = a[c][r]; }
                else {
                  if (l16 > lk) a[c][r] *= scal;
                  sv[row] = l16 > lk ? a[c][r] : (l16 == lk ? 1. : 0.);
                  if (l16 == lk) a[c][r] = beta;
                }
              }
              used |= 1u << c;
              if (l16 == 0) {
                s_tau[pb] = tau;
                s_perm[k] = pcol;
                const double ab = fabs(beta);
                if (k == 0) s_r00 = ab;
                const double r00 = (k == 0) ? ab : s_r00;
                // dgeqp3tol.f:225-232 (0/0 is NaN -> false, then the absolute test decides)
                if ((r00 != 0. && ab / r00 <= p.rtol) || ab <= p.atol) s_stop = 1;
              }
            }
          }
      }
      __syncthreads();
      if (s_stop) { rank = k; done = true; break; }
      const double tau = s_tau[pb];
      // ---- 3. apply H to the unused columns and down-date their norms (dlaqp2)
      double vr[RT];
#pragma unroll
      for (int r = 0; r < RT; r++) vr[r] = sv[l16 + 16 * r];
      // all dot products of the wave's slots first, their row sums stage by stage (independent chains in flight)
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
      double newk[CT];
#pragma unroll
      for (int c = 0; c < CT; c++) {
        const int col = grp + NC * c;
        const double f = (col < m && !((used >> c) & 1u)) ? dot[c] * tau : 0.;
#pragma unroll
        for (int r = 0; r < RT; r++) a[c][r] -= f * vr[r];
        newk[c] = hssk_shfl(a[c][rk], (lane & 48) | lk);  // R(k, col)
      }
#pragma unroll
      for (int c = 0; c < CT; c++) {
        const int col = grp + NC * c;
        const bool act = col < m && !((used >> c) & 1u);
        double n1 = 0., n2 = 0., newn1 = 0.;
        int recompute = 0;
        if (act) {
          n1 = s_vn1[col]; n2 = s_vn2[col];
          newn1 = n1 - newk[c] * newk[c];
          newn1 = newn1 > 0. ? newn1 : 0.;
          recompute = (n1 != 0.) && (newn1 <= tol3z * n2);
        }
        if (hssk_any(recompute)) {
          double s2 = 0.;
#pragma unroll
          for (int r = 0; r < RT; r++) {
            const int row = l16 + 16 * r;
            if (r > rk || (r == rk && l16 > lk)) s2 += a[c][r] * a[c][r];
          }
          s2 = hssk_row_sum(s2);
          if (recompute) {
            newn1 = s2;
            if (l16 == 0) s_vn2[col] = newn1;
          }
        }
        if (act && l16 == 0) s_vn1[col] = newn1;
      }
    }
  }
  if (rank > p.max_rank) rank = p.max_rank;
  // ---- pivoted column positions: skeleton columns first (pivot order), then the rest
  __syncthreads();
  for (int j = tid; j < rank; j += NW * 64) s_pos[s_perm[j]] = j;
  __syncthreads();
  {
    // a column that was never a pivot goes behind the skeleton, in index order: rank + (number of such columns before it)
    // (m <= 4 NW CT <= 64 NW: one column per thread)
    int mine = -1;
    if (tid < m && s_pos[tid] < 0) {
      int c = 0;
      for (int e = 0; e < tid; e++) c += s_pos[e] < 0;
      mine = rank + c;
    }
    __syncthreads();
    if (mine >= 0) s_pos[tid] = mine;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < CT; c++) {
    const int col = grp + NC * c;
    if (col < m) {
      const int pos = s_pos[col];
      // only R11 and R12 (the first `rank` rows) are read again: X = R11^{-1} R12, the rest of the panel is dead
#pragma unroll
      for (int r = 0; r < RT; r++) {
        const int row = l16 + 16 * r;
        if (16 * r < rank && row < d) p.W[row + (size_t)pos * ld] = a[c][r];
      }
      if (l16 == 0) p.perm[pos] = col;
    }
  }
  __syncthreads();
  return rank;
}
