// Back substitution with an upper triangular matrix of order <= 64, one right-hand side per thread:  x <- U^{-1} x  for the
// thread's own x (registers).  U's strictly upper part sits in LDS (s_R[i + l * HSSK_BACKSUB_LD], zero for i >= l and for
// l >= rank), s_rd holds the reciprocals of its diagonal (zero beyond the rank): all reads are broadcasts.  Unrolled over
// 8-row blocks, blocks at or above the rank are skipped by uniform branches: every register index is static and there is
// no cross-lane step -- rank^2 / 2 fmas per right-hand side.  Shared by the X = R11^{-1} R12 pass of the interpolative
// decomposition (hssk_id.hip) and the inversion of the 64 x 64 diagonal blocks of the ULV factors (hssk_sweep.hip).
#pragma once
#include "hssk_device.h"

constexpr int HSSK_BACKSUB_LD = 66;

__device__ __forceinline__ void hssk_backsub64(double (&x)[64], const double* s_R, const double* s_rd, int rank) {
  constexpr int LR = HSSK_BACKSUB_LD;
#pragma unroll
  for (int b = 7; b >= 0; b--) {
    if (8 * b < rank) {
#pragma unroll
      for (int lb = 7; lb > b; lb--) {
        if (8 * lb < rank) {
#pragma unroll
          for (int l = 8 * lb; l < 8 * lb + 8; l++) {
#pragma unroll
            for (int i = 8 * b; i < 8 * b + 8; i++) x[i] -= s_R[i + l * LR] * x[l];
            if (l & 1) hssk_sched_barrier();   // (keeps the scheduler from hoisting hundreds of LDS reads into registers)
          }
        }
      }
#pragma unroll
      for (int i = 8 * b + 7; i >= 8 * b; i--) {
#pragma unroll
        for (int l = i + 1; l < 8 * b + 8; l++) x[i] -= s_R[i + l * LR] * x[l];
        x[i] *= s_rd[i];
      }
      hssk_sched_barrier();
    }
  }
}
