// Entry (i, j) of a matrix given by a formula (hssk_gen, include/hssk.h): the ONE place the formulas live -- the fill
// kernels, the element gather and the tiles generated inside the sketch kernel all call this, so a generated operand is bit
// for bit the stored one.
#pragma once
#include "hssk.h"
#include "hssk_device.h"

__device__ __forceinline__ double hssk_gen_eval(const hssk_gen& g, int i, int j) {
  // Toeplitz test matrix of the reference's test driver (test/test_HSS_seq.cpp:75-78; upper triangle :86-90)
  const int d = i > j ? i - j : j - i;
  double v = 1.0 / (1.0 + (double)d);
  if (g.kind == HSSK_GEN_TOEPLITZ_UPPER && i > j) v = 0.;
  return v;
}
