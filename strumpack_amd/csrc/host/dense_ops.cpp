// strumpack_amd_dense_gemm: C = alpha op(A) op(B) + beta C for HOST operands on the MI355X -- what the free gemm() of
// DenseMatrix.hpp (dense/DenseMatrix.hpp:1346-1360 of the reference) hands over when the product is large enough to pay
// for the transfers.  One batched-GEMM launch of include/hssk.h on a private context; returns non-zero (the caller then
// computes on the host) when no device is available.
#include <mutex>

#include "hssk.h"

namespace {
std::mutex g_mu;
hssk_ctx* g_ctx = nullptr;
bool g_failed = false;
}  // namespace

extern "C" int strumpack_amd_dense_gemm(char ta, char tb, int m, int n, int k, double alpha, const double* A, int lda,
                                        const double* B, int ldb, double beta, double* C, int ldc) {
  if (m <= 0 || n <= 0) return 0;
  std::lock_guard<std::mutex> g(g_mu);
  if (g_failed) return 1;
  if (!g_ctx && hssk_ctx_create(&g_ctx, 0)) { g_failed = true; g_ctx = nullptr; return 1; }
  const bool TA = !(ta == 'N' || ta == 'n'), TB = !(tb == 'N' || tb == 'n');
  const int ar = TA ? k : m, ac = TA ? m : k, br = TB ? n : k, bc = TB ? k : n;
  const long long sa = (long long)sizeof(double) * ar * (ac > 0 ? ac : 1), sb = (long long)sizeof(double) * br * (bc > 0 ? bc : 1),
                  sc = (long long)sizeof(double) * m * n;
  double* dA = (double*)hssk_malloc(sa > 8 ? sa : 8);
  double* dB = (double*)hssk_malloc(sb > 8 ? sb : 8);
  double* dC = (double*)hssk_malloc(sc);
  int rc = (!dA || !dB || !dC) ? 1 : 0;
  if (!rc && k > 0) {
    rc |= hssk_memcpy2d_h2d(g_ctx, dA, sizeof(double) * ar, A, sizeof(double) * lda, sizeof(double) * ar, ac);
    rc |= hssk_memcpy2d_h2d(g_ctx, dB, sizeof(double) * br, B, sizeof(double) * ldb, sizeof(double) * br, bc);
  }
  if (!rc && beta != 0.0) rc |= hssk_memcpy2d_h2d(g_ctx, dC, sizeof(double) * m, C, sizeof(double) * ldc, sizeof(double) * m, n);
  if (!rc) {
    hssk_gemm_desc d{dA, dB, dC, m, n, k, ar > 0 ? ar : 1, br > 0 ? br : 1, m, TA ? 1 : 0, TB ? 1 : 0, alpha, beta};
    rc |= hssk_gemm_vbatched(g_ctx, &d, 1);
  }
  if (!rc) rc |= hssk_memcpy2d_d2h(g_ctx, C, sizeof(double) * ldc, dC, sizeof(double) * m, sizeof(double) * m, n);
  hssk_free(dA);
  hssk_free(dB);
  hssk_free(dC);
  return rc;
}
