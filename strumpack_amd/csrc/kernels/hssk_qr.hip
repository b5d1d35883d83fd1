// Batched Householder QR with optional explicit Q, one workgroup per HSS node.
//
// Serves two reference routines: DenseMatrix::orthogonalize (geqrf + orgqr, returns max/min |R_ii|;
// dense/DenseMatrix.cpp:721-744 -- the rank-adequacy test of the stable compression,
// HSS/HSSMatrix.compress_stable.hpp:390-442) and DenseMatrix::LQ (gelqf + orglq with the full
// m x m Q; dense/DenseMatrix.cpp:693-719 -- the ULV factorization, HSS/HSSMatrix.factor.hpp:122):
// the LQ of W0 is computed as the QR of W0^T, which the engine forms directly.
//
// Same wave64 mapping as the ID kernel: the reflector is built by one wave with shuffle reductions,
// the trailing update gives one column to each wave, lanes stride the (contiguous) column.
// Q is accumulated backwards (dorg2r order) in a separate rows x nq buffer.
// Bound: L2 latency/bandwidth (Level-2 BLAS on an L2-resident panel).
#include "hssk_device.h"
#include "hssk_internal.h"

namespace {

constexpr int QR_THREADS = 512;
constexpr int QR_WAVES = QR_THREADS / 64;

__global__ __launch_bounds__(QR_THREADS) void qr_kernel(const hssk_qr_desc* __restrict__ descs) {
  HSSK_SHARED double s_tau;
  const hssk_qr_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rows = p.rows, cols = p.cols, ld = p.lda;
  double* __restrict__ A = p.A;
  double* taus = p.work;  // kmax entries
  const int kmax = rows < cols ? rows : cols;
  double rmax = 0., rmin = 0.;

  for (int k = 0; k < kmax; k++) {
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
  if (p.rdiag && tid == 0) { p.rdiag[0] = rmax; p.rdiag[1] = rmin; }

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

}  // namespace

extern "C" int hssk_qr_vbatched(hssk_ctx* ctx, const hssk_qr_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  auto* dd = (const hssk_qr_desc*)ctx->stage(descs, sizeof(*descs) * count);
  HSSK_LAUNCH(qr_kernel, dim3((unsigned)count), dim3(QR_THREADS), 0, ctx->stream, dd);
  hssk_rt::check_launch();
  HSSK_API_END
}
