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
#include "hssk_internal.h"

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
  const int kmax = d < m ? d : m;
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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      double ov = hssk_shfl_xor(bv, o);
      int oi = hssk_shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
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

}  // namespace

extern "C" int hssk_id_vbatched(hssk_ctx* ctx, const hssk_id_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  auto* dd = (const hssk_id_desc*)ctx->stage(descs, sizeof(*descs) * count);
  HSSK_LAUNCH(id_kernel, dim3((unsigned)count), dim3(ID_THREADS), 0, ctx->stream, dd);
  hssk_rt::check_launch();
  HSSK_API_END
}
