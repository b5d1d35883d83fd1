// Row ID of a TALL sample panel W (d x m, d >> m) from its Gram matrix G = W^T W (m x m): the column-pivoted QR of W only
// depends on G -- pivot = the column of largest remaining norm^2 = largest remaining diagonal of the Schur complement of G,
// |R_kk| = its square root, row k of R = the pivot's row of that Schur complement over R_kk -- i.e. a diagonally pivoted
// Cholesky factorization of G, stopped by the reference's rule (dense/lapack/dgeqp3tol.f:225-232: the first k with
// |R_kk| / |R_00| <= rtol or |R_kk| <= atol).  Only `rank` steps are taken, each O(k m); the d-long part of the work is the
// product W^T W, which runs on the matrix cores (hssk_gemm_vbatched, K split into chunks summed in a fixed order by
// hssk_sum_partials) instead of a Householder sweep over d x m at one reflector per step (the TSQR of hss_compress.cpp).
//
// Accuracy: G carries the SQUARES of the singular values, so directions below sqrt(eps) ~ 1e-8 of the largest are noise.
// The caller uses this form only for tolerances >= 1e-6 (kernel matrices are compressed to 1e-2 .. 1e-4) and keeps the
// Householder path otherwise (DeviceHSS::id_panels).
//
// One workgroup per panel, thread j = column j (m <= 256).  Left-looking: the rows of R found so far live in the LDS
// (cap rows: what 150 KB hold; a panel whose rank reaches the cap reports rank -1 and the caller takes the QR path), the row of
// the Schur complement of the new pivot is G(:, p) minus its projection on them.
#include "hssk_device.h"
#include "hssk_internal.h"

#include <algorithm>
#include <vector>

namespace {

constexpr int PC_T = 256;
constexpr size_t PC_LDS = 150 * 1024;

__global__ __launch_bounds__(PC_T) void pchol_id_kernel(const hssk_pchol_desc* __restrict__ descs, int cap_rows_max) {
  HSSK_DYN_SHARED(double, Lr);   // [cap][mp] rows of R (original column order)
  HSSK_SHARED double s_v[4];
  HSSK_SHARED int s_i[4];
  HSSK_SHARED int s_piv[PC_T];
  HSSK_SHARED int s_pos[PC_T];
  const hssk_pchol_desc p = descs[blockIdx.x];
  const int j = threadIdx.x, lane = j & 63, wave = j >> 6;
  const int m = p.m, mp = (m + 7) & ~7;
  const int cap = min(min(cap_rows_max, (int)(PC_LDS / (sizeof(double) * (size_t)mp))), p.ldr);
  const bool col = j < m;
  double dj = col ? hssk_gload(p.G, (size_t)j + (size_t)j * p.ldg) : -1.;   // remaining squared norm of column j
  bool alive = col;
  double r00 = 0.;
  int rank = -2;
  const int kend = m;
  for (int k = 0; k <= kend; k++) {
    if (k == kend) { rank = kend; break; }
    // ---- the pivot: largest remaining squared norm, the first among equals
    double v = alive ? dj : -1.;
    int idx = j;
    hssk_wave_argmax(v, idx);
    if (lane == 0) { s_v[wave] = v; s_i[wave] = idx; }
    __syncthreads();
    double bv = s_v[0];
    int pc = s_i[0];
#pragma unroll
    for (int w = 1; w < 4; w++)
      if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < pc)) { bv = s_v[w]; pc = s_i[w]; }
    const double rkk = sqrt(bv > 0. ? bv : 0.);
    if (k == 0) r00 = rkk;
    // dgeqp3tol.f:225-232 (0 / 0 is NaN -> false, then the absolute test decides)
    if ((r00 != 0. && rkk / r00 <= p.rtol) || rkk <= p.atol) { rank = k; break; }
    if (k == cap) { rank = -1; break; }   // (more rows than the LDS holds: the caller's other path)
    // ---- row k of R: (G(p, :) - R(0:k, p)^T R(0:k, :)) / R_kk over the columns still in play
    double s = 0.;
    if (alive) {
      s = hssk_gload(p.G, (size_t)j + (size_t)pc * p.ldg);   // (G is symmetric: column p read along the threads)
      for (int i = 0; i < k; i++) s -= Lr[(size_t)i * mp + pc] * Lr[(size_t)i * mp + j];
    }
    const double rkj = j == pc ? rkk : (alive ? s / rkk : 0.);
    if (col) Lr[(size_t)k * mp + j] = rkj;
    if (alive) {
      if (j == pc) { alive = false; s_piv[k] = pc; }
      else { dj -= rkj * rkj; dj = dj > 0. ? dj : 0.; }
    }
    __syncthreads();
  }
  __syncthreads();
  if (rank < 0) {
    if (j == 0) *p.rank = -1;
    return;
  }
  const int rk = min(rank, p.max_rank);
  // ---- pivoted column positions: skeleton columns first (pivot order), then the rest in index order (as hssk_id_vbatched)
  if (col) s_pos[j] = -1;
  __syncthreads();
  if (j < rk) s_pos[s_piv[j]] = j;
  __syncthreads();
  {
    // a column that is not a skeleton column goes to rk + (number of such columns before it)
    const int mine = col && s_pos[j] < 0;
    const unsigned long long mk = hssk_ballot(mine);
    if (lane == 0) s_i[wave] = __builtin_popcountll(mk);
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; w++) before += s_i[w];
    const int pos = mine ? rk + before + __builtin_popcountll(mk & ((1ULL << lane) - 1ULL)) : (col ? s_pos[j] : -1);
    __syncthreads();
    if (col) { s_pos[j] = pos; p.perm[pos] = j; }
  }
  if (j == 0) *p.rank = rk;
  __syncthreads();
  // ---- [R11 R12]: row i of R in pivoted column order, zeros below the diagonal of R11
  if (col) {
    const int c = s_pos[j];
    for (int i = 0; i < rk; i++) hssk_gstore(p.R, (size_t)i + (size_t)c * p.ldr, (c >= i) ? Lr[(size_t)i * mp + j] : 0.);
  }
}

// out[e] = sum over the partials, in order
__global__ __launch_bounds__(256) void sum_partials_kernel(const hssk_sum_desc* __restrict__ descs, int per) {
  const hssk_sum_desc p = descs[blockIdx.x / per];
  const int b = blockIdx.x % per;
  for (long long e = (long long)b * 256 + threadIdx.x; e < p.n; e += (long long)per * 256) {
    double s = 0.;
    for (int q = 0; q < p.count; q++) s += hssk_gload(p.P, (size_t)q * p.stride + e);
    hssk_gstore(p.out, (size_t)e, s);
  }
}

}  // namespace

extern "C" int hssk_pchol_id_max_m(void) { return PC_T; }
extern "C" int hssk_pchol_id_rank_cap(int m) {
  const int mp = (std::max(m, 1) + 7) & ~7;
  const size_t lds = std::min(PC_LDS, hssk_rt::max_lds_per_workgroup() > 8192 ? hssk_rt::max_lds_per_workgroup() - 8192 : 0);
  return (int)std::min<size_t>(lds / (sizeof(double) * (size_t)mp), (size_t)m);
}

extern "C" int hssk_pchol_id_vbatched(hssk_ctx* ctx, const hssk_pchol_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  int mmax = 0;
  for (int i = 0; i < count; i++) {
    if (descs[i].m <= 0 || descs[i].m > PC_T) HSSK_UNSUPPORTED("panels of 1 .. 256 columns");
    if (descs[i].ldr < 1) throw std::invalid_argument("hssk_pchol_id_vbatched: ldr < 1");
    mmax = std::max(mmax, descs[i].m);
  }
  const int mp = (mmax + 7) & ~7;
  int cap = 0;
  for (int i = 0; i < count; i++) cap = std::max(cap, std::min(descs[i].ldr, descs[i].m));
  const size_t lds_max = std::min(PC_LDS, hssk_rt::max_lds_per_workgroup() > 8192 ? hssk_rt::max_lds_per_workgroup() - 8192 : 0);
  cap = (int)std::min<size_t>((size_t)cap, lds_max / (sizeof(double) * (size_t)mp));
  if (cap < 1) HSSK_UNSUPPORTED("no LDS for a row of R");
  const size_t shm = sizeof(double) * (size_t)cap * mp;
  auto* dd = (const hssk_pchol_desc*)ctx->stage(descs, sizeof(*descs) * count);
  hssk_rt::allow_dynamic_lds(pchol_id_kernel, shm);
  HSSK_LAUNCH(pchol_id_kernel, dim3((unsigned)count), dim3(PC_T), shm, ctx->stream, dd, cap);
  hssk_rt::check_launch();
  HSSK_API_END
}

extern "C" int hssk_sum_partials(hssk_ctx* ctx, const hssk_sum_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  long long nmax = 0;
  for (int i = 0; i < count; i++) nmax = std::max(nmax, descs[i].n);
  const int per = (int)std::max<long long>(1, std::min<long long>(64, (nmax + 1023) / 1024));
  auto* dd = (const hssk_sum_desc*)ctx->stage(descs, sizeof(*descs) * count);
  HSSK_LAUNCH(sum_partials_kernel, dim3((unsigned)(count * per)), dim3(256), 0, ctx->stream, dd, per);
  hssk_rt::check_launch();
  HSSK_API_END
}
