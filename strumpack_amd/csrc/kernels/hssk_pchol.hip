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
#include <cstdlib>
#include <vector>

namespace {

constexpr int PC_T = 256;
constexpr size_t PC_LDS = 150 * 1024;

__global__ __launch_bounds__(PC_T) void pchol_id_kernel(const hssk_pchol_desc* __restrict__ descs, int lds_doubles) {
  HSSK_DYN_SHARED(double, Lr);   // [cap][mp] rows of R (original column order)
  HSSK_SHARED double s_v[4];
  HSSK_SHARED int s_i[4];
  HSSK_SHARED int s_piv[PC_T];
  HSSK_SHARED int s_pos[PC_T];
  const hssk_pchol_desc p = descs[blockIdx.x];
  const int j = threadIdx.x, lane = j & 63, wave = j >> 6;
  const int m = p.m, mp = (m + 7) & ~7;
  const int cap = min(lds_doubles / mp, p.ldr);   // rows of R this panel may take
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
  // LDS of the launch: what the panel with the largest (rows it may take) x (padded columns) needs, within the device's
  const size_t lds_max = std::min(PC_LDS, hssk_rt::max_lds_per_workgroup() > 8192 ? hssk_rt::max_lds_per_workgroup() - 8192 : 0);
  size_t need = 0;
  for (int i = 0; i < count; i++) {
    const size_t mp = (size_t)((descs[i].m + 7) & ~7);
    need = std::max(need, sizeof(double) * mp * (size_t)std::min(descs[i].ldr, descs[i].m));
  }
  const size_t shm = std::min(need, lds_max);
  if (shm < sizeof(double) * (size_t)((mmax + 7) & ~7)) HSSK_UNSUPPORTED("no LDS for a row of R");
  auto* dd = (const hssk_pchol_desc*)ctx->stage(descs, sizeof(*descs) * count);
  hssk_rt::allow_dynamic_lds(pchol_id_kernel, shm);
  HSSK_LAUNCH(pchol_id_kernel, dim3((unsigned)count), dim3(PC_T), shm, ctx->stream, dd, (int)(shm / sizeof(double)));
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

// ---------------------------------------------------------------------------------------------
// G = W^T W for tall panels W (rows x m, column-major: K runs along the contiguous direction), FP64 matrix cores.
// A workgroup (4 waves) owns a 128 x 128 block (I, J >= I) of G for one panel, a wave a 64 x 64 quadrant = 4 x 4 MFMA tiles;
// quadrants and tiles outside the panel's columns or below the diagonal are skipped, off-diagonal tiles are written on both
// sides of it.  Both operands are column blocks of W staged through the LDS as [column][k] (17 doubles per column: a lane
// group of 16 columns falls on 16 banks), the next stage's loads in flight under the products of the current one.
// Bound: MFMA (a 128 x 128 block takes 16 KB of operands per 16 k-rows = 64 MFMAs per wave).
namespace {
constexpr int GR_B = 128, GR_K = 16, GR_KP = GR_K + 1;
struct GramTile { int prob, bi, bj; };

__global__ __launch_bounds__(256) void gram_kernel(const hssk_gram_desc* __restrict__ descs, const GramTile* __restrict__ tiles) {
  HSSK_DYN_SHARED(double, gr_lds);   // As[2][128 x 17] | Bs[2][128 x 17]
  const GramTile t = tiles[blockIdx.x];
  const hssk_gram_desc p = descs[t.prob];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int i0 = t.bi * GR_B, j0 = t.bj * GR_B, wi = i0 + (wave & 1) * 64, wj = j0 + (wave >> 1) * 64;
  const bool diag = t.bi == t.bj;
  // 16 x 16 tiles of this wave that lie inside the panel and on or above the diagonal
  bool on[4][4];
  bool any = false;
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) {
      on[a][b] = wi + 16 * a < p.m && wj + 16 * b < p.m && wi + 16 * a <= wj + 16 * b;
      any = any || on[a][b];
    }
  hssk_d4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = hssk_d4{0., 0., 0., 0.};
  // stage loads: thread -> k pair 2 (tid & 7), columns (tid >> 3) + 32 r
  const int ks = 2 * (tid & 7), cb = tid >> 3;
  hssk_d2 ra[4], rb[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int c = cb + 32 * r;
      const int ga = min(i0 + c, p.m - 1), gb = min(j0 + c, p.m - 1), gk = min(k0 + ks, max(p.rows - 2, 0));
      hssk_d2 va = hssk_gload2u(p.W, (size_t)gk + (size_t)ga * p.ldw);
      hssk_d2 vb = diag ? va : hssk_gload2u(p.W, (size_t)gk + (size_t)gb * p.ldw);
      // (rows beyond the panel's: zero; the clamped pair may straddle the end -- a panel has at least two rows)
      const bool k0ok = k0 + ks < p.rows, k1ok = k0 + ks + 1 < p.rows, shifted = k0 + ks > gk;
      const double a0 = shifted ? va[1] : va[0], b0 = shifted ? vb[1] : vb[0];
      ra[r] = hssk_d2{(k0ok && i0 + c < p.m) ? a0 : 0., (k1ok && i0 + c < p.m) ? va[1] : 0.};
      rb[r] = hssk_d2{(k0ok && j0 + c < p.m) ? b0 : 0., (k1ok && j0 + c < p.m) ? vb[1] : 0.};
    }
  };
  fetch(0);
  int buf = 0;
  for (int k0 = 0; k0 < p.rows; k0 += GR_K, buf ^= 1) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int c = cb + 32 * r;
      double* Aw = gr_lds + buf * GR_B * GR_KP;
      double* Bw = gr_lds + (2 + buf) * GR_B * GR_KP;
      Aw[c * GR_KP + ks] = ra[r][0]; Aw[c * GR_KP + ks + 1] = ra[r][1];
      if (!diag) { Bw[c * GR_KP + ks] = rb[r][0]; Bw[c * GR_KP + ks + 1] = rb[r][1]; }
    }
    __syncthreads();   // (one barrier per stage: the buffer written here was last read two stages ago)
    if (k0 + GR_K < p.rows) fetch(k0 + GR_K);
    if (any) {
      const double* Ab = gr_lds + buf * GR_B * GR_KP + ((wave & 1) * 64) * GR_KP;
      const double* Bb = gr_lds + ((diag ? 0 : 2) + buf) * GR_B * GR_KP + ((wave >> 1) * 64) * GR_KP;
#pragma unroll
      for (int kk = 0; kk < GR_K; kk += 4) {
        double av[4], bv[4];
#pragma unroll
        for (int a = 0; a < 4; a++) av[a] = Ab[(16 * a + l15) * GR_KP + kk + l4];
#pragma unroll
        for (int b = 0; b < 4; b++) bv[b] = Bb[(16 * b + l15) * GR_KP + kk + l4];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int b = 0; b < 4; b++)
            if (on[a][b]) acc[a][b] = hssk_mfma_f64_16x16x4(av[a], bv[b], acc[a][b]);
      }
    }
  }
  // ---- G(i, j) and G(j, i): lane l holds rows (l >> 4) + 4 r, column l & 15 of its tiles
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) {
      if (!on[a][b]) continue;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int gi = wi + 16 * a + l4 + 4 * r, gj = wj + 16 * b + l15;
        if (gi < p.m && gj < p.m) {
          hssk_gstore(p.G, (size_t)gi + (size_t)gj * p.ldg, acc[a][b][r]);
          if (wi + 16 * a != wj + 16 * b) hssk_gstore(p.G, (size_t)gj + (size_t)gi * p.ldg, acc[a][b][r]);
        }
      }
    }
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// The same product with ONE workgroup (8 waves) per panel and row chunk: the 16 x 16 tiles on and above the diagonal of the
// whole m x m matrix are dealt to the waves in row-major runs (91 tiles for the 195-column panels of a leaf: 11 or 12 per wave --
// the 128 x 128 blocks above left their waves 16, 16, 10 and 0 tiles of the same panel), every column of the chunk is staged once.
// m <= 256 (NT tiles per wave: 5 up to 96 columns, 12 up to 208, 17 up to 256).
namespace {
constexpr int G2_K = 16, G2_KP = G2_K + 1, G2_T = 512;

template <int NT>
__global__ __launch_bounds__(G2_T) void gram_panel_kernel(const hssk_gram_desc* __restrict__ descs) {
  HSSK_DYN_SHARED(double, g2_lds);   // two stages of [mp columns][17]
  const hssk_gram_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = hssk_uniform(tid >> 6), l15 = lane & 15, l4 = lane >> 4;
  const int m = p.m, nb = (m + 15) / 16, ntile = nb * (nb + 1) / 2, mp = nb * 16;
  // this wave's tiles: t0 .. t0 + cnt of the row-major list of (a, b >= a)
  const int per = (ntile + 7) / 8, t0 = wave * per, cnt = max(0, min(per, ntile - t0));
  int ta[NT], tb[NT];
  {
    int a = 0, rem = t0;   // row of tile t0: rows have nb, nb - 1, ... tiles
    while (a < nb && rem >= nb - a) { rem -= nb - a; a++; }
    int b = a + rem;
#pragma unroll
    for (int i = 0; i < NT; i++) {
      ta[i] = min(a, nb - 1); tb[i] = min(b, nb - 1);
      b++;
      if (b >= nb) { a++; b = a; }
    }
  }
  hssk_d4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; i++) acc[i] = hssk_d4{0., 0., 0., 0.};
  // stage loads: thread -> k pair 2 (tid & 7), columns (tid >> 3) + 64 r
  const int ks = 2 * (tid & 7), cb = tid >> 3;
  hssk_d2 rv[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int c = cb + 64 * r;
      if (64 * r < mp) {
        const int gc = min(c, m - 1), gk = min(k0 + ks, max(p.rows - 2, 0));
        const hssk_d2 v = hssk_gload2u(p.W, (size_t)gk + (size_t)gc * p.ldw);
        const bool k0ok = k0 + ks < p.rows, k1ok = k0 + ks + 1 < p.rows, shifted = k0 + ks > gk;
        rv[r] = hssk_d2{(k0ok && c < m) ? (shifted ? v[1] : v[0]) : 0., (k1ok && c < m) ? v[1] : 0.};
      }
    }
  };
  fetch(0);
  int buf = 0;
  for (int k0 = 0; k0 < p.rows; k0 += G2_K, buf ^= 1) {
    double* S = g2_lds + (size_t)buf * mp * G2_KP;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int c = cb + 64 * r;
      if (c < mp) { S[c * G2_KP + ks] = rv[r][0]; S[c * G2_KP + ks + 1] = rv[r][1]; }
    }
    __syncthreads();   // (one barrier per stage: the buffer written here was last read two stages ago)
    if (k0 + G2_K < p.rows) fetch(k0 + G2_K);
#pragma unroll
    for (int kk = 0; kk < G2_K; kk += 4) {
      double av = 0.;
      int arow = -1;
#pragma unroll
      for (int i = 0; i < NT; i++)
        if (i < cnt) {
          if (ta[i] != arow) { arow = ta[i]; av = S[(16 * arow + l15) * G2_KP + kk + l4]; }
          const double bv = S[(16 * tb[i] + l15) * G2_KP + kk + l4];
          acc[i] = hssk_mfma_f64_16x16x4(av, bv, acc[i]);
        }
    }
  }
  // ---- G(i, j) and G(j, i): lane l holds rows (l >> 4) + 4 r, column l & 15 of its tiles
#pragma unroll
  for (int i = 0; i < NT; i++)
    if (i < cnt) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int gi = 16 * ta[i] + l4 + 4 * r, gj = 16 * tb[i] + l15;
        if (gi < m && gj < m) {
          hssk_gstore(p.G, (size_t)gi + (size_t)gj * p.ldg, acc[i][r]);
          if (ta[i] != tb[i]) hssk_gstore(p.G, (size_t)gj + (size_t)gi * p.ldg, acc[i][r]);
        }
      }
    }
}

template <int NT> void gram_panel_launch(hssk_ctx* ctx, const hssk_gram_desc* descs, int count, int mmax) {
  const int mp = ((mmax + 15) / 16) * 16;
  const size_t shm = sizeof(double) * 2 * (size_t)mp * G2_KP;
  auto* dd = (const hssk_gram_desc*)ctx->stage(descs, sizeof(*descs) * count);
  hssk_rt::allow_dynamic_lds(gram_panel_kernel<NT>, shm);
  HSSK_LAUNCH(gram_panel_kernel<NT>, dim3((unsigned)count), dim3(G2_T), shm, ctx->stream, dd);
}
}  // namespace

extern "C" int hssk_gram_vbatched(hssk_ctx* ctx, const hssk_gram_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  // panels of up to 256 columns: a workgroup per panel (HSSK_GRAM_BLOCKS=1: the 128 x 128 blocks for everything)
  static const bool blocks_only = [] { const char* e = std::getenv("HSSK_GRAM_BLOCKS"); return e && e[0] == '1'; }();
  int mmax = 0;
  bool ok = !blocks_only;
  for (int q = 0; q < count; q++) {
    mmax = std::max(mmax, descs[q].m);
    ok = ok && descs[q].m > 0 && descs[q].m <= 256 && descs[q].rows >= 2;
  }
  if (ok && sizeof(double) * 2 * (size_t)(((mmax + 15) / 16) * 16) * G2_KP <= hssk_rt::max_lds_per_workgroup()) {
    const int nb = (mmax + 15) / 16, per = (nb * (nb + 1) / 2 + 7) / 8;
    if (per <= 5) gram_panel_launch<5>(ctx, descs, count, mmax);
    else if (per <= 12) gram_panel_launch<12>(ctx, descs, count, mmax);
    else gram_panel_launch<17>(ctx, descs, count, mmax);
    hssk_rt::check_launch();
    return 0;
  }
  std::vector<GramTile> tiles;
  for (int q = 0; q < count; q++) {
    if (descs[q].m <= 0) continue;
    if (descs[q].rows < 2) throw std::invalid_argument("hssk_gram_vbatched: a panel has at least two rows");
    const int nb = (descs[q].m + GR_B - 1) / GR_B;
    for (int bi = 0; bi < nb; bi++)
      for (int bj = bi; bj < nb; bj++) tiles.push_back(GramTile{q, bi, bj});
  }
  if (tiles.empty()) return 0;
  auto* dd = (const hssk_gram_desc*)ctx->stage(descs, sizeof(*descs) * count);
  auto* dt = (const GramTile*)ctx->stage(tiles.data(), sizeof(GramTile) * tiles.size());
  const size_t shm = sizeof(double) * 4 * GR_B * GR_KP;
  hssk_rt::allow_dynamic_lds(gram_kernel, shm);
  HSSK_LAUNCH(gram_kernel, dim3((unsigned)tiles.size()), dim3(256), shm, ctx->stream, dd, dt);
  hssk_rt::check_launch();
  HSSK_API_END
}

// ---------------------------------------------------------------------------------------------
// The same product for panels that are blocks of a KERNEL matrix, W(k, c) = K(x_row(k), x_col(c)) (Gauss / Laplace): the entries
// are evaluated from the points' coordinates while they are staged -- the panel (8 GB per compression at N = 1e5) is neither
// written by hssk_kernel_eval_vbatched nor read back here.  Columns' points sit in the LDS for the whole chunk, a stage's 16
// row points are fetched one stage ahead; a thread evaluates up to eight entries per stage under the products of the previous one.
// Point dimension <= 16.
namespace {
constexpr int GG_DP = 17;   // doubles per point in the LDS (16 coordinates + 1: lanes of consecutive points on different banks)

template <int NT>
__global__ __launch_bounds__(G2_T) void gram_gen_panel_kernel(hssk_kernel_spec ks, const hssk_gramgen_desc* __restrict__ descs) {
  HSSK_DYN_SHARED(double, gg_lds);   // two stages of [mp columns][17] | column points [mp][17] | row points of a stage [16][17]
  const hssk_gramgen_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = hssk_uniform(tid >> 6), l15 = lane & 15, l4 = lane >> 4;
  const int m = p.m, nb = (m + 15) / 16, ntile = nb * (nb + 1) / 2, mp = nb * 16, d = ks.d;
  double* Xc = gg_lds + 2 * (size_t)mp * G2_KP;
  double* Xr = Xc + (size_t)mp * GG_DP;
  const int per = (ntile + 7) / 8, t0 = wave * per, cnt = max(0, min(per, ntile - t0));
  int ta[NT], tb[NT];
  {
    int a = 0, rem = t0;
    while (a < nb && rem >= nb - a) { rem -= nb - a; a++; }
    int b = a + rem;
#pragma unroll
    for (int i = 0; i < NT; i++) {
      ta[i] = min(a, nb - 1); tb[i] = min(b, nb - 1);
      b++;
      if (b >= nb) { a++; b = a; }
    }
  }
  hssk_d4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; i++) acc[i] = hssk_d4{0., 0., 0., 0.};
  // the columns' points (W's column c = point ci[c] or c0 + c)
  for (int e = tid; e < mp * d; e += G2_T) {
    const int c = e / d, j = e % d;
    const int g = c < m ? (p.ci ? p.ci[c] : p.c0 + c) : 0;
    Xc[c * GG_DP + j] = c < m ? hssk_gload(ks.X, (size_t)g * d + j) : 0.;
  }
  // a stage's row points: thread t < 16 d fetches coordinate t % d of row k0 + t / d (one stage ahead, through a register)
  auto fetch_rows = [&](int k0) {
    double x = 0.;
    if (tid < 16 * d) {
      const int k = k0 + tid / d;
      const int g = k < p.rows ? (p.ri ? p.ri[k] : p.r0 + k) : 0;
      x = hssk_gload(ks.X, (size_t)g * d + tid % d);
    }
    return x;
  };
  const double scale = ks.type == 0 ? -1. / (2. * ks.h * ks.h) : -1. / ks.h;
  // entries of stage k0 this thread evaluates: k = tid & 15, columns (tid >> 4) + 32 r
  const int kt = tid & 15, ct = tid >> 4;
  double gv[8];
  auto generate = [&](int k0) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int c = ct + 32 * r;
      gv[r] = 0.;
      if (32 * r < mp) {
        double a = 0.;
        for (int j = 0; j < d; j++) {
          const double df = Xr[kt * GG_DP + j] - Xc[min(c, mp - 1) * GG_DP + j];
          a += ks.type == 0 ? df * df : fabs(df);
        }
        gv[r] = (k0 + kt < p.rows && c < m) ? exp(a * scale) : 0.;
      }
    }
  };
  double xr = fetch_rows(0);
  if (tid < 16 * d) Xr[(tid / d) * GG_DP + tid % d] = xr;
  __syncthreads();
  generate(0);
  xr = fetch_rows(G2_K);
  __syncthreads();   // (the first stage's row points are replaced at the top of the loop: every wave has evaluated its entries)
  int buf = 0;
  for (int k0 = 0; k0 < p.rows; k0 += G2_K, buf ^= 1) {
    double* S = gg_lds + (size_t)buf * mp * G2_KP;
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int c = ct + 32 * r;
      if (c < mp) S[c * G2_KP + kt] = gv[r];
    }
    if (tid < 16 * d) Xr[(tid / d) * GG_DP + tid % d] = xr;   // (the row points of stage k0 were last read before the previous barrier)
    __syncthreads();
    if (k0 + G2_K < p.rows) {
      generate(k0 + G2_K);
      xr = fetch_rows(k0 + 2 * G2_K);
    }
#pragma unroll
    for (int kk = 0; kk < G2_K; kk += 4) {
      double av = 0.;
      int arow = -1;
#pragma unroll
      for (int i = 0; i < NT; i++)
        if (i < cnt) {
          if (ta[i] != arow) { arow = ta[i]; av = S[(16 * arow + l15) * G2_KP + kk + l4]; }
          const double bv = S[(16 * tb[i] + l15) * G2_KP + kk + l4];
          acc[i] = hssk_mfma_f64_16x16x4(av, bv, acc[i]);
        }
    }
    __syncthreads();   // (the row points are replaced at the top of the next stage: every wave has evaluated its entries)
  }
#pragma unroll
  for (int i = 0; i < NT; i++)
    if (i < cnt) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int gi = 16 * ta[i] + l4 + 4 * r, gj = 16 * tb[i] + l15;
        if (gi < m && gj < m) {
          hssk_gstore(p.G, (size_t)gi + (size_t)gj * p.ldg, acc[i][r]);
          if (ta[i] != tb[i]) hssk_gstore(p.G, (size_t)gj + (size_t)gi * p.ldg, acc[i][r]);
        }
      }
    }
}

template <int NT> void gram_gen_launch(hssk_ctx* ctx, const hssk_kernel_spec& ks, const hssk_gramgen_desc* descs, int count, int mmax) {
  const int mp = ((mmax + 15) / 16) * 16;
  const size_t shm = sizeof(double) * (2 * (size_t)mp * G2_KP + (size_t)mp * GG_DP + 16 * GG_DP);
  auto* dd = (const hssk_gramgen_desc*)ctx->stage(descs, sizeof(*descs) * count);
  hssk_rt::allow_dynamic_lds(gram_gen_panel_kernel<NT>, shm);
  HSSK_LAUNCH(gram_gen_panel_kernel<NT>, dim3((unsigned)count), dim3(G2_T), shm, ctx->stream, ks, dd);
}
}  // namespace

extern "C" int hssk_gram_gen_supported(const hssk_kernel_spec* spec, int mmax) {
  return spec && (spec->type == 0 || spec->type == 1) && spec->d >= 1 && spec->d <= 16 && mmax >= 1 && mmax <= 256 &&
         sizeof(double) * (2 * (size_t)(((mmax + 15) / 16) * 16) * G2_KP + (size_t)(((mmax + 15) / 16) * 16) * GG_DP + 16 * GG_DP) <= hssk_rt::max_lds_per_workgroup();
}

extern "C" int hssk_gram_gen_vbatched(hssk_ctx* ctx, const hssk_kernel_spec* spec, const hssk_gramgen_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  int mmax = 0;
  for (int q = 0; q < count; q++) {
    if (descs[q].m <= 0 || descs[q].rows <= 0) throw std::invalid_argument("hssk_gram_gen_vbatched: empty panel");
    mmax = std::max(mmax, descs[q].m);
  }
  if (!hssk_gram_gen_supported(spec, mmax)) HSSK_UNSUPPORTED("Gauss / Laplace kernels over points of at most 16 coordinates, panels of at most 256 columns");
  const int nb = (mmax + 15) / 16, per = (nb * (nb + 1) / 2 + 7) / 8;
  if (per <= 5) gram_gen_launch<5>(ctx, *spec, descs, count, mmax);
  else if (per <= 12) gram_gen_launch<12>(ctx, *spec, descs, count, mmax);
  else gram_gen_launch<17>(ctx, *spec, descs, count, mmax);
  hssk_rt::check_launch();
  HSSK_API_END
}
