// Variable-size batched FP64 GEMM on the CDNA4 matrix cores (v_mfma_f64_16x16x4_f64).
//
// One launch == every small gemm of one HSS tree level:  C_i = alpha op(A_i) op(B_i) + beta C_i
// (reference call sites: leaf/inner sample updates HSS/HSSMatrix.compress.hpp:541-620, basis
// reduction HSS/HSSBasisID.hpp:189-203, ULV updates HSS/HSSMatrix.factor.hpp:68-137, solve/apply
// sweeps HSS/HSSMatrix.solve.hpp:88-181, HSS/HSSMatrix.apply.hpp:55-124).
//
// Mapping: the host flattens the batch into a work list of 64x64 output tiles (problem, tile row,
// tile col); workgroup b (256 threads = 4 wave64) owns tile b.  Each wave owns a 32x32 quadrant =
// 2x2 MFMA tiles, K advances 16 per LDS stage (4 MFMA k-steps).  Operands are staged through LDS as
// As[k][i] / Bs[k][j] with coalesced, zero-padded global reads for either transpose flag, so ragged
// sizes (HSS ranks are arbitrary) cost masking only at the edges.  The MFMA is issued with the B
// fragment as its first operand so that lane l ends up holding C[i = l&15][j = (l>>4)+4r]: a
// 16-lane group then writes 16 consecutive rows of column-major C (128 contiguous bytes).
// Bound: MFMA for the leaf level (k = leaf size), launch/latency for the upper levels.
#include "hssk_device.h"
#include "hssk_internal.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {

constexpr int TM = 64, TN = 64, TK = 16;
constexpr int LDS_LD = 80;  // 64 + 16: rows k and k+1 fall into different halves of the 64 banks

struct Tile {
  int prob, tm, tn;
};

__global__ __launch_bounds__(256) void gemm_vbatched_kernel(const hssk_gemm_desc* __restrict__ descs,
                                                            const Tile* __restrict__ tiles) {
  HSSK_SHARED double As[TK * LDS_LD];
  HSSK_SHARED double Bs[TK * LDS_LD];
  const Tile t = tiles[blockIdx.x];
  const hssk_gemm_desc p = descs[t.prob];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = t.tm * TM, j0 = t.tn * TN;
  const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
  const int l15 = lane & 15, l4 = lane >> 4;

  hssk_d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) acc[a][b] = hssk_d4{0., 0., 0., 0.};

  // The next K stage's operands are loaded into registers while the matrix cores work on the current one (the blocks of a
  // level stream from HBM: without the prefetch every stage exposes a memory round trip -- the 256 x 256 x 64 products of the
  // leaf level of a 64-right-hand-side mat-vec ran at 24 TFLOP/s).
  double ra[4], rb[4];
  auto fetch = [&](int k0) {
    // A tile: As[kk][i] = op(A)(i0+i, k0+kk);  B tile: Bs[kk][j] = op(B)(k0+kk, j0+j)   (clamped address + select: no branch
    // around the loads)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = p.transA ? (tid >> 4) + 16 * r : (tid & 63), kk = p.transA ? (tid & 15) : (tid >> 6) + 4 * r;
      const int gi = i0 + i, gk = k0 + kk;
      const bool ok = gi < p.m && gk < p.k;
      const double v = p.transA ? hssk_gload(p.A, min(gk, p.k - 1) + (size_t)min(gi, p.m - 1) * p.lda)
                                : hssk_gload(p.A, min(gi, p.m - 1) + (size_t)min(gk, p.k - 1) * p.lda);
      ra[r] = ok ? v : 0.;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int j = p.transB ? (tid & 63) : (tid >> 4) + 16 * r, kk = p.transB ? (tid >> 6) + 4 * r : (tid & 15);
      const int gj = j0 + j, gk = k0 + kk;
      const bool ok = gj < p.n && gk < p.k;
      const double v = p.transB ? hssk_gload(p.B, min(gj, p.n - 1) + (size_t)min(gk, p.k - 1) * p.ldb)
                                : hssk_gload(p.B, min(gk, p.k - 1) + (size_t)min(gj, p.n - 1) * p.ldb);
      rb[r] = ok ? v : 0.;
    }
  };
  if (p.k > 0) fetch(0);
  for (int k0 = 0; k0 < p.k; k0 += TK) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = p.transA ? (tid >> 4) + 16 * r : (tid & 63), kk = p.transA ? (tid & 15) : (tid >> 6) + 4 * r;
      As[kk * LDS_LD + i] = ra[r];
      const int j = p.transB ? (tid & 63) : (tid >> 4) + 16 * r, kb = p.transB ? (tid >> 6) + 4 * r : (tid & 15);
      Bs[kb * LDS_LD + j] = rb[r];
    }
    __syncthreads();
    if (k0 + TK < p.k) fetch(k0 + TK);
#pragma unroll
    for (int ks = 0; ks < TK; ks += 4) {
      const double a0 = As[(ks + l4) * LDS_LD + wm + l15];
      const double a1 = As[(ks + l4) * LDS_LD + wm + 16 + l15];
      const double b0 = Bs[(ks + l4) * LDS_LD + wn + l15];
      const double b1 = Bs[(ks + l4) * LDS_LD + wn + 16 + l15];
      // operands swapped on purpose: result lane layout is C[i = l15][j = l4 + 4r]
      acc[0][0] = hssk_mfma_f64_16x16x4(b0, a0, acc[0][0]);
      acc[0][1] = hssk_mfma_f64_16x16x4(b1, a0, acc[0][1]);
      acc[1][0] = hssk_mfma_f64_16x16x4(b0, a1, acc[1][0]);
      acc[1][1] = hssk_mfma_f64_16x16x4(b1, a1, acc[1][1]);
    }
    __syncthreads();
  }
  // ---- epilogue: all C reads of the tile are issued before the first is consumed
  double cv[2][2][4];
  if (p.beta != 0.) {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int gi = min(i0 + wm + a * 16 + l15, p.m - 1);
          const int gj = min(j0 + wn + b * 16 + l4 + 4 * r, p.n - 1);
          cv[a][b][r] = hssk_gload(p.C, gi + (size_t)gj * p.ldc);
        }
  }
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int gi = i0 + wm + a * 16 + l15;
        const int gj = j0 + wn + b * 16 + l4 + 4 * r;
        double v = p.alpha * acc[a][b][r];
        if (p.beta != 0.) v += p.beta * cv[a][b][r];
        if (gi < p.m && gj < p.n) hssk_gstore(p.C, gi + (size_t)gj * p.ldc, v);
      }
}

// k == 0 (rank-0 nodes): C = beta C
__global__ void gemm_scale_kernel(const hssk_gemm_desc* __restrict__ descs, const Tile* __restrict__ tiles) {
  const Tile t = tiles[blockIdx.x];
  const hssk_gemm_desc p = descs[t.prob];
  for (int e = threadIdx.x; e < TM * TN; e += blockDim.x) {
    int gi = t.tm * TM + (e & 63), gj = t.tn * TN + (e >> 6);
    if (gi < p.m && gj < p.n) {
      double* c = p.C + gi + (size_t)gj * p.ldc;
      *c = (p.beta == 0.) ? 0. : p.beta * (*c);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Panel variant for the sample-update GEMMs of a level:  C(m x n) = alpha A(m x k) op(B) + beta C with
// m = the sample count d (<= 192) -- the bulk of the batched-GEMM flops (leaf level: 2 x 192 x b x b per
// leaf).  One workgroup owns all m rows of a 32-column tile (4 wave64 stacked along M, 48 x 32 each = 3 x 2
// MFMA tiles), so the A panel (R^T, contiguous, 16-byte loads) is staged once per column tile and the
// narrow tile wastes little on ragged n (b = 195 / 196 -> 7 tiles, 87 % useful columns; 64-wide: 77 %).
// K advances 16 per stage through double-buffered LDS with a register prefetch of the next stage.
// ------------------------------------------------------------------------------------------------
constexpr int PBM = 192, PBN = 32, PBK = 16;
constexpr int PLDA = PBM + 16, PLDB = PBN + 16;

template <bool TRANSB>
__global__ __launch_bounds__(256, 2) void gemm_panel_kernel(const hssk_gemm_desc* __restrict__ descs,
                                                            const Tile* __restrict__ tiles) {
  HSSK_SHARED double As[2 * PBK * PLDA];
  HSSK_SHARED double Bs[2 * PBK * PLDB];
  const Tile t = tiles[blockIdx.x];
  if (t.prob < 0) return;  // padding entry of the XCD-aware work list
  const hssk_gemm_desc p = descs[t.prob];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int j0 = t.tn * PBN;
  const int wm = wave * 48;
  const int m = p.m, n = p.n, k = p.k;
  const double* __restrict__ A = p.A;
  const double* __restrict__ B = p.B;

  hssk_d4 acc[3][2];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) acc[a][b] = hssk_d4{0., 0., 0., 0.};

  // A tile: 192 x 16 doubles = 1536 pairs -> 6 per thread; pair e: rows (i, i+1), column kk
  int ldsA[6], kkA[6], iA[6];
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const int e = tid + 256 * r;
    iA[r] = 2 * (e % (PBM / 2));
    kkA[r] = e / (PBM / 2);
    ldsA[r] = kkA[r] * PLDA + iA[r];
  }
  // B tile: 32 x 16 doubles -> 2 per thread
  int jB[2], kkB[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int e = tid + 256 * r;
    if (TRANSB) { jB[r] = e % PBN; kkB[r] = e / PBN; }   // op(B)(k,j) = B(j,k): contiguous along j
    else { kkB[r] = e % PBK; jB[r] = e / PBK; }          // op(B)(k,j) = B(k,j): contiguous along k
  }
  // two register sets: the loads of stage s+2 are in flight while stage s computes (the D / B blocks of a
  // level stream from HBM: a one-stage prefetch leaves ~1.5 us of latency exposed per stage)
  hssk_d2 ra0[6], ra1[6];
  double rb0[2], rb1[2];
  auto load = [&](int k0, hssk_d2 (&ra)[6], double (&rb)[2]) {
#pragma unroll
    for (int r = 0; r < 6; r++) {
      // m is even: a pair is inside or outside.  Clamped address + select keeps the load unconditional.
      const bool ok = (iA[r] < m) && (k0 + kkA[r] < k);
      const hssk_d2 v = hssk_gload2(A, min(iA[r], m - 2) + (size_t)min(k0 + kkA[r], k - 1) * p.lda);
      ra[r] = ok ? v : hssk_d2{0., 0.};
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int gj = j0 + jB[r], gk = k0 + kkB[r];
      const bool ok = gj < n && gk < k;
      const int cj = min(gj, n - 1), ck = min(gk, k - 1);
      const double v = TRANSB ? hssk_gload(B, cj + (size_t)ck * p.ldb) : hssk_gload(B, ck + (size_t)cj * p.ldb);
      rb[r] = ok ? v : 0.;
    }
  };
  auto store = [&](int buf, const hssk_d2 (&ra)[6], const double (&rb)[2]) {
    double* as = As + buf * PBK * PLDA;
    double* bs = Bs + buf * PBK * PLDB;
#pragma unroll
    for (int r = 0; r < 6; r++) *reinterpret_cast<hssk_d2*>(as + ldsA[r]) = ra[r];
#pragma unroll
    for (int r = 0; r < 2; r++) bs[kkB[r] * PLDB + jB[r]] = rb[r];
  };
  auto compute = [&](int buf) {
    const double* as = As + buf * PBK * PLDA + wm + l15;
    const double* bs = Bs + buf * PBK * PLDB + l15;
#pragma unroll
    for (int ks = 0; ks < PBK; ks += 4) {
      double af[3], bf[2];
#pragma unroll
      for (int a = 0; a < 3; a++) af[a] = as[(ks + l4) * PLDA + a * 16];
#pragma unroll
      for (int b = 0; b < 2; b++) bf[b] = bs[(ks + l4) * PLDB + b * 16];
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = hssk_mfma_f64_16x16x4(bf[b], af[a], acc[a][b]);
    }
  };
  const int nst = (k + PBK - 1) / PBK;  // >= 1 (k > 0 for panel-eligible problems)
  load(0, ra0, rb0);
  store(0, ra0, rb0);
  if (nst > 1) load(PBK, ra1, rb1);
  __syncthreads();
  int st = 0;
  // LDS buffer and register set of stage s are both s & 1
  for (; st + 2 < nst; st += 2) {
    load((st + 2) * PBK, ra0, rb0);
    compute(0);
    store(1, ra1, rb1);
    __syncthreads();
    if (st + 3 < nst) load((st + 3) * PBK, ra1, rb1);
    compute(1);
    store(0, ra0, rb0);
    __syncthreads();
  }
  compute(0);
  if (st + 1 < nst) {
    store(1, ra1, rb1);
    __syncthreads();
    compute(1);
  }
  double cv[3][2][4];
  if (p.beta != 0.) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          cv[a][b][r] = hssk_gload(p.C, min(wm + a * 16 + l15, m - 1) + (size_t)min(j0 + b * 16 + l4 + 4 * r, n - 1) * p.ldc);
  }
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int gi = wm + a * 16 + l15;
        const int gj = j0 + b * 16 + l4 + 4 * r;
        double v = p.alpha * acc[a][b][r];
        if (p.beta != 0.) v += p.beta * cv[a][b][r];
        if (gi < m && gj < n) hssk_gstore(p.C, gi + (size_t)gj * p.ldc, v);
      }
}

// ------------------------------------------------------------------------------------------------
// Fused leaf sample update:  Sr(:, J) -= R D(J, :)^T  and  Sc(:, J) -= R D(:, J)  for one 32-column tile J of
// one leaf.  Same tiling as gemm_panel_kernel, but the staged R panel feeds BOTH products (two B tiles per
// stage), so every barrier covers 48 MFMAs per wave instead of 24 and the panel is read once, not twice.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void leaf_update_kernel(const hssk_leaf_update_desc* __restrict__ descs,
                                                             const Tile* __restrict__ tiles) {
  HSSK_SHARED double As[2 * PBK * PLDA];
  HSSK_SHARED double Bs[2 * 2 * PBK * PLDB];   // [buffer][product][k][j]
  const Tile t = tiles[blockIdx.x];
  if (t.prob < 0) return;
  const hssk_leaf_update_desc p = descs[t.prob];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int j0 = t.tn * PBN;
  const int wm = wave * 48;
  const int d = p.d, m = p.m;
  const double* __restrict__ A = p.R;
  const double* __restrict__ D = p.D;

  hssk_d4 acc[2][3][2];
#pragma unroll
  for (int q = 0; q < 2; q++)
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 2; b++) acc[q][a][b] = hssk_d4{0., 0., 0., 0.};

  int ldsA[6], kkA[6], iA[6];
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const int e = tid + 256 * r;
    iA[r] = 2 * (e % (PBM / 2));
    kkA[r] = e / (PBM / 2);
    ldsA[r] = kkA[r] * PLDA + iA[r];
  }
  // product 0 (Sr): op(B)(k, j) = D(j0+j, k)  -> contiguous along j ;  product 1 (Sc): D(k, j0+j) -> along k
  int j1[2], k1[2], j2[2], k2[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int e = tid + 256 * r;
    j1[r] = e % PBN; k1[r] = e / PBN;
    k2[r] = e % PBK; j2[r] = e / PBK;
  }
  hssk_d2 ra[6];
  double rb1[2], rb2[2];
  auto load = [&](int k0) {
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const bool ok = (iA[r] < d) && (k0 + kkA[r] < m);
      const hssk_d2 v = hssk_gload2(A, min(iA[r], d - 2) + (size_t)min(k0 + kkA[r], m - 1) * p.ldr);
      ra[r] = ok ? v : hssk_d2{0., 0.};
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int gj = j0 + j1[r], gk = k0 + k1[r];
      const double v1 = hssk_gload(D, min(gj, m - 1) + (size_t)min(gk, m - 1) * p.ldd);
      rb1[r] = (gj < m && gk < m) ? v1 : 0.;
      const int hj = j0 + j2[r], hk = k0 + k2[r];
      const double v2 = hssk_gload(D, min(hk, m - 1) + (size_t)min(hj, m - 1) * p.ldd);
      rb2[r] = (hj < m && hk < m) ? v2 : 0.;
    }
  };
  auto store = [&](int buf) {
    double* as = As + buf * PBK * PLDA;
    double* b1 = Bs + (buf * 2 + 0) * PBK * PLDB;
    double* b2 = Bs + (buf * 2 + 1) * PBK * PLDB;
#pragma unroll
    for (int r = 0; r < 6; r++) *reinterpret_cast<hssk_d2*>(as + ldsA[r]) = ra[r];
#pragma unroll
    for (int r = 0; r < 2; r++) {
      b1[k1[r] * PLDB + j1[r]] = rb1[r];
      b2[k2[r] * PLDB + j2[r]] = rb2[r];
    }
  };
  auto compute = [&](int buf) {
    const double* as = As + buf * PBK * PLDA + wm + l15;
    const double* b1 = Bs + (buf * 2 + 0) * PBK * PLDB + l15;
    const double* b2 = Bs + (buf * 2 + 1) * PBK * PLDB + l15;
#pragma unroll
    for (int ks = 0; ks < PBK; ks += 4) {
      double af[3], bf[2][2];
#pragma unroll
      for (int a = 0; a < 3; a++) af[a] = as[(ks + l4) * PLDA + a * 16];
#pragma unroll
      for (int b = 0; b < 2; b++) { bf[0][b] = b1[(ks + l4) * PLDB + b * 16]; bf[1][b] = b2[(ks + l4) * PLDB + b * 16]; }
#pragma unroll
      for (int q = 0; q < 2; q++)
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int b = 0; b < 2; b++) acc[q][a][b] = hssk_mfma_f64_16x16x4(bf[q][b], af[a], acc[q][a][b]);
    }
  };
  const int nst = (m + PBK - 1) / PBK;
  load(0);
  store(0);
  __syncthreads();
  int buf = 0;
  for (int st = 0; st + 1 < nst; st++) {
    load((st + 1) * PBK);
    compute(buf);
    store(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  compute(buf);
#pragma unroll
  for (int q = 0; q < 2; q++) {
    double* C = q == 0 ? p.Sr : p.Sc;
    double cv[3][2][4];   // every read of the tile in flight before the first is consumed
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          cv[a][b][r] = hssk_gload(C, min(wm + a * 16 + l15, d - 1) + (size_t)min(j0 + b * 16 + l4 + 4 * r, m - 1) * p.lds);
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int gi = wm + a * 16 + l15;
          const int gj = j0 + b * 16 + l4 + 4 * r;
          if (gi < d && gj < m) hssk_gstore(C, gi + (size_t)gj * p.lds, cv[a][b][r] - acc[q][a][b][r]);
        }
  }
}

inline bool panel_eligible(const hssk_gemm_desc& d) {
  return d.transA == 0 && d.m > 64 && d.m <= PBM && (d.m % 2 == 0) && (d.lda % 2 == 0) &&
         ((size_t)d.A % 16 == 0) && d.k > 0 && d.n > 0;
}


// ------------------------------------------------------------------------------------------------
// Tall variant for C(m x n) = alpha op(A) B + beta C with FEW columns (n <= 64) and a B block that fits the LDS
// (k <= 256): the leaf-level products of a mat-vec / solve with many right-hand sides (D x, Q~ y: 256 x 256 blocks
// against 64 columns).  The 64 x 64 tiles above stage both operands through LDS one K stage at a time and reach
// 24 TFLOP/s on those shapes; here B is loaded into LDS once ([k][65]: conflict-free for the transposing load and
// for the MFMA operand reads), a workgroup owns TALL_M rows of C and each wave streams the A fragments of its
// 16-row tiles straight from global memory into registers, TALL_CH k-steps ahead, against all four 16-column
// tiles of B.  The MFMA takes the B fragment first, so a lane holds C[i = l & 15][j = (l >> 4) + 4 r]: sixteen
// lanes store sixteen consecutive rows.
// ------------------------------------------------------------------------------------------------
// Round 6: 512 threads and B in the LDS HALF of K at a time.  The first form (1024 threads, all of B: 101 KB for the 195-row
// leaves of N = 1e5) was one workgroup per CU by registers and by LDS, so the 512 leaves ran as two rounds of workgroups whose
// phases -- load B, stream A, store C -- had nothing to overlap with: 116 us for the leaves' D X at 64 right-hand sides (2.2 TB/s),
// 193 - 245 us next to the inner levels' sweep, the longest launch of the mat-vec and of the solve.  Now a wave owns two 16-row
// tiles (accumulators of both live across the halves), two workgroups share a CU and every leaf is resident at once.
constexpr int TALL_T = 512, TALL_M = 256, TALL_N = 64, TALL_LDB = TALL_N + 1, TALL_CH = 8, TALL_KMAX = 256, TALL_KH = 128;

__global__ __launch_bounds__(TALL_T) HSSK_WAVES_PER_SIMD(4) void gemm_tall_kernel(const hssk_gemm_desc* __restrict__ descs, const Tile* __restrict__ tiles) {
  HSSK_DYN_SHARED(double, Bs);
  const Tile t = tiles[blockIdx.x];
  const hssk_gemm_desc p = descs[t.prob];
  const int tid = threadIdx.x, lane = tid & 63, wave = hssk_uniform(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int m = p.m, n = p.n, k = p.k;
  const int nh = (k + TALL_KH - 1) / TALL_KH;
  const int kc = (((k + nh - 1) / nh) + 3) & ~3;   // rows of B per pass (a multiple of the MFMA's four k)
  const int r0 = t.tm * TALL_M;
  constexpr int NW = TALL_T / 64, NTL = TALL_M / 16 / NW;   // tiles per wave: wave, wave + NW
  hssk_d4 acc[NTL][4];
#pragma unroll
  for (int q = 0; q < NTL; q++)
#pragma unroll
    for (int ct = 0; ct < 4; ct++) acc[q][ct] = hssk_d4{0., 0., 0., 0.};
  for (int kb = 0; kb < k; kb += kc) {
    const int kn = min(kc, k - kb);
    if (kb) __syncthreads();   // (every wave is done with the previous rows of B)
    // B(kb : kb + kn, :) -> LDS (lanes along k: contiguous in memory), zero beyond column n
    {
      const int cs = TALL_T / kn, kk = tid % kn, jq = tid / kn;   // kn <= 128: cs >= 4 columns per pass
      if (jq < cs)
        for (int j = jq; j < TALL_N; j += 8 * cs) {   // eight loads in flight (one at a time: a memory round trip per column)
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; u++) v[u] = hssk_gload(p.B, (size_t)(kb + kk) + (size_t)min(j + u * cs, n - 1) * p.ldb);
#pragma unroll
          for (int u = 0; u < 8; u++)
            if (j + u * cs < TALL_N) Bs[kk * TALL_LDB + j + u * cs] = j + u * cs < n ? v[u] : 0.;
        }
    }
    __syncthreads();
    const int nks = (kn + 3) >> 2;
    const int nch = hssk_uniform((nks + TALL_CH - 1) / TALL_CH);
#pragma unroll
    for (int q = 0; q < NTL; q++) {
      const int tl = wave + q * NW;
      if (r0 + tl * 16 >= m) continue;   // (uniform in the wave)
      const int i = r0 + tl * 16 + l15;
      // A fragment element, indices clamped into the block: no branch around the load and no select behind it (either makes the
      // compiler wait for the load long before the matrix cores need it).  Rows beyond m are never stored; k-steps beyond the
      // pass multiply by zeros read in place of B.
      auto aload = [&](int kk) -> double {
        const int ic = min(i, m - 1), kg = kb + min(kk, kn - 1);
        return p.transA ? hssk_gload(p.A, (size_t)kg + (size_t)ic * p.lda) : hssk_gload(p.A, (size_t)ic + (size_t)kg * p.lda);
      };
      auto chunk = [&](int k0, const double (&a)[TALL_CH]) {
#pragma unroll
        for (int u = 0; u < TALL_CH; u++) {
          const int kk = k0 + 4 * u;
          const double* xr = Bs + min(kk, kn - 1) * TALL_LDB + l15;
#pragma unroll
          for (int ct = 0; ct < 4; ct++) acc[q][ct] = hssk_mfma_f64_16x16x4(kk < kn ? xr[ct * 16] : 0., a[u], acc[q][ct]);
        }
      };
      // two register sets in turn: the fragments of the next chunk are in flight while the matrix cores work on the current one
      double a0[TALL_CH], a1[TALL_CH];
#pragma unroll
      for (int u = 0; u < TALL_CH; u++) a0[u] = aload(4 * u + l4);
      for (int c = 0; c < nch; c += 2) {
        const int k0 = 4 * TALL_CH * c + l4;
#pragma unroll
        for (int u = 0; u < TALL_CH; u++) a1[u] = aload(k0 + 4 * TALL_CH + 4 * u);
        chunk(k0, a0);
#pragma unroll
        for (int u = 0; u < TALL_CH; u++) a0[u] = aload(k0 + 8 * TALL_CH + 4 * u);
        chunk(k0 + 4 * TALL_CH, a1);   // (beyond the pass: zeros on the B side)
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NTL; q++) {
    const int i = r0 + (wave + q * NW) * 16 + l15;
    if (i < m) {
#pragma unroll
      for (int ct = 0; ct < 4; ct++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int j = ct * 16 + l4 + 4 * r;
          if (j < n) {
            double v = p.alpha * acc[q][ct][r];
            if (p.beta != 0.) v += p.beta * hssk_gload(p.C, (size_t)i + (size_t)j * p.ldc);
            hssk_gstore(p.C, (size_t)i + (size_t)j * p.ldc, v);
          }
        }
    }
  }
}
// ---- products with at most four columns (mat-vec shaped: the solve phases of a BLR front with one right-hand side -- V^T x
// and U t per low-rank tile --, the dense leaf products of an HSS mat-vec / solve at leaf sizes beyond the single-launch
// sweep).  The MFMA tile kernel spends a 64 x 64 tile's staging and its K loop of barriers on them: 38 - 44 us per launch for
// the tiles of a BLR block column, 212 us for the 196 leaf blocks of N = 1e5 at leaf 512 (411 MB: 1.9 TB/s).  Here the
// columns of op(B) sit in the LDS (k <= 1024) and A streams once: not transposed, a thread per row of C (coalesced along
// the rows, the waves of a workgroup along k, the vector entries read as LDS broadcasts); transposed, a wave per row of C (the lanes along the contiguous
// column of A, a wave sum per output).
constexpr int GV_T = 256, GV_N = 4, GV_K = 1024, GV_ROWS_T = 64;
__global__ __launch_bounds__(GV_T) void gemv_small_kernel(const hssk_gemm_desc* __restrict__ descs, const Tile* __restrict__ tiles) {
  HSSK_SHARED double s_x[GV_K * GV_N];
  const Tile t = tiles[blockIdx.x];
  const hssk_gemm_desc p = descs[t.prob];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = p.m, n = p.n, k = p.k;
  for (int e = tid; e < k * GV_N; e += GV_T) {
    const int kk = e % k, c = e / k;
    s_x[kk + c * GV_K] = c < n ? hssk_gload(p.B, p.transB ? (size_t)c + (size_t)kk * p.ldb : (size_t)kk + (size_t)c * p.ldb) : 0.;
  }
  __syncthreads();
  if (!p.transA) {
    // 64 rows of C per workgroup, a lane per row; the four waves take a quarter of k each (one long product -- the U panel of
    // a BLR block row against its stacked V^T x, k ~ 600 -- is then 19 batches of eight loads per lane instead of 75) and
    // meet in the LDS
    HSSK_SHARED double s_red[GV_T / 64][GV_N][64];
    const int row = t.tm * 64 + lane;
    const int kq = (k + GV_T / 64 - 1) / (GV_T / 64), k0 = wave * kq, k1 = min(k, k0 + kq);
    double acc[GV_N] = {0., 0., 0., 0.};
    if (row < m) {
      int kk = k0;
      for (; kk + 8 <= k1; kk += 8) {
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = hssk_gload(p.A, (size_t)row + (size_t)(kk + u) * p.lda);
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
          for (int c = 0; c < GV_N; c++) acc[c] += a[u] * s_x[kk + u + c * GV_K];
      }
      for (; kk < k1; kk++) {
        const double a = hssk_gload(p.A, (size_t)row + (size_t)kk * p.lda);
#pragma unroll
        for (int c = 0; c < GV_N; c++) acc[c] += a * s_x[kk + c * GV_K];
      }
    }
#pragma unroll
    for (int c = 0; c < GV_N; c++) s_red[wave][c][lane] = acc[c];
    __syncthreads();
    if (wave == 0 && row < m) {
#pragma unroll
      for (int c = 0; c < GV_N; c++)
        if (c < n) {
          double v = s_red[0][c][lane];
#pragma unroll
          for (int w = 1; w < GV_T / 64; w++) v += s_red[w][c][lane];
          const size_t o = (size_t)row + (size_t)c * p.ldc;
          hssk_gstore(p.C, o, p.beta == 0. ? p.alpha * v : p.alpha * v + p.beta * hssk_gload(p.C, o));
        }
    }
  } else {
    for (int jj = wave; jj < GV_ROWS_T; jj += GV_T / 64) {
      const int j = t.tm * GV_ROWS_T + jj;
      if (j >= m) break;
      double acc[GV_N] = {0., 0., 0., 0.};
      for (int kk = lane; kk < k; kk += 64) {
        const double a = hssk_gload(p.A, (size_t)kk + (size_t)j * p.lda);
#pragma unroll
        for (int c = 0; c < GV_N; c++) acc[c] += a * s_x[kk + c * GV_K];
      }
#pragma unroll
      for (int c = 0; c < GV_N; c++) acc[c] = hssk_wave_sum(acc[c]);
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < GV_N; c++)
          if (c < n) {
            const size_t o = (size_t)j + (size_t)c * p.ldc;
            hssk_gstore(p.C, o, p.beta == 0. ? p.alpha * acc[c] : p.alpha * acc[c] + p.beta * hssk_gload(p.C, o));
          }
      }
    }
  }
}
bool gemv_eligible(const hssk_gemm_desc& d) {
  static const bool off = [] { const char* e = std::getenv("HSSK_GEMM_NO_GEMV"); return e && e[0] == '1'; }();
  return !off && d.n <= GV_N && d.k >= 1 && d.k <= GV_K;
}
bool tall_eligible(const hssk_gemm_desc& d) {
  static const bool off = [] { const char* e = std::getenv("HSSK_GEMM_NO_TALL"); return e && e[0] == '1'; }();
  return !off && !d.transB && d.n > 16 && d.n <= TALL_N && d.k >= 32 && d.k <= TALL_KMAX && d.m >= 96;
}

}  // namespace

extern "C" int hssk_gemm_vbatched(hssk_ctx* ctx, const hssk_gemm_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  std::vector<Tile> tiles, ztiles, ptilesN, ptilesT, ttiles, vtiles;
  int kmax_tall = 0;
  for (int p = 0; p < count; p++) {
    const hssk_gemm_desc& d = descs[p];
    if (d.m <= 0 || d.n <= 0) continue;
    if (gemv_eligible(d)) {
      const int per = 64;   // rows of C per workgroup, either form (GV_ROWS_T)
      for (int tm = 0; tm * per < d.m; tm++) vtiles.push_back(Tile{p, tm, 0});
      continue;
    }
    if (tall_eligible(d)) {
      for (int tm = 0; tm * TALL_M < d.m; tm++) ttiles.push_back(Tile{p, tm, 0});
      {   // rows of B per pass of this problem (the kernel's own arithmetic): the LDS is sized for the largest
        const int nh = (d.k + TALL_KH - 1) / TALL_KH;
        kmax_tall = std::max(kmax_tall, (((d.k + nh - 1) / nh) + 3) & ~3);
      }
      continue;
    }
    if (panel_eligible(d)) {
      std::vector<Tile>& dst = d.transB ? ptilesT : ptilesN;
      for (int tn = 0; tn * PBN < d.n; tn++) dst.push_back(Tile{p, 0, tn});
      continue;
    }
    int ntm = (d.m + TM - 1) / TM, ntn = (d.n + TN - 1) / TN;
    std::vector<Tile>& dst = (d.k > 0) ? tiles : ztiles;
    for (int tn = 0; tn < ntn; tn++)
      for (int tm = 0; tm < ntm; tm++) dst.push_back(Tile{p, tm, tn});
  }
  if (tiles.empty() && ztiles.empty() && ptilesN.empty() && ptilesT.empty() && ttiles.empty() && vtiles.empty()) return 0;
  auto* d_descs = (const hssk_gemm_desc*)ctx->stage(descs, sizeof(hssk_gemm_desc) * count);
  // XCD-aware order for the panel tiles: workgroup b runs on XCD b % 8 and every XCD has its own L2, so
  // the column tiles of one problem (which share the 192 x k A panel) are placed on block ids that are
  // congruent mod 8 and adjacent in time; otherwise each of them re-fetches the panel from HBM / MALL
  auto xcd_order = [](std::vector<Tile>& v) {
    if (v.size() < 16) return;
    std::vector<std::vector<Tile>> lanes(8);
    int cur = -1, lane = -1;
    for (const Tile& t : v) {   // tiles of a problem are contiguous in v
      if (t.prob != cur) { cur = t.prob; lane = (lane + 1) & 7; }
      lanes[lane].push_back(t);
    }
    std::vector<Tile> out;
    out.reserve(v.size() + 64);
    size_t longest = 0;
    for (auto& l : lanes) longest = std::max(longest, l.size());
    for (size_t i = 0; i < longest; i++)
      for (int x = 0; x < 8; x++)
        out.push_back(i < lanes[x].size() ? lanes[x][i] : Tile{-1, 0, 0});  // padding keeps b % 8 == lane
    v.swap(out);
  };
  xcd_order(ptilesN);
  xcd_order(ptilesT);
  if (!ptilesN.empty()) {
    auto* d_tiles = (const Tile*)ctx->stage(ptilesN.data(), sizeof(Tile) * ptilesN.size());
    HSSK_LAUNCH((gemm_panel_kernel<false>), dim3((unsigned)ptilesN.size()), dim3(256), 0, ctx->stream, d_descs, d_tiles);
  }
  if (!ptilesT.empty()) {
    auto* d_tiles = (const Tile*)ctx->stage(ptilesT.data(), sizeof(Tile) * ptilesT.size());
    HSSK_LAUNCH((gemm_panel_kernel<true>), dim3((unsigned)ptilesT.size()), dim3(256), 0, ctx->stream, d_descs, d_tiles);
  }
  if (!vtiles.empty()) {
    auto* d_tiles = (const Tile*)ctx->stage(vtiles.data(), sizeof(Tile) * vtiles.size());
    HSSK_LAUNCH(gemv_small_kernel, dim3((unsigned)vtiles.size()), dim3(GV_T), 0, ctx->stream, d_descs, d_tiles);
  }
  if (!ttiles.empty()) {
    auto* d_tiles = (const Tile*)ctx->stage(ttiles.data(), sizeof(Tile) * ttiles.size());
    const size_t lds = sizeof(double) * (size_t)kmax_tall * TALL_LDB;   // (the largest pass of any problem)
    hssk_rt::allow_dynamic_lds(gemm_tall_kernel, lds);
    HSSK_LAUNCH(gemm_tall_kernel, dim3((unsigned)ttiles.size()), dim3(TALL_T), lds, ctx->stream, d_descs, d_tiles);
  }
  if (!tiles.empty()) {
    auto* d_tiles = (const Tile*)ctx->stage(tiles.data(), sizeof(Tile) * tiles.size());
    HSSK_LAUNCH(gemm_vbatched_kernel, dim3((unsigned)tiles.size()), dim3(256), 0, ctx->stream, d_descs, d_tiles);
  }
  if (!ztiles.empty()) {
    auto* d_tiles = (const Tile*)ctx->stage(ztiles.data(), sizeof(Tile) * ztiles.size());
    HSSK_LAUNCH(gemm_scale_kernel, dim3((unsigned)ztiles.size()), dim3(256), 0, ctx->stream, d_descs, d_tiles);
  }
  hssk_rt::check_launch();
  HSSK_API_END
}

extern "C" int hssk_leaf_update_vbatched(hssk_ctx* ctx, const hssk_leaf_update_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  std::vector<Tile> tiles;
  for (int p = 0; p < count; p++) {
    const hssk_leaf_update_desc& d = descs[p];
    if (d.d <= 0 || d.m <= 0) continue;
    if (d.d > PBM || (d.d % 2) || (d.ldr % 2) || ((size_t)d.R % 16)) HSSK_UNSUPPORTED("panel shape outside the fused leaf update");  // caller falls back to two GEMMs
    for (int tn = 0; tn * PBN < d.m; tn++) tiles.push_back(Tile{p, 0, tn});
  }
  if (tiles.empty()) return 0;
  auto* dd = (const hssk_leaf_update_desc*)ctx->stage(descs, sizeof(*descs) * count);
  auto* dt = (const Tile*)ctx->stage(tiles.data(), sizeof(Tile) * tiles.size());
  HSSK_LAUNCH(leaf_update_kernel, dim3((unsigned)tiles.size()), dim3(256), 0, ctx->stream, dd, dt);
  hssk_rt::check_launch();
  HSSK_API_END
}
