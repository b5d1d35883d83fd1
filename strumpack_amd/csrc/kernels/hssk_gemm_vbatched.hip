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

  for (int k0 = 0; k0 < p.k; k0 += TK) {
    // ---- stage A tile: As[kk][i] = op(A)(i0+i, k0+kk)
    if (!p.transA) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int i = tid & 63, kk = (tid >> 6) + 4 * r;
        int gi = i0 + i, gk = k0 + kk;
        double v = (gi < p.m && gk < p.k) ? p.A[gi + (size_t)gk * p.lda] : 0.;
        As[kk * LDS_LD + i] = v;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int kk = tid & 15, i = (tid >> 4) + 16 * r;
        int gi = i0 + i, gk = k0 + kk;
        double v = (gi < p.m && gk < p.k) ? p.A[gk + (size_t)gi * p.lda] : 0.;
        As[kk * LDS_LD + i] = v;
      }
    }
    // ---- stage B tile: Bs[kk][j] = op(B)(k0+kk, j0+j)
    if (!p.transB) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int kk = tid & 15, j = (tid >> 4) + 16 * r;
        int gj = j0 + j, gk = k0 + kk;
        double v = (gj < p.n && gk < p.k) ? p.B[gk + (size_t)gj * p.ldb] : 0.;
        Bs[kk * LDS_LD + j] = v;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int j = tid & 63, kk = (tid >> 6) + 4 * r;
        int gj = j0 + j, gk = k0 + kk;
        double v = (gj < p.n && gk < p.k) ? p.B[gj + (size_t)gk * p.ldb] : 0.;
        Bs[kk * LDS_LD + j] = v;
      }
    }
    __syncthreads();
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
  // ---- epilogue
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int gi = i0 + wm + a * 16 + l15;
        int gj = j0 + wn + b * 16 + l4 + 4 * r;
        if (gi < p.m && gj < p.n) {
          double* c = p.C + gi + (size_t)gj * p.ldc;
          double v = p.alpha * acc[a][b][r];
          if (p.beta != 0.) v += p.beta * (*c);
          *c = v;
        }
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

}  // namespace

extern "C" int hssk_gemm_vbatched(hssk_ctx* ctx, const hssk_gemm_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  std::vector<Tile> tiles, ztiles;
  for (int p = 0; p < count; p++) {
    const hssk_gemm_desc& d = descs[p];
    if (d.m <= 0 || d.n <= 0) continue;
    int ntm = (d.m + TM - 1) / TM, ntn = (d.n + TN - 1) / TN;
    std::vector<Tile>& dst = (d.k > 0) ? tiles : ztiles;
    for (int tn = 0; tn < ntn; tn++)
      for (int tm = 0; tm < ntm; tm++) dst.push_back(Tile{p, tm, tn});
  }
  if (tiles.empty() && ztiles.empty()) return 0;
  auto* d_descs = (const hssk_gemm_desc*)ctx->stage(descs, sizeof(hssk_gemm_desc) * count);
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
