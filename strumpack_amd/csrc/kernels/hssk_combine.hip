// hssk_gather_combine: out(:, j) = G(:, g_j) + alpha sum_k M(:, m_k) C(j, k) -- a column gather and a small product in ONE
// launch.  The inner levels of the HSS compression are chains of such pairs on tiny operands (the children's skeleton rows
// minus the coupling blocks times the sibling's reduced samples; the reduction of the random samples by the new basis,
// HSSMatrix.compress.hpp:524-629, 689-724): as separate gathers and batched GEMMs every link of the chain was a launch of
// ~10 us whatever its size, 9 dependent launches per level.
//
// Shape of the work: the long dimension (the rows: sample index, <= a few hundred) is contiguous in every operand, the
// output has J <= ~100 columns, the sum K <= ~200 terms.  One thread per row, JB = 64 output columns of the row in registers,
// the coefficients (uniform over the threads) staged in LDS and read as broadcasts; a basis column is read from global memory
// exactly once per row (coalesced along the rows), eight loads in flight.
#include "hssk_device.h"
#include "hssk_internal.h"

#include <vector>

namespace {

constexpr int GC_T = 256;    // rows per workgroup
// output columns per workgroup (registers): 64, or 16 when that leaves most of the chip idle (the top levels of a tree)
constexpr int GC_KC = 64;    // coefficient rows staged at a time
constexpr int GC_KP = 8;     // basis values loaded together
struct GcWork { int prob, rchunk, jchunk; };

__device__ __forceinline__ const double* gc_col(const double* s0, const double* s1, int split, int ld, int col) {
  return col < split ? s0 + (size_t)col * ld : s1 + (size_t)(col - split) * ld;
}

template <int GC_JB>
__global__ __launch_bounds__(GC_T) void gather_combine_kernel(const hssk_combine_desc* __restrict__ descs, const GcWork* __restrict__ work) {
  HSSK_SHARED double s_c[GC_KC * GC_JB];   // [k][j], alpha folded in
  HSSK_SHARED int s_m[GC_KC];              // source column of basis term k
  const GcWork w = work[blockIdx.x];
  const hssk_combine_desc p = descs[w.prob];
  const int tid = threadIdx.x, i = w.rchunk * GC_T + tid;
  const bool live = i < p.rows;
  const int j0 = w.jchunk * GC_JB, Jc = min(GC_JB, p.J - j0);
  double acc[GC_JB];
#pragma unroll
  for (int jb = 0; jb < GC_JB / 8; jb++) {
    if (8 * jb < Jc) {
#pragma unroll
      for (int j = 8 * jb; j < 8 * jb + 8; j++) {
        double v = 0.;
        if (live && j < Jc && p.G0) {
          const int c = p.gidx ? p.gidx[j0 + j] : j0 + j;
          v = hssk_gload(gc_col(p.G0, p.G1, p.gsplit, p.ldg, c), (size_t)i);
        }
        acc[j] = v;
      }
    } else {
#pragma unroll
      for (int j = 8 * jb; j < 8 * jb + 8; j++) acc[j] = 0.;
    }
  }
  for (int k0 = 0; k0 < p.K; k0 += GC_KC) {
    const int Kc = min(GC_KC, p.K - k0);
    __syncthreads();   // (the previous chunk has been consumed)
    for (int e = tid; e < GC_KC * GC_JB; e += GC_T) {
      const int j = e % GC_JB, kk = e / GC_JB;
      s_c[e] = (j < Jc && kk < Kc) ? p.alpha * p.C[(size_t)(j0 + j) * p.csj + (size_t)(k0 + kk) * p.csk] : 0.;
    }
    if (tid < GC_KC) s_m[tid] = tid < Kc ? (p.midx ? p.midx[k0 + tid] : k0 + tid) : 0;
    __syncthreads();
    // GC_KP basis values of the row are loaded together and one batch ahead of their use (independent loads in flight
    // under the fmas of the previous batch: a lone load per term left the loop waiting out one memory round trip per k)
    auto fetch = [&](double (&mk)[GC_KP], int kb) {
#pragma unroll
      for (int u = 0; u < GC_KP; u++)
        mk[u] = (live && kb + u < Kc) ? hssk_gload(gc_col(p.M0, p.M1, p.msplit, p.ldm, s_m[kb + u]), (size_t)i) : 0.;
    };
    double cur[GC_KP], nxt[GC_KP];
    fetch(cur, 0);
    for (int kb = 0; kb < Kc; kb += GC_KP) {
      if (kb + GC_KP < Kc) fetch(nxt, kb + GC_KP);
#pragma unroll
      for (int u = 0; u < GC_KP; u++) {
        const double* c = s_c + (kb + u) * GC_JB;
#pragma unroll
        for (int jb = 0; jb < GC_JB / 8; jb++) {
          if (8 * jb < Jc) {
#pragma unroll
            for (int j = 8 * jb; j < 8 * jb + 8; j++) acc[j] += cur[u] * c[j];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < GC_KP; u++) cur[u] = nxt[u];
    }
  }
  if (live) {
#pragma unroll
    for (int jb = 0; jb < GC_JB / 8; jb++) {
      if (8 * jb < Jc) {
#pragma unroll
        for (int j = 8 * jb; j < 8 * jb + 8; j++)
          if (j < Jc) hssk_gstore(p.out, (size_t)i + (size_t)(j0 + j) * p.ldo, acc[j]);
      }
    }
  }
}

}  // namespace

extern "C" int hssk_gather_combine(hssk_ctx* ctx, const hssk_combine_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  auto items = [&](int jb, std::vector<GcWork>* work) {
    long long n = 0;
    for (int p = 0; p < count; p++) {
      const hssk_combine_desc& d = descs[p];
      if (d.rows <= 0 || d.J <= 0) continue;
      for (int jc = 0; jc * jb < d.J; jc++)
        for (int rc = 0; rc * GC_T < d.rows; rc++, n++)
          if (work) work->push_back(GcWork{p, rc, jc});
    }
    return n;
  };
  for (int p = 0; p < count; p++)
    if (descs[p].rows > 0 && descs[p].J > 0 && descs[p].K > 0 && (!descs[p].M0 || !descs[p].C)) HSSK_UNSUPPORTED("product without its operands");
  const int jb = items(64, nullptr) >= 512 ? 64 : 16;
  std::vector<GcWork> work;
  if (items(jb, &work) == 0) return 0;
  auto* dd = (const hssk_combine_desc*)ctx->stage(descs, sizeof(*descs) * count);
  auto* dw = (const GcWork*)ctx->stage(work.data(), sizeof(GcWork) * work.size());
  if (jb == 64) HSSK_LAUNCH(gather_combine_kernel<64>, dim3((unsigned)work.size()), dim3(GC_T), 0, ctx->stream, dd, dw);
  else HSSK_LAUNCH(gather_combine_kernel<16>, dim3((unsigned)work.size()), dim3(GC_T), 0, ctx->stream, dd, dw);
  hssk_rt::check_launch();
  HSSK_API_END
}
