// Kernel-matrix front end on the device: entries K(I, J) of a Gauss / Laplace / ANOVA kernel
// evaluated straight from the d x n point coordinates in HBM, exact k-nearest-neighbour lists, and the
// prediction sum of kernel ridge regression.
//
// Reference behaviour restated: kernel::Kernel::operator()(I, J, B) and eval() (kernel/Kernel.hpp:122-147:
// k(x_i, x_j) + lambda on the diagonal), GaussKernel / LaplaceKernel / ANOVAKernel::eval_kernel_function
// (:333-399), Kernel::predict (kernel/KernelRegression.hpp:112-123), and the neighbour lists that
// HSSMatrix::compress_with_coordinates asks find_approximate_neighbors for (HSSMatrix.compress_kernel.hpp:58-66).
// The reference approximates the neighbours with random projection trees on the CPU; here they are exact --
// n^2 d flops are a few milliseconds on this machine -- which only improves the column sample the
// compression draws from them.
//
// Bounds: kernel_eval is exp-throughput / HBM-write bound (one exp per output double); knn is LDS-bandwidth
// bound (two LDS reads per squared difference); both are far from the sketch's MFMA bound and run once.
#include "hssk_device.h"
#include "hssk_internal.h"

#include <algorithm>
#include <vector>

namespace {

constexpr int KE_T = 64;   // output tile edge
constexpr int KE_DC = 32;  // coordinates staged per pass

struct KTile {
  int prob, tr, tc;
};

__device__ inline int row_id(const hssk_keval_desc& p, int a) { return p.ri ? p.ri[a] : p.r0 + a; }
__device__ inline int col_id(const hssk_keval_desc& p, int b) { return p.ci ? p.ci[b] : p.c0 + b; }

// Gauss (type 0): exp(-|x-y|_2^2 / (2 h^2)); Laplace (type 1): exp(-|x-y|_1 / h)
__global__ __launch_bounds__(256) void kernel_eval_kernel(hssk_kernel_spec ks, const hssk_keval_desc* __restrict__ descs,
                                                          const KTile* __restrict__ tiles) {
  HSSK_SHARED double xr[KE_T * (KE_DC + 1)];
  HSSK_SHARED double xc[KE_T * (KE_DC + 1)];
  HSSK_SHARED int idr[KE_T];
  HSSK_SHARED int idc[KE_T];
  const KTile t = tiles[blockIdx.x];
  const hssk_keval_desc p = descs[t.prob];
  const int tid = threadIdx.x, a = tid & 63, bq = tid >> 6;  // thread: row a, columns bq*16 .. bq*16+15
  const int a0 = t.tr * KE_T, b0 = t.tc * KE_T;
  if (tid < KE_T) {
    idr[tid] = a0 + tid < p.nr ? row_id(p, a0 + tid) : -1;
    idc[tid] = b0 + tid < p.nc ? col_id(p, b0 + tid) : -1;
  }
  __syncthreads();
  double acc[16];
#pragma unroll
  for (int b = 0; b < 16; b++) acc[b] = 0.;
  for (int d0 = 0; d0 < ks.d; d0 += KE_DC) {
    const int dc = min(KE_DC, ks.d - d0);
    for (int e = tid; e < KE_T * dc; e += 256) {
      const int pt = e / dc, j = e % dc;
      const int gr = idr[pt], gc = idc[pt];
      xr[pt * (KE_DC + 1) + j] = gr >= 0 ? hssk_gload(ks.X, (size_t)gr * ks.d + d0 + j) : 0.;
      xc[pt * (KE_DC + 1) + j] = gc >= 0 ? hssk_gload(ks.X, (size_t)gc * ks.d + d0 + j) : 0.;
    }
    __syncthreads();
    for (int j = 0; j < dc; j++) {
      const double x = xr[a * (KE_DC + 1) + j];
#pragma unroll
      for (int b = 0; b < 16; b++) {
        const double df = x - xc[(bq * 16 + b) * (KE_DC + 1) + j];
        acc[b] += ks.type == 0 ? df * df : fabs(df);
      }
    }
    __syncthreads();
  }
  const double scale = ks.type == 0 ? -1. / (2. * ks.h * ks.h) : -1. / ks.h;
  const int gr = idr[a];
  if (gr < 0) return;
#pragma unroll
  for (int b = 0; b < 16; b++) {
    const int gc = idc[bq * 16 + b];
    if (gc < 0) continue;
    const double v = exp(acc[b] * scale) + (gr == gc ? ks.lambda : 0.);
    hssk_gstore(p.out, (size_t)(a0 + a) + (size_t)(b0 + bq * 16 + b) * p.ldo, v);
  }
}

// ANOVA (type 2), degree p <= 8: one output per thread, coordinates straight from L2
__global__ __launch_bounds__(256) void kernel_eval_anova_kernel(hssk_kernel_spec ks, const hssk_keval_desc* __restrict__ descs,
                                                                const KTile* __restrict__ tiles) {
  const KTile t = tiles[blockIdx.x];
  const hssk_keval_desc p = descs[t.prob];
  const int P = ks.p;
  for (int e = threadIdx.x; e < KE_T * KE_T; e += 256) {
    const int a = t.tr * KE_T + (e & 63), b = t.tc * KE_T + (e >> 6);
    if (a >= p.nr || b >= p.nc) continue;
    const int gr = row_id(p, a), gc = col_id(p, b);
    double Kss[8], Kpp[9];
    for (int j = 0; j < P; j++) Kss[j] = 0.;
    for (int i = 0; i < ks.d; i++) {
      const double df = hssk_gload(ks.X, (size_t)gr * ks.d + i) - hssk_gload(ks.X, (size_t)gc * ks.d + i);
      const double tmp = exp(-(df * df) / (2. * ks.h * ks.h));
      double pw = tmp;
      for (int j = 0; j < P; j++) { Kss[j] += pw; pw *= tmp; }
    }
    Kpp[0] = 1.;
    for (int i = 1; i <= P; i++) {
      double s = 0.;
      for (int q = 1; q <= i; q++) s += ((q & 1) ? 1. : -1.) * Kpp[i - q] * Kss[q - 1];
      Kpp[i] = s / i;
    }
    hssk_gstore(p.out, (size_t)a + (size_t)b * p.ldo, Kpp[P] + (gr == gc ? ks.lambda : 0.));
  }
}

// ---------------------------------------------------------------------------------------------
// exact k nearest neighbours (Euclidean), one thread per query, pages of KNN_P neighbours:
// page q holds the KNN_P smallest keys (distance, index) above the largest key of page q-1.
// ---------------------------------------------------------------------------------------------
constexpr int KNN_P = 64;    // neighbours per page

constexpr int KNN_C = 64;    // candidates per LDS tile
constexpr int KNN_DMAX = 64; // largest point dimension

// DM >= d: coordinates per point, zero padded (instantiated for 8 / 16 / 32 / 64).  The query point sits in
// registers, the candidate tile in LDS as [candidate][coordinate] so that a wave reads one candidate's coordinates
// as broadcast 16-byte loads; four candidates are evaluated per trip to keep several loads in flight, the
// (rare) insertion into the LDS-resident page of the KNN_P best keys happens afterwards.
// Q queries per workgroup: Q / 64 waves share one candidate tile and -- unlike single-wave workgroups, which the
// dispatcher packed onto one SIMD of a CU -- spread over the CU's four SIMDs.
template <int DM, int Q>
__global__ __launch_bounds__(Q) void knn_kernel(const double* __restrict__ X, int d, int n, int q0, int q1, int kpage,
                                                    const float* __restrict__ lb_key, const int* __restrict__ lb_idx,
                                                    int* __restrict__ out_idx, int ldo, float* __restrict__ ub_key,
                                                    int* __restrict__ ub_idx) {
  HSSK_SHARED float hk[KNN_P * Q];
  HSSK_SHARED int hi[KNN_P * Q];
  HSSK_SHARED double xc[KNN_C * DM];
  const int tid = threadIdx.x;
  const int q = q0 + blockIdx.x * Q + tid;
  const bool live = q < q1;
  double xq[DM];
#pragma unroll
  for (int j = 0; j < DM; j++) xq[j] = (live && j < d) ? X[(size_t)q * d + j] : 0.;
  for (int s = 0; s < kpage; s++) { hk[s * Q + tid] = 3.0e38f; hi[s * Q + tid] = 0x7fffffff; }
  const float lbk = lb_key ? (live ? lb_key[q] : 0.f) : -1.f;
  const int lbi = lb_idx ? (live ? lb_idx[q] : 0) : -1;
  // the page is a binary max-heap on (key, id) in this lane's LDS column: the root is the worst kept key, an
  // insertion replaces it and sifts down (<= 6 levels for 64 entries) instead of re-scanning the page
  // (measured at N = 1e5, k = 64: 103 -> 54 ms)
  float worst = 3.0e38f;
  int worst_i = 0x7fffffff;
  auto greater = [](float ka, int ia, float kb, int ib) { return ka > kb || (ka == kb && ia > ib); };
  auto consider = [&](float key, int g) {
    const bool above = key > lbk || (key == lbk && g > lbi);
    const bool better = key < worst || (key == worst && g < worst_i);
    if (live && g != q && g < n && above && better) {
      int pos = 0;
      for (;;) {
        const int l = 2 * pos + 1, r = l + 1;
        if (l >= kpage) break;
        float kc = hk[l * Q + tid];
        int ic = hi[l * Q + tid], c = l;
        if (r < kpage) {
          const float kr = hk[r * Q + tid];
          const int ir = hi[r * Q + tid];
          if (greater(kr, ir, kc, ic)) { kc = kr; ic = ir; c = r; }
        }
        if (!greater(kc, ic, key, g)) break;
        hk[pos * Q + tid] = kc;
        hi[pos * Q + tid] = ic;
        pos = c;
      }
      hk[pos * Q + tid] = key;
      hi[pos * Q + tid] = g;
      worst = hk[tid];
      worst_i = hi[tid];
    }
  };
  // candidate tiles are visited starting at the queries' own position: after the clustering, index neighbours are
  // spatial neighbours, the page threshold tightens at once and later tiles rarely insert (same result set)
  const int ntile = (n + KNN_C - 1) / KNN_C, own = (q0 + blockIdx.x * Q) / KNN_C;
  for (int t = 0; t < ntile; t++) {
    // own, own+1, own-1, own+2, own-2, ...: in cluster order index distance tracks spatial distance
    const int off = (t + 1) >> 1;
    const int c0 = (((t & 1) ? own + off : own - off + ntile) % ntile) * KNN_C;
    __syncthreads();
    // tile load: DM independent loads per lane, all in flight together (clamped address + select instead of a
    // branch per element, which would serialise the round trips)
    {
      constexpr int NL = (KNN_C * DM + Q - 1) / Q;
      double v[NL];
#pragma unroll
      for (int r = 0; r < NL; r++) {
        const int e = tid + Q * r, pt = e / DM, j = e % DM;
        const int gp = min(c0 + pt, n - 1), gj = min(j, d - 1);
        v[r] = hssk_gload(X, (size_t)gp * d + gj);
      }
#pragma unroll
      for (int r = 0; r < NL; r++) {
        const int e = tid + Q * r, pt = e / DM, j = e % DM;
        if (e < KNN_C * DM) xc[e] = (c0 + pt < n && j < d) ? v[r] : 0.;
      }
    }
    __syncthreads();
    for (int c = 0; c < KNN_C; c += 4) {
      double s2[4] = {0., 0., 0., 0.};
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int j = 0; j < DM; j++) {
          const double df = xq[j] - xc[(c + u) * DM + j];
          s2[u] += df * df;
        }
      // branch-free acceptance test for the four candidates, ONE branch per trip into the (rare) insertion path:
      // taken branches cost an instruction-buffer refill each, which dominated this loop
      float key[4];
      bool pass[4], any = false;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        key[u] = (float)s2[u];
        const int g = c0 + c + u;
        pass[u] = live & (g != q) & (g < n) & ((key[u] > lbk) | ((key[u] == lbk) & (g > lbi))) &
                  ((key[u] < worst) | ((key[u] == worst) & (g < worst_i)));
        any |= pass[u];
      }
      if (any) {
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (pass[u]) consider(key[u], c0 + c + u);
      }
    }
  }
  if (!live) return;
  for (int s = 0; s < kpage; s++) {
    const int g = hi[s * Q + tid];
    out_idx[(size_t)q * ldo + s] = g == 0x7fffffff ? -1 : g;
  }
  if (ub_key) { ub_key[q] = worst; ub_idx[q] = worst_i; }
}

// prediction[c] = sum_r w[r] k(x_r, t_c)   (no lambda: train and test points are different sets)
constexpr int PR_T = 64;
__global__ __launch_bounds__(PR_T) void kernel_predict_kernel(hssk_kernel_spec ks, const double* __restrict__ w,
                                                              const double* __restrict__ T, int m,
                                                              double* __restrict__ pred) {
  HSSK_SHARED double xt[PR_T * (KNN_DMAX + 1)];
  HSSK_SHARED double xr[PR_T * (KNN_DMAX + 1)];
  HSSK_SHARED double wr[PR_T];
  const int tid = threadIdx.x, c = blockIdx.x * PR_T + tid, d = ks.d;
  const bool live = c < m;
  for (int j = 0; j < d; j++) xt[tid * (KNN_DMAX + 1) + j] = live ? T[(size_t)c * d + j] : 0.;
  double sum = 0.;
  for (long long r0 = 0; r0 < ks.n; r0 += PR_T) {
    __syncthreads();
    for (int e = tid; e < PR_T * d; e += PR_T) {
      const int pt = e / d, j = e % d;
      xr[pt * (KNN_DMAX + 1) + j] = r0 + pt < ks.n ? ks.X[(size_t)(r0 + pt) * d + j] : 0.;
    }
    wr[tid] = r0 + tid < ks.n ? w[r0 + tid] : 0.;
    __syncthreads();
    const int rend = (int)min((long long)PR_T, ks.n - r0);
    for (int r = 0; r < rend; r++) {
      double v;
      if (ks.type == 2) {
        double Kss[8], Kpp[9];
        for (int j = 0; j < ks.p; j++) Kss[j] = 0.;
        for (int i = 0; i < d; i++) {
          const double df = xr[r * (KNN_DMAX + 1) + i] - xt[tid * (KNN_DMAX + 1) + i];
          const double tmp = exp(-(df * df) / (2. * ks.h * ks.h));
          double pw = tmp;
          for (int j = 0; j < ks.p; j++) { Kss[j] += pw; pw *= tmp; }
        }
        Kpp[0] = 1.;
        for (int i = 1; i <= ks.p; i++) {
          double s = 0.;
          for (int q = 1; q <= i; q++) s += ((q & 1) ? 1. : -1.) * Kpp[i - q] * Kss[q - 1];
          Kpp[i] = s / i;
        }
        v = Kpp[ks.p];
      } else {
        double acc = 0.;
        for (int i = 0; i < d; i++) {
          const double df = xr[r * (KNN_DMAX + 1) + i] - xt[tid * (KNN_DMAX + 1) + i];
          acc += ks.type == 0 ? df * df : fabs(df);
        }
        v = exp(acc * (ks.type == 0 ? -1. / (2. * ks.h * ks.h) : -1. / ks.h));
      }
      sum += wr[r] * v;
    }
  }
  if (live) pred[c] = sum;
}

void check_spec(const hssk_kernel_spec& ks) {
  if (ks.type < 0 || ks.type > 2) throw std::invalid_argument("hssk kernel: type must be 0 (Gauss), 1 (Laplace) or 2 (ANOVA)");
  if (ks.d <= 0 || ks.n < 0 || !ks.X) throw std::invalid_argument("hssk kernel: bad point set");
  if (ks.type == 2 && (ks.p < 1 || ks.p > 8 || ks.p > ks.d)) throw std::invalid_argument("hssk kernel: ANOVA degree must be in [1, min(8, d)]");
}

}  // namespace

extern "C" int hssk_kernel_eval_vbatched(hssk_ctx* ctx, const hssk_kernel_spec* spec, const hssk_keval_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  check_spec(*spec);
  std::vector<KTile> tiles;
  for (int p = 0; p < count; p++)
    for (int tc = 0; tc * KE_T < descs[p].nc; tc++)
      for (int tr = 0; tr * KE_T < descs[p].nr; tr++) tiles.push_back(KTile{p, tr, tc});
  if (tiles.empty()) return 0;
  auto* dd = (const hssk_keval_desc*)ctx->stage(descs, sizeof(*descs) * count);
  auto* dt = (const KTile*)ctx->stage(tiles.data(), sizeof(KTile) * tiles.size());
  if (spec->type == 2)
    HSSK_LAUNCH(kernel_eval_anova_kernel, dim3((unsigned)tiles.size()), dim3(256), 0, ctx->stream, *spec, dd, dt);
  else
    HSSK_LAUNCH(kernel_eval_kernel, dim3((unsigned)tiles.size()), dim3(256), 0, ctx->stream, *spec, dd, dt);
  hssk_rt::check_launch();
  HSSK_API_END
}

extern "C" int hssk_knn(hssk_ctx* ctx, const double* X, int d, int n, int k, int q0, int q1, int* out_idx) {
  HSSK_API_BEGIN
  if (n <= 0 || k <= 0 || q1 <= q0) return 0;
  if (q0 < 0 || q1 > n) throw std::invalid_argument("hssk_knn: query range outside the point set");
  if (d <= 0 || d > KNN_DMAX) throw std::invalid_argument("hssk_knn: point dimension must be in [1, 64]");
  const int pages = (k + KNN_P - 1) / KNN_P;
  // page bounds (float key + index per query), ping-pong
  float* kb = (float*)ctx->scratch(sizeof(float) * 4 * (size_t)n + 64);
  int* ib = (int*)(kb + 2 * (size_t)n);

  for (int pg = 0; pg < pages; pg++) {
    const int kp = std::min(KNN_P, k - pg * KNN_P);
    const float* lk = pg ? kb + (size_t)((pg - 1) & 1) * n : nullptr;
    const int* li = pg ? ib + (size_t)((pg - 1) & 1) * n : nullptr;
    float* uk = kb + (size_t)(pg & 1) * n;
    int* ui = ib + (size_t)(pg & 1) * n;
    int* oi = out_idx + pg * KNN_P;
    if (d <= 8) HSSK_LAUNCH((knn_kernel<8, 256>), dim3((unsigned)((q1 - q0 + 256 - 1) / 256)), dim3(256), 0, ctx->stream, X, d, n, q0, q1, kp, lk, li, oi, k, uk, ui);
    else if (d <= 16) HSSK_LAUNCH((knn_kernel<16, 256>), dim3((unsigned)((q1 - q0 + 256 - 1) / 256)), dim3(256), 0, ctx->stream, X, d, n, q0, q1, kp, lk, li, oi, k, uk, ui);
    else if (d <= 32) HSSK_LAUNCH((knn_kernel<32, 256>), dim3((unsigned)((q1 - q0 + 256 - 1) / 256)), dim3(256), 0, ctx->stream, X, d, n, q0, q1, kp, lk, li, oi, k, uk, ui);
    else HSSK_LAUNCH((knn_kernel<64, 128>), dim3((unsigned)((q1 - q0 + 128 - 1) / 128)), dim3(128), 0, ctx->stream, X, d, n, q0, q1, kp, lk, li, oi, k, uk, ui);
  }
  hssk_rt::check_launch();
  HSSK_API_END
}

extern "C" int hssk_kernel_predict(hssk_ctx* ctx, const hssk_kernel_spec* spec, const double* w, const double* T, int m,
                                   double* pred) {
  HSSK_API_BEGIN
  if (m <= 0) return 0;
  check_spec(*spec);
  if (spec->d > KNN_DMAX) throw std::invalid_argument("hssk_kernel_predict: point dimension must be <= 64");
  HSSK_LAUNCH(kernel_predict_kernel, dim3((unsigned)((m + PR_T - 1) / PR_T)), dim3(PR_T), 0, ctx->stream, *spec, w, T, m, pred);
  hssk_rt::check_launch();
  HSSK_API_END
}
