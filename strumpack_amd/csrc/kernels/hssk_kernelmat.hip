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
// Bounds: kernel_eval is exp-throughput / HBM-write bound (one exp per output double); the heap form of the neighbour search
// is LDS-bandwidth bound (two LDS reads per squared difference), the filtered form (large point sets, round 6) runs its n^2
// part on the FP32 matrix cores.
#include "hssk_device.h"
#include "hssk_internal.h"

#include <algorithm>
#include <cstdio>
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

constexpr int KNN_TILE = 512; // coordinates per LDS tile (512 / DM candidates)
constexpr int KNN_DMAX = 64; // largest point dimension

// DM >= d: coordinates per point, zero padded (instantiated for 8 / 16 / 32 / 64).  The query point sits in
// registers, the candidate tile in LDS as [candidate][coordinate] so that a wave reads one candidate's coordinates
// as broadcast 16-byte loads; four candidates are evaluated per trip to keep several loads in flight.
// Q queries per workgroup: Q / 64 waves share one candidate tile and -- unlike single-wave workgroups, which the
// dispatcher packed onto one SIMD of a CU -- spread over the CU's four SIMDs.
//
// Keys: (float distance^2, index) packed into one 64-bit word, distance bits high -- distances are >= 0, so the
// unsigned order of the words IS the (distance, index) order; the acceptance test of a candidate is two 64-bit
// compares.  The page of a query is a binary max-heap of such words in the lane's LDS column.
//
// Insertions are BATCHED.  A candidate that passes the test is appended to the lane's pending list (KNN_PEND slots
// in LDS: an unconditional store + a conditional increment, no branch); the heap insertions run for the whole wave
// when some lane's list is nearly full.  Per query an insertion is rare (~300 in 1e5 candidates), per WAVE it is not:
// with 64 queries side by side every second trip of four candidates had some lane inserting, and the wave then ran the
// sift-down loop with one or two lanes active (PMC of the first version at N = 1e5, profiles/r02_pmc_knn.md: VALU busy
// 25 %, scalar 18 %, 1.4 us per trip against ~0.3 us of arithmetic).  Batched, the same loop runs with tens of lanes
// active and ~10 x less often; the threshold of a lane (its page's worst key) is only tightened at a flush, which lets a
// few more candidates through to the list -- they are tested again against the current threshold there.
constexpr int KNN_PEND = 8;
constexpr int KNN_JS = 2;    // coordinates per scheduling group of the distance loop
typedef unsigned long long knn_key_t;
__device__ __forceinline__ knn_key_t knn_pack(float key, int idx) { return ((knn_key_t)hssk_fbits(key) << 32) | (unsigned)idx; }
constexpr knn_key_t KNN_EMPTY = ((knn_key_t)0x7f61b1e6u << 32) | 0x7fffffffu;   // (3.0e38f, INT_MAX): above every real key

// LQ lanes per query: each of them takes every LQ-th candidate of a trip and keeps its own pending list; the page (heap) of
// the query is shared.  LQ = 1 is the layout described above.  LQ = 4 quarters the LDS footprint per wave (16 queries: 8 KB
// of heaps), so several waves share a SIMD and cover each other's LDS and FP64 latencies -- with one query per lane the
// 128 KB of heaps per 256 queries left ONE wave per SIMD, which issues one instruction per 4-cycle slot whatever its type
// (profiles/r02_pmc_knn.md).  The price is paid in the flush, which takes the LQ lane groups of a query in turn.
// TILEK / PEND: coordinates per LDS tile and lane, slots of a lane's pending list (the defaults above).  <8, 64, 4, false, 64,
// 11> is a ONE-WAVE workgroup of 17.5 KB: nine of them per CU, and the 6250 of N = 1e5 are 2.7 rounds of the chip's 2304
// slots -- the 1563 four-wave workgroups were 3.05 rounds of 512 slots, i.e. FOUR rounds of time for three of work.
template <int DM, int Q, int LQ, bool PIPE = false, int TILEK = KNN_TILE, int PEND = KNN_PEND>
__global__ __launch_bounds__(Q) void knn_kernel(const double* __restrict__ X, int d, int n, int q0, int q1, int kpage,
                                                    const knn_key_t* __restrict__ lb, int* __restrict__ out_idx, int ldo,
                                                    knn_key_t* __restrict__ ub, int window) {
  constexpr int TILE = TILEK * LQ;              // coordinates per tile: a tile lasts as many trips whatever LQ
  constexpr int CT = TILE / DM;                 // candidates per tile
  constexpr int U = DM <= 16 ? 4 : (DM <= 32 ? 2 : 1);   // candidates per lane and trip (their coordinates sit in registers)
  constexpr int NL = TILE / Q;                  // tile elements per lane
  constexpr int NQ = Q / LQ;                    // queries per workgroup
  constexpr int TRIP = U * LQ;                  // candidates per trip
  static_assert(TILE % Q == 0 && CT % (2 * TRIP) == 0 && Q % LQ == 0 && 64 % LQ == 0, "tile shape");
  HSSK_SHARED knn_key_t hh[KNN_P * NQ];
  HSSK_SHARED knn_key_t pend[PEND * Q];
  HSSK_SHARED double xc[2 * TILE];          // two tiles: the next one is written while the current one is read
  const int tid = threadIdx.x, ql = tid / LQ, part = tid % LQ;
  const int q = q0 + blockIdx.x * NQ + ql;
  const bool live = q < q1;
  double xq[DM];
#pragma unroll
  for (int j = 0; j < DM; j++) xq[j] = (live && j < d) ? X[(size_t)q * d + j] : 0.;
  for (int s = part; s < kpage; s += LQ) hh[s * NQ + ql] = KNN_EMPTY;
  // keys of this page lie in [lo, worst): lo = the previous page's largest key + 1; a lane without a query accepts nothing
  const knn_key_t lo = (lb && live) ? lb[q] + 1 : 0;
  knn_key_t worst = live ? KNN_EMPTY : 0;
  int cnt = 0;
  // replaces the root of the heap (the worst kept key) by K and sifts it down (<= 6 levels for 64 entries)
  auto insert = [&](knn_key_t K) {
    int pos = 0;
    for (;;) {
      const int l = 2 * pos + 1, r = l + 1;
      if (l >= kpage) break;
      knn_key_t kc = hh[l * NQ + ql];
      int c = l;
      if (r < kpage) {
        const knn_key_t kr = hh[r * NQ + ql];
        if (kr > kc) { kc = kr; c = r; }
      }
      if (kc <= K) break;
      hh[pos * NQ + ql] = kc;
      pos = c;
    }
    hh[pos * NQ + ql] = K;
    worst = hh[ql];
  };
  // the lanes of a query take turns (they sit in one wave: LDS operations of a wave execute in order, so a turn sees the
  // heap its predecessor left); within a turn all queries of the wave insert side by side
  auto flush = [&]() {
#pragma unroll
    for (int turn = 0; turn < LQ; turn++) {
      if (LQ > 1 && live) worst = hh[ql];
      for (int s = 0; hssk_any(part == turn && s < cnt); s++)
        if (part == turn && s < cnt) {
          const knn_key_t K = pend[s * Q + tid];
          if (K < worst) insert(K);
        }
    }
    if (LQ > 1 && live) worst = hh[ql];
    cnt = 0;
  };
  if (LQ > 1) __syncthreads();   // (the heaps are initialised by all lanes of a query)
  // candidate tiles are visited starting at the queries' own position: after the clustering, index neighbours are
  // spatial neighbours, the page threshold tightens at once and later tiles rarely insert (same result set)
  // window > 0: only the tiles nearest to the queries' own (about `window` candidates) are visited -- the page bound `ub` is
  // then an UPPER bound of the true one (the filtered search below starts from it); out_idx may be null
  const int ntile = (n + CT - 1) / CT, own = (q0 + blockIdx.x * NQ) / CT;
  const int nvis = window > 0 ? min(ntile, (window + CT - 1) / CT) : ntile;
  const double inf = __builtin_huge_val();
  // own, own+1, own-1, own+2, own-2, ...: in cluster order index distance tracks spatial distance
  auto tile_start = [&](int t) {
    const int off = (t + 1) >> 1;
    return (((t & 1) ? own + off : own - off + ntile) % ntile) * CT;
  };
  // tile t's coordinates: NL independent global loads per lane (clamped address + select instead of a branch per
  // element, which would serialise the round trips), issued one tile ahead of their use.  Candidates past the end
  // of the point set get an infinite first coordinate: their key is above every threshold.
  double v[NL];
  auto tile_fetch = [&](int t) {
    const int c0 = tile_start(t);
#pragma unroll
    for (int r = 0; r < NL; r++) {
      const int e = tid + Q * r, pt = e / DM, j = e % DM;
      const int gp = min(c0 + pt, n - 1), gj = min(j, d - 1);
      const double x = hssk_gload(X, (size_t)gp * d + gj);
      v[r] = c0 + pt < n ? (j < d ? x : 0.) : (j == 0 ? inf : 0.);
    }
  };
  tile_fetch(0);
  for (int t = 0; t < nvis; t++) {
    double* tile = xc + (t & 1) * TILE;
#pragma unroll
    for (int r = 0; r < NL; r++) tile[tid + Q * r] = v[r];
    // one barrier per tile: the buffer written here was last read two tiles ago, before the previous barrier
    __syncthreads();
    if (t + 1 < nvis) tile_fetch(t + 1);
    const int c0 = tile_start(t);
    // a trip: U candidates per lane (candidate c + u LQ + part: the LQ lanes of a query read neighbouring candidates, whose
    // 64-byte records fall on different LDS banks) against the lane's query, their keys appended to the pending list
    auto trip = [&](int c, const double (&b)[U][DM]) {
      // coordinate-major: the U chains (one per candidate) advance side by side, so consecutive instructions are independent
      double s2[U];
#pragma unroll
      for (int u = 0; u < U; u++) s2[u] = 0.;
#pragma unroll
      for (int j = 0; j < DM; j += KNN_JS) {
        double df[KNN_JS][U];
#pragma unroll
        for (int jj = 0; jj < KNN_JS; jj++)
#pragma unroll
          for (int u = 0; u < U; u++) df[jj][u] = xq[j + jj] - b[u][j + jj];
        hssk_sched_barrier();   // (the scheduler otherwise pairs every subtraction with its dependent fma)
#pragma unroll
        for (int jj = 0; jj < KNN_JS; jj++) {
#pragma unroll
          for (int u = 0; u < U; u++) s2[u] += df[jj][u] * df[jj][u];
        }
        hssk_sched_barrier();
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int g = c0 + c + u * LQ + part;
        const knn_key_t K = knn_pack((float)s2[u], g);
        pend[cnt * Q + tid] = K;   // kept only if the candidate passes (the slot is overwritten otherwise)
        cnt += (int)((K >= lo) & (K < worst) & (g != q));
      }
    };
    auto fetch = [&](int c, double (&b)[U][DM]) {
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int j = 0; j < DM; j++) b[u][j] = tile[(c + u * LQ + part) * DM + j];
    };
    if (LQ == 1 || PIPE) {
      // software pipeline over the tile: the LDS reads of the next trip are in flight while this one computes (one
      // wave per SIMD: nothing else hides their latency)
      double b0[U][DM], b1[U][DM];
      fetch(0, b0);
      for (int c = 0; c < CT; c += 2 * TRIP) {
        fetch(c + TRIP, b1);
        trip(c, b0);
        if (hssk_any(cnt > PEND - U)) flush();
        if (c + 2 * TRIP < CT) fetch(c + 2 * TRIP, b0);
        trip(c + TRIP, b1);
        if (hssk_any(cnt > PEND - U)) flush();
      }
    } else {
      // (several waves per SIMD: they cover each other.  With d <= 8 the LDS holds two workgroups per CU whatever the register
      // count, so the second register set is free there: PIPE, 18.9 -> 18.2 ms at N = 1e5.  Tried and dropped in round 4: the
      // page as an unsorted array with its largest key in registers and the four lanes of a query scanning for the next one
      // together -- 21.0 ms: every lane of the wave runs every insertion step of every query, a scan costs more than the six
      // levels of the heap it replaces)
      double b0[U][DM];
      for (int c = 0; c < CT; c += TRIP) {
        fetch(c, b0);
        trip(c, b0);
        if (hssk_any(cnt > PEND - U)) flush();
      }
    }
  }
  flush();
  if (!live) return;
  if (out_idx)
    for (int s = part; s < kpage; s += LQ) {
      const knn_key_t K = hh[s * NQ + ql];
      out_idx[(size_t)q * ldo + s] = K == KNN_EMPTY ? -1 : (int)(K & 0xffffffffu);
    }
  if (ub && part == 0) ub[q] = worst;
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

// ---- column sets: sorted unions through a bitmap in the LDS (hssk_colsets) ----------------------------------------
namespace {
constexpr int CS_T = 1024;
__global__ __launch_bounds__(CS_T) void colset_kernel(const hssk_colset_desc* __restrict__ descs, int words) {
  HSSK_DYN_SHARED(unsigned, cs_lds);
  unsigned* bits = cs_lds;                 // [words]
  int* base = (int*)(cs_lds + words);      // [CS_T + 1]: ids before each thread's stretch of the bitmap
  const hssk_colset_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x;
  for (int w = tid; w < words; w += CS_T) bits[w] = 0u;
  __syncthreads();
  for (int s = 0; s < 2; s++) {
    const int* src = s ? p.src1 : p.src0;
    const int* nd = s ? p.n1_dev : p.n0_dev;
    const int n = nd ? min(*nd, s ? p.n1 : p.n0) : (s ? p.n1 : p.n0);   // (a count an earlier launch left on the device)
    if (!src) continue;
    for (int e = tid; e < n; e += CS_T) {
      const int g = src[e];
      if (g >= 0 && (g < p.lo || g >= p.hi)) hssk_lds_or(&bits[g >> 5], 1u << (g & 31));
    }
  }
  __syncthreads();
  // every thread a contiguous stretch of words: count, exclusive scan over the threads, write in order
  const int per = (words + CS_T - 1) / CS_T, w0 = min(words, tid * per), w1 = min(words, w0 + per);
  int cnt = 0;
  for (int w = w0; w < w1; w++) cnt += __builtin_popcount(bits[w]);
  base[tid + 1] = cnt;
  if (tid == 0) base[0] = 0;
  __syncthreads();
  // (inclusive scan, Hillis-Steele over CS_T + 1 entries; the slots are re-read after a barrier each round)
  for (int off = 1; off <= CS_T; off <<= 1) {
    const int add = tid + 1 >= off ? base[tid + 1 - off] : 0;
    __syncthreads();
    base[tid + 1] += add;
    __syncthreads();
  }
  int o = base[tid];
  for (int w = w0; w < w1; w++) {
    unsigned b = bits[w];
    while (b) {
      const int t = __builtin_ctz(b);
      p.out[o++] = (w << 5) + t;
      b &= b - 1;
    }
  }
  if (tid == CS_T - 1) *p.count = base[CS_T];
}
}  // namespace

extern "C" long long hssk_colsets_max_universe(void) {
  const size_t cap = hssk_rt::max_lds_per_workgroup(), fixed = sizeof(int) * (CS_T + 1);
  return cap > fixed ? (long long)((cap - fixed) / sizeof(unsigned)) * 32 : 0;
}
extern "C" int hssk_colsets(hssk_ctx* ctx, const hssk_colset_desc* descs, int count, int universe) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  if (universe <= 0) throw std::invalid_argument("hssk_colsets: empty universe");
  const int words = (universe + 31) / 32;
  const size_t shm = sizeof(unsigned) * (size_t)words + sizeof(int) * (CS_T + 1);
  if (shm > hssk_rt::max_lds_per_workgroup()) HSSK_UNSUPPORTED("column sets over more ids than the LDS has bits");
  auto* dd = (const hssk_colset_desc*)ctx->stage(descs, sizeof(*descs) * count);
  hssk_rt::allow_dynamic_lds(colset_kernel, shm);
  HSSK_LAUNCH(colset_kernel, dim3((unsigned)count), dim3(CS_T), shm, ctx->stream, dd, words);
  hssk_rt::check_launch();
  HSSK_API_END
}

namespace {
// All pages of the search by the heap kernel.  window > 0: the search over the ~window candidates nearest in index to each
// query, nothing written but the last page's bound (returned: n keys, indexed by point).
knn_key_t* knn_exhaustive(hssk_ctx* ctx, const double* X, int d, int n, int k, int q0, int q1, int* out_idx, int window) {
  const int pages = (k + KNN_P - 1) / KNN_P;
  // page bounds (the largest key of the previous page, per query), ping-pong
  knn_key_t* kb = (knn_key_t*)ctx->scratch(sizeof(knn_key_t) * 2 * (size_t)n + 64);
  for (int pg = 0; pg < pages; pg++) {
    const int kp = std::min(KNN_P, k - pg * KNN_P);
    const knn_key_t* lb = pg ? kb + (size_t)((pg - 1) & 1) * n : nullptr;
    knn_key_t* ub = kb + (size_t)(pg & 1) * n;
    int* oi = out_idx ? out_idx + pg * KNN_P : nullptr;
    // (queries per workgroup: threads / lanes per query)
    static const bool lq1 = [] { const char* e = std::getenv("HSSK_KNN_LQ"); return e && e[0] == '1'; }();
    const int nqr = q1 - q0;
    static const bool pipe = [] { const char* e = std::getenv("HSSK_KNN_PIPE"); return !(e && e[0] == '0'); }();
    static const bool w1 = [] { const char* e = std::getenv("HSSK_KNN_W1"); return !(e && e[0] == '0'); }();   // (18.2 -> 17.3 ms at N = 1e5)
    if (d <= 8 && !lq1 && w1) {
      HSSK_LAUNCH((knn_kernel<8, 64, 4, false, 64, 11>), dim3((unsigned)((nqr + 15) / 16)), dim3(64), 0, ctx->stream, X, d, n, q0, q1, kp, lb, oi, k, ub, window);
      continue;
    }
    if (d <= 8 && !lq1 && pipe) HSSK_LAUNCH((knn_kernel<8, 256, 4, true>), dim3((unsigned)((nqr + 63) / 64)), dim3(256), 0, ctx->stream, X, d, n, q0, q1, kp, lb, oi, k, ub, window);
    else if (d <= 8 && !lq1) HSSK_LAUNCH((knn_kernel<8, 256, 4>), dim3((unsigned)((nqr + 63) / 64)), dim3(256), 0, ctx->stream, X, d, n, q0, q1, kp, lb, oi, k, ub, window);
    else if (d <= 8) HSSK_LAUNCH((knn_kernel<8, 256, 1>), dim3((unsigned)((nqr + 255) / 256)), dim3(256), 0, ctx->stream, X, d, n, q0, q1, kp, lb, oi, k, ub, window);
    else if (d <= 16) HSSK_LAUNCH((knn_kernel<16, 256, 4>), dim3((unsigned)((nqr + 63) / 64)), dim3(256), 0, ctx->stream, X, d, n, q0, q1, kp, lb, oi, k, ub, window);
    else if (d <= 32) HSSK_LAUNCH((knn_kernel<32, 128, 1>), dim3((unsigned)((nqr + 127) / 128)), dim3(128), 0, ctx->stream, X, d, n, q0, q1, kp, lb, oi, k, ub, window);
    else HSSK_LAUNCH((knn_kernel<64, 128, 1>), dim3((unsigned)((nqr + 127) / 128)), dim3(128), 0, ctx->stream, X, d, n, q0, q1, kp, lb, oi, k, ub, window);
  }
  hssk_rt::check_launch();
  return kb + (size_t)((pages - 1) & 1) * n;
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// The filtered search (large point sets).  The heap kernel above spends its time on the per-query heaps, not on the n^2
// distances.  Here the n^2 part only FILTERS:
//   * d2(q, c) = |c|^2 + |q|^2 - 2 c.q of all pairs on the FP32 matrix cores (v_mfma_f32_32x32x2_f32: 32 candidates along the
//     rows, a query per lane, K = d + 2 with the two norms as extra coordinates; points centred on their mean).  A pair
//     passes when its approximate d2 is <= tau_q + the worst-case error of that arithmetic, tau_q being the float key of the
//     k-th best candidate the query has met so far -- no true neighbour is lost; the id of a pair that passes is appended to
//     the query's list (a private cursor per lane: no atomics);
//   * a list that is nearly full is COMPACTED by the whole wave: exact keys of its entries -- the heap kernel's arithmetic:
//     FP64 differences, key = (float(d2), id) --, the k smallest kept (bisection on the key bits with ballot counts), tau_q
//     tightened.  Candidate tiles are visited outwards from the queries' own position (cluster order: index neighbours are
//     spatial neighbours), so tau_q is close to its final value after the first few tiles and a query lists a few hundred of
//     the n candidates; lists cannot overflow whatever the data (a loose tau_q only costs compactions);
//   * a last compaction writes the k ids.
// The result is the heap kernel's set (the k smallest keys are unique).  64 queries per wave, four independent waves per workgroup.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int K2_G = 64;         // workgroups of the mean / norm reductions
constexpr int K2_SLOTS = 6;      // entries per lane at a compaction: 64 x 6 = 384 = 2 halves x 128 listed + 128 kept

// partial sums of the coordinates: workgroup g adds the points g, g + G, ... (thread t always meets coordinate t % d)
__global__ __launch_bounds__(256) void knn2_mean_kernel(const double* __restrict__ X, int d, int n, double* __restrict__ part) {
  HSSK_SHARED double red[256];
  const int tid = threadIdx.x, Tp = (256 / d) * d, ppw = Tp / d;   // points per pass of the workgroup
  double s = 0.;
  if (tid < Tp) {
    const int j = tid % d, pl = tid / d;
    for (long long i = (long long)blockIdx.x * ppw + pl; i < n; i += (long long)K2_G * ppw) s += X[(size_t)i * d + j];
  }
  red[tid] = tid < Tp ? s : 0.;
  __syncthreads();
  if (tid < d) {
    double a = 0.;
    for (int m = tid; m < Tp; m += d) a += red[m];
    part[blockIdx.x * d + tid] = a;
  }
}

// Cf (KP x ldc floats, row k = coordinate k of every candidate): rows 0 .. d-1 = -2 (x - mean), row d = |x - mean|^2, rows d + 1 and
// d + 2 = 1, the rest 0; candidates n .. ldc-1 (padding) get an enormous norm.  nmax[g] = largest norm of workgroup g's points.
__global__ __launch_bounds__(256) void knn2_prep_kernel(const double* __restrict__ X, int d, int n, int ldc, int KP,
                                                        const double* __restrict__ part, float* __restrict__ Cf, float* __restrict__ nmax) {
  HSSK_SHARED double mean[KNN_DMAX];
  HSSK_SHARED float red[256];
  const int tid = threadIdx.x;
  if (tid < d) {
    double a = 0.;
    for (int g = 0; g < K2_G; g++) a += part[g * d + tid];
    mean[tid] = a / n;
  }
  __syncthreads();
  float big = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + tid; i < ldc; i += (long long)gridDim.x * 256) {
    if (i < n) {
      double s2 = 0.;
      for (int j = 0; j < d; j++) {
        const double c = X[(size_t)i * d + j] - mean[j];
        s2 += c * c;
        Cf[(size_t)j * ldc + i] = (float)(-2. * c);
      }
      const float nf = (float)s2;
      Cf[(size_t)d * ldc + i] = nf;
      big = fmaxf(big, nf);
    } else {
      for (int j = 0; j < d; j++) Cf[(size_t)j * ldc + i] = 0.f;
      Cf[(size_t)d * ldc + i] = 3.0e38f;
    }
    Cf[(size_t)(d + 1) * ldc + i] = 1.f;   // meets the query's norm
    Cf[(size_t)(d + 2) * ldc + i] = 1.f;   // meets minus the query's threshold
    for (int j = d + 3; j < KP; j++) Cf[(size_t)j * ldc + i] = 0.f;
  }
  red[tid] = big;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
    __syncthreads();
  }
  if (tid == 0) nmax[blockIdx.x] = red[0];
}

// KSM k-steps of two coordinates (KP = 2 KSM >= d + 3 rows of Cf), TC candidates per LDS tile.  A wave: queries
// q0 + 64 w .. + 63 in two groups of 32 (group g, query l & 31; the lane halves l >> 5 supply the two coordinates of a k-step and
// receive different candidate rows, so each half keeps a list of its own).  Row d + 2 of the product is 1 x (-tau_q): the
// accumulator IS d2 - tau_q and its sign bit the verdict; a lane shifts the 16 sign bits of a block into one word and appends
// (block, bits) to its list when any is set -- two instructions per pair and one predicated store per lane, group and block.
constexpr int K2_CAPH = 128;     // ids a lane half lists between two compactions of its query (and the words that hold them)
template <int KSM, int TC>
__global__ __launch_bounds__(256) void knn2_scan_kernel(const double* __restrict__ X, const float* __restrict__ Cf, int ldc, int d, int n,
                                                        int q0, int q1, int k, const float* __restrict__ nmax,
                                                        unsigned* __restrict__ list, int* __restrict__ kept, int* __restrict__ out_idx, int ldo,
                                                        long long* dbg) {
  constexpr int KP = 2 * KSM, NV = (KP * TC / 4 + 63) / 64;
  HSSK_DYN_SHARED(float, lds_all);   // per wave: a tile of KP x TC floats, then 512 ids (the entries of a compaction)
  // (four independent waves per workgroup: they never meet at a barrier)
  const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31, wv = threadIdx.x >> 6, wid = blockIdx.x * 4 + wv;
  float* tile = lds_all + wv * (KP * TC + 512);
  int* ent = (int*)(tile + KP * TC);
  const int qbase = q0 + wid * 64;
  if (qbase >= q1) return;
  long long dbg_comp = 0, dbg_listed = 0;
  float bq[2][KSM], marg[2];
  int nw[2] = {0, 0}, nh[2] = {0, 0}, kc[2] = {0, 0};   // words / ids listed since the query's last compaction, ids it kept
  unsigned* seg[2];
  float big = 0.f;
  for (int g = 0; g < K2_G; g++) big = fmaxf(big, nmax[g]);
  const int trow = d + 2;   // the threshold's row: supplied by the lanes of half trow & 1 in k-step trow / 2
#pragma unroll
  for (int g = 0; g < 2; g++) {
    const int q = qbase + 32 * g + l32;
    const bool live = q < q1 && !(dbg && dbg[2] == 1);
    const float nq = live ? Cf[(size_t)d * ldc + q] : 0.f;
    // the query's side of the product: coordinate k of (x - mean) = -Cf[k] / 2, then 1 (meets the candidate's norm), its own norm,
    // and minus the threshold: nothing met yet = every candidate passes (not the padding, whose norm is 3e38); a lane without a
    // query lists nothing
#pragma unroll
    for (int s = 0; s < KSM; s++) {
      const int kk = 2 * s + half;
      bq[g][s] = kk == trow ? (live ? -1.0e38f : 1.f)
                            : (!live ? 0.f : (kk < d ? -0.5f * Cf[(size_t)kk * ldc + q] : (kk == d ? 1.f : (kk == d + 1 ? nq : 0.f))));
    }
    // what the FP32 evaluation can be off by: inputs and norms rounded to FP32, K + 2 accumulations of terms bounded by
    // (|c| + |q|)^2 <= 2 (|c|^2 + |q|^2)
    marg[g] = fmaxf(2.02f * ((float)(KP + 6) * 1.1920929e-7f) * (nq + big), 1.0e-30f);
    seg[g] = list + ((size_t)(wid * 64 + 32 * g + l32) * 2 + half) * K2_CAPH;
  }
  // ---- compaction of query j (0 .. 63, uniform) of this wave; fin: the ids go out instead of into the kept list
  auto compact = [&](int j, bool fin) {
    const int g = j >> 5, j32 = j & 31, q = qbase + j;
    const int w0 = hssk_shfl(g ? nw[1] : nw[0], j32), w1 = hssk_shfl(g ? nw[1] : nw[0], j32 + 32), nk = hssk_shfl(g ? kc[1] : kc[0], j32);
    const unsigned* s0 = list + ((size_t)(wid * 64 + j) * 2) * K2_CAPH;
    int* kq = kept + (size_t)(wid * 64 + j) * 128;
    hssk_drain_stores();
    // the listed words back into ids: ent[0 .. m)
    int m = 0;
    for (int i0 = 0; i0 < w0 + w1; i0 += 64) {
      const int i = i0 + lane;
      const unsigned w = i < w0 + w1 ? (unsigned)hssk_flag_load((const int*)(i < w0 ? s0 + i : s0 + K2_CAPH + (i - w0))) : 0u;
      const int hf = i < w0 ? 0 : 1;
      int cnt = __builtin_popcount(w & 0xffffu), pre = cnt;
      for (int off = 1; off < 64; off <<= 1) {   // inclusive prefix sum over the lanes
        const int o = hssk_shfl(pre, max(lane - off, 0));
        if (lane >= off) pre += o;
      }
      int pos = m + pre - cnt;
      unsigned bits = w & 0xffffu;
      const int cb = (int)(w >> 16) * 32 + 4 * hf;
      while (bits) {
        const int bp = 31 - __builtin_clz(bits), r = 15 - bp;   // (register r of the block went into bit 15 - r)
        ent[pos++] = cb + 8 * (r / 4) + (r % 4);
        bits &= ~(1u << bp);
      }
      m += hssk_shfl(pre, 63);
    }
    for (int i = lane; i < nk; i += 64) ent[m + i] = hssk_flag_load(kq + i);
    m += nk;
    dbg_comp++; dbg_listed += m;
    hssk_wave_sync();
    // exact keys, the heap kernel's arithmetic
    // (all slots' coordinates are fetched side by side -- an empty slot reads the query's own: a compaction is a chain of memory
    //  round trips, not arithmetic)
    knn_key_t key[K2_SLOTS];
    const int nslot = (m + 63) / 64;
    int cg[K2_SLOTS];
    double s2[K2_SLOTS];
#pragma unroll
    for (int i = 0; i < K2_SLOTS; i++) {
      const int e = lane + 64 * i;
      cg[i] = e < m ? ent[e] : q;
      s2[i] = 0.;
    }
#pragma unroll 4
    for (int c = 0; c < d; c++) {
      const double xq = X[(size_t)q * d + c];
#pragma unroll
      for (int i = 0; i < K2_SLOTS; i++) {
        const double df = xq - X[(size_t)cg[i] * d + c];
        s2[i] += df * df;
      }
    }
#pragma unroll
    for (int i = 0; i < K2_SLOTS; i++) key[i] = cg[i] != q ? knn_pack((float)s2[i], cg[i]) : KNN_EMPTY;
    hssk_wave_sync();
    // the k-th smallest key: the largest P with fewer than k keys below it, bit by bit (bit 63 is the sign of a distance: clear;
    // the bits of the low word above the largest id are clear in every key, so they are clear in P)
    knn_key_t P = 0;
    for (int b = 62; b >= 0; b--) {
      if (b == 31) b = 31 - __builtin_clz((unsigned)n);
      const knn_key_t cand = P | (1ULL << b);
      int c = 0;
#pragma unroll
      for (int i = 0; i < K2_SLOTS; i++)
        if (i < nslot) c += __builtin_popcountll(hssk_ballot(key[i] < cand));
      if (c < k) P = cand;
    }
    int o = 0;
#pragma unroll
    for (int i = 0; i < K2_SLOTS; i++)
      if (i < nslot) {
        const int keep = key[i] <= P && key[i] != KNN_EMPTY;
        const unsigned long long mk = hssk_ballot(keep);
        if (keep) {
          const int pos = o + __builtin_popcountll(mk & ((1ULL << lane) - 1ULL)), id = (int)(key[i] & 0xffffffffu);
          if (fin) out_idx[(size_t)q * ldo + pos] = id;
          else kq[pos] = id;
        }
        o += __builtin_popcountll(mk);
      }
    if (fin) {
      for (int s = o + lane; s < k; s += 64) out_idx[(size_t)q * ldo + s] = -1;
      return;
    }
    // the query's lanes: empty lists, the threshold from the k-th key (once k candidates are known)
    const float kf = hssk_from_fbits((unsigned)(P >> 32));
    if (l32 == j32) {
      if (g) { nw[1] = 0; nh[1] = 0; kc[1] = o; }
      else { nw[0] = 0; nh[0] = 0; kc[0] = o; }
      if (o >= k && half == (trow & 1)) {
        const float t = -(kf * (1.f + 4.8e-7f) + (g ? marg[1] : marg[0]));
#pragma unroll
        for (int s = 0; s < KSM; s++)
          if (s == trow / 2) { if (g) bq[1][s] = t; else bq[0][s] = t; }
      }
    }
  };
  // ---- the scan: tiles outwards from the queries' own
  const int ntile = ldc / TC, own = min(qbase + 32, n - 1) / TC;
  auto tile_of = [&](int it) {
    const int off = (it + 1) >> 1;
    return ((it & 1) ? own + off : own - off + ntile) % ntile;
  };
  hssk_f4 v[NV];
  auto gfetch = [&](int t) {
#pragma unroll
    for (int r = 0; r < NV; r++) {
      const int e = min(lane + 64 * r, KP * TC / 4 - 1), row = e / (TC / 4), c4 = e % (TC / 4);
      v[r] = *(const hssk_f4*)(Cf + (size_t)row * ldc + (size_t)t * TC + 4 * c4);
    }
  };
  gfetch(tile_of(0));
  for (int it = 0; it < ntile; it++) {
    const int t = tile_of(it);
    hssk_wave_sync();   // (the tile is this wave's alone: its reads of the previous one are issued)
#pragma unroll
    for (int r = 0; r < NV; r++)
      if (lane + 64 * r < KP * TC / 4) *(hssk_f4*)(tile + 4 * (lane + 64 * r)) = v[r];
    hssk_wave_sync();
    if (it + 1 < ntile) gfetch(tile_of(it + 1));
    for (int b = 0; b < TC / 32; b++) {
      hssk_f16v acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < KSM; s++) {
        const float a = tile[(2 * s + half) * TC + b * 32 + l32];
        acc0 = hssk_mfma_f32_32x32x2(a, bq[0][s], acc0);
        acc1 = hssk_mfma_f32_32x32x2(a, bq[1][s], acc1);
      }
      unsigned m0 = 0, m1 = 0;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        m0 = (m0 << 1) | (hssk_fbits(acc0[r]) >> 31);
        m1 = (m1 << 1) | (hssk_fbits(acc1[r]) >> 31);
      }
      const unsigned blk = (unsigned)(t * (TC / 32) + b) << 16;
      if (m0) { seg[0][nw[0]] = blk | m0; nw[0]++; nh[0] += __builtin_popcount(m0); }
      if (m1) { seg[1][nw[1]] = blk | m1; nw[1]++; nh[1] += __builtin_popcount(m1); }
      // a lane half lists at most 16 ids per block: compact what could overflow on the next one
      for (int g = 0; g < 2; g++) {
        unsigned long long need = hssk_ballot((g ? nh[1] : nh[0]) > K2_CAPH - 16);
        while (need) {
          compact(32 * g + (__builtin_ctzll(need) & 31), false);
          need = hssk_ballot((g ? nh[1] : nh[0]) > K2_CAPH - 16);
        }
      }
    }
  }
  for (int j = 0; j < 64 && qbase + j < q1; j++) compact(j, true);
  if (dbg && lane == 0) {   // (HSSK_KNN_DEBUG: compactions and the entries they met, summed over the waves)
    hssk_gadd_ll(dbg, dbg_comp);
    hssk_gadd_ll(dbg + 1, dbg_listed);
  }
}

template <int KSM, int TC>
void knn2_launch_scan(hssk_ctx* ctx, int nwaves, const double* X, const float* Cf, int ldc, int d, int n, int q0, int q1, int k,
                      const float* nmax, unsigned* list, int* kept, int* out_idx, long long* dbg) {
  const size_t shm = 4 * (sizeof(float) * (2 * KSM) * TC + sizeof(int) * 512);
  hssk_rt::allow_dynamic_lds(knn2_scan_kernel<KSM, TC>, shm);
  HSSK_LAUNCH((knn2_scan_kernel<KSM, TC>), dim3((unsigned)((nwaves + 3) / 4)), dim3(256), shm, ctx->stream, X, Cf, ldc, d, n, q0, q1, k, nmax,
              list, kept, out_idx, k, dbg);
}
}  // namespace

extern "C" int hssk_knn(hssk_ctx* ctx, const double* X, int d, int n, int k, int q0, int q1, int* out_idx) {
  HSSK_API_BEGIN
  if (n <= 0 || k <= 0 || q1 <= q0) return 0;
  if (q0 < 0 || q1 > n) throw std::invalid_argument("hssk_knn: query range outside the point set");
  if (d <= 0 || d > KNN_DMAX) throw std::invalid_argument("hssk_knn: point dimension must be in [1, 64]");
  // the filtered search pays from a few thousand points on (HSSK_KNN_FILTER_MIN; HSSK_KNN_FILTER=0: always the heap kernel);
  // its lists name a candidate by (block of 32, bit): 2^16 blocks
  static const bool filt = [] { const char* e = std::getenv("HSSK_KNN_FILTER"); return !(e && e[0] == '0'); }();
  const char* fmin_env = std::getenv("HSSK_KNN_FILTER_MIN");   // (read per call: the tests take both searches in one process)
  const int fmin_n = fmin_env ? std::atoi(fmin_env) : 8192;
  if (!filt || n < fmin_n || k > 128 || d > 29 || n <= 4 * k || n > (1 << 21) - 256) {
    knn_exhaustive(ctx, X, d, n, k, q0, q1, out_idx, 0);
    return 0;
  }
  const int KSM = d <= 3 ? 3 : (d <= 9 ? 6 : (d <= 17 ? 10 : 16)), KP = 2 * KSM, TC = 128;
  const int ldc = ((n + TC - 1) / TC) * TC, nq = q1 - q0, nw = (nq + 63) / 64;
  // scratch: Cf | partial means | norms | lists (words) | kept ids
  const size_t o_cf = 0, b_cf = sizeof(float) * (size_t)KP * ldc;
  const size_t o_pm = (o_cf + b_cf + 255) & ~size_t(255), b_pm = sizeof(double) * K2_G * d;
  const size_t o_nm = (o_pm + b_pm + 255) & ~size_t(255), b_nm = sizeof(float) * K2_G;
  const size_t o_ls = (o_nm + b_nm + 255) & ~size_t(255), b_ls = sizeof(unsigned) * (size_t)nw * 64 * 2 * K2_CAPH;
  const size_t o_kp = (o_ls + b_ls + 255) & ~size_t(255), b_kp = sizeof(int) * (size_t)nw * 64 * 128;
  char* base = (char*)ctx->scratch(o_kp + b_kp + 256);
  float* Cf = (float*)(base + o_cf);
  double* part = (double*)(base + o_pm);
  float* nmax = (float*)(base + o_nm);
  unsigned* list = (unsigned*)(base + o_ls);
  int* kept = (int*)(base + o_kp);
  static const bool dbg_on = [] { const char* e = std::getenv("HSSK_KNN_DEBUG"); return e && e[0] == '1'; }();
  long long* dbg = nullptr;
  if (dbg_on) {
    dbg = (long long*)(base + o_kp + b_kp);
    hssk_rt::memset_async(dbg, 0, 32, ctx->stream);
    if (const char* e = std::getenv("HSSK_KNN_DRY")) { long long m = std::atoll(e); hssk_rt::h2d(dbg + 2, &m, 8, ctx->stream); hssk_rt::sync(ctx->stream); }
  }
  HSSK_LAUNCH(knn2_mean_kernel, dim3(K2_G), dim3(256), 0, ctx->stream, X, d, n, part);
  HSSK_LAUNCH(knn2_prep_kernel, dim3(K2_G), dim3(256), 0, ctx->stream, X, d, n, ldc, KP, part, Cf, nmax);
  if (KSM == 3) knn2_launch_scan<3, 128>(ctx, nw, X, Cf, ldc, d, n, q0, q1, k, nmax, list, kept, out_idx, dbg);
  else if (KSM == 6) knn2_launch_scan<6, 128>(ctx, nw, X, Cf, ldc, d, n, q0, q1, k, nmax, list, kept, out_idx, dbg);
  else if (KSM == 10) knn2_launch_scan<10, 128>(ctx, nw, X, Cf, ldc, d, n, q0, q1, k, nmax, list, kept, out_idx, dbg);
  else knn2_launch_scan<16, 128>(ctx, nw, X, Cf, ldc, d, n, q0, q1, k, nmax, list, kept, out_idx, dbg);
  hssk_rt::check_launch();
  if (dbg) {
    long long h[2] = {0, 0};
    hssk_rt::d2h(h, dbg, 16, ctx->stream);
    hssk_rt::sync(ctx->stream);
    std::fprintf(stderr, "hssk_knn filtered: per query %.2f compactions over %.1f entries (%d kept by each)\n", (double)h[0] / nq, (double)h[1] / nq, k);
  }
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
