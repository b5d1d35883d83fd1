// Binary-tree clustering of a point set by MEDIAN SPLITS on the device: the row / column ordering and
// the cluster tree of a kernel matrix, for the two partitioners of the reference whose tree shape does not depend on the
// data -- cobble (clustering/CobblePartitioning.cpp:36-78: median of the distances from the point farthest from the
// centroid) and kd (clustering/KDTree.cpp:36-95: median along the coordinate of largest extent) -- driven by
// clustering/Clustering.hpp:143-168.  A cluster of n >= cluster_size points is split into its n / 2 points of smallest
// key (in their original order) and the rest (in the order the reference's swap sequence leaves them); the halves are
// split again on the next level.  One launch per level, one workgroup per cluster, nothing read back between levels.
//
// The host form (host/Clustering.hpp) calls libstdc++ on the reference's data to reproduce its permutation exactly; this
// form reproduces it wherever the answer does not hang on the last bits:
//   * keys are computed with the host's operations in the host's order (no fused multiply-add: products rounded before
//     they are added; IEEE square root), so the cobble keys -- distances from ONE data point -- are the host's bit for bit;
//   * the centroid is a parallel sum (the host adds the points one after the other): it differs from the host's in the
//     last bits, which only matters when two points are the farthest from it to within rounding -- the workgroup then
//     raises the status word (so does an exact tie);
//   * the median is found by selection (a histogram of the keys over 2048 monotone buckets, then ranking inside the
//     median's bucket); WHICH points lie below it is what std::nth_element leaves unless equal keys straddle the median --
//     then the reference's own call decides, and the status word is raised;
//   * the reference moves the n / 2 points labelled 0 to the front with a sequence of swaps: the k-th of them (in index
//     order) is swapped with whatever stands at position k - 1.  Labelled-0 points therefore keep their order; a point
//     labelled 1 that stands at position p < n / 2 when position p is filled moves to where that 0-point stood,
//     z(p) = index of the (p + 1)-th 0-point, and again (z(z(p)), ...) until it lands at or behind n / 2.  These chains are
//     short unless the labels are 1 0 0 0 ...-like; a chain longer than CL_CHAIN raises the status word.
// A raised status word means: take the host form (the caller still holds the untouched points).
#include "hssk_device.h"
#include "hssk_internal.h"

#include <cstdlib>
#include <vector>

namespace {

constexpr int CL_T = 1024;      // threads per workgroup (16 waves)
constexpr int CL_NB = 2048;     // buckets of the selection histogram
constexpr int CL_CAP = 4096;    // keys of the median's bucket ranked in the LDS
constexpr int CL_DMAX = 64;     // largest point dimension
constexpr int CL_CHAIN = 256;   // longest displacement chain followed

struct ClDesc {
  int lo, n;
};

// kernel/Metrics.hpp:41-50 followed by sqrt (Euclidean_distance)
__device__ inline double cl_dist(int d, const double* a, const double* b) {
  double k = 0.;
  for (int i = 0; i < d; i++) k = hssk_sq_acc_rn(k, a[i] - b[i]);   // (the product rounded before it is added: the host's arithmetic)
  return sqrt(k);
}

// i = tid, tid + CL_T, ... < n in batches of U: the U loads (or load-and-compute chains) of a batch are independent and in flight
// together -- the top clusters of a large point set are streamed by ONE workgroup, which a load at a time per thread leaves at
// the memory latency (1.6 ms for the first split of 1e5 points in R^8)
template <int U, typename T, class L, class F>
__device__ __forceinline__ void cl_stream(int tid, int n, L load, F use) {
  for (int i0 = tid; i0 < n; i0 += U * CL_T) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = load(min(i0 + u * CL_T, n - 1));
#pragma unroll
    for (int u = 0; u < U; u++)
      if (i0 + u * CL_T < n) use(i0 + u * CL_T, v[u]);
  }
}

__global__ __launch_bounds__(256) void cluster_iota_kernel(int* p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = i;
}

// algo 4: cobble, 2: kd.  X, perm: the points (d x n, a point per column) and the permutation so far, rearranged in place;
// Xt, pt, key, zpos: scratch of the same extents (each workgroup uses its own cluster's stretch).
__global__ __launch_bounds__(CL_T) void cluster_split_kernel(double* __restrict__ X, double* __restrict__ Xt, int* __restrict__ perm,
                                                             int* __restrict__ pt, double* __restrict__ key, int* __restrict__ zpos,
                                                             const ClDesc* __restrict__ cl, int d, int algo, int* status) {
  HSSK_SHARED double red[CL_T];
  HSSK_SHARED double red2[CL_T];
  HSSK_SHARED int ired[CL_T];
  HSSK_SHARED double cen[CL_DMAX];
  HSSK_SHARED double ext[2 * CL_DMAX];
  HSSK_SHARED int hist[CL_NB];
  HSSK_SHARED double cand[CL_CAP];
  HSSK_SHARED int wcnt[CL_T / 64];
  HSSK_SHARED int shi[8];      // 0: farthest point / split coordinate, 1: bucket, 2: keys below the bucket, 3: keys in it,
                               // 4: gather cursor, 5: keys of the bucket below the median value, 7: give up
  HSSK_SHARED double shd[4];   // 0: median value, 1: smallest key, 2: largest key
  const ClDesc c = cl[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = c.lo, n = c.n, h = n / 2;
  double* P = X + (size_t)lo * d;
  double* K = key + lo;
  const int Tp = (CL_T / d) * d;   // threads of the coordinate-wise passes: thread t always meets coordinate t % d
  if (tid < 8) shi[tid] = 0;
  __syncthreads();
  double kmin, kmax;
  if (algo == 4) {
    // centroid
    double s = 0.;
    if (tid < Tp) {
      const long long nd = (long long)n * d;
      for (long long e0 = tid; e0 < nd; e0 += 8LL * Tp) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = P[min(e0 + (long long)u * Tp, nd - 1)];
#pragma unroll
        for (int u = 0; u < 8; u++) s += e0 + (long long)u * Tp < nd ? x[u] : 0.;
      }
    }
    red[tid] = tid < Tp ? s : 0.;
    __syncthreads();
    if (tid < d) {
      double a = 0.;
      for (int m = tid; m < Tp; m += d) a += red[m];
      cen[tid] = a / n;
    }
    __syncthreads();
    // the point farthest from it (the first among equals) and the runner-up's distance
    double v1 = -1., v2 = -1.;
    int i1 = 0x7fffffff;
    cl_stream<4, double>(tid, n, [&](int i) { return cl_dist(d, P + (size_t)i * d, cen); }, [&](int i, double dd) {
      if (dd > v1) { v2 = v1; v1 = dd; i1 = i; }
      else if (dd > v2) v2 = dd;
    });
    red[tid] = v1; red2[tid] = v2; ired[tid] = i1;
    __syncthreads();
    for (int s2 = CL_T / 2; s2 > 0; s2 >>= 1) {
      if (tid < s2) {
        const double va = red[tid], vb = red[tid + s2], wa = red2[tid], wb = red2[tid + s2];
        const int ia = ired[tid], ib = ired[tid + s2];
        if (vb > va || (vb == va && ib < ia)) { red[tid] = vb; ired[tid] = ib; red2[tid] = fmax(va, wb); }
        else red2[tid] = fmax(vb, wa);
      }
      __syncthreads();
    }
    const int first = ired[0];
    const double far = red[0], second = red2[0];
    if (!(far - second > 1e-10 * far)) {   // a tie to within the centroid's rounding (or NaNs): the host's sum decides
      if (tid == 0) hssk_flag_store(status, 1);
      return;
    }
    __syncthreads();
    if (tid < d) cen[tid] = P[(size_t)first * d + tid];
    __syncthreads();
    double mn = __builtin_huge_val(), mx = -__builtin_huge_val();
    cl_stream<4, double>(tid, n, [&](int i) { return cl_dist(d, P + (size_t)i * d, cen); }, [&](int i, double dd) {
      K[i] = dd;
      mn = fmin(mn, dd); mx = fmax(mx, dd);
    });
    red[tid] = mn; red2[tid] = mx;
    __syncthreads();
    for (int s2 = CL_T / 2; s2 > 0; s2 >>= 1) {
      if (tid < s2) { red[tid] = fmin(red[tid], red[tid + s2]); red2[tid] = fmax(red2[tid], red2[tid + s2]); }
      __syncthreads();
    }
    kmin = red[0]; kmax = red2[0];
  } else {
    // extent of every coordinate; the first of largest extent
    double mn = __builtin_huge_val(), mx = -__builtin_huge_val();
    if (tid < Tp) {
      const long long nd = (long long)n * d;
      for (long long e0 = tid; e0 < nd; e0 += 8LL * Tp) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = P[min(e0 + (long long)u * Tp, nd - 1)];
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (e0 + (long long)u * Tp < nd) { mn = fmin(mn, x[u]); mx = fmax(mx, x[u]); }   // (a clamped element is another coordinate's)
      }
    }
    red[tid] = mn; red2[tid] = mx;
    __syncthreads();
    if (tid < d) {
      double a = red[tid], b = red2[tid];
      for (int m = tid + d; m < Tp; m += d) { a = fmin(a, red[m]); b = fmax(b, red2[m]); }
      ext[tid] = a; ext[CL_DMAX + tid] = b;
    }
    __syncthreads();
    if (tid == 0) {
      int dim = 0;
      double e0 = ext[CL_DMAX] - ext[0];
      for (int j = 1; j < d; j++)
        if (ext[CL_DMAX + j] - ext[j] > e0) { e0 = ext[CL_DMAX + j] - ext[j]; dim = j; }
      shi[0] = dim;
    }
    __syncthreads();
    const int dim = shi[0];
    kmin = ext[dim]; kmax = ext[CL_DMAX + dim];
    cl_stream<8, double>(tid, n, [&](int i) { return P[(size_t)i * d + dim]; }, [&](int i, double x) { K[i] = x; });
  }
  if (!(kmax > kmin)) {   // all keys equal (or NaNs)
    if (tid == 0) hssk_flag_store(status, 2);
    return;
  }
  // ---- the median: the key of rank h (0-based).  Buckets are a monotone map of the keys, so the bucket of the median is
  // the one the cumulative counts say; inside it the keys are ranked against each other.
  for (int b = tid; b < CL_NB; b += CL_T) hist[b] = 0;
  __syncthreads();
  const double scale = (CL_NB - 1) / (kmax - kmin);
  auto bucket = [&](double k) {
    const int b = (int)((k - kmin) * scale);
    return min(CL_NB - 1, max(0, b));
  };
  cl_stream<8, double>(tid, n, [&](int i) { return K[i]; }, [&](int, double k) { hssk_lds_inc(&hist[bucket(k)]); });
  __syncthreads();
  {
    const int c0 = hist[2 * tid], c1 = hist[2 * tid + 1];
    ired[tid] = c0 + c1;
    __syncthreads();
    for (int off = 1; off < CL_T; off <<= 1) {   // inclusive scan over the pairs of buckets
      const int add = tid >= off ? ired[tid - off] : 0;
      __syncthreads();
      ired[tid] += add;
      __syncthreads();
    }
    const int base = ired[tid] - c0 - c1;
    if (base <= h && h < base + c0) { shi[1] = 2 * tid; shi[2] = base; shi[3] = c0; }
    else if (base + c0 <= h && h < base + c0 + c1) { shi[1] = 2 * tid + 1; shi[2] = base + c0; shi[3] = c1; }
  }
  __syncthreads();
  const int B = shi[1], below = shi[2], nb = shi[3], r = h - below;
  if (nb > CL_CAP || nb <= 0) {   // (a pile of near-equal keys)
    if (tid == 0) hssk_flag_store(status, 3);
    return;
  }
  cl_stream<8, double>(tid, n, [&](int i) { return K[i]; }, [&](int, double k) {
    if (bucket(k) == B) cand[hssk_lds_inc(&shi[4])] = k;
  });
  __syncthreads();
  for (int a = tid; a < nb; a += CL_T) {
    const double x = cand[a];
    int less = 0, eq = 0;
    for (int j = 0; j < nb; j++) { const double y = cand[j]; less += y < x; eq += y == x; }
    if (less <= r && r < less + eq) { shd[0] = x; shi[5] = less; }   // (equal keys write equal values)
  }
  __syncthreads();
  const double v = shd[0];
  if (shi[5] != r) {   // equal keys straddle the median: std::nth_element's arrangement decides (host)
    if (tid == 0) hssk_flag_store(status, 4);
    return;
  }
  // ---- the points below the median to the front, in order; every wave a contiguous stretch of the cluster
  const int seg = ((n + CL_T - 1) / CL_T) * 64, i0 = wave * seg, i1 = min(n, i0 + seg);
  int cnt = 0;
  for (int b0 = i0; b0 < i1; b0 += 8 * 64) {
    double k[8];
#pragma unroll
    for (int u = 0; u < 8; u++) k[u] = K[min(b0 + u * 64 + lane, n - 1)];
#pragma unroll
    for (int u = 0; u < 8; u++) cnt += __builtin_popcountll(hssk_ballot(b0 + u * 64 + lane < i1 && k[u] < v));
  }
  if (lane == 0) wcnt[wave] = cnt;
  __syncthreads();
  int run = 0;
  for (int u = 0; u < wave; u++) run += wcnt[u];
  for (int b0 = i0; b0 < i1; b0 += 4 * 64) {
    double k[4];
    int pm[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int i = min(b0 + u * 64 + lane, n - 1); k[u] = K[i]; pm[u] = perm[lo + i]; }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = b0 + u * 64 + lane;
      const int z = i < i1 && k[u] < v;
      const unsigned long long m = hssk_ballot(z);
      if (z) {
        const int rk = run + __builtin_popcountll(m & ((1ULL << lane) - 1ULL));
        zpos[lo + rk] = i;
        for (int j = 0; j < d; j++) Xt[(size_t)(lo + rk) * d + j] = P[(size_t)i * d + j];
        pt[lo + rk] = pm[u];
      }
      run += __builtin_popcountll(m);
    }
  }
  __syncthreads();
  // ---- the others: follow the displacements (see the head of the file)
  cl_stream<4, int>(tid, n, [&](int i) {
    if (K[i] < v) return -1;
    int p = i, steps = 0;
    while (p < h && steps <= CL_CHAIN) { p = zpos[lo + p]; steps++; }
    return p;
  }, [&](int i, int p) {
    if (p < 0) return;
    if (p < h) { shi[7] = 1; return; }
    for (int j = 0; j < d; j++) Xt[(size_t)(lo + p) * d + j] = P[(size_t)i * d + j];
    pt[lo + p] = perm[lo + i];
  });
  __syncthreads();
  if (shi[7]) {
    if (tid == 0) hssk_flag_store(status, 5);
    return;
  }
  {
    const long long nd = (long long)n * d;
    const double* src = Xt + (size_t)lo * d;
    for (long long e0 = tid; e0 < nd; e0 += 8LL * CL_T) {
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; u++) x[u] = src[min(e0 + (long long)u * CL_T, nd - 1)];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (e0 + (long long)u * CL_T < nd) P[e0 + (long long)u * CL_T] = x[u];
    }
  }
  cl_stream<8, int>(tid, n, [&](int i) { return pt[lo + i]; }, [&](int i, int q) { perm[lo + i] = q; });
}

// ---------------------------------------------------------------------------------------------
// Clusters of many points (the top levels: one workgroup streams 1e5 points 25 times in 1.5 ms) are split by SEVERAL workgroups:
// the same steps as cluster_split_kernel, each a launch of its own over (cluster, chunk of its points), what the workgroups of a
// cluster tell each other goes through a per-cluster workspace in global memory (partial sums and extrema per chunk in chunk
// order -- deterministic --, the key histogram by integer atomics).  Phases: 1 partial sums / extents, 2 centroid -> farthest
// candidates, 3 farthest (or split coordinate) -> keys, extrema, 4 histogram, 5 the median's bucket [one workgroup per cluster],
// 6 its keys gathered, 7 the median [one workgroup per cluster], 8 points below the median counted, 9 ... moved in order,
// 10 the others along their displacement chains, 11 the copy back.
constexpr int CW_NCH = 64;   // most chunks per cluster
struct CwDesc {
  int ci, lo, n, c0, cn, nch, chunk;   // cluster (index in the level's list), its points, this chunk's points [c0, c0 + cn), chunks, index
};
// per cluster: doubles  part[2 * NCH * DMAX] | v1[NCH] | v2[NCH] | mn[NCH] | mx[NCH] | pt0[DMAX] | scal[8] | cand[CAP]
//              ints     i1[NCH] | hist[NB] | zc[NCH] | sel[16]
constexpr size_t CW_DWS = 2 * (size_t)CW_NCH * CL_DMAX + 4 * CW_NCH + CL_DMAX + 8 + CL_CAP;
constexpr size_t CW_IWS = CW_NCH + CL_NB + CW_NCH + 16;

__global__ __launch_bounds__(CL_T) void cluster_wide_kernel(int phase, double* __restrict__ X, double* __restrict__ Xt, int* __restrict__ perm,
                                                            int* __restrict__ pt, double* __restrict__ key, int* __restrict__ zpos,
                                                            const CwDesc* __restrict__ cw, double* __restrict__ dws, int* __restrict__ iws,
                                                            int d, int algo, int* status) {
  HSSK_SHARED double red[CL_T];
  HSSK_SHARED double red2[CL_T];
  HSSK_SHARED int ired[CL_T];
  HSSK_SHARED double cen[CL_DMAX];
  HSSK_SHARED int hist[CL_NB];
  HSSK_SHARED int wcnt[CL_T / 64];
  HSSK_SHARED int shi[8];
  HSSK_SHARED double shd[4];
  if (hssk_flag_load(status) != 0) return;   // (an earlier phase or level gave up: the workspace may be stale)
  const CwDesc c = cw[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = c.lo, n = c.n, h = n / 2, c0 = c.c0, cn = c.cn, nch = c.nch;
  double* P = X + (size_t)lo * d;
  double* K = key + lo;
  double* D = dws + (size_t)c.ci * CW_DWS;
  int* I = iws + (size_t)c.ci * CW_IWS;
  double *part = D, *fv1 = D + 2 * CW_NCH * CL_DMAX, *fv2 = fv1 + CW_NCH, *cmn = fv2 + CW_NCH, *cmx = cmn + CW_NCH, *pt0 = cmx + CW_NCH,
         *scal = pt0 + CL_DMAX, *cand = scal + 8;   // scal: 0 median, 1 smallest key, 2 largest key
  int *fi1 = I, *ghist = I + CW_NCH, *zc = ghist + CL_NB, *sel = zc + CW_NCH;   // sel: 0 first / dim, 1 bucket, 2 below, 3 in it, 4 cursor
  const int Tp = (CL_T / d) * d;
  const double* Pc = P + (size_t)c0 * d;
  if (phase == 1) {
    if (c.chunk == 0) for (int b = tid; b < CL_NB; b += CL_T) ghist[b] = 0;
    // coordinate-wise sums (cobble) or extents (kd) of the chunk
    double s = 0., mn = __builtin_huge_val(), mx = -__builtin_huge_val();
    if (tid < Tp)
      for (long long e = tid; e < (long long)cn * d; e += Tp) { const double x = Pc[e]; s += x; mn = fmin(mn, x); mx = fmax(mx, x); }
    red[tid] = algo == 4 ? (tid < Tp ? s : 0.) : mn;
    red2[tid] = mx;
    __syncthreads();
    if (tid < d) {
      double a = red[tid], b = red2[tid];
      for (int m = tid + d; m < Tp; m += d) {
        if (algo == 4) a += red[m];
        else { a = fmin(a, red[m]); b = fmax(b, red2[m]); }
      }
      part[(size_t)c.chunk * d + tid] = a;
      part[(size_t)(CW_NCH + c.chunk) * d + tid] = b;
    }
    return;
  }
  if (phase == 2) {
    if (algo != 4) return;
    if (tid < d) {
      double a = 0.;
      for (int q = 0; q < nch; q++) a += part[(size_t)q * d + tid];
      cen[tid] = a / n;
    }
    __syncthreads();
    double v1 = -1., v2 = -1.;
    int i1 = 0x7fffffff;
    cl_stream<2, double>(tid, cn, [&](int i) { return cl_dist(d, Pc + (size_t)i * d, cen); }, [&](int i, double dd) {
      if (dd > v1) { v2 = v1; v1 = dd; i1 = c0 + i; }
      else if (dd > v2) v2 = dd;
    });
    red[tid] = v1; red2[tid] = v2; ired[tid] = i1;
    __syncthreads();
    for (int s2 = CL_T / 2; s2 > 0; s2 >>= 1) {
      if (tid < s2) {
        const double va = red[tid], vb = red[tid + s2], wa = red2[tid], wb = red2[tid + s2];
        const int ia = ired[tid], ib = ired[tid + s2];
        if (vb > va || (vb == va && ib < ia)) { red[tid] = vb; ired[tid] = ib; red2[tid] = fmax(va, wb); }
        else red2[tid] = fmax(vb, wa);
      }
      __syncthreads();
    }
    if (tid == 0) { fv1[c.chunk] = red[0]; fv2[c.chunk] = red2[0]; fi1[c.chunk] = ired[0]; }
    return;
  }
  if (phase == 3) {
    int dim = 0;
    if (algo == 4) {
      if (tid == 0) {
        double va = fv1[0], wa = fv2[0];
        int ia = fi1[0];
        for (int q = 1; q < nch; q++) {
          const double vb = fv1[q], wb = fv2[q];
          const int ib = fi1[q];
          if (vb > va || (vb == va && ib < ia)) { wa = fmax(va, wb); va = vb; ia = ib; }
          else wa = fmax(vb, wa);
        }
        shi[0] = ia;
        shi[1] = !(va - wa > 1e-10 * va);
      }
      __syncthreads();
      if (shi[1]) {
        if (tid == 0) hssk_flag_store(status, 1);
        return;
      }
      if (tid < d) cen[tid] = P[(size_t)shi[0] * d + tid];
      __syncthreads();
    } else {
      if (tid == 0) {
        double e0 = -1.;
        for (int j = 0; j < d; j++) {
          double a = part[j], b = part[(size_t)CW_NCH * d + j];
          for (int q = 1; q < nch; q++) { a = fmin(a, part[(size_t)q * d + j]); b = fmax(b, part[(size_t)(CW_NCH + q) * d + j]); }
          if (b - a > e0) { e0 = b - a; dim = j; }
        }
        shi[0] = dim;
      }
      __syncthreads();
      dim = shi[0];
    }
    if (c.chunk == 0 && tid == 0) sel[0] = shi[0];
    double mn = __builtin_huge_val(), mx = -__builtin_huge_val();
    cl_stream<2, double>(tid, cn, [&](int i) { return algo == 4 ? cl_dist(d, Pc + (size_t)i * d, cen) : Pc[(size_t)i * d + dim]; },
                         [&](int i, double dd) { K[c0 + i] = dd; mn = fmin(mn, dd); mx = fmax(mx, dd); });
    red[tid] = mn; red2[tid] = mx;
    __syncthreads();
    for (int s2 = CL_T / 2; s2 > 0; s2 >>= 1) {
      if (tid < s2) { red[tid] = fmin(red[tid], red[tid + s2]); red2[tid] = fmax(red2[tid], red2[tid + s2]); }
      __syncthreads();
    }
    if (tid == 0) { cmn[c.chunk] = red[0]; cmx[c.chunk] = red2[0]; }
    return;
  }
  // ---- from here on every workgroup of the cluster needs the key range
  double kmin = cmn[0], kmax = cmx[0];
  for (int q = 1; q < nch; q++) { kmin = fmin(kmin, cmn[q]); kmax = fmax(kmax, cmx[q]); }
  if (!(kmax > kmin)) {
    if (tid == 0) hssk_flag_store(status, 2);
    return;
  }
  const double scale = (CL_NB - 1) / (kmax - kmin);
  auto bucket = [&](double k) {
    const int b = (int)((k - kmin) * scale);
    return min(CL_NB - 1, max(0, b));
  };
  if (phase == 4) {
    for (int b = tid; b < CL_NB; b += CL_T) hist[b] = 0;
    __syncthreads();
    cl_stream<2, double>(tid, cn, [&](int i) { return K[c0 + i]; }, [&](int, double k) { hssk_lds_inc(&hist[bucket(k)]); });
    __syncthreads();
    for (int b = tid; b < CL_NB; b += CL_T)
      if (hist[b]) hssk_gadd_i(&ghist[b], hist[b]);
    return;
  }
  if (phase == 5) {   // (one workgroup per cluster)
    const int a0 = ghist[2 * tid], a1 = ghist[2 * tid + 1];
    ired[tid] = a0 + a1;
    __syncthreads();
    for (int off = 1; off < CL_T; off <<= 1) {
      const int add = tid >= off ? ired[tid - off] : 0;
      __syncthreads();
      ired[tid] += add;
      __syncthreads();
    }
    const int base = ired[tid] - a0 - a1;
    if (base <= h && h < base + a0) { sel[1] = 2 * tid; sel[2] = base; sel[3] = a0; }
    else if (base + a0 <= h && h < base + a0 + a1) { sel[1] = 2 * tid + 1; sel[2] = base + a0; sel[3] = a1; }
    if (tid == 0) sel[4] = 0;
    return;
  }
  const int B = sel[1], below = sel[2], nb = sel[3], r = h - below;
  if (nb > CL_CAP || nb <= 0) {
    if (tid == 0) hssk_flag_store(status, 3);
    return;
  }
  if (phase == 6) {
    cl_stream<2, double>(tid, cn, [&](int i) { return K[c0 + i]; }, [&](int, double k) {
      if (bucket(k) == B) {
        const int slot = hssk_gadd_i(&sel[4], 1);
        if (slot < CL_CAP) cand[slot] = k;
      }
    });
    return;
  }
  if (phase == 7) {   // (one workgroup per cluster)
    if (tid == 0) shi[5] = -1;
    __syncthreads();
    for (int a = tid; a < nb; a += CL_T) {
      const double x = cand[a];
      int less = 0, eq = 0;
      for (int j = 0; j < nb; j++) { const double y = cand[j]; less += y < x; eq += y == x; }
      if (less <= r && r < less + eq) { shd[0] = x; shi[5] = less; }
    }
    __syncthreads();
    if (tid == 0) { scal[0] = shd[0]; sel[5] = shi[5]; }
    return;
  }
  if (sel[5] != r) {
    if (tid == 0) hssk_flag_store(status, 4);
    return;
  }
  const double v = scal[0];
  // the chunk's points wave by wave: every wave a contiguous stretch
  const int seg = ((cn + CL_T - 1) / CL_T) * 64, i0 = c0 + wave * seg, i1 = min(c0 + cn, i0 + seg);
  if (phase == 8 || phase == 9) {
    int cnt = 0;
    for (int b0 = i0; b0 < i1; b0 += 64) {
      const int i = b0 + lane;
      cnt += __builtin_popcountll(hssk_ballot(i < i1 && K[min(i, n - 1)] < v));
    }
    if (lane == 0) wcnt[wave] = cnt;
    __syncthreads();
    if (phase == 8) {
      if (tid == 0) {
        int t = 0;
        for (int w = 0; w < CL_T / 64; w++) t += wcnt[w];
        zc[c.chunk] = t;
      }
      return;
    }
    int run = 0;
    for (int q = 0; q < c.chunk; q++) run += zc[q];
    for (int u = 0; u < wave; u++) run += wcnt[u];
    for (int b0 = i0; b0 < i1; b0 += 64) {
      const int i = b0 + lane;
      const int z = i < i1 && K[min(i, n - 1)] < v;
      const unsigned long long m = hssk_ballot(z);
      if (z) {
        const int rk = min(run + __builtin_popcountll(m & ((1ULL << lane) - 1ULL)), h - 1);
        zpos[lo + rk] = i;
        for (int j = 0; j < d; j++) Xt[(size_t)(lo + rk) * d + j] = P[(size_t)i * d + j];
        pt[lo + rk] = perm[lo + i];
      }
      run += __builtin_popcountll(m);
    }
    return;
  }
  if (phase == 10) {
    cl_stream<2, int>(tid, cn, [&](int ii) {
      const int i = c0 + ii;
      if (K[i] < v) return -1;
      int p = i, steps = 0;
      while (p >= 0 && p < h && steps <= CL_CHAIN) { p = zpos[lo + p]; steps++; }
      return p < n ? p : 0;
    }, [&](int ii, int p) {
      const int i = c0 + ii;
      if (p < 0) return;
      if (p < h) { hssk_flag_store(status, 5); return; }
      for (int j = 0; j < d; j++) Xt[(size_t)(lo + p) * d + j] = P[(size_t)i * d + j];
      pt[lo + p] = perm[lo + i];
    });
    return;
  }
  // phase 11: the chunk's stretch back (X and perm of the cluster are complete in Xt / pt: every point was moved)
  for (long long e = tid; e < (long long)cn * d; e += CL_T) P[(size_t)c0 * d + e] = Xt[(size_t)(lo + c0) * d + e];
  for (int i = tid; i < cn; i += CL_T) perm[lo + c0 + i] = pt[lo + c0 + i];
}

}  // namespace

// X (d x n, device) and perm (n ints, device; out, 0-based: new column i is old column perm[i]) rearranged into the cluster
// order of the binary tree whose clusters of >= cluster_size points are halved (n / 2 | n - n / 2).  *status (host, out) is 0
// when the device form reproduced the reference's arrangement; non-zero: ties or long displacement chains were met and X /
// perm are not to be used.  algo: 2 kd, 4 cobble (the reference's enumeration order: natural, 2means, kdtree, pca, cobble).
// Synchronises.
extern "C" int hssk_cluster_median(hssk_ctx* ctx, double* X, int d, int n, int algo, int cluster_size, int* perm, int* status) {
  HSSK_API_BEGIN
  if (algo != 2 && algo != 4) HSSK_UNSUPPORTED("median-split clustering on the device is cobble (4) or kd (2)");
  if (d <= 0 || d > CL_DMAX) HSSK_UNSUPPORTED("point dimension must be in [1, 64]");
  if (n <= 0 || cluster_size < 2) throw std::invalid_argument("hssk_cluster_median: empty point set or cluster size < 2");
  // scratch: Xt | key | pt | zpos | status word
  const size_t nd = (size_t)n * d;
  // clusters of CW_MIN points and more are split by several workgroups (HSSK_CLUSTER_WIDE_MIN; 0: never)
  const char* wenv = std::getenv("HSSK_CLUSTER_WIDE_MIN");   // (read per call: the tests take both forms in one process)
  const int wide_min = wenv ? std::atoi(wenv) : 16384;
  const int nwide_max = wide_min > 0 ? std::max(1, n / wide_min) : 0;
  char* base = (char*)ctx->scratch(sizeof(double) * (nd + n + CW_DWS * nwide_max) + sizeof(int) * (2 * (size_t)n + 16 + CW_IWS * nwide_max));
  double* Xt = (double*)base;
  double* key = Xt + nd;
  double* dws = key + n;
  int* pt = (int*)(dws + CW_DWS * nwide_max);
  int* zpos = pt + n;
  int* dstat = zpos + n;
  int* iws = dstat + 16;
  hssk_rt::memset_async(dstat, 0, sizeof(int), ctx->stream);
  HSSK_LAUNCH(cluster_iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, perm, n);
  std::vector<ClDesc> cur{ClDesc{0, n}}, next;
  while (!cur.empty()) {
    std::vector<ClDesc> split;
    std::vector<CwDesc> wide, wide1;   // (cluster, chunk) and one entry per cluster
    next.clear();
    for (const ClDesc& c : cur) {
      if (c.n < cluster_size) continue;   // (cluster_size >= 2: both halves non-empty)
      if (wide_min > 0 && c.n >= wide_min && (int)wide1.size() < nwide_max) {
        const int cs = std::max(2048, (((c.n + CW_NCH - 1) / CW_NCH) + 63) & ~63), nch = (c.n + cs - 1) / cs, ci = (int)wide1.size();
        for (int q = 0; q < nch; q++) wide.push_back(CwDesc{ci, c.lo, c.n, q * cs, std::min(cs, c.n - q * cs), nch, q});
        wide1.push_back(CwDesc{ci, c.lo, c.n, 0, std::min(cs, c.n), nch, 0});
      } else {
        split.push_back(c);
      }
      next.push_back(ClDesc{c.lo, c.n / 2});
      next.push_back(ClDesc{c.lo + c.n / 2, c.n - c.n / 2});
    }
    if (!wide.empty()) {
      auto* dw = (const CwDesc*)ctx->stage(wide.data(), sizeof(CwDesc) * wide.size());
      auto* d1 = (const CwDesc*)ctx->stage(wide1.data(), sizeof(CwDesc) * wide1.size());
      for (int phase = 1; phase <= 11; phase++) {
        if (phase == 2 && algo != 4) continue;
        const bool one = phase == 5 || phase == 7;
        HSSK_LAUNCH(cluster_wide_kernel, dim3((unsigned)(one ? wide1.size() : wide.size())), dim3(CL_T), 0, ctx->stream, phase, X, Xt, perm, pt, key,
                    zpos, one ? d1 : dw, dws, iws, d, algo, dstat);
      }
    }
    if (!split.empty()) {
      // (levels of many clusters: the table in pieces the staging ring takes)
      const size_t piece = 1 << 16;
      for (size_t o = 0; o < split.size(); o += piece) {
        const size_t cnt = std::min(piece, split.size() - o);
        auto* dd = (const ClDesc*)ctx->stage(split.data() + o, sizeof(ClDesc) * cnt);
        HSSK_LAUNCH(cluster_split_kernel, dim3((unsigned)cnt), dim3(CL_T), 0, ctx->stream, X, Xt, perm, pt, key, zpos, dd, d, algo, dstat);
      }
    }
    cur.swap(next);
  }
  hssk_rt::check_launch();
  int st = 0;
  hssk_rt::d2h(&st, dstat, sizeof(int), ctx->stream);
  hssk_rt::sync(ctx->stream);
  if (status) *status = st;
  HSSK_API_END
}
