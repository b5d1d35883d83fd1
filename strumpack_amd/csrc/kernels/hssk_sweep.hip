// Single-launch tree sweeps: the ULV solve (forward / backward) and the HSS mat-vec (up / down) of a whole tree in ONE
// kernel launch each (right-hand sides in groups of four along blockIdx.y).
//
// A sweep over an HSS tree is a chain of ~10 dependent levels of tiny per-node operations; launched level by level it
// is bound by launch-to-launch latency (N = 1e5: 20 launches, 0.55 ms apply / 1.07 ms solve for 0.19 / 0.52 GB of
// blocks).  Here every node of the sweep is one workgroup of the same launch, ordered so that a node only depends on
// workgroups with a LOWER index (children before parents going up, parents before children going down).  A workgroup
// waits for the vectors its dependencies hand over (coherent sc1 accesses, see below and hssk_device.h), does the node's
// arithmetic with the vectors in LDS, and stores its own results for the next level.  Workgroups are dispatched in
// index order (round-robin over the XCDs, in order within an XCD), so the lowest-indexed unfinished workgroup is always
// resident and never waits on a non-resident one: the scheme cannot deadlock, whatever the occupancy; a bounded spin
// count turns any violation of that assumption into an error code instead of a hang.  A recorded launch (hssk_plan_*)
// is replayed together with the small launch that re-arms the hand-off buffers.
//
// Reference arithmetic: HSSMatrix::solve_fwd / solve_bwd (HSS/HSSMatrix.solve.hpp:69-238), apply_fwd / apply_bwd
// (HSS/HSSMatrix.apply.hpp:55-220), HSSBasisID::apply / applyC (HSS/HSSBasisID.hpp:155-203).  Differences in the
// stored factors (see DeviceHSS::factor_sub): WQ = W1 Q~(:, 0:q) is formed once at factor time, so the forward sweep
// reads r x q instead of W1 (r x m) and Q~(:, 0:q) (m x q); the substitution with R~^T runs on 64-row blocks whose
// diagonal blocks were inverted at factor time (Tinv), so the dependent chain is q/64 block steps, not q scalar steps.
//
// Per-node GEMVs are written for memory-level parallelism (the blocks stream from HBM exactly once): thread per output
// row with 16 independent loads in flight (gemv_n), or four adjacent lanes per output column for column-contiguous
// operands (gemv_t).
#include "hssk_backsub.h"
#include "hssk_device.h"
#include "hssk_internal.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {

constexpr int SW_T = 256;     // threads per workgroup
constexpr int SW_NR = 4;      // right-hand sides handled at once (kernels are instantiated for NR = 4 and NR = 1)
constexpr int SW_MAX = 256;   // largest node dimension (rows of a basis)
constexpr int SW_NB = 64;     // block size of the substitution with R~^T (hssk_trtri_diag_vbatched)
// wide instantiation for many right-hand sides on small nodes (the inner levels of the hybrid path, hss_apply.cpp /
// hss_solve.cpp): sixteen right-hand sides per pass, node dimensions up to 128 -- a quarter of the passes (and of their
// barrier-separated stages) through every node; LDS 112 KB for the forward sweep
constexpr int SW_NRW = 16;
constexpr int SW_MAXW = 128;
constexpr long SW_SPIN_LIMIT = 1L << 22;

// ---- hand-off between workgroups of one launch ---------------------------------------------------------------------
// The vectors a node hands to its parent (or children) are a few hundred bytes.  They are written with coherent stores
// and the consumer polls THE DATA WORDS THEMSELVES: the hand-off buffers are filled with a sentinel (a NaN with a payload
// no computation produces) by one small launch before the sweep, and a consumer thread re-reads its element until it is
// no longer the sentinel.  One memory round trip per tree level -- the first version (data, drain, flag store; poll the
// flag, then load the data) paid three.  No flags, no consumer counts; the bounded spin turns a violated dispatch-order
// assumption into an error code.
constexpr unsigned long long SW_SENTINEL = 0x7FF8DEADBEEF5EEDull;
__device__ __forceinline__ bool is_sentinel(double v) { return hssk_bits(v) == SW_SENTINEL; }
__device__ __forceinline__ double sweep_take(const double* p, size_t off, int* err) {
  double v = hssk_cload(p, off);
  long spins = 0;
  while (is_sentinel(v)) {
    hssk_pause();
    if (++spins > SW_SPIN_LIMIT) { hssk_flag_raise(err); return 0.; }
    v = hssk_cload(p, off);
  }
  return v;
}
__global__ void sweep_fill_kernel(double* p, long long count) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (long long)gridDim.x * blockDim.x)
    hssk_cstore(p, (size_t)e, hssk_from_bits(SW_SENTINEL));
}

// ---- workgroup GEMVs on LDS vectors (leading dimension LDV per right-hand side) -----------------------------------
enum { OP_SET = 0, OP_ADD = 1, OP_SUB = 2 };
__device__ __forceinline__ void apply_op(double* o, double v, int op) {
  *o = op == OP_SET ? v : (op == OP_ADD ? *o + v : *o - v);
}

// Row-wise GEMV with per-row operands: out[i] (op)= sum_{k < K_i} a_i[k lda_i] x_i[k] for i < M, where rowop(i) names row
// i's operands (several stacked matrices that share a stage are ONE pass, so their loads are in flight together).  Lanes
// run along the rows (contiguous in memory); with M <= 128 the K range is split over 2 / 4 thread groups (partials meet in
// s_p) so that all 256 threads have loads in flight; M > 256 takes further passes.  Contains barriers: every thread of the
// workgroup must call.  x and out must not alias.
struct RowOp {
  const double* a;   // first element of the row
  int lda, K;
  const double* x;   // LDS vector (leading dimension LDV per right-hand side)
  double* o;         // LDS output element of the row (same leading dimension)
  int op;
};
// A thread's slice of the FIRST pass of gemv_rows (rows [0, SW_T)): row i, K range [k0, k1).  One definition for the pass
// itself and for rows_prefetch, which loads the slice's matrix elements into registers before the vectors exist.
constexpr int SW_GRP = 8;    // loads a thread keeps in flight in the streaming part of a pass
// prefetched elements per thread and pass (K <= 24 with two K partitions, <= 48 with four).  Twelve, not more: the slices set
// the kernels' register count, and with it how many workgroups a CU holds.  At 24 the one-vector mat-vec needed 189 registers
// and the forward solve 249 -- two workgroups per CU, 512 slots for the 2045 resp. 1023 workgroups of an N = 1e5 tree, so the
// inner nodes waiting for their turn held the slots the leaves needed and the leaves' D x (84 % of the bytes) only started once
// the up-sweep had ended; at 12 it is 116 / 153 registers, four / three workgroups per CU (mat-vec 0.089 -> 0.064 ms,
// profiles/r06_sweeps.md).  Nodes up to rank 48 still find their whole block in the slices.
constexpr int SW_PRE = 12;
struct Pre { double v[SW_PRE]; };
struct RowSlice { int i, part, P, M, M64, k0, k1; bool act; RowOp ro; };
template <class F>
__device__ __forceinline__ RowSlice rows_slice(int Mtot, int r0, F rowop) {
  RowSlice sl;
  const int tid = threadIdx.x;
  sl.M = min(Mtot - r0, SW_T);
  sl.M64 = max(64, (sl.M + 63) & ~63);
  sl.P = sl.M64 <= 64 ? 4 : (sl.M64 <= 128 ? 2 : 1);
  sl.i = sl.P == 1 ? tid : tid % sl.M64;
  sl.part = sl.P == 1 ? 0 : tid / sl.M64;
  sl.act = sl.i < sl.M && sl.part < sl.P;
  sl.ro = RowOp{};
  sl.k0 = sl.k1 = 0;
  if (sl.act) {
    sl.ro = rowop(r0 + sl.i);
    const int Kc = (sl.ro.K + sl.P - 1) / sl.P;
    sl.k0 = sl.part * Kc;
    sl.k1 = min(sl.ro.K, sl.k0 + Kc);
  }
  return sl;
}
template <class F>
__device__ __forceinline__ void rows_prefetch(int Mtot, F rowop, Pre& pre) {
  const RowSlice sl = rows_slice(Mtot, 0, rowop);
#pragma unroll
  for (int u = 0; u < SW_PRE; u++) pre.v[u] = (sl.act && sl.k0 + u < sl.k1) ? hssk_gload(sl.ro.a, (size_t)(sl.k0 + u) * sl.ro.lda) : 0.;
}
template <int NR, int LDV, class F>
__device__ __forceinline__ void gemv_rows(int Mtot, F rowop, int nrhs, double* s_p, const Pre& pre, bool use_pre) {
  for (int r0 = 0; r0 < Mtot || r0 == 0; r0 += SW_T) {
    const RowSlice sl = rows_slice(Mtot, r0, rowop);
    const int M = sl.M, M64 = sl.M64, P = sl.P, i = sl.i, part = sl.part;
    const RowOp ro = sl.ro;
    double acc[NR] = {};
    if (sl.act) {
      int k = sl.k0;
      if (use_pre && r0 == 0) {   // the first SW_PRE elements of the slice are already in registers (same order of summation)
#pragma unroll
        for (int u = 0; u < SW_PRE; u++) {
          if (sl.k0 + u < sl.k1) {
            const double t = pre.v[u];
#pragma unroll
            for (int c = 0; c < NR; c++) acc[c] += t * ro.x[sl.k0 + u + c * LDV];
          }
        }
        k = min(sl.k1, sl.k0 + SW_PRE);
      }
      // the rest in groups of SW_GRP predicated loads, all in flight together (a unrolled-by-n loop leaves up to n - 1
      // iterations to a remainder loop that waits out one memory round trip per element)
      for (; k < sl.k1; k += SW_GRP) {
        double t[SW_GRP];
#pragma unroll
        for (int u = 0; u < SW_GRP; u++) t[u] = k + u < sl.k1 ? hssk_gload(ro.a, (size_t)(k + u) * ro.lda) : 0.;
#pragma unroll
        for (int u = 0; u < SW_GRP; u++) {
          const int kk = min(k + u, sl.k1 - 1);
#pragma unroll
          for (int c = 0; c < NR; c++) acc[c] += t[u] * ro.x[kk + c * LDV];
        }
      }
    }
    if (P == 1) {
      if (i < M)
        for (int c = 0; c < nrhs; c++) apply_op(ro.o + c * LDV, acc[c], ro.op);
      __syncthreads();
      continue;
    }
    // partials: s_p[(part * NR + c) * M64 + i]   (P * M64 == 256)
    if (part < P)
      for (int c = 0; c < nrhs; c++) s_p[(part * NR + c) * M64 + i] = acc[c];
    __syncthreads();
    if (part == 0 && i < M)
      for (int c = 0; c < nrhs; c++) {
        double v = 0.;
        for (int q = 0; q < P; q++) v += s_p[(q * NR + c) * M64 + i];
        apply_op(ro.o + c * LDV, v, ro.op);
      }
    __syncthreads();
  }
}
// out[i] (op)= sum_{k < K} A[i + k lda] x[k],  i < M
template <int NR, int LDV>
__device__ __forceinline__ void gemv_n(const double* __restrict__ A, int lda, int M, int K, const double* x, double* out, int nrhs, int op,
                                       double* s_p, const Pre& pre, bool use_pre) {
  gemv_rows<NR, LDV>(M, [=](int i) { return RowOp{A + i, lda, K, x, out + i, op}; }, nrhs, s_p, pre, use_pre);
}
template <int NR, int LDV>
__device__ __forceinline__ void gemv_n(const double* __restrict__ A, int lda, int M, int K, const double* x, double* out, int nrhs, int op,
                                       double* s_p) {
  Pre none;
  gemv_n<NR, LDV>(A, lda, M, K, x, out, nrhs, op, s_p, none, false);
}
__device__ __forceinline__ void gemv_n_prefetch(const double* __restrict__ A, int lda, int M, int K, Pre& pre) {
  rows_prefetch(M, [=](int i) { return RowOp{A + i, lda, K, nullptr, nullptr, 0}; }, pre);
}

// The same for FEW rows (M <= 32: the interpolation block of a leaf, r x (m - r) with r ~ 10 - 20 and m - r ~ 180): the K range
// is split over all 256 / M thread groups, every thread's slice is at most a few groups of sixteen loads in flight -- gemv_rows
// gives such a block four K partitions of 45 columns each on a quarter of the threads, six dependent memory round trips
// (the leaves' V^T x: 7 -> 4.5 us).
template <int NR, int LDV>
__device__ __forceinline__ void gemv_n_few(const double* __restrict__ A, int lda, int M, int K, const double* x, double* out, int nrhs, int op,
                                           double* s_p) {
  const int tid = threadIdx.x;
  const int P = SW_T / M;                        // >= 8
  const int prt = tid / M, i = tid - prt * M;
  const bool act = prt < P;
  const int Kc = (K + P - 1) / P, k0 = prt * Kc, k1 = min(K, k0 + Kc);
  double acc[NR] = {};
  if (act)
    for (int k = k0; k < k1; k += 16) {
      double t[16];
#pragma unroll
      for (int u = 0; u < 16; u++) t[u] = hssk_gload(A + i, (size_t)min(k + u, k1 - 1) * lda);
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const double mk = k + u < k1 ? 1. : 0.;
#pragma unroll
        for (int c = 0; c < NR; c++) acc[c] += mk * t[u] * x[min(k + u, k1 - 1) + c * LDV];
      }
    }
  // partials: s_p[c * SW_T + prt * M + i]   (P * M <= SW_T)
  if (act)
    for (int c = 0; c < nrhs; c++) s_p[c * SW_T + prt * M + i] = acc[c];
  __syncthreads();
  if (tid < M)
    for (int c = 0; c < nrhs; c++) {
      double v = 0.;
      for (int q = 0; q < P; q++) v += s_p[c * SW_T + q * M + tid];
      apply_op(out + tid + c * LDV, v, op);
    }
  __syncthreads();
}

// pull `count` doubles at p towards this XCD's L2 (one load per 128-byte line) while the workgroup still waits for its
// dependencies: the dependent chain of a node then runs on L2 hits.  The values are folded into `sink` (see keep()).
__device__ __forceinline__ void touch(const double* p, size_t count, double& sink) {
  // (eight lines per thread in flight: a loop that folds each value into `sink` as it arrives waits out one memory round trip
  //  per line)
  if (p)
    for (size_t e = (size_t)threadIdx.x * 16; e < count; e += (size_t)SW_T * 16 * 8) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const size_t x = e + (size_t)u * SW_T * 16; t[u] = x < count ? hssk_gload(p, x) : 0.; }
#pragma unroll
      for (int u = 0; u < 8; u++) sink += t[u];
    }
}
__device__ __forceinline__ void keep(double sink, double* s_p) {
  if (sink == 1.234567e300) s_p[0] = sink;   // never true: keeps the prefetch loads alive
}

// out[j] (op)= sum_{i < K} A[i + j lda] x[i],  j < N: columns contiguous.  Four adjacent lanes share a column (each
// a contiguous quarter of it: whole cache lines per lane), 64 columns per pass; quad reduction by shuffles.
// Contains wave collectives and a barrier: every thread must call.  x and out must not alias.
// the thread's slice of the first pass (columns [0, 64)) of gemv_t, loaded ahead
__device__ __forceinline__ void gemv_t_prefetch(const double* __restrict__ A, int lda, int K, int N, Pre& pre) {
  const int tid = threadIdx.x;
  const int part = tid & 3, j = tid >> 2;
  const int Kc = (K + 3) >> 2;
  const int i0 = part * Kc, i1 = min(K, i0 + Kc);
#pragma unroll
  for (int u = 0; u < SW_PRE; u++) pre.v[u] = (j < N && i0 + u < i1) ? hssk_gload(A + (size_t)j * lda, (size_t)(i0 + u)) : 0.;
}
template <int NR, int LDV>
__device__ __forceinline__ void gemv_t(const double* __restrict__ A, int lda, int K, int N, const double* x, double* out, int nrhs, int op,
                                       const Pre& pre, bool use_pre) {
  const int tid = threadIdx.x;
  const int part = tid & 3, cj = tid >> 2;
  const int Kc = (K + 3) >> 2;
  const int i0 = part * Kc, i1 = min(K, i0 + Kc);
  for (int j0 = 0; j0 < N; j0 += SW_T / 4) {
    const int j = j0 + cj;
    double acc[NR] = {};
    if (j < N) {
      const double* a = A + (size_t)j * lda;
      int i = i0;
      if (use_pre && j0 == 0) {
#pragma unroll
        for (int u = 0; u < SW_PRE; u++) {
          if (i0 + u < i1) {
            const double t = pre.v[u];
#pragma unroll
            for (int c = 0; c < NR; c++) acc[c] += t * x[i0 + u + c * LDV];
          }
        }
        i = min(i1, i0 + SW_PRE);
      }
      for (; i < i1; i += SW_GRP) {
        double t[SW_GRP];
#pragma unroll
        for (int u = 0; u < SW_GRP; u++) t[u] = i + u < i1 ? hssk_gload(a, (size_t)(i + u)) : 0.;
#pragma unroll
        for (int u = 0; u < SW_GRP; u++) {
          const int ii = min(i + u, i1 - 1);
#pragma unroll
          for (int c = 0; c < NR; c++) acc[c] += t[u] * x[ii + c * LDV];
        }
      }
    }
    for (int c = 0; c < nrhs; c++) {
      double v = acc[c];
      v += hssk_shfl_xor(v, 1);
      v += hssk_shfl_xor(v, 2);
      if (part == 0 && j < N) apply_op(out + j + c * LDV, v, op);
    }
  }
  __syncthreads();
}

template <int NR, int LDV>
__device__ __forceinline__ void gemv_t(const double* __restrict__ A, int lda, int K, int N, const double* x, double* out, int nrhs, int op) {
  Pre none;
  gemv_t<NR, LDV>(A, lda, K, N, x, out, nrhs, op, none, false);
}

// The same for LONG columns (K >= 33 rows: the leaves' D^T x, the rows-below updates of the forward substitution): a WAVE per
// column, lanes along its rows -- every load instruction is 512 contiguous bytes, where gemv_t's four lanes per column put the
// 64 lanes of a load on 16 columns, i.e. on 64 different cache lines; the column sums over the DPP network (hssk_wave_sum).  A
// wave keeps four columns x up to four 64-row pieces in flight; x sits in registers.  K <= 256.  Contains wave collectives and
// a barrier: every thread must call.  x and out must not alias.
template <int NR, int LDV>
__device__ __forceinline__ void gemv_t_wave(const double* __restrict__ A, int lda, int K, int N, const double* x, double* out, int nrhs, int op) {
  const int lane = threadIdx.x & 63, wave = hssk_uniform((int)(threadIdx.x >> 6));
  constexpr int NW = SW_T / 64, U = 4;
  const int nc = (K + 63) >> 6;   // 64-row pieces of a column (<= 4)
  double xr[4][NR];
#pragma unroll
  for (int c = 0; c < 4; c++)
#pragma unroll
    for (int q = 0; q < NR; q++) xr[c][q] = (c < nc && lane + 64 * c < K) ? x[lane + 64 * c + q * LDV] : 0.;
  for (int j0 = wave * U; j0 < N; j0 += NW * U) {
    double t[U][4];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int j = min(j0 + u, N - 1);
#pragma unroll
      for (int c = 0; c < 4; c++) t[u][c] = c < nc ? hssk_gload(A + (size_t)j * lda, (size_t)min(lane + 64 * c, K - 1)) : 0.;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      double acc[NR] = {};
#pragma unroll
      for (int c = 0; c < 4; c++)
#pragma unroll
        for (int q = 0; q < NR; q++) acc[q] += t[u][c] * xr[c][q];   // (rows beyond K: xr is zero there)
      for (int q = 0; q < nrhs; q++) {
        const double v = hssk_wave_sum(acc[q]);
        if (lane == 0 && j0 + u < N) apply_op(out + j0 + u + q * LDV, v, op);
      }
    }
  }
  __syncthreads();
}

// ---- forward ULV sweep ---------------------------------------------------------------------------------------------
// Right-hand sides beyond SW_NR: blockIdx.y walks groups of SW_NR columns.  The groups are independent chains through the
// tree that run side by side (a 2-D grid is dispatched x-fastest, so within a group the index order still holds); each
// group streams the blocks again, which is what bounds it (nrhs = 64: 16 x the bytes of one group).
// (few groups: side by side along blockIdx.y as described; many -- the inner levels of the hybrid many-right-hand-side path,
// hss_apply.cpp / hss_solve.cpp --: ONE workgroup per node walks the groups in turn.  Side by side, sixteen groups of a
// 9-level tree cost sixteen latency chains one after the other -- a group's workgroups hold the slots while they wait --:
// 0.9 ms for the inner levels of N = 1e5 at nrhs = 64.  In turn, the chain is paid once and the levels pipeline: a parent
// works on group g while its children are on g + 1.)
// Chain block of an inner node of the forward sweep (hssk_sweep_fwd_desc::G): what the PARENT waits for -- ft1 and z -- is a
// linear function of what the children hand over, [ft1; z] = G [f; zc], G (r + rv) x (m + mv).  As written above it, the step is
// five dependent passes (the coupling products, X^T, the substitution, [WQ; Vt0^T]) with their barriers and partial sums: 6 - 8 us
// per tree level at N = 1e5, eight levels deep, against 1.6 - 2.4 us for the hand-off itself.  With G the chain is ONE pass whose
// matrix elements sit in registers before the children's vectors exist (SW_CH per thread: the rows over 256 / (r + rv) K
// partitions); y, which only the backward sweep reads, follows behind the hand-off through the node's blocks.
constexpr int SW_CH = 56;
__host__ __device__ inline bool chain_shape_ok(int m, int r, int mv, int rv, int ldv) {
  const int M = r + rv, K = m + mv;
  if (M <= 0 || M > SW_T || K > ldv || m <= r) return false;
  const int P = SW_T / M;
  return (K + P - 1) / P <= SW_CH;
}

template <int NR>
__device__ __forceinline__ int rhs_group(int nrhs_total, int group, int& c0) {
  c0 = group * NR;
  return min(NR, nrhs_total - c0);
}

template <int NR, int LDV, bool CHAIN>
__device__ __forceinline__ void ulv_fwd_body(const hssk_sweep_fwd_desc* __restrict__ descs, int nrhs_total, int* err, int group) {
  // (LDS carved from the launch's dynamic allocation: the NR = 16 instantiation needs more than the 64 KB static limit)
  HSSK_DYN_SHARED(double, s_dyn);
  double* s_f = s_dyn;                 // f, later the block right-hand side of the substitution
  double* s_y = s_f + LDV * NR;        // zc(permV[rv:]) first, then y
  double* s_a = s_y + LDV * NR;        // stacked children z (inner nodes)
  double* s_t = s_a + LDV * NR;        // ft1 (root: block right-hand side)
  double* s_z = s_t + LDV * NR;        // z
  double* s_p = s_z + LDV * NR;        // gemv partials (SW_T * NR)
  int* s_piv = (int*)(s_p + SW_T * NR);   // LDV ints
  hssk_sweep_fwd_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x;
  const int m = p.m, r = p.r, q = m - r, rv = p.rv, mv = p.mv;
  int c0;
  const int nrhs = rhs_group<NR>(nrhs_total, group, c0);
  if (c0) {   // this group's columns of every vector
    p.fsrc += (size_t)c0 * p.ldf;
    if (p.zc) p.zc += (size_t)c0 * p.ldz_in;
    if (p.ft1) p.ft1 += (size_t)c0 * p.ldp;
    if (p.y) p.y += (size_t)c0 * q;
    if (p.z) p.z += (size_t)c0 * p.ldz;
    if (p.xroot) p.xroot += (size_t)c0 * p.ldxr;
  }
  const bool inner = p.B01 != nullptr;
  // ---- before the dependencies arrive: permutations into registers, the node's blocks towards L2
  int pu = 0, pv = 0;
  if (p.LU) { if (tid < m) s_piv[tid] = p.piv[tid]; }
  else {
    if (tid < m) pu = p.permU[tid];
    if (inner && tid < mv) pv = p.permV[tid];
  }
  // a node that waits for its children loads ITS OWN slices of the blocks into registers first (every thread knows which
  // elements its passes below will read): once the vectors arrive, the dependent chain of the node runs on registers and LDS
  // alone -- with the blocks merely pulled towards L2 every pass still paid an L2 round trip, ~14 us per tree level.
  const bool zpart = inner && !p.LU && rv > 0;
  const int mz = (zpart && mv > rv) ? rv : 0;
  const int rU0 = p.rU0;
  auto rows1 = [=](int i) {   // f(0:rU0) -= B01 zc(rV0:), f(rU0:) -= B10 zc(0:rV0)  and  z += XV zc(permV[rv:])
    if (i < rU0) return RowOp{p.B01 + i, max(rU0, 1), p.rV1, s_a + p.rV0, s_f + i, OP_SUB};
    if (i < m) return RowOp{p.B10 + (i - rU0), max(p.rU1, 1), p.rV0, s_a, s_f + i, OP_SUB};
    return RowOp{p.XV + (i - m), rv, mv - rv, s_y, s_z + (i - m), OP_ADD};
  };
  auto rows4 = [=](int i) {   // ft1 -= WQ y  and  z += Vt0^T y
    if (i < r) return RowOp{p.WQ + i, r, q, s_y, s_t + i, OP_SUB};
    return RowOp{p.Vt0T + (i - r), rv, q, s_y, s_z + (i - r), OP_ADD};
  };
  // (CHAIN: an instantiation of its own, taken when every inner node of the launch has a block -- the slices of a block and the
  //  four prefetch slices of the plain step do not share registers, whatever the source says: 241 against 153)
  const bool chain = CHAIN && inner && !p.LU && p.G != nullptr && chain_shape_ok(m, r, mv, rv, LDV);
  const bool pf = !CHAIN && inner && !p.LU;
  // the chain block: row ci of G, columns [ck0, ck1)
  const int cM = chain ? r + rv : 1, cP = SW_T / cM, cpart = tid / cM, ci = tid - cpart * cM;
  const bool cact = chain && cpart < cP;
  const int cKc = (m + mv + cP - 1) / cP, ck0 = cpart * cKc, ck1 = min(m + mv, ck0 + cKc);
  Pre pre1, pre2, pre3, pre4;
  double gpre[CHAIN ? SW_CH : 1];
  if (CHAIN && chain) {
#pragma unroll
    for (int u = 0; u < SW_CH; u++) gpre[CHAIN ? u : 0] = (cact && ck0 + u < ck1) ? hssk_gload(p.G, (size_t)ci + (size_t)(ck0 + u) * p.ldg) : 0.;
  }
  if (pf) {
    rows_prefetch(m + mz, rows1, pre1);
    if (q > 0) {
      if (r > 0) gemv_t_prefetch(p.XU, r, r, q, pre2);
      gemv_n_prefetch(p.Tinv, SW_NB, min(SW_NB, q), min(SW_NB, q), pre3);
      rows_prefetch(r + rv, rows4, pre4);
    }
  }
  if (p.wait0 >= 0 || p.wait1 >= 0) {
    double sink = 0.;
    if (p.LU) {
      if (inner) { touch(p.B01, (size_t)p.rU0 * p.rV1, sink); touch(p.B10, (size_t)p.rU1 * p.rV0, sink); }
      touch(p.LU, (size_t)m * m, sink);
      touch(p.TinvL, (size_t)((m + SW_NB - 1) / SW_NB) * SW_NB * SW_NB, sink);
      touch(p.TinvU, (size_t)((m + SW_NB - 1) / SW_NB) * SW_NB * SW_NB, sink);
    } else if (q > SW_NB) {   // (the blocks behind the first 64-row step are not in the register slices)
      touch(p.Tinv + SW_NB * SW_NB, (size_t)((q - 1) / SW_NB) * SW_NB * SW_NB, sink);
      touch(p.Rlq, (size_t)m * q, sink);
    }
    keep(sink, s_p);
  }
  // ---- f = rhs rows (leaf) or [ft1_0; ft1_1] (inner: handed over by the children); zc = stacked children z
  for (int e = tid; e < m * nrhs; e += SW_T) {
    const int i = e % m, c = e / m;
    const double v = inner ? sweep_take(p.fsrc, i + (size_t)c * p.ldf, err) : hssk_gload(p.fsrc, i + (size_t)c * p.ldf);
    s_f[i + c * LDV] = v;
    if (chain) s_t[i + c * LDV] = v;   // ([f; zc] in one piece for the chain block)
  }
  if (inner)
    for (int e = tid; e < mv * nrhs; e += SW_T) {
      const double v = sweep_take(p.zc, (e % mv) + (size_t)(e / mv) * p.ldz_in, err);
      s_a[(e % mv) + (e / mv) * LDV] = v;
      if (chain) s_t[m + (e % mv) + (e / mv) * LDV] = v;
    }
  __syncthreads();
  if (CHAIN && chain) {
    // ---- [ft1; z] = G [f; zc]: the hand-off to the parent, one pass on registers and LDS
    double acc[NR] = {};
#pragma unroll
    for (int u = 0; u < SW_CH; u++) {
      const int k = min(ck0 + u, m + mv - 1);   // (elements beyond the slice are zeros)
#pragma unroll
      for (int c = 0; c < NR; c++) acc[c] += gpre[CHAIN ? u : 0] * s_t[k + c * LDV];
    }
    if (cact)
      for (int c = 0; c < nrhs; c++) s_p[(cpart * NR + c) * cM + ci] = acc[c];
    __syncthreads();
    if (tid < cM)
      for (int c = 0; c < nrhs; c++) {
        double v = 0.;
        for (int qq = 0; qq < cP; qq++) v += s_p[(qq * NR + c) * cM + tid];
        if (tid < r) hssk_cstore(p.ft1, tid + (size_t)c * p.ldp, v);
        else hssk_cstore(p.z, (tid - r) + (size_t)c * p.ldz, v);
      }
    __syncthreads();   // (s_p and s_t are written again below)
  }
  if (inner) {
    if (zpart) {
      // s_z (= s_t2) <- zc(permV[0:rv]);  s_y <- zc(permV[rv:])
      if (tid < mv)
        for (int c = 0; c < nrhs; c++) {
          const double v = s_a[pv + c * LDV];
          if (tid < rv) s_z[tid + c * LDV] = v;
          else s_y[(tid - rv) + c * LDV] = v;
        }
      __syncthreads();
    }
    // one pass: f(0:rU0) -= B01 zc(rV0:), f(rU0:) -= B10 zc(0:rV0)   and   z += XV zc(permV[rv:])   (XV is rv x (mv - rv))
    gemv_rows<NR, LDV>(m + mz, rows1, nrhs, s_p, pre1, pf);
  }
  if (p.LU) {
    // ---- root: x = U^{-1} L^{-1} P f   (DenseMatrix::solve / getrs, solve.hpp:133-135), block substitution with the
    // inverted 64 x 64 diagonal blocks
    if (tid < nrhs)
      for (int i = 0; i < m; i++) {
        const int pi = s_piv[i];
        if (pi != i) { const double a = s_f[i + tid * LDV]; s_f[i + tid * LDV] = s_f[pi + tid * LDV]; s_f[pi + tid * LDV] = a; }
      }
    __syncthreads();
    for (int b0 = 0, blk = 0; b0 < m; b0 += SW_NB, blk++) {
      const int nb = min(SW_NB, m - b0);
      for (int e = tid; e < nb * nrhs; e += SW_T) s_t[(e % nb) + (e / nb) * LDV] = s_f[b0 + (e % nb) + (e / nb) * LDV];
      __syncthreads();
      gemv_n<NR, LDV>(p.TinvL + (size_t)blk * SW_NB * SW_NB, SW_NB, nb, nb, s_t, s_f + b0, nrhs, OP_SET, s_p);
      const int rest = m - b0 - nb;
      if (rest > 0) {
        for (int e = tid; e < nb * nrhs; e += SW_T) s_t[(e % nb) + (e / nb) * LDV] = s_f[b0 + (e % nb) + (e / nb) * LDV];
        __syncthreads();
        gemv_n<NR, LDV>(p.LU + (b0 + nb) + (size_t)b0 * m, m, rest, nb, s_t, s_f + b0 + nb, nrhs, OP_SUB, s_p);
      }
    }
    for (int blk = (m - 1) / SW_NB; blk >= 0; blk--) {
      const int b0 = blk * SW_NB, nb = min(SW_NB, m - b0);
      for (int e = tid; e < nb * nrhs; e += SW_T) s_t[(e % nb) + (e / nb) * LDV] = s_f[b0 + (e % nb) + (e / nb) * LDV];
      __syncthreads();
      gemv_n<NR, LDV>(p.TinvU + (size_t)blk * SW_NB * SW_NB, SW_NB, nb, nb, s_t, s_f + b0, nrhs, OP_SET, s_p);
      if (b0 > 0) {
        for (int e = tid; e < nb * nrhs; e += SW_T) s_t[(e % nb) + (e / nb) * LDV] = s_f[b0 + (e % nb) + (e / nb) * LDV];
        __syncthreads();
        gemv_n<NR, LDV>(p.LU + (size_t)b0 * m, m, b0, nb, s_t, s_f, nrhs, OP_SUB, s_p);
      }
    }
    for (int e = tid; e < m * nrhs; e += SW_T) hssk_gstore(p.xroot, (e % m) + (size_t)(e / m) * p.ldxr, s_f[(e % m) + (e / m) * LDV]);
    return;
  }
  // ---- ft1 = f(perm[0:r]) -> s_t, y = f(perm[r:]) -> s_y
  if (tid < m)
    for (int c = 0; c < nrhs; c++) {
      const double v = s_f[pu + c * LDV];
      if (tid < r) s_t[tid + c * LDV] = v;
      else s_y[(tid - r) + c * LDV] = v;
    }
  if (!inner)
    for (int e = tid; e < rv * nrhs; e += SW_T) s_z[(e % rv) + (e / rv) * LDV] = 0.;
  __syncthreads();
  if (q > 0) {
    // ---- y -= X^T ft1   (X is r x q, column k contiguous)
    if (r > 0) gemv_t<NR, LDV>(p.XU, r, r, q, s_t, s_y, nrhs, OP_SUB, pre2, pf);
    // ---- y <- R~^{-T} y on 64-row blocks: y_b = Linv_b y_b, then rows below -= R~(b, below)^T y_b
    for (int b0 = 0, blk = 0; b0 < q; b0 += SW_NB, blk++) {
      const int nb = min(SW_NB, q - b0);
      for (int e = tid; e < nb * nrhs; e += SW_T) s_f[(e % nb) + (e / nb) * LDV] = s_y[b0 + (e % nb) + (e / nb) * LDV];
      __syncthreads();
      gemv_n<NR, LDV>(p.Tinv + (size_t)blk * SW_NB * SW_NB, SW_NB, nb, nb, s_f, s_y + b0, nrhs, OP_SET, s_p, pre3, pf && blk == 0);
      const int rest = q - b0 - nb;
      if (rest > 0) {
        // column k of R~ (rows b0 .. b0+nb contiguous) for k > b0 + nb;  x = y_b (now final) copied to s_f
        for (int e = tid; e < nb * nrhs; e += SW_T) s_f[(e % nb) + (e / nb) * LDV] = s_y[b0 + (e % nb) + (e / nb) * LDV];
        __syncthreads();
        if (NR == 1 && nb > 32) gemv_t_wave<NR, LDV>(p.Rlq + b0 + (size_t)(b0 + nb) * m, m, nb, rest, s_f, s_y + b0 + nb, nrhs, OP_SUB);
        else gemv_t<NR, LDV>(p.Rlq + b0 + (size_t)(b0 + nb) * m, m, nb, rest, s_f, s_y + b0 + nb, nrhs, OP_SUB);
      }
    }
    for (int e = tid; e < q * nrhs; e += SW_T) hssk_gstore(p.y, (e % q) + (size_t)(e / q) * q, s_y[(e % q) + (e / q) * LDV]);
    // ---- one pass over [WQ; Vt0^T] (both (.) x q, rows contiguous):  ft1 -= WQ y   and   z += Vt0^T y
    gemv_rows<NR, LDV>(r + rv, rows4, nrhs, s_p, pre4, pf);
  }
  if (chain) return;   // (ft1 and z left with the chain block)
  for (int e = tid; e < r * nrhs; e += SW_T) hssk_cstore(p.ft1, (e % r) + (size_t)(e / r) * p.ldp, s_t[(e % r) + (e / r) * LDV]);
  for (int e = tid; e < rv * nrhs; e += SW_T) hssk_cstore(p.z, (e % rv) + (size_t)(e / rv) * p.ldz, s_z[(e % rv) + (e / rv) * LDV]);
}

// LOOP: the groups of right-hand sides in turn inside the workgroup (many groups); otherwise one group per workgroup along
// blockIdx.y -- the few-right-hand-side form keeps the straight-line body (the loop and its closing barrier cost the
// single-vector solve 15 percent)
template <int NR, int LDV, bool LOOP, bool CHAIN = false>
__global__ __launch_bounds__(SW_T) void ulv_fwd_sweep_kernel(const hssk_sweep_fwd_desc* __restrict__ descs, int nrhs_total, int* err) {
  if (!LOOP) { ulv_fwd_body<NR, LDV, CHAIN>(descs, nrhs_total, err, (int)blockIdx.y); return; }
  for (int g = blockIdx.y; g * NR < nrhs_total; g += gridDim.y) {
    ulv_fwd_body<NR, LDV, CHAIN>(descs, nrhs_total, err, g);
    __syncthreads();   // (the LDS vectors are reused by the next group)
  }
}

// ---- backward ULV sweep:  x_c = Q~(:, 0:q) y + Q~(:, q:) xpart ; m == r (nothing eliminated): x_c = xpart ------------
template <int NR, int LDV>
__device__ __forceinline__ void ulv_bwd_body(const hssk_sweep_bwd_desc* __restrict__ descs, int nrhs_total, int* err, int group) {
  HSSK_DYN_SHARED(double, s_dyn);
  double* s_v = s_dyn;                 // [y; xpart]
  double* s_o = s_v + LDV * NR;
  double* s_p = s_o + LDV * NR;
  hssk_sweep_bwd_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x;
  const int m = p.m, r = p.r, q = m - r;
  int c0;
  const int nrhs = rhs_group<NR>(nrhs_total, group, c0);
  if (c0) {
    if (p.y) p.y += (size_t)c0 * q;
    p.xpart += (size_t)c0 * p.ldx;
    p.out += (size_t)c0 * p.ldo;
  }
  // the parent-independent part first: s_o = Q~(:, 0:q) y; Q~(:, q:) towards L2
  for (int e = tid; e < q * nrhs; e += SW_T) s_v[(e % q) + (e / q) * LDV] = hssk_gload(p.y, (e % q) + (size_t)(e / q) * q);
  __syncthreads();
  // (this thread's slice of Q~(:, q:) goes to registers before anything else: it is consumed after the wait)
  const bool pf = q > 0 && r > 0 && p.wait0 >= 0;
  Pre pre;
  if (pf) gemv_n_prefetch(p.Qt + (size_t)q * m, m, m, r, pre);
  if (q > 0) {
    gemv_n<NR, LDV>(p.Qt, m, m, q, s_v, s_o, nrhs, OP_SET, s_p);
    if (p.wait0 >= 0 && r > SW_PRE) { double sink = 0.; touch(p.Qt + (size_t)q * m, (size_t)m * r, sink); keep(sink, s_p); }
  }
  for (int e = tid; e < r * nrhs; e += SW_T) s_v[q + (e % r) + (e / r) * LDV] = sweep_take(p.xpart, (e % r) + (size_t)(e / r) * p.ldx, err);
  __syncthreads();
  if (q > 0) {
    if (r > 0) gemv_n<NR, LDV>(p.Qt + (size_t)q * m, m, m, r, s_v + q, s_o, nrhs, OP_ADD, s_p, pre, pf);
    for (int e = tid; e < m * nrhs; e += SW_T) hssk_cstore(p.out, (e % m) + (size_t)(e / m) * p.ldo, s_o[(e % m) + (e / m) * LDV]);
  } else {
    for (int e = tid; e < m * nrhs; e += SW_T) hssk_cstore(p.out, (e % m) + (size_t)(e / m) * p.ldo, s_v[(e % m) + (e / m) * LDV]);
  }
}

// LOOP: the groups of right-hand sides in turn inside the workgroup (many groups); otherwise one group per workgroup along
// blockIdx.y -- the few-right-hand-side form keeps the straight-line body (the loop and its closing barrier cost the
// single-vector solve 15 percent)
template <int NR, int LDV, bool LOOP>
__global__ __launch_bounds__(SW_T) void ulv_bwd_sweep_kernel(const hssk_sweep_bwd_desc* __restrict__ descs, int nrhs_total, int* err) {
  if (!LOOP) { ulv_bwd_body<NR, LDV>(descs, nrhs_total, err, (int)blockIdx.y); return; }
  for (int g = blockIdx.y; g * NR < nrhs_total; g += gridDim.y) {
    ulv_bwd_body<NR, LDV>(descs, nrhs_total, err, g);
    __syncthreads();
  }
}

// ---- mat-vec: up-sweep nodes [0, nup) then down-sweep nodes [nup, nup + ndown) in one launch -------------------------------
template <int NR, int LDV>
__device__ __forceinline__ void apply_body(const hssk_apply_up_desc* __restrict__ ups, int nup,
                                           const hssk_apply_down_desc* __restrict__ downs, int nrhs_total, int* err, int group) {
  HSSK_DYN_SHARED(double, s_dyn);
  double* s_x = s_dyn;
  double* s_g = s_x + LDV * NR;
  double* s_o = s_g + LDV * NR;
  double* s_p = s_o + LDV * NR;
  const int tid = threadIdx.x;
  int c0;
  const int nrhs = rhs_group<NR>(nrhs_total, group, c0);
  if ((int)blockIdx.x < nup) {
    // tmp1 = V^H src = src(perm[0:r]) + X src(perm[r:])   (X is r x (m - r), rows contiguous)
    hssk_apply_up_desc p = ups[blockIdx.x];
    p.src += (size_t)c0 * p.lds;
    p.dst += (size_t)c0 * p.ldd;
    const int m = p.m, r = p.r;
    const int pk = tid < m ? p.perm[tid] : 0;
    const bool handed = p.inner != 0;   // inner node: the children's results; leaf: rows of x
    const bool pf = handed && m > r && r > 0;   // (slices of the node's blocks to registers before the wait, see the solve sweeps)
    Pre pre;
    if (pf) gemv_n_prefetch(p.X, r, r, m - r, pre);
    if (tid < m)
      for (int c = 0; c < nrhs; c++) {
        const double v = handed ? sweep_take(p.src, pk + (size_t)c * p.lds, err) : hssk_gload(p.src, pk + (size_t)c * p.lds);
        if (tid < r) s_o[tid + c * LDV] = v;
        else s_g[(tid - r) + c * LDV] = v;
      }
    __syncthreads();
    if (!handed && m > r && r > 0 && r <= 32) gemv_n_few<NR, LDV>(p.X, r, r, m - r, s_g, s_o, nrhs, OP_ADD, s_p);
    else if (m > r && r > 0) gemv_n<NR, LDV>(p.X, r, r, m - r, s_g, s_o, nrhs, OP_ADD, s_p, pre, pf);
    for (int e = tid; e < r * nrhs; e += SW_T) hssk_cstore(p.dst, (e % r) + (size_t)(e / r) * p.ldd, s_o[(e % r) + (e / r) * LDV]);
    return;
  }
  hssk_apply_down_desc p = downs[blockIdx.x - nup];
  if (c0) {
    if (p.tmp2) p.tmp2 += (size_t)c0 * p.ld2;
    if (p.x) p.x += (size_t)c0 * p.ldx;
    if (p.t1) p.t1 += (size_t)c0 * p.ldt1;
    p.out += (size_t)c0 * p.ldo;
  }
  const int mo = p.mo, ro = p.ro;
  const bool expand = p.tmp2 && ro > 0;
  const int pk = (expand && tid < mo) ? p.perm[tid] : 0;
  // (slices of the blocks that are consumed after a wait go to registers first, see the solve sweeps)
  Pre preX;
  const bool pfX = expand && mo > ro && p.wait0 >= 0;
  if (pfX) gemv_t_prefetch(p.X, ro, ro, mo - ro, preX);
  if (p.D) {
    // ---- leaf: y = op(D) x + beta y + U tmp2.  op(D) x does not depend on the tree: it runs before the wait.
    const int m = p.m;
    for (int e = tid; e < m * nrhs; e += SW_T) s_x[(e % m) + (e / m) * LDV] = hssk_gload(p.x, (e % m) + (size_t)(e / m) * p.ldx);
    __syncthreads();
    if (p.trans && m > 32) gemv_t_wave<NR, LDV>(p.D, m, m, m, s_x, s_o, nrhs, OP_SET);
    else if (p.trans) gemv_t<NR, LDV>(p.D, m, m, m, s_x, s_o, nrhs, OP_SET);
    else gemv_n<NR, LDV>(p.D, m, m, m, s_x, s_o, nrhs, OP_SET, s_p);
    if (p.beta != 0.) {
      for (int e = tid; e < m * nrhs; e += SW_T) s_o[(e % m) + (e / m) * LDV] += p.beta * hssk_gload(p.out, (e % m) + (size_t)(e / m) * p.ldo);
      __syncthreads();
    }
  } else {
    // ---- inner: t = [B01 t1_1; B10 t1_0]  (transposed: [B10^T t1_1; B01^T t1_0]); t1 = the children's up-sweep results
    const int nt1 = p.ri_a + p.ri_b, nto = p.ro_a + p.ro_b;
    const int ro_a = p.ro_a;
    auto rowsB = [=](int i) {   // B01 is ro_a x ri_b, B10 is ro_b x ri_a: one pass over the stacked rows
      if (i < ro_a) return RowOp{p.B01 + i, max(ro_a, 1), p.ri_b, s_x + p.ri_a, s_o + i, OP_SET};
      return RowOp{p.B10 + (i - ro_a), max(p.ro_b, 1), p.ri_a, s_x, s_o + i, OP_SET};
    };
    Pre preB, preB2;
    if (!p.trans) rows_prefetch(nto, rowsB, preB);
    else {
      if (p.ro_a > 0 && p.ri_b > 0) gemv_t_prefetch(p.B10, p.ri_b, p.ri_b, p.ro_a, preB);
      if (p.ro_b > 0 && p.ri_a > 0) gemv_t_prefetch(p.B01, p.ri_a, p.ri_a, p.ro_b, preB2);
    }
    for (int e = tid; e < nt1 * nrhs; e += SW_T) s_x[(e % nt1) + (e / nt1) * LDV] = sweep_take(p.t1, (e % nt1) + (size_t)(e / nt1) * p.ldt1, err);
    for (int e = tid; e < nto * nrhs; e += SW_T) s_o[(e % nto) + (e / nto) * LDV] = 0.;
    __syncthreads();
    if (!p.trans) gemv_rows<NR, LDV>(nto, rowsB, nrhs, s_p, preB, true);
    else {            // B10 is ri_b x ro_a, B01 is ri_a x ro_b
      if (p.ro_a > 0 && p.ri_b > 0) gemv_t<NR, LDV>(p.B10, p.ri_b, p.ri_b, p.ro_a, s_x + p.ri_a, s_o, nrhs, OP_SET, preB, true);
      if (p.ro_b > 0 && p.ri_a > 0) gemv_t<NR, LDV>(p.B01, p.ri_a, p.ri_a, p.ro_b, s_x, s_o + p.ro_a, nrhs, OP_SET, preB2, true);
    }
  }
  // ---- + U tmp2:  out(perm[k]) += tmp2(k), k < ro ;  out(perm[ro + j]) += sum_k X(k, j) tmp2(k)   (X is ro x (mo - ro))
  if (expand) {
    if (p.wait0 >= 0 && mo - ro > SW_T / 4) { double sink = 0.; touch(p.X, (size_t)ro * (mo - ro), sink); keep(sink, s_p); }
    for (int e = tid; e < ro * nrhs; e += SW_T) s_x[(e % ro) + (e / ro) * LDV] = sweep_take(p.tmp2, (e % ro) + (size_t)(e / ro) * p.ld2, err);
    __syncthreads();
    if (mo > ro) gemv_t<NR, LDV>(p.X, ro, ro, mo - ro, s_x, s_g, nrhs, OP_SET, preX, pfX);
    if (tid < mo)
      for (int c = 0; c < nrhs; c++) s_o[pk + c * LDV] += tid < ro ? s_x[tid + c * LDV] : s_g[(tid - ro) + c * LDV];
    __syncthreads();
  }
  const int mout = p.D ? p.m : p.ro_a + p.ro_b;
  for (int e = tid; e < mout * nrhs; e += SW_T) hssk_cstore(p.out, (e % mout) + (size_t)(e / mout) * p.ldo, s_o[(e % mout) + (e / mout) * LDV]);
}

template <int NR, int LDV, bool LOOP>
__global__ __launch_bounds__(SW_T) void apply_sweep_kernel(const hssk_apply_up_desc* __restrict__ ups, int nup,
                                                           const hssk_apply_down_desc* __restrict__ downs, int nrhs_total, int* err) {
  if (!LOOP) { apply_body<NR, LDV>(ups, nup, downs, nrhs_total, err, (int)blockIdx.y); return; }
  for (int g = blockIdx.y; g * NR < nrhs_total; g += gridDim.y) {
    apply_body<NR, LDV>(ups, nup, downs, nrhs_total, err, g);
    __syncthreads();
  }
}

// ---- inverses of the 64 x 64 diagonal blocks of R~^T (factor time) --------------------------------------------------------
// One wave per block: lane j back-substitutes column j of U^{-1} (U = R~(b, b), upper triangular) in its registers
// (hssk_backsub64: U in LDS read as broadcasts, no cross-lane step); mode 0 stores it TRANSPOSED (Linv = U^{-T}, lower
// triangular, leading dimension 64) so that the sweep's y_b = Linv y_b reads rows contiguously; modes 1 / 2 serve the
// root's LU (plain inverses of the blocks of U and of the unit lower L).  (The first version kept both matrices in LDS,
// 66 KB per single-wave workgroup -- two waves per CU -- and took 0.26 ms for the leaf level of N = 1e5.)
__global__ __launch_bounds__(64) HSSK_WAVES_PER_SIMD(1) void trtri_diag_kernel(const hssk_trtri_desc* __restrict__ descs,
                                                                              const int* __restrict__ blk_prob,
                                                                              const int* __restrict__ blk_idx) {
  constexpr int LR = HSSK_BACKSUB_LD;
  HSSK_SHARED double s_U[SW_NB * LR];
  HSSK_SHARED double s_rd[SW_NB];
  const hssk_trtri_desc p = descs[blk_prob[blockIdx.x]];
  const int b = blk_idx[blockIdx.x], b0 = b * SW_NB;
  const int nb = min(SW_NB, p.n - b0);
  const int j = threadIdx.x;
  // lane j takes row j of the block, sixteen columns in flight at a time, from addresses clamped into the block (a load per
  // iteration under its own condition waited out a memory round trip each: 64 in a row, 0.10 ms for the launch that inverts the
  // blocks of a whole factorization at N = 1e5)
  {
    const int jr = min(j, nb - 1);
#pragma unroll
    for (int c0 = 0; c0 < SW_NB; c0 += 16) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int c = min(c0 + u, nb - 1);   // strictly upper part; mode 2: U = L^T (unit diagonal)
        v[u] = p.mode == 2 ? hssk_gload(p.R, (b0 + c) + (size_t)(b0 + jr) * p.ldr) : hssk_gload(p.R, (b0 + jr) + (size_t)(b0 + c) * p.ldr);
      }
#pragma unroll
      for (int u = 0; u < 16; u++) s_U[j + (c0 + u) * LR] = (j < c0 + u && c0 + u < nb) ? v[u] : 0.;
    }
  }
  s_rd[j] = j < nb ? (p.mode == 2 ? 1. : 1. / hssk_gload(p.R, (b0 + j) + (size_t)(b0 + j) * p.ldr)) : 0.;
  __syncthreads();
  // column j of U^{-1}: U x = e_j
  double x[SW_NB];
#pragma unroll
  for (int i = 0; i < SW_NB; i++) x[i] = (i == j && j < nb) ? 1. : 0.;
  hssk_backsub64(x, s_U, s_rd, nb);
  double* out = p.Tinv + (size_t)b * SW_NB * SW_NB;
  if (p.mode == 1) {   // out(i, j) = Uinv(i, j)
#pragma unroll
    for (int i = 0; i < SW_NB; i++) out[i + j * SW_NB] = x[i];
  } else {             // out(j, c) = Uinv(c, j): lanes along a row of the output
#pragma unroll
    for (int c = 0; c < SW_NB; c++) out[j + c * SW_NB] = x[c];
  }
}

// groups of right-hand sides along blockIdx.y: side by side, or in turn inside the workgroups (one workgroup per node).
// Measured at N = 1e5, nrhs = 64, inner levels only (profiles/r03_sweeps_nrhs64.md): mat-vec 0.90 ms side by side, 0.44 ms in
// turn; backward solve 0.19 / 0.17 ms; the forward solve -- the longest chain per node -- 0.45 ms side by side, 0.68 ms in turn.
// which: 0 mat-vec, 1 forward, 2 backward.  HSSK_SWEEP_GROUPS_Y = "a,f,b" overrides the three limits.
unsigned groups_y(int nrhs, int which) {
  static int lim[3] = {2, 1 << 20, 2};
  static const bool init = [] {
    if (const char* e = std::getenv("HSSK_SWEEP_GROUPS_Y")) {
      int a = lim[0], f = lim[1], b = lim[2];
      if (std::sscanf(e, "%d,%d,%d", &a, &f, &b) >= 1) { lim[0] = std::max(1, a); lim[1] = std::max(1, f); lim[2] = std::max(1, b); }
    }
    return true;
  }();
  (void)init;
  const int g = (nrhs + SW_NR - 1) / SW_NR;
  return (unsigned)(g <= lim[which] ? g : 1);
}
// which: 0 mat-vec, 1 forward, 2 backward.  Measured at N = 1e5, nrhs = 64, inner levels (profiles/r03_sweeps_nrhs64.md):
// mat-vec 0.44 ms (four right-hand sides per pass, in turn) -> 0.30 ms wide; backward 0.17 -> 0.12 ms; the forward sweep --
// the longest chain of dependent stages per node -- is fastest with its sixteen groups of four SIDE BY SIDE (0.44 ms; 0.56 ms
// wide, 0.68 ms in turn), so it keeps that form.  HSSK_SWEEP_WIDE = "a,f,b" (0 / 1 each) overrides.
bool wide_ok(int nrhs, int dmax, int which) {
  static int on[3] = {1, 0, 1};
  static const bool init = [] {
    if (const char* e = std::getenv("HSSK_SWEEP_WIDE")) std::sscanf(e, "%d,%d,%d", &on[0], &on[1], &on[2]);
    return true;
  }();
  (void)init;
  return on[which] && nrhs >= SW_NRW && dmax <= SW_MAXW;
}
#include "hssk_sweep_mma.h"

int* sweep_err(hssk_ctx* ctx) {
  if (!ctx->h_sweep_err) {
    ctx->h_sweep_err = (int*)hssk_rt::pinned_malloc(64);
    *ctx->h_sweep_err = 0;
  }
  return ctx->h_sweep_err;
}

}  // namespace

static std::atomic<long long> chain_launches{0};
extern "C" long long hssk_sweep_mma_launches(void) { return mma_launches; }
extern "C" long long hssk_sweep_chain_launches(void) { return chain_launches; }
extern "C" int hssk_sweep_mma_min_nrhs(void) { return mma_min_nrhs() > 0 ? mma_min_nrhs() : 1 << 30; }
extern "C" int hssk_sweep_require_mma(hssk_ctx* ctx, int on) { ctx->require_mma = on != 0; return 0; }

extern "C" int hssk_sweep_status(hssk_ctx* ctx) {
  if (!ctx->h_sweep_err) return 0;
  const int e = *(volatile int*)ctx->h_sweep_err;
  if (e) {
    *ctx->h_sweep_err = 0;
    hssk_set_error("hssk sweep: a workgroup timed out waiting for the vectors of its dependency (in-order dispatch violated?)");
  }
  return e;
}

// fills the hand-off buffers of a sweep with the sentinel (must precede, on the stream, every launch that writes them)
extern "C" int hssk_sweep_arm(hssk_ctx* ctx, double* buf, long long count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  const unsigned nb = (unsigned)std::min<long long>((count + 255) / 256, 1024);
  HSSK_LAUNCH(sweep_fill_kernel, dim3(nb), dim3(256), 0, ctx->stream, buf, count);
  hssk_rt::check_launch();
  HSSK_API_END
}

extern "C" int hssk_sweep_chain_ok(int m, int r, int mv, int rv) { return chain_shape_ok(m, r, mv, rv, SW_MAX) ? 1 : 0; }

extern "C" int hssk_ulv_fwd_sweep(hssk_ctx* ctx, const hssk_sweep_fwd_desc* descs, int count, int nrhs) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  if (nrhs < 1 || nrhs > SW_NR * 16384) HSSK_UNSUPPORTED("operands beyond the single-launch sweep");
  for (int i = 0; i < count; i++)
    if (descs[i].m > SW_MAX || descs[i].mv > SW_MAX || descs[i].m < 0 || descs[i].wait0 >= i || descs[i].wait1 >= i) HSSK_UNSUPPORTED("operands beyond the single-launch sweep");
  auto* dd = (const hssk_sweep_fwd_desc*)ctx->stage(descs, sizeof(*descs) * count);
  // (a single right-hand side runs the NR = 1 instantiation: a quarter of the LDS reads and fmas of every pass)
  int dmax = 0;
  for (int i = 0; i < count; i++) dmax = std::max(dmax, std::max(descs[i].m, descs[i].mv));
  if (mma_min_nrhs() > 0 && nrhs >= mma_min_nrhs()) {
    // many right-hand sides: the node arithmetic on the matrix cores, 64 right-hand sides per pass (hssk_sweep_mma.h)
    int rows = 0;
    for (int i = 0; i < count; i++) {
      const FwdRows R = mm_fwd_rows(descs[i].m, descs[i].r, descs[i].mv, descs[i].rv, descs[i].LU != nullptr, descs[i].B01 != nullptr);
      rows = std::max(rows, R.f + R.y + R.a + R.t + R.z);
    }
    if (const int nc = mma_width(nrhs, rows, 2 * SW_T, dmax)) {
      const size_t bytes = mma_lds_bytes(nc, rows, 2 * SW_T);
      mma_launches++;
      // (HSSK_SWEEP_MMA_T_BIG = 512 / 1024: more waves per workgroup for the launches with large nodes -- the leaf level.
      //  Measured at N = 1e5, nrhs = 64: solve 0.54 ms with 256 threads, 0.63 with 512, 0.69 with 1024: the stages of a node are
      //  a dependent chain, and their barriers cost more with more waves than the extra tiles in flight bring)
      static const int tb_big = [] { const char* e = std::getenv("HSSK_SWEEP_MMA_T_BIG"); const int v = e ? std::atoi(e) : SW_T; return (v == 512 || v == 1024) ? v : SW_T; }();
      const int tb = (dmax >= 160 && nc <= 32) ? tb_big : SW_T;
      bool leaves_only = true;
      for (int i = 0; i < count; i++) leaves_only = leaves_only && !descs[i].B01 && descs[i].wait0 < 0 && descs[i].wait1 < 0;
      auto go = [&](auto kernel) {
        hssk_rt::allow_dynamic_lds(kernel, bytes);
        const int ng = mma_groups(nrhs, nc);
        const int xcd = (leaves_only && ng > 1 && count >= 8) ? 1 : 0;
        const unsigned grid = xcd ? (unsigned)((count + 7) / 8) * 8u * (unsigned)ng : (unsigned)count * (unsigned)ng;
        HSSK_LAUNCH(kernel, dim3(grid), dim3((unsigned)tb), bytes, ctx->stream, dd, count, nrhs, ng, xcd, sweep_err(ctx));
      };
      if (tb == 1024 && nc == 32) go(ulv_fwd_sweep_mma_kernel<32, 1024>);
      else if (tb == 1024 && nc == 16) go(ulv_fwd_sweep_mma_kernel<16, 1024>);
      else if (tb == 512 && nc == 32) go(ulv_fwd_sweep_mma_kernel<32, 512>);
      else if (tb == 512 && nc == 16) go(ulv_fwd_sweep_mma_kernel<16, 512>);
      else if (nc == 64) go(ulv_fwd_sweep_mma_kernel<64, SW_T>);
      else if (nc == 32) go(ulv_fwd_sweep_mma_kernel<32, SW_T>);
      else go(ulv_fwd_sweep_mma_kernel<16, SW_T>);
      hssk_rt::check_launch();
      return 0;
    }
  }
  if (nrhs > 64 || ctx->require_mma) HSSK_UNSUPPORTED("operands beyond the matrix-core sweep");
  const unsigned gy = groups_y(nrhs, 1);
  auto lds = [](int nr, int ldv) { return sizeof(double) * ((size_t)5 * ldv * nr + (size_t)SW_T * nr) + sizeof(int) * (size_t)ldv; };
  // (chain blocks: the instantiation that uses them where every inner node below the root brought one the kernel takes)
  bool chain = nrhs == 1;
  int nchain = 0;
  for (int i = 0; i < count && chain; i++) {
    const hssk_sweep_fwd_desc& d = descs[i];
    if (!d.B01 || d.LU || d.m <= d.r) continue;
    chain = d.G != nullptr && chain_shape_ok(d.m, d.r, d.mv, d.rv, SW_MAX);
    nchain++;
  }
  if (nrhs == 1 && chain && nchain > 0) chain_launches++;
  if (nrhs == 1 && chain && nchain > 0) HSSK_LAUNCH((ulv_fwd_sweep_kernel<1, SW_MAX, false, true>), dim3((unsigned)count, 1u), dim3(SW_T), lds(1, SW_MAX), ctx->stream, dd, nrhs, sweep_err(ctx));
  else if (nrhs == 1) HSSK_LAUNCH((ulv_fwd_sweep_kernel<1, SW_MAX, false>), dim3((unsigned)count, 1u), dim3(SW_T), lds(1, SW_MAX), ctx->stream, dd, nrhs, sweep_err(ctx));
  else if ((int)gy * SW_NR >= nrhs && !wide_ok(nrhs, dmax, 1)) HSSK_LAUNCH((ulv_fwd_sweep_kernel<SW_NR, SW_MAX, false>), dim3((unsigned)count, gy), dim3(SW_T), lds(SW_NR, SW_MAX), ctx->stream, dd, nrhs, sweep_err(ctx));
  else if (wide_ok(nrhs, dmax, 1)) {
    // many right-hand sides, small nodes (the inner levels of the hybrid path): sixteen right-hand sides per pass
    hssk_rt::allow_dynamic_lds(ulv_fwd_sweep_kernel<SW_NRW, SW_MAXW, true>, lds(SW_NRW, SW_MAXW));
    HSSK_LAUNCH((ulv_fwd_sweep_kernel<SW_NRW, SW_MAXW, true>), dim3((unsigned)count, 1u), dim3(SW_T), lds(SW_NRW, SW_MAXW), ctx->stream, dd, nrhs, sweep_err(ctx));
  } else HSSK_LAUNCH((ulv_fwd_sweep_kernel<SW_NR, SW_MAX, true>), dim3((unsigned)count, gy), dim3(SW_T), lds(SW_NR, SW_MAX), ctx->stream, dd, nrhs, sweep_err(ctx));
  hssk_rt::check_launch();
  HSSK_API_END
}

extern "C" int hssk_ulv_bwd_sweep(hssk_ctx* ctx, const hssk_sweep_bwd_desc* descs, int count, int nrhs) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  if (nrhs < 1 || nrhs > SW_NR * 16384) HSSK_UNSUPPORTED("operands beyond the single-launch sweep");
  for (int i = 0; i < count; i++)
    if (descs[i].m > SW_MAX || descs[i].wait0 >= i) HSSK_UNSUPPORTED("operands beyond the single-launch sweep");
  auto* dd = (const hssk_sweep_bwd_desc*)ctx->stage(descs, sizeof(*descs) * count);
  int dmax = 0;
  for (int i = 0; i < count; i++) dmax = std::max(dmax, descs[i].m);
  if (mma_min_nrhs() > 0 && nrhs >= mma_min_nrhs()) {
    if (const int nc = mma_width(nrhs, 2 * std::max(dmax, 1), 0, dmax)) {
      const size_t bytes = mma_lds_bytes(nc, 2 * std::max(dmax, 1), 0);
      mma_launches++;
      auto go = [&](auto kernel) {
        hssk_rt::allow_dynamic_lds(kernel, bytes);
        const int ng = mma_groups(nrhs, nc);
        HSSK_LAUNCH(kernel, dim3((unsigned)count * ng), dim3(SW_T), bytes, ctx->stream, dd, nrhs, ng, sweep_err(ctx));
      };
      if (nc == 64) go(ulv_bwd_sweep_mma_kernel<64>);
      else if (nc == 32) go(ulv_bwd_sweep_mma_kernel<32>);
      else go(ulv_bwd_sweep_mma_kernel<16>);
      hssk_rt::check_launch();
      return 0;
    }
  }
  if (nrhs > 64 || ctx->require_mma) HSSK_UNSUPPORTED("operands beyond the matrix-core sweep");
  const unsigned gy = groups_y(nrhs, 2);
  auto lds = [](int nr, int ldv) { return sizeof(double) * ((size_t)2 * ldv * nr + (size_t)SW_T * nr); };
  if (nrhs == 1) HSSK_LAUNCH((ulv_bwd_sweep_kernel<1, SW_MAX, false>), dim3((unsigned)count, 1u), dim3(SW_T), lds(1, SW_MAX), ctx->stream, dd, nrhs, sweep_err(ctx));
  else if ((int)gy * SW_NR >= nrhs && !wide_ok(nrhs, dmax, 2)) HSSK_LAUNCH((ulv_bwd_sweep_kernel<SW_NR, SW_MAX, false>), dim3((unsigned)count, gy), dim3(SW_T), lds(SW_NR, SW_MAX), ctx->stream, dd, nrhs, sweep_err(ctx));
  else if (wide_ok(nrhs, dmax, 2)) {
    hssk_rt::allow_dynamic_lds(ulv_bwd_sweep_kernel<SW_NRW, SW_MAXW, true>, lds(SW_NRW, SW_MAXW));
    HSSK_LAUNCH((ulv_bwd_sweep_kernel<SW_NRW, SW_MAXW, true>), dim3((unsigned)count, 1u), dim3(SW_T), lds(SW_NRW, SW_MAXW), ctx->stream, dd, nrhs, sweep_err(ctx));
  } else HSSK_LAUNCH((ulv_bwd_sweep_kernel<SW_NR, SW_MAX, true>), dim3((unsigned)count, gy), dim3(SW_T), lds(SW_NR, SW_MAX), ctx->stream, dd, nrhs, sweep_err(ctx));
  hssk_rt::check_launch();
  HSSK_API_END
}

extern "C" int hssk_apply_sweep(hssk_ctx* ctx, const hssk_apply_up_desc* ups, int nup, const hssk_apply_down_desc* downs,
                                int ndown, int nrhs) {
  HSSK_API_BEGIN
  if (nup + ndown <= 0) return 0;
  if (nrhs < 1 || nrhs > SW_NR * 16384) HSSK_UNSUPPORTED("operands beyond the single-launch sweep");
  for (int i = 0; i < nup; i++)
    if (ups[i].m > SW_MAX || ups[i].wait0 >= i || ups[i].wait1 >= i) HSSK_UNSUPPORTED("operands beyond the single-launch sweep");
  for (int i = 0; i < ndown; i++) {
    const hssk_apply_down_desc& d = downs[i];
    if (d.mo > SW_MAX || d.m > SW_MAX || d.ri_a + d.ri_b > SW_MAX || d.ro_a + d.ro_b > SW_MAX) HSSK_UNSUPPORTED("operands beyond the single-launch sweep");
    if (d.wait0 >= nup + i || d.wait1 >= nup + i || d.wait2 >= nup + i) HSSK_UNSUPPORTED("operands beyond the single-launch sweep");
  }
  const hssk_apply_up_desc* du = nup ? (const hssk_apply_up_desc*)ctx->stage(ups, sizeof(*ups) * nup) : nullptr;
  const hssk_apply_down_desc* dn = ndown ? (const hssk_apply_down_desc*)ctx->stage(downs, sizeof(*downs) * ndown) : nullptr;
  int dmax = 0;
  for (int i = 0; i < nup; i++) dmax = std::max(dmax, ups[i].m);
  for (int i = 0; i < ndown; i++) dmax = std::max(dmax, std::max(std::max(downs[i].mo, downs[i].m), std::max(downs[i].ri_a + downs[i].ri_b, downs[i].ro_a + downs[i].ro_b)));
  const unsigned nwg = (unsigned)(nup + ndown);
  if (mma_min_nrhs() > 0 && nrhs >= mma_min_nrhs()) {
    int rows = 0;
    bool ok = true;
    for (int i = 0; i < nup; i++) rows = std::max(rows, std::max(ups[i].m, 1));
    for (int i = 0; i < ndown; i++) {
      const hssk_apply_down_desc& d = downs[i];
      if (d.D) { ok = false; break; }   // (leaves: the batched launches of the many-right-hand-side path, or the vector form)
      rows = std::max(rows, mm_apply_down_rows(d.ri_a + d.ri_b, d.ro, d.ro_a + d.ro_b, d.mo, d.acc));
    }
    const int nc = ok ? mma_width(nrhs, rows, SW_T, 0) : 0;
    if (nc) {
      const size_t bytes = mma_lds_bytes(nc, rows, SW_T);
      mma_launches++;
      auto go = [&](auto kernel) {
        hssk_rt::allow_dynamic_lds(kernel, bytes);
        const int ng = mma_groups(nrhs, nc);
        HSSK_LAUNCH(kernel, dim3(nwg * ng), dim3(SW_T), bytes, ctx->stream, du, nup, dn, nrhs, ng, sweep_err(ctx));
      };
      if (nc == 64) go(apply_sweep_mma_kernel<64>);
      else if (nc == 32) go(apply_sweep_mma_kernel<32>);
      else go(apply_sweep_mma_kernel<16>);
      hssk_rt::check_launch();
      return 0;
    }
  }
  if (nrhs > 64 || ctx->require_mma) HSSK_UNSUPPORTED("operands beyond the matrix-core sweep");
  for (int i = 0; i < ndown; i++)
    if (downs[i].acc) HSSK_UNSUPPORTED("accumulating leaves are served by the matrix-core sweep only");
  const unsigned gy = groups_y(nrhs, 0);
  auto lds = [](int nr, int ldv) { return sizeof(double) * ((size_t)3 * ldv * nr + (size_t)SW_T * nr); };
  if (nrhs == 1) HSSK_LAUNCH((apply_sweep_kernel<1, SW_MAX, false>), dim3(nwg, 1u), dim3(SW_T), lds(1, SW_MAX), ctx->stream, du, nup, dn, nrhs, sweep_err(ctx));
  else if ((int)gy * SW_NR >= nrhs && !wide_ok(nrhs, dmax, 0)) HSSK_LAUNCH((apply_sweep_kernel<SW_NR, SW_MAX, false>), dim3(nwg, gy), dim3(SW_T), lds(SW_NR, SW_MAX), ctx->stream, du, nup, dn, nrhs, sweep_err(ctx));
  else if (wide_ok(nrhs, dmax, 0)) {
    hssk_rt::allow_dynamic_lds(apply_sweep_kernel<SW_NRW, SW_MAXW, true>, lds(SW_NRW, SW_MAXW));
    HSSK_LAUNCH((apply_sweep_kernel<SW_NRW, SW_MAXW, true>), dim3(nwg, 1u), dim3(SW_T), lds(SW_NRW, SW_MAXW), ctx->stream, du, nup, dn, nrhs, sweep_err(ctx));
  } else HSSK_LAUNCH((apply_sweep_kernel<SW_NR, SW_MAX, true>), dim3(nwg, gy), dim3(SW_T), lds(SW_NR, SW_MAX), ctx->stream, du, nup, dn, nrhs, sweep_err(ctx));
  hssk_rt::check_launch();
  HSSK_API_END
}

extern "C" int hssk_trtri_diag_vbatched(hssk_ctx* ctx, const hssk_trtri_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  std::vector<int> prob, idx;
  for (int i = 0; i < count; i++)
    for (int b = 0; b * SW_NB < descs[i].n; b++) { prob.push_back(i); idx.push_back(b); }
  if (prob.empty()) return 0;
  auto* dd = (const hssk_trtri_desc*)ctx->stage(descs, sizeof(*descs) * count);
  auto* dp = (const int*)ctx->stage(prob.data(), sizeof(int) * prob.size());
  auto* di = (const int*)ctx->stage(idx.data(), sizeof(int) * idx.size());
  HSSK_LAUNCH(trtri_diag_kernel, dim3((unsigned)prob.size()), dim3(64), 0, ctx->stream, dd, dp, di);
  hssk_rt::check_launch();
  HSSK_API_END
}
