// hssk_tree_inner: ALL inner levels of one round of the randomized HSS compression as ONE launch
// (compress_recursive_stable above the leaves: HSS/HSSMatrix.compress_stable.hpp:165-278 -- block extraction :204-217,
//  compute_local_samples HSS/HSSMatrix.compress.hpp:555-629, the interpolative decompositions compress_stable.hpp:280-348,
//  reduce_local_samples compress.hpp:689-724, the skeleton index composition compress_stable.hpp:299-306 / 334-341).
//
// The level-synchronous form of these steps needs the ranks of level l on the host before it can size, allocate and launch
// level l + 1: one device -> host read-back and six dependent launches per level, each a latency chain of its own (at
// N = 1e5, leaf 256: 9 levels x ~180 us, of which the kernels' own chains are about half).  Here every (node, basis) pair is
// ONE workgroup of a single launch; the workgroups are ordered children before parents and a parent polls its children's
// completion flags (the in-order dispatch argument of the single-launch tree sweeps, hssk_sweep.hip: whatever a workgroup
// waits for was dispatched before it and is running or done).  Ranks never leave the device until the tree is finished:
// a node's storage is sized for a speculated rank bound (rcap, chosen by the host from the leaves' ranks), every array the
// node writes that depends on a rank is compact INSIDE that storage, and a rank beyond the bound raises a status the host
// answers by taking the level-synchronous path for the inner levels (nothing the leaves hold has been touched).
//
// One workgroup = 8 waves, side s of node nu (s = 0: row basis U from the row samples; s = 1: column basis V):
//   1. wait for both sides of both children; ranks ra[], rb[]; m = ra[s] + rb[s] rows enter the decomposition
//   2. coupling blocks B01 = A(Ir_a, Ic_b), B10 = A(Ir_b, Ic_a) gathered into the LDS (side 0 keeps B01, side 1 B10)
//   3. local samples, transposed (d x m):  S(:, j) = S_child(:, perm_child[j]) - sum_k Rred_sibling(:, k) C(j, k)
//   4. truncated column-pivoted QR of S in registers (id_reg_body, hssk_id_reg.h): rank r, pivots
//   5. X = R11^{-1} R12 (r x (m - r)), kept in the LDS for
//   6. the reduced random samples of the OTHER kind (U reduces Rc, V reduces Rr):
//      Rred(:, j) = R_children(:, perm[j]) + sum_k R_children(:, perm[r + k]) X(j, k)
//   7. global skeleton indices; publish r; raise the flag
// Everything another workgroup reads (S, Rred, perm, skeleton indices, ranks) is written with device-coherent stores and
// read with device-coherent loads (the L2 caches of the XCDs are not coherent for plain accesses inside a launch).
#include "hssk_device.h"
#include "hssk_backsub.h"
#include "hssk_gen.h"
#include "hssk_internal.h"
#include "hssk_id_reg.h"

#include <algorithm>
#include <vector>

namespace {

constexpr int TR_T = 512;              // threads per workgroup (8 waves: the register kernel's NW = 8)
constexpr long TR_SPIN_LIMIT = 1L << 23;   // polls of a child's flag before giving up (seconds; a tree finishes within milliseconds)

struct TreeParams {
  int d, lds, rcap, max_rank;
  double rtol, atol;
  const double* A;
  long long lda;
  hssk_gen gen;
  int use_gen;
};

__device__ __forceinline__ double tr_elem(const TreeParams& P, int i, int j) {
  return P.use_gen ? hssk_gen_eval(P.gen, i, j) : hssk_gload(P.A, (size_t)i + (size_t)j * (size_t)P.lda);
}

// accumulate acc[j] += mk * C(j, k) over j < J for one k; C(j, k) = c[j * sj + k * sk] in the LDS (broadcast reads)
template <int JB>
__device__ __forceinline__ void tr_fma_row(double (&acc)[JB], double mk, const double* c, int sj, int J) {
#pragma unroll
  for (int jb = 0; jb < JB / 8; jb++) {
    if (8 * jb < J) {
#pragma unroll
      for (int j = 8 * jb; j < 8 * jb + 8; j++) acc[j] += mk * c[j * sj];
    }
  }
}

template <int RT, int CT>
__global__ __launch_bounds__(TR_T) HSSK_WAVES_PER_SIMD(2) void tree_inner_kernel(hssk_tnode* __restrict__ nodes, const int* __restrict__ order,
                                                                                 TreeParams P, int* __restrict__ res, int* __restrict__ err) {
  constexpr int RC = 16 * CT;        // rank bound of this instantiation (= P.rcap), m <= 2 RC
  constexpr int KP = 8;              // sibling / children sample columns loaded together
  HSSK_DYN_SHARED(double, s_dyn);    // 2 RC^2 doubles (B01 | B10) during steps 2-3; then 64 x 66 (R11) + RC x 2 RC (X)
  HSSK_SHARED int s_I[4][RC];        // Ir_a, Ic_a, Ir_b, Ic_b
  HSSK_SHARED int s_pc[2][RC];       // the children's pivoted orders of this side (their skeleton columns first)
  HSSK_SHARED int s_perm[2 * RC];    // this node's pivoted order
  HSSK_SHARED int s_ok;
  const int tid = threadIdx.x;
  const int ord = order[blockIdx.x];
  const int id = ord >> 1, s = ord & 1;
  hssk_tnode* nd = nodes + id;
  const hssk_tnode* a = nodes + nd->c0;
  const hssk_tnode* b = nodes + nd->c1;
  const bool root = nd->lvl == 0;
  const int d = P.d, lds = P.lds;

  // ---- 1. the children (both sides of both)
  if (tid == 0) {
    int ok = 1;
    const int* fl[4] = {&a->flag[0], &a->flag[1], &b->flag[0], &b->flag[1]};
    for (int q = 0; q < 4 && ok; q++) {
      long spins = 0;
      while (hssk_flag_load(fl[q]) == 0) {
        hssk_pause();
        if (++spins > TR_SPIN_LIMIT) { hssk_flag_raise(err); ok = 0; break; }
      }
    }
    s_ok = ok;
  }
  __syncthreads();
  int ra[2], rb[2];
  ra[0] = hssk_flag_load(&a->r[0]); ra[1] = hssk_flag_load(&a->r[1]);
  rb[0] = hssk_flag_load(&b->r[0]); rb[1] = hssk_flag_load(&b->r[1]);
  const int m = ra[s] + rb[s];
  int status = (s_ok && hssk_flag_load(&a->status) == 0 && hssk_flag_load(&b->status) == 0) ? 0 : 1;
  if (ra[0] > RC || ra[1] > RC || rb[0] > RC || rb[1] > RC || (!root && m > d)) status = 1;
  auto publish = [&](int r, int st) {
    // (every store of this workgroup has left the CU before the flag goes up)
    hssk_drain_stores();
    __syncthreads();
    if (tid == 0) {
      hssk_flag_store(&nd->r[s], r);
      hssk_flag_store(&nd->m[s], m);
      if (st) hssk_flag_store(&nd->status, st);
      res[4 * id + s] = r;
      res[4 * id + 2 + s] = st;   // (every (node, side) writes its own word: the caller need not clear the array)
      hssk_drain_stores();
      hssk_flag_store(&nd->flag[s], 1);
      if (root) hssk_flag_store(&nd->flag[1], 1);
    }
  };
  if (status) { publish(0, 1); return; }

  // ---- 2. skeleton index lists, the children's pivoted orders, the coupling blocks
  for (int e = tid; e < 4 * RC; e += TR_T) {
    const int q = e / RC, i = e % RC;
    const hssk_tnode* c = q < 2 ? a : b;
    const int w = q & 1, rc = q < 2 ? ra[w] : rb[w];
    s_I[q][i] = i < rc ? hssk_flag_load(c->I[w] + i) : 0;
  }
  for (int e = tid; e < 2 * RC; e += TR_T) {
    const int q = e / RC, i = e % RC;
    const hssk_tnode* c = q == 0 ? a : b;
    const int rc = q == 0 ? ra[s] : rb[s];
    s_pc[q][i] = i < rc ? hssk_flag_load(c->perm[s] + i) : 0;
  }
  __syncthreads();
  double* sB01 = s_dyn;
  double* sB10 = s_dyn + RC * RC;
  {
    const int n01 = ra[0] * rb[1], n10 = rb[0] * ra[1];
    const int l01 = ra[0] > 0 ? ra[0] : 1, l10 = rb[0] > 0 ? rb[0] : 1;
    for (int e = tid; e < n01; e += TR_T) {
      const int i = e % ra[0], j = e / ra[0];
      const double v = tr_elem(P, s_I[0][i], s_I[3][j]);   // A(Ir_a[i], Ic_b[j])
      sB01[i + j * RC] = v;
      if (s == 0) hssk_gstore(nd->B01, (size_t)i + (size_t)j * l01, v);
    }
    for (int e = tid; e < n10; e += TR_T) {
      const int i = e % rb[0], j = e / rb[0];
      const double v = tr_elem(P, s_I[2][i], s_I[1][j]);   // A(Ir_b[i], Ic_a[j])
      sB10[i + j * RC] = v;
      if (s == 1 || root) hssk_gstore(nd->B10, (size_t)i + (size_t)j * l10, v);
    }
  }
  if (root) { publish(0, 0); return; }   // (the root has no basis: its coupling blocks are all there is)
  __syncthreads();

  // ---- 3. local samples (transposed).  Thread = sample row i; half 0 of the workgroup takes the columns that come from child a
  // (coupled with b's reduced samples), half 1 those from child b.
  double* S = nd->S[s];
  {
    const int i = tid & 255, hb = tid >> 8;
    const bool live = i < d;
    const hssk_tnode* cg = hb == 0 ? a : b;       // the child whose skeleton columns are gathered
    const hssk_tnode* cm = hb == 0 ? b : a;       // the sibling whose reduced samples are multiplied
    const int J = hb == 0 ? ra[s] : rb[s];
    const int K = hb == 0 ? rb[1 - s] : ra[1 - s];
    const double* G = cg->S[s];
    const double* M = cm->Rred[s];
    // C(j, k): s = 0: hb 0 -> B01(j, k), hb 1 -> B10(j, k);  s = 1: hb 0 -> B10(k, j), hb 1 -> B01(k, j)
    const double* C = (s == 0) == (hb == 0) ? sB01 : sB10;
    const int sj = s == 0 ? 1 : RC, sk = s == 0 ? RC : 1;
    double acc[RC];
#pragma unroll
    for (int j = 0; j < RC; j++) acc[j] = 0.;
    for (int k0 = 0; k0 < K; k0 += KP) {
      double mk[KP];
#pragma unroll
      for (int u = 0; u < KP; u++) mk[u] = (live && k0 + u < K) ? hssk_cload(M, (size_t)i + (size_t)(k0 + u) * lds) : 0.;
#pragma unroll
      for (int u = 0; u < KP; u++)
        if (k0 + u < K) tr_fma_row<RC>(acc, mk[u], C + (k0 + u) * sk, sj, J);
    }
    const int off = hb == 0 ? 0 : ra[s];
    // (the gathered columns eight at a time: eight coherent loads in flight, then their stores -- one load per store would be a
    // memory round trip per column)
#pragma unroll
    for (int jb = 0; jb < RC / 8; jb++) {
      if (8 * jb < J) {
        double g[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int j = 8 * jb + u;
          g[u] = (live && j < J) ? hssk_cload(G, (size_t)i + (size_t)s_pc[hb][j] * lds) : 0.;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int j = 8 * jb + u;
          if (live && j < J) hssk_cstore(S, (size_t)i + (size_t)(off + j) * lds, g[u] - acc[j]);
        }
      }
    }
  }
  hssk_drain_stores();
  __syncthreads();

  // ---- 4. the interpolative decomposition of the panel, in registers
  int rank = 0;
  const int ldw = 2 * RC;
  double* W = nd->W[s];
  int* perm = nd->perm[s];
  if (m > 0) {
    hssk_id_desc p;
    p.W = W; p.ldw = ldw; p.d = d; p.m = m;
    p.rtol = P.rtol / nd->lvl; p.atol = P.atol / nd->lvl;
    p.max_rank = P.max_rank;
    p.perm = perm; p.rank = nullptr; p.work = nullptr;
    p.src = S; p.lds = lds; p.defer_x = 1;
    // (a Householder step costs per column slot of the register tile: panels of up to 64 / 96 columns take the narrower tiles)
    if (CT >= 3 && m <= 64) rank = id_reg_body<RT, 2, 8>(p);
    else if (CT >= 4 && m <= 96) rank = id_reg_body<RT, 3, 8>(p);
    else rank = id_reg_body<RT, CT, 8>(p);
  }
  if (rank > RC) { publish(0, 1); return; }
  // the pivoted order: into the LDS for the steps below, and out again with coherent stores for the parent
  for (int j = tid; j < m; j += TR_T) { const int v = perm[j]; s_perm[j] = v; hssk_flag_store(perm + j, v); }
  __syncthreads();

  // ---- 5. X = R11^{-1} R12: one thread per column of R12, R11 staged in the LDS (rank <= RC <= 64)
  double* sR = s_dyn;                               // 64 x HSSK_BACKSUB_LD
  double* s_rd = s_dyn + 64 * HSSK_BACKSUB_LD;      // 64 reciprocals of the diagonal
  double* sX = s_rd + 64;                           // X(j, k) at sX[j + k * RC]
  const int K2 = m - rank;
  double* X = nd->X[s];
  if (rank > 0 && K2 > 0) {
    for (int e = tid; e < 64 * 64; e += TR_T) {
      const int i = e & 63, l = e >> 6;
      sR[i + l * HSSK_BACKSUB_LD] = (i < l && l < rank) ? W[i + (size_t)l * ldw] : 0.;
    }
    if (tid < 64) s_rd[tid] = tid < rank ? 1. / W[tid + (size_t)tid * ldw] : 0.;
    __syncthreads();
    if (tid < K2) {
      const double* bcol = W + (size_t)(rank + tid) * ldw;
      double x[64];
#pragma unroll
      for (int bq = 0; bq < 8; bq++) {
        if (8 * bq < rank) {
#pragma unroll
          for (int i = 8 * bq; i < 8 * bq + 8; i++) x[i] = i < rank ? bcol[i] : 0.;
        } else {
#pragma unroll
          for (int i = 8 * bq; i < 8 * bq + 8; i++) x[i] = 0.;
        }
      }
      hssk_backsub64(x, sR, s_rd, rank);
#pragma unroll
      for (int bq = 0; bq < 8; bq++) {
        if (8 * bq < rank) {
#pragma unroll
          for (int i = 8 * bq; i < 8 * bq + 8; i++)
            if (i < rank) { sX[i + tid * RC] = x[i]; hssk_gstore(X, (size_t)i + (size_t)tid * rank, x[i]); }
        }
      }
    }
  }
  __syncthreads();

  // ---- 6. reduced random samples of the other kind: [a.Rred | b.Rred](:, perm) combined through X
  if (rank > 0) {
    const int o = 1 - s;
    const int i = tid & 255, hb = tid >> 8;
    const bool live = i < d;
    const double* R0 = a->Rred[o];
    const double* R1 = b->Rred[o];
    const int split = ra[s];
    double* out = nd->Rred[o];
    // half hb of the workgroup takes the output columns [j0, j0 + J)
    const int JH = (rank + 1) / 2, j0 = hb * JH, J = hb == 0 ? JH : rank - JH;
    constexpr int JB = (RC / 2 + 7) / 8 * 8;   // >= ceil(rank / 2), a multiple of 8
    double acc[JB];
#pragma unroll
    for (int j = 0; j < JB; j++) acc[j] = 0.;
    auto col = [&](int c) { return c < split ? R0 + (size_t)c * lds : R1 + (size_t)(c - split) * lds; };
    for (int k0 = 0; k0 < K2; k0 += KP) {
      double mk[KP];
#pragma unroll
      for (int u = 0; u < KP; u++) mk[u] = (live && k0 + u < K2) ? hssk_cload(col(s_perm[rank + k0 + u]), (size_t)i) : 0.;
#pragma unroll
      for (int u = 0; u < KP; u++)
        if (k0 + u < K2) tr_fma_row<JB>(acc, mk[u], sX + j0 + (k0 + u) * RC, 1, J);
    }
#pragma unroll
    for (int jb = 0; jb < JB / 8; jb++) {
      if (8 * jb < J) {
        double g[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int j = 8 * jb + u;
          g[u] = (live && j < J) ? hssk_cload(col(s_perm[j0 + j]), (size_t)i) : 0.;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int j = 8 * jb + u;
          if (live && j < J) hssk_cstore(out, (size_t)i + (size_t)(j0 + j) * lds, g[u] + acc[j]);
        }
      }
    }
  }

  // ---- 7. global skeleton indices of this basis (children's indices composed with the pivots)
  for (int j = tid; j < rank; j += TR_T) {
    const int pj = s_perm[j];
    const int v = pj < ra[s] ? s_I[s][pj] : s_I[2 + s][pj - ra[s]];
    hssk_flag_store(nd->I[s] + j, v);
  }
  publish(rank, 0);
}

template <int RT, int CT>
void launch_tree(hssk_ctx* ctx, hssk_tnode* nodes, const int* dorder, int count, const TreeParams& P, int* res, int* err) {
  constexpr int RC = 16 * CT;
  const size_t dyn = sizeof(double) * std::max<size_t>(2 * RC * RC, 64 * HSSK_BACKSUB_LD + 64 + RC * 2 * RC);
  static const bool once = [&] { hssk_rt::allow_dynamic_lds(tree_inner_kernel<RT, CT>, dyn); return true; }();
  (void)once;
  HSSK_LAUNCH((tree_inner_kernel<RT, CT>), dim3((unsigned)count), dim3(TR_T), dyn, ctx->stream, nodes, dorder, P, res, err);
}

}  // namespace

// dynamic LDS of the launch for a rank bound (launch_tree) plus the kernel's static arrays (three register-ID instantiations,
// tables: ~20 KB): what a workgroup must be able to get
static size_t tree_lds_bytes(int rcap) {
  const size_t RC = (size_t)rcap;
  return sizeof(double) * std::max<size_t>(2 * RC * RC, 64 * HSSK_BACKSUB_LD + 64 + RC * 2 * RC) + (size_t)20 * 1024;
}
// largest rank bound whose launch fits the device's LDS (64 on gfx950: 100 KB of 160; a 64 KB part gets 32; 0: none fits)
extern "C" int hssk_tree_rcap_max(void) {
  const size_t cap = hssk_rt::max_lds_per_workgroup();
  for (int rcap : {64, 48, 32})
    if (tree_lds_bytes(rcap) <= cap) return rcap;
  return 0;
}

extern "C" int hssk_tree_inner(hssk_ctx* ctx, hssk_tnode* nodes, const int* order, int count, int d, int lds, int rcap, double rtol,
                               double atol, int max_rank, const hssk_elem_src* src, int* res) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  if (!src || (!src->use_gen && !src->A)) HSSK_UNSUPPORTED("no element source");
  if (d <= 0 || d > 256 || (rcap != 32 && rcap != 48 && rcap != 64)) HSSK_UNSUPPORTED("sample count / rank bound outside the kernel's variants");
  if (src->use_gen && src->gen.kind != HSSK_GEN_TOEPLITZ && src->gen.kind != HSSK_GEN_TOEPLITZ_UPPER) HSSK_UNSUPPORTED("unknown generator kind");
  if (tree_lds_bytes(rcap) > hssk_rt::max_lds_per_workgroup()) HSSK_UNSUPPORTED("rank bound beyond this device's LDS");   // (the caller takes the level path)
  if (!ctx->h_sweep_err) { ctx->h_sweep_err = (int*)hssk_rt::pinned_malloc(64); *ctx->h_sweep_err = 0; }
  TreeParams P;
  P.d = d; P.lds = lds; P.rcap = rcap; P.max_rank = max_rank; P.rtol = rtol; P.atol = atol;
  P.A = src->A; P.lda = src->lda; P.gen = src->gen; P.use_gen = src->use_gen;
  auto* dorder = (const int*)ctx->stage(order, sizeof(int) * (size_t)count);
  const int ct = rcap / 16;
  if (d <= 192) {
    if (ct == 2) launch_tree<12, 2>(ctx, nodes, dorder, count, P, res, ctx->h_sweep_err);
    else if (ct == 3) launch_tree<12, 3>(ctx, nodes, dorder, count, P, res, ctx->h_sweep_err);
    else launch_tree<12, 4>(ctx, nodes, dorder, count, P, res, ctx->h_sweep_err);
  } else {
    if (ct == 2) launch_tree<16, 2>(ctx, nodes, dorder, count, P, res, ctx->h_sweep_err);
    else if (ct == 3) launch_tree<16, 3>(ctx, nodes, dorder, count, P, res, ctx->h_sweep_err);
    else launch_tree<16, 4>(ctx, nodes, dorder, count, P, res, ctx->h_sweep_err);
  }
  hssk_rt::check_launch();
  HSSK_API_END
}
