// hssk_ulv_node: the ULV elimination of the INNER nodes of one tree level in ONE launch (HSSMatrix::factor_recursive,
// HSS/HSSMatrix.factor.hpp:57-147: the assembly of the reduced block :65-97, W1 / W0 :109-118, the LQ :122, Vt0 / Vt1 / Dt
// :123-137).
//
// Level by level these steps were five dependent launches per level -- two batched GEMMs (coupling blocks times the children's
// Vt1 into D-hat, the children's Vt1 times the dense column basis into V-hat), hssk_ulv_split, the register QR, the Q
// formation, four more small GEMMs -- each 10 - 17 us for a few 80 x 80 blocks: 0.69 ms for the nine inner levels of
// N = 1e5.  Here a node is ONE workgroup that walks the steps with barriers in between; the node's blocks stay in the L1 / L2
// of its CU and the explicit Q in the LDS for the products that read it.  The factorization itself is the register QR and
// Q formation of hssk_qr.hip, instantiated as device bodies (hssk_qr_reg.h): same arithmetic, same factors.
#include "hssk_device.h"
#include "hssk_internal.h"
#include "hssk_qr_reg.h"

#include <algorithm>
#include <atomic>
#include <vector>

namespace {

constexpr int UN_NW = 16, UN_T = UN_NW * 64;

// `count` doubles into the LDS, eight loads in flight per thread: element e comes from src(e) (an index into global memory,
// clamped by the caller's function for e >= count) and lands at dst(e).  A load per loop iteration waits out a memory round
// trip each -- with one workgroup per node and a handful of nodes on the chip nothing else hides it.
template <class FS, class FD>
__device__ __forceinline__ void un_stage(const double* g, int count, FS src, FD dst, double* lds) {
  for (int base = 0; base < count; base += 8 * UN_T) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int e = min(base + u * UN_T + (int)threadIdx.x, count - 1);
      v[u] = hssk_gload(g, src(e));
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int e = base + u * UN_T + (int)threadIdx.x;
      if (e < count) lds[dst(e)] = v[u];
    }
  }
}
// a column-major block (rows x cols, leading dimension ld) into the LDS, packed (leading dimension rows)
__device__ __forceinline__ void un_stage_block(const double* g, int rows, int cols, int ld, double* lds) {
  if (rows <= 0 || cols <= 0) return;
  un_stage(g, rows * cols, [=](int e) { return (size_t)(e % rows) + (size_t)(e / rows) * ld; }, [=](int e) { return e; }, lds);
}

// LDS doubles a node needs: the square region (D-hat in permuted row order, later Q~) followed by the region of the stacked left
// factors [Vh^T; W1]; the staged inputs of the assembly and X share the two
__host__ __device__ inline size_t un_lds_doubles(int m, int r, int rv, int ra, int rb, int rva, int rvb) {
  const size_t sq = (size_t)m * (m + 1), lf = (size_t)(r + rv) * m + (size_t)r * (m - r);
  const size_t in = (size_t)ra * rvb + (size_t)rb * rvb + (size_t)rb * rva + (size_t)ra * rva + (size_t)(rva + rvb) * rv;
  return sq + lf > in ? sq + lf : in;
}

template <int RT, int CT>
__global__ __launch_bounds__(UN_T) HSSK_WAVES_PER_SIMD(UN_NW / 4) void ulv_node_kernel(const hssk_ulvnode_desc* __restrict__ descs) {
  HSSK_DYN_SHARED(double, S);
  const hssk_ulvnode_desc p = descs[blockIdx.x];
  const int tid = threadIdx.x;
  const int m = p.m, r = p.r, q = m - r, rv = p.rv;
  const int ra = p.ra, rb = p.rb, rva = p.rva, rvb = p.rvb;
  const int lS = m + 1;
  // ---- the assembly's operands into the LDS: B01 | B10 | Vt1a | Vt1b | Vd
  double* sB01 = S;
  double* sB10 = sB01 + ra * rvb;
  double* sVa = sB10 + rb * rva;
  double* sVb = sVa + ra * rva;
  double* sVd = sVb + rb * rvb;
  const bool vh = p.Vd != nullptr && rv > 0;
  un_stage_block(p.B01, ra, rvb, max(ra, 1), sB01);
  un_stage_block(p.B10, rb, rva, max(rb, 1), sB10);
  un_stage_block(p.Vt1a, ra, rva, max(ra, 1), sVa);
  un_stage_block(p.Vt1b, rb, rvb, max(rb, 1), sVb);
  if (vh) un_stage_block(p.Vd, rva + rvb, rv, rva + rvb, sVd);
  __syncthreads();
  // D(0:ra, ra:) = B01 Vt1_b^T, D(ra:, 0:ra) = B10 Vt1_a^T; V-hat = [Vt1_a Vd(0:rva, :); Vt1_b Vd(rva:, :)]
  for (int e = tid; e < 2 * ra * rb; e += UN_T) {
    const bool up = e < ra * rb;
    const int f = up ? e : e - ra * rb;
    const int M = up ? ra : rb, K = up ? rvb : rva;
    const int i = f % M, j = f / M;
    const double* a = (up ? sB01 : sB10) + i;
    const double* b = (up ? sVb : sVa) + j;
    const int lb = up ? rb : ra;
    double s0 = 0., s1 = 0.;
    int k = 0;
    for (; k + 1 < K; k += 2) { s0 += a[k * M] * b[k * lb]; s1 += a[(k + 1) * M] * b[(k + 1) * lb]; }
    if (k < K) s0 += a[k * M] * b[k * lb];
    hssk_gstore(p.Dh, up ? (size_t)i + (size_t)(ra + j) * m : (size_t)(ra + i) + (size_t)j * m, s0 + s1);
  }
  if (vh) {
    const int mv = rva + rvb;
    for (int e = tid; e < m * rv; e += UN_T) {
      const int i = e % m, j = e / m;
      const bool top = i < ra;
      const double* a = top ? sVa + i : sVb + (i - ra);
      const int la = top ? ra : rb, K = top ? rva : rvb;
      const double* d = sVd + (top ? 0 : rva) + j * mv;
      double s0 = 0., s1 = 0.;
      int k = 0;
      for (; k + 1 < K; k += 2) { s0 += a[k * la] * d[k]; s1 += a[(k + 1) * la] * d[k + 1]; }
      if (k < K) s0 += a[k * la] * d[k];
      hssk_gstore(p.Vh, i + (size_t)j * m, s0 + s1);
    }
  }
  if (!p.eliminate) return;   // (the root: its LU is the caller's next launch)
  __syncthreads();
  // ---- the split (hssk_ulv_split): W1 = (P^T D)(0:r, :), W0^T = (P^T D)(r:, :)^T - W1^T X;  D-hat into the LDS in permuted row
  // order, X behind the left factors' region
  double* sL = S + m * lS;                 // [Vh^T; W1]: (rv + r) x m, leading dimension rv + r
  double* sX = sL + (r + rv) * m;          // r x q
  un_stage(p.Dh, m * m, [=](int e) { return (size_t)p.perm[e % m] + (size_t)(e / m) * m; }, [=](int e) { return (e % m) + (e / m) * lS; }, S);
  un_stage_block(p.X, r, q, max(r, 1), sX);
  __syncthreads();
  const int lL = r + rv;
  for (int e = tid; e < r * m; e += UN_T) {
    const int k = e % r, c = e / r;
    const double v = S[k + c * lS];
    hssk_gstore(p.W1, k + (size_t)c * max(r, 1), v);
    sL[rv + k + c * lL] = v;
  }
  for (int e = tid; e < m * q; e += UN_T) {
    const int c = e % m, j = e / m;
    const double* x = sX + j * r;
    const double* dc = S + c * lS;
    double s0 = dc[r + j], s1 = 0., s2 = 0., s3 = 0.;
    int k = 0;
    for (; k + 3 < r; k += 4) {
      s0 -= dc[k] * x[k];
      s1 -= dc[k + 1] * x[k + 1];
      s2 -= dc[k + 2] * x[k + 2];
      s3 -= dc[k + 3] * x[k + 3];
    }
    for (; k < r; k++) s0 -= dc[k] * x[k];
    hssk_gstore(p.Rlq, c + (size_t)j * m, (s0 + s1) + (s2 + s3));
  }
  __syncthreads();
  // ---- LQ(W0) == QR(W0^T): the factored panel and the explicit Q~ (m x m)
  hssk_qr_desc qd;
  qd.A = p.Rlq; qd.rows = m; qd.lda = m; qd.cols = q; qd.Q = p.Qt; qd.ldq = m; qd.nq = m; qd.rdiag = nullptr; qd.work = p.tau;
  qd.stair = 0; qd.r_only = 0; qd.stop_rel = 0.; qd.stop_abs = 0.;
  qr_reg_body<RT, CT, UN_NW>(qd);
  __syncthreads();
  formq_reg_body<RT, CT, UN_NW>(qd, 0);
  __syncthreads();
  // ---- Q~ and Vh^T into the LDS (W1 is there);  [Vt0^T, Vt1^T; WQ, Dt] = [Vh^T; W1] Q~
  un_stage(p.Qt, m * m, [=](int e) { return (size_t)e; }, [=](int e) { return (e % m) + (e / m) * lS; }, S);
  if (rv > 0) un_stage(p.Vh, m * rv, [=](int e) { return (size_t)e; }, [=](int e) { return (e / m) + (e % m) * lL; }, sL);
  __syncthreads();
  for (int e = tid; e < lL * m; e += UN_T) {
    const int i = e % lL, j = e / lL;
    const double* a = sL + i;
    const double* b = S + j * lS;
    double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
    int k = 0;
    for (; k + 3 < m; k += 4) {
      s0 += a[k * lL] * b[k];
      s1 += a[(k + 1) * lL] * b[k + 1];
      s2 += a[(k + 2) * lL] * b[k + 2];
      s3 += a[(k + 3) * lL] * b[k + 3];
    }
    for (; k < m; k++) s0 += a[k * lL] * b[k];
    const double v = (s0 + s1) + (s2 + s3);
    if (i < rv) {
      if (j < q) hssk_gstore(p.Vt0T, i + (size_t)j * rv, v);                      // Vt0^T = Vh^T Q~(:, 0:q)
      else hssk_gstore(p.Vt1, (j - q) + (size_t)i * max(r, 1), v);               // Vt1 = Q~(:, q:)^T Vh
    } else {
      if (j < q) hssk_gstore(p.WQ, (i - rv) + (size_t)j * max(r, 1), v);         // WQ = W1 Q~(:, 0:q)
      else hssk_gstore(p.Dt, (i - rv) + (size_t)(j - q) * p.ldt, v);             // Dt = W1 Q~(:, q:)
    }
  }
}

constexpr int UN_MMAX = 128;

}  // namespace

static std::atomic<long long> un_launches{0};
extern "C" long long hssk_ulv_node_launches(void) { return un_launches; }
// dynamic LDS a launch may take next to the static arrays of the QR / Q bodies (37 KB in the 128-row instantiation)
static size_t un_lds_cap() {
  const size_t cap = hssk_rt::max_lds_per_workgroup();
  return cap > (size_t)40 * 1024 ? cap - (size_t)40 * 1024 : 0;
}
extern "C" int hssk_ulv_node_fits(int m, int r, int rv, int ra, int rb, int rva, int rvb) {
  if (m <= 0 || m > UN_MMAX || r < 0 || r > m || ra + rb != m) return 0;
  return sizeof(double) * un_lds_doubles(m, r, rv, ra, rb, rva, rvb) <= un_lds_cap() ? 1 : 0;
}

extern "C" int hssk_ulv_node_vbatched(hssk_ctx* ctx, const hssk_ulvnode_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  int mmax = 0;
  size_t dyn = 0;
  for (int i = 0; i < count; i++) {
    const hssk_ulvnode_desc& d = descs[i];
    if ((d.eliminate && d.r >= d.m) || !hssk_ulv_node_fits(d.m, d.r, d.rv, d.ra, d.rb, d.rva, d.rvb)) HSSK_UNSUPPORTED("node beyond the fused step");
    mmax = std::max(mmax, d.m);
    dyn = std::max(dyn, sizeof(double) * un_lds_doubles(d.m, d.r, d.rv, d.ra, d.rb, d.rva, d.rvb));
  }
  auto* dd = (const hssk_ulvnode_desc*)ctx->stage(descs, sizeof(*descs) * count);
  if (mmax <= 64) {
    hssk_rt::allow_dynamic_lds(ulv_node_kernel<4, 1>, dyn);
    HSSK_LAUNCH((ulv_node_kernel<4, 1>), dim3((unsigned)count), dim3(UN_T), dyn, ctx->stream, dd);
  } else {
    hssk_rt::allow_dynamic_lds(ulv_node_kernel<8, 2>, dyn);
    HSSK_LAUNCH((ulv_node_kernel<8, 2>), dim3((unsigned)count), dim3(UN_T), dyn, ctx->stream, dd);
  }
  un_launches++;
  hssk_rt::check_launch();
  HSSK_API_END
}
