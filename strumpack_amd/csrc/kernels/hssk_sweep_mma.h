// Single-launch tree sweeps for MANY right-hand sides: the node arithmetic of hssk_sweep.hip on the FP64 matrix cores.
// Included by hssk_sweep.hip inside its anonymous namespace (shares its hand-off protocol, sentinel and helpers).
//
// The vector forms of the sweeps are GEMVs: every matrix element loaded costs one LDS read and one FMA per right-hand side
// and pass, and a node takes one pass of ~10 barrier-separated stages per group of 4 (16) right-hand sides.  With 64
// right-hand sides the per-node operations are (r x K) x (K x 64) GEMMs: here a workgroup keeps the node's vectors as
// [row][64] blocks in LDS (65 doubles per row: conflict-free both for the transposing loads / stores -- lanes along a column
// of the global block -- and for the MFMA operand reads -- lanes along a row), and each stage is a set of 16-row tiles of
// v_mfma_f64_16x16x4_f64: wave w owns tiles w, w + 4, ...; a tile reads its A fragment once (global, prefetched into
// registers before the workgroup waits for its dependencies) and sweeps the four 16-column tiles of the right-hand sides.
// One pass per node and 64 right-hand sides: a sixteenth (a quarter) of the passes of the 4-wide (16-wide) vector form.
//
// Same descriptors, same hand-off buffers, same arithmetic as the vector bodies (reference: HSSMatrix::solve_fwd / solve_bwd,
// HSS/HSSMatrix.solve.hpp:69-238; apply_fwd / apply_bwd, HSS/HSSMatrix.apply.hpp:55-220).

// right-hand sides per pass: NC = 64, or 32 / 16 when the node's vectors would not fit the LDS as 64-wide rows (and for fewer
// right-hand sides); an LDS row is NC + 1 doubles
#define MM_LDR (NC + 1)
#define MM_T ((int)blockDim.x)   /* threads of the workgroup: 256, or more waves for the launches with large nodes */
#define MM_NC NC
// k-steps (of 4) per chunk of the pipelined tile loop: 4 for the 16-wide form (small nodes: K = 41 is three chunks); 8 for the
// wider forms, which serve the launches with large nodes (the leaves: their blocks stream from HBM and a wave needs more
// loads in flight)
#define MM_CH (NC == 16 ? 4 : 8)
constexpr int MM_GRP = 8;         // hand-off loads a thread keeps in flight
constexpr size_t MM_LDS_BYTES = 160 * 1024 - 512;   // what a workgroup may take
// out (M x nc) (op)= op(A) x:  A is M x K (lda) or, trans, K x M (lda) applied transposed; x, out: LDS row blocks
struct MatOp {
  const double* A;
  int lda, M, K;
  int x, o;   // first elements of the operand / result row blocks: offsets into the workgroup's LDS (in doubles)
  int op, trans;
};

// A fragment element (i, k) of an operation, indices clamped into the block (no branch around the load, no select behind it:
// either would make the compiler wait for the load long before the matrix cores need it).  Rows beyond M give rows of the
// result that are never stored; steps beyond K are cancelled on the other operand (see mm_tile).  Only called with M, K > 0.
__device__ __forceinline__ double mm_aload(const MatOp& o, int i, int k) {
  const int ic = min(i, o.M - 1), kc = min(k, o.K - 1);
  return o.trans ? hssk_gload(o.A, (size_t)kc + (size_t)ic * o.lda) : hssk_gload(o.A, (size_t)ic + (size_t)kc * o.lda);
}
// one 16-row tile against the NC / 16 column tiles of the right-hand sides (all of them: columns beyond the last right-hand
// side hold whatever the LDS held and are never stored).  Wave collective: the whole wave calls.  The k loop runs on chunks
// of MM_CH k-steps with two register sets in turn: the A fragments of the next chunk are in flight while the matrix cores
// work on the current one, and nothing but the MFMAs themselves reads them.  (No conditional MFMAs -- the compiler copies
// the whole accumulator tuple around each --, k-steps beyond K multiply by zeros read in place of the right-hand sides.)
template <int NC>
__device__ __forceinline__ void mm_chunk(const MatOp& o, const double* xb, int k0, const double (&a)[MM_CH], hssk_d4 (&acc)[NC / 16]) {
  // (all LDS operands of the chunk first, unconditionally and from a clamped row; k-steps beyond K are cancelled by a ZERO
  // FACTOR on the A fragment.  Written as `k < K ? xr[..] : 0.` every read sat under a branch of its own with a full wait
  // behind it, right in front of its product: ~230 cycles per MFMA with one wave per SIMD -- round 3's "open" item.)
  double xv[MM_CH][NC / 16];
#pragma unroll
  for (int u = 0; u < MM_CH; u++) {
    const double* xr = xb + min(k0 + 4 * u, o.K - 1) * MM_LDR;
#pragma unroll
    for (int ct = 0; ct < NC / 16; ct++) xv[u][ct] = xr[ct * 16];
  }
#pragma unroll
  for (int u = 0; u < MM_CH; u++) {
    const double au = a[u] * (k0 + 4 * u < o.K ? 1. : 0.);
#pragma unroll
    for (int ct = 0; ct < NC / 16; ct++) acc[ct] = hssk_mfma_f64_16x16x4(au, xv[u][ct], acc[ct]);
  }
}
template <int NC>
__device__ __forceinline__ void mm_tile(const MatOp& o, int i0) {
  HSSK_DYN_SHARED(double, S);
  constexpr int NCT = NC / 16;
  const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
  hssk_d4 acc[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ct++) acc[ct] = hssk_d4{0., 0., 0., 0.};
  const int i = i0 + n;
  const int nch = hssk_uniform((o.K + 4 * MM_CH - 1) / (4 * MM_CH));
  const double* xb = S + o.x + n;
  if (nch > 0) {   // (K == 0: nothing to load, and no address to clamp to)
    double a0[MM_CH], a1[MM_CH];
#pragma unroll
    for (int u = 0; u < MM_CH; u++) a0[u] = mm_aload(o, i, 4 * u + kq);
    for (int c = 0; c < nch; c += 2) {
      const int k0 = 4 * MM_CH * c + kq;
#pragma unroll
      for (int u = 0; u < MM_CH; u++) a1[u] = mm_aload(o, i, k0 + 4 * MM_CH + 4 * u);
      mm_chunk<NC>(o, xb, k0, a0, acc);
      if (c + 1 < nch) {
#pragma unroll
        for (int u = 0; u < MM_CH; u++) a0[u] = mm_aload(o, i, k0 + 8 * MM_CH + 4 * u);
        mm_chunk<NC>(o, xb, k0 + 4 * MM_CH, a1, acc);
      }
    }
  }
  double* ob = S + o.o + n;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = i0 + kq + 4 * r;
    if (row < o.M) {
#pragma unroll
      for (int ct = 0; ct < NCT; ct++) apply_op(ob + row * MM_LDR + ct * 16, acc[ct][r], o.op);
    }
  }
}
// a stage: all tiles of the stacked operations (wave w: tiles w, w + 4, ...), then a barrier.  x and out of an operation must
// not overlap.
template <int NC, int NOPS>
__device__ __forceinline__ void mm_stage(const MatOp (&ops)[NOPS]) {
  const int wave = hssk_uniform((int)(threadIdx.x >> 6));   // (a scalar: the tile's operands stay in scalar registers)
  const int nw = MM_T >> 6;   // waves (a power of two)
  int t0 = 0;   // tiles of the operations before this one
#pragma unroll
  for (int j = 0; j < NOPS; j++) {
    const int nt = (ops[j].M + 15) >> 4;
    for (int lt = (wave - t0) & (nw - 1); lt < nt; lt += nw) mm_tile<NC>(ops[j], lt * 16);
    t0 += nt;
  }
  __syncthreads();
}
// pulls the operands of a stage towards this XCD's L2 while the workgroup still waits for its dependencies
__device__ __forceinline__ void mm_touch_block(const double* p, size_t count, double& sink) {
  if (!p) return;
  const size_t T = blockDim.x;
  for (size_t e = (size_t)threadIdx.x * 16; e < count; e += T * 16 * 8) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { const size_t x = e + (size_t)u * T * 16; t[u] = x < count ? hssk_gload(p, x) : 0.; }
#pragma unroll
    for (int u = 0; u < 8; u++) sink += t[u];
  }
}
template <int NOPS>
__device__ __forceinline__ void mm_touch(const MatOp (&ops)[NOPS], double& sink) {
#pragma unroll
  for (int j = 0; j < NOPS; j++)
    if (ops[j].M > 0 && ops[j].K > 0) mm_touch_block(ops[j].A, (size_t)ops[j].lda * ((ops[j].trans ? ops[j].M : ops[j].K) - 1) + (ops[j].trans ? ops[j].K : ops[j].M), sink);
}

// LDS row block (rows x nc) <- column-major global block (leading dimension ld), rows through `perm` (LDS or global ints)
// if given.  HANDED: the block is handed over inside the launch -- every element is polled until it is no longer the
// sentinel; the loads of a batch are all in flight before the first one is examined.
template <int NC, bool HANDED>
__device__ __forceinline__ void mm_take(const double* src, size_t ld, int rows, int nc, double* dst, const int* perm, int* err) {
  if (rows <= 0) return;
  const int tid = threadIdx.x;
  const int cs = MM_T / rows;   // rows <= MM_T
  const int i = tid % rows, cq = tid / rows;
  if (cq >= cs) return;
  const size_t ri = perm ? (size_t)perm[i] : (size_t)i;
  for (int c = cq; c < nc; c += cs * MM_GRP) {
    double v[MM_GRP];
#pragma unroll
    for (int u = 0; u < MM_GRP; u++) {
      const int cc = c + u * cs;
      v[u] = cc < nc ? (HANDED ? hssk_cload(src, ri + (size_t)cc * ld) : hssk_gload(src, ri + (size_t)cc * ld)) : 0.;
    }
    if (HANDED) {
      // (one loop over the whole batch: per-element polling loops make the compiler shuffle the batch's registers around)
      long spins = 0;
      for (;;) {
        bool again = false;
#pragma unroll
        for (int u = 0; u < MM_GRP; u++) again = again || is_sentinel(v[u]);
        if (!again) break;
        hssk_pause();
        if (++spins > SW_SPIN_LIMIT) { hssk_flag_raise(err); break; }
#pragma unroll
        for (int u = 0; u < MM_GRP; u++)
          if (is_sentinel(v[u])) v[u] = hssk_cload(src, ri + (size_t)(c + u * cs) * ld);
      }
    }
#pragma unroll
    for (int u = 0; u < MM_GRP; u++) {
      const int cc = c + u * cs;
      if (cc < nc) dst[(size_t)i * MM_LDR + cc] = v[u];
    }
  }
}
// column-major global block <- LDS row block; COHERENT: the block is handed to another workgroup of the launch
template <int NC, bool COHERENT>
__device__ __forceinline__ void mm_put(double* dst, size_t ld, int rows, int nc, const double* src) {
  if (rows <= 0) return;
  const int tid = threadIdx.x;
  const int cs = MM_T / rows;
  const int i = tid % rows, cq = tid / rows;
  if (cq >= cs) return;
  for (int c = cq; c < nc; c += cs) {
    const double v = src[(size_t)i * MM_LDR + c];
    if (COHERENT) hssk_cstore(dst, (size_t)i + (size_t)c * ld, v);
    else hssk_gstore(dst, (size_t)i + (size_t)c * ld, v);
  }
}
// LDS -> LDS: dst rows [0, rows) <- src rows [r0, r0 + rows)  (lanes along the right-hand sides)
template <int NC>
__device__ __forceinline__ void mm_copy(double* dst, const double* src, int rows, int nc) {
  for (int e = threadIdx.x; e < rows * MM_NC; e += MM_T) {
    const int c = e % NC, i = e / NC;
    if (c < nc) dst[(size_t)i * MM_LDR + c] = src[(size_t)i * MM_LDR + c];
  }
}

// ---- LDS budgets (rows of MM_LDR doubles) of the three bodies: the host checks them, the kernels lay their blocks out by them
struct FwdRows { int f, y, a, t, z; };
// A leaf below the root takes the slim layout: its right-hand-side rows go straight through permU into [ft1; y] (no block for f,
// none for the children's z), and f only names the 64-row block buffer of the substitution -- 300 rows instead of 550 for a
// 195-row leaf of rank 41: three workgroups per CU at 16 right-hand sides per pass.
__host__ __device__ inline FwdRows mm_fwd_rows(int m, int r, int mv, int rv, bool root, bool inner) {
  FwdRows R;
  const int q = m - r;
  if (!inner && !root) {
    R.t = r;
    R.y = q;
    R.z = rv > 1 ? rv : 1;
    R.f = q < SW_NB ? (q > 1 ? q : 1) : SW_NB;
    R.a = 1;   // (unused; keeps the sum an upper bound when r == 0)
    return R;
  }
  if (!inner) mv = rv;   // (a leaf has no children's z: nothing to stack, nothing beyond its own z)
  R.f = m > 1 ? m : 1;
  R.y = (mv - rv > q ? mv - rv : q);
  if (R.y < 1) R.y = 1;
  R.a = (inner && mv > 1) ? mv : 1;
  R.t = root ? (m < SW_NB ? m : SW_NB) : r;
  if (R.t < 1) R.t = 1;
  R.z = rv > 1 ? rv : 1;
  return R;
}

template <int NC>
__device__ __forceinline__ void ulv_fwd_body_mma(const hssk_sweep_fwd_desc* __restrict__ descs, int node, int nrhs_total, int* err, int group) {
  HSSK_DYN_SHARED(double, s_dyn);
  hssk_sweep_fwd_desc p = descs[node];
  const int tid = threadIdx.x;
  const int m = p.m, r = p.r, q = m - r, rv = p.rv, mv = p.mv;
  const bool root = p.LU != nullptr;
  const FwdRows R = mm_fwd_rows(m, r, mv, rv, root, p.B01 != nullptr);
  // (row blocks by their offsets into the workgroup's LDS: the stages address them through s_dyn itself, so the compiler
  //  emits LDS instructions, not flat ones)
  const bool slim = p.B01 == nullptr && !root;   // (mm_fwd_rows: [ft1; y], z, block buffer)
  const int o_f = slim ? (R.t + R.y + R.z) * MM_LDR : 0;   // f, later the block right-hand side of the substitution
  const int o_y = slim ? R.t * MM_LDR : o_f + R.f * MM_LDR;   // zc(permV[rv:]) first, then y
  const int o_a = slim ? o_f : o_y + R.y * MM_LDR;          // stacked children z (inner nodes)
  const int o_t = slim ? 0 : o_a + R.a * MM_LDR;            // ft1 (root: block right-hand side)
  const int o_z = slim ? (R.t + R.y) * MM_LDR : o_t + R.t * MM_LDR;   // z
  const int o_end = slim ? o_f + (R.f + R.a) * MM_LDR : o_z + R.z * MM_LDR;
  double *s_f = s_dyn + o_f, *s_y = s_dyn + o_y, *s_a = s_dyn + o_a, *s_t = s_dyn + o_t, *s_z = s_dyn + o_z;
  int* s_pu = (int*)(s_dyn + o_end);   // permU or the root's pivots (m), then permV (mv)
  int* s_pv = s_pu + max(m, 1);
  const int c0 = group * MM_NC;
  const int nc = min(MM_NC, nrhs_total - c0);
  if (c0) {
    p.fsrc += (size_t)c0 * p.ldf;
    if (p.zc) p.zc += (size_t)c0 * p.ldz_in;
    if (p.ft1) p.ft1 += (size_t)c0 * p.ldp;
    if (p.y) p.y += (size_t)c0 * q;
    if (p.z) p.z += (size_t)c0 * p.ldz;
    if (p.xroot) p.xroot += (size_t)c0 * p.ldxr;
  }
  const bool inner = p.B01 != nullptr;
  if (tid < m) s_pu[tid] = root ? p.piv[tid] : p.permU[tid];
  if (inner && !root && tid < mv) s_pv[tid] = p.permV[tid];
  const bool zpart = inner && !root && rv > 0;
  const int mz = (zpart && mv > rv) ? rv : 0;
  // f(0:rU0) -= B01 zc(rV0:), f(rU0:) -= B10 zc(0:rV0)  and  z += XV zc(permV[rv:])
  const MatOp ops1[3] = {{p.B01, max(p.rU0, 1), inner ? p.rU0 : 0, p.rV1, o_a + p.rV0 * MM_LDR, o_f, OP_SUB, 0},
                         {p.B10, max(p.rU1, 1), inner ? m - p.rU0 : 0, p.rV0, o_a, o_f + p.rU0 * MM_LDR, OP_SUB, 0},
                         {p.XV, max(rv, 1), mz, mv - rv, o_y, o_z, OP_ADD, 0}};
  // y -= XU^T ft1
  const MatOp ops2[1] = {{p.XU, max(r, 1), (q > 0 && r > 0) ? q : 0, r, o_t, o_y, OP_SUB, 1}};
  // ft1 -= WQ y  and  z += Vt0^T y
  const MatOp ops4[2] = {{p.WQ, max(r, 1), q > 0 ? r : 0, q, o_y, o_t, OP_SUB, 0}, {p.Vt0T, max(rv, 1), q > 0 ? rv : 0, q, o_y, o_z, OP_ADD, 0}};
  {   // (leaves too: their blocks come from HBM, and the stages below would meet them one memory round trip at a time)
    double sink = 0.;
    mm_touch(ops1, sink);
    if (root) {
      mm_touch_block(p.LU, (size_t)m * m, sink);
      mm_touch_block(p.TinvL, (size_t)((m + SW_NB - 1) / SW_NB) * SW_NB * SW_NB, sink);
      mm_touch_block(p.TinvU, (size_t)((m + SW_NB - 1) / SW_NB) * SW_NB * SW_NB, sink);
    } else if (q > 0) {
      mm_touch(ops2, sink);
      mm_touch(ops4, sink);
      mm_touch_block(p.Tinv, (size_t)((q + SW_NB - 1) / SW_NB) * SW_NB * SW_NB, sink);
      if (q > SW_NB) mm_touch_block(p.Rlq, (size_t)m * q, sink);
    }
    keep(sink, s_f);
  }
  // ---- f = rhs rows (leaf) or [ft1_0; ft1_1] (inner: handed over by the children); zc = stacked children z
  if (inner) {
    mm_take<NC, true>(p.fsrc, (size_t)p.ldf, m, nc, s_f, nullptr, err);
    mm_take<NC, true>(p.zc, (size_t)p.ldz_in, mv, nc, s_a, nullptr, err);
  } else if (slim) {
    __syncthreads();   // (permU is in LDS)
    mm_take<NC, false>(p.fsrc, (size_t)p.ldf, m, nc, s_t, s_pu, err);   // rows through permU: [ft1; y] in place
  } else mm_take<NC, false>(p.fsrc, (size_t)p.ldf, m, nc, s_f, nullptr, err);
  __syncthreads();
  if (inner) {
    if (zpart) {
      // s_z <- zc(permV[0:rv]);  s_y <- zc(permV[rv:])
      for (int e = tid; e < mv * MM_NC; e += MM_T) {
        const int c = e % NC, i = e / NC;
        if (c < nc) {
          const double v = s_a[(size_t)s_pv[i] * MM_LDR + c];
          if (i < rv) s_z[(size_t)i * MM_LDR + c] = v;
          else s_y[(size_t)(i - rv) * MM_LDR + c] = v;
        }
      }
      __syncthreads();
    }
    mm_stage<NC>(ops1);
  }
  if (root) {
    // ---- root: x = U^{-1} L^{-1} P f, block substitution with the inverted 64 x 64 diagonal blocks
    if (tid < nc)
      for (int i = 0; i < m; i++) {
        const int pi = s_pu[i];
        if (pi != i) { const double a = s_f[(size_t)i * MM_LDR + tid]; s_f[(size_t)i * MM_LDR + tid] = s_f[(size_t)pi * MM_LDR + tid]; s_f[(size_t)pi * MM_LDR + tid] = a; }
      }
    __syncthreads();
    for (int b0 = 0, blk = 0; b0 < m; b0 += SW_NB, blk++) {
      const int nb = min(SW_NB, m - b0);
      mm_copy<NC>(s_t, s_f + (size_t)b0 * MM_LDR, nb, nc);
      __syncthreads();
      const MatOp oL[1] = {{p.TinvL + (size_t)blk * SW_NB * SW_NB, SW_NB, nb, nb, o_t, o_f + b0 * MM_LDR, OP_SET, 0}};
      mm_stage<NC>(oL);
      const int rest = m - b0 - nb;
      if (rest > 0) {
        mm_copy<NC>(s_t, s_f + (size_t)b0 * MM_LDR, nb, nc);
        __syncthreads();
        const MatOp oR[1] = {{p.LU + (b0 + nb) + (size_t)b0 * m, m, rest, nb, o_t, o_f + (b0 + nb) * MM_LDR, OP_SUB, 0}};
        mm_stage<NC>(oR);
      }
    }
    for (int blk = (m - 1) / SW_NB; blk >= 0; blk--) {
      const int b0 = blk * SW_NB, nb = min(SW_NB, m - b0);
      mm_copy<NC>(s_t, s_f + (size_t)b0 * MM_LDR, nb, nc);
      __syncthreads();
      const MatOp oU[1] = {{p.TinvU + (size_t)blk * SW_NB * SW_NB, SW_NB, nb, nb, o_t, o_f + b0 * MM_LDR, OP_SET, 0}};
      mm_stage<NC>(oU);
      if (b0 > 0) {
        mm_copy<NC>(s_t, s_f + (size_t)b0 * MM_LDR, nb, nc);
        __syncthreads();
        const MatOp oR[1] = {{p.LU + (size_t)b0 * m, m, b0, nb, o_t, o_f, OP_SUB, 0}};
        mm_stage<NC>(oR);
      }
    }
    mm_put<NC, false>(p.xroot, (size_t)p.ldxr, m, nc, s_f);
    return;
  }
  // ---- ft1 = f(perm[0:r]) -> s_t, y = f(perm[r:]) -> s_y
  for (int e = tid; e < (slim ? 0 : m * MM_NC); e += MM_T) {
    const int c = e % NC, i = e / NC;
    if (c < nc) {
      const double v = s_f[(size_t)s_pu[i] * MM_LDR + c];
      if (i < r) s_t[(size_t)i * MM_LDR + c] = v;
      else s_y[(size_t)(i - r) * MM_LDR + c] = v;
    }
  }
  if (!inner)
    for (int e = tid; e < rv * MM_NC; e += MM_T) s_z[(size_t)(e / NC) * MM_LDR + (e % NC)] = 0.;
  __syncthreads();
  if (q > 0) {
    if (r > 0) mm_stage<NC>(ops2);
    // ---- y <- R~^{-T} y on 64-row blocks: y_b = Linv_b y_b, then rows below -= R~(b, below)^T y_b
    for (int b0 = 0, blk = 0; b0 < q; b0 += SW_NB, blk++) {
      const int nb = min(SW_NB, q - b0);
      mm_copy<NC>(s_f, s_y + (size_t)b0 * MM_LDR, nb, nc);
      __syncthreads();
      const MatOp o3[1] = {{p.Tinv + (size_t)blk * SW_NB * SW_NB, SW_NB, nb, nb, o_f, o_y + b0 * MM_LDR, OP_SET, 0}};
      mm_stage<NC>(o3);
      const int rest = q - b0 - nb;
      if (rest > 0) {
        mm_copy<NC>(s_f, s_y + (size_t)b0 * MM_LDR, nb, nc);
        __syncthreads();
        const MatOp oR[1] = {{p.Rlq + b0 + (size_t)(b0 + nb) * m, m, rest, nb, o_f, o_y + (b0 + nb) * MM_LDR, OP_SUB, 1}};
        mm_stage<NC>(oR);
      }
    }
    mm_put<NC, false>(p.y, (size_t)q, q, nc, s_y);
    mm_stage<NC>(ops4);
  }
  mm_put<NC, true>(p.ft1, (size_t)p.ldp, r, nc, s_t);
  mm_put<NC, true>(p.z, (size_t)p.ldz, rv, nc, s_z);
}

// xcd != 0 (a launch of leaves only: no dependencies between its workgroups): the groups of a node sit on workgroup ids that
// are congruent mod 8 and adjacent in time -- workgroup b runs on XCD b % 8 and every XCD has its own L2, so the groups,
// which stream the same blocks, fetch them from HBM once; the grid is padded to a multiple of 8 x groups.
template <int NC, int TB>
__global__ __launch_bounds__(TB) void ulv_fwd_sweep_mma_kernel(const hssk_sweep_fwd_desc* __restrict__ descs, int count, int nrhs_total, int ngroups, int xcd, int* err) {
  const int b = blockIdx.x;
  const int node = xcd ? (b / (8 * ngroups)) * 8 + (b & 7) : b / ngroups;
  if (node >= count) return;
  for (int g = xcd ? (b >> 3) % ngroups : b % ngroups; g * MM_NC < nrhs_total; g += ngroups) {
    ulv_fwd_body_mma<NC>(descs, node, nrhs_total, err, g);
    __syncthreads();
  }
}

// ---- backward:  x_c = Q~(:, 0:q) y + Q~(:, q:) xpart ; m == r: x_c = xpart
template <int NC>
__device__ __forceinline__ void ulv_bwd_body_mma(const hssk_sweep_bwd_desc* __restrict__ descs, int node, int nrhs_total, int* err, int group) {
  HSSK_DYN_SHARED(double, s_dyn);
  hssk_sweep_bwd_desc p = descs[node];
  const int m = p.m, r = p.r, q = m - r;
  const int o_v = 0, o_o = max(m, 1) * MM_LDR;          // [y; xpart], the result
  double *s_v = s_dyn + o_v, *s_o = s_dyn + o_o;
  const int c0 = group * MM_NC;
  const int nc = min(MM_NC, nrhs_total - c0);
  if (c0) {
    if (p.y) p.y += (size_t)c0 * q;
    p.xpart += (size_t)c0 * p.ldx;
    p.out += (size_t)c0 * p.ldo;
  }
  const MatOp oy[1] = {{p.Qt, max(m, 1), q > 0 ? m : 0, q, o_v, o_o, OP_SET, 0}};
  const MatOp ox[1] = {{p.Qt + (size_t)q * m, max(m, 1), (q > 0 && r > 0) ? m : 0, r, o_v + q * MM_LDR, o_o, OP_ADD, 0}};
  { double sink = 0.; mm_touch(oy, sink); mm_touch(ox, sink); keep(sink, s_v); }
  // the parent-independent part first
  mm_take<NC, false>(p.y, (size_t)q, q, nc, s_v, nullptr, err);
  __syncthreads();
  if (q > 0) mm_stage<NC>(oy);
  mm_take<NC, true>(p.xpart, (size_t)p.ldx, r, nc, s_v + (size_t)q * MM_LDR, nullptr, err);
  __syncthreads();
  if (q > 0) {
    if (r > 0) mm_stage<NC>(ox);
    mm_put<NC, true>(p.out, (size_t)p.ldo, m, nc, s_o);
  } else mm_put<NC, true>(p.out, (size_t)p.ldo, m, nc, s_v);
}

template <int NC>
__global__ __launch_bounds__(SW_T) void ulv_bwd_sweep_mma_kernel(const hssk_sweep_bwd_desc* __restrict__ descs, int nrhs_total, int ngroups, int* err) {
  const int node = blockIdx.x / ngroups;
  for (int g = blockIdx.x % ngroups; g * MM_NC < nrhs_total; g += ngroups) {
    ulv_bwd_body_mma<NC>(descs, node, nrhs_total, err, g);
    __syncthreads();
  }
}

// ---- mat-vec (inner nodes; the leaves of a many-right-hand-side product run as batched launches)
__host__ __device__ inline int mm_apply_down_rows(int nt1, int ro, int nto, int mo, int acc) {
  const int x = nt1 > ro ? nt1 : ro;
  return (x > 1 ? x : 1) + (acc ? 0 : (nto > mo ? nto : mo)) + (mo - ro > 1 ? mo - ro : 1);
}
template <int NC>
__device__ __forceinline__ void apply_body_mma(const hssk_apply_up_desc* __restrict__ ups, int nup,
                                               const hssk_apply_down_desc* __restrict__ downs, int node, int nrhs_total, int* err, int group) {
  HSSK_DYN_SHARED(double, s_dyn);
  const int tid = threadIdx.x;
  const int c0 = group * MM_NC;
  const int nc = min(MM_NC, nrhs_total - c0);
  if (node < nup) {
    // tmp1 = V^H src = src(perm[0:r]) + X src(perm[r:])   (X is r x (m - r))
    hssk_apply_up_desc p = ups[node];
    p.src += (size_t)c0 * p.lds;
    p.dst += (size_t)c0 * p.ldd;
    const int m = p.m, r = p.r;
    double* s_o = s_dyn;                                 // rows [0, r): src(perm[0:r]); rows [r, m): src(perm[r:])
    const MatOp ou[1] = {{p.X, max(r, 1), (m > r && r > 0) ? r : 0, m - r, r * MM_LDR, 0, OP_ADD, 0}};
    const bool handed = p.inner != 0;
    { double sink = 0.; mm_touch(ou, sink); keep(sink, s_o); }
    if (handed) mm_take<NC, true>(p.src, (size_t)p.lds, m, nc, s_o, p.perm, err);
    else mm_take<NC, false>(p.src, (size_t)p.lds, m, nc, s_o, p.perm, err);
    __syncthreads();
    mm_stage<NC>(ou);
    mm_put<NC, true>(p.dst, (size_t)p.ldd, r, nc, s_o);
    return;
  }
  hssk_apply_down_desc p = downs[node - nup];
  if (c0) {
    if (p.tmp2) p.tmp2 += (size_t)c0 * p.ld2;
    if (p.t1) p.t1 += (size_t)c0 * p.ldt1;
    p.out += (size_t)c0 * p.ldo;
  }
  const int mo = p.mo, ro = p.ro;
  const bool expand = p.tmp2 && ro > 0;
  const int nt1 = p.ri_a + p.ri_b, nto = p.ro_a + p.ro_b;
  const int o_x = 0;                                          // t1, later tmp2
  const int o_o = o_x + max(max(nt1, ro), 1) * MM_LDR;        // the node's result (nto rows; mo == nto below the root)
  const int o_g = o_o + (p.acc ? 0 : max(nto, mo)) * MM_LDR;  // X^T tmp2
  double *s_x = s_dyn + o_x, *s_o = s_dyn + o_o, *s_g = s_dyn + o_g;
  int* s_perm = (int*)(s_g + (size_t)max(mo - ro, 1) * MM_LDR);
  if (expand && tid < mo) s_perm[tid] = p.perm[tid];
  // t = [B01 t1_1; B10 t1_0]  (transposed: [B10^T t1_1; B01^T t1_0])
  const MatOp oB[2] = {
      p.trans ? MatOp{p.B10, max(p.ri_b, 1), p.ro_a, p.ri_b, o_x + p.ri_a * MM_LDR, o_o, OP_SET, 1}
              : MatOp{p.B01, max(p.ro_a, 1), p.ro_a, p.ri_b, o_x + p.ri_a * MM_LDR, o_o, OP_SET, 0},
      p.trans ? MatOp{p.B01, max(p.ri_a, 1), p.ro_b, p.ri_a, o_x, o_o + p.ro_a * MM_LDR, OP_SET, 1}
              : MatOp{p.B10, max(p.ro_b, 1), p.ro_b, p.ri_a, o_x, o_o + p.ro_a * MM_LDR, OP_SET, 0}};
  const MatOp oX[1] = {{p.X, max(ro, 1), (expand && mo > ro) ? mo - ro : 0, ro, o_x, o_g, OP_SET, 1}};
  { double sink = 0.; mm_touch(oB, sink); mm_touch(oX, sink); keep(sink, s_x); }
  if (p.acc) {
    // a leaf whose op(D) x + beta y is already in `out` (a batched launch next to the sweep): out += U tmp2 in place
    if (expand) {
      mm_take<NC, true>(p.tmp2, (size_t)p.ld2, ro, nc, s_x, nullptr, err);
      __syncthreads();
      if (mo > ro) mm_stage<NC>(oX);
      const int cs = MM_T / mo, i = tid % mo, cq = tid / mo;   // mo <= MM_T
      if (cq < cs) {
        const size_t row = (size_t)s_perm[i];
        for (int c = cq; c < nc; c += cs) {
          const double v = i < ro ? s_x[(size_t)i * MM_LDR + c] : s_g[(size_t)(i - ro) * MM_LDR + c];
          hssk_gstore(p.out, row + (size_t)c * p.ldo, hssk_gload(p.out, row + (size_t)c * p.ldo) + v);
        }
      }
    }
    return;
  }
  mm_take<NC, true>(p.t1, (size_t)p.ldt1, nt1, nc, s_x, nullptr, err);
  __syncthreads();
  mm_stage<NC>(oB);
  if (expand) {
    // + U tmp2:  out(perm[k]) += tmp2(k), k < ro ;  out(perm[ro + j]) += sum_k X(k, j) tmp2(k)
    mm_take<NC, true>(p.tmp2, (size_t)p.ld2, ro, nc, s_x, nullptr, err);
    __syncthreads();
    if (mo > ro) mm_stage<NC>(oX);
    for (int e = tid; e < mo * MM_NC; e += MM_T) {
      const int c = e % NC, i = e / NC;
      if (c < nc) s_o[(size_t)s_perm[i] * MM_LDR + c] += i < ro ? s_x[(size_t)i * MM_LDR + c] : s_g[(size_t)(i - ro) * MM_LDR + c];
    }
    __syncthreads();
  }
  mm_put<NC, true>(p.out, (size_t)p.ldo, nto, nc, s_o);
}

template <int NC>
__global__ __launch_bounds__(SW_T) void apply_sweep_mma_kernel(const hssk_apply_up_desc* __restrict__ ups, int nup,
                                                               const hssk_apply_down_desc* __restrict__ downs, int nrhs_total, int ngroups, int* err) {
  const int node = blockIdx.x / ngroups;
  for (int g = blockIdx.x % ngroups; g * MM_NC < nrhs_total; g += ngroups) {
    apply_body_mma<NC>(ups, nup, downs, node, nrhs_total, err, g);
    __syncthreads();
  }
}

// the matrix-core bodies serve nrhs >= HSSK_SWEEP_MMA (default 16; 0 switches them off) when every node fits the LDS budget
int mma_min_nrhs() {
  static const int v = [] {
    const char* e = std::getenv("HSSK_SWEEP_MMA");
    return e ? std::atoi(e) : 16;
  }();
  return v;
}
std::atomic<long long> mma_launches{0};   // sweeps issued in the matrix-core form (tests assert that the path was taken)
size_t mma_lds_bytes(int nc, int rows, int ints) { return sizeof(double) * (size_t)rows * (nc + 1) + sizeof(int) * (size_t)ints; }
// Right-hand sides per workgroup pass and groups side by side.  A node's GEMMs are FP64-rate bound (a 16x16x4 MFMA is 64
// cycles on a SIMD: ~7 us for the forward step of a rank-41 inner node with 64 right-hand sides on one CU), so the
// right-hand sides are split into groups of 16 that run as SEPARATE workgroups, interleaved in the launch order (workgroup
// index = node * groups + group: every group of a child still precedes every group of its parent, and the groups of a level
// fill the chip together instead of one chain after the other).  HSSK_SWEEP_MMA_NC = 16 / 32 / 64 overrides the width,
// HSSK_SWEEP_MMA_GROUPS the number of groups side by side (further groups in turn inside the workgroups).
int mma_width(int nrhs, int rows, int ints, int dmax) {
  static const int want = [] { const char* e = std::getenv("HSSK_SWEEP_MMA_NC"); return e ? std::atoi(e) : 16; }();
  // (launches with large nodes -- the leaf level --: every group of right-hand sides streams the node's blocks again, so wider
  //  groups; HSSK_SWEEP_MMA_NC_BIG overrides)
  static const int want_big = [] { const char* e = std::getenv("HSSK_SWEEP_MMA_NC_BIG"); return e ? std::atoi(e) : 16; }();
  const int w = dmax >= 160 ? want_big : want;
  for (int nc = (w == 64 || w == 32) ? w : 16; nc >= 16; nc /= 2) {
    if (nc >= 2 * nrhs && nc > 16) continue;
    if (mma_lds_bytes(nc, rows, ints) <= MM_LDS_BYTES) return nc;
  }
  return 0;
}
int mma_groups(int nrhs, int nc) {
  static const int lim = [] { const char* e = std::getenv("HSSK_SWEEP_MMA_GROUPS"); return e ? std::max(1, std::atoi(e)) : 16; }();
  return std::min((nrhs + nc - 1) / nc, lim);
}

#undef MM_LDR
#undef MM_T
#undef MM_CH
#undef MM_NC
