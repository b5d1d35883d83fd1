// HBM-bound helper kernels of the HSS engine: generators, gathers/scatters, transposes, norms.
// All are pure streaming kernels (bound: HBM / L2 bandwidth); consecutive lanes touch consecutive
// addresses of the column-major operands wherever the access pattern allows it.
#include "hssk_device.h"
#include "hssk_internal.h"
#include "hssk_gen.h"

#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------
// Toeplitz test matrix (test/test_HSS_seq.cpp:75-78, :86-90), generated in place in HBM.
// ------------------------------------------------------------------------------------------------
__global__ void fill_toeplitz_kernel(double* __restrict__ A, int rows, long long lda, int upper, int i0, int j0) {
  // grid: (ceil(rows/256), cols): blockIdx.y = column, x covers rows -> coalesced column writes
  const int jl = blockIdx.y, j = j0 + jl;
  hssk_gen g;
  g.kind = upper ? HSSK_GEN_TOEPLITZ_UPPER : HSSK_GEN_TOEPLITZ;
  for (int il = blockIdx.x * blockDim.x + threadIdx.x; il < rows; il += gridDim.x * blockDim.x) {
    A[il + (size_t)jl * lda] = hssk_gen_eval(g, i0 + il, j);
  }
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator + Box-Muller.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3,
                                             unsigned k0, unsigned k1) {
  const unsigned long long M0 = 0xD2511F53ull, M1 = 0xCD9E8D57ull;
  unsigned long long p0 = M0 * c0, p1 = M1 * c2;
  unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
  unsigned n1 = (unsigned)p1;
  unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
  unsigned n3 = (unsigned)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ double philox_normal(unsigned long long seed, unsigned long long ctr) {
  unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0x9E3779B9u, c3 = 0x243F6A88u;
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; r++) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  // two uniforms in (0,1): 53-bit mantissas
  unsigned long long a = ((unsigned long long)c0 << 21) ^ (unsigned long long)(c1 >> 11);
  unsigned long long b = ((unsigned long long)c2 << 21) ^ (unsigned long long)(c3 >> 11);
  double u1 = ((double)(a & ((1ull << 53) - 1)) + 0.5) * (1.0 / 9007199254740992.0);
  double u2 = ((double)(b & ((1ull << 53) - 1)) + 0.5) * (1.0 / 9007199254740992.0);
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}
__global__ void randn_kernel(double* __restrict__ P, int rows, long long cols, long long ld, int row0,
                             long long stride, unsigned long long seed) {
  long long total = (long long)rows * cols;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    int r = (int)(e % rows);
    long long c = e / rows;
    P[r + c * ld] = philox_normal(seed, (unsigned long long)(row0 + r) * (unsigned long long)stride + (unsigned long long)c);
  }
}

// ------------------------------------------------------------------------------------------------
// column gather / scatter: one workgroup per (problem, column chunk)
// ------------------------------------------------------------------------------------------------
struct Work2 {
  int prob, chunk;
};
constexpr int COLS_PER_WG = 8;

__global__ void gather_cols_kernel(const hssk_colgather_desc* __restrict__ descs, const Work2* __restrict__ work) {
  const Work2 w = work[blockIdx.x];
  const hssk_colgather_desc p = descs[w.prob];
  int jend = min(p.ncols, (w.chunk + 1) * COLS_PER_WG);
  for (int j = w.chunk * COLS_PER_WG; j < jend; j++) {
    int js = p.idx ? p.idx[j] : j;
    const double* s = p.scatter ? p.src + (size_t)j * p.lds : p.src + (size_t)js * p.lds;
    double* d = p.scatter ? p.dst + (size_t)js * p.ldd : p.dst + (size_t)j * p.ldd;
    for (int i = threadIdx.x; i < p.rows; i += blockDim.x) d[i] = s[i];
  }
}

// row gather / scatter (vectors of the apply / solve sweeps: rows = HSS node rows, cols = nrhs)
__global__ void gather_rows_kernel(const hssk_rowgather_desc* __restrict__ descs, const Work2* __restrict__ work) {
  const Work2 w = work[blockIdx.x];
  const hssk_rowgather_desc p = descs[w.prob];
  int jend = min(p.cols, (w.chunk + 1) * COLS_PER_WG);
  for (int j = w.chunk * COLS_PER_WG; j < jend; j++) {
    const double* s = p.src + (size_t)j * p.lds;
    double* d = p.dst + (size_t)j * p.ldd;
    for (int i = threadIdx.x; i < p.nrows; i += blockDim.x) {
      int ii = p.idx ? p.idx[i] : i;
      double v = p.scatter ? s[i] : s[ii];
      double* o = p.scatter ? d + ii : d + i;
      *o = p.accumulate ? (*o + v) : v;
    }
  }
}

// element gather B(i,j) = A(I[i], J[j])
__global__ void gather_elems_kernel(const hssk_elem_desc* __restrict__ descs, const Work2* __restrict__ work) {
  const Work2 w = work[blockIdx.x];
  const hssk_elem_desc p = descs[w.prob];
  int jend = min(p.n, (w.chunk + 1) * COLS_PER_WG);
  for (int j = w.chunk * COLS_PER_WG; j < jend; j++) {
    long long gj = p.J ? p.J[j] : (p.j0 + j);
    const bool cin = p.chi <= p.clo || (gj >= p.clo && gj < p.chi);
    const double* col = p.A + (long long)gj * p.lda;
    for (int i = threadIdx.x; i < p.m; i += blockDim.x) {
      long long gi = p.I ? p.I[i] : (p.i0 + i);
      const bool rin = p.rhi <= p.rlo || (gi >= p.rlo && gi < p.rhi);
      double v = (cin && rin) ? col[gi] : 0.;
      if (p.transpose) p.B[j + (size_t)i * p.ldb] = v;
      else p.B[i + (size_t)j * p.ldb] = v;
    }
  }
}

// transposed gathers of large blocks, B(j, i) = A(I[i], J[j]): a 64 (i) x 32 (j) tile through the LDS, so that the reads run
// along the columns of A and the writes along the columns of B (the loop above writes with stride ldb: the 196 blocks
// 472 x 512 of a leaf-512 factorization took 0.52 ms = 1.6 TB/s)
struct Work3e {
  int prob, ti, tj;
};
__global__ __launch_bounds__(256) void gather_elems_t_kernel(const hssk_elem_desc* __restrict__ descs, const Work3e* __restrict__ work) {
  HSSK_SHARED double tile[32 * 65];
  const Work3e w = work[blockIdx.x];
  const hssk_elem_desc p = descs[w.prob];
  const int tid = threadIdx.x;
  {
    const int i = w.ti * 64 + (tid & 63);
    long long gi = 0;
    bool rin = false;
    if (i < p.m) {
      gi = p.I ? p.I[i] : (p.i0 + i);
      rin = p.rhi <= p.rlo || (gi >= p.rlo && gi < p.rhi);
    }
#pragma unroll
    for (int jj = tid >> 6; jj < 32; jj += 4) {
      const int j = w.tj * 32 + jj;
      double v = 0.;
      if (i < p.m && j < p.n) {
        const long long gj = p.J ? p.J[j] : (p.j0 + j);
        const bool cin = p.chi <= p.clo || (gj >= p.clo && gj < p.chi);
        if (cin && rin) v = p.A[gi + gj * p.lda];
      }
      tile[jj * 65 + (tid & 63)] = v;
    }
  }
  __syncthreads();
  const int jj = tid & 31, j = w.tj * 32 + jj;
  for (int ii = tid >> 5; ii < 64; ii += 8) {
    const int i = w.ti * 64 + ii;
    if (i < p.m && j < p.n) p.B[j + (size_t)i * p.ldb] = tile[jj * 65 + ii];
  }
}

// the same with the matrix given by a formula: B(i,j) = G(I[i], J[j])
__global__ void gen_elems_kernel(hssk_gen g, const hssk_elem_desc* __restrict__ descs, const Work2* __restrict__ work) {
  const Work2 w = work[blockIdx.x];
  const hssk_elem_desc p = descs[w.prob];
  int jend = min(p.n, (w.chunk + 1) * COLS_PER_WG);
  for (int j = w.chunk * COLS_PER_WG; j < jend; j++) {
    long long gj = p.J ? p.J[j] : (p.j0 + j);
    const bool cin = p.chi <= p.clo || (gj >= p.clo && gj < p.chi);
    for (int i = threadIdx.x; i < p.m; i += blockDim.x) {
      long long gi = p.I ? p.I[i] : (p.i0 + i);
      const bool rin = p.rhi <= p.rlo || (gi >= p.rlo && gi < p.rhi);
      double v = (cin && rin) ? hssk_gen_eval(g, (int)gi, (int)gj) : 0.;
      if (p.transpose) p.B[j + (size_t)i * p.ldb] = v;
      else p.B[i + (size_t)j * p.ldb] = v;
    }
  }
}
// a dense block of it: A(il, jl) = trans ? G(j0 + jl, i0 + il) : G(i0 + il, j0 + jl); grid (row chunks, columns)
__global__ void gen_fill_kernel(hssk_gen g, double* __restrict__ A, long long rows, long long lda, long long i0, long long j0, int trans) {
  const long long jl = blockIdx.y;
  for (long long il = (long long)blockIdx.x * blockDim.x + threadIdx.x; il < rows; il += (long long)gridDim.x * blockDim.x)
    A[il + jl * lda] = trans ? hssk_gen_eval(g, (int)(j0 + jl), (int)(i0 + il)) : hssk_gen_eval(g, (int)(i0 + il), (int)(j0 + jl));
}

// real image of a block of another scalar type (hssk_expand_image): grid (row chunks, columns), a thread per scalar
template <typename R, bool CPLX>
__global__ void expand_image_kernel(double* __restrict__ dst, long long ldd, const R* __restrict__ src, long long lds, long long rows) {
  const long long j = blockIdx.y;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (long long)gridDim.x * blockDim.x) {
    if (!CPLX) {
      dst[i + j * ldd] = (double)src[i + j * lds];
    } else {
      const double re = (double)src[2 * (i + j * lds)], im = (double)src[2 * (i + j * lds) + 1];
      double* c0 = dst + 2 * i + 2 * j * ldd;
      c0[0] = re; c0[1] = im;
      c0[ldd] = -im; c0[ldd + 1] = re;
    }
  }
}

__global__ void sum_slabs_kernel(const double* __restrict__ slabs, long long count, long long stride, int nslab, double* __restrict__ out) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (long long)gridDim.x * blockDim.x) {
    double s = 0.;
    for (int g = 0; g < nslab; g++) s += slabs[(size_t)g * stride + e];
    out[e] = s;
  }
}

// upper-trapezoidal copy: dst(i, j) = src(i, j) for i <= j, 0 below (stacking the R factors of a TSQR tree)
__global__ void copy_triu_kernel(const hssk_triu_desc* __restrict__ descs, const Work2* __restrict__ work) {
  const Work2 w = work[blockIdx.x];
  const hssk_triu_desc p = descs[w.prob];
  const int jend = min(p.cols, (w.chunk + 1) * COLS_PER_WG);
  const int st = p.dstride > 1 ? p.dstride : 1;
  for (int j = w.chunk * COLS_PER_WG; j < jend; j++)
    for (int i = threadIdx.x; i < p.rows; i += blockDim.x)
      hssk_gstore(p.dst, (size_t)i * st + (size_t)j * p.ldd, i <= j ? hssk_gload(p.src, i + (size_t)j * p.lds) : 0.);
}

// transpose through a padded LDS tile: dst(c, r) = src(r, c)
struct Work3 {
  int prob, tr, tc;
};
__global__ void transpose_kernel(const hssk_transpose_desc* __restrict__ descs, const Work3* __restrict__ work) {
  HSSK_SHARED double tile[32 * 33];
  const Work3 w = work[blockIdx.x];
  const hssk_transpose_desc p = descs[w.prob];
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int y = ty; y < 32; y += 8) {
    int r = w.tr * 32 + tx, c = w.tc * 32 + y;
    tile[y * 33 + tx] = (r < p.rows && c < p.cols) ? p.src[r + (size_t)c * p.lds] : 0.;
  }
  __syncthreads();
  for (int y = ty; y < 32; y += 8) {
    int c = w.tc * 32 + tx, r = w.tr * 32 + y;
    if (r < p.rows && c < p.cols) p.dst[c + (size_t)r * p.ldd] = tile[tx * 33 + y];
  }
}

// sum of squares of a panel, one workgroup per problem
__global__ void sumsq_kernel(const hssk_norm_desc* __restrict__ descs) {
  HSSK_SHARED double part[4];
  const hssk_norm_desc p = descs[blockIdx.x];
  double s = 0.;
  long long total = (long long)p.rows * p.cols;
  for (long long e = threadIdx.x; e < total; e += blockDim.x) {
    int r = (int)(e % p.rows);
    long long c = e / p.rows;
    double v = p.P[r + c * p.ld];
    s += v * v;
  }
  s = hssk_wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *p.out = part[0] + part[1] + part[2] + part[3];
}

// dense interpolative basis: out(perm[k], :) = row k of [I; X^T]
__global__ void basis_dense_kernel(const hssk_basis_desc* __restrict__ descs) {
  const hssk_basis_desc p = descs[blockIdx.x];
  for (int e = threadIdx.x; e < p.m * p.r; e += blockDim.x) {
    int k = e % p.m, j = e / p.m;
    double v = (k < p.r) ? (k == j ? 1. : 0.) : p.X[j + (size_t)(k - p.r) * p.ldx];
    p.out[p.perm[k] + (size_t)j * p.ldo] = v;
  }
}

__global__ void shift_diag_kernel(const hssk_shift_desc* __restrict__ descs, double sigma) {
  const hssk_shift_desc p = descs[blockIdx.x];
  for (int i = threadIdx.x; i < p.n; i += blockDim.x) p.A[i + (size_t)i * p.lda] += sigma;
}

__global__ void shift_diag_cplx_kernel(const hssk_shift_desc* __restrict__ descs, double re, double im) {
  const hssk_shift_desc p = descs[blockIdx.x];
  for (int i = threadIdx.x; i < p.n; i += blockDim.x) {
    p.A[i + (size_t)i * p.lda] += re;
    if ((i & 1) == 0 && i + 1 < p.n) {
      p.A[i + (size_t)(i + 1) * p.lda] -= im;
      p.A[(i + 1) + (size_t)i * p.lda] += im;
    }
  }
}

// FP64 matrix-core peak probe: 4 independent accumulators per wave, no memory traffic
__global__ __launch_bounds__(256) void mfma_peak_kernel(double* out, int iters) {
  hssk_d4 c0 = {0., 0., 0., 0.}, c1 = c0, c2 = c0, c3 = c0;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int i = 0; i < iters; i++) {
    c0 = hssk_mfma_f64_16x16x4(a, b, c0);
    c1 = hssk_mfma_f64_16x16x4(a, b, c1);
    c2 = hssk_mfma_f64_16x16x4(a, b, c2);
    c3 = hssk_mfma_f64_16x16x4(a, b, c3);
  }
  hssk_d4 s = c0 + c1 + c2 + c3;
  if (s[0] + s[1] + s[2] + s[3] == -1.0) out[0] = s[0];
}

__global__ __launch_bounds__(256, 2) void mfma_probe_kernel(double* out, long long* clk, int iters, double av, double bv) {
  // 12 independent accumulators, 48 MFMAs per iteration, no memory traffic: the pure issue rate
  hssk_d4 c[12];
#pragma unroll
  for (int i = 0; i < 12; i++) c[i] = hssk_d4{0., 0., 0., 0.};
  double a0 = av * (1.0 + threadIdx.x * 1e-3), b0 = bv * (1.0 - threadIdx.x * 1e-3);
  double a1 = a0 * 0.5, b1 = b0 * 0.25;
  long long t0 = hssk_clock(), w0 = hssk_wallclock();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
#pragma unroll
      for (int q = 0; q < 12; q += 2) {
        c[q] = hssk_mfma_f64_16x16x4(a0, b0, c[q]);
        c[q + 1] = hssk_mfma_f64_16x16x4(a1, b0, c[q + 1]);
      }
#pragma unroll
      for (int q = 0; q < 12; q += 2) {
        c[q] = hssk_mfma_f64_16x16x4(a0, b1, c[q]);
        c[q + 1] = hssk_mfma_f64_16x16x4(a1, b1, c[q + 1]);
      }
    }
  }
  long long t1 = hssk_clock(), w1 = hssk_wallclock();
  double s = 0.;
#pragma unroll
  for (int i = 0; i < 12; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  if (s == -1.0) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <class Desc, class F>
std::vector<Work2> make_work2(const Desc* descs, int count, F ncols_of) {
  std::vector<Work2> w;
  for (int p = 0; p < count; p++) {
    int nc = ncols_of(descs[p]);
    for (int c = 0; c * COLS_PER_WG < nc; c++) w.push_back(Work2{p, c});
  }
  return w;
}

}  // namespace

extern "C" {

int hssk_fill_toeplitz(hssk_ctx* ctx, double* A, int n, long long lda, char kind) {
  HSSK_API_BEGIN
  if (n <= 0) return 0;
  dim3 grid((unsigned)std::min(64, (n + 255) / 256), (unsigned)n);
  HSSK_LAUNCH(fill_toeplitz_kernel, grid, dim3(256), 0, ctx->stream, A, n, lda, (int)(kind == 'U'), 0, 0);
  hssk_rt::check_launch();
  HSSK_API_END
}

int hssk_fill_toeplitz_block(hssk_ctx* ctx, double* A, int rows, int cols, long long lda, int i0, int j0, char kind) {
  HSSK_API_BEGIN
  if (rows <= 0 || cols <= 0) return 0;
  dim3 grid((unsigned)std::min(64, (rows + 255) / 256), (unsigned)cols);
  HSSK_LAUNCH(fill_toeplitz_kernel, grid, dim3(256), 0, ctx->stream, A, rows, lda, (int)(kind == 'U'), i0, j0);
  hssk_rt::check_launch();
  HSSK_API_END
}

int hssk_randn(hssk_ctx* ctx, double* P, int rows, long long cols, long long ld, int row0,
               long long stride, unsigned long long seed) {
  HSSK_API_BEGIN
  long long total = (long long)rows * cols;
  if (total <= 0) return 0;
  unsigned blocks = (unsigned)std::min<long long>((total + 255) / 256, 8192);
  HSSK_LAUNCH(randn_kernel, dim3(blocks), dim3(256), 0, ctx->stream, P, rows, cols, ld, row0, stride, seed);
  hssk_rt::check_launch();
  HSSK_API_END
}

int hssk_gather_cols(hssk_ctx* ctx, const hssk_colgather_desc* descs, int count) {
  HSSK_API_BEGIN
  auto w = make_work2(descs, count, [](const hssk_colgather_desc& d) { return d.rows > 0 ? d.ncols : 0; });
  if (w.empty()) return 0;
  auto* dd = (const hssk_colgather_desc*)ctx->stage(descs, sizeof(*descs) * count);
  auto* dw = (const Work2*)ctx->stage(w.data(), sizeof(Work2) * w.size());
  HSSK_LAUNCH(gather_cols_kernel, dim3((unsigned)w.size()), dim3(256), 0, ctx->stream, dd, dw);
  hssk_rt::check_launch();
  HSSK_API_END
}

int hssk_copy_triu(hssk_ctx* ctx, const hssk_triu_desc* descs, int count) {
  HSSK_API_BEGIN
  auto w = make_work2(descs, count, [](const hssk_triu_desc& d) { return d.rows > 0 ? d.cols : 0; });
  if (w.empty()) return 0;
  auto* dd = (const hssk_triu_desc*)ctx->stage(descs, sizeof(*descs) * count);
  auto* dw = (const Work2*)ctx->stage(w.data(), sizeof(Work2) * w.size());
  HSSK_LAUNCH(copy_triu_kernel, dim3((unsigned)w.size()), dim3(256), 0, ctx->stream, dd, dw);
  hssk_rt::check_launch();
  HSSK_API_END
}

int hssk_gather_rows(hssk_ctx* ctx, const hssk_rowgather_desc* descs, int count) {
  HSSK_API_BEGIN
  auto w = make_work2(descs, count, [](const hssk_rowgather_desc& d) { return d.nrows > 0 ? d.cols : 0; });
  if (w.empty()) return 0;
  auto* dd = (const hssk_rowgather_desc*)ctx->stage(descs, sizeof(*descs) * count);
  auto* dw = (const Work2*)ctx->stage(w.data(), sizeof(Work2) * w.size());
  HSSK_LAUNCH(gather_rows_kernel, dim3((unsigned)w.size()), dim3(256), 0, ctx->stream, dd, dw);
  hssk_rt::check_launch();
  HSSK_API_END
}

int hssk_gather_elems(hssk_ctx* ctx, const hssk_elem_desc* descs, int count) {
  HSSK_API_BEGIN
  auto tiled = [](const hssk_elem_desc& d) { return d.transpose && d.m >= 32 && d.n >= 32; };
  auto w = make_work2(descs, count, [&](const hssk_elem_desc& d) { return d.m > 0 && !tiled(d) ? d.n : 0; });
  std::vector<Work3e> wt;
  for (int p = 0; p < count; p++)
    if (tiled(descs[p]))
      for (int tj = 0; tj * 32 < descs[p].n; tj++)
        for (int ti = 0; ti * 64 < descs[p].m; ti++) wt.push_back(Work3e{p, ti, tj});
  if (w.empty() && wt.empty()) return 0;
  auto* dd = (const hssk_elem_desc*)ctx->stage(descs, sizeof(*descs) * count);
  if (!w.empty()) {
    auto* dw = (const Work2*)ctx->stage(w.data(), sizeof(Work2) * w.size());
    HSSK_LAUNCH(gather_elems_kernel, dim3((unsigned)w.size()), dim3(256), 0, ctx->stream, dd, dw);
  }
  if (!wt.empty()) {
    auto* dw = (const Work3e*)ctx->stage(wt.data(), sizeof(Work3e) * wt.size());
    HSSK_LAUNCH(gather_elems_t_kernel, dim3((unsigned)wt.size()), dim3(256), 0, ctx->stream, dd, dw);
  }
  hssk_rt::check_launch();
  HSSK_API_END
}

static void check_gen(const hssk_gen* g) {
  if (!g || (g->kind != HSSK_GEN_TOEPLITZ && g->kind != HSSK_GEN_TOEPLITZ_UPPER)) throw std::invalid_argument("hssk_gen: unknown generator kind");
}
int hssk_gen_elems(hssk_ctx* ctx, const hssk_gen* g, const hssk_elem_desc* descs, int count) {
  HSSK_API_BEGIN
  check_gen(g);
  auto w = make_work2(descs, count, [](const hssk_elem_desc& d) { return d.m > 0 ? d.n : 0; });
  if (w.empty()) return 0;
  auto* dd = (const hssk_elem_desc*)ctx->stage(descs, sizeof(*descs) * count);
  auto* dw = (const Work2*)ctx->stage(w.data(), sizeof(Work2) * w.size());
  HSSK_LAUNCH(gen_elems_kernel, dim3((unsigned)w.size()), dim3(256), 0, ctx->stream, *g, dd, dw);
  hssk_rt::check_launch();
  HSSK_API_END
}
int hssk_gen_fill(hssk_ctx* ctx, const hssk_gen* g, double* A, long long rows, long long cols, long long lda, long long i0,
                  long long j0, int trans) {
  HSSK_API_BEGIN
  check_gen(g);
  if (rows <= 0 || cols <= 0) return 0;
  if (i0 + rows > 0x7fffffffLL || j0 + cols > 0x7fffffffLL) throw std::invalid_argument("hssk_gen_fill: indices beyond 2^31");
  for (long long c0 = 0; c0 < cols; c0 += 65535) {   // (grid.y limit)
    const long long nc = std::min<long long>(65535, cols - c0);
    dim3 grid((unsigned)std::min<long long>(64, (rows + 255) / 256), (unsigned)nc);
    HSSK_LAUNCH(gen_fill_kernel, grid, dim3(256), 0, ctx->stream, *g, A + c0 * lda, rows, lda, i0, j0 + c0, trans);
  }
  hssk_rt::check_launch();
  HSSK_API_END
}

int hssk_sum_slabs(hssk_ctx* ctx, const double* slabs, long long count, long long stride, int nslab, double* out) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  const unsigned nb = (unsigned)std::min<long long>((count + 255) / 256, 4096);
  HSSK_LAUNCH(sum_slabs_kernel, dim3(nb), dim3(256), 0, ctx->stream, slabs, count, stride, nslab, out);
  hssk_rt::check_launch();
  HSSK_API_END
}

int hssk_transpose(hssk_ctx* ctx, const hssk_transpose_desc* descs, int count) {
  HSSK_API_BEGIN
  std::vector<Work3> w;
  for (int p = 0; p < count; p++)
    for (int tc = 0; tc * 32 < descs[p].cols; tc++)
      for (int tr = 0; tr * 32 < descs[p].rows; tr++) w.push_back(Work3{p, tr, tc});
  if (w.empty()) return 0;
  auto* dd = (const hssk_transpose_desc*)ctx->stage(descs, sizeof(*descs) * count);
  auto* dw = (const Work3*)ctx->stage(w.data(), sizeof(Work3) * w.size());
  HSSK_LAUNCH(transpose_kernel, dim3((unsigned)w.size()), dim3(256), 0, ctx->stream, dd, dw);
  hssk_rt::check_launch();
  HSSK_API_END
}

int hssk_sumsq_vbatched(hssk_ctx* ctx, const hssk_norm_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  auto* dd = (const hssk_norm_desc*)ctx->stage(descs, sizeof(*descs) * count);
  HSSK_LAUNCH(sumsq_kernel, dim3((unsigned)count), dim3(256), 0, ctx->stream, dd);
  hssk_rt::check_launch();
  HSSK_API_END
}

int hssk_basis_dense(hssk_ctx* ctx, const hssk_basis_desc* descs, int count) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  auto* dd = (const hssk_basis_desc*)ctx->stage(descs, sizeof(*descs) * count);
  HSSK_LAUNCH(basis_dense_kernel, dim3((unsigned)count), dim3(256), 0, ctx->stream, dd);
  hssk_rt::check_launch();
  HSSK_API_END
}

int hssk_shift_diag(hssk_ctx* ctx, const hssk_shift_desc* descs, int count, double sigma) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  auto* dd = (const hssk_shift_desc*)ctx->stage(descs, sizeof(*descs) * count);
  HSSK_LAUNCH(shift_diag_kernel, dim3((unsigned)count), dim3(256), 0, ctx->stream, dd, sigma);
  hssk_rt::check_launch();
  HSSK_API_END
}

int hssk_shift_diag_cplx(hssk_ctx* ctx, const hssk_shift_desc* descs, int count, double re, double im) {
  HSSK_API_BEGIN
  if (count <= 0) return 0;
  auto* dd = (const hssk_shift_desc*)ctx->stage(descs, sizeof(*descs) * count);
  HSSK_LAUNCH(shift_diag_cplx_kernel, dim3((unsigned)count), dim3(256), 0, ctx->stream, dd, re, im);
  hssk_rt::check_launch();
  HSSK_API_END
}

double hssk_mfma_f64_peak_tflops(hssk_ctx* ctx, int iters) {
  try {
    const int blocks = 256 * 8;
    double* d = (double*)ctx->scratch(64);
    HSSK_LAUNCH(mfma_peak_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d, 16);  // warm-up
    hssk_rt::event_record(ctx->ev0, ctx->stream);
    HSSK_LAUNCH(mfma_peak_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d, iters);
    hssk_rt::event_record(ctx->ev1, ctx->stream);
    hssk_rt::check_launch();
    float ms = hssk_rt::event_elapsed_ms(ctx->ev0, ctx->ev1);
    ctx->dgemm_timed = false;
    double flops = (double)blocks * 4 /*waves*/ * (double)iters * 4 /*mfma*/ * 2048.0;
    return flops / (ms * 1e-3) * 1e-12;
  } catch (const std::exception& e) {
    hssk_set_error(e.what());
    return -1.;
  }
}

int hssk_mfma_f64_probe(hssk_ctx* ctx, int iters, int waves_per_simd, int zero_data, double* out) {
  HSSK_API_BEGIN
  // 256-thread blocks = 4 waves = one per SIMD; waves_per_simd blocks per CU
  const int blocks = 256 * std::max(1, std::min(2, waves_per_simd));
  double* d = (double*)ctx->scratch(256);
  long long* clk = (long long*)(d + 8);
  const double v = zero_data ? 0.0 : 0.7310585786300049;
  HSSK_LAUNCH(mfma_probe_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d, clk, 64, v, v);
  hssk_rt::event_record(ctx->ev0, ctx->stream);
  HSSK_LAUNCH(mfma_probe_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d, clk, iters, v, v);
  hssk_rt::event_record(ctx->ev1, ctx->stream);
  hssk_rt::check_launch();
  float ms = hssk_rt::event_elapsed_ms(ctx->ev0, ctx->ev1);
  ctx->dgemm_timed = false;
  long long h[2] = {0, 0};
  hssk_rt::d2h(h, clk, sizeof(h), ctx->stream);
  hssk_rt::sync(ctx->stream);
  out[0] = (double)blocks * 4 * (double)iters * 48 * 2048.0 / (ms * 1e-3) * 1e-12;
  out[1] = (double)h[0] / ((double)iters * 48);
  out[2] = h[1] > 0 ? (double)h[0] / ((double)h[1] / 100e6) * 1e-9 : 0.;
  HSSK_API_END
}

int hssk_expand_image(hssk_ctx* ctx, double* dst, long long ldd, const void* src, long long lds, long long rows, long long cols,
                      int dtype) {
  HSSK_API_BEGIN
  if (rows <= 0 || cols <= 0) return 0;
  if (dtype != HSSK_DT_F32 && dtype != HSSK_DT_C32 && dtype != HSSK_DT_C64) HSSK_UNSUPPORTED("unknown scalar type");
  if (lds < rows || ldd < (dtype == HSSK_DT_F32 ? rows : 2 * rows)) HSSK_UNSUPPORTED("leading dimension smaller than the block");
  for (long long c0 = 0; c0 < cols; c0 += 65535) {   // (grid.y limit)
    const long long nc = std::min<long long>(65535, cols - c0);
    dim3 grid((unsigned)std::min<long long>(256, (rows + 255) / 256), (unsigned)nc);
    if (dtype == HSSK_DT_F32)
      HSSK_LAUNCH((expand_image_kernel<float, false>), grid, dim3(256), 0, ctx->stream, dst + c0 * ldd, ldd, (const float*)src + c0 * lds, lds, rows);
    else if (dtype == HSSK_DT_C32)
      HSSK_LAUNCH((expand_image_kernel<float, true>), grid, dim3(256), 0, ctx->stream, dst + 2 * c0 * ldd, ldd, (const float*)src + 2 * c0 * lds, lds, rows);
    else
      HSSK_LAUNCH((expand_image_kernel<double, true>), grid, dim3(256), 0, ctx->stream, dst + 2 * c0 * ldd, ldd, (const double*)src + 2 * c0 * lds, lds, rows);
    hssk_rt::check_launch();
  }
  HSSK_API_END
}

}  // extern "C"
