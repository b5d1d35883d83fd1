// The sketch GEMM  C(m x n) = alpha A(m x k) op(B) + beta C  with m = number of random samples
// (<= a few hundred) and n, k = the matrix dimension (1e5): 98% of all flops of HSS compression
// (reference: AFunctor, HSS/HSSExtra.hpp:236-239, Sr = A R and Sc = A^H R; here in the transposed
// sample layout S^T = R^T op(A), so every operand panel is read contiguously).
//
// Bound: FP64 MFMA (v_mfma_f64_16x16x4_f64; gfx950 FP64 matrix peak 78.6 TFLOP/s).  Arithmetic
// intensity per HBM byte of B is m/4 flop/B (48 at m = 192), far above the 10 flop/B ridge.
//
// Tiling: workgroup = 256 threads (4 wave64 as 2x2), output tile BM x 64 with BM in {64,128,192}
// chosen so one workgroup covers all m sample rows when m <= 192 (B, the N x N matrix, is then
// streamed from HBM exactly once; the A panel -- R^T, 24 KB per k-stage -- is shared by all
// workgroups through L2).  K advances 16 per stage through double-buffered LDS (As[k][i],
// Bs[k][j], row stride +16 doubles so the two 16-lane halves of a ds_read_b64 hit different bank
// halves) with the next stage prefetched into registers while the MFMAs of the current one issue.
// The K range is split over gridDim.z workgroups (deterministic: partial tiles go to scratch and a
// second kernel reduces them in fixed order) so that the workgroup count fills 256 CUs evenly.
#include "hssk_device.h"
#include "hssk_internal.h"
#include "hssk_gen.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BN = 64, BK = 16;

// FULL = true : interior tiles -- m % BM == 0, whole BN columns, k-chunk a multiple of BK, operands
//               16-byte aligned with even leading dimensions: unmasked 16-byte global loads.
// FULL = false: edge tiles / unaligned operands -- masked 8-byte loads.
// TAG only separates the symbol of the short tail launch from the main one (per-kernel profiles stay readable)
template <int BM, bool TRANSB, bool FULL, int TAG = 0>
__global__ __launch_bounds__(256, 2) void dgemm_kernel(int m, long long n, long long k, const double* __restrict__ A,
                                                       long long lda, const double* __restrict__ B, long long ldb,
                                                       double* __restrict__ P, long long ldp, long long pstride,
                                                       long long kchunk, int jtile0, long long* __restrict__ clk) {
  const long long t0_ = clk ? hssk_clock() : 0, w0_ = clk ? hssk_wallclock() : 0;
  constexpr int LDA_S = BM + 16, LDB_S = BN + 16;
  constexpr int WM = BM / 2;       // rows per wave
  constexpr int MT = WM / 16;      // MFMA tiles per wave along M
  constexpr int NT = 2;            // 32 columns per wave
  constexpr int A_PER_THREAD = BM * BK / 256;   // doubles per thread and stage
  constexpr int B_PER_THREAD = BN * BK / 256;
  constexpr int A2 = A_PER_THREAD / 2, B2 = B_PER_THREAD / 2;  // 16-byte pieces
  HSSK_SHARED double As[2 * BK * LDA_S];
  HSSK_SHARED double Bs[2 * BK * LDB_S];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs, so XCD x (= id % 8) is handed a
  // contiguous range of column tiles -- neighbouring tiles share DRAM pages / L2 lines of B within one L2
  int bx = blockIdx.x;
  {
    const int nx = gridDim.x, x = bx & 7, q = nx >> 3, r = nx & 7;
    bx = x * q + (x < r ? x : r) + (bx >> 3);
  }
  const long long j0 = (long long)(bx + jtile0) * BN;
  const int i0 = blockIdx.y * BM;
  const long long kbeg = (long long)blockIdx.z * kchunk;
  const long long kend = kbeg + kchunk < k ? kbeg + kchunk : k;
  const int wm = (wave & 1) * WM, wn = (wave >> 1) * 32;

  hssk_d4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; a++)
#pragma unroll
    for (int b = 0; b < NT; b++) acc[a][b] = hssk_d4{0., 0., 0., 0.};

  const double* Ak = A + i0 + kbeg * lda;  // (advanced by the FULL-path loader)
  const double* Bk = TRANSB ? B + j0 + kbeg * ldb : B + j0 * ldb + kbeg;
  const long long stepA = (long long)BK * lda, stepB = TRANSB ? (long long)BK * ldb : (long long)BK;
  const long long nst = (kend - kbeg + BK - 1) / BK;  // stages

  // MFMAs of one k-stage; the LDS fragments of sub-step ks+1 are requested before the MFMAs of
  // sub-step ks issue, so their latency hides under the 12 x 64-cycle MFMA group
  auto compute = [&](int buf) {
    const double* as = As + buf * BK * LDA_S + wm + l15;
    const double* bs = Bs + buf * BK * LDB_S + wn + l15;
    double af[2][MT], bf[2][NT];
#pragma unroll
    for (int a = 0; a < MT; a++) af[0][a] = as[l4 * LDA_S + a * 16];
#pragma unroll
    for (int b = 0; b < NT; b++) bf[0][b] = bs[l4 * LDB_S + b * 16];
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      const int cur = (ks >> 2) & 1, nxt = cur ^ 1;
      if (ks + 4 < BK) {
#pragma unroll
        for (int a = 0; a < MT; a++) af[nxt][a] = as[(ks + 4 + l4) * LDA_S + a * 16];
#pragma unroll
        for (int b = 0; b < NT; b++) bf[nxt][b] = bs[(ks + 4 + l4) * LDB_S + b * 16];
      }
#pragma unroll
      for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < NT; b++)  // swapped operands: lane holds C[i = l15][j = l4 + 4r]
          acc[a][b] = hssk_mfma_f64_16x16x4(bf[cur][b], af[cur][a], acc[a][b]);
    }
  };

  if (FULL) {
    // k-invariant per-thread global offsets (32-bit, relative to the stage base) and LDS slots
    int offA[A2], ldsA[A2], offB[B2], ldsB[B2];
#pragma unroll
    for (int r = 0; r < A2; r++) {
      int e = tid + 256 * r;                  // pair index: (i2, kk) with i = 2 i2
      int i = 2 * (e % (BM / 2)), kk = e / (BM / 2);
      offA[r] = i + kk * (int)lda;
      ldsA[r] = kk * LDA_S + i;
    }
#pragma unroll
    for (int r = 0; r < B2; r++) {
      int e = tid + 256 * r;
      if (TRANSB) {  // op(B)(k,j) = B(j,k): pairs along j
        int j = 2 * (e % (BN / 2)), kk = e / (BN / 2);
        offB[r] = j + kk * (int)ldb;
        ldsB[r] = kk * LDB_S + j;
      } else {       // op(B)(k,j) = B(k,j): pairs along k
        int kk = 2 * (e % (BK / 2)), j = e / (BK / 2);
        offB[r] = kk + j * (int)ldb;
        ldsB[r] = kk * LDB_S + j;
      }
    }
    // two register sets: the global loads of stage s+2 are issued while stage s computes, so every
    // load has two full MFMA stages (~2.5 us) to land before it is written to LDS
    hssk_d2 ra0[A2], rb0[B2], ra1[A2], rb1[B2];
    auto load = [&](hssk_d2 (&ra)[A2], hssk_d2 (&rb)[B2]) {
#pragma unroll
      for (int r = 0; r < A2; r++) ra[r] = *reinterpret_cast<const hssk_d2*>(Ak + offA[r]);
#pragma unroll
      for (int r = 0; r < B2; r++) rb[r] = *reinterpret_cast<const hssk_d2*>(Bk + offB[r]);
      Ak += stepA; Bk += stepB;
    };
    auto store = [&](int buf, const hssk_d2 (&ra)[A2], const hssk_d2 (&rb)[B2]) {
      double* as = As + buf * BK * LDA_S;
      double* bs = Bs + buf * BK * LDB_S;
#pragma unroll
      for (int r = 0; r < A2; r++) *reinterpret_cast<hssk_d2*>(as + ldsA[r]) = ra[r];
#pragma unroll
      for (int r = 0; r < B2; r++) {
        if (TRANSB) *reinterpret_cast<hssk_d2*>(bs + ldsB[r]) = rb[r];
        else { bs[ldsB[r]] = rb[r][0]; bs[ldsB[r] + LDB_S] = rb[r][1]; }
      }
    };
    load(ra0, rb0);                      // stage 0
    store(0, ra0, rb0);
    if (nst > 1) load(ra1, rb1);         // stage 1
    __syncthreads();
    long long st = 0;
    // LDS buffer of stage s is s & 1; register set of stage s is s & 1 as well
    for (; st + 2 < nst; st += 2) {
      load(ra0, rb0);                    // stage st+2
      compute(0);                        // stage st
      store(1, ra1, rb1);                // stage st+1 (loaded one full stage ago)
      __syncthreads();
      if (st + 3 < nst) load(ra1, rb1);  // stage st+3
      compute(1);                        // stage st+1
      store(0, ra0, rb0);                // stage st+2
      __syncthreads();
    }
    // here LDS[st & 1 == 0] holds stage st; stage st+1 (if any) sits in (ra1, rb1)
    compute(0);
    if (st + 1 < nst) {
      store(1, ra1, rb1);
      __syncthreads();
      compute(1);
    }
  } else {
    double ra[A_PER_THREAD], rb[B_PER_THREAD];
    auto load = [&](long long k0) {
#pragma unroll
      for (int r = 0; r < A_PER_THREAD; r++) {
        int e = tid + 256 * r;
        int i = e % BM, kk = e / BM;
        ra[r] = (i0 + i < m && k0 + kk < kend) ? A[i0 + i + (k0 + kk) * lda] : 0.;
      }
#pragma unroll
      for (int r = 0; r < B_PER_THREAD; r++) {
        int e = tid + 256 * r;
        if (TRANSB) {
          int j = e % BN, kk = e / BN;
          rb[r] = (j0 + j < n && k0 + kk < kend) ? B[j0 + j + (k0 + kk) * ldb] : 0.;
        } else {
          int kk = e % BK, j = e / BK;
          rb[r] = (j0 + j < n && k0 + kk < kend) ? B[k0 + kk + (j0 + j) * ldb] : 0.;
        }
      }
    };
    auto store = [&](int buf) {
      double* as = As + buf * BK * LDA_S;
      double* bs = Bs + buf * BK * LDB_S;
#pragma unroll
      for (int r = 0; r < A_PER_THREAD; r++) {
        int e = tid + 256 * r;
        as[(e / BM) * LDA_S + (e % BM)] = ra[r];
      }
#pragma unroll
      for (int r = 0; r < B_PER_THREAD; r++) {
        int e = tid + 256 * r;
        if (TRANSB) bs[(e / BN) * LDB_S + (e % BN)] = rb[r];
        else bs[(e % BK) * LDB_S + (e / BK)] = rb[r];
      }
    };
    if (nst > 0) { load(kbeg); store(0); }
    __syncthreads();
    int buf = 0;
    long long k0 = kbeg;
    for (long long st = 0; st + 1 < nst; st++) {
      k0 += BK;
      load(k0);
      compute(buf);
      store(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
    if (nst > 0) compute(buf);
  }
  // partial tile -> P (slice blockIdx.z), plain stores; the reduce kernel applies alpha/beta
  double* Pz = P + (long long)blockIdx.z * pstride;
#pragma unroll
  for (int a = 0; a < MT; a++)
#pragma unroll
    for (int b = 0; b < NT; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int gi = i0 + wm + a * 16 + l15;
        long long gj = j0 + wn + b * 16 + l4 + 4 * r;
        if (gi < m && gj < n) Pz[gi + gj * ldp] = acc[a][b][r];
      }
  // shader-clock probe (workgroup 0: elapsed shader cycles and 100 MHz ticks) + per-workgroup trace
  // {start tick, end tick, hw id, tile} for hssk_last_dgemm_trace
  if (clk && threadIdx.x == 0) {
    const long long w1 = hssk_wallclock();
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
      clk[0] = hssk_clock() - t0_;
      clk[1] = w1 - w0_;
    }
    const long long fid = blockIdx.x + (long long)gridDim.x * (blockIdx.y + (long long)gridDim.y * blockIdx.z);
    long long* rec = clk + 4 + 4 * fid;
    rec[0] = w0_; rec[1] = w1; rec[2] = hssk_hwid(); rec[3] = bx;
  }
}

// ---- second form of the interior tiles (round 4): BM x 128 tile on EIGHT waves, operands by LDS DMA ------------------
// What the first form left on the table (main launches at 0.85 of the FP64 MFMA roof): every stage a wave issued 8 global
// loads into registers, 8 LDS stores, a drain of both counters and a barrier in front of LDS reads whose latency only
// the other workgroup of the CU could cover.  Here
//  * one workgroup of 8 waves (2 x 4, 96 x 32 each at BM = 192) owns the CU and a BM x 128 tile: the shared A panel
//    (R^T) is read from L2 once per 128 columns instead of once per 64;
//  * the operands go global -> LDS by global_load_lds_dwordx4 (no staging registers, no ds_write, no address VALU in
//    the loop) into a ring of THREE stages: the copy of stage s+3 is issued right after the barrier that ends the reads
//    of stage s and has two whole stages (~5 us) to land;
//  * one bare s_barrier per stage, placed BEFORE the last k-step's MFMAs of the stage (its fragments are already in
//    registers): after the barrier a wave still has 12 MFMAs to issue while the first fragments of the next stage come
//    back from the LDS, so the matrix pipe never waits for an LDS round trip;
//  * an LDS DMA writes lane-linear (base + 16 lane), so the layout is chosen on the SOURCE side: j- / i-contiguous
//    operands as [64-wide block][k][64] images with the two k rows a 32-lane read group touches in different bank
//    halves (column ^ 16 (k & 1)); a k-contiguous operand (B not transposed) as a [j][16 k] image whose 16-byte k
//    pairs sit at pair ^ ((j >> 1) & 7) -- both conflict-free for ds_read_b64.  The k index a lane group l4 feeds to
//    MFMA sub-step s is pi(s, l4) = 2 s + (l4 & 1) + 8 (l4 >> 1) for BOTH operands (any bijection does: the sum over
//    k is what it is), which is what lets one 16-byte piece of the k-contiguous operand serve two lane groups.
//  * GEN: op(B) is not in memory at all -- it is a formula (hssk_gen) of the DIFFERENCE of its indices, G(i, j) = tau(i - j)
//    (the Toeplitz kinds).  The 16 x 128 tile of a stage then holds only 143 distinct values -- tau on a run of consecutive
//    differences, tile position (k, j) at entry k + 127 - j (a 32-lane read group covers 17 consecutive entries:
//    conflict-free) -- and the run of the next stage is the same run moved on by 16.  The run is the B part of a ring
//    slot: two stages ahead of its use, two waves copy 127 entries from the slot of the stage before (one LDS read, one LDS
//    write per lane) and a third evaluates the 16 new ones.  Only the A panel still travels.  Same tiles, stages and
//    summation order as the stored operand: the results are bitwise equal.
//    (Why so frugal, measured at N = 1e5 against 51.2 ms per launch with the stored operand: every thread evaluating its
//    four entries of the full tile, 61 ms -- the FP64 vector instructions of an IEEE division run on the units the FP64
//    MFMA runs on; one entry per thread of the run with a branch around the MFMAs of the evaluating waves, 64 ms -- the
//    compiler copied the accumulators; a circular window with computed read addresses, 53.4 ms -- two integer
//    instructions per operand read, issued by both waves of a SIMD at once.)
template <int MBLK, bool TRANSB, int TAG = 0, bool GEN = false>
__global__ __launch_bounds__(512, 2) void sketch_kernel(long long n, long long k, const double* __restrict__ A, long long lda,
                                                        const double* __restrict__ B, long long ldb,
                                                        double* __restrict__ P, long long ldp, long long pstride,
                                                        long long kchunk, int jtile0, long long* __restrict__ clk,
                                                        hssk_gen gen, long long jg0) {
  const long long t0_ = clk ? hssk_clock() : 0, w0_ = clk ? hssk_wallclock() : 0;
  constexpr int BM = 64 * MBLK, BN2 = 128;
  constexpr int WM = BM / 2, MT = WM / 16, NT = 2;
  constexpr int A_DBL = BM * BK, B_DBL = BN2 * BK, SLOT = A_DBL + B_DBL;   // doubles per ring stage
  constexpr int NCH = GEN ? MBLK : MBLK + 2;                               // 1 KB copies per wave and stage
  constexpr bool KJ_IMAGE = TRANSB && !GEN;                                // B image [block][k][64] (else [j][16 k], or the run)
  HSSK_DYN_SHARED(double, lds);

  const int tid = threadIdx.x, lane = tid & 63, wave = hssk_uniform(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  int bx = blockIdx.x;
  {  // XCD-aware tile order (see dgemm_kernel)
    const int nx = gridDim.x, x = bx & 7, q = nx >> 3, r = nx & 7;
    bx = x * q + (x < r ? x : r) + (bx >> 3);
  }
  const long long j0 = (long long)(bx + jtile0) * BN2;
  const int i0 = blockIdx.y * BM;
  const long long kbeg = (long long)blockIdx.z * kchunk;
  const long long kend = kbeg + kchunk < k ? kbeg + kchunk : k;
  const int nst = (int)((kend - kbeg) / BK);   // whole stages (the host only sends aligned chunks here)
  const int wm = (wave & 1) * WM, wn = (wave >> 1) * 32;

  // ---- copy side: wave w moves the k rows {2w, 2w+1} of every 64-wide block (i- / j-contiguous operands), or the
  // eight j rows 8 c .. 8 c + 7 of chunk c = w, w + 8 (k-contiguous operand)
  const int kk = lane >> 5, pos = 2 * (lane & 31);
  const double* srcA = A + i0 + (pos ^ (16 * kk)) + (kbeg + 2 * wave + kk) * lda;
  const long long stepA = (long long)BK * lda;
  const double* srcB[2] = {nullptr, nullptr};
  long long stepB = 0;
  if (GEN) {
  } else if (TRANSB) {
    // (the RAGGED last tile -- n is not a multiple of 128; the host sends it only when n is even -- reads the last valid pair
    //  instead of columns beyond n: its surplus columns hold garbage that the reduce pass never folds)
    const long long jp = j0 + (pos ^ (16 * kk));
    srcB[0] = B + (jp < n - 2 ? jp : n - 2) + (kbeg + 2 * wave + kk) * ldb;
    srcB[1] = B + (jp + 64 < n - 2 ? jp + 64 : n - 2) + (kbeg + 2 * wave + kk) * ldb;
    stepB = (long long)BK * ldb;
  } else {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int j = 8 * (wave + 8 * r) + (lane >> 3), q = lane & 7;
      const long long jc = j0 + j < n - 1 ? j0 + j : n - 1;   // (ragged last tile: the last valid column again)
      srcB[r] = B + jc * ldb + kbeg + 2 * (q ^ ((j >> 1) & 7));
    }
    stepB = BK;
  }
  // copy number c (0 .. NCH-1) of a stage: the A blocks first, then the two B pieces.  The stage is clamped: copies past the
  // chunk re-read its last stage into a slot nobody reads any more
  auto copy_one = [&](int stage, int slot, int c) {
    const int sc = stage < nst ? stage : nst - 1;
    double* base = lds + slot * SLOT;
    if (c < MBLK) hssk_glds16(srcA + sc * stepA + 64 * c, base + c * 1024 + wave * 128);
    else if (GEN) {}
    else if (TRANSB) hssk_glds16(srcB[c - MBLK] + sc * stepB, base + A_DBL + (c - MBLK) * 1024 + wave * 128);
    else hssk_glds16(srcB[c - MBLK] + sc * stepB, base + A_DBL + (wave + 8 * (c - MBLK)) * 128);
  };
  // generated operand: entry e of stage s's run is tau(d_first + 16 s + e), d_first = first k of the chunk minus the last
  // column of the tile (transposed, the formula takes the negated difference); it sits at B-part offset e of the stage's slot
  const int gen_dfirst = (int)kbeg - ((int)(jg0 + j0) + 127);
  auto gen_tau = [&](int i) {
    const int d = gen_dfirst + i;
    return TRANSB ? hssk_gen_eval(gen, 0, d) : hssk_gen_eval(gen, d, 0);
  };
  auto copy_stage = [&](int stage, int slot) {
#pragma unroll
    for (int c = 0; c < NCH; c++) copy_one(stage, slot, c);
  };

  // ---- fragment side: offsets (doubles, within a stage) of this lane's operand words for sub-step 0; sub-step s adds
  // 128 (two k rows) in the [k][64] images, and moves to pair (s ^ h) in the [j][16 k] image
  int offA[MT], offB[NT][4];
#pragma unroll
  for (int a = 0; a < MT; a++) {
    const int il = wm + a * 16 + l15;
    offA[a] = hssk_opaque((il >> 6) * 1024 + ((l4 & 1) + 8 * (l4 >> 1)) * 64 + ((il & 63) ^ (16 * (l4 & 1))));
  }
#pragma unroll
  for (int b = 0; b < NT; b++) {
    const int jl = wn + b * 16 + l15;
    if (GEN) {   // run entry of tile position (k of sub-step s, j): k + 127 - j
      const int o = hssk_opaque(A_DBL + (l4 & 1) + 8 * (l4 >> 1) + 127 - jl);
#pragma unroll
      for (int s = 0; s < 4; s++) offB[b][s] = o + 2 * s;
    } else if (KJ_IMAGE) {
      const int o = hssk_opaque(A_DBL + (jl >> 6) * 1024 + ((l4 & 1) + 8 * (l4 >> 1)) * 64 + ((jl & 63) ^ (16 * (l4 & 1))));
#pragma unroll
      for (int s = 0; s < 4; s++) offB[b][s] = o + s * 128;
    } else {
      const int h = (jl >> 1) & 7;
#pragma unroll
      for (int s = 0; s < 4; s++) offB[b][s] = hssk_opaque(A_DBL + jl * 16 + 2 * ((s + 4 * (l4 >> 1)) ^ h) + (l4 & 1));
    }
  }

  hssk_d4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; a++)
#pragma unroll
    for (int b = 0; b < NT; b++) acc[a][b] = hssk_d4{0., 0., 0., 0.};
  double af[2][MT], bf[2][NT];
  auto frags = [&](int slot, int s, int set) {
    const double* base = lds + slot * SLOT;
#pragma unroll
    for (int a = 0; a < MT; a++) af[set][a] = base[offA[a] + s * 128];
#pragma unroll
    for (int b = 0; b < NT; b++) bf[set][b] = base[offB[b][s]];
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
      for (int b = 0; b < NT; b++)  // swapped operands: lane holds C[i = l15][j = l4 + 4r]
        acc[a][b] = hssk_mfma_f64_16x16x4(bf[set][b], af[set][a], acc[a][b]);
  };
  // one stage out of ring slot S; on entry the fragments of its sub-step 0 are in set 0
  auto stage = [&](int st, auto slot_tag) {
    constexpr int S = decltype(slot_tag)::value;
    // (scheduling fences: left alone, the compiler hoists the reads of later sub-steps, fuses them into half-rate
    // ds_read2st64_b64 pairs and then waits for ALL of them in front of the next MFMA)
    // generated operand: the run of stage st + 2 goes into slot S + 2 (released by the previous barrier, read after the next
    // one) -- waves 0 and 1 move 127 entries over from the run of stage st + 1 (read here, written after the first group of
    // MFMAs: the LDS round trip rides them), wave 2 + (st & 1) evaluates the sixteen new ones
    double gen_moved = 0.;
    if (GEN && wave < 2 && tid < 127) gen_moved = lds[((S + 1) % 3) * SLOT + A_DBL + tid + 16];
    frags(S, 1, 1);
    hssk_sched_barrier();
    mfmas(0);
    hssk_sched_barrier();
    if (GEN && wave < 2 && tid < 127) lds[((S + 2) % 3) * SLOT + A_DBL + tid] = gen_moved;
    if (GEN && wave == 2 + (st & 1)) lds[((S + 2) % 3) * SLOT + A_DBL + 127 + l15] = gen_tau(16 * (st + 2) + 127 + l15);
    hssk_sched_barrier();
    frags(S, 2, 0);
    hssk_sched_barrier();
    mfmas(1);
    hssk_sched_barrier();
    frags(S, 3, 1);
    hssk_sched_barrier();
    mfmas(0);
    hssk_sched_barrier();
    // stage st+1 has landed (this wave's pieces; the barrier makes it everyone's), every read of stage st has returned
    hssk_wait_glds<NCH>();
    hssk_wg_barrier();
    hssk_sched_barrier();
    // the last sub-step's MFMAs (fragments already in registers) carry the issue of the next stage's first fragment reads
    // and of the copies of stage st+3 into the slot just released: one copy behind each pair of MFMAs
#pragma unroll
    for (int a = 0; a < MT; a++) {
#pragma unroll
      for (int b = 0; b < NT; b++) acc[a][b] = hssk_mfma_f64_16x16x4(bf[1][b], af[1][a], acc[a][b]);
      hssk_sched_barrier();
      if (a == 0) frags((S + 1) % 3, 0, 0);
      else if (a - 1 < NCH) copy_one(st + 3, S, a - 1);
      hssk_sched_barrier();
    }
#pragma unroll
    for (int c = MT - 1; c < NCH; c++) copy_one(st + 3, S, c);
    hssk_sched_barrier();
  };

  if (nst > 0) {
    copy_stage(0, 0);
    copy_stage(1, 1);
    copy_stage(2, 2);
    if (GEN && tid < 143) {   // the runs of stages 0 and 1 (stage st builds the run of stage st + 2)
      lds[A_DBL + tid] = gen_tau(tid);
      lds[SLOT + A_DBL + tid] = gen_tau(16 + tid);
    }
    hssk_wait_glds<2 * NCH>();
    hssk_wg_barrier();
    frags(0, 0, 0);
    int st = 0;
    for (; st + 3 <= nst; st += 3) {
      stage(st, std::integral_constant<int, 0>());
      stage(st + 1, std::integral_constant<int, 1>());
      stage(st + 2, std::integral_constant<int, 2>());
    }
    if (st < nst) stage(st, std::integral_constant<int, 0>());
    if (st + 1 < nst) stage(st + 1, std::integral_constant<int, 1>());
    hssk_wait_glds<0>();   // the clamped copies of the last stages still target this workgroup's LDS
  }
  double* Pz = P + (long long)blockIdx.z * pstride;
#pragma unroll
  for (int a = 0; a < MT; a++)
#pragma unroll
    for (int b = 0; b < NT; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int gi = i0 + wm + a * 16 + l15;
        const long long gj = j0 + wn + b * 16 + l4 + 4 * r;
        Pz[gi + gj * ldp] = acc[a][b][r];
      }
  if (clk && threadIdx.x == 0) {
    const long long w1 = hssk_wallclock();
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
      clk[0] = hssk_clock() - t0_;
      clk[1] = w1 - w0_;
    }
    const long long fid = blockIdx.x + (long long)gridDim.x * (blockIdx.y + (long long)gridDim.y * blockIdx.z);
    long long* rec = clk + 4 + 4 * fid;
    rec[0] = w0_; rec[1] = w1; rec[2] = hssk_hwid(); rec[3] = bx;
  }
}

// C = alpha * sum_z P_z + beta * C   (fixed summation order -> deterministic)
__global__ void dgemm_reduce_kernel(int m, long long n, const double* __restrict__ P, long long ldp,
                                    long long pstride, int nz, double alpha, double beta,
                                    double* __restrict__ C, long long ldc) {
  long long total = (long long)m * n;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    int i = (int)(e % m);
    long long j = e / m;
    double s = 0.;
    for (int z = 0; z < nz; z++) s += P[i + j * ldp + z * pstride];
    double* c = C + i + j * ldc;
    double v = alpha * s;
    if (beta != 0.) v += beta * (*c);
    *c = v;
  }
}

// The same for FEW output elements with MANY partials (the ragged edge tile of the sketch: 192 x 32 elements, 256 partials --
// a thread per element walked them one dependent-latency step at a time, 104 us at N = 1e5): 16 elements x 16 z-lanes per
// workgroup, every lane sums its partials z = lane, lane + 16, ... , the 16 lane sums are added in lane order (fixed
// summation order -> deterministic).
__global__ void dgemm_reduce_wide_kernel(int m, long long n, const double* __restrict__ P, long long ldp,
                                         long long pstride, int nz, double alpha, double beta,
                                         double* __restrict__ C, long long ldc) {
  HSSK_SHARED double s_part[256];
  const int el = threadIdx.x & 15, zl = threadIdx.x >> 4;
  const long long total = (long long)m * n;
  for (long long e0 = (long long)blockIdx.x * 16; e0 < total; e0 += (long long)gridDim.x * 16) {
    const long long e = e0 + el;
    const int i = (int)(e % m);
    const long long j = e / m;
    double s = 0.;
    if (e < total) {
      const double* p = P + i + j * ldp;
      for (int z = zl; z < nz; z += 16) s += p[z * pstride];
    }
    s_part[threadIdx.x] = s;
    __syncthreads();
    if (zl == 0 && e < total) {
      double t = 0.;
      for (int q = 0; q < 16; q++) t += s_part[el + 16 * q];
      double* c = C + i + j * ldc;
      double v = alpha * t;
      if (beta != 0.) v += beta * (*c);
      *c = v;
    }
    __syncthreads();
  }
}

template <int BM, bool FULL, int TAG>
void launch_dgemm(hssk_ctx* ctx, int transB, dim3 grid, int m, long long n, long long k, const double* A,
                  long long lda, const double* B, long long ldb, double* P, long long ldp, long long pstride,
                  long long kchunk, int jtile0, long long* clk) {
  if (grid.x == 0) return;
  if (transB)
    HSSK_LAUNCH((dgemm_kernel<BM, true, FULL, TAG>), grid, dim3(256), 0, ctx->stream, m, n, k, A, lda, B, ldb, P, ldp, pstride, kchunk, jtile0, clk);
  else
    HSSK_LAUNCH((dgemm_kernel<BM, false, FULL, TAG>), grid, dim3(256), 0, ctx->stream, m, n, k, A, lda, B, ldb, P, ldp, pstride, kchunk, jtile0, clk);
}

}  // namespace

namespace {
template <bool FULL, int TAG>
void launch_bm(int BM, hssk_ctx* ctx, int transB, dim3 grid, int m, long long n, long long k, const double* A,
               long long lda, const double* B, long long ldb, double* P, long long ldp, long long pstride,
               long long kchunk, int jtile0, long long* clk) {
  if (BM == 192) launch_dgemm<192, FULL, TAG>(ctx, transB, grid, m, n, k, A, lda, B, ldb, P, ldp, pstride, kchunk, jtile0, clk);
  else if (BM == 128) launch_dgemm<128, FULL, TAG>(ctx, transB, grid, m, n, k, A, lda, B, ldb, P, ldp, pstride, kchunk, jtile0, clk);
  else launch_dgemm<64, FULL, TAG>(ctx, transB, grid, m, n, k, A, lda, B, ldb, P, ldp, pstride, kchunk, jtile0, clk);
}

// the eight-wave LDS-DMA form (sketch_kernel): BM = 64 MBLK rows, 128 columns per workgroup, three ring stages
constexpr int BN2 = 128;
inline size_t sketch_lds_bytes(int mblk) { return sizeof(double) * 3 * (size_t)(64 * mblk + BN2) * BK; }
template <int MBLK, int TAG>
void launch_sketch_m(hssk_ctx* ctx, int transB, dim3 grid, long long n, long long k, const double* A, long long lda,
                     const double* B, long long ldb, double* P, long long ldp, long long pstride, long long kchunk,
                     int jtile0, long long* clk, const hssk_gen* gen, long long jg0) {
  const size_t shm = sketch_lds_bytes(MBLK);
  const hssk_gen g0{0, 0, {0., 0., 0., 0.}};
  if (gen) {
    if (transB) HSSK_LAUNCH((sketch_kernel<MBLK, true, TAG, true>), grid, dim3(512), shm, ctx->stream, n, k, A, lda, B, ldb, P, ldp, pstride, kchunk, jtile0, clk, *gen, jg0);
    else HSSK_LAUNCH((sketch_kernel<MBLK, false, TAG, true>), grid, dim3(512), shm, ctx->stream, n, k, A, lda, B, ldb, P, ldp, pstride, kchunk, jtile0, clk, *gen, jg0);
  } else {
    if (transB) HSSK_LAUNCH((sketch_kernel<MBLK, true, TAG, false>), grid, dim3(512), shm, ctx->stream, n, k, A, lda, B, ldb, P, ldp, pstride, kchunk, jtile0, clk, g0, jg0);
    else HSSK_LAUNCH((sketch_kernel<MBLK, false, TAG, false>), grid, dim3(512), shm, ctx->stream, n, k, A, lda, B, ldb, P, ldp, pstride, kchunk, jtile0, clk, g0, jg0);
  }
}
// asks the runtime for the ring's LDS once per instantiation; false where the device cannot give it (the caller then keeps
// the four-wave form, which needs 74 KB at most)
template <int MBLK>
bool sketch_prepare_m() {
  static const bool ok = [] {
    try {
      const size_t b = sketch_lds_bytes(MBLK);
      if (b > hssk_rt::max_lds_per_workgroup()) return false;
      hssk_rt::allow_dynamic_lds(sketch_kernel<MBLK, true, 0, false>, b);
      hssk_rt::allow_dynamic_lds(sketch_kernel<MBLK, false, 0, false>, b);
      hssk_rt::allow_dynamic_lds(sketch_kernel<MBLK, true, 1, false>, b);
      hssk_rt::allow_dynamic_lds(sketch_kernel<MBLK, false, 1, false>, b);
      hssk_rt::allow_dynamic_lds(sketch_kernel<MBLK, true, 0, true>, b);
      hssk_rt::allow_dynamic_lds(sketch_kernel<MBLK, false, 0, true>, b);
      hssk_rt::allow_dynamic_lds(sketch_kernel<MBLK, true, 1, true>, b);
      hssk_rt::allow_dynamic_lds(sketch_kernel<MBLK, false, 1, true>, b);
      return true;
    } catch (const std::exception&) {
      return false;
    }
  }();
  return ok;
}
inline bool sketch_prepare(int BM) {
  return BM == 192 ? sketch_prepare_m<3>() : (BM == 128 ? sketch_prepare_m<2>() : sketch_prepare_m<1>());
}
template <int TAG>
void launch_sketch(int BM, hssk_ctx* ctx, int transB, dim3 grid, long long n, long long k, const double* A, long long lda,
                   const double* B, long long ldb, double* P, long long ldp, long long pstride, long long kchunk,
                   int jtile0, long long* clk, const hssk_gen* gen, long long jg0) {
  if (grid.x == 0) return;
  if (BM == 192) launch_sketch_m<3, TAG>(ctx, transB, grid, n, k, A, lda, B, ldb, P, ldp, pstride, kchunk, jtile0, clk, gen, jg0);
  else if (BM == 128) launch_sketch_m<2, TAG>(ctx, transB, grid, n, k, A, lda, B, ldb, P, ldp, pstride, kchunk, jtile0, clk, gen, jg0);
  else launch_sketch_m<1, TAG>(ctx, transB, grid, n, k, A, lda, B, ldb, P, ldp, pstride, kchunk, jtile0, clk, gen, jg0);
}
}  // namespace

// gen != nullptr: op(B)(kk, j) = transB ? G(jg0 + j, kk) : G(kk, jg0 + j) is evaluated inside the kernel (B, ldb unused)
static void dgemm_impl(hssk_ctx* ctx, int transB, int m, long long n, long long k, double alpha, const double* A, long long lda,
                       const double* B, long long ldb, const hssk_gen* gen, long long jg0, double beta, double* C, long long ldc) {
  if (m <= 0 || n <= 0) return;
  const int BM = m > 128 ? 192 : (m > 64 ? 128 : 64);
  const unsigned gm = (unsigned)((m + BM - 1) / BM);
  const long long ksteps = (k + BK - 1) / BK;
  // interior tiles take an unmasked kernel; the ragged last columns (and any unaligned / odd-sized problem) the masked one
  const bool aligned = (m % BM == 0) && (k % BK == 0) && (lda % 2 == 0) && (gen || ldb % 2 == 0) &&
                       (((size_t)A | (gen ? (size_t)0 : (size_t)B)) % 16 == 0);
  // Interior tiles: the eight-wave LDS-DMA form (BM x 128 per workgroup, ONE workgroup per CU) where the device has the
  // LDS for its three ring stages (120 KB at BM = 192; gfx950: 160 KB per workgroup), else the four-wave form (BM x 64,
  // two workgroups per CU).  HSSK_DGEMM_V1=1 forces the latter (A/B runs).
  static const int cus = hssk_rt::cu_count();
  const bool force_v1 = [] { const char* e = std::getenv("HSSK_DGEMM_V1"); return e && std::atoi(e) != 0; }();
  const bool v2 = aligned && (gen || !force_v1) && k > 0 && n >= BN2 && sketch_prepare(BM);
  if (gen && !v2) {
    // a generated operand outside the eight-wave form's reach (ragged k, odd sample counts, narrow outputs): blocks of
    // columns are written out and multiplied as stored operands
    const long long nb_max = std::max<long long>(64, std::min<long long>(1024, (long long)(size_t(1) << 28) / std::max<long long>(k, 1)));
    const long long ldg = k + (k & 1);
    for (long long c0 = 0; c0 < n; c0 += nb_max) {
      const long long nb = std::min(nb_max, n - c0);
      double* G = ctx->gen_block(sizeof(double) * (size_t)ldg * nb);
      if (hssk_gen_fill(ctx, gen, G, k, nb, ldg, 0, jg0 + c0, transB)) throw std::runtime_error(hssk_last_error());
      dgemm_impl(ctx, 0, m, nb, k, alpha, A, lda, G, ldg, nullptr, 0, beta, C + c0 * ldc, ldc);
    }
    return;
  }
  const int BNt = v2 ? BN2 : BN;
  // The ragged last columns (n not a multiple of the tile width) of the eight-wave form are ONE MORE TILE of its grid, padded: the
  // kernel reads the last valid columns again instead of columns beyond n, writes the tile's partials (the scratch has the
  // room) and the reduce pass folds only the valid ones.  They used to be a launch of their own through the masked
  // four-wave kernel, split 256 ways along K to fill the chip: 1.0 ms for the 32 last columns of N = 1e5 (1.2 TFLOP/s), twice
  // per round, against 0.07 ms for one more tile.  (An odd n keeps the masked launch: the pair that straddles n would be read.)
  static const bool no_fold = [] { const char* e = std::getenv("HSSK_DGEMM_NO_FOLD"); return e && std::atoi(e) != 0; }();
  const bool fold_edge = v2 && !no_fold && n % BNt != 0 && n % 2 == 0;
  const unsigned gn_full = aligned ? (unsigned)(n / BNt) + (fold_edge ? 1u : 0u) : 0u;
  const long long edge_col0 = std::min<long long>(n, (long long)gn_full * BNt);   // the masked kernel starts here (64-column tiles)
  const unsigned gn_edge = (unsigned)((n - edge_col0 + BN - 1) / BN);
  // Work decomposition.  The CUs hold `slots` workgroups (two four-wave ones per CU, or one eight-wave one); a grid that is
  // not a multiple of that ends in a partly filled round.  The full tiles are therefore cut into a MAIN group whose grid
  // (tiles x K-split s) fills r whole rounds exactly, and a short TAIL group (the remaining < slots / s tiles) with a
  // deeper K-split that fills one last round of short workgroups; the ragged edge keeps its own masked launch.  Every
  // group writes K-partials that one deterministic reduce pass per group folds into C.
  const long long slots = v2 ? cus : 2LL * cus;
  // (cols: columns of the group's tiles, the stride of its partials; vcols: those of them that exist)
  struct Group { long long col0 = 0, ntiles = 0; int split = 1; long long kchunk = BK; int nz = 0; double* P = nullptr; long long cols = 0, vcols = 0; };
  auto chunk_of = [&](int split) {
    long long c = ((ksteps + split - 1) / split) * BK;
    return c > 0 ? c : (long long)BK;
  };
  auto max_split = [&]() { return (int)std::max<long long>(1, std::min<long long>(256, ksteps / 24)); };
  // split of a group that should fill (at most) one round
  auto one_round_split = [&](long long tiles) {
    if (tiles <= 0 || k <= 0) return 1;
    return (int)std::max<long long>(1, std::min<long long>(max_split(), slots / tiles));
  };
  Group gmain, gtail, gedge;
  if (gn_full) {
    const long long T = (long long)gm * gn_full;   // gm == 1 on the sketch path (m = d <= 192)
    // cost model (units: one k-step of one workgroup): rounds x (steps per chunk + epilogue) + reduce traffic.
    // The reduce term is ABSOLUTE (per K-chunk and tile: the partials written and folded), calibrated on N = 1e5 (7 chunks x
    // 1536 tiles of 64 columns: 1.06 GB of partials, 0.21 ms = 72 units).  It used to be relative to the number of tiles,
    // which made splits look three to twenty times too expensive on the narrow outputs of a sharded sketch: 12 500 columns
    // per rank of 8 took 146 tiles x 7 + a tail round (90 % of the slots busy, 16.4 ms) instead of 192 x 8 = three full rounds.
    const double epi = 3.0, red = 0.0067 * (BNt / BN);
    double best = 1e300;
    int best_s = 1;
    long long best_main = T;
    // K-chunks of at most ~1024 stages: the workgroups running together on an XCD then stay within a window of
    // the shared A panel (R^T, 24 KB per stage) that its 4 MB L2 can hold even though their speeds differ by
    // +-20 % (two workgroups share a CU's MFMA pipes unevenly).  Measured at N = 1e5 (TCC_EA0_RDREQ_DRAM_32B):
    // memory-side reads per launch 207 GB with one chunk, 123 GB with 6, 96 GB with 12; algorithmic 80 GB.
    const int sp_min = (int)std::min<long long>(std::min(max_split(), 64), (ksteps + 1023) / 1024);
    for (int sp = std::max(1, sp_min); sp <= std::min(max_split(), 64); sp++) {
      const long long r = (T * sp) / slots;                       // whole rounds
      long long tm = r > 0 ? std::min<long long>(T, (r * slots) / sp) : 0;
      if (gm > 1) tm = T;                                          // tall outputs: no tile regrouping
      const long long tt = T - tm;
      double cost = 0.;
      if (tm) cost += (double)((tm * sp + slots - 1) / slots) * ((double)(ksteps + sp - 1) / sp + epi) + red * sp * (double)tm;
      if (tt) {
        const int st = one_round_split(tt);
        cost += (double)((tt * st + slots - 1) / slots) * ((double)(ksteps + st - 1) / st + epi) + red * st * (double)tt + 2.0;
      }
      if (cost < best - 1e-9) { best = cost; best_s = sp; best_main = tm; }
    }
    if (k <= 0) { best_s = 1; best_main = T; }
    if (const char* e = std::getenv("HSSK_DGEMM_SPLIT")) {   // tuning override: K-split of the main group
      const int sp = std::max(1, std::min(max_split(), std::atoi(e)));
      const long long r = (T * sp) / slots;
      best_s = sp;
      best_main = (gm > 1 || r == 0) ? T : std::min<long long>(T, (r * slots) / sp);
    }
    gmain.col0 = 0; gmain.ntiles = gm > 1 ? gn_full : best_main; gmain.split = best_s; gmain.cols = gmain.ntiles * BNt;
    gtail.col0 = gmain.cols; gtail.ntiles = gn_full - gmain.ntiles; gtail.split = one_round_split(gtail.ntiles); gtail.cols = gtail.ntiles * BNt;
  }
  gedge.col0 = edge_col0; gedge.ntiles = gn_edge; gedge.split = one_round_split((long long)gm * gn_edge); gedge.cols = n - edge_col0;
  const long long ldp = m;
  size_t ptot = 0;
  for (Group* g : {&gmain, &gtail, &gedge}) {
    if (!g->ntiles) continue;
    g->vcols = std::min(g->cols, n - g->col0);
    g->kchunk = chunk_of(g->split);
    g->nz = (int)std::max<long long>(1, (k + g->kchunk - 1) / g->kchunk);
    ptot += (size_t)ldp * g->cols * g->nz;
  }
  const size_t ntrace = gmain.ntiles ? (size_t)gmain.ntiles * gm * gmain.nz : 0;
  double* P = ctx->scratch(sizeof(double) * (ptot + 4 + 4 * ntrace));
  long long* clk = (long long*)(P + ptot);  // clock probe of workgroup 0 of the main launch
  {
    double* q = P;
    for (Group* g : {&gmain, &gtail, &gedge}) {
      if (!g->ntiles) continue;
      g->P = q;
      q += (size_t)ldp * g->cols * g->nz;
    }
  }
  // partials of a group are addressed by absolute column: shift its base by the group's first column
  auto shifted = [&](const Group& g) { return g.P - g.col0 * ldp; };
  // the timed launch (hssk_last_dgemm_ms / _flops): the main group, or whatever carries the bulk
  const Group* timed = gmain.ntiles ? &gmain : (gtail.ntiles ? &gtail : &gedge);
  auto bracket = [&](const Group* g, auto&& launch) {
    if (!g->ntiles) return;
    if (g == timed) hssk_rt::event_record(ctx->ev0, ctx->stream);
    launch();
    if (g == timed) hssk_rt::event_record(ctx->ev1, ctx->stream);
  };
  auto grid_of = [&](const Group& g) { return dim3((unsigned)g.ntiles, gm, (unsigned)g.nz); };
  if (gen && gedge.ntiles) {
    // the ragged rest of a generated operand (< 128 columns) is written out and takes the masked kernel as a stored block
    const long long ldg = k + (k & 1);
    double* G = ctx->gen_block(sizeof(double) * (size_t)ldg * gedge.cols);
    if (hssk_gen_fill(ctx, gen, G, k, gedge.cols, ldg, 0, jg0 + gedge.col0, transB)) throw std::runtime_error(hssk_last_error());
    const double* Gs = G - gedge.col0 * ldg;   // (the kernel addresses columns by their absolute index)
    bracket(&gedge, [&] { launch_bm<false, 0>(BM, ctx, 0, grid_of(gedge), m, n, k, A, lda, Gs, ldg, shifted(gedge), ldp, ldp * gedge.cols, gedge.kchunk, (int)(gedge.col0 / BN), nullptr); });
  } else
  bracket(&gedge, [&] { launch_bm<false, 0>(BM, ctx, transB, grid_of(gedge), m, n, k, A, lda, B, ldb, shifted(gedge), ldp, ldp * gedge.cols, gedge.kchunk, (int)(gedge.col0 / BN), nullptr); });
  if (v2) {
    bracket(&gtail, [&] { launch_sketch<1>(BM, ctx, transB, grid_of(gtail), n, k, A, lda, B, ldb, shifted(gtail), ldp, ldp * gtail.cols, gtail.kchunk, (int)(gtail.col0 / BN2), nullptr, gen, jg0); });
    bracket(&gmain, [&] { launch_sketch<0>(BM, ctx, transB, grid_of(gmain), n, k, A, lda, B, ldb, shifted(gmain), ldp, ldp * gmain.cols, gmain.kchunk, (int)(gmain.col0 / BN2), clk, gen, jg0); });
  } else {
    bracket(&gtail, [&] { launch_bm<true, 1>(BM, ctx, transB, grid_of(gtail), m, n, k, A, lda, B, ldb, shifted(gtail), ldp, ldp * gtail.cols, gtail.kchunk, (int)(gtail.col0 / BN), nullptr); });
    bracket(&gmain, [&] { launch_bm<true, 0>(BM, ctx, transB, grid_of(gmain), m, n, k, A, lda, B, ldb, shifted(gmain), ldp, ldp * gmain.cols, gmain.kchunk, (int)(gmain.col0 / BN), clk); });
  }
  ctx->d_clk = gmain.ntiles ? clk : nullptr;
  ctx->dgemm_trace_wgs = (long long)ntrace;
  ctx->dgemm_timed = true;
  ctx->dgemm_timed_flops = 2.0 * (double)m * (double)timed->vcols * (double)k;   // (algorithmic: the columns that exist)
  for (const Group* g : {&gmain, &gtail, &gedge}) {
    if (!g->ntiles) continue;
    long long total = (long long)m * g->vcols;
    unsigned rb = (unsigned)std::min<long long>((total + 255) / 256, 4096);
    if (g->nz >= 32 && total <= 65536)   // too few elements to hide the latency of a serial walk over the partials
      HSSK_LAUNCH(dgemm_reduce_wide_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, ctx->stream, m, g->vcols, (const double*)g->P, ldp, ldp * g->cols, g->nz, alpha, beta, C + g->col0 * ldc, ldc);
    else
      HSSK_LAUNCH(dgemm_reduce_kernel, dim3(rb), dim3(256), 0, ctx->stream, m, g->vcols, (const double*)g->P, ldp, ldp * g->cols, g->nz, alpha, beta, C + g->col0 * ldc, ldc);
  }
  hssk_rt::check_launch();
}

extern "C" int hssk_dgemm(hssk_ctx* ctx, int transB, int m, long long n, long long k, double alpha,
                          const double* A, long long lda, const double* B, long long ldb, double beta,
                          double* C, long long ldc) {
  HSSK_API_BEGIN
  dgemm_impl(ctx, transB, m, n, k, alpha, A, lda, B, ldb, nullptr, 0, beta, C, ldc);
  HSSK_API_END
}

extern "C" int hssk_sketch_gen(hssk_ctx* ctx, const hssk_gen* g, int transG, int m, long long n, long long k, long long j0,
                               double alpha, const double* A, long long lda, double beta, double* C, long long ldc) {
  HSSK_API_BEGIN
  if (!g || (g->kind != HSSK_GEN_TOEPLITZ && g->kind != HSSK_GEN_TOEPLITZ_UPPER)) throw std::invalid_argument("hssk_sketch_gen: unknown generator kind");
  if (k > 0x7fffffffLL || j0 + n > 0x7fffffffLL) throw std::invalid_argument("hssk_sketch_gen: indices beyond 2^31");
  dgemm_impl(ctx, transG, m, n, k, alpha, A, lda, nullptr, 0, g, j0, beta, C, ldc);
  HSSK_API_END
}
