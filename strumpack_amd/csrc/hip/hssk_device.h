// Device-side primitives for the gfx950 (CDNA4, wave64) kernels of the HSS engine.
// Everything a kernel needs beyond plain C++ goes through this header: MFMA, wave shuffles,
// launch syntax.  (tests/emu/ carries a same-named header that maps these onto a CPU fiber
// emulator so index arithmetic and host orchestration can be unit-tested without a GPU; the
// product is only ever built from THIS file with hipcc --offload-arch=gfx950.)
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <vector>

namespace hssk_rec {
extern thread_local std::vector<std::function<void()>>* sink;   // non-null while a plan is being recorded
}

#define HSSK_WAVE 64

typedef double hssk_d4 __attribute__((ext_vector_type(4)));
typedef double hssk_d2 __attribute__((ext_vector_type(2)));

// D(16x16) += A(16x4) * B(4x16), FP64 matrix core (v_mfma_f64_16x16x4_f64).
// lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
// lane l receives, in c[r], D[row = (l >> 4) + 4 r][col = l & 15].
__device__ __forceinline__ hssk_d4 hssk_mfma_f64_16x16x4(double a, double b, hssk_d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// D(32x32) += A(32x2) * B(2x32), FP32 matrix core (v_mfma_f32_32x32x2_f32).
// lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
// lane l receives, in c[r], D[row = 8 (r / 4) + 4 (l >> 5) + r % 4][col = l & 31].
typedef float hssk_f16v __attribute__((ext_vector_type(16)));
typedef float hssk_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ hssk_f16v hssk_mfma_f32_32x32x2(float a, float b, hssk_f16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ double hssk_shfl_xor(double v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ int hssk_shfl_xor(int v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ double hssk_shfl(double v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int hssk_shfl(int v, int src) { return __shfl(v, src, 64); }

// 64-lane sum on the DPP network (no LDS crossbar): quad swaps, half-row / row mirrors, then the
// GFX9 row broadcasts; lane 63 ends up with the total, which is read back through an SGPR.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double hssk_dpp_mov0(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double hssk_wave_sum(double v) {
  v += hssk_dpp_mov0<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += hssk_dpp_mov0<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += hssk_dpp_mov0<0x141, 0xF>(v);  // row_half_mirror
  v += hssk_dpp_mov0<0x140, 0xF>(v);  // row_mirror      -> every lane holds its 16-lane row sum
  v += hssk_dpp_mov0<0x142, 0xA>(v);  // row_bcast15 into rows 1, 3
  v += hssk_dpp_mov0<0x143, 0xC>(v);  // row_bcast31 into rows 2, 3 -> lane 63 holds the total
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
// sum over each 16-lane DPP row (lanes 16 q .. 16 q + 15); every lane of a row ends up with its row's total.
// The register QR / ID kernels keep one matrix column per row of lanes, so a column dot product is these four DPP steps.
__device__ __forceinline__ double hssk_row_sum(double v) {
  v += hssk_dpp_mov0<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += hssk_dpp_mov0<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += hssk_dpp_mov0<0x141, 0xF>(v);  // row_half_mirror
  v += hssk_dpp_mov0<0x140, 0xF>(v);  // row_mirror
  return v;
}
// first arg max over each 16-lane DPP row: on return every lane of a row holds the largest v of the row and the smallest idx
// among the lanes that held it (the exchanges are symmetric, so all lanes take the same decisions)
template <int CTRL>
__device__ __forceinline__ void hssk_argmax_step(double& v, int& idx) {
  const double ov = hssk_dpp_mov0<CTRL, 0xF>(v);
  const int oi = __builtin_amdgcn_update_dpp(0, idx, CTRL, 0xF, 0xF, false);
  if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
}
__device__ __forceinline__ void hssk_row_argmax(double& v, int& idx) {
  hssk_argmax_step<0xB1>(v, idx);    // quad_perm [1,0,3,2]
  hssk_argmax_step<0x4E>(v, idx);    // quad_perm [2,3,0,1]
  hssk_argmax_step<0x141>(v, idx);   // row_half_mirror
  hssk_argmax_step<0x140>(v, idx);   // row_mirror
}
// the same for N independent values, stage by stage: N dependent chains in flight instead of one after the other
template <int N>
__device__ __forceinline__ void hssk_row_sum_n(double (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; i++) v[i] += hssk_dpp_mov0<0xB1, 0xF>(v[i]);
#pragma unroll
  for (int i = 0; i < N; i++) v[i] += hssk_dpp_mov0<0x4E, 0xF>(v[i]);
#pragma unroll
  for (int i = 0; i < N; i++) v[i] += hssk_dpp_mov0<0x141, 0xF>(v[i]);
#pragma unroll
  for (int i = 0; i < N; i++) v[i] += hssk_dpp_mov0<0x140, 0xF>(v[i]);
}
// value of `v` in lane `src` (src must be wave-uniform): v_readlane, no LDS crossbar
__device__ __forceinline__ double hssk_bcast_lane(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int hssk_bcast_lane_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
// first arg max over the 64 lanes of the wave, same rule; every lane ends up with the result.  The rows on the DPP network,
// the four row results through SGPRs (v_readlane): no LDS crossbar.  (The pivot searches of the LU / QRCP kernels ran six
// stages of three ds_bpermute each per elimination step -- eighteen dependent round trips through the LDS pipeline.)
__device__ __forceinline__ void hssk_wave_argmax(double& v, int& idx) {
  hssk_row_argmax(v, idx);
  double bv = hssk_bcast_lane(v, 0);
  int bi = hssk_bcast_lane_i(idx, 0);
#pragma unroll
  for (int r = 1; r < 4; r++) {
    const double ov = hssk_bcast_lane(v, 16 * r);
    const int oi = hssk_bcast_lane_i(idx, 16 * r);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  v = bv; idx = bi;
}
// value of `v` in lane Q of the caller's quad (lanes 4 g .. 4 g + 3): one DPP quad_perm move per half
template <int Q>
__device__ __forceinline__ double hssk_quad_bcast(double v) { return hssk_dpp_mov0<Q | (Q << 2) | (Q << 4) | (Q << 6), 0xF>(v); }
// value of `v` in lane Q (0 / 1) of the caller's pair of lanes (2 g, 2 g + 1)
template <int Q>
__device__ __forceinline__ double hssk_pair_bcast(double v) { return hssk_dpp_mov0<Q | (Q << 2) | ((2 + Q) << 4) | ((2 + Q) << 6), 0xF>(v); }
// non-zero if `pred` holds in any lane of the wave
__device__ __forceinline__ int hssk_any(int pred) { return __any(pred); }
// bit l = lane l's predicate (all 64 lanes of the wave take part)
__device__ __forceinline__ unsigned long long hssk_ballot(int pred) { return __ballot(pred); }
__device__ __forceinline__ double hssk_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, hssk_shfl_xor(v, o));
  return v;
}

// shader-clock and constant-rate (100 MHz) counters, for the peak probe
__device__ __forceinline__ long long hssk_clock() { return (long long)__builtin_readcyclecounter(); }
__device__ __forceinline__ long long hssk_wallclock() { return (long long)__builtin_amdgcn_s_memrealtime(); }
// (XCC_ID << 32) | HW_ID: which XCD / SE / CU / SIMD this wave runs on (profiling aid)
__device__ __forceinline__ long long hssk_hwid() {
  return ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
}


// ---- cross-workgroup hand-off inside one launch (single-launch tree sweeps, kernels/hssk_sweep.hip) ---------------
// The vectors one workgroup hands to another (a few hundred bytes) and the flag that announces them are written and
// read with agent-scope RELAXED atomics: on gfx950 these are sc1 accesses served at the device coherence point, so no
// cache-wide maintenance is needed.  (Agent-scope release / acquire FENCES would write back / invalidate the whole L2
// of the XCD -- measured ~100 us per tree level.)  Ordering: the producer drains its stores (s_waitcnt 0) before the
// workgroup barrier that precedes the flag store; the consumer's loads are issued after the barrier that follows the
// poll.  s_sleep keeps the polling wave off the issue ports.
__device__ __forceinline__ int hssk_flag_load(const int* f) { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void hssk_flag_store(int* f, int v) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int hssk_flag_sub(int* f, int v) { return __hip_atomic_fetch_sub(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void hssk_flag_raise(int* f) { __hip_atomic_store(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }   // error word in pinned host memory
__device__ __forceinline__ double hssk_cload(const double* p, size_t off) {
  return __longlong_as_double(__hip_atomic_load((const long long*)(p + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void hssk_cstore(double* p, size_t off, double v) {
  __hip_atomic_store((long long*)(p + off), __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// bit patterns of doubles (sentinel test of the sweep hand-off)
__device__ __forceinline__ unsigned long long hssk_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ double hssk_from_bits(unsigned long long b) { return __longlong_as_double((long long)b); }
__device__ __forceinline__ unsigned hssk_fbits(float v) { return __float_as_uint(v); }
__device__ __forceinline__ float hssk_from_fbits(unsigned b) { return __uint_as_float(b); }
// the lanes of a wave exchange data through the LDS without a workgroup barrier: LDS operations of a wave execute in order;
// this keeps the compiler from moving them across the point (the emulator's lanes are fibers: they meet here)
__device__ __forceinline__ void hssk_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
// instruction-scheduling fence: nothing is moved across it
__device__ __forceinline__ void hssk_sched_barrier() { __builtin_amdgcn_sched_barrier(0); }
// request to the instruction scheduler: the next `n` instructions of class MASK of this scheduling region go here
// (__builtin_amdgcn_sched_group_barrier; classes: matrix core, vector ALU incl. transcendental, LDS write)
#define HSSK_SG_MFMA 0x008
#define HSSK_SG_VALU 0x002
#define HSSK_SG_DSWRITE 0x200
template <int MASK, int N> __device__ __forceinline__ void hssk_sched_group() { __builtin_amdgcn_sched_group_barrier(MASK, N, 0); }
__device__ __forceinline__ void hssk_drain_stores() { __builtin_amdgcn_s_waitcnt(0); }
__device__ __forceinline__ void hssk_pause() { __builtin_amdgcn_s_sleep(2); }

// occupancy the register allocator should plan for (workgroups of N wave64 per SIMD): the full register budget of that
// occupancy is then available to the kernel's register tile
#define HSSK_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#define HSSK_SHARED __shared__ __attribute__((aligned(16)))

// Global-memory accessors for pointers that arrive through a descriptor in memory: the compiler cannot infer
// their address space and would emit FLAT loads, which also count on lgkmcnt and so serialise against every
// LDS wait of a pipelined kernel.  These cast to address space 1 (global_load / global_store, vmcnt only).
#define HSSK_GLOBAL_AS __attribute__((address_space(1)))
__device__ __forceinline__ double hssk_gload(const double* p, size_t off) { return ((const double HSSK_GLOBAL_AS*)p)[off]; }
__device__ __forceinline__ hssk_d2 hssk_gload2(const double* p, size_t off) {
  return *(const hssk_d2 HSSK_GLOBAL_AS*)((const double HSSK_GLOBAL_AS*)p + off);
}
// the same from an address that is only 8-byte aligned (one 16-byte load all the same: global memory takes unaligned accesses)
__device__ __forceinline__ hssk_d2 hssk_gload2u(const double* p, size_t off) { return hssk_gload2(p, off); }
__device__ __forceinline__ void hssk_gstore(double* p, size_t off, double v) { ((double HSSK_GLOBAL_AS*)p)[off] = v; }
// ---- asynchronous global -> LDS copies (global_load_lds_dwordx4, "LDS DMA") -------------------------------------
// Every lane names 16 bytes of global memory; lane l's piece lands at lds_base + 16 l (lds_base wave-uniform: it
// travels in M0), without passing through registers.  The copy counts on vmcnt and nothing else orders it against
// LDS reads: the issuing wave waits (hssk_wait_glds<N>: at most N of ITS copies still in flight), then a workgroup
// barrier, then the reads.  hssk_wg_barrier is the bare s_barrier (__syncthreads would drain every counter).
// (Written as an asm statement: through __builtin_amdgcn_global_load_lds the compiler's wait-count pass books the copy as
// a FLAT access that may touch the LDS, after which every wait it places in front of an LDS read's consumer is a full
// lgkmcnt(0) instead of the counted one -- the pipelined fragment reads of the sketch kernel then stall on the newest
// read instead of the oldest.  M0 is the compiler's: saved and restored inside the statement.)
__device__ __forceinline__ void hssk_glds16(const double* gsrc, double* lds_base) {
  unsigned keep;
  const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_base;
  __asm__ volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
template <int N> __device__ __forceinline__ void hssk_wait_glds() {   // s_waitcnt vmcnt(N) lgkmcnt(0)
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0070);
}
__device__ __forceinline__ void hssk_wg_barrier() { __builtin_amdgcn_s_barrier(); }
// an index the compiler may not reason about (keeps two LDS reads from being fused into one half-rate ds_read2_b64)
__device__ __forceinline__ int hssk_opaque(int v) { __asm__ volatile("" : "+v"(v)); return v; }
// LDS accumulate without a return value (ds_add_f64): lanes / waves of a workgroup summing into shared slots
__device__ __forceinline__ void hssk_lds_add(double* p, double v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// bits set in an LDS word by many lanes at once
__device__ __forceinline__ void hssk_lds_or(unsigned* p, unsigned v) {
  (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// a slot counter in the LDS: returns the value before the increment
__device__ __forceinline__ int hssk_lds_inc(int* p) { return __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// a value known to be the same in every lane of the wave, moved to a scalar register (loads indexed by it become s_load)
// device-wide counter (statistics)
// k + t * t with the product rounded before the addition (what a host compiler without fused multiply-add computes)
__device__ __forceinline__ double hssk_sq_acc_rn(double k, double t) { return __dadd_rn(k, __dmul_rn(t, t)); }
__device__ __forceinline__ int hssk_gadd_i(int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void hssk_gadd_ll(long long* p, long long v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int hssk_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// compile-time only: memory operations are not moved across this point
#define HSSK_COMPILER_FENCE() __asm__ volatile("" ::: "memory")
#define HSSK_DYN_SHARED(type, name) extern __shared__ __attribute__((aligned(16))) type name[]

// kernel<<<grid, block, shmem, stream>>>(args...).  While a sweep plan is being recorded on this thread
// (hssk_plan_begin / _end, hssk_internal.h) the launch is also remembered, arguments by value, for replay.
#define HSSK_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                  \
  do {                                                                                                        \
    hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)(stream), __VA_ARGS__);                       \
    if (hssk_rec::sink)                                                                                       \
      hssk_rec::sink->push_back([=]() { hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)(stream), __VA_ARGS__); }); \
  } while (0)
