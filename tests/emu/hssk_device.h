// TEST INFRASTRUCTURE ONLY.  CPU stand-in for strumpack_amd/csrc/hip/hssk_device.h: lets the
// *same kernel sources* be compiled with g++ and run on a fiber-based SIMT emulator
// (emu_runtime.cpp) so that index arithmetic and the host orchestration can be unit-tested in the
// GPU-less build container.  Never part of the product: libstrumpack_amd.so is built by hipcc for
// gfx950 from csrc/hip/hssk_device.h, and the Python package refuses to load anything else.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <vector>

#define HSSK_WAVE 64
#define HSSK_EMU 1

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define HSSK_WAVES_PER_SIMD(n)

namespace hssk_rec {
extern thread_local std::vector<std::function<void()>>* sink;   // non-null while a plan is being recorded
}
typedef double hssk_d4 __attribute__((vector_size(32)));
typedef double hssk_d2 __attribute__((vector_size(16)));
typedef float hssk_f16v __attribute__((vector_size(64)));
typedef float hssk_f4 __attribute__((vector_size(16)));

namespace emu {
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void* dyn_shared();
void block_barrier();
double wave_xchg(double v, int src_lane);  // every live lane of the wave must call
hssk_d4 mfma_f64_16x16x4(double a, double b, hssk_d4 c);
hssk_f16v mfma_f32_32x32x2(float a, float b, hssk_f16v c);
unsigned long long wave_ballot(int pred);   // every live lane of the wave must call
}  // namespace emu

inline void __syncthreads() { emu::block_barrier(); }

inline hssk_d4 hssk_mfma_f64_16x16x4(double a, double b, hssk_d4 c) {
  return emu::mfma_f64_16x16x4(a, b, c);
}
inline hssk_f16v hssk_mfma_f32_32x32x2(float a, float b, hssk_f16v c) { return emu::mfma_f32_32x32x2(a, b, c); }
inline double hssk_shfl_xor(double v, int mask) {
  return emu::wave_xchg(v, (int)((threadIdx.x & 63) ^ mask));
}
inline int hssk_shfl_xor(int v, int mask) { return (int)hssk_shfl_xor((double)v, mask); }
inline double hssk_shfl(double v, int src) { return emu::wave_xchg(v, src & 63); }
inline int hssk_shfl(int v, int src) { return (int)hssk_shfl((double)v, src); }
inline double hssk_bcast_lane(double v, int src) { return emu::wave_xchg(v, src & 63); }
inline double hssk_wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += hssk_shfl_xor(v, o);
  return v;
}
inline int hssk_bcast_lane_i(int v, int src) { return (int)emu::wave_xchg((double)v, src & 63); }
inline int hssk_any(int pred) {
  double v = pred ? 1. : 0.;
  for (int o = 32; o > 0; o >>= 1) v = std::fmax(v, hssk_shfl_xor(v, o));
  return v > 0.;
}
inline double hssk_row_sum(double v) {
  for (int o = 8; o > 0; o >>= 1) v += hssk_shfl_xor(v, o);
  return v;
}
inline void hssk_row_argmax(double& v, int& idx) {
  for (int o = 8; o > 0; o >>= 1) {
    const double ov = hssk_shfl_xor(v, o);
    const int oi = hssk_shfl_xor(idx, o);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
}
inline void hssk_wave_argmax(double& v, int& idx) {
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = hssk_shfl_xor(v, o);
    const int oi = hssk_shfl_xor(idx, o);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
}
inline unsigned long long hssk_ballot(int pred) { return emu::wave_ballot(pred); }
template <int Q> inline double hssk_quad_bcast(double v) { return emu::wave_xchg(v, (int)((threadIdx.x & 60) | Q)); }
template <int Q> inline double hssk_pair_bcast(double v) { return emu::wave_xchg(v, (int)((threadIdx.x & 62) | Q)); }
template <int N>
inline void hssk_row_sum_n(double (&v)[N]) {
  for (int i = 0; i < N; i++) v[i] = hssk_row_sum(v[i]);
}
using std::min;
using std::max;
inline double hssk_wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = std::fmax(v, hssk_shfl_xor(v, o));
  return v;
}

inline long long hssk_clock() { return 0; }
inline long long hssk_wallclock() { return 0; }
inline long long hssk_hwid() { return 0; }



// cross-workgroup dependency flags (single-launch tree sweeps): workgroups of a launch are taken in index order by the
// emulator's worker threads, so a workgroup that polls a lower-indexed one always finds it running or finished
inline int hssk_flag_load(const int* f) { return __atomic_load_n(f, __ATOMIC_ACQUIRE); }
inline void hssk_flag_store(int* f, int v) { __atomic_store_n(f, v, __ATOMIC_RELEASE); }
inline int hssk_flag_sub(int* f, int v) { return __atomic_fetch_sub(f, v, __ATOMIC_ACQ_REL); }
inline void hssk_flag_raise(int* f) { __atomic_store_n(f, 1, __ATOMIC_RELAXED); }
inline double hssk_cload(const double* p, size_t off) { double v; __atomic_load((const double*)(p + off), &v, __ATOMIC_ACQUIRE); return v; }
inline void hssk_cstore(double* p, size_t off, double v) { __atomic_store(p + off, &v, __ATOMIC_RELEASE); }
inline void hssk_sched_barrier() {}
#define HSSK_SG_MFMA 0x008
#define HSSK_SG_VALU 0x002
#define HSSK_SG_DSWRITE 0x200
template <int MASK, int N> inline void hssk_sched_group() {}
inline unsigned hssk_fbits(float v) { unsigned b; std::memcpy(&b, &v, 4); return b; }
inline float hssk_from_fbits(unsigned b) { float v; std::memcpy(&v, &b, 4); return v; }
inline void hssk_wave_sync() { (void)emu::wave_ballot(1); }
inline unsigned long long hssk_bits(double v) { unsigned long long b; std::memcpy(&b, &v, 8); return b; }
inline double hssk_from_bits(unsigned long long b) { double v; std::memcpy(&v, &b, 8); return v; }
inline void hssk_drain_stores() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
void hssk_pause();   // emu_runtime.cpp: sched_yield

#define HSSK_SHARED alignas(16) static thread_local
inline double hssk_gload(const double* p, size_t off) { return p[off]; }
inline hssk_d2 hssk_gload2(const double* p, size_t off) { return *reinterpret_cast<const hssk_d2*>(p + off); }
inline hssk_d2 hssk_gload2u(const double* p, size_t off) { hssk_d2 v; std::memcpy(&v, p + off, sizeof v); return v; }
inline void hssk_gstore(double* p, size_t off, double v) { p[off] = v; }
// asynchronous global -> LDS copies: immediate on the emulator (lane l's 16 bytes land at lds_base + 16 l)
inline void hssk_glds16(const double* gsrc, double* lds_base) { std::memcpy(lds_base + 2 * (threadIdx.x & 63), gsrc, 16); }
template <int N> inline void hssk_wait_glds() {}
inline void hssk_wg_barrier() { emu::block_barrier(); }
inline int hssk_opaque(int v) { return v; }
inline void hssk_lds_add(double* p, double v) { *p += v; }   // fibers are cooperative: a plain update is atomic
inline void hssk_lds_or(unsigned* p, unsigned v) { *p |= v; }
inline int hssk_lds_inc(int* p) { return (*p)++; }
inline double hssk_sq_acc_rn(double k, double t) { volatile double p = t * t; return k + p; }   // (the product rounded on its own)
inline int hssk_gadd_i(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline void hssk_gadd_ll(long long* p, long long v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline int hssk_uniform(int v) { return v; }
// compile-time only: memory operations are not moved across this point
#define HSSK_COMPILER_FENCE() __asm__ volatile("" ::: "memory")
#define HSSK_DYN_SHARED(type, name) type* name = (type*)emu::dyn_shared()

#define HSSK_LAUNCH(kernel, grid, block, shmem, stream, ...)                                          \
  do {                                                                                                \
    emu::launch(grid, block, shmem, [=]() { kernel(__VA_ARGS__); });                                  \
    if (hssk_rec::sink)                                                                               \
      hssk_rec::sink->push_back([=]() { emu::launch(grid, block, shmem, [=]() { kernel(__VA_ARGS__); }); }); \
  } while (0)
