// nf_internal.h -- shared device helpers for libnormflow_hip (gfx950 only).
// Wave = 64 lanes, 256-thread workgroups (4 waves, one per SIMD of a CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/normflow_hip.h"

namespace nf {

constexpr int kBlock = 256;   // threads per workgroup
constexpr int kWave = 64;     // CDNA wavefront
constexpr int kMaxBlocksX = 65535 * 16;

void set_error(const char *fmt, ...);
int check_launch(const char *what);
int option(int which);      // nf_set_option / nf_get_option (nf_common.hip)

#define NF_REQUIRE(cond, ...)                \
  do {                                       \
    if (!(cond)) {                           \
      ::nf::set_error(__VA_ARGS__);          \
      return NF_EINVAL;                      \
    }                                        \
  } while (0)

// ---------------------------------------------------------------- scalar math
// f32 uses the native transcendental instructions (v_exp_f32 = 2^x, v_log_f32 =
// log2 x, both ~1 ulp); softplus(beta = ln 2) of the reference IS log2(1 + 2^x),
// so it maps onto them with no base change.  Divisions are IEEE (no fast-math):
// the kernels are HBM-bound and the 1e-5 parity budget is better spent elsewhere.
template <typename T> struct Num;

template <> struct Num<float> {
  static constexpr float kLog2e = 1.4426950408889634f;
  static constexpr float kLn2 = 0.6931471805599453f;
  static __device__ __forceinline__ float exp2(float x) { return __builtin_amdgcn_exp2f(x); }
  static __device__ __forceinline__ float log2(float x) { return __builtin_amdgcn_logf(x); }
  static __device__ __forceinline__ float sqrt(float x) { return __builtin_sqrtf(x); }
  static __device__ __forceinline__ float abs(float x) { return __builtin_fabsf(x); }
  static __device__ __forceinline__ float max(float a, float b) { return __builtin_fmaxf(a, b); }
  static __device__ __forceinline__ float tiny_log1p_cut() { return 2.44140625e-4f; }  // 2^-12
  static __device__ __forceinline__ float softplus_cut() { return 28.853900817779268f; }  // 20/ln2
};

template <> struct Num<double> {
  static constexpr double kLog2e = 1.4426950408889634;
  static constexpr double kLn2 = 0.6931471805599453;
  static __device__ __forceinline__ double exp2(double x) { return ::exp2(x); }
  static __device__ __forceinline__ double log2(double x) { return ::log2(x); }
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  static __device__ __forceinline__ double abs(double x) { return ::fabs(x); }
  static __device__ __forceinline__ double max(double a, double b) { return ::fmax(a, b); }
  static __device__ __forceinline__ double tiny_log1p_cut() { return 1.52587890625e-5; }  // 2^-16
  static __device__ __forceinline__ double softplus_cut() { return 28.853900817779268; }
};

// e^x and ln x through the base-2 units
template <typename T> __device__ __forceinline__ T nf_exp(T x) { return Num<T>::exp2(x * Num<T>::kLog2e); }
template <typename T> __device__ __forceinline__ T nf_log(T x) { return Num<T>::log2(x) * Num<T>::kLn2; }

// softplus with beta = ln2: log2(1 + 2^c); equals c once ln2*c > 20 (torch's
// threshold, which the reference inherits: couplings_.py:172).  For very negative
// c the series of log1p keeps the relative accuracy that log2(1 + u) loses.
// `dsig` returns d/dc = u / (1 + u).
template <typename T> __device__ __forceinline__ T softplus2(T c, T *dsig = nullptr) {
  if (c > Num<T>::softplus_cut()) {
    if (dsig) *dsig = T(1);
    return c;
  }
  const T u = Num<T>::exp2(c);
  if (dsig) *dsig = u / (T(1) + u);
  if (u < Num<T>::tiny_log1p_cut()) return u * (T(1) - T(0.5) * u) * Num<T>::kLog2e;
  return Num<T>::log2(T(1) + u);
}

// ---------------------------------------------------------------- reductions
// Per-sample log-det sums are accumulated in double from the wave upward; the
// per-thread partial (a handful of sites) stays in T.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;  // valid in lane 0
}

// Sum over the workgroup (blockDim.x <= kBlock); result valid in thread 0.  `smem` needs
// kBlock/kWave doubles.
__device__ __forceinline__ double block_sum(double v, double *smem) {
  v = wave_sum(v);
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  if (lane == 0) smem[w] = v;
  __syncthreads();
  double r = 0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + kWave - 1) / kWave;
    for (int i = 0; i < nw; ++i) r += smem[i];
  }
  return r;
}

// Stage 2 of the per-sample reduction: logj[b] = log0[b] + sum_i partial[b, i].
template <typename T>
int launch_finalize(const double *partial, int64_t n_part, const void *log0, void *logj, int64_t B,
                    hipStream_t stream);

// How a (B, units) problem is cut into workgroups: every workgroup owns `per_block`
// consecutive units of ONE sample, so its partial log-det needs no atomics.
struct Tiling {
  int64_t units;      // work units per sample
  int iters;          // units per thread
  int64_t blocks_x;   // workgroups per sample
};
inline Tiling make_tiling(int64_t units, int64_t B, int block = kBlock) {
  Tiling t;
  t.units = units;
  // aim for >= ~4 waves of workgroups over 256 CUs before growing the per-thread loop
  int iters = 1;
  while (iters < 8 && (units / (int64_t(block) * iters * 2)) * B >= 8192) iters *= 2;
  t.iters = iters;
  t.blocks_x = (units + int64_t(block) * iters - 1) / (int64_t(block) * iters);
  return t;
}

}  // namespace nf
