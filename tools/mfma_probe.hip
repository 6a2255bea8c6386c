// mfma_probe.hip -- what rate does v_mfma_f32_16x16x4_f32 reach on this chip, alone and with the
// operand traffic pattern of the convolution kernel?  (diagnostic; build: hipcc -O3 --offload-arch=gfx950)
//   mode 0: MFMAs only (6 independent accumulators, operands fixed in registers)
//   mode 1: + A fragments from LDS (6 ds_read_b32 per 18 MFMAs, double buffered)
//   mode 2: + B fragments from global memory (9 loads of 256 B per 18 MFMAs, 124 KB table, double buffered)
//   mode 3: A from LDS and B from global
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ w, float *out, int iters, int nrows) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = float(i & 7) * 0.125f;
  __syncthreads();
  f32x4 acc[2][3];
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
  float a0[3][2], b0[3][3], a1[3][2], b1[3][3];
  for (int j = 0; j < 3; ++j) { for (int m = 0; m < 2; ++m) a0[j][m] = a1[j][m] = 1.f + lane; for (int n = 0; n < 3; ++n) b0[j][n] = b1[j][n] = 0.5f; }
  const float *wl = w + lane;
  int row = 0, off = lane;
  auto request = [&](float (&a)[3][2], float (&b)[3][3]) {
    if (MODE & 4) {        // one 16-byte load per tap: the lane's three column-tile values are adjacent (+1 pad)
      const f32x4 *wt = reinterpret_cast<const f32x4 *>(w) + lane + row * (3 * 2 * 64);
      for (int j = 0; j < 3; ++j) { const f32x4 v = wt[j * 2 * 64]; b[j][0] = v[0]; b[j][1] = v[1]; b[j][2] = v[2]; }
    } else if (MODE & 2) {
      const float *wt = wl + row * (3 * 2 * 192);
      for (int j = 0; j < 3; ++j) for (int n = 0; n < 3; ++n) b[j][n] = wt[j * 2 * 192 + n * 64];
    }
    if (MODE & 1) {
      for (int m = 0; m < 2; ++m) for (int j = 0; j < 3; ++j) a[j][m] = lds[off + m * 1024 + j];
    }
  };
  auto multiply = [&](const float (&a)[3][2], const float (&b)[3][3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][m], b[j][n], acc[m][n], 0, 0, 0);
  };
  auto next = [&]() { row = row + 1 < nrows ? row + 1 : 0; off = (off + 34) & 2047; };
  request(a0, b0);
  for (int it = 0; it < iters; ++it) {
    next(); request(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    multiply(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    next(); request(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    multiply(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0;
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 3; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE> double run(const float *w, float *out, int blocks_per_cu, int lds_bytes, int iters, int nrows = 27) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), lds_bytes, 0, w, out, iters, nrows);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), lds_bytes, 0, w, out, iters, nrows);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = double(grid) * 4 * iters * 36.0 * 2048.0;
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  float *w, *out;
  hipMalloc(&w, 27 * 3 * 2 * 256 * 4 + 4096); hipMemset(w, 0, 27 * 3 * 2 * 256 * 4 + 4096);
  hipMalloc(&out, 64);
  const int iters = 4000;
  // LDS per block picks the residency: 150 KB -> 1 block/CU (1 wave/SIMD), 75 KB -> 2, 50 KB -> 3, 36 KB -> 4
  const int ldsz[4] = {150 * 1024, 75 * 1024, 50 * 1024, 36 * 1024};
  for (int r = 0; r < 4; ++r) {
    printf("waves/SIMD %d:  mfma only %.1f  +LDS A %.1f  +global B %.1f  both %.1f  TFLOP/s\n", r + 1,
           run<0>(w, out, r + 1, ldsz[r], iters), run<1>(w, out, r + 1, ldsz[r], iters),
           run<2>(w, out, r + 1, ldsz[r], iters), run<3>(w, out, r + 1, ldsz[r], iters));
  }
  for (int r = 0; r < 4; ++r) {
    printf("waves/SIMD %d:  global B, 1 row (L1 resident) %.1f   B as 3 x 16-byte loads %.1f   same + LDS A %.1f   same, 1 row %.1f\n", r + 1,
           run<2>(w, out, r + 1, ldsz[r], iters, 1), run<4>(w, out, r + 1, ldsz[r], iters), run<5>(w, out, r + 1, ldsz[r], iters),
           run<5>(w, out, r + 1, ldsz[r], iters, 1));
  }
  return 0;
}
