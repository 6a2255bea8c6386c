// mfma_probe3.hip -- feasibility probe for the next step of the conv kernels: fp32 products as THREE fp16 MFMAs
// (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, f32 accumulate; the fp16 matrix pipe is 16x the fp32 one per product).
// Weight-stationary shape: a wave owns one 16-column tile and keeps its B fragments (hi and lo) for the whole K
// in registers; A fragments (hi, lo) of 8 site tiles come from LDS (channel-last fp16, 8 bytes per lane and tile).
// Reports fp32-EQUIVALENT TFLOP/s (2*M*N*K per product, the three MFMAs counted once).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int NSL, int MTL, bool LDS_A>
__global__ __launch_bounds__(256, 1) void probe(const f16x4 *__restrict__ w, float *out, int iters) {
  extern __shared__ f16x4 lds4[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 256) lds4[i] = f16x4{(_Float16)0.5f, (_Float16)0.25f, (_Float16)1.f, (_Float16)0.125f};
  __syncthreads();
  f16x4 bh[NSL], bl[NSL];
#pragma unroll
  for (int s = 0; s < NSL; ++s) { bh[s] = w[(s * 2) * 64 + lane]; bl[s] = w[(s * 2 + 1) * 64 + lane]; }
  f32x4 acc[MTL];
#pragma unroll
  for (int m = 0; m < MTL; ++m) acc[m] = f32x4{0, 0, 0, 0};
  f16x4 ah[MTL], al[MTL];
#pragma unroll
  for (int m = 0; m < MTL; ++m) { ah[m] = lds4[lane + 64 * m]; al[m] = lds4[lane + 64 * m + 512]; }
  int off = lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      f16x4 nh[MTL], nl[MTL];
      if (LDS_A) {
        off = (off + 37) & 4095;
#pragma unroll
        for (int m = 0; m < MTL; ++m) { nh[m] = lds4[off + 64 * m]; nl[m] = lds4[off + 64 * m + 2048]; }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MTL; ++m) {
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[m], bh[s], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[m], bl[s], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x16f16(al[m], bh[s], acc[m], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (LDS_A) {
#pragma unroll
        for (int m = 0; m < MTL; ++m) { ah[m] = nh[m]; al[m] = nl[m]; }
      }
    }
  }
  float sum = 0;
  for (int m = 0; m < MTL; ++m) for (int r = 0; r < 4; ++r) sum += acc[m][r];
  if (sum == 12345.678f) out[0] = sum;
}

template <int NSL, int MTL, bool LDS_A> double run(const f16x4 *w, float *out, int blocks_per_cu, int lds_bytes, int iters) {
  auto k = probe<NSL, MTL, LDS_A>;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_bytes, 0, w, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_bytes, 0, w, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = double(grid) * 4 * iters * NSL * MTL * (2.0 * 16 * 16 * 16);   // one fp32-equivalent product per (tile, slice)
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  f16x4 *w; float *out;
  hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20);
  hipMalloc(&out, 64);
  const int iters = 6000;
  printf("split-fp16 (3 MFMA 16x16x16 per product), B in registers, fp32-equivalent TFLOP/s (fp32 MFMA peak 157):\n");
  printf("  40 slices, 8 site tiles, A fixed in registers : 1 wg/CU %.0f   2 wg/CU %.0f\n", run<40, 8, false>(w, out, 1, 150 << 10, iters), run<40, 8, false>(w, out, 2, 75 << 10, iters));
  printf("  40 slices, 8 site tiles, A (hi, lo) from LDS   : 1 wg/CU %.0f   2 wg/CU %.0f\n", run<40, 8, true>(w, out, 1, 150 << 10, iters), run<40, 8, true>(w, out, 2, 75 << 10, iters));
  printf("  40 slices, 4 site tiles, A (hi, lo) from LDS   : 1 wg/CU %.0f   2 wg/CU %.0f\n", run<40, 4, true>(w, out, 1, 150 << 10, iters), run<40, 4, true>(w, out, 2, 75 << 10, iters));
  return 0;
}
