// mfma_probe4.hip -- as mfma_probe3 but with the gfx950 K=32 instruction (v_mfma_f32_16x16x32_f16): a slice is
// 4 taps x 8 channels, A fragments are 16 bytes (hi) + 16 bytes (lo) per lane and site tile.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NSL, int MTL, bool LDS_A, int NWAVE>
__global__ __launch_bounds__(256, 1) void probe(const f16x8 *__restrict__ w, float *out, int iters) {
  extern __shared__ f16x8 lds8[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8192; i += 256) lds8[i] = f16x8{(_Float16)0.5f, (_Float16)0.25f, (_Float16)1.f, (_Float16)0.125f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)1.f, (_Float16)0.125f};
  __syncthreads();
  if (wave >= NWAVE) return;
  f16x8 bh[NSL], bl[NSL];
#pragma unroll
  for (int s = 0; s < NSL; ++s) { bh[s] = w[(s * 2) * 64 + lane]; bl[s] = w[(s * 2 + 1) * 64 + lane]; }
  f32x4 acc[MTL];
#pragma unroll
  for (int m = 0; m < MTL; ++m) acc[m] = f32x4{0, 0, 0, 0};
  f16x8 ah[MTL], al[MTL];
#pragma unroll
  for (int m = 0; m < MTL; ++m) { ah[m] = lds8[lane + 64 * m]; al[m] = lds8[lane + 64 * m + 512]; }
  int off = lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      f16x8 nh[MTL], nl[MTL];
      if (LDS_A) {
        off = (off + 37) & 2047;
#pragma unroll
        for (int m = 0; m < MTL; ++m) { nh[m] = lds8[off + 64 * m]; nl[m] = lds8[off + 64 * m + 2048]; }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MTL; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[s], acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MTL; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[s], acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MTL; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[s], acc[m], 0, 0, 0);
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

template <int NSL, int MTL, bool LDS_A, int NWAVE> double run(const f16x8 *w, float *out, int iters) {
  auto k = probe<NSL, MTL, LDS_A, NWAVE>;
  const int lds_bytes = 150 << 10;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  const int grid = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_bytes, 0, w, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_bytes, 0, w, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = double(grid) * NWAVE * iters * NSL * MTL * (2.0 * 16 * 16 * 32);
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  f16x8 *w; float *out;
  hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20);
  hipMalloc(&out, 64);
  const int iters = 2000;
  printf("split-fp16 with v_mfma_f32_16x16x32_f16, B in registers, 1 workgroup/CU, fp32-equivalent TFLOP/s:\n");
  printf("  21 slices, 8 tiles, A in registers : 4 waves %.0f   3 waves %.0f\n", run<21, 8, false, 4>(w, out, iters), run<21, 8, false, 3>(w, out, iters));
  printf("  21 slices, 8 tiles, A from LDS     : 4 waves %.0f   3 waves %.0f\n", run<21, 8, true, 4>(w, out, iters), run<21, 8, true, 3>(w, out, iters));
  return 0;
}
