// lds_probe.hip -- what a 16-byte LDS read costs when only half of the wave's lanes take part (K5g's k-groups 2, 3 hold the
// data of k-groups 0, 1 shifted by one entry: they could come from a cross-lane move instead of the LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>      // 0: all 64 lanes read; 1: lanes 0-31 read; 2: lanes 0-31 read + rebuild the upper half (DPP rotate + v_permlane32_swap)
__global__ __launch_bounds__(256, 1) void probe(float *out, int iters) {
  extern __shared__ __align__(16) unsigned char sm[];
  const int lane = threadIdx.x & 63, g = lane >> 4, p = lane & 15;
  for (int i = threadIdx.x; i < 32768; i += 256) reinterpret_cast<float *>(sm)[i] = float(i);
  __syncthreads();
  f32x4 acc = {0, 0, 0, 0};
  int off = (g & 1) * 16384 + ((g >> 1) + p) % 16 * 16 + (threadIdx.x >> 6) * 4096;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      f32x4 v = {0, 0, 0, 0};
      if (MODE == 0 || lane < 32) v = *reinterpret_cast<const f32x4 *>(sm + ((off + r * 256) & 0x1ffff));
      if (MODE == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k]), 0x12F /* row_ror:15 */, 0xf, 0xf, false);
          int a = __builtin_bit_cast(int, v[k]);
          asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(t));
          v[k] = __builtin_bit_cast(float, a);
        }
      }
      acc += v;
    }
    off += 16;
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

template <int MODE> double run(float *out, int iters) {
  auto k = probe<MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 140 << 10);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(256), 140 << 10, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(256), dim3(256), 140 << 10, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 * 2.38e9 / (double(iters) * 16);      // cycles per read instruction per wave (4 waves share the LDS)
}

int main() {
  float *out; hipMalloc(&out, 64);
  const int iters = 20000;
  printf("cycles per ds_read_b128 per wave, 4 waves per CU:\n");
  printf("  all 64 lanes                     : %.1f\n", run<0>(out, iters));
  printf("  lanes 0-31 only                  : %.1f\n", run<1>(out, iters));
  printf("  lanes 0-31 + rebuild upper half  : %.1f\n", run<2>(out, iters));
  return 0;
}
