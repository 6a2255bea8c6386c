// mfma_probe5.hip -- the K-split loop of conv_h_kernel in isolation: one wave holds the B fragments of 7 K slices x 3
// column tiles (hi, lo: 168 registers) and 8 x 3 accumulators; A fragments fixed in registers.  Which MFMA ORDER keeps
// the matrix pipe full?  V = 0: per (slice, site tile) 9 MFMAs with the same A, B changing (what the kernel did);
// V = 1: per (slice, product, column tile) 8 MFMAs with the same B, A changing.  fp32-equivalent TFLOP/s, 3 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int V, int NSET, int BAR = 0>
__global__ __launch_bounds__(256, 1) void probe(const f16x8 *__restrict__ w, float *out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= 3) {           // BAR: the fourth wave only keeps the barriers company, like an idle mover
    if (BAR) for (int it = 0; it < iters * NSET * BAR; ++it) __builtin_amdgcn_s_barrier();
    return;
  }
  f16x8 bh[7][3], bl[7][3];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int n = 0; n < 3; ++n) { bh[i][n] = w[((i * 3 + n) * 2) * 64 + lane]; bl[i][n] = w[((i * 3 + n) * 2 + 1) * 64 + lane]; }
  f32x4 acc[NSET][8][3];
#pragma unroll
  for (int s = 0; s < NSET; ++s)
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int n = 0; n < 3; ++n) acc[s][m][n] = f32x4{0, 0, 0, 0};
  f16x8 ah[8], al[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) { ah[m] = w[lane + 64 * m]; al[m] = w[lane + 64 * m + 512]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        if (V == 0) {
#pragma unroll
          for (int m = 0; m < 8; ++m) {
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[s][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[i][n], acc[s][m][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[s][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[i][n], acc[s][m][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[s][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[i][n], acc[s][m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (BAR && (i * 8 + m + 1) % (56 / BAR) == 0) __builtin_amdgcn_s_barrier();
          }
        } else {
#pragma unroll
          for (int n = 0; n < 3; ++n) {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[s][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[i][n], acc[s][m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[s][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[i][n], acc[s][m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[s][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[i][n], acc[s][m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
  }
  float sum = 0;
  for (int s = 0; s < NSET; ++s) for (int m = 0; m < 8; ++m) for (int n = 0; n < 3; ++n) for (int r = 0; r < 4; ++r) sum += acc[s][m][n][r];
  if (sum == 12345.678f) out[0] = sum;
}

template <int V, int NSET, int BAR = 0> double run(const f16x8 *w, float *out, int iters) {
  auto k = probe<V, NSET, BAR>;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, w, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, w, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * 3 * iters * NSET * 7 * 8 * 3 * (2.0 * 16 * 16 * 32);     // one of the three products counted
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  f16x8 *w; float *out;
  hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20);
  hipMalloc(&out, 64);
  const int iters = 1000;
  printf("K-split loop, 3 waves/CU, fp32-equivalent TFLOP/s (ceiling 625):\n");
  printf("  same A, B changing (9 per step), 1 accumulator set : %.0f\n", run<0, 1>(w, out, iters));
  printf("  same A, B changing (9 per step), 2 accumulator sets: %.0f\n", run<0, 2>(w, out, iters));
  printf("  same B, A changing (8 per run),  1 accumulator set : %.0f\n", run<1, 1>(w, out, iters));
  printf("  same B, A changing (8 per run),  2 accumulator sets: %.0f\n", run<1, 2>(w, out, iters));
  printf("  same A, 2 sets, 4 barriers per 56 steps (+ idle 4th wave): %.0f\n", run<0, 2, 4>(w, out, iters));
  printf("  same A, 2 sets, 1 barrier per 56 steps: %.0f\n", run<0, 2, 1>(w, out, iters));
  printf("  same A, 2 sets, 8 barriers per 56 steps: %.0f\n", run<0, 2, 8>(w, out, iters));
  return 0;
}
