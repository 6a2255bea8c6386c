// mfma_probe6.hip -- how much of the fp16 matrix pipe's rate survives REAL operands?  The same dependent-accumulator MFMA
// stream with all-zero operands and with random fp16 operands, for the 16x16x32 and the 32x32x16 shape (the larger tile
// reads half as many operand registers per flop), 3 and 4 waves per CU.  TFLOP/s of executed fp16 (peak 2500).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int NWAVE>
__global__ __launch_bounds__(256, 1) void probe(const f16x8 *__restrict__ w, float *out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= NWAVE) return;
  f16x8 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = w[(wave * 16 + i) * 64 + lane]; b[i] = w[(wave * 16 + 8 + i) * 64 + lane]; }
  float sum = 0;
  if (SHAPE == 0) {
    f32x4 acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[j], acc[m], 0, 0, 0);
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 4; ++r) sum += acc[m][r];
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b[j], acc[m], 0, 0, 0);
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) sum += acc[m][r];
  }
  if (sum == 12345.678f) out[0] = sum;
}

template <int SHAPE, int NWAVE> double run(const f16x8 *w, float *out, int iters) {
  auto k = probe<SHAPE, NWAVE>;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, w, out, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, w, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double per = SHAPE == 0 ? 64.0 * 2 * 16 * 16 * 32 : 32.0 * 2 * 32 * 32 * 16;     // flops per iteration and wave
  return 256.0 * NWAVE * iters * per / (ms * 1e-3) / 1e12;
}

int main() {
  f16x8 *w; float *out;
  const size_t n = 1 << 20;
  (void)hipMalloc(&w, n); (void)hipMalloc(&out, 64);
  for (int real = 0; real < 2; ++real) {
    std::vector<_Float16> h(n / 2);
    srand(7);
    for (auto &x : h) x = real ? _Float16((rand() / double(RAND_MAX) - 0.5) * 0.25) : _Float16(0);     // small: the accumulators stay finite
    (void)hipMemcpy(w, h.data(), n, hipMemcpyHostToDevice);
    const int iters = 4000;
    printf("%s operands, executed fp16 TFLOP/s (peak 2500; 1875 with 3 of 4 SIMDs):\n", real ? "random" : "zero");
    printf("  16x16x32: 3 waves %.0f   4 waves %.0f\n", run<0, 3>(w, out, iters), run<0, 4>(w, out, iters));
    printf("  32x32x16: 3 waves %.0f   4 waves %.0f\n", run<1, 3>(w, out, iters), run<1, 4>(w, out, iters));
  }
  return 0;
}
