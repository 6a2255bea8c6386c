// mfma_probe2.hip -- candidate B-operand deliveries for the conv kernel (diagnostic).
//   P1<MT>: per-wave global B loads (9 per row-step) amortised over MT site tiles (A from LDS)
//   P2: B fragments of a row loaded cooperatively (each wave a quarter) into an LDS ring, one barrier per row-step
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MT>
__global__ __launch_bounds__(256) void p1(const float *__restrict__ w, float *out, int iters, int nrows) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = float(i & 7) * 0.125f;
  __syncthreads();
  f32x4 acc[MT][3];
  for (int m = 0; m < MT; ++m) for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
  float a0[3][MT], b0[3][3], a1[3][MT], b1[3][3];
  const float *wl = w + lane;
  int row = 0, off = lane;
  auto request = [&](float (&a)[3][MT], float (&b)[3][3]) {
    const float *wt = wl + row * (3 * 2 * 192);
    for (int j = 0; j < 3; ++j) for (int n = 0; n < 3; ++n) b[j][n] = wt[j * 2 * 192 + n * 64];
    for (int m = 0; m < MT; ++m) for (int j = 0; j < 3; ++j) a[j][m] = lds[off + m * 1024 + j];
  };
  auto multiply = [&](const float (&a)[3][MT], const float (&b)[3][3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int m = 0; m < MT; ++m)
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
  for (int m = 0; m < MT; ++m) for (int n = 0; n < 3; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
  if (s == 12345.678f) out[0] = s;
}

template <int MT>
__global__ __launch_bounds__(256) void p5(const float *__restrict__ w, float *out, int iters, int nrows) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = float(i & 7) * 0.125f;
  __syncthreads();
  f32x4 acc[MT][3];
  for (int m = 0; m < MT; ++m) for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
  float a0[3][MT], b0[3][3], a1[3][MT], b1[3][3];
  const float *wl = w + lane;
  int row = 0, off = lane;
  auto request = [&](float (&a)[3][MT], float (&b)[3][3]) {
    const f32x4 *wt = reinterpret_cast<const f32x4 *>(w) + lane + row * (3 * 2 * 64);
    for (int j = 0; j < 3; ++j) { const f32x4 v = wt[j * 2 * 64]; b[j][0] = v[0]; b[j][1] = v[1]; b[j][2] = v[2]; }
    for (int m = 0; m < MT; ++m) for (int j = 0; j < 3; ++j) a[j][m] = lds[off + m * 1024 + j];
  };
  auto multiply = [&](const float (&a)[3][MT], const float (&b)[3][3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int m = 0; m < MT; ++m)
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
  for (int m = 0; m < MT; ++m) for (int n = 0; n < 3; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
  if (s == 12345.678f) out[0] = s;
}

template <int MT>
__global__ __launch_bounds__(256) void p4(const float *__restrict__ w, float *out, int iters, int nrows) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = float(i & 7) * 0.125f;
  __syncthreads();
  f32x4 acc[MT][3];
  for (int m = 0; m < MT; ++m) for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
  float a0[3][MT], b0[3][3], a1[3][MT], b1[3][3];
  const float *wl = w + lane;
  int row = 0, off = lane;
  auto request = [&](float (&a)[3][MT], float (&b)[3][3]) {
    const float *wt = lds + 8192 + lane + row * (3 * 192);     // the whole chunk's fragments live in LDS (27 x 9 x 256 B = 62 KB)
    for (int j = 0; j < 3; ++j) for (int n = 0; n < 3; ++n) b[j][n] = wt[j * 192 + n * 64];
    for (int m = 0; m < MT; ++m) for (int j = 0; j < 3; ++j) a[j][m] = lds[off + m * 1024 + j];
  };
  auto multiply = [&](const float (&a)[3][MT], const float (&b)[3][3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int m = 0; m < MT; ++m)
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
  for (int m = 0; m < MT; ++m) for (int n = 0; n < 3; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
  if (s == 12345.678f) out[0] = s;
}

template <int MT>
__global__ __launch_bounds__(256) void p3(const float *__restrict__ w, float *out, int iters, int nrows) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = float(i & 7) * 0.125f;
  __syncthreads();
  f32x4 acc[MT][3];
  for (int m = 0; m < MT; ++m) for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
  float a0[3][MT], b0[3][3], a1[3][MT], b1[3][3], a2[3][MT], b2[3][3];
  const float *wl = w + lane;
  int row = 0, off = lane;
  auto request = [&](float (&a)[3][MT], float (&b)[3][3]) {
    const float *wt = wl + row * (3 * 2 * 192);
    for (int j = 0; j < 3; ++j) for (int n = 0; n < 3; ++n) b[j][n] = wt[j * 2 * 192 + n * 64];
    for (int m = 0; m < MT; ++m) for (int j = 0; j < 3; ++j) a[j][m] = lds[off + m * 1024 + j];
  };
  auto multiply = [&](const float (&a)[3][MT], const float (&b)[3][3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][m], b[j][n], acc[m][n], 0, 0, 0);
  };
  auto next = [&]() { row = row + 1 < nrows ? row + 1 : 0; off = (off + 34) & 2047; };
  request(a0, b0);
  next(); request(a1, b1);
  for (int it = 0; it < iters; ++it) {
    next(); request(a2, b2);
    __builtin_amdgcn_sched_barrier(0);
    multiply(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    next(); request(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    multiply(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    next(); request(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    multiply(a2, b2);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0;
  for (int m = 0; m < MT; ++m) for (int n = 0; n < 3; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
  if (s == 12345.678f) out[0] = s;
}

template <int MT>
__global__ __launch_bounds__(256) void p6(const float *__restrict__ w, float *out, int iters, int nrows) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = float(i & 7) * 0.125f;
  __syncthreads();
  f32x4 acc[MT][3];
  for (int m = 0; m < MT; ++m) for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
  float a0[3][MT], b0[3][3], a1[3][MT], b1[3][3], a2[3][MT], b2[3][3];
  const float *wl = w + lane;
  int row = 0, off = lane;
  auto request = [&](float (&a)[3][MT], float (&b)[3][3]) {
    const f32x4 *wt = reinterpret_cast<const f32x4 *>(w) + lane + row * (3 * 2 * 64);
    for (int j = 0; j < 3; ++j) { const f32x4 v = wt[j * 2 * 64]; b[j][0] = v[0]; b[j][1] = v[1]; b[j][2] = v[2]; }
    for (int m = 0; m < MT; ++m) for (int j = 0; j < 3; ++j) a[j][m] = lds[off + m * 1024 + j];
  };
  auto multiply = [&](const float (&a)[3][MT], const float (&b)[3][3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][m], b[j][n], acc[m][n], 0, 0, 0);
  };
  auto next = [&]() { row = row + 1 < nrows ? row + 1 : 0; off = (off + 34) & 2047; };
  request(a0, b0);
  next(); request(a1, b1);
  for (int it = 0; it < iters; ++it) {
    next(); request(a2, b2);
    __builtin_amdgcn_sched_barrier(0);
    multiply(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    next(); request(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    multiply(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    next(); request(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    multiply(a2, b2);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0;
  for (int m = 0; m < MT; ++m) for (int n = 0; n < 3; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
  if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void p2(const float *__restrict__ w, float *out, int iters, int nrows) {
  extern __shared__ float lds[];
  float *ring = lds + 8192;                       // [3 slots][9 fragments][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8192 + 3 * 9 * 64; i += blockDim.x) lds[i] = float(i & 7) * 0.125f;
  __syncthreads();
  f32x4 acc[2][3];
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
  float a0[3][2], b0[9], a1[3][2], b1[9], g[3];
  int row = 0, off = lane, step = 0;
  auto gload = [&](int r) {                       // this wave's share of row r's fragments
    const float *wt = w + lane + r * (2 * 9 * 64);
    g[0] = wt[wave * 64]; g[1] = wt[(wave + 4) * 64]; g[2] = wt[(wave == 0 ? 8 : wave) * 64];
  };
  auto gstore = [&](int slot) {
    float *d = ring + slot * 576 + lane;
    d[wave * 64] = g[0]; d[(wave + 4) * 64] = g[1];
    if (wave == 0) d[8 * 64] = g[2];
  };
  auto request = [&](float (&a)[3][2], float (&b)[9], int slot) {
    const float *rb = ring + slot * 576 + lane;
    for (int f = 0; f < 9; ++f) b[f] = rb[f * 64];
    for (int m = 0; m < 2; ++m) for (int j = 0; j < 3; ++j) a[j][m] = lds[off + m * 1024 + j];
  };
  auto multiply = [&](const float (&a)[3][2], const float (&b)[9]) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][m], b[j * 3 + n], acc[m][n], 0, 0, 0);
  };
  auto next = [&]() { row = row + 1 < nrows ? row + 1 : 0; off = (off + 34) & 2047; };
  gload(0);
  request(a0, b0, 0);
  int s0 = 0;                                     // slot of the row held in (a0, b0)
  for (int it = 0; it < iters; ++it) {
    // step A: compute (a0,b0); read row+1 from slot s0+1; write row+2 (loaded last step) to slot s0+2; load row+3
    const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s1 == 2 ? 0 : s1 + 1;
    gstore(s2);
    next(); gload(row);
    request(a1, b1, s1);
    __builtin_amdgcn_sched_barrier(0);
    multiply(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    gstore(s0);
    next(); gload(row);
    request(a0, b0, s2);
    __builtin_amdgcn_sched_barrier(0);
    multiply(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    s0 = s1 == 2 ? 0 : s1 + 1;   // advanced by two
    s0 = s2;                       // (slot of the row now in a0,b0)
  }
  float s = 0;
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 3; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
  if (s == 12345.678f) out[0] = s + step;
}

template <class K> double run(K kern, int threads, int mt, const float *w, float *out, int blocks_per_cu, int lds_bytes, int iters, double rpi = 2.0) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds_bytes, 0, w, out, iters, 27);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds_bytes, 0, w, out, iters, 27);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = double(grid) * (threads / 64) * iters * rpi * (9 * mt) * 2048.0;
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  float *w, *out;
  hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20);
  hipMalloc(&out, 64);
  const int iters = 3000;
  printf("P1 MT=2, 4 waves/block : 1 blk/CU %.1f   2 blk/CU %.1f   3 blk/CU %.1f\n", run(p1<2>, 256, 2, w, out, 1, 150 << 10, iters), run(p1<2>, 256, 2, w, out, 2, 75 << 10, iters), run(p1<2>, 256, 2, w, out, 3, 50 << 10, iters));
  printf("P1 MT=4, 4 waves/block : 1 blk/CU %.1f   2 blk/CU %.1f\n", run(p1<4>, 256, 4, w, out, 1, 150 << 10, iters), run(p1<4>, 256, 4, w, out, 2, 75 << 10, iters));
  printf("P1 MT=4, 2 waves/block : 2 blk/CU %.1f   4 blk/CU %.1f\n", run(p1<4>, 128, 4, w, out, 2, 75 << 10, iters), run(p1<4>, 128, 4, w, out, 4, 38 << 10, iters));
  printf("P1 MT=8, 4 waves/block : 1 blk/CU %.1f\n", run(p1<8>, 256, 8, w, out, 1, 150 << 10, iters));
  printf("P3 MT=2 distance 2 (x1.5 rows/iter: scale) : 1 blk/CU %.1f   2 blk/CU %.1f  3 blk/CU %.1f\n", run(p3<2>, 256, 2, w, out, 1, 150 << 10, iters, 3.0) , run(p3<2>, 256, 2, w, out, 2, 75 << 10, iters, 3.0), run(p3<2>, 256, 2, w, out, 3, 50 << 10, iters, 3.0));
  printf("P3 MT=4 distance 2: 1 blk/CU %.1f   2 blk/CU %.1f ; MT=3: 1 blk/CU %.1f\n", run(p3<4>, 256, 4, w, out, 1, 150 << 10, iters, 3.0), run(p3<4>, 256, 4, w, out, 2, 75 << 10, iters, 3.0), run(p3<3>, 256, 3, w, out, 1, 150 << 10, iters, 3.0));
  printf("P4 A and B from LDS: MT=2 1 blk/CU %.1f  MT=2 2 blk/CU (8 waves, 2x62KB) %.1f  MT=4 1 blk/CU %.1f\n", run(p4<2>, 256, 2, w, out, 1, 150 << 10, iters), run(p4<2>, 256, 2, w, out, 2, 79 << 10, iters), run(p4<4>, 256, 4, w, out, 1, 150 << 10, iters));
  printf("P5 16-byte B loads, distance 1: MT=2 1 blk %.1f  2 blk %.1f  3 blk %.1f | MT=4 1 blk %.1f  2 blk %.1f\n", run(p5<2>, 256, 2, w, out, 1, 150 << 10, iters), run(p5<2>, 256, 2, w, out, 2, 75 << 10, iters), run(p5<2>, 256, 2, w, out, 3, 50 << 10, iters), run(p5<4>, 256, 4, w, out, 1, 150 << 10, iters), run(p5<4>, 256, 4, w, out, 2, 75 << 10, iters));
  printf("P6 16-byte B loads, distance 2: MT=2 1 blk %.1f  2 blk %.1f  3 blk %.1f | MT=4 1 blk %.1f  2 blk %.1f\n", run(p6<2>, 256, 2, w, out, 1, 150 << 10, iters, 3.0), run(p6<2>, 256, 2, w, out, 2, 75 << 10, iters, 3.0), run(p6<2>, 256, 2, w, out, 3, 50 << 10, iters, 3.0), run(p6<4>, 256, 4, w, out, 1, 150 << 10, iters, 3.0), run(p6<4>, 256, 4, w, out, 2, 75 << 10, iters, 3.0));
  printf("P2 ring,  4 waves/block : 1 blk/CU %.1f   2 blk/CU %.1f   3 blk/CU %.1f\n", run(p2, 256, 2, w, out, 1, 150 << 10, iters), run(p2, 256, 2, w, out, 2, 75 << 10, iters), run(p2, 256, 2, w, out, 3, 50 << 10, iters));
  return 0;
}
