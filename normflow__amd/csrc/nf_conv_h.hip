// nf_conv_h.hip -- K5h: the fused last ConvAct layer (8 -> 46 channels at the active sites + RQ-spline coupling)
// with every fp32 product computed as THREE fp16 matrix-core products,
//        a * w  ~=  a_hi*w_hi + a_hi*w_lo + a_lo*w_hi        (x_hi = fp16(x), x_lo = fp16(x - x_hi)),
// accumulated in fp32 (v_mfma_f32_16x16x16_f16).  The dropped term a_lo*w_lo is 2^-22 relative; measured against the
// fp64 definition the layer's error is ~1.7x that of an fp32 fmaf chain over the same 648 terms (rms 1.1e-6 vs 6.5e-7
// on O(1) outputs), inside the 1e-5 budget -- and the fp16 pipe is 16x the fp32 one per product, so a product costs
// 3/16 of an fp32 MFMA slot (tools/mfma_probe3.hip: 370-410 fp32-equivalent TFLOP/s where the fp32 loop tops out at ~130).
// fp16 has a short exponent: the kernel is only used when the caller guarantees |input| <= 1 (NF_CONV_UNIT_INPUT:
// the hidden activations are tanh outputs) and the weights are finite fp16-range numbers (checked on the host side).
//
// Shape of the computation (weight-stationary; reference: src/nn/scalar/modules.py:120-145 + couplings_.py:178-200):
//   * one persistent workgroup per CU, 4 waves.  Waves 0-2 each own one 16-column tile of the 46 logit channels and keep
//     its B fragments -- hi and lo, 41 K-slices of 16 = (2 taps x 8 input channels) -- in 164 registers for the whole
//     launch; per item (sample, 2x2x2x32 box -> 128 active sites = 8 site tiles) a wave issues 8 x 41 x 3 MFMAs and reads
//     only A fragments from LDS.
//   * wave 3 is the data mover: while the others multiply item m it (a) runs the RQ-spline epilogue of item m-1 on the
//     logits the compute waves left in LDS (the logits never reach HBM) and (b) stages item m+1: 64 halo rows x 8
//     channel planes -> split into hi/lo fp16 -> two channel-last LDS images of 16 bytes per site.  With the taps of a
//     slice adjacent along the fastest axis and the active sites at stride 2, the 64 lanes of an A read cover 512
//     CONTIGUOUS bytes (no bank conflicts); slices pairing the third taps of two kernel rows pay a 2-way conflict.
//   * two barriers per item; LDS = 2 x (2 x 34 KB) tile images + 23 KB of logits = 159 KB.
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "nf_conv_core.h"

namespace nf {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace h {
constexpr int H0 = 4, H1 = 4, H2 = 4, H3 = 34;            // halo box of a 2x2x2x32 box under a 3^4 kernel
constexpr int NSITE = H0 * H1 * H2 * H3;                  // 2176
constexpr int NROW = H0 * H1 * H2;                        // 64 halo rows: one per lane of the loader wave
constexpr int IMG = NSITE * 16;                           // bytes of one fp16 image (8 channels per site)
// An image is four sub-images [channel quad q][parity pi of the halo index z3][row][17 entries of 8 bytes]: the 16 lanes of a
// k-group of an A fragment read sites of ONE parity (active sites sit at stride 2) and ONE channel quad, i.e. 16
// consecutive 8-byte entries = 128 contiguous bytes, all banks once.  (With 16 bytes per site in site order the same
// read strides 32 bytes and is a 4-way bank conflict: measured 35 instead of ~18 cycles per MFMA.)
constexpr int ROWB = 17 * 8;                              // bytes of a row in a sub-image
constexpr int SUB = NROW * ROWB;                          // 8704 bytes
__host__ __device__ constexpr int rowidx(int r) { return ((r / 9) * H1 + (r / 3) % 3) * H2 + r % 3; }   // kernel row -> halo row step
constexpr int NS = 41;                                    // K slices: 27 (taps 0,1 of a kernel row) + 14 (third taps of two rows)
constexpr int UNITS = 128;                                // active sites per box
constexpr int C = 46, M = 16;
constexpr int PT = C * UNITS * 4;                         // bytes of the logit scratch
constexpr int LDS_BYTES = 4 * IMG + PT;

__host__ __device__ constexpr int rowoff(int r) {         // halo-site offset of kernel row r = (j0, j1, j2)
  return (((r / 9) * H1 + (r / 3) % 3) * H2 + r % 3) * H3;
}
}  // namespace h

template <int FUSE>
__global__ __launch_bounds__(256, 1) void conv_h_kernel(ConvArgs A) {
  using namespace h;
  extern __shared__ __align__(16) unsigned char smem_h[];
  float *pt = reinterpret_cast<float *>(smem_h + 4 * IMG);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4;

  const int nb = gridDim.x;
  const int vb = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);
  if (vb >= A.nitems) return;
  const int n_my = int((A.nitems - vb + nb - 1) / nb);
  auto decode = [&](int it, int &b, int (&o)[4]) {
    b = it / A.nboxes;
    int bid = it - b * A.nboxes;
#pragma unroll
    for (int mu = 3; mu >= 0; --mu) {
      o[mu] = (bid % A.nbox[mu]) * A.box[mu];
      bid /= A.nbox[mu];
    }
  };

  if (wave < 3) {
    // ============================================================ compute waves: column tile `wave`
    const f16x4 *__restrict__ wsp = static_cast<const f16x4 *>(A.wfrag) + (wave * NS * 2) * 64 + lane;
    f16x4 bh[NS], bl[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      bh[s] = wsp[(2 * s) * 64];
      bl[s] = wsp[(2 * s + 1) * 64];
    }
    const int co = (wave << 4) + (lane & 15);
    const float bv = (A.bias && co < A.cout) ? static_cast<const float *>(A.bias)[co] : 0.f;
    // byte offsets of this lane's A reads inside an image, per site tile (= box row mt: z0 = mt>>2, z1 = (mt>>1)&1, z2 = mt&1).
    // Box extents are even, so the parity of a row does not depend on the box.
    int ta[8], tb[8];      // byte offsets in an image: ta for taps (0, 1) of a kernel row [tap = g>>1], tb for tap 2
    {
      const int q = g & 1, ts = g >> 1, p = lane & 15;
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const int z0 = mt >> 2, z1 = (mt >> 1) & 1, z2 = mt & 1;
        const int par = (A.parity + z0 + z1 + z2) & 1;
        const int r0 = (z0 * H1 + z1) * H2 + z2;
        const int za = 2 * p + par + ts;                  // halo index of tap ts
        ta[mt] = (q * 2 + (za & 1)) * SUB + (r0 * 17 + (za >> 1)) * 8;
        tb[mt] = (q * 2 + par) * SUB + (r0 * 17 + p + 1) * 8;       // tap 2: same parity as the site, one entry on
      }
    }
    const bool second = (g >> 1) != 0;
    lds_barrier();            // P: the mover has staged the first image
    for (int m = 0; m < n_my; ++m) {
      const unsigned char *imgH = smem_h + (m & 1) * 2 * IMG;
      const unsigned char *imgL = imgH + IMG;
      f32x4 acc[8];
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      // 41 slices of 8 site tiles; the A fragments (hi, lo) of a slice are read TWO slices ahead into three named
      // buffers (one wave per SIMD: nobody else hides the LDS latency), and the 24 MFMAs of a slice run hi*hi over
      // the 8 tiles, then hi*lo, then lo*hi, so that an accumulator is touched every 8th MFMA only.
      f16x4 ahA[8], alA[8], ahB[8], alB[8], ahC[8], alC[8];
      auto fetch = [&](f16x4 (&ah)[8], f16x4 (&al)[8], int s) {
        if (s < 27) {                       // taps 0 and 1 of kernel row s: the second tap is the next site
          const int off = rowidx(s) * ROWB;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            ah[i] = *reinterpret_cast<const f16x4 *>(imgH + ta[i] + off);
            al[i] = *reinterpret_cast<const f16x4 *>(imgL + ta[i] + off);
          }
        } else {                            // third taps of kernel rows 2i and 2i+1 (the last slice: row 26 and padding)
          const int i2 = s - 27;
          const int offA = rowidx(2 * i2) * ROWB;
          const int offB = rowidx(2 * i2 + 1 < 27 ? 2 * i2 + 1 : 26) * ROWB;
          const int sel = second ? offB : offA;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            ah[i] = *reinterpret_cast<const f16x4 *>(imgH + tb[i] + sel);
            al[i] = *reinterpret_cast<const f16x4 *>(imgL + tb[i] + sel);
          }
        }
      };
      auto mult = [&](const f16x4 (&ah)[8], const f16x4 (&al)[8], int s) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[i], bh[s], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[i], bl[s], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(al[i], bh[s], acc[i], 0, 0, 0);
      };
      if (!(A.dbg & 256)) {     // dbg 256: timing ablation, no MFMA loop
        fetch(ahA, alA, 0);
        fetch(ahB, alB, 1);
#pragma unroll
        for (int s3 = 0; s3 < NS; s3 += 3) {
          if (s3 + 2 < NS) fetch(ahC, alC, s3 + 2);
          __builtin_amdgcn_sched_barrier(0);
          mult(ahA, alA, s3);
          __builtin_amdgcn_sched_barrier(0);
          if (s3 + 1 < NS) {
            if (s3 + 3 < NS) fetch(ahA, alA, s3 + 3);
            __builtin_amdgcn_sched_barrier(0);
            mult(ahB, alB, s3 + 1);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (s3 + 2 < NS) {
            if (s3 + 4 < NS) fetch(ahB, alB, s3 + 4);
            __builtin_amdgcn_sched_barrier(0);
            mult(ahC, alC, s3 + 2);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      lds_barrier();            // B1: this image is consumed; the mover is done with the previous logits and the next image
      // logits (+bias) -> pt[channel][unit]: unit = 16*mt + 4g + r
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        f32x4 v = acc[mt];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv;
        if (co < C) *reinterpret_cast<f32x4 *>(pt + co * UNITS + (mt << 4) + (g << 2)) = v;
      }
      lds_barrier();            // B2: logits complete
    }
    return;
  }

  // ================================================================ wave 3: stage the next item, finish the previous one
  const float *__restrict__ in = static_cast<const float *>(A.in);
  // lane l holds halo row l: its (z0, z1, z2) and, per box, its offset in a channel plane of the input
  const int rz0 = lane / (H1 * H2), rz1 = (lane / H2) % H1, rz2 = lane % H2;
  auto row_offsets = [&](const int (&o)[4]) {
    int x0 = o[0] + rz0 - 1, x1 = o[1] + rz1 - 1, x2 = o[2] + rz2 - 1;
    x0 = x0 < 0 ? x0 + A.L[0] : (x0 >= A.L[0] ? x0 - A.L[0] : x0);
    x1 = x1 < 0 ? x1 + A.L[1] : (x1 >= A.L[1] ? x1 - A.L[1] : x1);
    x2 = x2 < 0 ? x2 + A.L[2] : (x2 >= A.L[2] ? x2 - A.L[2] : x2);
    return ((x0 * A.L[1] + x1) * A.L[2] + x2) * A.L[3];
  };
  // Two halo rows per pass: lanes 0-31 take the 32 interior sites of row 2i, lanes 32-63 those of row 2i+1 (one 128-byte
  // line per channel and row); the two halo sites of a row are copies of its own end sites (the box spans the axis).
  constexpr int PB = 4;                          // passes per batch: 32 loads in flight per batch, two batches in flight
  const int rs = lane >> 5, xs = lane & 31;
  auto stage = [&](int b, const int (&o)[4], unsigned char *imgH) {
    unsigned char *imgL = imgH + IMG;
    const float *__restrict__ src = in + int64_t(b) * 8 * A.V + xs;
    const int myoff = row_offsets(o);
    float v0[PB][8], v1[PB][8];
    auto issue = [&](float (&v)[PB][8], int p0) {
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        const int oa = __builtin_amdgcn_readlane(myoff, 2 * (p0 + j));
        const int ob = __builtin_amdgcn_readlane(myoff, 2 * (p0 + j) + 1);
        const int off = rs ? ob : oa;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[j][c] = src[int64_t(c) * A.V + off];
      }
    };
    auto commit = [&](const float (&v)[PB][8], int p0) {
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        f16x8 hi, lo;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const _Float16 hh = static_cast<_Float16>(v[j][c]);
          hi[c] = hh;
          lo[c] = static_cast<_Float16>(v[j][c] - static_cast<float>(hh));
        }
        const int row = 2 * (p0 + j) + rs;
        const f16x4 h0 = {hi[0], hi[1], hi[2], hi[3]}, h1 = {hi[4], hi[5], hi[6], hi[7]};
        const f16x4 l0 = {lo[0], lo[1], lo[2], lo[3]}, l1 = {lo[4], lo[5], lo[6], lo[7]};
        auto put = [&](int z3) {                  // halo index z3 of this row <- the lane's site
          const int d = (z3 & 1) * SUB + (row * 17 + (z3 >> 1)) * 8;
          *reinterpret_cast<f16x4 *>(imgH + d) = h0;
          *reinterpret_cast<f16x4 *>(imgH + d + 2 * SUB) = h1;
          *reinterpret_cast<f16x4 *>(imgL + d) = l0;
          *reinterpret_cast<f16x4 *>(imgL + d + 2 * SUB) = l1;
        };
        put(xs + 1);
        if (xs == 0) put(H3 - 1);                 // periodic copies: site 0 -> right halo, site 31 -> left halo
        if (xs == 31) put(0);
      }
    };
    issue(v0, 0);
#pragma unroll 1
    for (int p0 = 0; p0 < NROW / 2; p0 += 2 * PB) {
      issue(v1, p0 + PB);
      commit(v0, p0);
      if (p0 + 2 * PB < NROW / 2) issue(v0, p0 + 2 * PB);
      commit(v1, p0 + PB);
    }
  };
  auto epilogue = [&](int b, const int (&o)[4], int64_t pidx) {
    double lacc = 0.0;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const int u = pass * 64 + lane;
      const int mt = u >> 4, p3 = u & 15;
      const int x0 = o[0] + (mt >> 2), x1 = o[1] + ((mt >> 1) & 1), x2 = o[2] + (mt & 1);
      const int offp = (A.parity + x0 + x1 + x2) & 1;
      const int64_t pair = int64_t(b) * (A.V / 2) + ((int64_t(x0) * A.L[1] + x1) * A.L[2] + x2) * (A.L[3] / 2) + p3;
      const float2 xv = reinterpret_cast<const float2 *>(A.xact)[pair];
      RegCol<float, C> a;
#pragma unroll
      for (int c = 0; c < C; ++c) a[c] = pt[c * UNITS + u];
      float val, logd;
      rqs_site<float, M, FUSE == 2>(a, A.P, offp ? xv.y : xv.x, val, logd);
      float2 ov;
      ov.x = offp ? 0.f : val;
      ov.y = offp ? val : 0.f;
      reinterpret_cast<float2 *>(A.yout)[pair] = ov;
      lacc += double(logd);
    }
    const double tot = wave_sum(lacc);
    if (lane == 0) A.partial[pidx] = tot;
  };

  int pb = 0, po[4] = {0, 0, 0, 0};            // the item whose logits sit in pt
  int cb, co4[4];
  decode(vb, cb, co4);
  stage(cb, co4, smem_h);
  lds_barrier();                                // P: image 0 ready
  for (int m = 0; m < n_my; ++m) {
    if (m > 0 && !(A.dbg & 128)) epilogue(pb, po, int64_t(vb) + int64_t(m - 1) * nb);     // dbg 128: timing ablation
    int nb_ = cb, no_[4] = {co4[0], co4[1], co4[2], co4[3]};
    if (m + 1 < n_my) {
      decode(vb + (m + 1) * nb, nb_, no_);
      if (!(A.dbg & 64)) stage(nb_, no_, smem_h + ((m + 1) & 1) * 2 * IMG);                 // dbg 64: timing ablation
    }
    lds_barrier();                              // B1
    pb = cb;
#pragma unroll
    for (int mu = 0; mu < 4; ++mu) { po[mu] = co4[mu]; co4[mu] = no_[mu]; }
    cb = nb_;
    lds_barrier();                              // B2: logits of item m are in pt
  }
  epilogue(pb, po, int64_t(vb) + int64_t(n_my - 1) * nb);
}

// 1 = launched (dry: would launch), 0 = not this kernel's layer, < 0 error.  A0 is nf_conv.hip's planned argument block.
int launch_conv_h(const ConvArgs &A0, int64_t B, int64_t nboxes, int fuse, hipStream_t stream, bool dry) {
  using namespace h;
  static const int off = getenv("NF_CONV_SPLIT16") ? (atoi(getenv("NF_CONV_SPLIT16")) == 0) : 0;
  if (off || !fuse) return 0;
  ConvArgs A = A0;
  if (A.cin != 8 || A.cout != C || A.P.m != M || A.P.fx || A.P.fy || (A.dbg & 15) || A.stamps) return 0;
  for (int mu = 0; mu < 4; ++mu)
    if (A.k[mu] != 3) return 0;
  if (A.box[0] != 2 || A.box[1] != 2 || A.box[2] != 2 || A.box[3] != 32 || A.L[3] != 32) return 0;
  for (int mu = 0; mu < 3; ++mu)
    if (A.L[mu] < 2 || (A.L[mu] & 1)) return 0;           // even extents: whole boxes, row parity independent of the box
  if (dry) return 1;
  A.nitems = B * nboxes;
  A.nboxes = int(nboxes);
  if (A.nitems >= (int64_t(1) << 31) - 4096) return -2;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    ncu = prop.multiProcessorCount;
  }
  int64_t grid = ncu;
  if (grid > A.nitems) grid = A.nitems;
  grid = (grid + 7) & ~int64_t(7);
  if (fuse == 1) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_h_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -1;
    hipLaunchKernelGGL((conv_h_kernel<1>), dim3(unsigned(grid)), dim3(256), LDS_BYTES, stream, A);
  } else {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_h_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -1;
    hipLaunchKernelGGL((conv_h_kernel<2>), dim3(unsigned(grid)), dim3(256), LDS_BYTES, stream, A);
  }
  return 1;
}

}  // namespace nf
