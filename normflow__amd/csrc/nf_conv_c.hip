// nf_conv_c.hip -- K5c: the FIRST ConvAct layer (1 -> 8 channels, 3^4 kernel, tanh / logistic) in front of split-fp16
// consumers: fp32 field in, fp16 (hi, lo) pair tensor out (include/normflow_hip.h, nf_conv_first_split16; reference: the first
// Conv4d + activation of src/nn/scalar/modules.py:120-145, src/nn/scalar/convNd.py:86-126, fed by the frozen half of the
// field, src/nn/scalar/couplings_.py:179-181).
//
// The layer is 36 bytes per site of data movement (4 in, 32 out) and 648 MACs: with fp32 matrix instructions (27 x
// v_mfma_f32_16x16x4_f32 per 16 site pairs, 864 cycles) the arithmetic alone is ~3 ms per 256 samples of 32^4 -- twice the
// HBM time.  Here the products are split as in the other two layers (x = x_hi + x_lo in fp16, weights scaled by 2^10 and
// split the same way, three v_mfma_f32_16x16x32_f16 per K = 32 slice, fp32 accumulation): two-site columns make
// K = 27 kernel rows x 4 taps = 108 -> 4 slices, 12 MFMAs (192 cycles) per tile.  |x| must stay below the fp16 range (6.5e4;
// beyond it the outputs are NaN, never silently wrong); small |x| lose nothing that matters: the lo part carries an ABSOLUTE
// error of 3e-8.
//
// Shape (the pattern of nf_conv_g.hip): persistent workgroups of 8 waves march columns of C0 x C1 lattice rows along axis 2,
// one plane per step.  The field is tiny next to the output (4 B against 32 B per site), so the halo'd input plane
// ((C0+2) x (C1+2) rows) is loaded into registers two steps ahead, split into (hi, lo) halves once and kept in a 4-plane LDS
// ring; an A fragment is two 8-byte windows of that image.  Every wave owns the tiles (rows) w, w+8 of the plane and runs
// their epilogue itself, straight from the accumulators (the weights are the MFMA's A operand, so a lane ends up with four
// channels of one site): bias, activation, split, 8-byte stores into the row-major pair tensor.  Scalar bookkeeping is incremental; one barrier per step.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "nf_conv_core.h"

namespace nf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace c2 {
constexpr int SEG = 32;                       // sites of a SEGMENT of the fastest axis: one MFMA tile of 16 pairs.  A lattice row of L3
                                              // sites is ceil(L3 / 32) segments; a column of the march is (cross-section, segment)
#ifndef NF_C2_IROW
#define NF_C2_IROW 80
#endif
constexpr int IROW = NF_C2_IROW;                      // bytes of an input row image: 34 halves (sites -1 .. 32) + pad, 8-byte multiple;
                                              // (positions 0 .. 33 = sites 32 h - 1 .. 32 h + 32 of segment h, periodic in L3)
constexpr int MAXROWS = 36;                   // (4 + 2) x (4 + 2) halo rows of a plane
constexpr int PLANE = 2 * MAXROWS * IROW;     // hi image + lo image of one plane: 5760 B
constexpr int NPL = 4;                        // ring: planes z-1, z, z+1 being read + one being written
constexpr int LDS_BYTES = NPL * PLANE;                      // 23040
constexpr float kInvWScale = 1.0f / 1024.0f;
}  // namespace c2

// tanh / logistic of a * scale + bias in one branch-free form: al / (1 + 2^(c1 a + c0)) + ga (nf_conv_g.hip, act_affine)
__device__ __forceinline__ float act_affine_c(float a, float c1, float c0, float al, float ga) {
  const float t = __builtin_amdgcn_exp2f(__builtin_fmaf(a, c1, c0));
  return __builtin_fmaf(al, __builtin_amdgcn_rcpf(1.0f + t), ga);
}

// C0 x C1 = rows of the cross-section (2 or 4 each).  HK (round 3): the launch covers the 8-pair last segment of a 48-, 80-, ... wide
// row and packs a wave's two rows (w, w + 8 of a 4 x 4 cross-section) into ONE tile -- lanes p < 8 row w, lanes p >= 8 row w + 8,
// pair 16 hs + (p & 7) --: half the MFMAs and fragment reads; the full segments go through the plain instance first.
template <bool HK>
__global__ __launch_bounds__(512, 2) void conv_c2_kernel(ConvArgs A) {
  using namespace c2;
  extern __shared__ __align__(16) unsigned char smem_c2[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, p = lane & 15;
  const int C0 = A.box[0], C1 = A.box[1];                 // cross-section (rows), from the launcher
  const int H1 = C1 + 2, NROWS = (C0 + 2) * H1;           // halo rows of a plane
  const int NT = C0 * C1;                                 // tiles (output rows) per plane
  const int n0 = A.L[0] / C0, n1 = A.L[1] / C1, ncol = n0 * n1;
  const int L2 = A.L[2], L3 = A.L[3];
  const int RB = L3 * 32, HB = L3 * 16;                    // bytes of an output row of the pair tensor / of its hi block
  const int HP = L3 >> 1, NSEG = (HP + 15) >> 4;           // pairs per row; segments per row
  const int seg_lo = A.seg_lo, seg_n = A.seg_n > 0 ? A.seg_n : NSEG;      // the segments this launch covers
  const int total = int(A.nitems);                        // B * ncol * seg_n columns
  const int nwg = gridDim.x;
  // my columns: blockIdx, blockIdx + nwg, ...  (a column of this layer reads 4 B and writes 32 B per site: no halo traffic to
  // speak of, so no XCD-aware grouping is needed)
  if (int(blockIdx.x) >= total) return;
  const int ncols_my = (total - int(blockIdx.x) + nwg - 1) / nwg;
  auto decode = [&](int ci, int &b, int &i0, int &i1, int &hs) {
    int gc = int(blockIdx.x) + ci * nwg;
    hs = seg_lo + gc % seg_n;
    gc /= seg_n;
    b = gc / ncol;
    const int c = gc - b * ncol;
    i0 = c / n1;
    i1 = c - i0 * n1;
  };

  // ---- weights: 4 slices, hi and lo: [slice][hi|lo][lane][8]
  f16x8 bh[4], bl[4];
  {
    const f16x8 *__restrict__ wsp = static_cast<const f16x8 *>(A.wfrag) + lane;
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      bh[sl] = wsp[(2 * sl) * 64];
      bl[sl] = wsp[(2 * sl + 1) * 64];
    }
  }
  float bv4[4], kc0[4];                        // bias of the four channels this lane ends up with: 4 (g & 1) + r
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    bv4[r] = A.bias ? static_cast<const float *>(A.bias)[4 * (g & 1) + r] : 0.f;
  }
  const bool is_tanh = A.act != kActSigmoid;
  const float kcs = is_tanh ? 2.885390081777927f : -1.4426950408889634f;       // 2 log2(e) / -log2(e)
  const float kal = is_tanh ? -2.0f : 1.0f, kga = is_tanh ? 1.0f : 0.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r) kc0[r] = kcs * bv4[r];
  const float kc1 = kcs * kInvWScale;

  // ---- A fragments.  K index = 4 r + t: kernel row r = (j0 * 3 + j1) * 3 + j2 (27, padded to 32 with zero weights), tap t of
  // the pair (sites 2p - 1 .. 2p + 2).  Slice sl, k-group g: rows rA = 8 sl + 2 g and rA + 1.  Lane (p, g) reads, for each, the
  // 8-byte window of 4 halves that starts at image position 2p (the image holds site x at position x + 1).
  // Per tile (row zt = (z0, z1) of the cross-section): window address = ring plane (zs + j2) + ((z0 + j0) * H1 + z1 + j1) * IROW + 4 p.
  int rj2[4][2], roff[4][2];                   // per slice: the two kernel rows' j2 and halo-row byte offset relative to the tile
#pragma unroll
  for (int sl = 0; sl < 4; ++sl)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int r = 8 * sl + 2 * g + h;
      r = r < 27 ? r : 26;                     // padding rows: any valid address (their weights are zero)
      rj2[sl][h] = r % 3;
      roff[sl][h] = ((r / 9) * H1 + (r / 3) % 3) * IROW + 4 * p;
    }

  // ---- staging: the 34 positions of every halo row of a plane are spread over the threads, one float per thread and pass:
  // row = id / 34, position = id % 34 = site 32 hs - 1 + position of the row (mod L3)
  constexpr int NPOS = SEG + 2;
  const int NPASS = (NROWS * NPOS + 511) / 512;  // <= 3 (36 rows x 34 = 1224)
  const float *__restrict__ inb = static_cast<const float *>(A.in);
  float sv[3] = {0.f, 0.f, 0.f};               // loads in flight (issued for plane z+2, committed one step later)
  int srow[3] = {0, 0, 0};                     // element index of this thread's site at plane 0 in pass k, -1 = none; per column
  int sdst[3] = {0, 0, 0};                     // its byte offset in a plane's hi image
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int id = k * 512 + int(threadIdx.x);
    const int row = id / NPOS, pos = id - row * NPOS;
    sdst[k] = row * IROW + pos * 2;
  }
  auto open_stage_column = [&](int ci) {
    int b, i0, i1, hs;
    decode(ci, b, i0, i1, hs);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int id = k * 512 + int(threadIdx.x);
      const int row = id / NPOS, pos = id - row * NPOS;
      if (k < NPASS && row < NROWS) {
        const int hz0 = row / H1, hz1 = row - hz0 * H1;
        int x0 = C0 * i0 + hz0 - 1, x1 = C1 * i1 + hz1 - 1;
        x0 = x0 < 0 ? x0 + A.L[0] : (x0 >= A.L[0] ? x0 - A.L[0] : x0);
        x1 = x1 < 0 ? x1 + A.L[1] : (x1 >= A.L[1] ? x1 - A.L[1] : x1);
        int site = SEG * hs + pos - 1;
        site = site < 0 ? site + L3 : (site >= L3 ? site - L3 : site);           // (32 hs + 32 <= L3 + 16: one wrap is enough)
        srow[k] = ((b * A.L[0] + x0) * A.L[1] + x1) * (L2 * L3) + site;      // < 2^31: checked by the launcher
      } else {
        srow[k] = -1;
      }
    }
  };
  auto stage_issue = [&](int x2) {             // plane x2 (wrapped) of the staging column -> registers
    const int z = x2 < 0 ? x2 + L2 : (x2 >= L2 ? x2 - L2 : x2);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k < NPASS && srow[k] >= 0) sv[k] = inb[srow[k] + z * L3];
  };
  auto stage_commit = [&](int slot) {          // registers -> (hi, lo) images of ring plane `slot`
    unsigned char *ph = smem_c2 + slot * PLANE;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (k < NPASS && srow[k] >= 0) {
        const _Float16 h = static_cast<_Float16>(sv[k]);
        const _Float16 l = static_cast<_Float16>(sv[k] - static_cast<float>(h));
        *reinterpret_cast<_Float16 *>(ph + sdst[k]) = h;
        *reinterpret_cast<_Float16 *>(ph + MAXROWS * IROW + sdst[k]) = l;
      }
    }
  };

  // ---- epilogue addressing: lane (p, g) ends up with channels 4 (g & 1) .. + 3 of site 2 q + (g >> 1), q = 16 hs + p its pair
  unsigned char *__restrict__ outb = static_cast<unsigned char *>(A.out);
  const int64_t sampleB = A.V * 32;
  // my tiles of a plane: rows zt = wave and wave + 8 of the cross-section (row-major (z0, z1))
  int tz[2], toff[2];
  unsigned char *ocol[2] = {nullptr, nullptr};
  unsigned lane_o = 0;
  bool lane_ok = true;                         // my pair exists (a partial last segment has 8 of 16)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    tz[t] = wave + 8 * t;
    const int z0 = tz[t] / C1, z1 = tz[t] - z0 * C1;
    toff[t] = (z0 * H1 + z1) * IROW;
  }
  auto open_column = [&](int ci) {
    int b, i0, i1, hs;
    decode(ci, b, i0, i1, hs);
    const int q = 16 * hs + (HK ? (p & 7) : p);
    lane_ok = q < HP;
    lane_o = unsigned(pair_row_offset(lane_ok ? 2 * q + (g >> 1) : 0, L3) + (g & 1) * 8);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int z0 = tz[t] / C1, z1 = tz[t] - z0 * C1;
      ocol[t] = outb + int64_t(b) * sampleB + int64_t(((C0 * i0 + z0) * A.L[1] + C1 * i1 + z1) * L2) * RB;
    }
  };

  // ---- the march.  Ring entry e = plane (e mod EPC) - 1 of my column e / EPC (EPC = L2 + 2 planes per column: -1 .. L2), kept
  // in slot e & 3.  Iteration e multiplies the output plane whose halo window is entries e, e+1, e+2 -- when those belong
  // to one column (all but the last two iterations of a column: two staging-only iterations per L2, which keeps the loop
  // uniform across column seams) -- then commits entry e + 3 (loaded an iteration ago) and issues the loads of entry e + 4.
  const int EPC = L2 + 2;
  const int nent = ncols_my * EPC;
  int sci = 0, spl = -1;                       // staging cursor: column, plane
  auto stage_advance = [&]() {
    if (++spl > L2) {
      spl = -1;
      if (++sci < ncols_my) open_stage_column(sci);
    }
  };
  open_stage_column(0);
#pragma unroll 1
  for (int e = 0; e < 3; ++e) {                // entries 0, 1, 2 (EPC >= 4: the same column)
    stage_issue(spl);
    stage_commit(e);
    stage_advance();
  }
  bool stage_live = sci < ncols_my;            // entry 3 exists
  if (stage_live) stage_issue(spl);
  lds_barrier();

  int cci = 0, pl = -1;                        // the column and plane of entry e
  open_column(0);
  for (int e = 0; e < nent; ++e) {
    const bool compute = pl <= L2 - 2;         // window planes pl, pl+1, pl+2 inside the column: output plane z = pl + 1
    const int z = pl + 1;
    f32x4 am[2], ac[2];
    if (compute) {
      // (1) my tiles: 4 slices x 3 products
      am[0] = am[1] = ac[0] = ac[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      int poff[3];                             // byte offset of the ring planes z-1, z, z+1
#pragma unroll
      for (int j = 0; j < 3; ++j) poff[j] = ((e + j) & (NPL - 1)) * PLANE;
      int fbase[4][2];                         // this lane's window offsets for the step (shared by its tiles)
#pragma unroll
      for (int sl = 0; sl < 4; ++sl)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j2 = rj2[sl][h];
          fbase[sl][h] = (j2 == 0 ? poff[0] : (j2 == 1 ? poff[1] : poff[2])) + roff[sl][h];
        }
      __builtin_amdgcn_s_setprio(2);           // four waves share a SIMD: the one that multiplies issues first
      // HK: this lane's row of the merged tile (its window starts at pair p & 7 of that row)
      [[maybe_unused]] const int tmerged = (p & 8) ? toff[1] - 32 : toff[0];
#pragma unroll
      for (int t = 0; t < (HK ? 1 : 2); ++t) {
        if (tz[t] < NT) {
#pragma unroll
          for (int sl = 0; sl < 4; ++sl) {
            union { f16x8 v; unsigned w[4]; } fh, fl;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const unsigned char *src = smem_c2 + (HK ? tmerged : toff[t]) + fbase[sl][h];      // 4-byte aligned: two dword reads
              fh.w[2 * h] = *reinterpret_cast<const unsigned *>(src);
              fh.w[2 * h + 1] = *reinterpret_cast<const unsigned *>(src + 4);
              fl.w[2 * h] = *reinterpret_cast<const unsigned *>(src + MAXROWS * IROW);
              fl.w[2 * h + 1] = *reinterpret_cast<const unsigned *>(src + MAXROWS * IROW + 4);
            }
            // weights as the A operand (rows m = column (shift, co) of the layer), site pairs as the B operand: D[m][pair] puts
            // the FOUR CHANNELS 4 (g & 1) .. + 3 of ONE site (2p + (g >> 1)) into each lane -- what a store needs, no transpose
            am[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[sl], fh.v, am[t], 0, 0, 0);
            ac[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[sl], fh.v, ac[t], 0, 0, 0);
            ac[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[sl], fl.v, ac[t], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
    }
    // (2) entry e + 3 (loaded an iteration ago) goes into the ring slot nobody reads; the loads of entry e + 4 are issued
    if (stage_live) {
      stage_commit((e + 3) & (NPL - 1));
      stage_advance();
      stage_live = sci < ncols_my;
      if (stage_live) stage_issue(spl);
    }
    // (3) epilogue of my tiles, straight from the accumulators: lane (p, g) holds channels 4 (g & 1) + r of site 2p + (g >> 1)
    if (compute) {
#pragma unroll
      for (int t = 0; t < (HK ? 1 : 2); ++t) {
        if (tz[t] < NT) {
          // (the packed forms v_pk_add_f32 / v_pk_fma_f32 / v_cvt_pk_f16_f32 for two channels at a time were measured: 2 % slower)
          f16x4 hi, lo;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a = am[t][r] + ac[t][r];
            const float v = act_affine_c(a, kc1, kc0[r], kal, kga);
            const _Float16 h0 = static_cast<_Float16>(v);
            hi[r] = h0;
            lo[r] = static_cast<_Float16>(v - static_cast<float>(h0));
          }
          if (lane_ok) {
            unsigned char *d = (HK ? ((p & 8) ? ocol[1] : ocol[0]) : ocol[t]) + unsigned(z) * unsigned(RB) + lane_o;
            *reinterpret_cast<f16x4 *>(d) = hi;
            *reinterpret_cast<f16x4 *>(d + HB) = lo;
          }
        }
      }
    }
    lds_barrier();                             // entry e + 3 is complete; entry e's slot is free
    if (++pl > L2) {                           // entry e + 1 opens the next column
      pl = -1;
      if (++cci < ncols_my) open_column(cci);
    }
  }
}

}  // namespace nf

using namespace nf;

// 1 if nf_conv_first_split16 takes this layer: 1 -> 8 channels, 3^4 kernel, a fastest axis of 32, 48, 64, ... sites (a multiple of 16), even other extents,
// an activation that keeps |out| <= 1.
extern "C" int nf_conv_first_split16_supported(const int32_t *lattice, const int32_t *ksize, int cout, int act) {
  if (!nf::option(NF_OPT_SPLIT16) || !lattice || !ksize || cout != 8) return 0;
  if (act != kActTanh && act != kActSigmoid) return 0;
  for (int mu = 0; mu < 4; ++mu)
    if (ksize[mu] != 3) return 0;
  if (lattice[3] < 32 || (lattice[3] & 15)) return 0;                     // whole or half segments of 32 sites
  for (int mu = 0; mu < 3; ++mu)
    if (lattice[mu] < 2 || (lattice[mu] & 1)) return 0;
  return 1;
}

extern "C" int nf_conv_first_split16(const void *in, const void *wsplit, const void *bias, void *out16, int64_t B,
                                     const int32_t *lattice, int act, void *stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  NF_REQUIRE(in && wsplit && out16 && lattice, "nf_conv_first_split16: NULL pointer");
  const int32_t k3[4] = {3, 3, 3, 3};
  NF_REQUIRE(nf_conv_first_split16_supported(lattice, k3, 8, act), "nf_conv_first_split16: layer not supported (needs a fastest axis of 32 + 16 n sites, even other extents, tanh / sigmoid)");
  NF_REQUIRE(B >= 0 && B <= 65535, "nf_conv_first_split16: batch outside [0, 65535]");
  if (B == 0) return NF_OK;
  ConvArgs A{};
  A.in = in; A.wfrag = wsplit; A.bias = bias; A.out = out16;
  A.V = 1;
  for (int mu = 0; mu < 4; ++mu) {
    A.L[mu] = lattice[mu]; A.k[mu] = 3;
    A.V *= lattice[mu];
  }
  A.cin = 1; A.cout = 8; A.act = act;
  A.box[0] = lattice[0] % 4 == 0 ? 4 : 2;                              // cross-section of a column
  A.box[1] = lattice[1] % 4 == 0 ? 4 : 2;
  A.nitems = B * int64_t(lattice[0] / A.box[0]) * int64_t(lattice[1] / A.box[1]) * int64_t((lattice[3] / 2 + 15) / 16);   // columns x segments
  NF_REQUIRE(B * A.V < (int64_t(1) << 31), "nf_conv_first_split16: batch x volume >= 2^31 sites, split the batch");
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    NF_REQUIRE(hipGetDeviceProperties(&prop, dev) == hipSuccess, "nf_conv_first_split16: no device properties");
    ncu = prop.multiProcessorCount;
  }
  const int nseg = (lattice[3] / 2 + 15) / 16;
  const int64_t per_seg = A.nitems / nseg;
  auto go = [&](bool hk) {
    int64_t grid = int64_t(2) * ncu;           // two persistent workgroups per CU
    if (grid > A.nitems) grid = A.nitems;
    if (hk)
      hipLaunchKernelGGL(conv_c2_kernel<true>, dim3(unsigned(grid)), dim3(512), c2::LDS_BYTES, stream, A);
    else
      hipLaunchKernelGGL(conv_c2_kernel<false>, dim3(unsigned(grid)), dim3(512), c2::LDS_BYTES, stream, A);
  };
  if (((lattice[3] / 2) & 15) == 8 && A.box[0] == 4 && A.box[1] == 4) {
    // a last segment of 8 pairs: the full segments through the plain kernel, the half columns packed two rows to a tile
    if (nseg > 1) {
      A.seg_lo = 0; A.seg_n = nseg - 1; A.nitems = per_seg * (nseg - 1);
      go(false);
    }
    A.seg_lo = nseg - 1; A.seg_n = 1; A.nitems = per_seg;
    go(true);
  } else {
    A.seg_lo = 0; A.seg_n = nseg;
    go(false);
  }
  return check_launch("conv split-fp16 first-layer kernel");
}
