// nf_conv_w.hip -- weight gradient of a 3^4 circular conv layer on the fp16 matrix cores (gfx950), fp32 in and out.
//
//   gw[o][t cin + i] = sum_{b, n} gz[b, o, n] * in[b, i, (n + t - 1) mod L]      (t = the 81 taps, row-major)
//   gw[o][81 cin]    = sum_{b, n} gz[b, o, n]                                       (bias)
//
// what autograd derives for ConvAct's layers in Fitter.step (reference: src/_normflowcore.py:275-294 differentiating
// src/nn/scalar/modules.py:120-145).  A GEMM with the SITES as the reduction axis: M = cout (<= 48), N = 81 cin (+ 1),
// K = B V.  The generic kernel (nf_conv.hip, conv_wgrad_kernel) feeds the fp32 MFMA from per-lane LDS gathers and runs at
// ~38 TFLOP/s; this one is for the shapes of the lattice networks (4-D, 32 sites on the fastest axis, cin 1 or 8) and
// computes every fp32 product as three fp16 products (a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32 accumulation), like the
// forward chain (nf_conv_c/g/h.hip).
//
// One MFMA k-step (K = 32) = one lattice row: lane (column, kq) holds 8 consecutive sites 8 kq .. 8 kq + 7 of the row.
//   A (gz):  lane (m = channel o, kq): 16 contiguous bytes of o's row image -- no shift.
//   B (in):  column n = (kernel row c = (j0, j1, j2), input channel i); the tap j3 along the fastest axis shifts the 8 sites by
//            j3 - 1: a lane reads the 24 bytes around its run once (sites 8 kq - 2 .. 8 kq + 9: one aligned 16-byte read and
//            the two words beside it) and forms the three shifted fragments with v_alignbyte -- the three taps j3 share one
//            read.  Row images carry one wrapped site on either side.
// Work: a workgroup (7 multiplying waves + 1 staging wave) marches "columns" (sample, x0, x1) along axis 2, one lattice row
// per step: the 3 x 3 neighbour rows of plane x2 + 2 are staged (fp32 -> (hi, lo) halfs) into a 4-plane LDS ring while
// plane x2 is multiplied; gz's row is double-buffered.  Multiplying wave w owns column groups 2w, 2w + 1 (a group = two
// kernel rows x 8 channels, or 16 kernel rows x 1 channel; x 3 taps j3) for all cout: 18 accumulator tiles, kept for the
// whole launch; they leave as one partial (cout, 81 cin + 1) matrix per workgroup, summed in a fixed order by a second
// kernel: deterministic, no atomics.
// gz is scaled by a power of two that brings its largest magnitude to ~2^13 (cotangents of a mean over the batch are far
// below fp16's normal range): one max-reduction pass over gz first.
#include <hip/hip_fp16.h>
#include "nf_conv_core.h"

namespace nf {
namespace wg {

constexpr int kThreads = 512;                 // 7 multiplying waves + the stager
constexpr int RS = 208;                       // bytes of a (row, channel) image: 48 hi halfs | 48 lo halfs | pad (conflict-free B reads)
constexpr int LO = 96;                        // offset of the lo halfs in it; site x sits at half 8 + x, wrapped copies at 7 and 40
constexpr int GS = 144;                       // bytes of a gz row image: 32 hi halfs | 32 lo halfs | pad
constexpr int GLO = 64;
constexpr int NSLOT = 4;                      // ring of x2 planes
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Args {
  const float *in, *gz;
  float *partial;                             // (gridDim.x, 48, ncols)
  const unsigned *absmax;                     // bits of max |gz|
  int64_t V;
  int L[4];
  int B, cin, cout, ncols;
  int64_t ncolumns;                           // B L0 L1
  int gz_compact, parity;                     // gz is pair-compact (B, cout, V/2): the active sites (coordinate sum == parity mod 2) only
};

extern __shared__ __align__(16) unsigned char smem_w[];

__device__ __forceinline__ float gz_scale(const unsigned *absmax) { return pow2_scale_for(absmax); }

__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ p, int64_t n, unsigned *out) {
  float m = 0.f;
  const int64_t n4 = n >> 2;
  const float4 *p4 = reinterpret_cast<const float4 *>(p);
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n4; i += int64_t(gridDim.x) * 256) {
    const float4 v = p4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (int64_t i = (n4 << 2) + int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) m = fmaxf(m, fabsf(p[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));      // non-negative floats order like their bits
}

// fp32 -> (hi, lo) halfs of four values
__device__ __forceinline__ void split4(const float4 v, f16x4 &hi, f16x4 &lo) {
  const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const _Float16 h = static_cast<_Float16>(a[r]);
    hi[r] = h;
    lo[r] = static_cast<_Float16>(a[r] - static_cast<float>(h));
  }
}

template <int MT, int CIN, bool GZC, bool SEGM>
__global__ __launch_bounds__(kThreads, 4) void wgrad16_kernel(Args A) {
  constexpr int PLANE = 9 * CIN * RS;           // one x2 plane of the ring: 3 x 3 neighbour rows x channels
  constexpr int NG = CIN == 8 ? 14 : 2;         // column groups of 16: (2 kernel rows x 8 channels) or (16 kernel rows x 1)
  unsigned char *ring = smem_w;
  unsigned char *gbuf = smem_w + NSLOT * PLANE; // two gz row images of 48 channels
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, kq = lane >> 4;
  const int L0 = A.L[0], L1 = A.L[1], L2 = A.L[2];
  const int L3 = A.L[3], NSEG = (L3 + 31) >> 5; // a lattice row is cut into segments of 32 sites (the last one may hold 16)
  const int64_t rowsz = L3;                     // sites of a lattice row (the fastest axis)
  const float scale = gz_scale(A.absmax);

  // zero the gz images once: channels cout .. 47 are never written
  for (int i = threadIdx.x; i < 2 * 48 * GS / 4; i += kThreads) reinterpret_cast<unsigned *>(gbuf)[i] = 0u;
  for (int i = threadIdx.x; i < NSLOT * PLANE / 4; i += kThreads) reinterpret_cast<unsigned *>(ring)[i] = 0u;      // (a partial segment's tail is multiplied by zeros: keep it finite)

  // ---- multiplying waves: per-lane constants of their two groups
  int boff[2] = {0, 0}, bj2[2] = {0, 0};
  bool bvalid[2] = {false, false}, bones[2] = {false, false}, gact[2] = {false, false};
  if (wave < 7) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int g = 2 * wave + q;
      gact[q] = g < NG;
      const int c = CIN == 8 ? 2 * g + (n >> 3) : 16 * g + n;       // kernel row (j0, j1, j2) of this lane's column
      const int ci = CIN == 8 ? (n & 7) : 0;
      bvalid[q] = gact[q] && c < 27;
      bones[q] = gact[q] && c == 27 && ci == 0;                    // the bias column: B = 1
      const int cc = c < 27 ? c : 26;
      const int j0 = cc / 9, j1 = (cc / 3) % 3;
      bj2[q] = cc % 3;
      boff[q] = ((j0 * 3 + j1) * CIN + ci) * RS + 12 + 16 * kq;    // word 0 of the lane's 24 bytes (sites 8 kq - 2 ..)
    }
  }
  f32x4 acc[MT][2][3];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int s = 0; s < 3; ++s) acc[mi][q][s] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- the stager's view: lane -> (channel, quad of sites) of a row
  auto site_row = [&](int x0, int x1, int x2) {         // first site of lattice row (x0, x1, x2), coordinates wrapped
    x0 = x0 < 0 ? x0 + L0 : (x0 >= L0 ? x0 - L0 : x0);
    x1 = x1 < 0 ? x1 + L1 : (x1 >= L1 ? x1 - L1 : x1);
    x2 = x2 < 0 ? x2 + L2 : (x2 >= L2 ? x2 - L2 : x2);
    return ((int64_t(x0) * L1 + x1) * L2 + x2) * rowsz;
  };
  // sites 4 q4 .. 4 q4 + 3 of a segment's row image (zeros past the segment's nv sites) with the neighbour site on either side:
  // for a whole-row segment (L3 = 32) the row's own ends, held by the lanes q4 = 7 / 0; otherwise `hal`, which the lanes
  // q4 = 0 (left neighbour) and q4 = 7 (right neighbour) loaded
  auto put_row = [&](unsigned char *img, int q4, float4 v, float hal, int nv) {
    if (4 * q4 >= nv) v = float4{0.f, 0.f, 0.f, 0.f};
    f16x4 hi, lo;
    split4(v, hi, lo);
    *reinterpret_cast<f16x4 *>(img + (8 + 4 * q4) * 2) = hi;
    *reinterpret_cast<f16x4 *>(img + LO + (8 + 4 * q4) * 2) = lo;
    if constexpr (!SEGM) {
      if (q4 == 0) {
        *reinterpret_cast<_Float16 *>(img + 40 * 2) = hi[0];
        *reinterpret_cast<_Float16 *>(img + LO + 40 * 2) = lo[0];
      }
      if (q4 == 7) {
        *reinterpret_cast<_Float16 *>(img + 7 * 2) = hi[3];
        *reinterpret_cast<_Float16 *>(img + LO + 7 * 2) = lo[3];
      }
    } else if (q4 == 0 || q4 == 7) {
      const _Float16 h = static_cast<_Float16>(hal), l = static_cast<_Float16>(hal - static_cast<float>(h));
      const int at = q4 == 0 ? 7 : 8 + nv;
      *reinterpret_cast<_Float16 *>(img + at * 2) = h;
      *reinterpret_cast<_Float16 *>(img + LO + at * 2) = l;
    }
  };
  // A plane / a gz row travel in two steps, global -> registers and registers -> LDS images, so that the stager can have the
  // loads of step t + 1 in flight while the multiplying waves are still on step t (one round trip to memory per lattice
  // row, taken in line, was most of the kernel's time).
  constexpr int NPV = CIN == 8 ? 9 : 2;         // float4 per lane of a plane
  constexpr int NGV = (16 * MT * 8 + 63) / 64;  // ... of a gz row
  struct Staged { float4 pv[NPV]; float ph[NPV]; float4 gv[NGV]; };
  int seg0 = 0, nv = 32;                        // the open column's segment: first site, sites (32 or 16)
  auto load_plane = [&](Staged &S, int b, int x0, int x1, int lx2) {      // logical plane lx2 (-1 .. L2) of column (b, x0, x1)
    const float *src = A.in + int64_t(b) * CIN * A.V;
    // (straight-line loads: with a load under a lane-dependent condition the compiler waits for each row's data before it
    //  asks for the next row's, and the staging wave lives on having all of a step's loads in flight together)
    int xs = (lane & 7) == 0 ? seg0 - 1 : seg0 + nv;      // the neighbour site the lanes q4 = 0 / 7 bring (any valid site for the others)
    xs = xs < 0 ? xs + L3 : (xs >= L3 ? xs - L3 : xs);
    const int q4c = 4 * (lane & 7) < nv ? (lane & 7) : 0; // (a partial segment's idle lanes re-read its first quad)
    auto one = [&](const float *rowp, int q4, float4 &v, float &hal) {
      if constexpr (SEGM) {
        v = *reinterpret_cast<const float4 *>(rowp + seg0 + 4 * q4c);
        hal = rowp[xs];
      } else {
        v = *reinterpret_cast<const float4 *>(rowp + 4 * q4);
        hal = 0.f;
      }
    };
    if (CIN == 8) {
      const int ci = lane >> 3, q4 = lane & 7;
#pragma unroll
      for (int rs = 0; rs < 9; ++rs)
        one(src + int64_t(ci) * A.V + site_row(x0 + rs / 3 - 1, x1 + rs % 3 - 1, lx2), q4, S.pv[rs], S.ph[rs]);
    } else {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int f = lane + 64 * it, rs = f < 72 ? f >> 3 : 8, q4 = f & 7;      // (lanes past the 72 quads re-read row 8)
        one(src + site_row(x0 + rs / 3 - 1, x1 + rs % 3 - 1, lx2), q4, S.pv[it], S.ph[it]);
      }
    }
  };
  auto commit_plane = [&](const Staged &S, int lx2) {            // -> ring slot of logical plane lx2
    unsigned char *pl = ring + ((lx2 + NSLOT) & (NSLOT - 1)) * PLANE;
    if (CIN == 8) {
      const int ci = lane >> 3, q4 = lane & 7;
#pragma unroll
      for (int rs = 0; rs < 9; ++rs) put_row(pl + (rs * CIN + ci) * RS, q4, S.pv[rs], S.ph[rs], nv);
    } else {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int f = lane + 64 * it, rs = f >> 3, q4 = f & 7;
        if (f < 72) put_row(pl + rs * RS, q4, S.pv[it], S.ph[it], nv);
      }
    }
  };
  // gz full (B, cout, V): 8 float4 per (channel, row); pair-compact (B, cout, V/2): 4 float4 = the row's 16 active sites,
  // written with zeros between them (the active site of pair p of row (x0, x1, x2) is 2p + ((parity + x0 + x1 + x2) & 1))
  auto load_gz = [&](Staged &S, int b, int x0, int x1, int x2) {
    if constexpr (GZC) {
      const float *src = A.gz + (int64_t(b) * A.cout * A.V + site_row(x0, x1, x2) + seg0) / 2;
#pragma unroll
      for (int it = 0; it < (NGV + 1) / 2; ++it) {
        const int f = lane + 64 * it, co = f >> 2, q4 = f & 3;
        const int coc = co < A.cout ? co : A.cout - 1, qc = 8 * q4 < nv ? q4 : 0;      // (straight-line: idle lanes re-read valid data)
        S.gv[it] = *reinterpret_cast<const float4 *>(src + int64_t(coc) * (A.V / 2) + 4 * qc);
      }
      return;
    }
    const float *src = A.gz + int64_t(b) * A.cout * A.V + site_row(x0, x1, x2) + seg0;
#pragma unroll
    for (int it = 0; it < NGV; ++it) {
      const int f = lane + 64 * it, co = f >> 3, q4 = f & 7;
      const int coc = co < A.cout ? co : A.cout - 1, qc = 4 * q4 < nv ? q4 : 0;
      S.gv[it] = *reinterpret_cast<const float4 *>(src + int64_t(coc) * A.V + 4 * qc);
    }
  };
  auto commit_gz = [&](const Staged &S, int buf, int off) {
    unsigned char *gb = gbuf + buf * 48 * GS;
    if constexpr (GZC) {
#pragma unroll
      for (int it = 0; it < (NGV + 1) / 2; ++it) {
        const int f = lane + 64 * it, co = f >> 2, q4 = f & 3;
        if (co < A.cout) {
          float4 s = 8 * q4 < nv ? S.gv[it] : float4{0.f, 0.f, 0.f, 0.f};
          s.x *= scale; s.y *= scale; s.z *= scale; s.w *= scale;
          f16x4 hi, lo;
          split4(s, hi, lo);
          const _Float16 z = static_cast<_Float16>(0.f);
          const f16x8 h8 = off ? f16x8{z, hi[0], z, hi[1], z, hi[2], z, hi[3]} : f16x8{hi[0], z, hi[1], z, hi[2], z, hi[3], z};
          const f16x8 l8 = off ? f16x8{z, lo[0], z, lo[1], z, lo[2], z, lo[3]} : f16x8{lo[0], z, lo[1], z, lo[2], z, lo[3], z};
          *reinterpret_cast<f16x8 *>(gb + co * GS + 16 * q4) = h8;
          *reinterpret_cast<f16x8 *>(gb + co * GS + GLO + 16 * q4) = l8;
        }
      }
      return;
    }
#pragma unroll
    for (int it = 0; it < NGV; ++it) {
      const int f = lane + 64 * it, co = f >> 3, q4 = f & 7;
      if (co < A.cout) {
        float4 s = 4 * q4 < nv ? S.gv[it] : float4{0.f, 0.f, 0.f, 0.f};
        s.x *= scale; s.y *= scale; s.z *= scale; s.w *= scale;
        f16x4 hi, lo;
        split4(s, hi, lo);
        *reinterpret_cast<f16x4 *>(gb + co * GS + 8 * q4) = hi;
        *reinterpret_cast<f16x4 *>(gb + co * GS + GLO + 8 * q4) = lo;
      }
    }
  };

  // ---- one lattice row: 3 x MT x (2 groups x 3 taps) MFMAs per multiplying wave
  auto multiply = [&](int t) {
    const unsigned char *gb = gbuf + (t & 1) * 48 * GS;
    f16x8 ah[MT], al[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const unsigned char *pa = gb + (16 * mi + n) * GS + 16 * kq;
      ah[mi] = *reinterpret_cast<const f16x8 *>(pa);
      al[mi] = *reinterpret_cast<const f16x8 *>(pa + GLO);
    }
    int slot[3];
#pragma unroll
    for (int j2 = 0; j2 < 3; ++j2) slot[j2] = ((t + j2 - 1 + NSLOT) & (NSLOT - 1)) * PLANE;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (!gact[q]) continue;                   // (wave-uniform)
      const unsigned char *pb = ring + (bj2[q] == 0 ? slot[0] : (bj2[q] == 1 ? slot[1] : slot[2])) + boff[q];
      unsigned wh[6], wl[6];
      wh[0] = *reinterpret_cast<const unsigned *>(pb);
      wl[0] = *reinterpret_cast<const unsigned *>(pb + LO);
      const u32x4 mh = *reinterpret_cast<const u32x4 *>(pb + 4), ml = *reinterpret_cast<const u32x4 *>(pb + LO + 4);
      wh[5] = *reinterpret_cast<const unsigned *>(pb + 20);
      wl[5] = *reinterpret_cast<const unsigned *>(pb + LO + 20);
#pragma unroll
      for (int r = 0; r < 4; ++r) { wh[1 + r] = mh[r]; wl[1 + r] = ml[r]; }
#pragma unroll
      for (int s = 0; s < 3; ++s) {             // tap j3 = s: sites shifted by s - 1
        u32x4 fh, fl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (s == 1) { fh[r] = wh[1 + r]; fl[r] = wl[1 + r]; }
          else if (s == 0) { fh[r] = __builtin_amdgcn_alignbyte(wh[1 + r], wh[r], 2); fl[r] = __builtin_amdgcn_alignbyte(wl[1 + r], wl[r], 2); }
          else { fh[r] = __builtin_amdgcn_alignbyte(wh[2 + r], wh[1 + r], 2); fl[r] = __builtin_amdgcn_alignbyte(wl[2 + r], wl[1 + r], 2); }
        }
        if (!bvalid[q]) {                       // columns past the 27 kernel rows: zero, or the bias column's ones (tap 1 only)
          const unsigned one = (bones[q] && s == 1) ? 0x3c003c00u : 0u;
          fh = u32x4{one, one, one, one};
          fl = u32x4{0u, 0u, 0u, 0u};
        }
        const f16x8 bh = __builtin_bit_cast(f16x8, fh), bl = __builtin_bit_cast(f16x8, fl);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          acc[mi][q][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bh, acc[mi][q][s], 0, 0, 0);
          acc[mi][q][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bl, acc[mi][q][s], 0, 0, 0);
          acc[mi][q][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mi], bh, acc[mi][q][s], 0, 0, 0);
        }
      }
    }
  };

  __syncthreads();
  auto column_of = [&](int64_t col, int &b, int &x0, int &x1) {      // (the segment is the fastest index)
    const int hs = int(col % NSEG);
    col /= NSEG;
    seg0 = 32 * hs;
    nv = L3 - seg0 < 32 ? L3 - seg0 : 32;
    b = int(col / (int64_t(L0) * L1));
    const int rem = int(col - int64_t(b) * L0 * L1);
    x0 = rem / L1;
    x1 = rem - x0 * L1;
  };
  // Two loops with the same barriers, one per role: in one loop the stager's registers would be live across the other waves'
  // MFMA code (the register allocator does not know that `wave` never changes), 98 of them spilled.
  if (wave == 7) {
    Staged S;
    for (int64_t col = blockIdx.x; col < A.ncolumns; col += gridDim.x) {
      int b, x0, x1;
      column_of(col, b, x0, x1);
      // planes -1, 0, 1 and gz's first row; plane 2 and the second row set off
#pragma unroll 1
      for (int lx2 = -1; lx2 <= 1; ++lx2) {
        load_plane(S, b, x0, x1, lx2);
        commit_plane(S, lx2);
      }
      load_gz(S, b, x0, x1, 0);
      commit_gz(S, 0, (A.parity + x0 + x1) & 1);
      if (L2 > 1) {
        load_plane(S, b, x0, x1, 2);
        load_gz(S, b, x0, x1, 1);
      }
      lds_barrier();
      for (int t = 0; t < L2; ++t) {
        if (t + 1 < L2) {                         // what was loaded a step ago: plane t + 2, gz row t + 1
          commit_plane(S, t + 2);
          commit_gz(S, (t + 1) & 1, (A.parity + x0 + x1 + t + 1) & 1);
        }
        if (t + 2 < L2) {                         // lands while the others multiply row t
          load_plane(S, b, x0, x1, t + 3);
          load_gz(S, b, x0, x1, t + 2);
        }
        lds_barrier();                            // (LDS only: the loads stay in flight across it)
      }
    }
    return;
  }
  for (int64_t col = blockIdx.x; col < A.ncolumns; col += gridDim.x) {
    lds_barrier();
    for (int t = 0; t < L2; ++t) {
      multiply(t);
      lds_barrier();
    }
  }

  // ---- this workgroup's partial matrix: D[m][n] of tile (mi, group, tap): lane (n, g4) holds rows 4 g4 .. 4 g4 + 3
  {
    float *out = A.partial + int64_t(blockIdx.x) * 48 * A.ncols;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (!gact[q]) continue;
      const int g = 2 * wave + q;
      const int c = CIN == 8 ? 2 * g + (n >> 3) : 16 * g + n;
      const int ci = CIN == 8 ? (n & 7) : 0;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        int colx = -1;
        if (c < 27) colx = (c * 3 + s) * CIN + ci;
        else if (c == 27 && ci == 0 && s == 1) colx = 81 * CIN;
        if (colx < 0) continue;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) out[int64_t(16 * mi + 4 * kq + r) * A.ncols + colx] = acc[mi][q][s][r];
      }
    }
  }
}

// pair-compact (rows, V/2) -> full (rows, V): the value of pair p of a lattice row goes to its active site (coordinate sum
// == parity mod 2), the other site of the pair gets 0.  One pass (the host-side version was five).
template <typename T, typename T2>
__global__ __launch_bounds__(256) void expand_pairs_kernel(const T *__restrict__ src, T2 *__restrict__ dst, int64_t rows, int64_t Vh,
                                                           int L0, int L1, int L2, int HP, int parity) {
  const int64_t total = rows * Vh;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += int64_t(gridDim.x) * 256) {
    const int64_t pr = i % Vh;                 // pair index inside the sample's lattice
    const int64_t row = pr / HP;               // lattice row (x0, x1, x2)
    const int x2 = int(row % L2), x1 = int((row / L2) % L1), x0 = int(row / (int64_t(L2) * L1));
    const int first = ((x0 + x1 + x2) & 1) == parity;      // the pair's even site is the active one
    const T v = src[i];
    T2 o;
    o.x = first ? v : T(0);
    o.y = first ? T(0) : v;
    dst[i] = o;
  }
  (void)L0;
}

// fp32 channel planes (B, C, V) -> G = ceil(C / 8) pair tensors (G, B, V, 16 halfs) of 8 channels each (zeros past C), every
// value multiplied by the power of two of *absmax: the cotangent as the split-fp16 conv kernels read it (nf_conv_dgrad_split16).
__global__ __launch_bounds__(256) void planes_to_pairs_kernel(const float *__restrict__ src, unsigned char *__restrict__ dst, int64_t B,
                                                              int C, int64_t V, int L3, const unsigned *absmax, int compact, int parity,
                                                              int L1, int L2) {
  const float scale = pow2_scale_for(absmax);
  const int G = (C + 7) >> 3;
  const int64_t total = int64_t(G) * B * V;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += int64_t(gridDim.x) * 256) {
    const int64_t site = i % V;
    const int64_t gb = i / V;                  // g * B + b
    const int b = int(gb % B), g = int(gb / B);
    const int64_t row = site / L3;
    const int x3 = int(site - row * L3);
    bool live = true;                          // compact input: only the active sites carry a value
    if (compact) {
      const int x2 = int(row % L2), x1 = int((row / L2) % L1), x0 = int(row / (int64_t(L2) * L1));
      live = ((x0 + x1 + x2 + x3) & 1) == parity;
    }
    f16x8 hi, lo;
    const int64_t per = compact ? V / 2 : V, at = compact ? (site >> 1) : site;
    float raw[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {             // (straight-line loads: channels past C re-read the last one)
      const int ch = 8 * g + c < C ? 8 * g + c : C - 1;
      raw[c] = src[(int64_t(b) * C + ch) * per + at];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float v = (8 * g + c < C && live) ? raw[c] * scale : 0.f;
      const _Float16 h = static_cast<_Float16>(v);
      hi[c] = h;
      lo[c] = static_cast<_Float16>(v - static_cast<float>(h));
    }
    unsigned char *d = dst + (gb * V + row * L3) * 32 + pair_row_offset(x3, L3);
    *reinterpret_cast<f16x8 *>(d) = hi;
    *reinterpret_cast<f16x8 *>(d + L3 * 16) = lo;
  }
}

// gw[o][c] += (sum over the workgroups' partials, in order) / scale
__global__ __launch_bounds__(256) void wgrad16_reduce_kernel(const float *__restrict__ partial, float *__restrict__ gw, int nparts, int ncols,
                                                             int nused, int rows, const unsigned *absmax) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * nused) return;
  const int o = idx / nused, c = idx - o * nused;
  double s = 0.0;
  for (int p = 0; p < nparts; ++p) s += double(partial[(int64_t(p) * 48 + o) * ncols + c]);
  gw[int64_t(o) * ncols + c] += float(s / double(gz_scale(absmax)));
}

}  // namespace wg
}  // namespace nf

using namespace nf;

extern "C" int nf_conv_wgrad_cols(int cin, int ntaps);

extern "C" int nf_conv_wgrad_split16_supported(const int32_t *lattice, const int32_t *ksize, int cin, int cout) {
  if (!lattice || !ksize) return 0;
  for (int mu = 0; mu < 4; ++mu)
    if (ksize[mu] != 3 || lattice[mu] < 1) return 0;
  if (lattice[3] < 32 || (lattice[3] & 15)) return 0;       // segments of 32 sites, the last one 32 or 16
  if (cin != 1 && cin != 8) return 0;
  return cout >= 1 && cout <= 48;
}

static int wgrad16_grid(int64_t ncolumns) { return int(ncolumns < 512 ? ncolumns : 512); }

extern "C" size_t nf_conv_wgrad_split16_workspace(int64_t B, const int32_t *lattice, int cin) {
  if (!lattice) return 0;
  const int64_t ncolumns = B * lattice[0] * lattice[1] * ((lattice[3] + 31) / 32);
  return 256 + size_t(wgrad16_grid(ncolumns)) * 48 * size_t(nf_conv_wgrad_cols(cin, 81)) * sizeof(float);
}

extern "C" int nf_conv_wgrad_split16(const void *in, const void *gz, void *gw, int64_t B, const int32_t *lattice,
                                     const int32_t *ksize, int cin, int cout, const void *absmax_bits, int compact_parity,
                                     void *workspace, size_t workspace_bytes, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  NF_REQUIRE(in && gz && gw && lattice && ksize, "nf_conv_wgrad_split16: NULL pointer");
  NF_REQUIRE(nf_conv_wgrad_split16_supported(lattice, ksize, cin, cout),
             "nf_conv_wgrad_split16: needs a 4-D lattice with 32 + 16 n sites on the fastest axis, 3^4 kernels, cin 1 or 8, cout <= 48");
  NF_REQUIRE(B >= 0 && B < (int64_t(1) << 24), "nf_conv_wgrad_split16: bad batch");
  if (B == 0) return NF_OK;
  wg::Args A{};
  A.V = 1;
  for (int mu = 0; mu < 4; ++mu) { A.L[mu] = lattice[mu]; A.V *= lattice[mu]; }
  NF_REQUIRE(A.V < (int64_t(1) << 31), "nf_conv_wgrad_split16: lattice volume must be < 2^31");
  A.in = static_cast<const float *>(in);
  A.gz = static_cast<const float *>(gz);
  A.B = int(B); A.cin = cin; A.cout = cout;
  A.gz_compact = compact_parity >= 0 ? 1 : 0;
  A.parity = compact_parity & 1;
  A.ncols = nf_conv_wgrad_cols(cin, 81);
  A.ncolumns = B * A.L[0] * A.L[1] * ((A.L[3] + 31) / 32);
  const int grid = wgrad16_grid(A.ncolumns);
  const size_t need = nf_conv_wgrad_split16_workspace(B, lattice, cin);
  if (!workspace || workspace_bytes < need) {
    set_error("nf_conv_wgrad_split16: workspace %zu B < %zu B needed", workspace_bytes, need);
    return NF_EWORKSPACE;
  }
  NF_REQUIRE(absmax_bits != nullptr, "nf_conv_wgrad_split16: absmax_bits is NULL (nf_absmax_bits of gz)");
  const unsigned *absmax = static_cast<const unsigned *>(absmax_bits);
  A.absmax = absmax;
  A.partial = reinterpret_cast<float *>(static_cast<unsigned char *>(workspace) + 256);
  NF_REQUIRE(hipMemsetAsync(workspace, 0, need, s) == hipSuccess, "nf_conv_wgrad_split16: hipMemsetAsync failed");
  int rc;
  const int MT = (cout + 15) >> 4;
  const size_t lds = size_t(wg::NSLOT) * 9 * cin * wg::RS + 2 * 48 * wg::GS;
#define NF_W16S(MTV, CINV, GZ, SG)                                                                                     \
  {                                                                                                                    \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&wg::wgrad16_kernel<MTV, CINV, GZ, SG>),                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));                                  \
    hipLaunchKernelGGL((wg::wgrad16_kernel<MTV, CINV, GZ, SG>), dim3(grid), dim3(wg::kThreads), lds, s, A);            \
  }
#define NF_W16G(MTV, CINV, GZ)                                                                                         \
  {                                                                                                                    \
    if (A.L[3] != 32) NF_W16S(MTV, CINV, GZ, true) else NF_W16S(MTV, CINV, GZ, false)                                  \
  }
#define NF_W16(MTV, CINV)                                                                                              \
  {                                                                                                                    \
    if (A.gz_compact) NF_W16G(MTV, CINV, true) else NF_W16G(MTV, CINV, false)                                          \
  }
  if (cin == 8) {
    if (MT == 1) NF_W16(1, 8) else if (MT == 2) NF_W16(2, 8) else NF_W16(3, 8)
  } else {
    if (MT == 1) NF_W16(1, 1) else if (MT == 2) NF_W16(2, 1) else NF_W16(3, 1)
  }
#undef NF_W16
#undef NF_W16G
#undef NF_W16S
  rc = check_launch("wgrad16 kernel");
  if (rc) return rc;
  const int nused = 81 * cin + 1;
  const int total = cout * nused;
  hipLaunchKernelGGL(wg::wgrad16_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, A.partial,
                     static_cast<float *>(gw), grid, A.ncols, nused, cout, absmax);
  return check_launch("wgrad16 reduce kernel");
}

extern "C" int nf_expand_pairs(const void *compact, void *full, int64_t rows, const int32_t *lattice, int parity, int dtype,
                               void *stream) {
  NF_REQUIRE(compact && full && lattice, "nf_expand_pairs: NULL pointer");
  NF_REQUIRE(rows >= 0 && (parity == 0 || parity == 1), "nf_expand_pairs: bad arguments");
  NF_REQUIRE(lattice[3] >= 2 && (lattice[3] & 1) == 0, "nf_expand_pairs: the fastest axis must be even");
  const int64_t Vh = int64_t(lattice[0]) * lattice[1] * lattice[2] * (lattice[3] / 2);
  if (rows == 0 || Vh == 0) return NF_OK;
  const int64_t want = (rows * Vh + 255) / 256;
  const unsigned grid = unsigned(want < 16384 ? want : 16384);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == NF_F32)
    hipLaunchKernelGGL((wg::expand_pairs_kernel<float, float2>), dim3(grid), dim3(256), 0, s, static_cast<const float *>(compact),
                       static_cast<float2 *>(full), rows, Vh, lattice[0], lattice[1], lattice[2], lattice[3] / 2, parity);
  else if (dtype == NF_F64)
    hipLaunchKernelGGL((wg::expand_pairs_kernel<double, double2>), dim3(grid), dim3(256), 0, s, static_cast<const double *>(compact),
                       static_cast<double2 *>(full), rows, Vh, lattice[0], lattice[1], lattice[2], lattice[3] / 2, parity);
  else {
    set_error("nf_expand_pairs: unsupported dtype %d", dtype);
    return NF_EINVAL;
  }
  return check_launch("expand pairs kernel");
}

// max |x| of an fp32 tensor as the bits of a float in device memory (4 bytes): the measure the split-fp16 training kernels
// scale a cotangent by (one pass, shared by the weight gradient and the input gradient of a layer).
extern "C" int nf_absmax_bits(const void *x, int64_t n, void *bits, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  NF_REQUIRE(x && bits && n >= 0, "nf_absmax_bits: bad arguments");
  NF_REQUIRE(hipMemsetAsync(bits, 0, 4, s) == hipSuccess, "nf_absmax_bits: hipMemsetAsync failed");
  if (n == 0) return NF_OK;
  hipLaunchKernelGGL(wg::absmax_kernel, dim3(2048), dim3(256), 0, s, static_cast<const float *>(x), n, static_cast<unsigned *>(bits));
  return check_launch("absmax kernel");
}

// fp32 channel planes (B, C, V) as the split-fp16 kernels read them: ceil(C / 8) pair tensors (G, B, V, 16 halfs), multiplied
// by the power of two that brings the maximum in *absmax_bits to [2^12, 2^13) (absmax_bits NULL: as they are -- activations).
extern "C" int nf_planes_to_split16(const void *gz, void *out16, const void *absmax_bits, int64_t B, int C,
                                    const int32_t *lattice, int compact_parity, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  NF_REQUIRE(gz && out16 && lattice, "nf_planes_to_split16: NULL pointer");
  NF_REQUIRE(B >= 0 && C >= 1, "nf_planes_to_split16: bad sizes");
  NF_REQUIRE(lattice[3] >= 2 && (lattice[3] & 1) == 0, "nf_planes_to_split16: the fastest axis must be even");
  const int64_t V = int64_t(lattice[0]) * lattice[1] * lattice[2] * lattice[3];
  if (B == 0 || V == 0) return NF_OK;
  const int64_t total = int64_t((C + 7) / 8) * B * V;
  const int64_t want = (total + 255) / 256;
  hipLaunchKernelGGL(wg::planes_to_pairs_kernel, dim3(unsigned(want < 16384 ? want : 16384)), dim3(256), 0, s,
                     static_cast<const float *>(gz), static_cast<unsigned char *>(out16), B, C, V, lattice[3],
                     static_cast<const unsigned *>(absmax_bits), compact_parity >= 0 ? 1 : 0, compact_parity & 1, lattice[1],
                     lattice[2]);
  return check_launch("planes to pairs kernel");
}
