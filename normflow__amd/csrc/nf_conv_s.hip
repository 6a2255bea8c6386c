// nf_conv_s.hip -- K5s: ONE kernel per coupling layer for SMALL 3-D lattices (fastest axis 16 sites, e.g. BASELINE config 3's
// 16^3): the parameter net ConvAct 1 -> h -> h -> 3m-2 (3^3 circular kernels, h <= 8) AND the RQ-spline coupling of a sample
// run inside one workgroup, with the sample resident in the CU's 160 KB of LDS -- the hidden activations never touch HBM and
// there is no halo exchange at all: the periodic wrap is an address computation inside LDS.
// Reference chain replaced (one launch per layer instead of ~160 eager ops): src/nn/scalar/modules.py:120-145 (ConvAct),
// src/nn/scalar/couplings_.py:178-262 (atomic_forward / backward, make_spline), src/lib/spline/spline.py:154-287.
//
// Arithmetic: every fp32 product as THREE fp16 matrix-core products (a_hi w_hi + a_lo w_hi + a_hi w_lo, fp32 accumulate,
// v_mfma_f32_16x16x32_f16), as in nf_conv_c / g / h.hip: the input field is split x = x_hi + x_lo (|x| < 6.5e4, beyond: NaN,
// never silently wrong), hidden activations are tanh / logistic outputs (|h| <= 1), weights are scaled by 2^10 and split on
// the host (checked finite and in range there).
//
// Shape of the computation.  A persistent workgroup of 4 waves takes samples one at a time and marches the slowest axis:
//   step t:  A  H1[t+2] = act(conv1(x))            one 16-site lattice row per MFMA tile (weights: the A operand, 16 = 8 + 8 pad)
//            -- barrier --
//            B  H2[t+1] = act(conv2(H1[t .. t+2]))  two rows = 16 site PAIRS per tile, two-site columns (nf_conv_g.hip's trick):
//                                                  9 kernel rows x 3 products = 27 MFMAs per tile
//            -- barrier --
//            C  logits[t] = conv3(H2[t-1 .. t+1]) at the 128 active sites of the plane: 7 K-slices of 4 taps x 8 channels,
//               3 column tiles, 63 MFMAs per 16-site tile; then the RQ-spline map of those sites from a per-wave logit scratch
// H1 / H2 live in rings of 4 planes of fp16 (hi, lo) pairs (16 B per site and half); the planes beyond the ends of the periodic
// axis (-2, -1, L0, L0+1) are computed again rather than kept (+25 % of the cheap first layer, +12 % of the second).  Every wave
// holds ALL weights in registers (62 fragments) and owns whole tiles: no cross-wave sums, two barriers per plane.
// log|J| of a sample is summed inside its workgroup (fixed order: bitwise reproducible), no second kernel.
//
// The same kernel serves (i) the AFFINE coupling (KIND 1: the net ends in 2 channels (t, s); y = t + x e^{-|s|}, log|J| -= |s|,
// src/nn/scalar/couplings_.py:123-139; one column tile instead of three) and (ii) 2-D lattices (L1, 16) -- BASELINE config 2's
// 16 x 16 --: `flat` = one plane, 3^2 kernels embedded as the middle plane of 3^3 ones (zero weights elsewhere), the ring
// slots of the absent neighbour planes zeroed once, one step A, B, C per sample.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "nf_conv_core.h"

namespace nf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace s3 {
constexpr int LX = 16;            // sites of the fastest axis
constexpr int RING = 4;           // planes kept of each hidden layer
constexpr int C = 46, M = 16;     // most logit channels / knots (3 column tiles)
constexpr int PTS = 68;           // floats per channel row of a wave's logit scratch: 2 planes x 32 sites + 4 (bank spread, 16-B rows)
constexpr int PTW = 48 * PTS * 4; // bytes of a wave's logit scratch
constexpr float kInvWScale = 1.0f / 1024.0f;       // the host packs the weights scaled by 2^10 (normflow__amd/_hip.py: SPLIT16_WEIGHT_SCALE)
constexpr int PX = LX + 2;        // the input field is kept with a one-site periodic halo on every axis: taps are plain offsets
__host__ __device__ constexpr size_t x_bytes(int L0, int L1) { return ((size_t(L0 + 2) * (L1 + 2) * PX * 2 + 15) / 16) * 16; }   // one half array (hi or lo)
__host__ __device__ constexpr size_t lds_bytes(int L0, int L1) {
  return 2 * x_bytes(L0, L1) + size_t(2 * RING) * L1 * LX * 32 + 4 * PTW + 64;
}
}  // namespace s3

struct SmallArgs {
  const float *xf, *xa;
  float *y;
  const float *log0;
  float *logj;
  const f16x8 *w1, *w2, *w3;
  const float *b1, *b2, *b3;
  int64_t B;
  int L0, L1, parity, cout, act1, act2;
  int flat;                 // 2-D lattice: a single plane, no marching
  RqsParams P;
};

template <bool INV, int KIND>      // KIND 0: RQ-spline coupling, 1: affine coupling
__global__ __launch_bounds__(256, 1) void conv_small3d_kernel(SmallArgs A) {
  using namespace s3;
  extern __shared__ __align__(16) unsigned char smem_s[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int L0 = A.L0, L1 = A.L1;
  const int V = L0 * L1 * LX, PSITES = L1 * LX;
  const int PB = PSITES * 32, HL = PSITES * 16;     // bytes of a plane of pairs / offset of its lo half
  const int PY = L1 + 2;
  const int XB = int(x_bytes(L0, L1));
  unsigned char *Xh = smem_s, *Xl = smem_s + XB;      // (L0+2, L1+2, 18) halfs each: x_hi, x_lo with the periodic halo
  unsigned char *H1 = smem_s + 2 * XB;
  unsigned char *H2 = H1 + RING * PB;
  float *pt = reinterpret_cast<float *>(H2 + RING * PB) + wave * (48 * PTS);
  double *red = reinterpret_cast<double *>(H2 + RING * PB + 4 * PTW);

  // ---- all weights, for the whole launch
  const f16x8 a1h = A.w1[lane], a1l = A.w1[64 + lane];
  f16x8 a2h[9], a2l[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    a2h[r] = A.w2[(2 * r) * 64 + lane];
    a2l[r] = A.w2[(2 * r + 1) * 64 + lane];
  }
  f16x8 b3h[7][3], b3l[7][3];
#pragma unroll
  for (int t = 0; t < (KIND == 1 ? 1 : 3); ++t)
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      b3h[i][t] = A.w3[((t * 7 + i) * 2) * 64 + lane];
      b3l[i][t] = A.w3[((t * 7 + i) * 2 + 1) * 64 + lane];
    }
  // Both hidden activations are one branch-free form: act(v) = alpha * s(beta v) + gamma with the logistic function
  // s(u) = 1 / (1 + 2^(-u log2 e))  (tanh: 2 s(2v) - 1; logistic: s(v)), evaluated on the accumulator directly:
  // 2^(c1 * acc + c0[channel]) with the weight scale, the bias and -beta log2 e folded into c1, c0 (one FMA, v_exp_f32,
  // v_rcp_f32, one FMA; absolute error ~1e-7 on an O(1) activation).
  const float be1 = A.act1 == kActTanh ? 2.f : 1.f, be2 = A.act2 == kActTanh ? 2.f : 1.f;
  const float al1 = be1, ga1 = 1.f - be1, al2 = be2, ga2 = 1.f - be2;
  const float c11 = -be1 * Num<float>::kLog2e * kInvWScale, c12 = -be2 * Num<float>::kLog2e * kInvWScale;
  float c01[4], c02[4], b3v[3];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    c01[r] = -be1 * Num<float>::kLog2e * ((A.b1 && g < 2) ? A.b1[4 * g + r] : 0.f);   // stage A: lane holds channels 4g .. 4g+3 of one site (g < 2)
    c02[r] = -be2 * Num<float>::kLog2e * (A.b2 ? A.b2[4 * (g & 1) + r] : 0.f);        // stage B: channels 4(g&1) .. of site 2q + (g >> 1)
  }
  auto act_of = [](float acc, float c1, float c0, float al, float ga) {
    const float t = Num<float>::exp2(__builtin_fmaf(acc, c1, c0));
    return __builtin_fmaf(al, __builtin_amdgcn_rcpf(1.f + t), ga);
  };
#pragma unroll
  for (int t = 0; t < 3; ++t) b3v[t] = (A.b3 && 16 * t + n < A.cout) ? A.b3[16 * t + n] : 0.f;

  // ---- per-lane tap tables: stage A's K index 8g + i (27 taps of the 3^3 kernel, padded to 32), stage C's 4i + g (28)
  int tapA[8], tapC[7];            // tapA: byte offset of the tap in the haloed field; tapC: packed (dz + 1) | (dy + 1) << 2 | (dx + 1) << 4
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = 8 * g + i;
    tapA[i] = k < 27 ? (((k / 9 - 1) * PY + ((k / 3) % 3 - 1)) * PX + (k % 3 - 1)) * 2 : 0;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int k = 4 * i + g;
    tapC[i] = k < 27 ? (k / 9) | (((k / 3) % 3) << 2) | ((k % 3) << 4) : (1 | (1 << 2) | (1 << 4));
  }
  constexpr int NT = KIND == 1 ? 1 : 3;                         // column tiles of the last layer (affine: t and s only)
  const int ntw = (L1 / 2 - wave + 3) / 4;                      // tiles (row pairs) of a plane this wave owns: T = wave, wave + 4, ...
  auto wrap1 = [](int v, int L) { return v < 0 ? v + L : (v >= L ? v - L : v); };
  auto ring = [&](int p) { return ((p + 8) & (RING - 1)) * PB; };

  // ================================================================= stage A: H1[p] = act(conv1(x)), one row per tile
  auto stageA = [&](int p) {
    int pz = p % L0;
    pz = pz < 0 ? pz + L0 : pz;
    unsigned char *dst = H1 + ring(p);
    for (int T = wave; T < L1 / 2; T += 4) {
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int yrow = 2 * T + rr;
        f16x8 xh, xl;
        const int ctr = (((pz + 1) * PY + yrow + 1) * PX + n + 1) * 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          xh[i] = *reinterpret_cast<const _Float16 *>(Xh + ctr + tapA[i]);
          xl[i] = *reinterpret_cast<const _Float16 *>(Xl + ctr + tapA[i]);
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1l, xh, acc, 0, 0, 0);
        if (g < 2) {                // D[channel 4g + r][site n]
          f16x4 hi, lo;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = act_of(acc[r], c11, c01[r], al1, ga1);
            hi[r] = static_cast<_Float16>(v);
            lo[r] = static_cast<_Float16>(v - static_cast<float>(hi[r]));
          }
          unsigned char *d = dst + (yrow * LX + n) * 16 + g * 8;
          *reinterpret_cast<f16x4 *>(d) = hi;
          *reinterpret_cast<f16x4 *>(d + HL) = lo;
        }
      }
    }
  };

  // ================================================================= stage B: H2[p] = act(conv2(H1[p-1 .. p+1])), 16 site pairs per tile
  auto stageB = [&](int p) {
    unsigned char *dst = H2 + ring(p);
    const int rr = n >> 3, q = n & 7;
    const int xs = (2 * q + g - 1) & (LX - 1);         // k-group g = tap g of the pair (sites 2q-1 .. 2q+2)
    for (int T = wave; T < L1 / 2; T += 4) {
      const int yrow = 2 * T + rr;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j0 = 0; j0 < 3; ++j0) {
        const unsigned char *pl = H1 + ring(p + j0 - 1);
#pragma unroll
        for (int j1 = 0; j1 < 3; ++j1) {
          const int yr = wrap1(yrow + j1 - 1, L1);
          const unsigned char *src = pl + (yr * LX + xs) * 16;
          const f16x8 fh = *reinterpret_cast<const f16x8 *>(src);
          const f16x8 fl = *reinterpret_cast<const f16x8 *>(src + HL);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2h[3 * j0 + j1], fh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2h[3 * j0 + j1], fl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2l[3 * j0 + j1], fh, acc, 0, 0, 0);
        }
      }
      // D[(site-in-pair s, channel)][pair n]: this lane holds channels 4(g&1) .. +3 of site 2q + (g >> 1)
      f16x4 hi, lo;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = act_of(acc[r], c12, c02[r], al2, ga2);
        hi[r] = static_cast<_Float16>(v);
        lo[r] = static_cast<_Float16>(v - static_cast<float>(hi[r]));
      }
      unsigned char *d = dst + (yrow * LX + 2 * q + (g >> 1)) * 16 + (g & 1) * 8;
      *reinterpret_cast<f16x4 *>(d) = hi;
      *reinterpret_cast<f16x4 *>(d + HL) = lo;
    }
  };

  // ================================================================= stage C: logits of plane z at its active sites, then the spline
  // The logits of two consecutive planes are collected in the wave's scratch (32 sites each) and mapped together: the
  // spline pass is a long dependent chain, so 64 lanes cost what 32 do.  The field values of those sites are requested at
  // the top of the step (prefetch_x), two barriers ahead of their use.
  double lacc = 0.0;
  float xpre = 0.f;
  int64_t spre = -1;                 // the lane's site of the coming spline pass (-1: none)
  auto spline_due = [&](int z) { return (z & 1) || z == L0 - 1; };
  auto prefetch_x = [&](int z, int64_t sbase) {
    spre = -1;
    if (z < 0 || !spline_due(z)) return;
    const int z0 = (z & 1) ? z - 1 : z;                         // first plane of the pass (a lone last plane: z itself)
    const int u = lane, zz = z0 + (u >> 5), ul = u & 31;
    if (zz <= z && ul < ntw * 16) {
      const int T = wave + 4 * (ul >> 4), m_ = ul & 15;
      const int yrow = 2 * T + (m_ >> 3), qq = m_ & 7;
      const int xsite = 2 * qq + ((A.parity + zz + yrow) & 1);
      spre = sbase + (int64_t(zz) * L1 + yrow) * LX + xsite;
      xpre = A.xa[spre];
    }
  };
  auto stageC = [&](int z) {
    const int rr = n >> 3, q = n & 7;
    int nt = 0;
    const int half = (z & 1) * 32;                              // this plane's half of the scratch rows
    for (int T = wave; T < L1 / 2; T += 4, ++nt) {
      const int yrow = 2 * T + rr;
      const int xa = 2 * q + ((A.parity + z + yrow) & 1);       // the active site of pair q in this row
      f32x4 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int yr = wrap1(yrow + ((tapC[i] >> 2) & 3) - 1, L1);
        const int xr = (xa + ((tapC[i] >> 4) & 3) - 1) & (LX - 1);
        const unsigned char *src = H2 + ring(z + (tapC[i] & 3) - 1) + (yr * LX + xr) * 16;
        const f16x8 fh = *reinterpret_cast<const f16x8 *>(src);
        const f16x8 fl = *reinterpret_cast<const f16x8 *>(src + HL);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, b3h[i][t], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl, b3h[i][t], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, b3l[i][t], acc[t], 0, 0, 0);
        }
      }
      // D[site 4g + r of the tile][channel 16 t + n] -> the wave's logit scratch [channel][site]
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int ch = 16 * t + n;
        if (ch < A.cout) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[t][r] * kInvWScale + b3v[t];
          *reinterpret_cast<f32x4 *>(pt + ch * PTS + half + nt * 16 + 4 * g) = v;
        }
      }
    }
    if (!spline_due(z)) return;
    // the spline map of the wave's own sites of this plane and the one before (lane u: plane u >> 5, site u & 31 of it)
    if (spre >= 0) {
      const int u = (z & 1) ? lane : (lane & 31);               // a lone last plane sits in the first half
      float val, logd;
      if constexpr (KIND == 1) {
        // affine coupling: (t, s) = the net's two channels; s enters as |s| (couplings_.py:123-139)
        const float tt = pt[u], ss = fabsf(pt[PTS + u]);
        val = INV ? (xpre - tt) * __expf(ss) : tt + xpre * __expf(-ss);
        logd = INV ? ss : -ss;
      } else if (A.P.m == M) {
        RegCol<float, C> col;
#pragma unroll
        for (int c = 0; c < C; ++c) col[c] = pt[c * PTS + u];
        rqs_site<float, M, INV>(col, A.P, xpre, val, logd);
      } else if (A.P.m == 8) {          // the other common knots_len: its own unrolled instance (the run-time form below is ~25 % slower)
        RegCol<float, 22> col;
#pragma unroll
        for (int c = 0; c < 22; ++c) col[c] = pt[c * PTS + u];
        rqs_site<float, 8, INV>(col, A.P, xpre, val, logd);
      } else {
        LdsCol<float> col{pt + u, PTS};
        rqs_site<float, 0, INV>(col, A.P, xpre, val, logd);
      }
      A.y[spre] = val;
      A.y[spre ^ 1] = 0.f;          // the frozen site of the pair
      lacc += double(logd);
    }
  };

  if (A.flat) {                       // the ring slots of planes -1 and +1 are read (with zero weights) and never written: zero them once
    for (int i = threadIdx.x * 16; i < 2 * RING * PB; i += 256 * 16) *reinterpret_cast<f32x4 *>(H1 + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int64_t b = blockIdx.x; b < A.B; b += gridDim.x) {
    const int64_t sbase = b * int64_t(V);
    for (int i = threadIdx.x; i < (L0 + 2) * PY * PX; i += 256) {       // the haloed copy: source site = index - 1, wrapped
      const int hz = i / (PY * PX), rem = i - hz * (PY * PX), hy = rem / PX, hx = rem - hy * PX;
      const int sz = wrap1(hz - 1, L0), sy = wrap1(hy - 1, L1), sx = (hx - 1) & (LX - 1);
      const float v = A.xf[sbase + (sz * L1 + sy) * LX + sx];
      const _Float16 hi = static_cast<_Float16>(v);
      reinterpret_cast<_Float16 *>(Xh)[i] = hi;
      reinterpret_cast<_Float16 *>(Xl)[i] = static_cast<_Float16>(v - static_cast<float>(hi));
    }
    lacc = 0.0;
    lds_barrier();
    if (A.flat) {                     // one plane: its neighbours along the absent axis are the zeroed ring slots
      prefetch_x(0, sbase);
      stageA(0);
      lds_barrier();
      stageB(0);
      lds_barrier();
      stageC(0);
    } else
    for (int t = -4; t < L0; ++t) {
      prefetch_x(t, sbase);
      stageA(t + 2);
      lds_barrier();
      if (t >= -2) stageB(t + 1);
      lds_barrier();
      if (t >= 0) stageC(t);
    }
    const double tot = wave_sum(lacc);
    if (lane == 0) red[wave] = tot;
    lds_barrier();                  // also: every wave is done with this sample's Xs / rings
    if (threadIdx.x == 0) A.logj[b] = float((A.log0 ? double(A.log0[b]) : 0.0) + ((red[0] + red[1]) + (red[2] + red[3])));
    lds_barrier();
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same layer with EIGHT waves in two roles, two waves per SIMD (K5s above holds all weights in every wave: ~420 registers,
// one wave per SIMD, and its counters say it is bound by vector issue and latency, not by the matrix pipe: a wave's
// dependent chains have nothing to hide behind).  Waves 0-3 ("X"): stages A and B and the THIRD column tile of the last layer
// (136 weight registers); waves 4-7 ("Y"): the first two column tiles and the spline passes (112 weight registers + the logit
// column).  Wave w and wave w + 4 own the same site tiles and share the logit scratch: C(t) fills half t & 1 of it during
// interval t (X: channels 32.., Y: channels 0 .. 31), Y maps those sites in the second phase of interval t + 1.
// Interval t (two barriers, as before):
//     X:  A(t+3), third column tile of C(t), first site tile   | bar | B(t+2), third column tile, second site tile       | bar
//     Y:  column tiles 0, 1 of C(t), both site tiles           | bar | spline(t-1), request the field values of plane t  | bar
// Ring hazards: C(t) reads H2[t-1 .. t+1] while B(t+2) writes slot t+2 = t-2 (mod 4); A(t+3) writes H1 slot t-1, last read by
// B(t) one interval earlier.  (Tried: stage A moved to the C waves and run one interval ahead, two rows per phase -- the C
// waves then need 256 registers + 60 bytes of scratch and the kernel is 10 % slower: 3.33 against 3.02 ms per config-3 step.)
template <bool INV, int KIND>
__global__ __launch_bounds__(512, 1) void conv_small3d_kernel8(SmallArgs A) {
  using namespace s3;
  extern __shared__ __align__(16) unsigned char smem_s[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int role = wave >> 2, w4 = wave & 3;
  const int g = lane >> 4, n = lane & 15;
  const int L0 = A.L0, L1 = A.L1;
  const int V = L0 * L1 * LX, PSITES = L1 * LX;
  const int PB = PSITES * 32, HL = PSITES * 16;
  const int PY = L1 + 2;
  const int XB = int(x_bytes(L0, L1));
  unsigned char *Xh = smem_s, *Xl = smem_s + XB;
  unsigned char *H1 = smem_s + 2 * XB;
  unsigned char *H2 = H1 + RING * PB;
  float *pt = reinterpret_cast<float *>(H2 + RING * PB) + w4 * (48 * PTS);
  double *red = reinterpret_cast<double *>(H2 + RING * PB + 4 * PTW);
  const int ntw = (L1 / 2 - w4 + 3) / 4;
  auto wrap1 = [](int v, int L) { return v < 0 ? v + L : (v >= L ? v - L : v); };
  auto ring = [&](int p) { return ((p + 8) & (RING - 1)) * PB; };
  const bool flat = A.flat != 0;
  const int t0 = flat ? -3 : -5;
  auto a_valid = [&](int p) { return flat ? p == 0 : (p >= -2 && p <= L0 + 1); };
  auto b_valid = [&](int p) { return flat ? p == 0 : (p >= -1 && p <= L0); };

  if (flat) {
    for (int i = threadIdx.x * 16; i < 2 * RING * PB; i += 512 * 16) *reinterpret_cast<f32x4 *>(H1 + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto load_x = [&](int64_t sbase) {
    for (int i = threadIdx.x; i < (L0 + 2) * PY * PX; i += 512) {
      const int hz = i / (PY * PX), rem = i - hz * (PY * PX), hy = rem / PX, hx = rem - hy * PX;
      const int sz = wrap1(hz - 1, L0), sy = wrap1(hy - 1, L1), sx = (hx - 1) & (LX - 1);
      const float v = A.xf[sbase + (sz * L1 + sy) * LX + sx];
      const _Float16 hi = static_cast<_Float16>(v);
      reinterpret_cast<_Float16 *>(Xh)[i] = hi;
      reinterpret_cast<_Float16 *>(Xl)[i] = static_cast<_Float16>(v - static_cast<float>(hi));
    }
  };

  if (role == 0) {
    // ================================================================================ waves 0-3: stages A, B and the spline
    const f16x8 a1h = A.w1[lane], a1l = A.w1[64 + lane];
    f16x8 a2h[9], a2l[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      a2h[r] = A.w2[(2 * r) * 64 + lane];
      a2l[r] = A.w2[(2 * r + 1) * 64 + lane];
    }
    const float be1 = A.act1 == kActTanh ? 2.f : 1.f, be2 = A.act2 == kActTanh ? 2.f : 1.f;
    const float al1 = be1, ga1 = 1.f - be1, al2 = be2, ga2 = 1.f - be2;
    const float c11 = -be1 * Num<float>::kLog2e * kInvWScale, c12 = -be2 * Num<float>::kLog2e * kInvWScale;
    float c01[4], c02[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      c01[r] = -be1 * Num<float>::kLog2e * ((A.b1 && g < 2) ? A.b1[4 * g + r] : 0.f);
      c02[r] = -be2 * Num<float>::kLog2e * (A.b2 ? A.b2[4 * (g & 1) + r] : 0.f);
    }
    auto act_of = [](float acc, float c1, float c0, float al, float ga) {
      const float t = Num<float>::exp2(__builtin_fmaf(acc, c1, c0));
      return __builtin_fmaf(al, __builtin_amdgcn_rcpf(1.f + t), ga);
    };
    int tapA[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = 8 * g + i;
      tapA[i] = k < 27 ? (((k / 9 - 1) * PY + ((k / 3) % 3 - 1)) * PX + (k % 3 - 1)) * 2 : 0;
    }
    auto stageA = [&](int p) {
      int pz = p % L0;
      pz = pz < 0 ? pz + L0 : pz;
      unsigned char *dst = H1 + ring(p);
      for (int T = w4; T < L1 / 2; T += 4) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int yrow = 2 * T + rr;
          f16x8 xh, xl;
          const int ctr = (((pz + 1) * PY + yrow + 1) * PX + n + 1) * 2;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            xh[i] = *reinterpret_cast<const _Float16 *>(Xh + ctr + tapA[i]);
            xl[i] = *reinterpret_cast<const _Float16 *>(Xl + ctr + tapA[i]);
          }
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, xh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, xl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1l, xh, acc, 0, 0, 0);
          if (g < 2) {
            f16x4 hi, lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = act_of(acc[r], c11, c01[r], al1, ga1);
              hi[r] = static_cast<_Float16>(v);
              lo[r] = static_cast<_Float16>(v - static_cast<float>(hi[r]));
            }
            unsigned char *d = dst + (yrow * LX + n) * 16 + g * 8;
            *reinterpret_cast<f16x4 *>(d) = hi;
            *reinterpret_cast<f16x4 *>(d + HL) = lo;
          }
        }
      }
    };
    auto stageB = [&](int p) {
      unsigned char *dst = H2 + ring(p);
      const int rr = n >> 3, q = n & 7;
      const int xs = (2 * q + g - 1) & (LX - 1);
      for (int T = w4; T < L1 / 2; T += 4) {
        const int yrow = 2 * T + rr;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j0 = 0; j0 < 3; ++j0) {
          const unsigned char *pl = H1 + ring(p + j0 - 1);
#pragma unroll
          for (int j1 = 0; j1 < 3; ++j1) {
            const int yr = wrap1(yrow + j1 - 1, L1);
            const unsigned char *src = pl + (yr * LX + xs) * 16;
            const f16x8 fh = *reinterpret_cast<const f16x8 *>(src);
            const f16x8 fl = *reinterpret_cast<const f16x8 *>(src + HL);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2h[3 * j0 + j1], fh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2h[3 * j0 + j1], fl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2l[3 * j0 + j1], fh, acc, 0, 0, 0);
          }
        }
        f16x4 hi, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = act_of(acc[r], c12, c02[r], al2, ga2);
          hi[r] = static_cast<_Float16>(v);
          lo[r] = static_cast<_Float16>(v - static_cast<float>(hi[r]));
        }
        unsigned char *d = dst + (yrow * LX + 2 * q + (g >> 1)) * 16 + (g & 1) * 8;
        *reinterpret_cast<f16x4 *>(d) = hi;
        *reinterpret_cast<f16x4 *>(d + HL) = lo;
      }
    };
    // the third column tile of the last layer (logit channels 32 .. 47: the derivative logits' tail) is multiplied here, the
    // first two by the other role: 21 of the 63 MFMAs per tile
    constexpr bool XC = KIND == 0;
    f16x8 x3h[7], x3l[7];
    float x3b = 0.f;
    int tapCx[7];
    if constexpr (XC) {
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        x3h[i] = A.w3[((2 * 7 + i) * 2) * 64 + lane];
        x3l[i] = A.w3[((2 * 7 + i) * 2 + 1) * 64 + lane];
        const int k = 4 * i + g;
        tapCx[i] = k < 27 ? (k / 9) | (((k / 3) % 3) << 2) | ((k % 3) << 4) : (1 | (1 << 2) | (1 << 4));
      }
      x3b = (A.b3 && 32 + n < A.cout) ? A.b3[32 + n] : 0.f;
    }
    auto logits_tile_x = [&](int z, int nt) {
      if constexpr (XC) {
        const int T = w4 + 4 * nt;
        if (T >= L1 / 2) return;
        const int rr = n >> 3, q = n & 7;
        const int yrow = 2 * T + rr;
        const int xa = 2 * q + ((A.parity + z + yrow) & 1);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const int yr = wrap1(yrow + ((tapCx[i] >> 2) & 3) - 1, L1);
          const int xr = (xa + ((tapCx[i] >> 4) & 3) - 1) & (LX - 1);
          const unsigned char *src = H2 + ring(z + (tapCx[i] & 3) - 1) + (yr * LX + xr) * 16;
          const f16x8 fh = *reinterpret_cast<const f16x8 *>(src);
          const f16x8 fl = *reinterpret_cast<const f16x8 *>(src + HL);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, x3h[i], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl, x3h[i], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, x3l[i], acc, 0, 0, 0);
        }
        const int ch = 32 + n;
        if (ch < A.cout) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[r] * kInvWScale + x3b;
          *reinterpret_cast<f32x4 *>(pt + ch * PTS + (z & 1) * 32 + nt * 16 + 4 * g) = v;
        }
      }
    };
    for (int64_t b = blockIdx.x; b < A.B; b += gridDim.x) {
      const int64_t sbase = b * int64_t(V);
      load_x(sbase);
      lds_barrier();
      for (int t = t0; t <= L0; ++t) {
        const bool on = t >= 0 && t < L0;
        if (a_valid(t + 3)) stageA(t + 3);
        if (on) logits_tile_x(t, 0);
        lds_barrier();
        if (b_valid(t + 2)) stageB(t + 2);
        if (on) logits_tile_x(t, 1);
        lds_barrier();
      }
      lds_barrier();
      if (threadIdx.x == 0) A.logj[b] = float((A.log0 ? double(A.log0[b]) : 0.0) + ((red[0] + red[1]) + (red[2] + red[3])));
      lds_barrier();
    }
    return;
  }

  // ===================================================== waves 4-7: the last layer's first two column tiles, and the spline
  constexpr int NTY = KIND == 1 ? 1 : 2;
  f16x8 b3h[7][NTY], b3l[7][NTY];
#pragma unroll
  for (int t = 0; t < NTY; ++t)
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      b3h[i][t] = A.w3[((t * 7 + i) * 2) * 64 + lane];
      b3l[i][t] = A.w3[((t * 7 + i) * 2 + 1) * 64 + lane];
    }
  float b3v[NTY];
#pragma unroll
  for (int t = 0; t < NTY; ++t) b3v[t] = (A.b3 && 16 * t + n < A.cout) ? A.b3[16 * t + n] : 0.f;
  int tapC[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int k = 4 * i + g;
    tapC[i] = k < 27 ? (k / 9) | (((k / 3) % 3) << 2) | ((k % 3) << 4) : (1 | (1 << 2) | (1 << 4));
  }
  auto logits_tile = [&](int z, int nt) {               // tile nt (0, 1) of this wave in plane z
    const int T = w4 + 4 * nt;
    if (T >= L1 / 2) return;
    const int rr = n >> 3, q = n & 7;
    const int yrow = 2 * T + rr;
    const int xa = 2 * q + ((A.parity + z + yrow) & 1);
    f32x4 acc[NTY];
#pragma unroll
    for (int t = 0; t < NTY; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int yr = wrap1(yrow + ((tapC[i] >> 2) & 3) - 1, L1);
      const int xr = (xa + ((tapC[i] >> 4) & 3) - 1) & (LX - 1);
      const unsigned char *src = H2 + ring(z + (tapC[i] & 3) - 1) + (yr * LX + xr) * 16;
      const f16x8 fh = *reinterpret_cast<const f16x8 *>(src);
      const f16x8 fl = *reinterpret_cast<const f16x8 *>(src + HL);
#pragma unroll
      for (int t = 0; t < NTY; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, b3h[i][t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl, b3h[i][t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, b3l[i][t], acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < NTY; ++t) {
      const int ch = 16 * t + n;
      if (ch < A.cout) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[t][r] * kInvWScale + b3v[t];
        *reinterpret_cast<f32x4 *>(pt + ch * PTS + (z & 1) * 32 + nt * 16 + 4 * g) = v;
      }
    }
  };
  double lacc = 0.0;
  float xpre = 0.f;
  int64_t spre = -1;
  auto prefetch_x = [&](int z, int64_t sbase) {       // the field values of plane z's sites (this wave's tiles), one interval ahead
    spre = -1;
    const int u = lane;
    if (z >= 0 && z < L0 && u < ntw * 16) {
      const int T = w4 + 4 * (u >> 4), m_ = u & 15;
      const int yrow = 2 * T + (m_ >> 3), qq = m_ & 7;
      const int xsite = 2 * qq + ((A.parity + z + yrow) & 1);
      spre = sbase + (int64_t(z) * L1 + yrow) * LX + xsite;
      xpre = A.xa[spre];
    }
  };
  auto spline = [&](int z) {                          // plane z's logits: half z & 1 of the scratch shared with wave w4 (its third column tile)
    if (spre >= 0) {
      const int u = (z & 1) * 32 + lane;
      float val, logd;
      if constexpr (KIND == 1) {
        const float tt = pt[u], ss = fabsf(pt[PTS + u]);
        val = INV ? (xpre - tt) * __expf(ss) : tt + xpre * __expf(-ss);
        logd = INV ? ss : -ss;
      } else if (A.P.m == M) {
        RegCol<float, C> col;
#pragma unroll
        for (int c = 0; c < C; ++c) col[c] = pt[c * PTS + u];
        rqs_site<float, M, INV>(col, A.P, xpre, val, logd);
      } else if (A.P.m == 8) {
        RegCol<float, 22> col;
#pragma unroll
        for (int c = 0; c < 22; ++c) col[c] = pt[c * PTS + u];
        rqs_site<float, 8, INV>(col, A.P, xpre, val, logd);
      } else {
        LdsCol<float> col{pt + u, PTS};
        rqs_site<float, 0, INV>(col, A.P, xpre, val, logd);
      }
      A.y[spre] = val;
      A.y[spre ^ 1] = 0.f;
      lacc += double(logd);
    }
  };
  for (int64_t b = blockIdx.x; b < A.B; b += gridDim.x) {
    const int64_t sbase = b * int64_t(V);
    load_x(sbase);
    lacc = 0.0;
    spre = -1;
    lds_barrier();
    for (int t = t0; t <= L0; ++t) {
      const bool on = t >= 0 && t < L0;
      if (on) { logits_tile(t, 0); logits_tile(t, 1); }
      lds_barrier();
      if (t >= 1) spline(t - 1);          // plane t-1: its three column tiles were complete when interval t-1 ended
      prefetch_x(t, sbase);
      lds_barrier();
    }
    const double tot = wave_sum(lacc);
    if (lane == 0) red[w4] = tot;
    lds_barrier();
    lds_barrier();
  }
}

}  // namespace nf

using namespace nf;

// kind: 0 RQ-spline (cout = 3m - 2), 1 affine (cout = 2); ndim 3: lattice (L0, L1, 16), ndim 2: (L1, 16)
static int small_supported(const int32_t *lattice, int ndim, int kind, int cout, int m, int act1, int act2) {
  if (!lattice || !option(NF_OPT_SPLIT16) || (ndim != 2 && ndim != 3)) return 0;
  const int L0 = ndim == 3 ? lattice[0] : 1, L1 = lattice[ndim - 2], L2 = lattice[ndim - 1];
  if (kind == 0 && (m < 2 || m > s3::M || cout != 3 * m - 2)) return 0;
  if (kind == 1 && cout != 2) return 0;
  if (kind != 0 && kind != 1) return 0;
  if (L2 != s3::LX || L0 < 1 || L1 < 2 || (L1 & 1) || L1 > 16) return 0;      // (a wave's logit scratch holds 2 tiles)
  if ((act1 != kActTanh && act1 != kActSigmoid) || (act2 != kActTanh && act2 != kActSigmoid)) return 0;
  if (s3::lds_bytes(L0, L1) > 160 * 1024) return 0;
  return 1;
}

extern "C" int nf_small3d_rqs_supported(const int32_t *lattice3, int cout, int m, int act1, int act2) {
  return small_supported(lattice3, 3, 0, cout, m, act1, act2);
}
extern "C" int nf_small_lattice_supported(const int32_t *lattice, int ndim, int kind, int cout, int m, int act1, int act2) {
  return small_supported(lattice, ndim, kind, cout, m, act1, act2);
}

static int small_launch(const char *who, int kind, const void *x_frozen, const void *x_active, const void *w1, const void *b1,
                        const void *w2, const void *b2, const void *w3, const void *b3, const void *log0, void *y, void *logj,
                        int64_t B, const int32_t *lattice, int ndim, int active_parity, int cout, int act1, int act2,
                        const nf_rqs_opts *opts, int inverse, hipStream_t stream) {
  NF_REQUIRE(lattice && (kind == 1 || opts), "%s: NULL pointer", who);
  NF_REQUIRE(B >= 0, "%s: negative batch", who);
  NF_REQUIRE(small_supported(lattice, ndim, kind, cout, kind == 0 ? opts->m : 0, act1, act2),
             "%s: needs a lattice (L0, L1 even <= 16, 16) or (L1 even <= 16, 16) that fits the LDS, %s, tanh / logistic hidden "
             "activations (got ndim=%d, cout=%d)", who, kind == 0 ? "knots_len 2..16 with cout = 3m-2" : "cout = 2", ndim, cout);
  if (kind == 0) {
    NF_REQUIRE(!opts->fixed_knots_x && !opts->fixed_knots_y, "%s: fixed knots are not fused", who);
    NF_REQUIRE(opts->xhi > opts->xlo && opts->yhi > opts->ylo, "%s: empty xlim/ylim", who);
  }
  if (B == 0) return NF_OK;
  NF_REQUIRE(x_frozen && x_active && w1 && w2 && w3 && y && logj, "%s: NULL tensor pointer", who);
  SmallArgs A{};
  A.xf = static_cast<const float *>(x_frozen); A.xa = static_cast<const float *>(x_active);
  A.y = static_cast<float *>(y); A.log0 = static_cast<const float *>(log0); A.logj = static_cast<float *>(logj);
  A.w1 = static_cast<const f16x8 *>(w1); A.w2 = static_cast<const f16x8 *>(w2); A.w3 = static_cast<const f16x8 *>(w3);
  A.b1 = static_cast<const float *>(b1); A.b2 = static_cast<const float *>(b2); A.b3 = static_cast<const float *>(b3);
  A.B = B; A.L0 = ndim == 3 ? lattice[0] : 1; A.L1 = lattice[ndim - 2]; A.flat = ndim == 2 ? 1 : 0;
  A.parity = active_parity & 1; A.cout = cout; A.act1 = act1; A.act2 = act2;
  if (kind == 0) {
    A.P.xlo = opts->xlo; A.P.xhi = opts->xhi; A.P.ylo = opts->ylo; A.P.yhi = opts->yhi;
    A.P.fx = nullptr; A.P.fy = nullptr; A.P.m = opts->m; A.P.el = opts->extrap_left; A.P.er = opts->extrap_right;
  }
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { set_error("%s: no device properties", who); return NF_ELAUNCH; }
    ncu = prop.multiProcessorCount;
  }
  const int64_t grid = B < ncu ? B : ncu;
  const int lds = int(s3::lds_bytes(A.L0, A.L1));
  auto go = [&](auto kern, int threads) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1;
    hipLaunchKernelGGL(kern, dim3(unsigned(grid)), dim3(threads), lds, stream, A);
    return 0;
  };
  int rc;
  if (option(NF_OPT_SMALL8)) {       // eight waves in two roles (the default)
    if (kind == 0) rc = inverse ? go(&conv_small3d_kernel8<true, 0>, 512) : go(&conv_small3d_kernel8<false, 0>, 512);
    else rc = inverse ? go(&conv_small3d_kernel8<true, 1>, 512) : go(&conv_small3d_kernel8<false, 1>, 512);
  } else {
    if (kind == 0) rc = inverse ? go(&conv_small3d_kernel<true, 0>, 256) : go(&conv_small3d_kernel<false, 0>, 256);
    else rc = inverse ? go(&conv_small3d_kernel<true, 1>, 256) : go(&conv_small3d_kernel<false, 1>, 256);
  }
  if (rc != 0) {
    set_error("%s: could not configure the kernel's LDS (%d bytes)", who, lds);
    return NF_ELAUNCH;
  }
  return check_launch("small-lattice fused layer kernel");
}

extern "C" int nf_small3d_rqs(const void *x_frozen, const void *x_active, const void *w1, const void *b1, const void *w2,
                              const void *b2, const void *w3, const void *b3, const void *log0, void *y, void *logj,
                              int64_t B, const int32_t *lattice3, int active_parity, int cout, int act1, int act2,
                              const nf_rqs_opts *opts, int inverse, void *stream) {
  return small_launch("nf_small3d_rqs", 0, x_frozen, x_active, w1, b1, w2, b2, w3, b3, log0, y, logj, B, lattice3, 3,
                      active_parity, cout, act1, act2, opts, inverse, static_cast<hipStream_t>(stream));
}

extern "C" int nf_small_lattice_coupling(int kind, const void *x_frozen, const void *x_active, const void *w1, const void *b1,
                                         const void *w2, const void *b2, const void *w3, const void *b3, const void *log0,
                                         void *y, void *logj, int64_t B, const int32_t *lattice, int ndim, int active_parity,
                                         int cout, int act1, int act2, const nf_rqs_opts *opts, int inverse, void *stream) {
  return small_launch("nf_small_lattice_coupling", kind, x_frozen, x_active, w1, b1, w2, b2, w3, b3, log0, y, logj, B, lattice,
                      ndim, active_parity, cout, act1, act2, opts, inverse, static_cast<hipStream_t>(stream));
}
