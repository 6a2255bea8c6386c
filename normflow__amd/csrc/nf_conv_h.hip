// nf_conv_h.hip -- K5h: the fused last ConvAct layer (8 -> 46 channels at the active sites + RQ-spline coupling)
// with every fp32 product computed as THREE fp16 matrix-core products,
//        a * w  ~=  a_hi*w_hi + a_hi*w_lo + a_lo*w_hi        (x_hi = fp16(x), x_lo = fp16(x - x_hi)),
// accumulated in fp32 (v_mfma_f32_16x16x32_f16).  The dropped term a_lo*w_lo is 2^-22 relative; measured against the
// fp64 definition the layer's error is ~1.7x that of an fp32 fmaf chain over the same 648 terms (rms 1.1e-6 vs 6.5e-7
// on O(1) outputs), inside the 1e-5 budget -- and the fp16 pipe is 16x the fp32 one per product, so a product costs
// 3/16 of an fp32 MFMA slot (tools/mfma_probe3.hip: 370-410 fp32-equivalent TFLOP/s where the fp32 loop tops out at ~130).
// fp16 has a short exponent: the kernel is only used when the caller guarantees |input| <= 1 (NF_CONV_UNIT_INPUT:
// the hidden activations are tanh outputs) and the weights are finite fp16-range numbers (checked on the host side).
//
// Shape of the computation (weight-stationary; reference: src/nn/scalar/modules.py:120-145 + couplings_.py:178-200):
//   * one persistent workgroup per CU, 4 waves, items = (sample, 2x2x2x32 box -> 128 active sites = 8 site tiles).
//   * K = 648 is cut into 21 slices of 32 = (4 kernel rows x 8 input channels) at one tap j3 of the fastest axis
//     (v_mfma_f32_16x16x32_f16).  Wave w in {0, 1, 2} owns tap j3 = w: its 7 slices for ALL three 16-column tiles of the
//     46 logit channels, B fragments (hi, lo) in 168 registers for the whole launch, 8 x 3 accumulators, 504 MFMAs per
//     item -- 9 per A fragment pair, and no fragment is read by two waves.
//   * the three partial sums of a column tile meet in LDS: stored straight from the accumulator registers into the logit
//     scratch and two planes laid over the consumed image, then added up 16 bytes at a time by the 192 lanes.
//   * wave 3 is the data mover: while the others multiply item m it (a) runs the RQ-spline epilogue of item m-1 on the
//     logits in LDS (they never reach HBM) and (b) stages item m+1 from the (hi, lo) fp16 pairs the previous layer wrote
//     (or splits fp32 planes itself: the fallback) into two parity-split channel-last LDS images of 16 bytes per site.
//   * three barriers per item; LDS = 2 x (2 x 32 KB) images + 24 KB of logits = 152 KB.
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <type_traits>
#include "nf_conv_core.h"

namespace nf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace h {
constexpr int H0 = 4, H1 = 4, H2 = 4;                     // halo rows of a 2x2x2x32 box under a 3^4 kernel; the box spans the fastest axis
constexpr int SEGW = 32;                                  // sites of a SEGMENT of the fastest axis = 16 pairs = one site tile per box row.
// A box is 2 x 2 x 2 lattice rows x one segment.  SEGM = false: the segment is the whole periodic row (L3 = 32): its taps wrap by
// address (slot index mod 16), no halo sites, no periodic copies.  SEGM = true (L3 = 48, 64, ...): a row image holds the 17
// slots 16 h .. 16 h + 16 (mod L3 / 2) of the four blocks of the pair tensor's row.
constexpr int NROW = H0 * H1 * H2;                        // 64 halo rows: one per lane of the loader wave
// The LDS image of an item is rows of the pair layout (pair_row_offset, nf_conv_core.h): 64 halo rows of
// [hi | lo][even sites | odd sites][16 slots x 16 B] (1 KiB, one LDS-DMA piece = one wave-instruction) + when SEGM the four
// 17th slots (64 B, a second 4-lane piece) -- no staging registers and no ds_write on the way.  The 16 lanes of a k-group of
// an A fragment (v_mfma_f32_16x16x32_f16: k-group g = the 8 channels of ONE tap) read 16 consecutive slots of one parity
// block (rotated, or shifted by one into the 17th): 256 contiguous bytes, no bank conflict.
template <bool SEGM>
struct Img {
  static constexpr int RBL = SEGM ? 1088 : 1024;           // bytes of a row image
  static constexpr int ITEM = NROW * RBL;                  // 64 / 68 KiB: one item's image (hi and lo)
};
__host__ __device__ constexpr int rowidx(int r) { return ((r / 9) * H1 + (r / 3) % 3) * H2 + r % 3; }   // kernel row -> halo row step
// K slices of 32 = (4 kernel rows) x (8 channels) at ONE tap j3 of the fastest axis: slice sl = 7*j3 + i holds kernel
// rows 4i .. 4i+3 (row 27 is padding: zero weights).  Same j3 for the four k-groups => same parity sub-image.
constexpr int NS = 21;
constexpr int UNITS = 128;                                // active sites per box
constexpr float kWScale = 1024.0f, kInvWScale = 1.0f / 1024.0f;      // normflow__amd/_hip.py: SPLIT16_WEIGHT_SCALE
constexpr int C = 46, M = 16;                            // the LARGEST logit count / knots_len (3 column tiles); a layer may have fewer:
                                                          // cout = 3 m - 2 <= 46 at run time (columns >= cout: zero weights, never stored)
constexpr int PTS = UNITS + 4;                            // row stride of the logit scratch in floats: +4 spreads the 16 channels a
                                                          // wave writes at once over the banks (stride 128 put them all on one)
constexpr int PT = C * PTS * 4;                           // bytes of the logit scratch
template <bool SEGM> constexpr int lds_bytes() { return 2 * Img<SEGM>::ITEM + PT; }
static_assert(2 * PT <= Img<false>::ITEM, "two planes of partial sums must fit a consumed image");
static_assert(lds_bytes<true>() <= 160 * 1024, "two item images and the logit scratch must fit the CU's LDS");

}  // namespace h

#ifndef NF_H_DMA_EARLY
#define NF_H_DMA_EARLY NROW
#endif
#ifndef NF_H_EPI
#define NF_H_EPI 1          // 1: the mover's spline pass with the bin fetched by index (rqs_site_pt), 0: the generic rqs_site
#endif
#if !defined(NF_DIAG) || !defined(NF_H_ABL)
#undef NF_H_ABL
#define NF_H_ABL 0      // timing ablations of the compute waves (diagnostic builds only, make DIAG=1 with -DNF_H_ABL=..): 1 no fragment reads, 2 no reduction
#endif

typedef __attribute__((address_space(3))) float lds_f;

template <int LO, int HI, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (LO < HI) {
    f(std::integral_constant<int, LO>{});
    static_for<LO + 1, HI>(f);
  }
}

// The mover's spline pass: rqs_site (nf_rqs_core.h) for m = 16 free knots with the logits of the site in an LDS column
// (`col[c * PTS]`, c = 0..45), restated so that the bin is FOUND by the scan and FETCHED by its index: the scan carries the
// two running knots and a counter (2 FMA-able adds, 1 compare, 2 selects, 1 add per knot) instead of selecting seven values
// per knot, and the bin's width / height numerators and the two derivative logits are read back from the column at
// [j] -- 34 LDS reads and ~300 vector instructions per site where the generic form takes 46 and ~450.  Same operations
// in the same order wherever a value is formed (running sums, widths as numerator x scale); two roundings differ from the
// generic form: the softmax argument is one FMA, and quotients sharing a divisor share its reciprocal (both ~1e-7
// relative; the parity tests hold the fused layer to the unfused one and to the oracle at the same bounds as before).
// The mover's two passes per item are what it has to hide in an MFMA phase.
template <bool INV>
__device__ __forceinline__ void rqs_site_pt(const lds_f *col, const RqsParams &A, float v, float &val, float &logd) {
  using namespace h;
  constexpr int NB = M - 1, OX = 0, OY = NB, OD = 2 * NB;
  const float xlo = float(A.xlo), W = float(A.xhi) - float(A.xlo), ylo = float(A.ylo), H = float(A.yhi) - float(A.ylo);
  const float in_lo = INV ? ylo : xlo, in_hi = INV ? ylo + H : xlo + W;
  const float out_lo = INV ? xlo : ylo, out_hi = INV ? xlo + W : ylo + H;
  const bool refl_l = (A.el == NF_EXTRAP_ANTI) && (v < in_lo);
  const bool refl_r = (A.er == NF_EXTRAP_ANTI) && (v > in_hi);
  v = refl_l ? 2.f * in_lo - v : (refl_r ? 2.f * in_hi - v : v);
  float a[2 * NB];
#pragma unroll
  for (int c = 0; c < 2 * NB; ++c) a[c] = col[c * PTS];
  float amax = a[OX], bmax = a[OY];
#pragma unroll
  for (int k = 1; k < NB; ++k) {
    amax = Num<float>::max(amax, a[OX + k]);
    bmax = Num<float>::max(bmax, a[OY + k]);
  }
  // exp(a - max) as exp2(a log2e - max log2e): one FMA per logit (the generic form subtracts, then scales)
  const float am2 = -amax * Num<float>::kLog2e, bm2 = -bmax * Num<float>::kLog2e;
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    a[OX + k] = Num<float>::exp2(__builtin_fmaf(a[OX + k], Num<float>::kLog2e, am2));
    sa += a[OX + k];
  }
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    a[OY + k] = Num<float>::exp2(__builtin_fmaf(a[OY + k], Num<float>::kLog2e, bm2));
    sb += a[OY + k];
  }
  const float wx = W / sa, wy = H / sb;
  // running knots; knot k (k = 1..NB-1) strictly below the value moves the bin up
  float cx = xlo + a[OX] * wx, cy = ylo + a[OY] * wy;
  float x0 = xlo, y0 = ylo;
  int j = 0;
#pragma unroll
  for (int k = 1; k < NB; ++k) {
    const bool sel = (INV ? cy : cx) < v;
    x0 = sel ? cx : x0;
    y0 = sel ? cy : y0;
    j += sel ? 1 : 0;
    cx = cx + a[OX + k] * wx;
    cy = cy + a[OY + k] * wy;
  }
  const float xe = cx, ye = cy;      // last knot as accumulated (the reference's cumsum end)
  const lds_f *cj = col + j * PTS;
  const float bw = Num<float>::exp2(__builtin_fmaf(cj[OX * PTS], Num<float>::kLog2e, am2)) * wx;
  const float bh = Num<float>::exp2(__builtin_fmaf(cj[OY * PTS], Num<float>::kLog2e, bm2)) * wy;
  const float c0 = cj[OD * PTS], c1 = cj[(OD + 1) * PTS];
  const bool tail_l = (A.el == NF_EXTRAP_LINEAR) && !(in_lo < v);
  const bool tail_r = (A.er == NF_EXTRAP_LINEAR) && ((INV ? ye : xe) < v);
  const float d0 = softplus2(c0), d1 = softplus2(c1);
  const float ibw = 1.f / bw;         // one correctly rounded reciprocal for the slope and for theta, one for the two
  const float sl = bh * ibw;          // quotients by den (the generic form divides four times: ~10 instructions each)
  const float curv = d0 + d1 - 2.f * sl;
  float th, g;
  if (!INV) {
    th = (v - x0) * ibw;
    const float t1 = th * (1.f - th);
    const float den = sl + curv * t1, iden = 1.f / den;
    val = y0 + bh * (sl * th * th + d0 * t1) * iden;
    const float P = d1 * th * th + 2.f * sl * t1 + d0 * (1.f - th) * (1.f - th);
    g = sl * sl * P * (iden * iden);
    val = tail_l ? ylo + d0 * (v - xlo) : (tail_r ? ye + d1 * (v - xe) : val);
    g = tail_l ? d0 : (tail_r ? d1 : g);
    logd = nf_log(g);
  } else {
    const float eta = (v - y0) / bh;
    const float a2 = -curv * eta + d0 - sl;
    const float bb = a2 + sl;
    const float a0 = sl * eta;
    const float disc = Num<float>::sqrt(Num<float>::max(bb * bb - 4.f * a0 * a2, 0.f));
    th = (bb >= 0.f) ? 2.f * a0 / (bb + disc) : (bb - disc) / (2.f * a2);
    const float t1 = th * (1.f - th);
    const float den = sl + curv * t1;
    const float P = d1 * th * th + 2.f * sl * t1 + d0 * (1.f - th) * (1.f - th);
    g = sl * sl * P / (den * den);
    val = x0 + bw * th;
    val = tail_l ? xlo + (v - ylo) / d0 : (tail_r ? xe + (v - ye) / d1 : val);
    g = tail_l ? d0 : (tail_r ? d1 : g);
    logd = -nf_log(g);
  }
  val = refl_l ? 2.f * out_lo - val : (refl_r ? 2.f * out_hi - val : val);
}

template <int FUSE, bool SEGM>
__global__ __launch_bounds__(256, 1) void conv_h_kernel(ConvArgs A) {
  using namespace h;
  constexpr int RBL = Img<SEGM>::RBL, ITEM = Img<SEGM>::ITEM;
  const int L3 = A.L[3], HP = L3 >> 1;                    // sites / pairs of a lattice row (L3 = 32 when !SEGM)
  const int HB = L3 * 16, PBK = L3 * 8;                   // bytes of the hi block / of a parity block of a row of the pair tensor
  extern __shared__ __align__(16) unsigned char smem_h[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4;

  // Item order (round 3).  An item = (sample, box, segment).  Every XCD (= blockIdx & 7: workgroups are dealt to the XCDs round-robin)
  // owns a CONTIGUOUS eighth of the launch's (sample, box) pairs -- boxes of a sample numbered [tile of 4 x 4 (4 x 8 when a row is
  // one segment) boxes in axes 1, 2][box along axis 0][place in the tile] -- and goes through it once per segment, its workgroups
  // taking consecutive ones.  The 32 workgroups of an XCD then work on one compact tile and march it along axis 0: three
  // quarters of the halo rows an item asks for were asked for by a neighbour on the same XCD at the same time or one step
  // earlier (its L2 has them).  Rounds 1-2 numbered the boxes row-major and dealt every step's 256 items out in eighths, which
  // brought the neighbour along axis 0 back one step later when a sample has 16 x 16 x 16 boxes of one segment (32^4: 15.3 GB
  // fetched per 256 samples, 1.8 x the tensor) and never otherwise (48^4: 37.8 GB per 50 samples, 4.4 x).  Segment by segment
  // rather than segment-fastest: a workgroup's consecutive items are then of one kind (full or half, below), and the phases
  // of the mover and of the multiplying waves stay matched.
  const int nb = gridDim.x, nx = nb >> 3;
  const int nsg = SEGM ? A.nbox[3] : 1;                     // (all counts below fit 31 bits: the launcher checks nitems)
  const int nbs = A.nboxes / nsg;                           // boxes of a sample (segments apart)
  const int nb0 = int(A.nitems) / nsg;                      // ... of the launch
  const int nper = (nb0 + 7) >> 3;                          // an XCD's share of them
  const int g0 = int(blockIdx.x & 7) * nper;
  if (g0 >= nb0) return;
  const int R = nb0 - g0 < nper ? nb0 - g0 : nper;
  const int l0 = blockIdx.x >> 3;                           // local item numbers of the XCD: l = segment * R + r, box g0 + r
  if (l0 >= R * nsg) return;
  const int n_my = (R * nsg - l0 + nx - 1) / nx;
  // mixed radix of a box number, least significant digit last: [T1][T2][n0][t1][t2]
  const int td1 = (A.nbox[1] & 3) == 0 ? 4 : ((A.nbox[1] & 1) == 0 ? 2 : 1);
  const int td2 = (nsg == 1 && (A.nbox[2] & 7) == 0) ? 8 : ((A.nbox[2] & 3) == 0 ? 4 : ((A.nbox[2] & 1) == 0 ? 2 : 1));
  const int rad[5] = {A.nbox[1] / td1, A.nbox[2] / td2, A.nbox[0], td1, td2};
  struct Item {
    int l, seg, b, d[5];
  };
  auto decode = [&](int l, Item &it) {
    it.l = l;
    it.seg = SEGM ? l / R : 0;
    const int gl = g0 + (l - it.seg * R);
    it.b = gl / nbs;
    int q = gl - it.b * nbs;
#pragma unroll
    for (int k = 4; k >= 0; --k) {
      it.d[k] = q % rad[k];
      q /= rad[k];
    }
  };
  auto coords = [&](const Item &it, int (&o)[4]) {          // box origin in sites (boxes are 2 x 2 x 2 rows x one 32-site segment)
    o[0] = 2 * it.d[2];
    o[1] = 2 * (it.d[0] * td1 + it.d[3]);
    o[2] = 2 * (it.d[1] * td2 + it.d[4]);
    o[3] = SEGW * it.seg;
  };
  auto slot_of = [&](const Item &it) {                      // the item's place among the log-det partials: [sample][box][segment]
    const int q = (((it.d[0] * rad[1] + it.d[1]) * rad[2] + it.d[2]) * rad[3] + it.d[3]) * rad[4] + it.d[4];
    return int64_t(it.b) * A.nboxes + int64_t(q) * nsg + it.seg;
  };
  // a workgroup's items are nx apart: (sample, digits) advance by mixed-radix counters, no divisions -- but for the (rare) step
  // into the next segment, which goes back to the start of the XCD's range
  int sb_ = nx / nbs, sd_[5];
  {
    int q = nx - sb_ * nbs;
#pragma unroll
    for (int k = 4; k >= 0; --k) {
      sd_[k] = q % rad[k];
      q /= rad[k];
    }
  }
  auto advance = [&](Item &it) {
    const int ln = it.l + nx;
    if (SEGM && ln - it.seg * R >= R) {
      decode(ln, it);
      return;
    }
    it.l = ln;
    int carry = 0;
#pragma unroll
    for (int k = 4; k >= 0; --k) {
      it.d[k] += sd_[k] + carry;
      carry = it.d[k] >= rad[k] ? 1 : 0;
      it.d[k] -= carry ? rad[k] : 0;
    }
    it.b += sb_ + carry;
  };

  // the input may come scaled by a power of two (training: activations of unknown range, nf_conv_last_logits_split16)
  const float in_scale = pow2_scale_for(A.gscale_bits), out_scale = kInvWScale / in_scale;
  // HALF ITEMS (round 3).  When a row is a whole number of segments plus 8 pairs (L3 = 48, 80, ...), the last segment of a box
  // holds 8 pairs per row: as one site tile per row half of every tile's lanes were padding (48^4 ran at 0.71 of the per-site
  // rate of 32^4).  Such an item now packs the two rows (z2 = 0, 1) of a (z0, z1) position into ONE site tile -- lanes 0-7 row
  // z2 = 0, lanes 8-15 row z2 = 1 -- : 4 site tiles, half the MFMAs, half the exchange, one spline pass.  Unit u of its logit
  // scratch is (T = u >> 4 = 2 z0 + z1, z2 = (u >> 3) & 1, pair 16 (nseg - 1) + (u & 7)).  The image and its staging are as before.
  const bool has_half = SEGM && (HP & 15) == 8;
  const int l_half = R * (nsg - 1);                         // local item numbers >= l_half are the last segment's

  if (wave < 3) {
    // ============================================================ compute waves: K third `wave` = fastest-axis tap j3
    // Wave w multiplies the 7 slices of tap j3 = w (kernel rows 4i..4i+3, i = 0..6) into ALL three column tiles of all 8
    // site tiles: 36 MFMAs per 8 fragment reads and no fragment read by two waves (with one column tile per wave the
    // three waves read the same 1 MB per item and the LDS pipe, not the matrix pipe, set the pace).  The three partial
    // sums meet in the logit scratch in three rounds, every wave working on a different column tile in each round
    // (round 0 stores, rounds 1 and 2 add in the LDS itself; fixed order per column tile: deterministic).
    // Column SLOT k of wave w is column tile (w + k) % 3: the code is the same for the three waves, only data differ.
    const int W = wave;
    f16x8 bh[7][3], bl[7][3];
    {
      const f16x8 *__restrict__ wsp = static_cast<const f16x8 *>(A.wfrag) + lane;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int t = (W + k) % 3;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          bh[i][k] = wsp[((t * NS + 7 * W + i) * 2) * 64];
          bl[i][k] = wsp[((t * NS + 7 * W + i) * 2 + 1) * 64];
        }
      }
    }
    // byte offset of this lane's A read = T[site tile] + RG[i]: T places the lane's site (box row mt: z0 = mt>>2,
    // z1 = (mt>>1)&1, z2 = mt&1; halo index 2p + parity + j3) in its parity sub-image, RG adds the halo rows of kernel
    // row 4i + g.  Box extents are even, so the parity of a box row does not depend on the box.
    int TP[2], TL[2], RG[7];  // hi / lo offsets of this lane's slot in a row image, by parity of the box row; halo rows of the slices
    {
      const int p = lane & 15;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        // tap W of the active site 2p + a of pair p (a = (parity + e) & 1): site 2p + a + W - 1 = tap index gI = a + W of the pair:
        // parity block (gI + 1) & 1, local slot p + (gI >> 1) (nf_conv_g.hip has the same four taps as its k-groups)
        const int gI = ((A.parity + e) & 1) + W, ja = p + (gI >> 1), para = (gI + 1) & 1;
        const bool x17 = SEGM && ja == 16;
        TP[e] = x17 ? 1024 + para * 16 : para * 256 + (ja & 15) * 16;
        TL[e] = TP[e] + (x17 ? 32 : 512);
      }
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int r = 4 * i + g;
        const int rr = r < 27 ? r : 26;
        RG[i] = (((rr / 9) * H1 + (rr / 3) % 3) * H2 + rr % 3) * RBL;
      }
    }
    int TPm[2] = {0, 0};      // half items: the lane's slot in the MERGED tile of position (z0, z1), by parity of z0 + z1 (lo: + 512)
    if (SEGM) {
      const int pm = lane & 7, z2l = (lane >> 3) & 1;
#pragma unroll
      for (int e0 = 0; e0 < 2; ++e0) {
        const int gI = ((A.parity + (e0 ^ z2l)) & 1) + W, ja = pm + (gI >> 1), para = (gI + 1) & 1;      // ja <= 8: inside the main piece
        TPm[e0] = para * 256 + ja * 16 + z2l * RBL;
      }
    }
    // The three partial sums of a column tile: slot 0 (the wave's OWN column tile, w) stays in the accumulators; slots 1 and
    // 2 are stored straight from the accumulator registers (inline asm: hipcc would copy them to VGPRs first) into two
    // planes [channel][unit] laid over the image the item has just consumed (unit = 16 mt + 4g + r: a 16-byte store per
    // site tile, conflict-free with the row stride PTS); one barrier; every wave then reads the two partials for its own
    // tile, adds ((slot 0 + slot 1) + slot 2: deterministic), rescales and stores the logits in pt.  LDS atomics for the
    // adds (ds_add_f32: ~170 cycles per instruction) and three read-modify-write rounds through pt both cost far more;
    // what the rounds cost is LDS bytes, and this form moves the fewest.  Slot 0's accumulators start from the bias,
    // scaled by 2^10 like the weights.
    const int lds0 = int(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char *)smem_h));
    int ptk[3];               // LDS byte address of this lane's 16 bytes in plane k, site tile 0 (planes 1, 2: relative to the image)
    bool okk[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int co = (((W + k) % 3) << 4) + (lane & 15);
      okk[k] = co < A.cout;
      ptk[k] = lds0 + ((okk[k] ? co : 0) * PTS + (g << 2)) * 4 + (k == 0 ? 2 * ITEM : (k - 1) * PT);
    }
    const int home = (((W << 4) + (lane & 15)) * PTS + (g << 2)) * 4;      // byte offset of this lane's unit quad (site tile 0) in a plane, own column tile
    f32x4 acc0;               // what slot 0 starts from
    {
      const int co = (W << 4) + (lane & 15);
      const float b0 = (A.bias && co < A.cout) ? static_cast<const float *>(A.bias)[co] * kWScale * in_scale : 0.f;
      acc0 = f32x4{b0, b0, b0, b0};
    }
    f32x4 acc[8][3];
    f16x8 fa[2][2], fl[2][2];                    // fragments (hi, lo) of two quarter-slices (2 site tiles each), read one ahead
#if NF_H_ABL & 1
    fa[0][0] = fa[0][1] = fa[1][0] = fa[1][1] = fl[0][0] = fl[0][1] = fl[1][0] = fl[1][1] = bh[0][0];
#endif
    lds_barrier();            // P: the mover has staged the first image
    for (int m = 0; m < n_my; ++m) {
      const int ioff = (m & 1) * ITEM;
      const bool half = has_half && l0 + m * nx >= l_half;
      const int ntile = half ? 4 : 8;           // site tiles of the item (their accumulators: acc[0 .. ntile))
      const unsigned char *img[2] = {smem_h + ioff + TP[0], smem_h + ioff + TP[1]};
      const unsigned char *iml[2] = {smem_h + ioff + TL[0], smem_h + ioff + TL[1]};
      const unsigned char *imgm[2] = {smem_h + ioff + TPm[0], smem_h + ioff + TPm[1]};
      // half item: quarter-slice qs = (slice i, position T = qs & 3 = 2 z0 + z1): ONE merged tile, 9 MFMAs; accumulators acc[T]
      auto fetch_h = [&](auto QC) {
        constexpr int qs = decltype(QC)::value;
        constexpr int i = qs >> 2, T = qs & 3, q = qs & 1;
        constexpr int z0 = T >> 1, z1 = T & 1;
        const int ro = RG[i] + ((z0 * H1 + z1) * H2) * RBL;
        fa[q][0] = *reinterpret_cast<const f16x8 *>(imgm[(z0 + z1) & 1] + ro);
        fl[q][0] = *reinterpret_cast<const f16x8 *>(imgm[(z0 + z1) & 1] + 512 + ro);
      };
      auto mult_h = [&](auto QC) {
        constexpr int qs = decltype(QC)::value;
        constexpr int i = qs >> 2, T = qs & 3, q = qs & 1;
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[T][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[q][0], bh[i][n], acc[T][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[T][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[q][0], bl[i][n], acc[T][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[T][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[q][0], bh[i][n], acc[T][n], 0, 0, 0);
      };
      [[maybe_unused]] auto step_h = [&](auto QC) {
        constexpr int qs = decltype(QC)::value;
        if constexpr (qs + 1 < 28) fetch_h(std::integral_constant<int, qs + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        mult_h(QC);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto fetch = [&](auto QC) {
        constexpr int qs = decltype(QC)::value;
        constexpr int i = qs >> 2, t0 = (qs & 3) * 2, q = qs & 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int mt = t0 + t;
          const int z0 = mt >> 2, z1 = (mt >> 1) & 1, z2 = mt & 1;
          const int ro = RG[i] + ((z0 * H1 + z1) * H2 + z2) * RBL;
          fa[q][t] = *reinterpret_cast<const f16x8 *>(img[(z0 + z1 + z2) & 1] + ro);
          fl[q][t] = *reinterpret_cast<const f16x8 *>(iml[(z0 + z1 + z2) & 1] + ro);
        }
      };
      auto mult = [&](auto QC) {
        constexpr int qs = decltype(QC)::value;
        constexpr int i = qs >> 2, t0 = (qs & 3) * 2, q = qs & 1;
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[t0 + t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[q][t], bh[i][n], acc[t0 + t][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[t0 + t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[q][t], bl[i][n], acc[t0 + t][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[t0 + t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[q][t], bh[i][n], acc[t0 + t][n], 0, 0, 0);
      };
      [[maybe_unused]] auto step = [&](auto QC) {                 // quarter-slice qs: read qs + 1, multiply qs (18 MFMAs)
        constexpr int qs = decltype(QC)::value;
        if constexpr (qs + 1 < 28 && !(NF_H_ABL & 1)) fetch(std::integral_constant<int, qs + 1>{});
        __builtin_amdgcn_sched_barrier(0);
#if NF_H_ABL & 1
        asm volatile("" : "+v"(fa[qs & 1][0]), "+v"(fl[qs & 1][0]), "+v"(fa[qs & 1][1]), "+v"(fl[qs & 1][1]));
#endif
        mult(QC);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto store_slot = [&](auto KC) {
        constexpr int k = decltype(KC)::value;
        if (okk[k]) {
          const int ad = ptk[k] + (k == 0 ? 0 : ioff);
#pragma unroll
          for (int mt = 0; mt < 8; ++mt) {
            const f32x4 val = acc[mt][k];
#if NF_H_ABL & 16
            asm volatile("" ::"v"(ad), "a"(val));
#else
            if (mt < 4 || ntile == 8) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(ad), "a"(val), "n"(mt << 6) : "memory");
#endif
          }
        }
      };
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[mt][n] = n == 0 ? acc0 : f32x4{0.f, 0.f, 0.f, 0.f};
#if !(NF_H_ABL & 32)
      if (half) {
        fetch_h(std::integral_constant<int, 0>{});
        static_for<0, 28>(step_h);
      } else {
        if constexpr (!(NF_H_ABL & 1)) fetch(std::integral_constant<int, 0>{});
        static_for<0, 28>(step);
      }
#endif
      asm volatile("s_nop 15\n\ts_nop 3");      // the LDS instructions below read accumulators the compiler does not know they read: let the last MFMA land
      lds_barrier();            // B1: image m is consumed; the mover has read the logits of item m-1
      store_slot(std::integral_constant<int, 1>{});
      store_slot(std::integral_constant<int, 2>{});
      lds_barrier();            // B2: the partial sums of the other two waves for this wave's own column tile are in place
      // own tile (slot 0, still in the accumulators) + slot 1 of the previous wave + slot 2 of the one before: logits -> pt
      if (okk[0]) {
        typedef __attribute__((address_space(3))) f32x4 lds_q;
        const lds_f *pa = (const lds_f *)(smem_h + ioff + home);       // plane 1, this lane's channel and unit quad of site tile 0
        lds_f *po = (lds_f *)(smem_h + 2 * ITEM + home);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
          if (mt >= 4 && ntile == 4) break;
          const f32x4 s1 = *(const lds_q *)(pa + (mt << 4));
          const f32x4 s2 = *(const lds_q *)(pa + PT / 4 + (mt << 4));
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = ((acc[mt][0][r] + s1[r]) + s2[r]) * out_scale;      // the weights were packed scaled by 2^10 (and the input by in_scale)
          *(lds_q *)(po + (mt << 4)) = v;
        }
      }
      lds_barrier();            // Bs: the logits of item m are in pt
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
  // Two halo rows per pass: lanes 0-31 take the 32 interior sites of row 2i, lanes 32-63 those of row 2i+1; the two halo
  // sites of a row are copies of its own end sites (the box spans the axis).  Input either as 8 fp32 channel planes
  // (split here: 8 loads and ~40 conversions per site and pass) or already split by the producing layer into
  // channel-last fp16 pairs, 32 bytes per site (NF_CONV_SPLIT16_INPUT: two 16-byte loads, no arithmetic).
  constexpr int PB = 4;                          // passes per batch, two batches in flight
  const int rs = lane >> 5, xs = lane & 31;
  const bool pre = A.in_split16 != 0;       // input already split (flag carried in the high bits of dbg)
  auto put = [&](unsigned char *imgH, int row, int x3, const f16x8 &hi, const f16x8 &lo) {
    const int d = row * RBL + pair_row_offset(x3, SEGW);         // (whole-row segments only: the image row IS the pair row)
    *reinterpret_cast<f16x8 *>(imgH + d) = hi;
    *reinterpret_cast<f16x8 *>(imgH + 512 + d) = lo;
  };
  auto put3 = [&](unsigned char *imgH, int row, const f16x8 &hi, const f16x8 &lo) { put(imgH, row, xs, hi, lo); };
  auto stage = [&](int b, const int (&o)[4], unsigned char *imgH) {
    const int myoff = row_offsets(o);
    if (pre) {
      const unsigned char *__restrict__ src = static_cast<const unsigned char *>(A.in) + int64_t(b) * A.V * 32 + pair_row_offset(xs, SEGW);
      f16x8 h0[PB], l0[PB], h1[PB], l1[PB];
      auto issue = [&](f16x8 (&h)[PB], f16x8 (&l)[PB], int p0) {
#pragma unroll
        for (int j = 0; j < PB; ++j) {
          const int oa = __builtin_amdgcn_readlane(myoff, 2 * (p0 + j));
          const int ob = __builtin_amdgcn_readlane(myoff, 2 * (p0 + j) + 1);
          const unsigned char *q = src + int64_t(rs ? ob : oa) * 32;
          h[j] = *reinterpret_cast<const f16x8 *>(q);
          l[j] = *reinterpret_cast<const f16x8 *>(q + 512);
        }
      };
      auto commit = [&](const f16x8 (&h)[PB], const f16x8 (&l)[PB], int p0) {
#pragma unroll
        for (int j = 0; j < PB; ++j) put3(imgH, 2 * (p0 + j) + rs, h[j], l[j]);
      };
      issue(h0, l0, 0);
#pragma unroll 1
      for (int p0 = 0; p0 < NROW / 2; p0 += 2 * PB) {
        issue(h1, l1, p0 + PB);
        commit(h0, l0, p0);
        if (p0 + 2 * PB < NROW / 2) issue(h0, l0, p0 + 2 * PB);
        commit(h1, l1, p0 + PB);
      }
      return;
    }
    const float *__restrict__ src = in + int64_t(b) * 8 * A.V + xs;
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
          const float sv = v[j][c] * in_scale;
          const _Float16 hh = static_cast<_Float16>(sv);
          hi[c] = hh;
          lo[c] = static_cast<_Float16>(sv - static_cast<float>(hh));
        }
        put3(imgH, 2 * (p0 + j) + rs, hi, lo);
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
  auto is_half = [&](const int (&o)[4]) { return has_half && o[3] == SEGW * (nsg - 1); };
  // unit u of the logit scratch -> box row (z0, z1, z2) and pair p of the segment (full item: u = 16 (4 z0 + 2 z1 + z2) + p;
  // half item: u = 16 (2 z0 + z1) + 8 z2 + p, 64 units)
  auto unit_row = [&](bool hf, int u, int &z0, int &z1, int &z2, int &p) {
    const int mt = u >> 4;
    z0 = hf ? (mt >> 1) & 1 : mt >> 2;
    z1 = hf ? mt & 1 : (mt >> 1) & 1;
    z2 = hf ? (u >> 3) & 1 : mt & 1;
    p = hf ? u & 7 : u & 15;
  };
  auto pair_of = [&](int b, const int (&o)[4], int pass, bool &ok) {
    const bool hf = is_half(o);
    int z0, z1, z2, p;
    unit_row(hf, pass * 64 + lane, z0, z1, z2, p);
    const int p3 = p + (o[3] >> 1);                               // pair of the row: 16 hs + p
    ok = hf ? pass == 0 : p3 < HP;                                // (a half item has 64 units; without packing a partial segment has 8 of 16)
    const int x0 = o[0] + z0, x1 = o[1] + z1, x2 = o[2] + z2;
    return int64_t(b) * (A.V / 2) + ((int64_t(x0) * A.L[1] + x1) * A.L[2] + x2) * HP + (ok ? p3 : 0);
  };
  auto unit_parity = [&](const int (&o)[4], int u) {             // which site of unit u's pair is the active one
    int z0, z1, z2, p;
    unit_row(is_half(o), u, z0, z1, z2, p);
    return (A.parity + o[0] + z0 + o[1] + z1 + o[2] + z2) & 1;
  };
  float2 xpre[2] = {{0.f, 0.f}, {0.f, 0.f}};     // the x pairs of the item whose logits are (about to be) in pt
  float2 gpre[2] = {{0.f, 0.f}, {0.f, 0.f}};     // FUSE 4 / 5: and the pairs of the value's cotangent
  auto prefetch_x = [&](int b, const int (&o)[4]) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      if (pass && is_half(o)) break;             // (a half item has one pass of units)
      bool ok;
      const int64_t pr = pair_of(b, o, pass, ok);
      xpre[pass] = load_field_pair(A, pr);
      if constexpr (FUSE >= 4) gpre[pass] = reinterpret_cast<const float2 *>(A.gyout)[pr];
    }
  };
  auto epilogue = [&](int b, const int (&o)[4], int64_t pidx) {
    const lds_f *ptl = (const lds_f *)(smem_h + 2 * ITEM);
    if constexpr (FUSE == 3) {
      // no coupling: the logits themselves, pair-compact (B, C, V/2) -- the form the training path differentiates
      // (nf_conv_last_logits_split16)
      float *outp = static_cast<float *>(A.out);
      const int64_t Vh = A.V / 2;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        if (pass && is_half(o)) break;
        const int u = pass * 64 + lane;
        bool pok;
        const int64_t pair = pair_of(b, o, pass, pok);
        if (pok) {
          float *d = outp + (int64_t(b) * A.cout) * Vh + (pair - int64_t(b) * Vh);
          if (A.accumulate) {                  // (a second group of 8 input channels adds to the first one's logits:
            float old[C];                      //  all loads in flight, then the stores)
#pragma unroll
            for (int c = 0; c < C; ++c) old[c] = c < A.cout ? d[int64_t(c) * Vh] : 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c)
              if (c < A.cout) d[int64_t(c) * Vh] = old[c] + ptl[c * PTS + u];
          } else if (A.cout == C) {
#pragma unroll
            for (int c = 0; c < C; ++c) d[int64_t(c) * Vh] = ptl[c * PTS + u];
          } else {
            for (int c = 0; c < A.cout; ++c) d[int64_t(c) * Vh] = ptl[c * PTS + u];
          }
        }
      }
      (void)pidx;
      return;
    }
    if constexpr (FUSE >= 4) {
      // VJP of the fused layer's coupling (training: Fitter.step differentiates every layer, src/_normflowcore.py:275-294):
      // the logits were recomputed by the matrix-core part above and sit in pt; the site's cotangents (value, log-det)
      // go back through the spline (rqs_site_vjp: segment, softmax-cumsum, softplus) -- the logit column is overwritten by
      // its cotangent and leaves pair-compact (B, C, V/2), the form the weight- and input-gradient kernels read; the
      // field's cotangent goes where the forward pass writes y.  FUSE 4: of the forward map, 5: of the inverse.
      float *outp = static_cast<float *>(A.out);
      const int64_t Vh = A.V / 2;
      const float gl = A.glogj[b];
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {       // (both passes in flight: two independent dependent chains)
        if (pass && is_half(o)) break;
        const int u = pass * 64 + lane;
        const int offp = unit_parity(o, u);
        bool pok;
        const int64_t pair = pair_of(b, o, pass, pok);
        const float2 xv = xpre[pass], gv = gpre[pass];
        float *d = outp + (int64_t(b) * A.cout) * Vh + (pair - int64_t(b) * Vh);
        float gin;
        if (A.P.m == M) {            // knots_len 16: the column in registers, cotangents straight from them to memory
          RegCol<float, C> colr;
#pragma unroll
          for (int c = 0; c < C; ++c) colr[c] = ptl[c * PTS + u];
          gin = rqs_site_vjp<float, M, FUSE == 5>(colr, A.P, offp ? xv.y : xv.x, offp ? gv.y : gv.x, gl);
          if (pok) {
#pragma unroll
            for (int c = 0; c < C; ++c) d[int64_t(c) * Vh] = colr[c];
          }
        } else {                     // any other knots_len: the generic form on the LDS column
          LdsCol<float> colv{reinterpret_cast<float *>(const_cast<unsigned char *>(smem_h + 2 * ITEM)) + u, PTS};
          gin = rqs_site_vjp<float, 0, FUSE == 5>(colv, A.P, offp ? xv.y : xv.x, offp ? gv.y : gv.x, gl);
          if (pok)
            for (int c = 0; c < A.cout; ++c) d[int64_t(c) * Vh] = ptl[c * PTS + u];
        }
        if (pok) {
          float2 ov;
          ov.x = offp ? 0.f : gin;
          ov.y = offp ? gin : 0.f;
          reinterpret_cast<float2 *>(A.yout)[pair] = ov;
        }
      }
      (void)pidx;
      return;
    }
    double lacc = 0.0;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      if (pass && is_half(o)) break;
      const int u = pass * 64 + lane;
      const int offp = unit_parity(o, u);
      bool pok;
      const int64_t pair = pair_of(b, o, pass, pok);
      const float2 xv = xpre[pass];
      float val, logd;
#if NF_H_EPI
      if (A.P.m == M) {
        rqs_site_pt<FUSE == 2>(ptl + u, A.P, offp ? xv.y : xv.x, val, logd);
      } else {
        // any other knots_len (2..15): the generic site function on the logit column where it lies (run-time m; the
        // softmax numerators overwrite the consumed logits) -- fewer channels than the m = 16 the kernel is tuned for
        LdsCol<float> colm{reinterpret_cast<float *>(const_cast<unsigned char *>(smem_h + 2 * ITEM)) + u, PTS};
        rqs_site<float, 0, FUSE == 2>(colm, A.P, offp ? xv.y : xv.x, val, logd);
      }
#else
      RegCol<float, C> a;
#pragma unroll
      for (int c = 0; c < C; ++c) a[c] = ptl[c * PTS + u];
      rqs_site<float, M, FUSE == 2>(a, A.P, offp ? xv.y : xv.x, val, logd);
#endif
      float2 ov;
      ov.x = offp ? 0.f : val;
      ov.y = offp ? val : 0.f;
      if (pok) {
        store_field_pair(A, pair, ov);
        lacc += double(logd);
      }
    }
    const double tot = wave_sum(lacc);
    if (lane == 0) A.partial[pidx] = tot;
  };

  int pb = 0, po[4] = {0, 0, 0, 0};            // the item whose logits sit in pt
  int64_t pslot = 0, cslot;                    // ... and its place among the log-det partials
  int cb, co4[4];
  Item n1;
  decode(l0, n1);
  cb = n1.b;
  coords(n1, co4);
  cslot = slot_of(n1);
  if (!pre) stage(cb, co4, smem_h);            // (pre-split input: by LDS-DMA below, once its helpers exist)
  // Pre-split input (the pipeline's path): the next item's image is brought in by LDS-DMA, one halo row = one 1 KiB piece =
  // one wave-instruction (the pair tensor's rows ARE the image rows): no staging registers, no ds_write -- the mover's 128
  // 16-byte LDS stores per item used to take ~13 % of the kernel from the compute waves' fragment reads.
  const unsigned lds0m = unsigned(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char *)smem_h));
  const unsigned lane16 = unsigned(lane * 16);
  // Source of halo row (z0, z1, z2): sample base + 32 (x0 L1 L2 L3 + x1 L2 L3 + x2 L3) bytes -- separable, so an item costs
  // 16 scalar bases (z0, z1) and 4 per-lane offsets (z2, with the lane's place in the row folded in) and a row costs
  // nothing but its LDS address and the instruction (4 scalar instructions where a row offset read out of a lane and a
  // 64-bit address built per row were 8).
  bool dma_half = false;                         // the item being staged is a half item: slots 0 .. 8 of its rows are all it reads
  const unsigned char *dma_ab[H0 * H1];          // uniform
  unsigned dma_v[H2], dma_v2[H2];               // per lane: main piece / (SEGM) the 17th slots, lanes 0..3
  auto dma_open = [&](int b, const int (&o)[4]) {
    dma_half = is_half(o);
    const unsigned char *base = static_cast<const unsigned char *>(A.in) + int64_t(b) * A.V * 32;
    unsigned lane_d = lane16, lane_d2 = 0;       // per-lane source offsets inside a row of the pair tensor
    if (SEGM) {
      const int hs16 = o[3] >> 1;                // first pair of the segment = 16 hs
      const int blk = lane >> 4;                 // piece 1: block (hl = blk >> 1, parity = blk & 1), local slot lane & 15
      int gs = hs16 + (lane & 15);
      gs = gs >= HP ? gs - HP : gs;
      lane_d = unsigned((blk >> 1) * HB + (blk & 1) * PBK + gs * 16);
      int g17 = hs16 + 16;                       // piece 2 (lanes 0..3): block = lane, the 17th slot
      g17 = g17 >= HP ? g17 - HP : g17;
      lane_d2 = unsigned(((lane >> 1) & 1) * HB + (lane & 1) * PBK + g17 * 16);
    }
    const unsigned s2 = unsigned(A.L[3]) * 32u, s1 = s2 * unsigned(A.L[2]), s0 = s1 * unsigned(A.L[1]);
    unsigned c0[H0], c1[H1];
#pragma unroll
    for (int z = 0; z < H0; ++z) {
      int x = o[0] + z - 1;
      x = x < 0 ? x + A.L[0] : (x >= A.L[0] ? x - A.L[0] : x);
      c0[z] = unsigned(x) * s0;
    }
#pragma unroll
    for (int z = 0; z < H1; ++z) {
      int x = o[1] + z - 1;
      x = x < 0 ? x + A.L[1] : (x >= A.L[1] ? x - A.L[1] : x);
      c1[z] = unsigned(x) * s1;
    }
#pragma unroll
    for (int z = 0; z < H0 * H1; ++z) dma_ab[z] = base + (c0[z / H1] + c1[z % H1]);
#pragma unroll
    for (int z = 0; z < H2; ++z) {
      int x = o[2] + z - 1;
      x = x < 0 ? x + A.L[2] : (x >= A.L[2] ? x - A.L[2] : x);
      dma_v[z] = lane_d + unsigned(x) * s2;
      dma_v2[z] = lane_d2 + unsigned(x) * s2;
    }
  };
  auto dma_rows = [&](unsigned buf, auto R0, auto R1) {     // halo rows [R0, R1) -> image buffer at LDS byte address buf
    constexpr int r0 = decltype(R0)::value, r1 = decltype(R1)::value;
    if (SEGM && dma_half) {
      if ((lane & 15) <= 8) {
#pragma unroll
        for (int i = r0; i < r1; ++i) dma_row(dma_ab[i / H2], dma_v[i % H2], buf + unsigned(i * RBL));
      }
      return;
    }
#pragma unroll
    for (int i = r0; i < r1; ++i) dma_row(dma_ab[i / H2], dma_v[i % H2], buf + unsigned(i * RBL));
    if (SEGM) {
      if (lane < 4) {
#pragma unroll
        for (int i = r0; i < r1; ++i) dma_row(dma_ab[i / H2], dma_v2[i % H2], buf + unsigned(i * RBL + 1024));
      }
    }
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, NF_H_DMA_EARLY> IH;     // rows issued before B1 (the rest behind it)
  typedef std::integral_constant<int, NROW> IN;
  if (pre) {                                    // image 0
    dma_open(cb, co4);
    dma_rows(lds0m, I0{}, IN{});
    wait_vm<0>();
  }
  int n1b = cb, n1o[4] = {co4[0], co4[1], co4[2], co4[3]};      // item m + 1
  int64_t n1slot = cslot;
  if (n_my > 1) {
    advance(n1);
    n1b = n1.b;
    coords(n1, n1o);
    n1slot = slot_of(n1);
  }
  if (FUSE != 3) prefetch_x(cb, co4);
  lds_barrier();                                // P: image 0 ready
  // n_my + 1 rounds, the last one only the epilogue of the last item: ONE copy of the epilogue in the code, so that every
  // item's spline runs through the same instructions whatever its place in the workgroup's sequence.  (With a second,
  // separately inlined copy behind the loop the compiler is free to contract a * b + c differently in the two, and the last
  // bit of log|J| then depends on the batch order: seen in an experiment of round 2)
  for (int m = 0; m <= n_my; ++m) {
    if (m > 0 && !NF_DBG(A, 128)) epilogue(pb, po, pslot);     // dbg 128: timing ablation
    if (m == n_my) break;
    if (FUSE != 3) prefetch_x(cb, co4);         // for the epilogue of item m, one iteration from now
    // image m+1 goes where image m-1 and then the partial sums of item m-1 were: after the epilogue above, all rows ahead of
    // B1 (they need the whole MFMA phase and the add-up behind it to land)
    const bool more = m + 1 < n_my && !NF_DBG(A, 64);          // dbg 64: timing ablation
    const unsigned nbuf = lds0m + unsigned(((m + 1) & 1) * ITEM);
    if (more) {
      if (pre) {
        dma_open(n1b, n1o);
        dma_rows(nbuf, I0{}, IH{});
      } else {
        stage(n1b, n1o, smem_h + ((m + 1) & 1) * ITEM);
      }
    }
    lds_barrier();                              // B1: image m is consumed, pt is free
    if (more && pre) dma_rows(nbuf, IH{}, IN{});
    pb = cb;
    cb = n1b;
    pslot = cslot;
    cslot = n1slot;
#pragma unroll
    for (int mu = 0; mu < 4; ++mu) { po[mu] = co4[mu]; co4[mu] = n1o[mu]; }
    if (m + 2 < n_my) {
      advance(n1);
      n1b = n1.b;
      coords(n1, n1o);
      n1slot = slot_of(n1);
    }
    lds_barrier();                              // B2: the partial sums of item m are in pt and in image m
    if (pre) wait_vm<0>();                      // image m+1 has landed (a whole MFMA phase after its first row was issued)
    lds_barrier();                              // Bs: the compute waves have added them up: logits of item m in pt
  }
}

// Is this fused layer the split-fp16 kernel's, and with how many boxes per sample?  (The kernel has its own box: 2 x 2 x 2
// lattice rows x one 32-site segment of the fastest axis.)
int conv_h_eligible(const ConvArgs &A, int fuse, int64_t *nboxes) {
  using namespace h;
  if (!option(NF_OPT_SPLIT16) || !fuse) return 0;
  if (A.cin != 8 || A.cout < 1 || A.cout > C || NF_DBG(A, 15) || NF_STAMPS(A)) return 0;
  if (fuse != 3 && (A.P.m < 2 || A.P.m > M || A.cout != 3 * A.P.m - 2 || A.P.fx || A.P.fy)) return 0;
  for (int mu = 0; mu < 4; ++mu)
    if (A.k[mu] != 3) return 0;
  if (A.L[3] < 32 || (A.L[3] & 15)) return 0;                // whole or half segments
  if (A.L[3] != 32 && !A.in_split16) return 0;               // fp32 planes are staged by the whole-row fallback only
  if (A.in_split16 && A.V * 32 >= (int64_t(1) << 32)) return 0;       // the mover's row offsets inside a sample are 32-bit
  for (int mu = 0; mu < 3; ++mu)
    if (A.L[mu] < 2 || (A.L[mu] & 1)) return 0;             // even extents: whole boxes, row parity independent of the box
  if (nboxes) *nboxes = int64_t(A.L[0] / 2) * (A.L[1] / 2) * (A.L[2] / 2) * ((A.L[3] / 2 + 15) / 16);
  return 1;
}

// 1 = launched (dry: would launch), 0 = not this kernel's layer, < 0 error.  A0 is nf_conv.hip's planned argument block.
int launch_conv_h(const ConvArgs &A0, int64_t B, int fuse, hipStream_t stream, bool dry) {
  using namespace h;
  int64_t nboxes = 0;
  if (!conv_h_eligible(A0, fuse, &nboxes)) return 0;
  if (dry) return 1;
  ConvArgs A = A0;
  const bool segm = A.L[3] != 32;
  A.box[0] = A.box[1] = A.box[2] = 2; A.box[3] = SEGW;
  for (int mu = 0; mu < 3; ++mu) A.nbox[mu] = A.L[mu] / 2;
  A.nbox[3] = (A.L[3] / 2 + 15) / 16;
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
  auto go = [&](auto kern, int lds) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1;
    hipLaunchKernelGGL(kern, dim3(unsigned(grid)), dim3(256), lds, stream, A);
    return 1;
  };
  if (fuse == 3) return segm ? go(&conv_h_kernel<3, true>, lds_bytes<true>()) : go(&conv_h_kernel<3, false>, lds_bytes<false>());
  if (fuse == 4) return segm ? go(&conv_h_kernel<4, true>, lds_bytes<true>()) : go(&conv_h_kernel<4, false>, lds_bytes<false>());
  if (fuse == 5) return segm ? go(&conv_h_kernel<5, true>, lds_bytes<true>()) : go(&conv_h_kernel<5, false>, lds_bytes<false>());
  if (fuse == 1) return segm ? go(&conv_h_kernel<1, true>, lds_bytes<true>()) : go(&conv_h_kernel<1, false>, lds_bytes<false>());
  return segm ? go(&conv_h_kernel<2, true>, lds_bytes<true>()) : go(&conv_h_kernel<2, false>, lds_bytes<false>());
}

}  // namespace nf

using namespace nf;

// The last conv layer 8 -> 46 of a spline coupling's net at the active sites, on the split-fp16 kernel, with the LOGITS written
// out pair-compact (B, 46, V/2) instead of consumed by the coupling: the forward pass of a training step, whose autograd
// differentiates the spline separately (reference: src/nn/scalar/modules.py:120-145 under Fitter.step).  in: fp32 channel
// planes (B, 8, V) (fastest axis of 32 sites) or, with in_split16, the (B, V, 16) pair tensor; absmax_bits (or NULL): the
// input is / gets scaled by the matching power of two (nf_absmax_bits), the logits are descaled.  wsplit: NF_WLAYOUT_SPLIT16.
static int last_logits_split16(const void *in, int in_split16, const void *wsplit, const void *bias, void *logits, int64_t B,
                               const int32_t *lattice, int active_parity, const void *absmax_bits, int accumulate, int cout, void *stream_);

extern "C" int nf_conv_last_logits_split16(const void *in, int in_split16, const void *wsplit, const void *bias, void *logits,
                                           int64_t B, const int32_t *lattice, int active_parity, const void *absmax_bits,
                                           void *stream_) {
  return last_logits_split16(in, in_split16, wsplit, bias, logits, B, lattice, active_parity, absmax_bits, 0, h::C, stream_);
}

// ... the same for any cout <= 46 (logits (B, cout, V/2)), ADDED to the logits already there when `accumulate`: the second group of
// 8 input channels of a 16 -> cout layer (hidden width 16 on the split-fp16 kernels: normflow__amd/_hip.py, conv_wide_logits_split16)
extern "C" int nf_conv_last_logits_split16_acc(const void *in, int in_split16, const void *wsplit, const void *bias, void *logits,
                                               int64_t B, const int32_t *lattice, int active_parity, const void *absmax_bits,
                                               int accumulate, int cout, void *stream_) {
  NF_REQUIRE(cout >= 1 && cout <= h::C, "nf_conv_last_logits_split16_acc: cout outside [1, 46]");
  return last_logits_split16(in, in_split16, wsplit, bias, logits, B, lattice, active_parity, absmax_bits, accumulate, cout, stream_);
}

static int last_logits_split16(const void *in, int in_split16, const void *wsplit, const void *bias, void *logits, int64_t B,
                               const int32_t *lattice, int active_parity, const void *absmax_bits, int accumulate, int cout, void *stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  NF_REQUIRE(in && wsplit && logits && lattice, "nf_conv_last_logits_split16: NULL pointer");
  NF_REQUIRE(B >= 0, "nf_conv_last_logits_split16: negative batch");
  if (B == 0) return NF_OK;
  ConvArgs A{};
  A.V = 1;
  for (int mu = 0; mu < 4; ++mu) { A.L[mu] = lattice[mu]; A.k[mu] = 3; A.V *= lattice[mu]; }
  A.in = in; A.wfrag = wsplit; A.bias = bias; A.out = logits;
  A.cin = 8; A.cout = cout;
  A.parity = active_parity & 1;
  A.in_split16 = in_split16 ? 1 : 0;
  A.gscale_bits = static_cast<const unsigned *>(absmax_bits);
  A.accumulate = accumulate ? 1 : 0;
  const int pr = launch_conv_h(A, B, 3, stream, false);
  if (pr == 0) {
    set_error("nf_conv_last_logits_split16: layer not supported (4-D lattice, even extents, fastest axis 32 + 16 n sites; fp32 planes need 32)");
    return NF_EINVAL;
  }
  if (pr == -2) { set_error("nf_conv_last_logits_split16: batch x boxes >= 2^31 work items, split the batch"); return NF_EINVAL; }
  if (pr != 1) { set_error("nf_conv_last_logits_split16: could not launch the kernel"); return NF_ELAUNCH; }
  return check_launch("conv split-fp16 logits kernel");
}

// The fused last layer + RQ-spline coupling as a DIFFERENTIABLE node of a training step (reference: Fitter.step,
// src/_normflowcore.py:275-294, differentiates src/nn/scalar/couplings_.py:178-200 through autograd; here the logits of the
// layer never exist in memory, forward or backward).  Shared argument block of the forward and the VJP entry.
static int rqs_train_args(ConvArgs &A, const char *who, const void *in, int in_split16, const void *wsplit, const void *bias,
                          int cout, const void *x, int64_t B, const int32_t *lattice, int active_parity,
                          const void *absmax_bits, const nf_rqs_opts *opts) {
  NF_REQUIRE(in && wsplit && x && lattice && opts, "%s: NULL pointer", who);
  NF_REQUIRE(B >= 0, "%s: negative batch", who);
  NF_REQUIRE(!opts->fixed_knots_x && !opts->fixed_knots_y, "%s: fixed knots are not fused", who);
  NF_REQUIRE(opts->xhi > opts->xlo && opts->yhi > opts->ylo, "%s: empty xlim/ylim", who);
  A.V = 1;
  for (int mu = 0; mu < 4; ++mu) { A.L[mu] = lattice[mu]; A.k[mu] = 3; A.V *= lattice[mu]; }
  A.in = in; A.wfrag = wsplit; A.bias = bias;
  A.cin = 8; A.cout = cout;
  A.parity = active_parity & 1;
  A.in_split16 = in_split16 ? 1 : 0;
  A.gscale_bits = static_cast<const unsigned *>(absmax_bits);
  A.xact = static_cast<const float *>(x);
  A.P.xlo = opts->xlo; A.P.xhi = opts->xhi; A.P.ylo = opts->ylo; A.P.yhi = opts->yhi;
  A.P.fx = nullptr; A.P.fy = nullptr; A.P.m = opts->m; A.P.el = opts->extrap_left; A.P.er = opts->extrap_right;
  return NF_OK;
}

extern "C" int nf_conv_rqs_split16_train(const void *in, int in_split16, const void *wsplit, const void *bias, int cout,
                                         const void *x_active, const void *log0, void *y, void *logj, int64_t B,
                                         const int32_t *lattice, int active_parity, const void *absmax_bits,
                                         const nf_rqs_opts *opts, int inverse, void *workspace, size_t workspace_bytes,
                                         void *stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ConvArgs A{};
  const int rc = rqs_train_args(A, "nf_conv_rqs_split16_train", in, in_split16, wsplit, bias, cout, x_active, B, lattice,
                                active_parity, absmax_bits, opts);
  if (rc) return rc;
  NF_REQUIRE(y && logj, "nf_conv_rqs_split16_train: NULL output");
  if (B == 0) return NF_OK;
  A.yout = static_cast<float *>(y);
  A.partial = static_cast<double *>(workspace);
  const int fuse = inverse ? 2 : 1;
  int64_t nboxes = 0;
  if (!conv_h_eligible(A, fuse, &nboxes)) {
    set_error("nf_conv_rqs_split16_train: layer not supported (4-D lattice, even extents, fastest axis 32 + 16 n sites -- fp32 planes need 32 --, knots_len 2..16, cout = 3m-2)");
    return NF_EINVAL;
  }
  NF_REQUIRE(workspace && workspace_bytes >= size_t(B) * size_t(nboxes) * sizeof(double),
             "nf_conv_rqs_split16_train: workspace %zu B < %zu B needed", workspace_bytes, size_t(B) * size_t(nboxes) * sizeof(double));
  const int pr = launch_conv_h(A, B, fuse, stream, false);
  if (pr == -2) { set_error("nf_conv_rqs_split16_train: batch x boxes >= 2^31 work items, split the batch"); return NF_EINVAL; }
  if (pr != 1) { set_error("nf_conv_rqs_split16_train: could not launch the kernel"); return NF_ELAUNCH; }
  const int rl = check_launch("conv split-fp16 training forward kernel");
  if (rl) return rl;
  return launch_finalize<float>(A.partial, nboxes, log0, logj, B, stream);
}

extern "C" int nf_conv_rqs_split16_vjp(const void *in, int in_split16, const void *wsplit, const void *bias, int cout,
                                       const void *x_point, const void *grad_y, const void *grad_logj, void *grad_logits,
                                       void *grad_x, int64_t B, const int32_t *lattice, int active_parity,
                                       const void *absmax_bits, const nf_rqs_opts *opts, int inverse, void *stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ConvArgs A{};
  const int rc = rqs_train_args(A, "nf_conv_rqs_split16_vjp", in, in_split16, wsplit, bias, cout, x_point, B, lattice,
                                active_parity, absmax_bits, opts);
  if (rc) return rc;
  NF_REQUIRE(grad_y && grad_logj && grad_logits && grad_x, "nf_conv_rqs_split16_vjp: NULL pointer");
  if (B == 0) return NF_OK;
  A.gyout = static_cast<const float *>(grad_y);
  A.glogj = static_cast<const float *>(grad_logj);
  A.out = grad_logits;
  A.yout = static_cast<float *>(grad_x);
  const int fuse = inverse ? 5 : 4;
  if (!conv_h_eligible(A, fuse, nullptr)) {
    set_error("nf_conv_rqs_split16_vjp: layer not supported (4-D lattice, even extents, fastest axis 32 + 16 n sites -- fp32 planes need 32 --, knots_len 2..16, cout = 3m-2)");
    return NF_EINVAL;
  }
  const int pr = launch_conv_h(A, B, fuse, stream, false);
  if (pr == -2) { set_error("nf_conv_rqs_split16_vjp: batch x boxes >= 2^31 work items, split the batch"); return NF_EINVAL; }
  if (pr != 1) { set_error("nf_conv_rqs_split16_vjp: could not launch the kernel"); return NF_ELAUNCH; }
  return check_launch("conv split-fp16 coupling VJP kernel");
}
