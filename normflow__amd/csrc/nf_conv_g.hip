// nf_conv_g.hip -- K5g: a hidden 8 -> 8 ConvAct layer (3^4 kernel, tanh / logistic) between two split-fp16 consumers, fp16
// (hi, lo) pair tensors in and out (include/normflow_hip.h, nf_conv_fwd_split16; reference: one Conv4d + activation of
// src/nn/scalar/modules.py:120-145, src/nn/scalar/convNd.py:86-126).
//
// Arithmetic as in nf_conv_h.hip: every fp32 product = three fp16 matrix-core products (a_hi w_hi + a_hi w_lo + a_lo w_hi,
// fp32 accumulation, v_mfma_f32_16x16x32_f16).  Two-site columns (column n = 8*shift + co = channel co at site 2p + shift)
// make the four fastest-axis taps -1..+2 of a site pair the four k-groups of one MFMA: a kernel row (j0, j1, j2) is exactly
// one K = 32 slice, 27 slices per output tile (a tile = the 16 site pairs of one lattice row).
//
// Shape of the computation (what changed against the one-box-per-item kernel it replaces, and why):
//   * MARCHING COLUMNS.  A persistent workgroup owns a 2 x 2 cross-section (axes 0, 1) and marches along axis 2, two
//     lattice planes per step.  The halo planes (4 x 4 rows of L3 sites, 16 KiB) live in an 8-slot LDS ring: a step reads
//     four of them, and only the two new ones are loaded -- 4 rows in per row out instead of 6 (2x2x4 boxes) or 8.5.
//   * LDS-DMA.  The pair tensor is row-major with exactly the LDS image's row layout, so a halo row is one
//     global_load_lds_dwordx4 wave-instruction (1 KiB, fully coalesced): no staging registers, no ds_write, and the copy
//     of the planes for step s+2 flies while steps s and s+1 multiply (counted vmcnt, LDS-only barriers).
//   * TWO WAVES PER SIMD, K SPLIT BETWEEN THEM.  The freed registers let 8 waves fit (<= 256 VGPRs each).  Waves w ("A") and
//     w+4 ("B") share a SIMD and the two output tiles of position (z0, z1) of the cross-section; A multiplies slices 0..13,
//     B slices 14..26, for BOTH tiles.  The partial sums cross through a 2 KiB LDS exchange (A gets tile 0's, B tile 1's),
//     double-buffered, and each wave runs the epilogue of ITS tile of step s (add, bias, activation, split into (hi, lo),
//     store -- straight from the accumulators: the weights are the MFMA's A operand, so a lane holds 4 channels of one site) at the start of step s+1 -- beside its partner's MFMAs.  While one wave of a SIMD waits or does vector
//     work, the other's MFMAs keep the matrix pipe busy; the one-wave-per-SIMD form spent 42 % of an item outside its MFMAs.
//   * ONE barrier per step; scalar bookkeeping per step is a handful of instructions (addresses are advanced, not recomputed:
//     with 64-bit multiplications per DMA piece the pieces cost ~450 cycles each and the waves were scalar-bound).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "nf_conv_core.h"

namespace nf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#if !defined(NF_DIAG) || !defined(NF_G2_ABL)
#undef NF_G2_ABL
#define NF_G2_ABL 0     // timing ablations (diagnostic builds only): 1 no ring refill, 2 no epilogue, 4 no MFMAs, 8 no fragment reads
#endif

namespace g2 {
// A lattice row of L3 sites is cut into SEGMENTS of 32 sites = 16 pairs = one MFMA tile; a column of the march is
// (cross-section, segment).  The LDS image of a halo row of a segment holds the 17 slots 16 h .. 16 h + 16 (mod L3 / 2) of both
// parity blocks of the pair tensor's row, hi and lo: [hi | lo][even | odd][16 slots] (1 KiB, one DMA piece) + the 17th slots
// (64 B, a second, 4-lane piece).  SEGM = false is the L3 = 32 case: one segment is the whole periodic row, slot 16 IS slot 0,
// no second piece (the taps wrap by address).
template <bool SEGM>
struct Geo {
  static constexpr int RBL = SEGM ? 1088 : 1024;            // bytes of a row image in LDS
  static constexpr int PLROWS = 16;                         // halo rows of a plane: (z0, z1) in {-1..2}^2, row index 4*hz0 + hz1
  static constexpr int PLANE = PLROWS * RBL;                // 16 / 17 KiB
  static constexpr int NSLOT = 8;                           // ring: 4 planes being read + 4 ahead
  static constexpr int RING = NSLOT * PLANE;                // 128 / 136 KiB
  static constexpr int XBUF = 4 * 2 * 64 * 16;              // partial-sum exchange: [wave pair][direction][lane] x 16 B, one of two buffers
  static constexpr int LDS_BYTES = RING + 2 * XBUF;         // 147456 / 155648 <= 163840
  static_assert(LDS_BYTES <= 160 * 1024, "ring + exchange must fit the CU's LDS");
};
#ifndef NF_G2_PRIO
#define NF_G2_PRIO 1
#endif
constexpr int NSA = 14;                       // slices of the A waves (0..13); B waves: 14..26
constexpr float kInvWScale = 1.0f / 1024.0f;  // the weights are packed scaled by 2^10 (normflow__amd/_hip.py: SPLIT16_WEIGHT_SCALE)
}  // namespace g2

// tanh(v) for v = a * scale + bias given as (a, c1 = 2 log2(e) scale, c0 = 2 log2(e) bias): 1 - 2 / (1 + 2^(c1 a + c0)); five
// instructions, two of them transcendental; absolute error ~1e-7 (the rounding of a number near 1), exact limits at +-inf.
__device__ __forceinline__ float tanh_affine(float a, float c1, float c0) {
  const float t = __builtin_amdgcn_exp2f(__builtin_fmaf(a, c1, c0));
  return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + t), 1.0f);
}
// The two activations of the chain in ONE branch-free form: al / (1 + 2^(c1 a + c0)) + ga -- tanh: (al, ga) = (-2, 1) and
// c = 2 log2(e) (scale, bias); logistic: (1, 0) and c = -log2(e) (scale, bias).  (A select on the run-time activation code
// around every element had become a tree of scalar branches in the epilogue.)
__device__ __forceinline__ float act_affine(float a, float c1, float c0, float al, float ga) {
  const float t = __builtin_amdgcn_exp2f(__builtin_fmaf(a, c1, c0));
  return __builtin_fmaf(al, __builtin_amdgcn_rcpf(1.0f + t), ga);
}

// EPI = 0: bias + activation -> the pair tensor (a hidden layer).  EPI = 1 / 2: the LAST layer of an affine coupling's net
// (output channel 0 = t, 1 = s; the other six columns carry zero weights) fused with the coupling itself
// (src/nn/scalar/couplings_.py:123-139): y = t + x e^{-|s|} (1) or x = (y - t) e^{|s|} (2) at the active site of every pair,
// 0 at the frozen one, and the per-sample log-det partials -- the (B, 2, V/2) parameter tensor never exists in memory.
template <bool SEGM, int EPI, bool HK = false>
__global__ __launch_bounds__(512, 2) void conv_g2_kernel(ConvArgs A) {
  using namespace g2;
  typedef Geo<SEGM> G;
  constexpr int RBL = G::RBL, PLANE = G::PLANE, NSLOT = G::NSLOT, RING = G::RING, XBUF = G::XBUF;
  extern __shared__ __align__(16) unsigned char smem_g2[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool isB = wave >= 4;                 // A waves: slices 0..13, epilogue of tile 0; B waves: slices 14..26, tile 1
  const int q = wave & 3;                     // wave pair = position (z0, z1) = (q >> 1, q & 1) in the cross-section
  const int g = lane >> 4, p = lane & 15;
  const unsigned lds0 = unsigned(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char *)smem_g2));

  // ---- my columns.  Global column id gc = sample * ncol + (i0 * n1 + i1); groups of 32 consecutive columns go to one XCD
  // (blockIdx & 7: consecutive workgroups sit on consecutive XCDs), member j = blockIdx >> 3 of a group to workgroup j of that
  // XCD: the 32 columns an XCD marches at the same time are neighbours and share their halo rows in its L2.
  const int n0 = A.L[0] >> 1, n1 = A.L[1] >> 1, ncol = n0 * n1;
  const int L2 = A.L[2], nstep = L2 >> 1;                       // steps per column; its ring entries are the planes -1 .. L2
  const int L3 = A.L[3], HP = L3 >> 1, NSEG = (HP + 15) >> 4;    // sites / pairs of a row; segments per row (1 when !SEGM)
  const int RB = L3 * 32, HB = L3 * 16, PB = L3 * 8;             // bytes of a row of the pair tensor / its hi block / a parity block
  const int seg_lo = SEGM ? A.seg_lo : 0, seg_n = SEGM ? (A.seg_n > 0 ? A.seg_n : NSEG) : 1;      // the segments this launch covers
  const int total = int(A.nitems);                                // B * ncol * seg_n columns (< 2^31: checked by the launcher)
  const int xcd = blockIdx.x & 7, jm = blockIdx.x >> 3;
  auto col_id = [&](int ci) { return (xcd + 8 * ci) * 32 + jm; };
  int ncols_my = 0;
  {
    const int first = col_id(0);
    if (first < total) ncols_my = (total - first + 255) / 256;
  }
  if (ncols_my == 0) return;
  // (rows of several segments only: 48^4 fetches 13.8 instead of 19.0 GB per 50 samples and runs 1.4 % faster.  With one
  // segment per row the 32 columns of an XCD are two whole periodic lines along axis 1, which have no halo along that axis and
  // are long contiguous address ranges: 4 x 8 tiles fetched 6 % less there and ran 0.8 % slower)
  const int td0 = NSEG == 1 ? 1 : ((n0 & 3) == 0 ? 4 : ((n0 & 1) == 0 ? 2 : 1));
  const int td1 = NSEG == 1 ? 1 : ((n1 & 3) == 0 ? 4 : ((n1 & 1) == 0 ? 2 : 1));
  auto decode = [&](int ci, int &b, int &i0, int &i1, int &hs) {
    int gc = col_id(ci);
    hs = 0;
    if (SEGM) {                                // the segment is the fastest index: a row's segments run at the same time
      hs = seg_lo + gc % seg_n;
      gc /= seg_n;
    }
    b = gc / ncol;
    // columns of a sample are numbered tile by tile (td0 x td1 columns in axes 0, 1, as many -- with their segments -- as an XCD
    // marches at a time): a compact cross-section shares more halo rows in the XCD's L2 than a line of 32 does
    int c = gc - b * ncol;
    const int t1 = c % td1; c /= td1;
    const int t0 = c % td0; c /= td0;
    const int nt1 = n1 / td1;
    i0 = (c / nt1) * td0 + t0;
    i1 = (c % nt1) * td1 + t1;
  };

  // ---- weights: my slices, hi and lo (A: 0..13; B: 14..26 -- its 14th register pair is never used)
  const int sl0 = isB ? NSA : 0;
  f16x8 bh[NSA], bl[NSA];
  {
    const f16x8 *__restrict__ wsp = static_cast<const f16x8 *>(A.wfrag) + lane;
#pragma unroll
    for (int r = 0; r < NSA; ++r) {
      const int rr = sl0 + r;
      const int rc = rr < 27 ? rr : 26;
      bh[r] = wsp[(2 * rc) * 64];
      bl[r] = wsp[(2 * rc + 1) * 64];
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
  [[maybe_unused]] const float descale = kInvWScale / pow2_scale_for(A.gscale_bits);      // EPI = 3

  // ---- A-fragment addressing.  Lane (pair p, k-group g) reads tap g of its pair q = 16 hs + p: site 2q + g - 1 -- parity block
  // (g + 1) & 1, slot q + (g >> 1) of the pair tensor's row (pair_row_offset), i.e. LOCAL slot j = p + (g >> 1) of the segment's
  // image: j < 16 in the main piece, j = 16 (lane p = 15 of taps 2, 3) its 17th slot -- or, when the segment is the whole
  // periodic row, slot 0.  The row of tile t (plane 2s + t of the step) at tap j2 of combo (j0, j1) is halo row
  // (z0 + j0, z1 + j1) of ring plane t + j2.
  const int ja = p + (g >> 1), para = (g + 1) & 1;
  const unsigned rowsel = unsigned(((q >> 1) * 4 + (q & 1)) * RBL);
  const unsigned lane_a = rowsel + unsigned((SEGM && ja == 16) ? 1024 + para * 16 : para * 256 + (ja & 15) * 16);
  const unsigned lane_al = lane_a + unsigned((SEGM && ja == 16) ? 32 : 512);       // its lo half
  // HALF COLUMNS (round 3): the last segment of a 48-, 80-, ... wide row holds 8 pairs; its columns pack the two planes of a step
  // into ONE tile per cross-section position -- lanes p < 8 plane 2s, lanes p >= 8 plane 2s + 1, pair 16 hs + (p & 7) -- : half
  // the MFMAs of a wave, the B wave of a pair finishes the tile.  (Not the training form EPI = 3.)
  // They run in a launch of their own (template parameter HK; the launcher sends the full segments through the plain kernel
  // first): with both forms in one kernel the register allocation went past 256 and spilled.
  constexpr bool HALF = SEGM && EPI != 3 && HK;
  const int psh = p >> 3, pm = p & 7;
  const int lane_am_delta = int(rowsel + unsigned(para * 256 + (pm + (g >> 1)) * 16)) - int(lane_a);      // (local slot <= 8: the main piece)
  // ---- DMA: wave w copies halo rows 2w and 2w + 1 of every plane; lane l brings slot (16 hs + (l & 15)) mod HP of block l >> 4
  // ([hi | lo][even | odd]) of the row; lanes 0..3 of a second piece bring the 17th slot of the four blocks
  const int hz0 = wave >> 1, hz1a = 2 * (wave & 1);
  unsigned lane_d = unsigned(lane * 16), lane_d2 = 0;          // per-lane source offsets inside a row (per column when SEGM)
  const unsigned char *__restrict__ inb = static_cast<const unsigned char *>(A.in);
  const int64_t sampleB = A.V * 32;           // bytes of one sample's pair tensor

  // Issue cursor over the ring entries of my columns: column ici, plane ipl in -1 .. L2, next slot = head & 7.  Per COLUMN the
  // global address of the wave's two rows at plane 0, per piece one shift-add.
  int ici = 0, ipl = -1, head = 0;
  const unsigned char *rowp[2] = {nullptr, nullptr};
  auto open_issue_column = [&]() {
    int ib, ii0, ii1, hs;
    decode(ici, ib, ii0, ii1, hs);
    int x0 = 2 * ii0 + hz0 - 1;
    x0 = x0 < 0 ? x0 + A.L[0] : (x0 >= A.L[0] ? x0 - A.L[0] : x0);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      int x1 = 2 * ii1 + hz1a + k - 1;
      x1 = x1 < 0 ? x1 + A.L[1] : (x1 >= A.L[1] ? x1 - A.L[1] : x1);
      rowp[k] = inb + int64_t(ib) * sampleB + int64_t((x0 * A.L[1] + x1) * L2) * RB;
    }
    if (SEGM) {
      const int blk = lane >> 4;               // piece 1: block (hl = blk >> 1, parity = blk & 1), local slot lane & 15
      int gs = 16 * hs + (lane & 15);
      gs = gs >= HP ? gs - HP : gs;
      lane_d = unsigned((blk >> 1) * HB + (blk & 1) * PB + gs * 16);
      int g17 = 16 * hs + 16;                  // piece 2 (lanes 0..3): block = lane, the 17th slot
      g17 = g17 >= HP ? g17 - HP : g17;
      lane_d2 = unsigned(((lane >> 1) & 1) * HB + (lane & 1) * PB + g17 * 16);
    }
  };
  open_issue_column();
  const unsigned lds_rows = lds0 + unsigned((hz0 * 4 + hz1a) * RBL);
  // One piece = one halo row image (1 KiB, + the 64-byte piece of 17th slots when SEGM).  A wave brings rows (hz0, hz1a) and
  // (hz0, hz1a + 1) of every plane: pieces k = 0, 1 of the entry under the cursor; the cursor moves on after the second.
  constexpr int PPP = SEGM ? 2 : 1;           // DMA instructions per piece
  auto issue_piece = [&](int k) {
    if (ici >= ncols_my) return false;
    const int x2 = ipl < 0 ? L2 - 1 : (ipl >= L2 ? 0 : ipl);
    const unsigned char *src = rowp[k] + unsigned(x2) * unsigned(RB);
    const unsigned dst = lds_rows + unsigned((head & (NSLOT - 1)) * PLANE + k * RBL);
    dma_row(src, lane_d, dst);
    if (SEGM) {
      if (lane < 4) dma_row(src, lane_d2, dst + 1024u);
    }
    if (k == 1) {
      ++head;
      if (++ipl > L2) {
        ipl = -1;
        if (++ici < ncols_my) open_issue_column();
      }
    }
    return true;
  };
  auto issue_entry = [&]() { return issue_piece(0) && issue_piece(1); };

  // ---- prologue: fill the ring (8 entries, or all there are)
#pragma unroll 1
  for (int e = 0; e < NSLOT; ++e) (void)issue_entry();
  wait_vm<0>();
  lds_barrier();

  // ---- the step machine
  const int te = isB ? 1 : 0;                 // the tile whose epilogue is mine (plane 2s + te of a step)
  unsigned char *__restrict__ outb = static_cast<unsigned char *>(A.out);
  unsigned char *ocol = nullptr;              // the current column's output row of this wave at plane te
  unsigned lane_o = 0;                        // the epilogue's store: lane (p, g) holds channels 4 (g & 1) .. + 3 of site 2q + (g >> 1)
  bool lane_ok = true;                        // my pair exists (a partial last segment has 8 of 16)
  int64_t fcol = 0;                           // EPI > 0: pair index of this lane's pair in the column's row at plane te
  int col_slot = 0;                           // EPI > 0: slot of this (column, wave) among the sample's log-det partials
  auto open_column = [&](int ci) {
    int b, i0, i1, hs;
    decode(ci, b, i0, i1, hs);
    const int qq = 16 * hs + p;
    lane_ok = qq < HP;
    if (EPI == 3) {
      // fp32 channel planes (B, 8, V): element of channel 0 at this lane's site, row (x0, x1, plane te)
      const int x0 = 2 * i0 + (q >> 1), x1 = 2 * i1 + (q & 1);
      fcol = int64_t(b) * 8 * A.V + ((int64_t(x0) * A.L[1] + x1) * L2 + te) * int64_t(L3) + (lane_ok ? 2 * qq + (g >> 1) : 0);
    } else if (EPI > 0) {
      const int x0 = 2 * i0 + (q >> 1), x1 = 2 * i1 + (q & 1);
      // the active site of a pair of row (x0, x1, x2): site parity (A.parity + x0 + x1 + x2) & 1; planes 2s + te: x2 parity = te
      const int a = (A.parity + x0 + x1 + te) & 1;
      lane_ok = lane_ok && g == 2 * a;        // the lanes that hold channels 0..3 (t, s, -, -) of the ACTIVE site
      fcol = ((int64_t(b) * A.L[0] + x0) * A.L[1] + x1) * int64_t(L2) * HP + int64_t(te) * HP + (qq < HP ? qq : 0);
      const int per = ncol * NSEG;            // columns per sample (all segments)
      const int gcl = col_id(ci);             // the launch's own numbering -> the global one [sample][cross-section][segment]
      const int gc = (gcl / seg_n) * NSEG + hs;
      col_slot = (gc - (gc / per) * per) * 8 + wave + (gc / per) * per * 8;      // sample-major: [sample][column][wave]
    }
    lane_o = unsigned(pair_row_offset(lane_ok ? 2 * qq + (g >> 1) : 0, L3) + (g & 1) * 8);
    ocol = outb + int64_t(b) * sampleB + int64_t(((2 * i0 + (q >> 1)) * A.L[1] + 2 * i1 + (q & 1)) * L2 + te) * RB;
    if constexpr (HALF) {
      // lane (p, g) of the merged tile: plane 2s + psh, pair 16 hs + pm; the B wave stores, the A wave's lanes are off
      const int qh = 16 * hs + pm;
      lane_ok = isB;
      if (EPI > 0) {
        const int x0 = 2 * i0 + (q >> 1), x1 = 2 * i1 + (q & 1);
        const int a = (A.parity + x0 + x1 + psh) & 1;
        lane_ok = isB && g == 2 * a;
        fcol = ((int64_t(b) * A.L[0] + x0) * A.L[1] + x1) * int64_t(L2) * HP + int64_t(psh) * HP + qh;
      }
      lane_o = unsigned(pair_row_offset(2 * qh + (g >> 1), L3) + (g & 1) * 8) + unsigned(psh) * unsigned(RB);
      ocol = outb + int64_t(b) * sampleB + int64_t(((2 * i0 + (q >> 1)) * A.L[1] + 2 * i1 + (q & 1)) * L2) * RB;
    }
  };
  open_column(0);
  int cci = 0, s = 0;                         // the column's index in my list, the step inside it
  int rbase = 0;                              // ring entry of plane 2s - 1 of the current column (mod 8 gives the slot)
  f32x4 prev = f32x4{0.f, 0.f, 0.f, 0.f};     // my partial sums of MY tile from the previous step (its epilogue is pending)
  unsigned char *pout = nullptr;              // ... and the row it goes to (with the lane's offset and validity in THAT column)
  unsigned plane_o = 0;
  bool plane_ok = true;
  int64_t pfield = 0;                         // EPI > 0: pair index of the step whose epilogue is pending
  int pslot = -1, aslot = -1;                 // ... its partial slot; the slot the running log-det sum belongs to
  double lacc = 0.0;
  bool have_prev = false;
  int free_next = 0;                          // ring entries freed by the previous step (2, or 4 at a column end)
  const int nsteps_total = ncols_my * nstep;
  // exchange slots of this pair in a buffer: [0] A -> B (tile 1's partial sums of A), [1] B -> A (tile 0's of B)
  const unsigned xsend = unsigned(RING + (q * 2 + (isB ? 1 : 0)) * 1024 + lane * 16);
  const unsigned xrecv = unsigned(RING + (q * 2 + (isB ? 0 : 1)) * 1024 + lane * 16);

#if defined(NF_DIAG) && defined(NF_G2_TIMING)      // diagnostic build: cycle counters around the phases of a step
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define NF_G2TICK(i) { const unsigned long long tn = __builtin_readcyclecounter(); tacc[i] += tn - tprev; tprev = tn; }
#else
#define NF_G2TICK(i)
#endif
  f32x4 nold = f32x4{0.f, 0.f, 0.f, 0.f}, pold = nold;      // EPI = 3, accumulate: what the planes hold at this step's / the pending tile's sites
  auto flush = [&]() {                        // EPI > 0: the finished column's log-det partial of this wave
    const double tot = wave_sum(lacc);
    if (lane == 0 && aslot >= 0) A.partial[aslot] = tot;
    lacc = 0.0;
  };
  auto epilogue = [&](int xbuf) {
    // partner's partial sums of my tile + mine
    const f32x4 pa = *reinterpret_cast<const f32x4 *>(smem_g2 + xbuf * XBUF + xrecv);
    if constexpr (EPI == 0) {
      // -> bias, activation -> (hi, lo) halves of this lane's 4 channels of its site
      f16x4 hi, lo;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = prev[r] + pa[r];
        const float v = act_affine(a, kc1, kc0[r], kal, kga);
        const _Float16 h0 = static_cast<_Float16>(v);
        hi[r] = h0;
        lo[r] = static_cast<_Float16>(v - static_cast<float>(h0));
      }
      if (plane_ok) {
        unsigned char *d = pout + plane_o;
        *reinterpret_cast<f16x4 *>(d) = hi;
        *reinterpret_cast<f16x4 *>(d + HB) = lo;
      }
    } else if constexpr (EPI == 3) {
      // -> descaled, as they are (an input gradient has no activation), to fp32 channel planes; optionally added to them
      if (plane_ok) {
        float *d = static_cast<float *>(A.out) + pfield + int64_t(4 * (g & 1)) * A.V;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = (prev[r] + pa[r]) * descale + bv4[r];
          if (A.accumulate) v += pold[r];                            // (the old values were requested a step ago)
          if (A.act == kActTanh) v = tanh_affine(v, 2.885390081777927f, 0.f);      // of the SUM: the last group pass of a layer with more than 8 input channels
          d[int64_t(r) * A.V] = v;
        }
      }
    } else {
      // -> (t, s) of the active site -> the affine map on the field, log-det -= / += |s|
      if (pslot != aslot) {                   // (wave-uniform) a new column: hand the finished one's sum over
        flush();
        aslot = pslot;
      }
      if (plane_ok) {
        const float t = (prev[0] + pa[0]) * kInvWScale + bv4[0];
        const float sa = __builtin_fabsf((prev[1] + pa[1]) * kInvWScale + bv4[1]);
        const float2 xv = load_field_pair(A, pfield);
        const int a = g >> 1;                 // which site of the pair is the active one (this lane holds it)
        const float v = a ? xv.y : xv.x;
        const float val = EPI == 2 ? (v - t) * nf_exp(sa) : t + v * nf_exp(-sa);
        float2 ov;
        ov.x = a ? 0.f : val;
        ov.y = a ? val : 0.f;
        store_field_pair(A, pfield, ov);
        lacc += double(EPI == 2 ? sa : -sa);
      }
    }
  };

  for (int k = 0; k < nsteps_total; ++k) {
    // (1) The ring slots the previous step released (every wave has passed the barrier: nobody reads them any more) are
    // refilled ONE PIECE AT A TIME between the MFMA blocks below: a CU takes in 10-30 bytes per clock (global loads and
    // LDS-DMA alike, MI355X_MICROARCH.md), so a burst of 32 pieces fills the address FIFO and the waves behind it stall at
    // issue.  Two entries = four pieces per wave; a column's first step has four entries: the extra two go out right here.
    int ndma = 0;
    if (EPI == 3 && A.accumulate) {
      // the values this step's tile will be added to: requested now, used by its epilogue a step from now (read at the
      // epilogue they put a memory round trip on the critical path of every step: 1.0 against 0.6 ms per pass)
      if (lane_ok) {
        const float *d = static_cast<const float *>(A.out) + fcol + int64_t(2 * s) * L3 + int64_t(4 * (g & 1)) * A.V;
#pragma unroll
        for (int r = 0; r < 4; ++r) nold[r] = d[int64_t(r) * A.V];
      }
    }
    if (!(NF_G2_ABL & 1) && free_next == 4) {
      ndma += issue_entry() ? 2 * PPP : 0;
      ndma += issue_entry() ? 2 * PPP : 0;
    }
    const bool refill = !(NF_G2_ABL & 1) && free_next >= 2;
    auto dma_slot = [&](int i) {               // slot i of 4: piece i & 1 of the entry under the cursor
      if (refill && issue_piece(i & 1)) ndma += PPP;
    };
    NF_G2TICK(0)      // step head
    // (2) the epilogue of MY tile of the previous step: B waves run it now, A waves after their MFMAs -- so that right behind
    // the barrier one wave of every SIMD multiplies while the other does vector work, and the other way round at the end
    // of the step (with both epilogues up front the matrix pipe idled for ~1000 cycles of a ~4500-cycle step)
    if (isB && have_prev && !(NF_G2_ABL & 2)) epilogue((k - 1) & 1);
    NF_G2TICK(1)      // epilogue (B)
    // (3) my share of the 27 slices, for both tiles
#if NF_G2_PRIO
    __builtin_amdgcn_s_setprio(2);             // the wave that multiplies goes first at the SIMD's issue port
#endif
    f32x4 am[2], ac[2];                        // per tile: hi*hi sums, and the two correction products
    am[0] = am[1] = ac[0] = ac[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned rowa[4], rowl[4];                 // LDS offset of this lane's fragment (hi, lo) in ring plane (rbase + i), combo (0, 0)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned po = unsigned(((rbase + i) & (NSLOT - 1)) * PLANE);
      rowa[i] = po + lane_a;
      rowl[i] = po + lane_al;
    }
    // rows I0 .. I1 of combo offset coff (tap j2 of tile t reads row t + j2); half column: taps H0 .. H1 of the merged tile
    auto fetch = [&](f16x8 (&fh)[4], f16x8 (&fl)[4], int coff, int I0, int I1, int H0, int H1) {
      if constexpr (HALF) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          if (i < H0 || i > H1) continue;
          // this lane's fragment of tap i: plane i (+ 1 for the upper half of the lanes) of the ring window, its own slot
          // (arithmetic on the lane's plane index, not a select between rowa[i] and rowa[i + 1]: the compiler turned that
          //  select into a scratch array indexed per lane)
          const unsigned rm = unsigned(((rbase + i + psh) & (NSLOT - 1)) * PLANE) + unsigned(int(lane_a) + lane_am_delta);
          fh[i] = *reinterpret_cast<const f16x8 *>(smem_g2 + rm + coff);
          fl[i] = *reinterpret_cast<const f16x8 *>(smem_g2 + rm + 512 + coff);
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < I0 || i > I1) continue;
        if (NF_G2_ABL & 8) {
          fh[i] = bh[i]; fl[i] = bl[i];
          asm volatile("" : "+v"(fh[i]), "+v"(fl[i]));
        } else {
          fh[i] = *reinterpret_cast<const f16x8 *>(smem_g2 + rowa[i] + coff);
          fl[i] = *reinterpret_cast<const f16x8 *>(smem_g2 + rowl[i] + coff);
        }
      }
    };
    // taps J0 .. J1 of a combo whose tap 0 is my local slice `base` (compile-time at every call site).  The WEIGHTS are the
    // MFMA's A operand (rows m = column (shift, co) of the layer), the site pairs its B operand: D[m][pair] leaves the four
    // channels 4 (g & 1) .. + 3 of ONE site (2p + (g >> 1)) in each lane -- what a store needs, no transpose
    auto mult = [&](const f16x8 (&fh)[4], const f16x8 (&fl)[4], int base, int J0, int J1) {
      if (NF_G2_ABL & 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fh[i]), "v"(fl[i]));
        return;
      }
      if constexpr (HALF) {                    // one merged tile: tap j2 multiplies fragment j2
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2)
          if (j2 >= J0 && j2 <= J1) am[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[base + j2], fh[j2], am[0], 0, 0, 0);
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2)
          if (j2 >= J0 && j2 <= J1) ac[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[base + j2], fh[j2], ac[0], 0, 0, 0);
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2)
          if (j2 >= J0 && j2 <= J1) ac[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[base + j2], fl[j2], ac[0], 0, 0, 0);
        return;
      }
#pragma unroll
      for (int j2 = 0; j2 < 3; ++j2)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (j2 >= J0 && j2 <= J1) am[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[base + j2], fh[t + j2], am[t], 0, 0, 0);
#pragma unroll
      for (int j2 = 0; j2 < 3; ++j2)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (j2 >= J0 && j2 <= J1) ac[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[base + j2], fh[t + j2], ac[t], 0, 0, 0);
#pragma unroll
      for (int j2 = 0; j2 < 3; ++j2)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (j2 >= J0 && j2 <= J1) ac[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[base + j2], fl[t + j2], ac[t], 0, 0, 0);
    };
    auto combo_off = [](int jj) { return ((jj / 3) * 4 + jj % 3) * RBL; };      // halo row (j0, j1) relative to the pair's own
    f16x8 fh0[4], fl0[4], fh1[4], fl1[4];
    // (sched_barrier: left alone the compiler sinks every fragment read next to its first use and exposes the LDS latency
    //  at each MFMA group; the reads of the next combo must issue BEFORE the current combo's MFMAs)
#define NF_SB __builtin_amdgcn_sched_barrier(0)
    if (!isB) {                                // slices 0..13: combos 0..3 and taps 0, 1 of combo 4
      fetch(fh0, fl0, combo_off(0), 0, 3, 0, 2);
      fetch(fh1, fl1, combo_off(1), 0, 3, 0, 2);
      NF_SB; mult(fh0, fl0, 0, 0, 2); NF_SB;
      dma_slot(0);
      fetch(fh0, fl0, combo_off(2), 0, 3, 0, 2);
      NF_SB; mult(fh1, fl1, 3, 0, 2); NF_SB;
      dma_slot(1);
      fetch(fh1, fl1, combo_off(3), 0, 3, 0, 2);
      NF_SB; mult(fh0, fl0, 6, 0, 2); NF_SB;
      dma_slot(2);
      fetch(fh0, fl0, combo_off(4), 0, 2, 0, 1);
      NF_SB; mult(fh1, fl1, 9, 0, 2); NF_SB;
      dma_slot(3);
      mult(fh0, fl0, 12, 0, 1);
      NF_SB;
    } else {                                   // slices 14..26: tap 2 of combo 4 and combos 5..8 (local index = global - 14)
      fetch(fh0, fl0, combo_off(4), 2, 3, 2, 2);
      fetch(fh1, fl1, combo_off(5), 0, 3, 0, 2);
      NF_SB; mult(fh0, fl0, 12 - NSA, 2, 2); NF_SB;
      fetch(fh0, fl0, combo_off(6), 0, 3, 0, 2);
      NF_SB; mult(fh1, fl1, 15 - NSA, 0, 2); NF_SB;
      dma_slot(0);
      fetch(fh1, fl1, combo_off(7), 0, 3, 0, 2);
      NF_SB; mult(fh0, fl0, 18 - NSA, 0, 2); NF_SB;
      dma_slot(1);
      fetch(fh0, fl0, combo_off(8), 0, 3, 0, 2);
      NF_SB; mult(fh1, fl1, 21 - NSA, 0, 2); NF_SB;
      dma_slot(2);
      mult(fh0, fl0, 24 - NSA, 0, 2);
      NF_SB;
      dma_slot(3);
    }
#undef NF_SB
#if NF_G2_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    if (!isB && have_prev && !(NF_G2_ABL & 2)) epilogue((k - 1) & 1);
    NF_G2TICK(2)      // fragment reads + MFMAs + DMA pieces (+ the A waves' epilogue)
    // (4) the partial sums of the partner's tile cross over; mine stay for the next step's epilogue
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) am[t][r] += ac[t][r];
    constexpr bool mh = HALF;                  // (half column: the A wave hands its half of the K sum of the merged tile to B)
    *reinterpret_cast<f32x4 *>(smem_g2 + (k & 1) * XBUF + xsend) = (isB || mh) ? am[0] : am[1];
    prev = (isB && !mh) ? am[1] : am[0];
    pout = ocol + unsigned(2 * s) * unsigned(RB);
    plane_o = lane_o;
    plane_ok = lane_ok;
    if (EPI == 3) {
      pfield = fcol + int64_t(2 * s) * L3;
      pold = nold;
    } else if (EPI > 0) {
      pfield = fcol + int64_t(2 * s) * HP;
      pslot = col_slot;
    }
    have_prev = true;
    // (5) the planes of the NEXT step were issued a full step ago or earlier: everything but this step's own DMAs (the
    // youngest operations: the epilogue's stores precede them) must have landed before the barrier.  A column's first step
    // reads four planes, the last two of which were issued during THIS step when this is a column's last step: then they
    // must land as well.
    const int allow = s + 1 == nstep ? 0 : ndma;
    if (allow >= 16) wait_vm<16>();
    else if (allow >= 12) wait_vm<12>();
    else if (allow >= 8) wait_vm<8>();
    else if (allow >= 6) wait_vm<6>();
    else if (allow >= 4) wait_vm<4>();
    else if (allow >= 2) wait_vm<2>();
    else wait_vm<0>();
    NF_G2TICK(3)      // hand-over + the wait for the next step's planes
    lds_barrier();
    NF_G2TICK(4)      // barrier
    // advance the compute cursor
    if (++s == nstep) {
      s = 0;
      free_next = 4;
      rbase += 4;                              // the next column's plane -1 follows this column's plane L2 in the ring
      if (++cci < ncols_my) open_column(cci);
    } else {
      free_next = 2;
      rbase += 2;
    }
  }
#if defined(NF_DIAG) && defined(NF_G2_TIMING)
  if (blockIdx.x == 9 && lane == 0 && (wave == 0 || wave == 4) && nsteps_total > 100)
    printf("[g2 timing] wave %d steps %d | cycles per step: head %.0f  epilogue %.0f  mfma %.0f  wait %.0f  barrier %.0f\n", wave, nsteps_total,
           double(tacc[0]) / nsteps_total, double(tacc[1]) / nsteps_total, double(tacc[2]) / nsteps_total, double(tacc[3]) / nsteps_total, double(tacc[4]) / nsteps_total);
#endif
  // ---- the last step's epilogue
  if (have_prev) epilogue((nsteps_total - 1) & 1);
  if (EPI == 1 || EPI == 2) flush();
  wait_vm<0>();                               // no DMA may outlive the workgroup's LDS allocation
}

}  // namespace nf

using namespace nf;

extern "C" int nf_conv_split16_supported(const int32_t *lattice, const int32_t *ksize, int cin, int cout, int act) {
  if (!nf::option(NF_OPT_SPLIT16) || !lattice || !ksize || cin != 8 || cout != 8) return 0;
  if (act != kActTanh && act != kActSigmoid) return 0;                    // the OUTPUT must be fp16-safe as well
  for (int mu = 0; mu < 4; ++mu)
    if (ksize[mu] != 3) return 0;
  if (lattice[3] < 32 || (lattice[3] & 15)) return 0;                     // whole or half segments of 32 sites
  for (int mu = 0; mu < 3; ++mu)
    if (lattice[mu] < 2 || (lattice[mu] & 1)) return 0;
  return 1;
}

static int launch_g2(ConvArgs &A, const int32_t *lattice, int64_t B, int epi, hipStream_t stream, const char *what) {
  A.V = 1;
  for (int mu = 0; mu < 4; ++mu) {
    A.L[mu] = lattice[mu]; A.k[mu] = 3;
    A.V *= lattice[mu];
  }
  A.cin = 8; A.cout = 8;
  const bool segm = lattice[3] != 32;
  A.nitems = B * int64_t(lattice[0] / 2) * int64_t(lattice[1] / 2) * int64_t(segm ? (lattice[3] / 2 + 15) / 16 : 1);      // columns (x segments)
  NF_REQUIRE(A.V * 32 < (int64_t(1) << 32), "%s: a sample's pair tensor must stay below 4 GiB", what);
  NF_REQUIRE(A.nitems < (int64_t(1) << 28) - 4096, "%s: batch x columns >= 2^28, split the batch", what);
  // one persistent workgroup per CU of an MI355X; workgroup (xcd = id & 7, j = id >> 3) takes member j of every 8th group of
  // 32 columns (on a part with fewer CUs the surplus workgroups simply queue: there is no inter-workgroup dependency)
  const int64_t grid = 256;
  auto go = [&](auto kern, int lds) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      set_error("%s: cannot reserve %d B of LDS", what, lds);
      return int(NF_ELAUNCH);
    }
    hipLaunchKernelGGL(kern, dim3(unsigned(grid)), dim3(512), lds, stream, A);
    return int(NF_OK);
  };
  int rc;
  if (segm) {
    constexpr int lds = g2::Geo<true>::LDS_BYTES;
    const int nseg = (lattice[3] / 2 + 15) / 16;
    const int64_t per_seg = A.nitems / nseg;
    if (epi != 3 && ((lattice[3] / 2) & 15) == 8) {
      // a last segment of 8 pairs: the full segments through the plain kernel, then the half columns in their packed form
      rc = NF_OK;
      if (nseg > 1) {
        A.seg_lo = 0; A.seg_n = nseg - 1; A.nitems = per_seg * (nseg - 1);
        rc = epi == 0 ? go(&conv_g2_kernel<true, 0>, lds) : (epi == 1 ? go(&conv_g2_kernel<true, 1>, lds) : go(&conv_g2_kernel<true, 2>, lds));
      }
      if (rc == NF_OK) {
        A.seg_lo = nseg - 1; A.seg_n = 1; A.nitems = per_seg;
        rc = epi == 0 ? go(&conv_g2_kernel<true, 0, true>, lds) : (epi == 1 ? go(&conv_g2_kernel<true, 1, true>, lds) : go(&conv_g2_kernel<true, 2, true>, lds));
      }
    } else {
      A.seg_lo = 0; A.seg_n = nseg;
      rc = epi == 0 ? go(&conv_g2_kernel<true, 0>, lds) : (epi == 1 ? go(&conv_g2_kernel<true, 1>, lds) : (epi == 2 ? go(&conv_g2_kernel<true, 2>, lds) : go(&conv_g2_kernel<true, 3>, lds)));
    }
  } else {
    constexpr int lds = g2::Geo<false>::LDS_BYTES;
    rc = epi == 0 ? go(&conv_g2_kernel<false, 0>, lds) : (epi == 1 ? go(&conv_g2_kernel<false, 1>, lds) : (epi == 2 ? go(&conv_g2_kernel<false, 2>, lds) : go(&conv_g2_kernel<false, 3>, lds)));
  }
  if (rc) return rc;
  return check_launch(what);
}

extern "C" int nf_conv_fwd_split16(const void *in16, const void *wsplit, const void *bias, void *out16, int64_t B,
                                   const int32_t *lattice, int act, void *stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  NF_REQUIRE(in16 && wsplit && out16 && lattice, "nf_conv_fwd_split16: NULL pointer");
  const int32_t k3[4] = {3, 3, 3, 3};
  NF_REQUIRE(nf_conv_split16_supported(lattice, k3, 8, 8, act), "nf_conv_fwd_split16: layer not supported (needs a fastest axis of 32 + 16 n sites, even other extents, tanh / sigmoid)");
  NF_REQUIRE(B >= 0 && B <= 65535, "nf_conv_fwd_split16: batch outside [0, 65535]");
  NF_REQUIRE(in16 != out16, "nf_conv_fwd_split16: in place is not possible (a layer reads its neighbours' inputs)");
  if (B == 0) return NF_OK;
  ConvArgs A{};
  A.in = in16; A.wfrag = wsplit; A.bias = bias; A.out = out16;
  A.act = act;
  return launch_g2(A, lattice, B, 0, stream, "nf_conv_fwd_split16");
}

// Partials per sample of nf_conv_affine_split16: one per (column, segment, wave).
static int64_t affine_split16_blocks(const int32_t *lattice) {
  return int64_t(lattice[0] / 2) * (lattice[1] / 2) * (lattice[3] != 32 ? (lattice[3] / 2 + 15) / 16 : 1) * 8;
}

extern "C" int nf_conv_affine_split16(const void *in16, const void *wsplit, const void *bias, const void *x_active,
                                      const void *log0, void *y, void *logj, int64_t B, const int32_t *lattice,
                                      int active_parity, int inverse, int flags, void *workspace, size_t workspace_bytes,
                                      void *stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  NF_REQUIRE(in16 && wsplit && x_active && y && logj && lattice, "nf_conv_affine_split16: NULL pointer");
  const int32_t k3[4] = {3, 3, 3, 3};
  NF_REQUIRE(nf_conv_split16_supported(lattice, k3, 8, 8, kActTanh), "nf_conv_affine_split16: lattice not supported (needs a fastest axis of 32 + 16 n sites, even other extents)");
  NF_REQUIRE(B >= 0 && B <= 65535, "nf_conv_affine_split16: batch outside [0, 65535]");
  if (B == 0) return NF_OK;
  const int64_t blocks = affine_split16_blocks(lattice);
  const size_t need = size_t(B) * size_t(blocks) * sizeof(double);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("nf_conv_affine_split16: workspace %zu B < %zu B needed", workspace_bytes, need);
    return NF_EWORKSPACE;
  }
  ConvArgs A{};
  A.in = in16; A.wfrag = wsplit; A.bias = bias;
  A.xact = static_cast<const float *>(x_active);
  A.yout = static_cast<float *>(y);
  A.partial = static_cast<double *>(workspace);
  A.parity = active_parity & 1;
  A.field16 = (flags & NF_CONV_FIELD_F16) ? 1 : 0;
  const int rc = launch_g2(A, lattice, B, inverse ? 2 : 1, stream, "nf_conv_affine_split16");
  if (rc) return rc;
  return launch_finalize<float>(A.partial, blocks, log0, logj, B, stream);
}

// The hidden-layer kernel with fp32 channel planes (B, 8, V) out -- for training, whose other kernels read planes:
//  * the input gradient of an 8 -> 8 layer, or one 8-channel group of the input gradient of a wider layer: the layer's weights
//    flipped and transposed (packed like a forward layer's), bias NULL, act 0; in16 is the cotangent's pair tensor, scaled by
//    the power of two that belongs to *absmax_bits (nf_planes_to_split16); the planes are written, or added to when
//    `accumulate` (the groups of a 46 -> 8 gradient take turns);
//  * a forward 8 -> 8 layer whose output autograd keeps: bias, act = tanh (or 0), absmax_bits NULL.
extern "C" int nf_conv_dgrad_split16(const void *in16, const void *wsplit, const void *bias, void *gx, int64_t B,
                                     const int32_t *lattice, const void *absmax_bits, int accumulate, int act,
                                     void *stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  NF_REQUIRE(in16 && wsplit && gx && lattice, "nf_conv_dgrad_split16: NULL pointer");
  const int32_t k3[4] = {3, 3, 3, 3};
  NF_REQUIRE(nf_conv_split16_supported(lattice, k3, 8, 8, kActTanh), "nf_conv_dgrad_split16: lattice not supported (needs a fastest axis of 32 + 16 n sites, even other extents)");
  NF_REQUIRE(B >= 0 && B <= 65535, "nf_conv_dgrad_split16: batch outside [0, 65535]");
  if (B == 0) return NF_OK;
  ConvArgs A{};
  NF_REQUIRE(act == kActNone || act == kActTanh, "nf_conv_dgrad_split16: activation %d (none or tanh)", act);
  A.in = in16; A.wfrag = wsplit; A.bias = bias; A.out = gx;
  A.act = act;
  A.gscale_bits = static_cast<const unsigned *>(absmax_bits);
  A.accumulate = accumulate ? 1 : 0;
  return launch_g2(A, lattice, B, 3, stream, "nf_conv_dgrad_split16");
}
