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
constexpr int L3 = 32;                        // sites of the fastest axis (a lattice row = one MFMA tile of 16 pairs)
constexpr int RB = L3 * 32, HB = L3 * 16, PB = L3 * 8;      // bytes of a row / of its hi block / of a parity block
constexpr int PLROWS = 16;                    // halo rows of a plane: (z0, z1) in {-1..2}^2, row index 4*hz0 + hz1
constexpr int PLANE = PLROWS * RB;            // 16 KiB
constexpr int NSLOT = 8;                      // ring: 4 planes being read + 4 ahead
constexpr int RING = NSLOT * PLANE;           // 128 KiB
constexpr int XBUF = 4 * 2 * 64 * 16;         // partial-sum exchange: [wave pair][direction][lane] x 16 B, one of two buffers
constexpr int LDS_BYTES = RING + 2 * XBUF;                  // 131072 + 16384 = 147456 <= 163840
constexpr int NSA = 14;                       // slices of the A waves (0..13); B waves: 14..26
constexpr float kInvWScale = 1.0f / 1024.0f;  // the weights are packed scaled by 2^10 (normflow__amd/_hip.py: SPLIT16_WEIGHT_SCALE)
static_assert(LDS_BYTES <= 160 * 1024, "ring + exchange must fit the CU's LDS");
}  // namespace g2

// tanh(v) for v = a * scale + bias given as (a, c1 = 2 log2(e) scale, c0 = 2 log2(e) bias): 1 - 2 / (1 + 2^(c1 a + c0)); five
// instructions, two of them transcendental; absolute error ~1e-7 (the rounding of a number near 1), exact limits at +-inf.
__device__ __forceinline__ float tanh_affine(float a, float c1, float c0) {
  const float t = __builtin_amdgcn_exp2f(__builtin_fmaf(a, c1, c0));
  return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + t), 1.0f);
}

__global__ __launch_bounds__(512, 2) void conv_g2_kernel(ConvArgs A) {
  using namespace g2;
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
  const int total = int(A.nitems);                                // B * ncol columns (< 2^31: checked by the launcher)
  const int xcd = blockIdx.x & 7, jm = blockIdx.x >> 3;
  auto col_id = [&](int ci) { return (xcd + 8 * ci) * 32 + jm; };
  int ncols_my = 0;
  {
    const int first = col_id(0);
    if (first < total) ncols_my = (total - first + 255) / 256;
  }
  if (ncols_my == 0) return;
  auto decode = [&](int ci, int &b, int &i0, int &i1) {
    const int gc = col_id(ci);
    b = gc / ncol;
    const int c = gc - b * ncol;
    i0 = c / n1;
    i1 = c - i0 * n1;
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
    kc0[r] = 2.885390081777927f * bv4[r];      // 2 log2(e) x bias
  }
  const float kc1 = 2.885390081777927f * kInvWScale;

  // ---- A-fragment addressing.  Lane (pair p, k-group g) reads tap g of its pair: site 2p + g - 1 (mod L3) -- parity block
  // (g + 1) & 1, slot p + (g >> 1) (mod 16) of the row image (pair_row_offset).  The row of tile t (plane 2s + t of the step) at
  // tap j2 of combo (j0, j1) is halo row (z0 + j0, z1 + j1) of ring plane t + j2.
  const unsigned lane_a = unsigned((((g + 1) & 1) * PB) + (((p + (g >> 1)) & 15) * 16) + (((q >> 1) * 4 + (q & 1)) * RB));
  // ---- DMA: wave w copies halo rows 2w and 2w + 1 of every plane; lane l brings bytes [16 l, 16 l + 16) of the row
  const int hz0 = wave >> 1, hz1a = 2 * (wave & 1);
  const unsigned lane_d = unsigned(lane * 16);
  const unsigned char *__restrict__ inb = static_cast<const unsigned char *>(A.in);
  const int64_t sampleB = A.V * 32;           // bytes of one sample's pair tensor

  // Issue cursor over the ring entries of my columns: column ici, plane ipl in -1 .. L2, next slot = head & 7.  Per COLUMN the
  // global address of the wave's two rows at plane 0, per piece one shift-add.
  int ici = 0, ipl = -1, head = 0;
  const unsigned char *rowp[2] = {nullptr, nullptr};
  auto open_issue_column = [&]() {
    int ib, ii0, ii1;
    decode(ici, ib, ii0, ii1);
    int x0 = 2 * ii0 + hz0 - 1;
    x0 = x0 < 0 ? x0 + A.L[0] : (x0 >= A.L[0] ? x0 - A.L[0] : x0);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      int x1 = 2 * ii1 + hz1a + k - 1;
      x1 = x1 < 0 ? x1 + A.L[1] : (x1 >= A.L[1] ? x1 - A.L[1] : x1);
      rowp[k] = inb + int64_t(ib) * sampleB + int64_t((x0 * A.L[1] + x1) * L2) * RB;
    }
  };
  open_issue_column();
  const unsigned lds_rows = lds0 + unsigned((hz0 * 4 + hz1a) * RB);
  // One piece = one halo row (1 KiB).  A wave brings rows (hz0, hz1a) and (hz0, hz1a + 1) of every plane: pieces k = 0, 1 of
  // the entry under the cursor; the cursor moves on after the second.
  auto issue_piece = [&](int k) {
    if (ici >= ncols_my) return false;
    const int x2 = ipl < 0 ? L2 - 1 : (ipl >= L2 ? 0 : ipl);
    dma_row(rowp[k] + unsigned(x2) * unsigned(RB), lane_d, lds_rows + unsigned((head & (NSLOT - 1)) * PLANE + k * RB));
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
  auto open_column = [&](int ci) {
    int b, i0, i1;
    decode(ci, b, i0, i1);
    ocol = outb + int64_t(b) * sampleB + int64_t(((2 * i0 + (q >> 1)) * A.L[1] + 2 * i1 + (q & 1)) * L2 + te) * RB;
  };
  open_column(0);
  int cci = 0, s = 0;                         // the column's index in my list, the step inside it
  int rbase = 0;                              // ring entry of plane 2s - 1 of the current column (mod 8 gives the slot)
  f32x4 prev = f32x4{0.f, 0.f, 0.f, 0.f};     // my partial sums of MY tile from the previous step (its epilogue is pending)
  unsigned char *pout = nullptr;              // ... and the row it goes to
  bool have_prev = false;
  int free_next = 0;                          // ring entries freed by the previous step (2, or 4 at a column end)
  const int nsteps_total = ncols_my * nstep;
  // exchange slots of this pair in a buffer: [0] A -> B (tile 1's partial sums of A), [1] B -> A (tile 0's of B)
  const unsigned xsend = unsigned(RING + (q * 2 + (isB ? 1 : 0)) * 1024 + lane * 16);
  const unsigned xrecv = unsigned(RING + (q * 2 + (isB ? 0 : 1)) * 1024 + lane * 16);
  // the epilogue's store: lane (p, g) holds channels 4 (g & 1) .. + 3 of site 2p + (g >> 1)
  const unsigned lane_o = unsigned(pair_row_offset(2 * p + (g >> 1), L3) + (g & 1) * 8);

#if defined(NF_DIAG) && defined(NF_G2_TIMING)      // diagnostic build: cycle counters around the phases of a step
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define NF_G2TICK(i) { const unsigned long long tn = __builtin_readcyclecounter(); tacc[i] += tn - tprev; tprev = tn; }
#else
#define NF_G2TICK(i)
#endif
  auto epilogue = [&](int xbuf) {
    // partner's partial sums of my tile + mine -> bias, activation -> (hi, lo) halves of this lane's 4 channels of its site
    const f32x4 pa = *reinterpret_cast<const f32x4 *>(smem_g2 + xbuf * XBUF + xrecv);
    f16x4 hi, lo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a = prev[r] + pa[r];
      const float v = A.act == kActTanh ? tanh_affine(a, kc1, kc0[r]) : activate(a * kInvWScale + bv4[r], kActSigmoid);
      const _Float16 h0 = static_cast<_Float16>(v);
      hi[r] = h0;
      lo[r] = static_cast<_Float16>(v - static_cast<float>(h0));
    }
    unsigned char *d = pout + lane_o;
    *reinterpret_cast<f16x4 *>(d) = hi;
    *reinterpret_cast<f16x4 *>(d + HB) = lo;
  };

  for (int k = 0; k < nsteps_total; ++k) {
    // (1) The ring slots the previous step released (every wave has passed the barrier: nobody reads them any more) are
    // refilled ONE PIECE AT A TIME between the MFMA blocks below: a CU takes in 10-30 bytes per clock (global loads and
    // LDS-DMA alike, MI355X_MICROARCH.md), so a burst of 32 pieces fills the address FIFO and the waves behind it stall at
    // issue.  Two entries = four pieces per wave; a column's first step has four entries: the extra two go out right here.
    int ndma = 0;
    if (!(NF_G2_ABL & 1) && free_next == 4) {
      ndma += issue_entry() ? 2 : 0;
      ndma += issue_entry() ? 2 : 0;
    }
    const bool refill = !(NF_G2_ABL & 1) && free_next >= 2;
    auto dma_slot = [&](int i) {               // slot i of 4: piece i & 1 of the entry under the cursor
      if (refill && issue_piece(i & 1)) ++ndma;
    };
    NF_G2TICK(0)      // step head
    // (2) the epilogue of MY tile of the previous step: B waves run it now, A waves after their MFMAs -- so that right behind
    // the barrier one wave of every SIMD multiplies while the other does vector work, and the other way round at the end
    // of the step (with both epilogues up front the matrix pipe idled for ~1000 cycles of a ~4500-cycle step)
    if (isB && have_prev && !(NF_G2_ABL & 2)) epilogue((k - 1) & 1);
    NF_G2TICK(1)      // epilogue (B)
    // (3) my share of the 27 slices, for both tiles
    f32x4 am[2], ac[2];                        // per tile: hi*hi sums, and the two correction products
    am[0] = am[1] = ac[0] = ac[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned rowa[4];                          // LDS offset of this lane's fragment in ring plane (rbase + i), combo (0, 0)
#pragma unroll
    for (int i = 0; i < 4; ++i) rowa[i] = unsigned(((rbase + i) & (NSLOT - 1)) * PLANE) + lane_a;
    // rows I0 .. I1 of combo offset coff (tap j2 of tile t reads row t + j2)
    auto fetch = [&](f16x8 (&fh)[4], f16x8 (&fl)[4], int coff, int I0, int I1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < I0 || i > I1) continue;
        if (NF_G2_ABL & 8) {
          fh[i] = bh[i]; fl[i] = bl[i];
          asm volatile("" : "+v"(fh[i]), "+v"(fl[i]));
        } else {
          fh[i] = *reinterpret_cast<const f16x8 *>(smem_g2 + rowa[i] + coff);
          fl[i] = *reinterpret_cast<const f16x8 *>(smem_g2 + rowa[i] + coff + HB);
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
    auto combo_off = [](int jj) { return ((jj / 3) * 4 + jj % 3) * RB; };      // halo row (j0, j1) relative to the pair's own
    f16x8 fh0[4], fl0[4], fh1[4], fl1[4];
    // (sched_barrier: left alone the compiler sinks every fragment read next to its first use and exposes the LDS latency
    //  at each MFMA group; the reads of the next combo must issue BEFORE the current combo's MFMAs)
#define NF_SB __builtin_amdgcn_sched_barrier(0)
    if (!isB) {                                // slices 0..13: combos 0..3 and taps 0, 1 of combo 4
      fetch(fh0, fl0, combo_off(0), 0, 3);
      fetch(fh1, fl1, combo_off(1), 0, 3);
      NF_SB; mult(fh0, fl0, 0, 0, 2); NF_SB;
      dma_slot(0);
      fetch(fh0, fl0, combo_off(2), 0, 3);
      NF_SB; mult(fh1, fl1, 3, 0, 2); NF_SB;
      dma_slot(1);
      fetch(fh1, fl1, combo_off(3), 0, 3);
      NF_SB; mult(fh0, fl0, 6, 0, 2); NF_SB;
      dma_slot(2);
      fetch(fh0, fl0, combo_off(4), 0, 2);
      NF_SB; mult(fh1, fl1, 9, 0, 2); NF_SB;
      dma_slot(3);
      mult(fh0, fl0, 12, 0, 1);
      NF_SB;
    } else {                                   // slices 14..26: tap 2 of combo 4 and combos 5..8 (local index = global - 14)
      fetch(fh0, fl0, combo_off(4), 2, 3);
      fetch(fh1, fl1, combo_off(5), 0, 3);
      NF_SB; mult(fh0, fl0, 12 - NSA, 2, 2); NF_SB;
      fetch(fh0, fl0, combo_off(6), 0, 3);
      NF_SB; mult(fh1, fl1, 15 - NSA, 0, 2); NF_SB;
      dma_slot(0);
      fetch(fh1, fl1, combo_off(7), 0, 3);
      NF_SB; mult(fh0, fl0, 18 - NSA, 0, 2); NF_SB;
      dma_slot(1);
      fetch(fh0, fl0, combo_off(8), 0, 3);
      NF_SB; mult(fh1, fl1, 21 - NSA, 0, 2); NF_SB;
      dma_slot(2);
      mult(fh0, fl0, 24 - NSA, 0, 2);
      NF_SB;
      dma_slot(3);
    }
#undef NF_SB
    if (!isB && have_prev && !(NF_G2_ABL & 2)) epilogue((k - 1) & 1);
    NF_G2TICK(2)      // fragment reads + MFMAs + DMA pieces (+ the A waves' epilogue)
    // (4) the partial sums of the partner's tile cross over; mine stay for the next step's epilogue
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) am[t][r] += ac[t][r];
    *reinterpret_cast<f32x4 *>(smem_g2 + (k & 1) * XBUF + xsend) = isB ? am[0] : am[1];
    prev = isB ? am[1] : am[0];
    pout = ocol + unsigned(2 * s) * unsigned(RB);
    have_prev = true;
    // (5) the planes of the NEXT step were issued a full step ago or earlier: everything but this step's own DMAs (the
    // youngest operations: the epilogue's stores precede them) must have landed before the barrier.  A column's first step
    // reads four planes, the last two of which were issued during THIS step when this is a column's last step: then they
    // must land as well.
    const int allow = s + 1 == nstep ? 0 : ndma;
    if (allow >= 8) wait_vm<8>();
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
  wait_vm<0>();                               // no DMA may outlive the workgroup's LDS allocation
}

}  // namespace nf

using namespace nf;

extern "C" int nf_conv_split16_supported(const int32_t *lattice, const int32_t *ksize, int cin, int cout, int act) {
  if (!nf::option(NF_OPT_SPLIT16) || !lattice || !ksize || cin != 8 || cout != 8) return 0;
  if (act != kActTanh && act != kActSigmoid) return 0;                    // the OUTPUT must be fp16-safe as well
  for (int mu = 0; mu < 4; ++mu)
    if (ksize[mu] != 3) return 0;
  if (lattice[3] != g2::L3) return 0;
  for (int mu = 0; mu < 3; ++mu)
    if (lattice[mu] < 2 || (lattice[mu] & 1)) return 0;
  return 1;
}

extern "C" int nf_conv_fwd_split16(const void *in16, const void *wsplit, const void *bias, void *out16, int64_t B,
                                   const int32_t *lattice, int act, void *stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  NF_REQUIRE(in16 && wsplit && out16 && lattice, "nf_conv_fwd_split16: NULL pointer");
  const int32_t k3[4] = {3, 3, 3, 3};
  NF_REQUIRE(nf_conv_split16_supported(lattice, k3, 8, 8, act), "nf_conv_fwd_split16: layer not supported (needs a 32-site fastest axis, even other extents, tanh / sigmoid)");
  NF_REQUIRE(B >= 0 && B <= 65535, "nf_conv_fwd_split16: batch outside [0, 65535]");
  NF_REQUIRE(in16 != out16, "nf_conv_fwd_split16: in place is not possible (a layer reads its neighbours' inputs)");
  if (B == 0) return NF_OK;
  ConvArgs A{};
  A.in = in16; A.wfrag = wsplit; A.bias = bias; A.out = out16;
  A.V = 1;
  for (int mu = 0; mu < 4; ++mu) {
    A.L[mu] = lattice[mu]; A.k[mu] = 3;
    A.V *= lattice[mu];
  }
  A.cin = 8; A.cout = 8; A.act = act;
  A.nitems = B * int64_t(lattice[0] / 2) * int64_t(lattice[1] / 2);      // columns
  NF_REQUIRE(A.V * 32 < (int64_t(1) << 32), "nf_conv_fwd_split16: a sample's pair tensor must stay below 4 GiB");
  NF_REQUIRE(A.nitems < (int64_t(1) << 31) - 4096, "nf_conv_fwd_split16: batch x columns >= 2^31, split the batch");
  // one persistent workgroup per CU of an MI355X; workgroup (xcd = id & 7, j = id >> 3) takes member j of every 8th group of
  // 32 columns (on a part with fewer CUs the surplus workgroups simply queue: there is no inter-workgroup dependency)
  const int64_t grid = 256;
  NF_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_g2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, g2::LDS_BYTES) == hipSuccess,
             "nf_conv_fwd_split16: cannot reserve %d B of LDS", g2::LDS_BYTES);
  hipLaunchKernelGGL(conv_g2_kernel, dim3(unsigned(grid)), dim3(512), g2::LDS_BYTES, stream, A);
  return check_launch("conv split-fp16 hidden-layer kernel");
}
