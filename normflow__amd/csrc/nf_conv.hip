// nf_conv.hip -- K5: circular 'same' convolution (lattice dimension 1..4) as an implicit
// GEMM on the f32 matrix cores (v_mfma_f32_16x16x4_f32: exact f32 products and
// accumulation, same numerics as an fmaf chain), with bias + activation fused and an
// optional "active sites only, pair-compact" output for the layer that feeds a coupling
// kernel (include/normflow_hip.h, NF_LAYOUT_PAIR).
//
// Restates what ConvAct computes (reference: src/nn/scalar/modules.py:120-145 -- a chain of
// torch Conv{1,2,3}d(padding='same', padding_mode='circular') / Conv4d + activation;
// Conv4d itself is src/nn/scalar/convNd.py:86-126: k0 shifted 3-d convolutions summed):
//     out[b,o,n] = bias[o] + sum_{i,j} W[o,i,j] * in[b,i,(n + j - k//2) mod L]
//
// Mapping (one workgroup = one sample x one box of output sites):
//   * the input box plus its halo, all input channels, is staged in LDS as channel planes
//     (plane stride chosen so that the 4 k-groups of an MFMA A-fragment hit disjoint banks);
//   * GEMM view: rows M = output sites (16 per MFMA tile), columns N = output channels
//     (16 per tile), reduction K = (tap, input channel), 4 per MFMA;
//   * A fragments (site, channel) are LDS reads at  lane_base + tap_offset + plane;
//     B fragments (weights) are pre-packed on the host side in fragment order and read
//     straight from global memory (a few KB per layer, L1/L2 resident, one coalesced
//     256-B row per fragment) -- no LDS spent on them;
//   * every wave owns MT site tiles x NT channel tiles of accumulators (MT*NT independent
//     chains cover the 40-cycle MFMA latency);
//   * epilogue: bias, activation, store as channel planes (B, Cout, V) -- the layout torch
//     uses, so the kernel drops in for the torch convolution -- or, in pair-compact mode,
//     only the ACTIVE site of every aligned site pair, to (B, Cout, V/2).
#include <cstdio>
#include <cstdlib>
#include "nf_conv_core.h"

namespace nf {

template <typename T, int MT, int NT, int KQ>
__device__ __forceinline__ void mma_taps(const ConvArgs &A, const T *tile, const int (&abase)[MT],
                                         const T *__restrict__ wf, typename Mma<T>::vec4 (&acc)[MT][NT],
                                         int kq0, int kq_n) {
  const int ntaps = A.k[0] * A.k[1] * A.k[2] * A.kt3;
  const int wstep = A.nt_total << 6;             // floats per (tap, kq)
  const int S4 = 4 * A.S;
  if constexpr (KQ == 0) {                       // any channel count: plain loop
    TapWalk w{0, 0, 0, 0, 0};
    for (int t = 0; t < ntaps; ++t) {
      for (int q = 0; q < kq_n; ++q) {
        T a[MT], b[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = wf[(int64_t(t) * A.kq_total + kq0 + q) * wstep + (nt << 6)];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = tile[abase[mt] + w.off + q * S4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = Mma<T>::mma(a[mt], b[nt], acc[mt][nt]);
      }
      w.next(A, ntaps);
    }
  } else {
    T a0[KQ][MT], b0[KQ][NT], a1[KQ][MT], b1[KQ][NT];
    auto request = [&](T (&a)[KQ][MT], T (&b)[KQ][NT], const TapWalk &w) {
      const T *__restrict__ wt = wf + (int64_t(w.tap) * A.kq_total + kq0) * wstep;
#pragma unroll
      for (int q = 0; q < KQ; ++q)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[q][nt] = wt[q * wstep + (nt << 6)];
#pragma unroll
      for (int q = 0; q < KQ; ++q)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[q][mt] = tile[abase[mt] + w.off + q * S4];
    };
    auto multiply = [&](const T (&a)[KQ][MT], const T (&b)[KQ][NT]) {
#pragma unroll
      for (int q = 0; q < KQ; ++q)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = Mma<T>::mma(a[q][mt], b[q][nt], acc[mt][nt]);
    };
    TapWalk w{0, 0, 0, 0, 0};
    request(a0, b0, w);
    for (int t = 0; t < ntaps; t += 2) {
      w.next(A, ntaps);
      request(a1, b1, w);
      __builtin_amdgcn_sched_barrier(0);     // keep the requests AHEAD of this tap's MFMAs
      multiply(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < ntaps) {
        w.next(A, ntaps);
        request(a0, b0, w);
        __builtin_amdgcn_sched_barrier(0);
        multiply(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// Kernel rows: when the fastest kernel axis has extent 3 (the ConvAct default) one pipelined
// iteration covers the 3 taps of a row at once -- their LDS offsets differ by 1 element, which the
// ds_read encodes as an immediate -- so the uniform walk / address arithmetic is paid once per
// 3*KQ*MT*NT MFMAs instead of once per KQ*MT*NT.
template <typename T, int MT, int NT, int KQ, int K3>
__device__ __forceinline__ void mma_rows(const ConvArgs &A, const T *tile, const int (&abase)[MT],
                                         const T *__restrict__ wf, typename Mma<T>::vec4 (&acc)[MT][NT],
                                         int kq0) {
  const int nrows = A.k[0] * A.k[1] * A.k[2];
  const int wstep = A.nt_total << 6;
  const int S4 = 4 * A.S;
  const int h3 = A.hal[3], h2 = A.hal[2], h1 = A.hal[1];
  T a0[K3][KQ][MT], b0[K3][KQ][NT], a1[K3][KQ][MT], b1[K3][KQ][NT];
  int j1 = 0, j2 = 0, off = 0, row = 0;                 // uniform walk over (j0, j1, j2)
  auto next = [&]() {
    if (row + 1 >= nrows) return;                       // clamp at the last row (harmless re-read)
    ++row;
    off += h3;
    if (++j2 == A.k[2]) {
      j2 = 0;
      off += (h2 - A.k[2]) * h3;
      if (++j1 == A.k[1]) {
        j1 = 0;
        off += (h1 - A.k[1]) * h2 * h3;
      }
    }
  };
  auto request = [&](T (&a)[K3][KQ][MT], T (&b)[K3][KQ][NT]) {
    const T *__restrict__ wt = wf + (int64_t(NF_DBG(A, 8) ? 0 : row) * K3 * A.kq_total + kq0) * wstep;   // dbg 8: L1-resident weights (ablation)
#pragma unroll
    for (int j3 = 0; j3 < K3; ++j3)
#pragma unroll
      for (int q = 0; q < KQ; ++q)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[j3][q][nt] = wt[(int64_t(j3) * A.kq_total + q) * wstep + (nt << 6)];
#pragma unroll
    for (int q = 0; q < KQ; ++q)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const T *pa = tile + abase[mt] + off + q * S4;
#pragma unroll
        for (int j3 = 0; j3 < K3; ++j3) a[j3][q][mt] = pa[j3];
      }
  };
  auto multiply = [&](const T (&a)[K3][KQ][MT], const T (&b)[K3][KQ][NT]) {
#pragma unroll
    for (int j3 = 0; j3 < K3; ++j3)
#pragma unroll
      for (int q = 0; q < KQ; ++q)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = Mma<T>::mma(a[j3][q][mt], b[j3][q][nt], acc[mt][nt]);
  };
  // Rows are taken in pairs inside ONE basic block (no branch between a request and the multiply it
  // overlaps with: a conditional there let the optimizer sink the prefetch loads to their use and
  // exposed a full L2 latency per row); an odd last row is peeled off after the loop.
  request(a0, b0);
  int r = 0;
  for (; r + 1 < nrows; r += 2) {
    next();
    request(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    multiply(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    next();
    request(a0, b0);                                    // (clamped re-read when r + 2 == nrows)
    __builtin_amdgcn_sched_barrier(0);
    multiply(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (r < nrows) multiply(a0, b0);
}

// K-packed variant for cin % 4 != 0 (e.g. the 1 -> 8 first layer of ConvAct): the reduction
// index kk = tap * cin + ci is cut into steps of 4 regardless of tap boundaries, so no MFMA
// k-slot is wasted on channel padding.  The k-group g of a lane then needs its OWN LDS offset
// per step; the offsets live in a small LDS table koff[g][step] (built once per workgroup) and
// are fetched four steps at a time with one ds_read_b128.
template <typename T, int MT, int NT>
__device__ __forceinline__ void mma_packed(const ConvArgs &A, const T *tile, const int *koff,
                                           const int (&abase)[MT], const T *__restrict__ wf,
                                           typename Mma<T>::vec4 (&acc)[MT][NT]) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const int wstep = A.nt_total << 6;
  const int ngroups = A.ns >> 2;
  T a0[4][MT], b0[4][NT], a1[4][MT], b1[4][NT];
  auto request = [&](T (&a)[4][MT], T (&b)[4][NT], int grp) {
    const i32x4 off = *reinterpret_cast<const i32x4 *>(koff + (grp << 2));
    const T *__restrict__ wt = wf + int64_t(grp) * (4 * wstep);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[i][nt] = wt[i * wstep + (nt << 6)];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[i][mt] = tile[abase[mt] + off[i]];
  };
  auto multiply = [&](const T (&a)[4][MT], const T (&b)[4][NT]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = Mma<T>::mma(a[i][mt], b[i][nt], acc[mt][nt]);
  };
  request(a0, b0, 0);
  for (int grp = 0; grp < ngroups; grp += 2) {
    request(a1, b1, grp + 1 < ngroups ? grp + 1 : grp);
    __builtin_amdgcn_sched_barrier(0);
    multiply(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    if (grp + 1 < ngroups) {
      request(a0, b0, grp + 2 < ngroups ? grp + 2 : grp + 1);
      __builtin_amdgcn_sched_barrier(0);
      multiply(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Register budget: the fp32 MT=2 variants are asked to fit 3 waves per SIMD (<= 168 VGPR+AGPR, no spills);
// left alone the allocator takes 201 registers for the 8->46 kernel and only 2 workgroups fit a CU
// (census by HW_ID: 1.78 resident workgroups per CU), which starves the matrix pipe during staging.
template <typename T, int MT, int NT, bool COMPACT, int FUSE>
__global__ __launch_bounds__(kBlock, (sizeof(T) == 4 && MT == 2) ? 3 : 2) void conv_kernel(ConvArgs A) {
  static_assert(FUSE == 0 || sizeof(T) == 4, "the fused coupling epilogue is fp32 only");
  typedef typename Mma<T>::vec4 acc_t;
  extern __shared__ __align__(16) unsigned char smem_conv[];
  T *tile = reinterpret_cast<T *>(smem_conv);
  __shared__ double red[kBlock / kWave];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int nwaves = kBlock / kWave;
  stamp(A, 0);
  if (NF_STAMPS(A) && threadIdx.x == 0) {     // diagnostic: which CU hosts this workgroup
    const unsigned id = blockIdx.y * gridDim.x + blockIdx.x;
    if (id < 4096u) {
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      A.stamps[id * 8 + 7] = ((unsigned long long)(xcc & 0xf) << 16) | ((hw >> 8) & 0xff);   // xcc | se,sh,cu
    }
  }
  // ---- which box
  int bid = blockIdx.x;
  int o[4];
#pragma unroll
  for (int mu = 3; mu >= 0; --mu) {
    o[mu] = (bid % A.nbox[mu]) * A.box[mu];
    bid /= A.nbox[mu];
  }
  const int b = blockIdx.y;
  const T *__restrict__ in_b = static_cast<const T *>(A.in) + int64_t(b) * A.cin * A.V;
  const int r0 = A.k[0] >> 1, r1 = A.k[1] >> 1, r2 = A.k[2] >> 1, r3 = A.k[3] >> 1;
  const int h0 = A.hal[0], h1 = A.hal[1], h2 = A.hal[2], h3 = A.hal[3];

  // ---- stage the input box + halo.
  // (1) every thread of the workgroup resolves ONE halo row (z0, z1, z2) -> (offset of that
  //     row in a channel plane of the input, offset in a channel plane of the LDS tile); the
  //     divisions / wrap-arounds happen once per row here, in parallel, not in the copy loop;
  // (2) each wave then copies rows w, w+4, ...: two broadcast LDS reads per row, then per
  //     channel one coalesced global load and one LDS store, four channels in flight.
  const int R = h0 * h1 * h2;
  const int tile_ints = int(sizeof(T) / 4) * A.cchunk * A.S;    // ints occupied by the staged channel planes
  int *rowsrc = reinterpret_cast<int *>(tile) + tile_ints;
  int *rowdst = rowsrc + R;
  if (!NF_DBG(A, 1)) {
    for (int t = threadIdx.x; t < R; t += kBlock) {
      const int z0 = t / (h1 * h2), rem = t - z0 * (h1 * h2);
      const int z1 = rem / h2, z2 = rem - z1 * h2;
      const int x0 = wrap(o[0] + z0 - r0, A.L[0]), x1 = wrap(o[1] + z1 - r1, A.L[1]),
                x2 = wrap(o[2] + z2 - r2, A.L[2]);
      rowsrc[t] = ((x0 * A.L[1] + x1) * A.L[2] + x2) * A.L[3];     // < V <= 2^31 (checked by the launcher)
      rowdst[t] = t * h3;
    }
    if (A.packed) {                                     // koff[g][step], see mma_packed
      int *koff = reinterpret_cast<int *>(tile) + ((tile_ints + 2 * R + 3) & ~3);   // 16-B aligned
      const int ktot = A.k[0] * A.k[1] * A.k[2] * A.kt3 * A.cin;
      for (int t = threadIdx.x; t < 4 * A.ns; t += kBlock) {
        const int gq = t / A.ns, st = t - gq * A.ns;
        const int kk = 4 * st + gq;
        int off = 0;
        if (kk < ktot) {
          int tap = kk / A.cin;
          const int ci = kk - tap * A.cin;
          const int j3 = tap % A.kt3; tap /= A.kt3;
          const int j2 = tap % A.k[2]; tap /= A.k[2];
          const int j1 = tap % A.k[1];
          const int j0 = tap / A.k[1];
          off = ((j0 * h1 + j1) * h2 + j2) * h3 + j3 + ci * A.S;
        }
        koff[t] = off;
      }
    }
  }
  // copies channels [c0, c0 + cc) of the box + halo into planes [0, cc) of the tile
  auto stage = [&](int c0, int cc) {
    if (NF_DBG(A, 1)) return;
    const int creal = A.cin - c0 < cc ? (A.cin - c0 > 0 ? A.cin - c0 : 0) : cc;
    for (int i = threadIdx.x; i < (cc - creal) * A.S; i += kBlock) tile[creal * A.S + i] = T(0);
    for (int z3b = 0; z3b < h3; z3b += kWave) {         // 64-wide chunks of the fastest axis
      const int z3 = z3b + lane;
      const bool in_row = z3 < h3;
      const int x3 = wrap(o[3] + z3 - r3, A.L[3]);
      // RG rows x 4 channels = 16 independent loads in flight per wave before the first LDS
      // store (4 in flight left the copy latency-bound: ~300 cycles per load, measured by stamps)
      constexpr int RG = 4;
      for (int r0 = wave; r0 < R; r0 += nwaves * RG) {
        int src[RG], dst[RG];
#pragma unroll
        for (int j = 0; j < RG; ++j) {
          const int r = r0 + j * nwaves;
          const int rr = r < R ? r : r0;                 // clamp (harmless duplicate of row r0)
          src[j] = rowsrc[rr] + x3;
          dst[j] = r < R ? rowdst[rr] + z3 : -1;
        }
        int c = 0;
        for (; c + 4 <= creal; c += 4) {
          T v[RG][4];
#pragma unroll
          for (int j = 0; j < RG; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
              v[j][i] = in_row ? in_b[int64_t(c0 + c + i) * A.V + src[j]] : T(0);
#pragma unroll
          for (int j = 0; j < RG; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (in_row && dst[j] >= 0) tile[(c + i) * A.S + dst[j]] = v[j][i];
        }
        for (; c < creal; ++c) {
          T v[RG];
#pragma unroll
          for (int j = 0; j < RG; ++j) v[j] = in_row ? in_b[int64_t(c0 + c) * A.V + src[j]] : T(0);
#pragma unroll
          for (int j = 0; j < RG; ++j)
            if (in_row && dst[j] >= 0) tile[c * A.S + dst[j]] = v[j];
        }
      }
    }
  };

  // ---- per-lane A-fragment bases: unit -> box coordinates (box dims are powers of two)
  const int g = lane >> 4;                       // k-group of the MFMA fragment
  const bool sh2 = !COMPACT && A.sh2;                 // two-site column packing (NT == 1 launches only)
  const int lb3 = (COMPACT || sh2) ? A.lbox[3] - 1 : A.lbox[3];
  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int u = ((wave * MT + mt) << 4) + (lane & 15);
    const int p3 = u & ((1 << lb3) - 1);
    u >>= lb3;
    const int z2 = u & (A.box[2] - 1);
    u >>= A.lbox[2];
    const int z1 = u & (A.box[1] - 1);
    u >>= A.lbox[1];
    const int z0 = u;
    int z3 = sh2 ? 2 * p3 : p3;
    if (COMPACT) z3 = 2 * p3 + ((A.parity + o[0] + z0 + o[1] + z1 + o[2] + z2) & 1);
    abase[mt] = ((z0 * h1 + z1) * h2 + z2) * h3 + z3 + (A.packed ? 0 : g * A.S);
  }

  acc_t acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc_t{T(0), T(0), T(0), T(0)};

  // ---- K loop: channel chunks (all of cin at once when it fits LDS) x taps.  Taps are software
  // pipelined by hand: the A (LDS) and B (global, L1/L2) fragments of tap t+1 are requested
  // before the KQ*MT*NT MFMAs of tap t issue; the loop is unrolled twice over two named fragment
  // buffers so that no register rotation (and hence no early s_waitcnt) sits between a request
  // and its use one tap later.
  const T *__restrict__ wf = static_cast<const T *>(A.wfrag) + (int64_t(A.nt0) << 6) + lane;
  __syncthreads();                                      // row tables are visible
  for (int c0 = 0; c0 < A.cin_pad; c0 += A.cchunk) {
    const int cc = A.cin_pad - c0 < A.cchunk ? A.cin_pad - c0 : A.cchunk;
    if (c0) __syncthreads();                            // every wave is done reading the previous chunk
    stamp(A, c0 ? 3 : 1);
    stage(c0, cc);
    __syncthreads();
    stamp(A, c0 ? 4 : 2);
    if (NF_DBG(A, 2)) continue;
    if (A.packed) {
      mma_packed<T, MT, NT>(A, tile, reinterpret_cast<const int *>(tile) + ((tile_ints + 2 * R + 3) & ~3) + g * A.ns,
                            abase, wf, acc);
    } else {
      // whole-row pipelining doubles the fragment registers: only where the accumulator tile leaves room
      // (MT*NT = 12 would exceed 256 VGPR+AGPR and drop to one wave per SIMD)
      const int rows = (NF_DBG(A, 4) || MT * NT > 8) ? 0 : A.kt3;
      switch (cc >> 2) {
        case 1: if (rows == 3) mma_rows<T, MT, NT, 1, 3>(A, tile, abase, wf, acc, c0 >> 2);
                else if (rows == 4) mma_rows<T, MT, NT, 1, 4>(A, tile, abase, wf, acc, c0 >> 2);
                else mma_taps<T, MT, NT, 1>(A, tile, abase, wf, acc, c0 >> 2, 1);
                break;
        case 2: if (rows == 3) mma_rows<T, MT, NT, 2, 3>(A, tile, abase, wf, acc, c0 >> 2);
                else if (rows == 4) mma_rows<T, MT, NT, 2, 4>(A, tile, abase, wf, acc, c0 >> 2);
                else mma_taps<T, MT, NT, 2>(A, tile, abase, wf, acc, c0 >> 2, 2);
                break;
        case 4: mma_taps<T, MT, NT, 4>(A, tile, abase, wf, acc, c0 >> 2, 4); break;
        default: mma_taps<T, MT, NT, 0>(A, tile, abase, wf, acc, c0 >> 2, cc >> 2); break;
      }
    }
  }

  stamp(A, 5);
  conv_epilogue<T, MT, NT, COMPACT, FUSE>(A, o, b, int64_t(b) * gridDim.x + blockIdx.x, acc, tile, red, wave, lane);
  stamp(A, 6);
}

static int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

template <typename T, int MT, int NT, bool COMPACT, int FUSE>
static void launch_one(const ConvArgs &A, dim3 grid, size_t lds, hipStream_t stream) {
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_kernel<T, MT, NT, COMPACT, FUSE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
  hipLaunchKernelGGL((conv_kernel<T, MT, NT, COMPACT, FUSE>), grid, dim3(kBlock), lds, stream, A);
}

template <typename T, int MT, int NT>
static void launch(const ConvArgs &A, dim3 grid, size_t lds, int fuse, hipStream_t stream) {
  if constexpr (sizeof(T) == 4) {
    if (fuse == 1) return launch_one<T, MT, NT, true, 1>(A, grid, lds, stream);
    if (fuse == 2) return launch_one<T, MT, NT, true, 2>(A, grid, lds, stream);
  }
  if (A.compact) launch_one<T, MT, NT, true, 0>(A, grid, lds, stream);
  else launch_one<T, MT, NT, false, 0>(A, grid, lds, stream);
}

}  // namespace nf

using namespace nf;

// 1 if nf_conv_fwd computes this layer with two-site column packing (then the weights must be
// packed as 16 output columns over k3+1 taps along the fastest axis, see include/normflow_hip.h)
extern "C" int nf_conv_two_site(int cout, int compact, int l3, int k3) {
  static const int off = NF_DIAG_ENV_INT("NF_CONV_NO_TWO_SITE", 0);
  return !off && cout <= 8 && !compact && l3 % 2 == 0 && l3 >= 4 && (k3 & 1);
}
extern "C" int nf_conv_cin_pad(int cin) { return (cin + 3) & ~3; }
// number of 4-wide reduction steps of the K-packed weight layout (cin % 4 != 0), 0 otherwise
extern "C" int nf_conv_packed_steps(int cin, int ntaps) {
  if (cin % 4 == 0 || cin >= 8) return 0;
  return ((((cin * ntaps + 3) / 4) + 3) / 4) * 4;
}
extern "C" int nf_conv_ntiles(int cout) { return (cout + 15) >> 4; }

static thread_local int g_last_path = 0;
static thread_local int *g_dry_layout = nullptr;     // non-null: plan only (nf_conv_weight_layout)
extern "C" int nf_conv_last_path(void) { return g_last_path; }

struct FuseInfo {           // non-null => the coupling epilogue replaces the store
  int flags;                 // NF_CONV_UNIT_INPUT: |input| <= 1 guaranteed -> the split-fp16 kernel (nf_conv_h.hip) may run
  int mode;                 // 1 forward, 2 inverse
  const float *xact;
  float *yout;
  double *partial;
  size_t partial_bytes;
  RqsParams P;
  int64_t *blocks_out;
};

template <typename T>
static int run_conv_t(const void *in, const void *wfrag, const void *bias, void *out, int64_t B,
                      const int32_t *lattice, const int32_t *ksize, int cin, int cout, int act, int compact,
                      int active_parity, hipStream_t stream, const FuseInfo *fz) {
  NF_REQUIRE(in && wfrag && (out || fz) && lattice && ksize, "nf_conv_fwd: NULL pointer");
  NF_REQUIRE(B >= 0 && B <= 65535, "nf_conv_fwd: batch %lld outside [0, 65535]", (long long)B);
  NF_REQUIRE(cin >= 1 && cout >= 1, "nf_conv_fwd: bad channel counts");
  NF_REQUIRE(act >= kActNone && act <= kActSigmoid, "nf_conv_fwd: unknown activation code %d", act);
  ConvArgs A{};
  int64_t V = 1;
  for (int mu = 0; mu < 4; ++mu) {
    NF_REQUIRE(lattice[mu] >= 1 && ksize[mu] >= 1 && (ksize[mu] & 1), "nf_conv_fwd: lattice dims >= 1 and odd kernel sizes needed");
    A.L[mu] = lattice[mu];
    A.k[mu] = ksize[mu];
    V *= lattice[mu];
  }
  const int out_split16 = compact == 2 ? 1 : 0;     // NF_OUT_SPLIT16: full lattice, fp16 (hi, lo) pairs, channel-last
  if (out_split16) compact = 0;
  if (compact) NF_REQUIRE(A.L[3] % 2 == 0, "nf_conv_fwd: pair-compact output needs an even fastest axis");
  if (B == 0 || V == 0) return NF_OK;
  A.in = in;
  A.wfrag = wfrag;
  A.bias = bias;
  A.out = out;
  A.V = V;
  A.cin = cin; A.cin_pad = (cin + 3) & ~3; A.cout = cout; A.kq = A.cin_pad / 4;
  A.kq_total = A.kq; A.cchunk = A.cin_pad;
  A.sh2 = (!fz && nf_conv_two_site(cout, compact, A.L[3], A.k[3])) ? 1 : 0;
  A.kt3 = A.k[3] + A.sh2;
  if (A.sh2) A.nt_total = 1;
  A.nt_total = (cout + 15) >> 4;
  A.act = act; A.compact = compact ? 1 : 0; A.parity = active_parity & 1;
  A.out_split16 = out_split16;
  if (out_split16)
    NF_REQUIRE(sizeof(T) == 4 && cout == 8 && A.sh2 && A.L[3] % 4 == 0 && !fz,
               "nf_conv_fwd: split-fp16 output needs fp32, 8 output channels and a fastest axis that is a multiple of 4");
  A.packed = (cin % 4) != 0 && cin < 8;   // larger odd channel counts: zero-pad to a multiple of 4 and chunk
  if (A.packed) {
    const int ktot = cin * ksize[0] * ksize[1] * ksize[2] * A.kt3;
    A.ns = ((((ktot + 3) / 4) + 3) / 4) * 4;   // steps of 4 k, rounded up to groups of 4 steps
    A.cin_pad = cin;                           // no channel padding in this mode
    A.kq = 0; A.kq_total = 0; A.cchunk = cin;
  }
  {
    static const int dbg = NF_DIAG_ENV_INT("NF_CONV_DBG", 0);
    A.dbg = dbg;
  }

  // ---- box: (4 waves x MT tiles x 16) output units per workgroup, powers of two, long along
  // the fastest axis (coalescing), then as cubic as the lattice allows (least halo).  MT = 4
  // unless the staged box would then exceed ~80 KiB of LDS (two workgroups per CU keep one
  // staging while the other multiplies); then MT = 2.
  static const int lds_cap_kb = NF_DIAG_ENV_INT("NF_CONV_LDS_KB", 40);
  static const int box3_cap = NF_DIAG_ENV_INT("NF_CONV_BOX3", 32);
  static const int mt_first = NF_DIAG_ENV_INT("NF_CONV_MT", 4);
  // layers the persistent kernel (nf_conv_pipe.hip) can take are planned with its MT = 2 boxes straight away
  const int pipe_off = !option(NF_OPT_PIPE);
  const bool pipe_candidate = sizeof(T) == 4 && !pipe_off && !A.packed && A.k[3] == 3 &&
                              A.k[0] * A.k[1] * A.k[2] >= 2 && A.nt_total <= 3 && !NF_DBG(A, 15);   // (dbg bits >= 16 are timing ablations inside the persistent kernels)
  int MT = (mt_first == 2 || pipe_candidate) ? 2 : 4;
  int box[4];
  for (int attempt = 0; attempt < 2; ++attempt) {
    const int units = (kBlock / kWave) * MT * 16;
    const int target = (compact || A.sh2) ? 2 * units : units;        // sites in the box
    int cap[4];
    for (int mu = 0; mu < 4; ++mu) { cap[mu] = 1 << ilog2(A.L[mu]); box[mu] = 1; }
    box[3] = cap[3] < box3_cap ? cap[3] : box3_cap;
    const int min3 = (compact || A.sh2) ? 8 : 4;            // >= 4 units along the fastest axis
    while (box[3] > min3 && A.L[3] % box[3] != 0) box[3] >>= 1;   // prefer boxes that tile the fastest axis (48 -> 3 x 16)
    if (box[3] < min3) box[3] = min3;
    int vol = box[3];
    while (vol < target) {
      int best = -1;
      for (int mu = 2; mu >= 0; --mu)
        if (box[mu] < cap[mu] && (best < 0 || box[mu] < box[best])) best = mu;
      if (best < 0) {
        if (box[3] < cap[3]) best = 3; else break;
      }
      box[best] *= 2;
      vol *= 2;
    }
    while (vol < target) {   // tiny lattice: pad the fastest axis (extra units are masked out)
      box[3] *= 2;
      vol *= 2;
    }
    while (vol > target && box[3] > min3) {   // min3 may have overshot a tiny target
      box[3] /= 2;
      vol /= 2;
    }
    int64_t hv = 1;
    for (int mu = 0; mu < 4; ++mu) hv *= box[mu] + A.k[mu] - 1;
    static const int cplan_max = NF_DIAG_ENV_INT("NF_CONV_CPLAN", 8);
    const int cplan = A.packed ? cin : (A.cin_pad < cplan_max ? A.cin_pad : cplan_max);   // channels planned per K pass
    if (MT == 2 || hv * cplan * int64_t(sizeof(T)) <= int64_t(lds_cap_kb) * 1024) break;
    MT = 2;
  }
  int64_t nblocks = 1;
  int64_t halvol = 1;
  for (int mu = 0; mu < 4; ++mu) {
    A.box[mu] = box[mu];
    A.lbox[mu] = ilog2(box[mu]);
    A.nbox[mu] = (A.L[mu] + box[mu] - 1) / box[mu];
    A.hal[mu] = box[mu] + A.k[mu] - 1;
    nblocks *= A.nbox[mu];
    halvol *= A.hal[mu];
  }
  // plane stride: odd for the stride-2 reads of compact mode, = 16 mod 32 otherwise, so the
  // four k-groups of an A fragment fall on disjoint LDS banks
  int S = int(halvol);
  if (compact || A.sh2) S |= 1; else S = ((S + 15) & ~31) + 16;
  A.S = S;
  int64_t rows = 1;
  for (int mu = 0; mu < 3; ++mu) rows *= A.hal[mu];
  NF_REQUIRE(V < (int64_t(1) << 31), "nf_conv_fwd: lattice volume must be < 2^31");
  if (!A.packed) {   // channels per K pass: as many as fit ~78 KiB (at least 4), in steps the pipelined loop knows
    const int fit = int((int64_t(lds_cap_kb) * 1024) / (int64_t(S) * sizeof(T)));
    int cc = A.cin_pad;
    if (cc > fit) cc = fit >= 16 ? 16 : (fit >= 8 ? 8 : 4);
    A.cchunk = cc;
  }
  size_t lds = size_t(A.cchunk) * S * sizeof(T) + size_t(rows) * 2 * sizeof(int) +
                     (A.packed ? size_t(4) * A.ns * sizeof(int) + 16 : 0);
  if (fz) {
    const size_t stage = size_t(48) * ((kBlock / kWave) * MT * 16 + 4) * sizeof(float);
    if (lds < stage) lds = stage;
  }
  if (A.sh2) {       // the two-site epilogue transposes the box through LDS: 8 channels x (sites of the box + 8)
    const size_t stage = size_t(8) * (2 * (kBlock / kWave) * MT * 16 + 8) * sizeof(T);
    if (lds < stage) lds = stage;
  }
  {
    static const int pad_kb = NF_DIAG_ENV_INT("NF_CONV_LDS_PAD_KB", 0);   // occupancy experiments
    lds += size_t(pad_kb) * 1024;
  }
  NF_REQUIRE(lds <= 160 * 1024, "nf_conv_fwd: input box needs %zu B of LDS (> 160 KiB): cin=%d, kernel %dx%dx%dx%d",
             lds, cin, A.k[0], A.k[1], A.k[2], A.k[3]);
  NF_REQUIRE(nblocks <= 0x7fffffff, "nf_conv_fwd: lattice too large");
  const dim3 grid = dim3(static_cast<unsigned>(nblocks), static_cast<unsigned>(B), 1u);
  int fuse = 0;
  if (fz) {
    fuse = fz->mode;
    A.xact = fz->xact; A.yout = fz->yout; A.partial = fz->partial; A.P = fz->P;
    A.field16 = (fz->flags & NF_CONV_FIELD_F16) ? 1 : 0;
    NF_REQUIRE(A.nt_total <= 3, "nf_conv_rqs: at most 48 logit channels can be fused");
  }
  g_last_path = 0;
  // the split-fp16 fused kernel (nf_conv_h.hip) first: it has its own box (2 x 2 x 2 rows x a 32-site segment), hence its own
  // number of log-det partials per sample
  if (fz && (fz->flags & NF_CONV_UNIT_INPUT)) {
    if constexpr (sizeof(T) == 4) {
      if (fz->flags & NF_CONV_SPLIT16_INPUT) A.in_split16 = 1;
      int64_t nbh = 0;
      if (conv_h_eligible(A, fuse, &nbh)) {
        if (g_dry_layout) { *g_dry_layout = NF_WLAYOUT_SPLIT16; return NF_OK; }
        const size_t need = size_t(B) * size_t(nbh) * sizeof(double);
        if (fz->partial == nullptr || fz->partial_bytes < need) {
          set_error("nf_conv_rqs: workspace %zu B < %zu B needed", fz->partial_bytes, need);
          return NF_EWORKSPACE;
        }
        *fz->blocks_out = nbh;
        const int pr = launch_conv_h(A, B, fuse, stream, false);
        if (pr == -2) { set_error("nf_conv_rqs: batch x boxes >= 2^31 work items, split the batch"); return NF_EINVAL; }
        if (pr != 1) { set_error("nf_conv_rqs: could not launch the split-fp16 kernel"); return NF_ELAUNCH; }
        g_last_path = 3;
        return check_launch("conv split-fp16 kernel");
      }
    }
  }
  if (fz && (fz->flags & NF_CONV_SPLIT16_INPUT) && !g_dry_layout) {
    set_error("nf_conv_rqs: split-fp16 input but the layer is not eligible for the split-fp16 kernel");
    return NF_EINVAL;
  }
  if (fz) {
    const size_t need = size_t(B) * size_t(nblocks) * sizeof(double);
    if (fz->partial == nullptr || fz->partial_bytes < need) {
      set_error("nf_conv_rqs: workspace %zu B < %zu B needed", fz->partial_bytes, need);
      return NF_EWORKSPACE;
    }
    *fz->blocks_out = nblocks;
  }
  if (g_dry_layout) {                  // nf_conv_weight_layout: report which kernel (hence weight layout) this layer gets
    int pr = 0;
    if constexpr (sizeof(T) == 4) {
      if (MT == 2) { A.nt0 = 0; pr = launch_conv_pipe(A, B, nblocks, fuse, stream, true); }
    }
    *g_dry_layout = pr == 1 ? 1 : 0;
    return NF_OK;
  }
  if constexpr (sizeof(T) == 4) {
    if (cin == 1 && A.sh2 && !fz) {     // first ConvAct layer: data-movement kernel (nf_conv_pipe.hip, K5c)
      const int pr = launch_conv_c1(A, MT, B, nblocks, stream);
      if (pr == -2) { set_error("nf_conv_fwd: batch x boxes >= 2^31 work items, split the batch"); return NF_EINVAL; }
      if (pr < 0) { set_error("nf_conv_fwd: could not launch the single-channel kernel"); return NF_ELAUNCH; }
      if (pr == 1) { g_last_path = 2; return check_launch("conv c1 kernel"); }
    }
  }
  if constexpr (sizeof(T) == 4) {
    if (MT == 2) {     // persistent, staging-overlapped variant (nf_conv_pipe.hip) when the layer is eligible
      A.nt0 = 0;
      const int pr = launch_conv_pipe(A, B, nblocks, fuse, stream, false);
      if (pr == -2) { set_error("nf_conv_fwd: batch x boxes >= 2^31 work items, split the batch"); return NF_EINVAL; }
      if (pr < 0) { set_error("nf_conv_fwd: could not launch the pipelined kernel"); return NF_ELAUNCH; }
      if (pr == 1) { g_last_path = 1; return check_launch("conv pipe kernel"); }
    }
  }
  static const int want_stamps = NF_DIAG_ENV_INT("NF_CONV_STAMPS", 0);
  unsigned long long *d_stamps = nullptr;
  if (want_stamps) {      // DIAGNOSTIC ONLY: allocates and synchronises, never enabled in product use
    (void)hipMalloc(&d_stamps, 4096 * 8 * sizeof(unsigned long long));
    (void)hipMemset(d_stamps, 0, 4096 * 8 * sizeof(unsigned long long));
    A.stamps = d_stamps;
  }
  for (int nt0 = 0; nt0 < A.nt_total; nt0 += 3) {
    A.nt0 = nt0;
    const int n = A.nt_total - nt0 >= 3 ? 3 : A.nt_total - nt0;
    if (MT == 4) {
      if (n == 3) launch<T, 4, 3>(A, grid, lds, fuse, stream);
      else if (n == 2) launch<T, 4, 2>(A, grid, lds, fuse, stream);
      else launch<T, 4, 1>(A, grid, lds, fuse, stream);
    } else {
      if (n == 3) launch<T, 2, 3>(A, grid, lds, fuse, stream);
      else if (n == 2) launch<T, 2, 2>(A, grid, lds, fuse, stream);
      else launch<T, 2, 1>(A, grid, lds, fuse, stream);
    }
    const int rc = check_launch("conv kernel");
    if (rc) return rc;
  }
  if (want_stamps) {
    (void)hipStreamSynchronize(stream);
    static unsigned long long h[4096 * 8];
    (void)hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d_stamps);
    double seg[6] = {0, 0, 0, 0, 0, 0};
    int n = 0;
    for (int i = 1024; i < 4096; ++i) {     // skip the first dispatch round
      const unsigned long long *t = h + i * 8;
      if (!t[0] || !t[5]) continue;
      ++n;
      seg[0] += double(t[1] - t[0]); seg[1] += double(t[2] - t[1]);
      seg[2] += t[3] ? double(t[3] - t[2]) : double(t[5] - t[2]);
      seg[3] += t[3] ? double(t[4] - t[3]) : 0.0;
      seg[4] += t[3] ? double(t[5] - t[4]) : 0.0;
      seg[5] += t[6] ? double(t[6] - t[5]) : 0.0;
    }
    {
      // mean number of co-resident workgroups per CU (per-CU windows: clock counters differ between XCDs)
      static double busy[16 * 256];
      static unsigned long long lo[16 * 256], hi[16 * 256];
      for (int k = 0; k < 16 * 256; ++k) { busy[k] = 0; lo[k] = ~0ull; hi[k] = 0; }
      for (int i = 0; i < 4096; ++i) {
        const unsigned long long *t = h + i * 8;
        const unsigned long long end = t[6] ? t[6] : t[5];
        if (!t[0] || !end) continue;
        const int key = int(((t[7] >> 16) & 0xf) * 256 + (t[7] & 0xff));
        busy[key] += double(end - t[0]);
        if (t[0] < lo[key]) lo[key] = t[0];
        if (end > hi[key]) hi[key] = end;
      }
      double conc = 0; int cus = 0;
      for (int k = 0; k < 16 * 256; ++k)
        if (hi[k] > lo[k]) { conc += busy[k] / double(hi[k] - lo[k]); ++cus; }
      if (cus) fprintf(stderr, "[nf_conv stamps] %d CUs seen, mean co-resident workgroups per CU (first 4096 workgroups) %.2f\n", cus, conc / cus);
    }
    if (n) fprintf(stderr, "[nf_conv stamps] cin=%d cout=%d compact=%d fuse=%d blocks=%d | cycles: prologue %.0f  stage0 %.0f  mma0 %.0f  stage1 %.0f  mma1 %.0f  epilogue %.0f\n",
                   cin, cout, compact, fuse, n, seg[0] / n, seg[1] / n, seg[2] / n, seg[3] / n, seg[4] / n, seg[5] / n);
  }
  return NF_OK;
}

static int run_conv(const void *in, const void *wfrag, const void *bias, void *out, int64_t B,
                    const int32_t *lattice, const int32_t *ksize, int cin, int cout, int act, int compact,
                    int active_parity, int dtype, hipStream_t stream, const FuseInfo *fz) {
  if (dtype == NF_F32)
    return run_conv_t<float>(in, wfrag, bias, out, B, lattice, ksize, cin, cout, act, compact, active_parity, stream, fz);
  if (dtype == NF_F64 && fz == nullptr)
    return run_conv_t<double>(in, wfrag, bias, out, B, lattice, ksize, cin, cout, act, compact, active_parity, stream, fz);
  set_error("nf_conv: unsupported dtype %d (fp32 and fp64; the fused coupling epilogue is fp32 only)", dtype);
  return NF_EINVAL;
}

extern "C" int nf_conv_fwd(const void *in, const void *wfrag, const void *bias, void *out, int64_t B,
                           const int32_t *lattice, const int32_t *ksize, int cin, int cout, int act,
                           int compact, int active_parity, int dtype, void *stream) {
  return run_conv(in, wfrag, bias, out, B, lattice, ksize, cin, cout, act, compact, active_parity, dtype,
                  static_cast<hipStream_t>(stream), nullptr);
}

extern "C" int nf_conv_weight_layout(const int32_t *lattice, const int32_t *ksize, int cin, int cout, int compact,
                                     int fused, int dtype) {
  if (!lattice || !ksize) return -1;
  int layout = 0;
  int64_t blocks = 0;
  FuseInfo fz{};
  fz.mode = 1;
  fz.flags = ((fused & 2) ? NF_CONV_UNIT_INPUT : 0) | ((fused & 4) ? NF_CONV_SPLIT16_INPUT : 0);
  fz.P.m = (cout + 2) / 3;                                  // knots_len of the fused spline (planning only)
  fz.partial = reinterpret_cast<double *>(uintptr_t(8));
  fz.partial_bytes = ~size_t(0);
  fz.blocks_out = &blocks;
  void *dummy = reinterpret_cast<void *>(uintptr_t(8));   // never dereferenced: planning only
  g_dry_layout = &layout;
  const int rc = run_conv(dummy, dummy, nullptr, dummy, 1, lattice, ksize, cin, cout, 0, fused ? 1 : compact, 0, dtype,
                          nullptr, fused ? &fz : nullptr);
  g_dry_layout = nullptr;
  return rc ? -1 : layout;
}

extern "C" int nf_conv_rqs_supported(int cout, int m) {
  return (m == 4 || m == 8 || m == 16) && cout == 3 * m - 2;
}

// The split-fp16 fused kernel (nf_conv_h.hip) takes any knots_len 2..16 on its lattices (4-D, even extents, fastest axis
// 32 + 16 n sites, 3^4 kernel, 8 hidden channels): a host-side planning query, no launch.
extern "C" int nf_conv_rqs_split16_supported(const int32_t *lattice, int cout, int m) {
  if (!lattice || m < 2 || m > 16 || cout != 3 * m - 2 || !option(NF_OPT_SPLIT16)) return 0;
  if (lattice[3] < 32 || (lattice[3] & 15)) return 0;
  for (int mu = 0; mu < 3; ++mu)
    if (lattice[mu] < 2 || (lattice[mu] & 1)) return 0;
  return 1;
}

extern "C" int nf_conv_rqs(const void *in, const void *wfrag, const void *bias, const void *x_active,
                           const void *log0, void *y, void *logj, int64_t B, const int32_t *lattice,
                           const int32_t *ksize, int cin, int cout, int active_parity,
                           const nf_rqs_opts *opts, int inverse, int flags, void *workspace, size_t workspace_bytes,
                           int dtype, void *stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  NF_REQUIRE(opts && x_active && y && logj && lattice, "nf_conv_rqs: NULL pointer");
  NF_REQUIRE(nf_conv_rqs_supported(cout, opts->m) ||
                 ((flags & NF_CONV_UNIT_INPUT) && (flags & NF_CONV_SPLIT16_INPUT) && cin == 8 &&
                  nf_conv_rqs_split16_supported(lattice, cout, opts->m)),
             "nf_conv_rqs: needs knots_len in {4, 8, 16} (or 2..16 on the split-fp16 chain) and cout = 3m-2 (got m=%d, cout=%d)", opts->m, cout);
  NF_REQUIRE(!opts->fixed_knots_x && !opts->fixed_knots_y, "nf_conv_rqs: fixed knots are not fused");
  NF_REQUIRE(lattice[3] % 2 == 0, "nf_conv_rqs: needs an even fastest axis");
  NF_REQUIRE(opts->xhi > opts->xlo && opts->yhi > opts->ylo, "nf_conv_rqs: empty xlim/ylim");
  if (B == 0) return NF_OK;
  int64_t blocks = 0;
  FuseInfo fz{};
  fz.mode = inverse ? 2 : 1;
  fz.flags = flags;
  fz.xact = static_cast<const float *>(x_active);
  fz.yout = static_cast<float *>(y);
  fz.partial = static_cast<double *>(workspace);
  fz.partial_bytes = workspace_bytes;
  fz.P.xlo = opts->xlo; fz.P.xhi = opts->xhi; fz.P.ylo = opts->ylo; fz.P.yhi = opts->yhi;
  fz.P.fx = nullptr; fz.P.fy = nullptr;
  fz.P.m = opts->m; fz.P.el = opts->extrap_left; fz.P.er = opts->extrap_right;
  fz.blocks_out = &blocks;
  const int rc = run_conv(in, wfrag, bias, nullptr, B, lattice, ksize, cin, cout, 0, 1, active_parity, dtype,
                          stream, &fz);
  if (rc) return rc;
  return launch_finalize<float>(fz.partial, blocks, log0, logj, B, stream);
}

// =====================================================================================
// VJP of the conv layer w.r.t. weights and bias (what autograd derives for ConvAct in
// Fitter.step, src/_normflowcore.py:288):
//     gW[o, i, j] = sum_{b, n} gz[b, o, n] * in[b, i, (n + j - k//2) mod L],   gb[o] = sum gz[b, o, n]
// GEMM view: rows M = output channels, columns N = (tap, input channel) [+ one all-ones column
// that yields the bias gradient], reduction K = lattice sites x batch -- enormous K, tiny M x N.
// So the accumulators stay in registers for the whole launch: workgroups are PERSISTENT, each
// walks over (sample, box) items, stages the input box + halo and the matching gz box in LDS,
// and multiplies with K = the 256 sites of the box; wave w owns the N tiles w*NTW .. w*NTW+NTW-1
// of all M tiles.  One round of float atomics per workgroup at the very end.
namespace nf {

struct WgArgs {
  const void *in;       // (B, cin, V)
  const void *gz;       // (B, cout, V)
  void *gw;             // (cout_pad16, ncols_pad16) accumulators, zeroed by the caller
  int64_t V, nitems;
  int L[4], k[4], box[4], lbox[4], nbox[4], hal[4];
  int S, PS, units;     // in-tile plane stride, gz-tile plane stride, sites per box
  int cin, cout, ntot, ncols_pad, nt0, nboxes;
  int parity;           // conv_wgrad_sites_kernel: >= 0: gz is pair-compact (B, cout, V/2), values at the active sites of this parity
  void *part;           // conv_wgrad_sites_kernel: (gridDim.x, MTW * 16, ncols_pad) partial sums, one slot per workgroup
};

template <typename T> __device__ __forceinline__ void atomic_add(T *p, T v) { atomicAdd(p, v); }

template <typename T, int MTW, int NTW>
__global__ __launch_bounds__(kBlock) void conv_wgrad_kernel(WgArgs A) {
  typedef typename Mma<T>::vec4 acc_t;
  extern __shared__ __align__(16) unsigned char smem_conv[];
  T *tile = reinterpret_cast<T *>(smem_conv);                 // cin planes of the box + halo
  T *gzt = tile + A.cin * A.S;                                // MTW*16 planes of the box
  const int h0 = A.hal[0], h1 = A.hal[1], h2 = A.hal[2], h3 = A.hal[3];
  const int R = h0 * h1 * h2;
  int *rowsrc = reinterpret_cast<int *>(gzt + MTW * 16 * A.PS);
  int *rowdst = rowsrc + R;
  const int lane = threadIdx.x & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int nwaves = kBlock / kWave;
  const int r0 = A.k[0] >> 1, r1 = A.k[1] >> 1, r2 = A.k[2] >> 1, r3 = A.k[3] >> 1;

  // per-lane constants of this wave's columns: LDS offset of (tap, ci), or the ones column
  int coff[NTW];
  bool one[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int n = ((A.nt0 + wave * NTW + nt) << 4) + (lane & 15);
    coff[nt] = 0;
    one[nt] = n == A.ntot;
    if (n < A.ntot) {
      int tap = n / A.cin;
      const int ci = n - tap * A.cin;
      const int j3 = tap % A.k[3]; tap /= A.k[3];
      const int j2 = tap % A.k[2]; tap /= A.k[2];
      const int j1 = tap % A.k[1];
      const int j0 = tap / A.k[1];
      coff[nt] = ((j0 * h1 + j1) * h2 + j2) * h3 + j3 + ci * A.S;
    }
  }
  acc_t acc[MTW][NTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = acc_t{T(0), T(0), T(0), T(0)};

  for (int64_t item = blockIdx.x; item < A.nitems; item += gridDim.x) {
    const int b = int(item / A.nboxes);
    int bid = int(item - int64_t(b) * A.nboxes);
    int o[4];
#pragma unroll
    for (int mu = 3; mu >= 0; --mu) {
      o[mu] = (bid % A.nbox[mu]) * A.box[mu];
      bid /= A.nbox[mu];
    }
    const T *__restrict__ in_b = static_cast<const T *>(A.in) + int64_t(b) * A.cin * A.V;
    const T *__restrict__ gz_b = static_cast<const T *>(A.gz) + int64_t(b) * A.cout * A.V;
    __syncthreads();                                  // previous item fully consumed
    for (int t = threadIdx.x; t < R; t += kBlock) {
      const int z0 = t / (h1 * h2), rem = t - z0 * (h1 * h2);
      const int z1 = rem / h2, z2 = rem - z1 * h2;
      const int x0 = wrap(o[0] + z0 - r0, A.L[0]), x1 = wrap(o[1] + z1 - r1, A.L[1]),
                x2 = wrap(o[2] + z2 - r2, A.L[2]);
      rowsrc[t] = ((x0 * A.L[1] + x1) * A.L[2] + x2) * A.L[3];
      rowdst[t] = t * h3;
    }
    // gz box: plane co, unit u (box order); 0 outside the lattice / beyond cout
    for (int idx = threadIdx.x; idx < MTW * 16 * A.units; idx += kBlock) {
      const int co = idx / A.units;
      int u = idx - co * A.units;
      const int uu = u;
      const int z3 = u & (A.box[3] - 1); u >>= A.lbox[3];
      const int z2 = u & (A.box[2] - 1); u >>= A.lbox[2];
      const int z1 = u & (A.box[1] - 1); u >>= A.lbox[1];
      const int x0 = o[0] + u, x1 = o[1] + z1, x2 = o[2] + z2, x3 = o[3] + z3;
      T v = T(0);
      if (co < A.cout && x0 < A.L[0] && x1 < A.L[1] && x2 < A.L[2] && x3 < A.L[3])
        v = gz_b[int64_t(co) * A.V + ((int64_t(x0) * A.L[1] + x1) * A.L[2] + x2) * A.L[3] + x3];
      gzt[co * A.PS + uu] = v;
    }
    __syncthreads();                                  // row tables visible
    for (int z3b = 0; z3b < h3; z3b += kWave) {
      const int z3 = z3b + lane;
      const bool in_row = z3 < h3;
      const int x3 = wrap(o[3] + z3 - r3, A.L[3]);
      for (int r = wave; r < R; r += nwaves) {
        const int src = rowsrc[r] + x3, dst = rowdst[r] + z3;
        for (int c = 0; c < A.cin; ++c)
          if (in_row) tile[c * A.S + dst] = in_b[int64_t(c) * A.V + src];
      }
    }
    __syncthreads();
    // K loop over the sites of the box, 4 per MFMA
    for (int s = 0; s < A.units; s += 4) {
      int u = s + g;
      const int uu = u;
      const int z3 = u & (A.box[3] - 1); u >>= A.lbox[3];
      const int z2 = u & (A.box[2] - 1); u >>= A.lbox[2];
      const int z1 = u & (A.box[1] - 1); u >>= A.lbox[1];
      const int ab = ((u * h1 + z1) * h2 + z2) * h3 + z3;
      T a[MTW], bq[NTW];
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) a[mt] = gzt[((mt << 4) + (lane & 15)) * A.PS + uu];
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) bq[nt] = one[nt] ? T(1) : tile[ab + coff[nt]];
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = Mma<T>::mma(a[mt], bq[nt], acc[mt][nt]);
    }
  }
  // flush: D col = lane & 15 (n), rows (co) 4g + r (f32) / g + 4r (f64)
  T *gw = static_cast<T *>(A.gw);
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int n = ((A.nt0 + wave * NTW + nt) << 4) + (lane & 15);
      if (n >= A.ncols_pad) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = (mt << 4) + (Mma<T>::kStridedRows ? g + (r << 2) : (g << 2) + r);
        atomic_add(gw + int64_t(co) * A.ncols_pad + n, acc[mt][nt][r]);
      }
    }
}

constexpr int kSitesStage = 10;                // input elements a thread stages per channel (a halo plane of <= 2560 elements)

// The same GEMM for layers with FEW columns (N = taps x cin + 1 <= 14 tiles: 1-, 2- and 3-D kernels), where the split of
// the column tiles over the waves above leaves most waves idle (14 tiles over 4 x 6 slots: 2.3 waves of 4 work; 2 tiles: one
// wave): every wave holds ALL column tiles and the waves split the SITES of a box (K) instead.  A pair-compact cotangent
// (the layer that is evaluated at the active sites only) is read as it stands and only its active sites are walked: half
// the K steps, no expanded copy.  No atomics: the four waves add up through LDS in a fixed order, every workgroup stores one
// partial matrix, wgrad_sites_reduce_kernel adds them to gw in workgroup order -- the gradient is bitwise reproducible.
template <typename T, int MTW, int NTW>
__global__ __launch_bounds__(kBlock) void conv_wgrad_sites_kernel(WgArgs A) {
  typedef typename Mma<T>::vec4 acc_t;
  extern __shared__ __align__(16) unsigned char smem_conv[];
  T *tile = reinterpret_cast<T *>(smem_conv);                 // cin planes of the box + halo
  T *gzt = tile + A.cin * A.S;                                // MTW*16 planes of the box's K entries
  const int h0 = A.hal[0], h1 = A.hal[1], h2 = A.hal[2], h3 = A.hal[3];
  const int R = h0 * h1 * h2;
  int *rowsrc = reinterpret_cast<int *>(gzt + MTW * 16 * A.PS);
  const int lane = threadIdx.x & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int nwaves = kBlock / kWave;
  const int r0 = A.k[0] >> 1, r1 = A.k[1] >> 1, r2 = A.k[2] >> 1, r3 = A.k[3] >> 1;
  const bool compact = A.parity >= 0;
  const int KU = compact ? A.units >> 1 : A.units;            // K entries of a box: its sites, or its active sites (a power of two)
  const int lKU = 31 - __builtin_clz(unsigned(KU));
  const int lb3 = compact ? A.lbox[3] - 1 : A.lbox[3];        // bits of the position along the fastest axis (pairs / sites)
  const int64_t gzV = compact ? A.V >> 1 : A.V;

  int coff[NTW];
  bool one[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int n = (nt << 4) + (lane & 15);
    coff[nt] = 0;
    one[nt] = n == A.ntot;
    if (n < A.ntot) {
      int tap = n / A.cin;
      const int ci = n - tap * A.cin;
      const int j3 = tap % A.k[3]; tap /= A.k[3];
      const int j2 = tap % A.k[2]; tap /= A.k[2];
      const int j1 = tap % A.k[1];
      const int j0 = tap / A.k[1];
      coff[nt] = ((j0 * h1 + j1) * h2 + j2) * h3 + j3 + ci * A.S;
    }
  }
  acc_t acc[MTW][NTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = acc_t{T(0), T(0), T(0), T(0)};

  // K entry ku of a box -> box coordinates (z0, z1, z2, z3); compact: entry = (row, pair), the pair's active site
  auto decode = [&](int ku, const int *o, int &z0, int &z1, int &z2, int &z3) {
    const int p = ku & ((1 << lb3) - 1);
    int u = ku >> lb3;
    z2 = u & (A.box[2] - 1); u >>= A.lbox[2];
    z1 = u & (A.box[1] - 1); u >>= A.lbox[1];
    z0 = u;
    z3 = compact ? 2 * p + ((o[0] + z0 + o[1] + z1 + o[2] + z2 + A.parity) & 1) : p;
  };

  // staging of the input box: element e = threadIdx + 256 j of the (halo rows x fastest-axis positions) plane of a channel
  // -- every lane loads (a row of the halo has 6 .. 34 positions: a wave per row left most lanes idle), and a thread's
  // loads of one channel are issued together, then stored (one load per round trip was the kernel's time).  The split of
  // e into (row, position) does not depend on the item.
  constexpr int ES = kSitesStage;                      // >= ceil(halo plane / 256): checked by the launcher
  const int plane = R * h3;
  int er[ES], ez[ES];
#pragma unroll
  for (int j = 0; j < ES; ++j) {
    const int e = int(threadIdx.x) + kBlock * j;
    er[j] = e < plane ? e / h3 : -1;
    ez[j] = e < plane ? e - er[j] * h3 : 0;
  }

  for (int64_t item = blockIdx.x; item < A.nitems; item += gridDim.x) {
    const int b = int(item / A.nboxes);
    int bid = int(item - int64_t(b) * A.nboxes);
    int o[4];
#pragma unroll
    for (int mu = 3; mu >= 0; --mu) {
      o[mu] = (bid % A.nbox[mu]) * A.box[mu];
      bid /= A.nbox[mu];
    }
    const T *__restrict__ in_b = static_cast<const T *>(A.in) + int64_t(b) * A.cin * A.V;
    const T *__restrict__ gz_b = static_cast<const T *>(A.gz) + int64_t(b) * A.cout * gzV;
    __syncthreads();                                  // previous item fully consumed
    for (int t = threadIdx.x; t < R; t += kBlock) {
      const int z0 = t / (h1 * h2), rem = t - z0 * (h1 * h2);
      const int z1 = rem / h2, z2 = rem - z1 * h2;
      const int x0 = wrap(o[0] + z0 - r0, A.L[0]), x1 = wrap(o[1] + z1 - r1, A.L[1]),
                x2 = wrap(o[2] + z2 - r2, A.L[2]);
      rowsrc[t] = ((x0 * A.L[1] + x1) * A.L[2] + x2) * A.L[3];
    }
    // the cotangent's K entries, eight loads in flight per thread
    for (int base = threadIdx.x; base < ((MTW * 16) << lKU); base += 8 * kBlock) {
      T v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = base + j * kBlock;
        const int co = idx >> lKU, ku = idx & (KU - 1);
        int z0, z1, z2, z3;
        decode(ku, o, z0, z1, z2, z3);
        const int x0 = o[0] + z0, x1 = o[1] + z1, x2 = o[2] + z2, x3 = o[3] + z3;
        v[j] = T(0);
        if (idx < ((MTW * 16) << lKU) && co < A.cout && x0 < A.L[0] && x1 < A.L[1] && x2 < A.L[2] && x3 < A.L[3]) {
          const int64_t flat = ((int64_t(x0) * A.L[1] + x1) * A.L[2] + x2) * A.L[3] + x3;
          v[j] = gz_b[int64_t(co) * gzV + (compact ? flat >> 1 : flat)];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = base + j * kBlock;
        if (idx < ((MTW * 16) << lKU)) gzt[(idx >> lKU) * A.PS + (idx & (KU - 1))] = v[j];
      }
    }
    __syncthreads();                                  // row table visible
    int esrc[ES];
#pragma unroll
    for (int j = 0; j < ES; ++j)
      esrc[j] = er[j] >= 0 ? rowsrc[er[j]] + wrap(o[3] + ez[j] - r3, A.L[3]) : -1;
    for (int c = 0; c < A.cin; ++c) {
      T v[ES];
#pragma unroll
      for (int j = 0; j < ES; ++j) v[j] = esrc[j] >= 0 ? in_b[int64_t(c) * A.V + esrc[j]] : T(0);
#pragma unroll
      for (int j = 0; j < ES; ++j)
        if (esrc[j] >= 0) tile[c * A.S + int(threadIdx.x) + kBlock * j] = v[j];
    }
    __syncthreads();
    // K loop: the waves take turns over the box's K entries, 4 per MFMA
    for (int s = 4 * wave; s < KU; s += 4 * nwaves) {
      const int ku = s + g;
      int z0, z1, z2, z3;
      decode(ku, o, z0, z1, z2, z3);
      const int ab = ((z0 * h1 + z1) * h2 + z2) * h3 + z3;
      T a[MTW], bq[NTW];
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) a[mt] = gzt[((mt << 4) + (lane & 15)) * A.PS + ku];
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) bq[nt] = one[nt] ? T(1) : tile[ab + coff[nt]];
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = Mma<T>::mma(a[mt], bq[nt], acc[mt][nt]);
    }
  }
  // waves 1..3 hand their sums to wave 0 through LDS, one accumulator tile at a time, added in wave order; wave 0 stores
  // the workgroup's partial matrix: D col = lane & 15 (n), rows (co) 4g + r (f32) / g + 4r (f64)
  T *xch = reinterpret_cast<T *>(smem_conv);             // [wave 1..3][lane][4]
  T *part = static_cast<T *>(A.part) + int64_t(blockIdx.x) * (MTW * 16) * A.ncols_pad;
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      __syncthreads();
      if (wave > 0)
#pragma unroll
        for (int r = 0; r < 4; ++r) xch[((wave - 1) * 64 + lane) * 4 + r] = acc[mt][nt][r];
      __syncthreads();
      const int n = (nt << 4) + (lane & 15);
      if (wave == 0 && n < A.ncols_pad) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          T v = acc[mt][nt][r];
          for (int w = 0; w < nwaves - 1; ++w) v += xch[(w * 64 + lane) * 4 + r];
          const int co = (mt << 4) + (Mma<T>::kStridedRows ? g + (r << 2) : (g << 2) + r);
          part[int64_t(co) * A.ncols_pad + n] = v;
        }
      }
    }
}

// gw[i] += part[0][i] + part[1][i] + ... (i over the MTW * 16 x ncols_pad matrix), always in the same order: thread (i, j) of a
// workgroup adds the slots of chunk j one after the other (eight loads in flight), the chunks' sums are added in chunk order.
constexpr int kRedChunks = 16;
template <typename T>
__global__ __launch_bounds__(256) void wgrad_sites_reduce_kernel(const T *__restrict__ part, T *__restrict__ gw, int nslots, int64_t n) {
  __shared__ T sums[256];
  const int j = threadIdx.x >> 4, il = threadIdx.x & 15;
  const int64_t i = int64_t(blockIdx.x) * 16 + il;
  const int per = (nslots + kRedChunks - 1) / kRedChunks;
  const int s0 = j * per, s1 = s0 + per < nslots ? s0 + per : nslots;
  T v = T(0);
  if (i < n) {
    for (int sl = s0; sl < s1; sl += 8) {
      T t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t[q] = sl + q < s1 ? part[int64_t(sl + q) * n + i] : T(0);
#pragma unroll
      for (int q = 0; q < 8; ++q) v += t[q];
    }
  }
  sums[threadIdx.x] = v;
  __syncthreads();
  if (j == 0 && i < n) {
    T tot = sums[il];
    for (int c = 1; c < kRedChunks; ++c) tot += sums[c * 16 + il];
    gw[i] += tot;
  }
}

template <typename T, int MTW>
static int launch_wgrad(const WgArgs &A0, int ntiles, size_t lds, int grid, hipStream_t stream) {
  WgArgs A = A0;
  int nt0 = 0;
  while (nt0 < ntiles) {
    const int left = ntiles - nt0;
    A.nt0 = nt0;
    int per_wave;
#define NF_WG(NTW)                                                                                        \
  {                                                                                                       \
    if (lds > 64 * 1024)                                                                                  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wgrad_kernel<T, MTW, NTW>),          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));                    \
    hipLaunchKernelGGL((conv_wgrad_kernel<T, MTW, NTW>), dim3(grid), dim3(kBlock), lds, stream, A);       \
    per_wave = NTW;                                                                                       \
  }
    if (left <= 8) NF_WG(2) else if (left <= 24) NF_WG(6) else NF_WG(11)
#undef NF_WG
    const int rc = check_launch("conv wgrad kernel");
    if (rc) return rc;
    nt0 += 4 * per_wave;
  }
  return NF_OK;
}

// box geometry, LDS layout and column count of a weight-gradient launch (both kernels)
template <typename T>
static int wgrad_setup(WgArgs &A, const void *in, const void *gz, void *gw, int64_t B, const int32_t *lattice, const int32_t *ksize,
                       int cin, int cout, int *ntiles_out, size_t *lds_out) {
  NF_REQUIRE(in && gz && gw && lattice && ksize, "nf_conv_wgrad: NULL pointer");
  NF_REQUIRE(B >= 0 && cin >= 1 && cout >= 1 && cout <= 48, "nf_conv_wgrad: bad sizes (cout <= 48 per call)");
  int64_t V = 1;
  int ntaps = 1;
  for (int mu = 0; mu < 4; ++mu) {
    NF_REQUIRE(lattice[mu] >= 1 && ksize[mu] >= 1 && (ksize[mu] & 1), "nf_conv_wgrad: lattice dims >= 1 and odd kernel sizes needed");
    A.L[mu] = lattice[mu];
    A.k[mu] = ksize[mu];
    V *= lattice[mu];
    ntaps *= ksize[mu];
  }
  NF_REQUIRE(V < (int64_t(1) << 31), "nf_conv_wgrad: lattice volume must be < 2^31");
  const int MTW = (cout + 15) >> 4;
  // box of 256 (fp32) / 128 (fp64) sites, same shape rule as the forward kernel
  const int target = sizeof(T) == 4 ? 256 : 128;
  int box[4], cap[4];
  for (int mu = 0; mu < 4; ++mu) { cap[mu] = 1 << ilog2(A.L[mu]); box[mu] = 1; }
  box[3] = cap[3] < 32 ? cap[3] : 32;
  if (box[3] < 4) box[3] = 4;
  int vol = box[3];
  while (vol < target) {
    int best = -1;
    for (int mu = 2; mu >= 0; --mu)
      if (box[mu] < cap[mu] && (best < 0 || box[mu] < box[best])) best = mu;
    if (best < 0) { if (box[3] < cap[3]) best = 3; else break; }
    box[best] *= 2;
    vol *= 2;
  }
  while (vol < target) { box[3] *= 2; vol *= 2; }
  int64_t halvol = 1, nboxes = 1, rows = 1;
  for (int mu = 0; mu < 4; ++mu) {
    A.box[mu] = box[mu];
    A.lbox[mu] = ilog2(box[mu]);
    A.nbox[mu] = (A.L[mu] + box[mu] - 1) / box[mu];
    A.hal[mu] = box[mu] + A.k[mu] - 1;
    halvol *= A.hal[mu];
    nboxes *= A.nbox[mu];
    if (mu < 3) rows *= A.hal[mu];
  }
  A.units = vol;
  A.S = int(halvol) | 1;
  A.PS = vol + 2;                                    // gz planes: banks 2*co + k-group, conflict-free
  A.cin = cin; A.cout = cout; A.ntot = ntaps * cin;
  const int ntiles = (A.ntot + 1 + 15) >> 4;
  A.ncols_pad = ntiles << 4;
  A.nboxes = int(nboxes);
  A.nitems = int64_t(B) * nboxes;
  A.V = V; A.in = in; A.gz = gz; A.gw = gw;
  A.parity = -1; A.part = nullptr;
  const size_t lds = (size_t(cin) * A.S + size_t(MTW) * 16 * A.PS) * sizeof(T) + size_t(rows) * 2 * sizeof(int);
  NF_REQUIRE(lds <= 160 * 1024, "nf_conv_wgrad: needs %zu B of LDS (> 160 KiB): cin=%d", lds, cin);
  *ntiles_out = ntiles;
  *lds_out = lds;
  return NF_OK;
}

template <typename T>
static int run_wgrad(const void *in, const void *gz, void *gw, int64_t B, const int32_t *lattice,
                     const int32_t *ksize, int cin, int cout, hipStream_t stream) {
  WgArgs A{};
  int ntiles = 0;
  size_t lds = 0;
  const int rc = wgrad_setup<T>(A, in, gz, gw, B, lattice, ksize, cin, cout, &ntiles, &lds);
  if (rc) return rc;
  if (B == 0 || A.V == 0) return NF_OK;
  const int MTW = (cout + 15) >> 4;
  const int grid = int(A.nitems < 512 ? A.nitems : 512);
  if (MTW == 1) return launch_wgrad<T, 1>(A, ntiles, lds, grid, stream);
  if (MTW == 2) return launch_wgrad<T, 2>(A, ntiles, lds, grid, stream);
  return launch_wgrad<T, 3>(A, ntiles, lds, grid, stream);
}

// the K-split kernel: layers of at most kSitesMaxTiles column tiles
constexpr int kSitesMaxTiles = 14;
constexpr int kSitesMaxGrid = 512;

static int wgrad_sites_tiles(const int32_t *ksize, int cin) {
  int64_t ntaps = 1;
  for (int mu = 0; mu < 4; ++mu) {
    if (ksize[mu] < 1 || !(ksize[mu] & 1)) return 0;
    ntaps *= ksize[mu];
    if (ntaps > 4096) return 0;
  }
  return int((ntaps * cin + 1 + 15) >> 4);
}

template <typename T, int MTW>
static int launch_wgrad_sites(const WgArgs &A, int ntiles, size_t lds, int grid, hipStream_t stream) {
#define NF_WGS(NTW)                                                                                        \
  {                                                                                                        \
    if (lds > 64 * 1024)                                                                                   \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wgrad_sites_kernel<T, MTW, NTW>),     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));                     \
    hipLaunchKernelGGL((conv_wgrad_sites_kernel<T, MTW, NTW>), dim3(grid), dim3(kBlock), lds, stream, A);  \
  }
  if (ntiles <= 2) NF_WGS(2) else if (ntiles <= 6) NF_WGS(6) else NF_WGS(14)
#undef NF_WGS
  return check_launch("conv wgrad (sites) kernel");
}

template <typename T>
static int run_wgrad_sites(const void *in, const void *gz, void *gw, int64_t B, const int32_t *lattice, const int32_t *ksize,
                           int cin, int cout, int compact_parity, void *workspace, size_t workspace_bytes, hipStream_t stream) {
  WgArgs A{};
  int ntiles = 0;
  size_t lds = 0;
  const int rc = wgrad_setup<T>(A, in, gz, gw, B, lattice, ksize, cin, cout, &ntiles, &lds);
  if (rc) return rc;
  NF_REQUIRE(ntiles <= kSitesMaxTiles, "nf_conv_wgrad_sites: %d column tiles (taps x cin + 1 must be <= %d)", ntiles, 16 * kSitesMaxTiles);
  NF_REQUIRE(compact_parity < 0 || (lattice[3] % 2 == 0 && compact_parity <= 1), "nf_conv_wgrad_sites: a pair-compact cotangent needs an even fastest axis and parity 0 / 1");
  if (B == 0 || A.V == 0) return NF_OK;
  const int MTW = (cout + 15) >> 4;
  const int grid = int(A.nitems < kSitesMaxGrid ? A.nitems : kSitesMaxGrid);
  const int64_t n = int64_t(MTW) * 16 * A.ncols_pad;
  NF_REQUIRE(workspace && workspace_bytes >= size_t(grid) * size_t(n) * sizeof(T), "nf_conv_wgrad_sites: workspace too small (nf_conv_wgrad_sites_workspace)");
  A.parity = compact_parity;
  A.part = workspace;
  if (compact_parity >= 0) {                                  // half the K entries: a smaller cotangent tile (two workgroups per CU)
    const size_t old_gz = size_t(MTW) * 16 * A.PS * sizeof(T);
    A.PS = (A.units >> 1) + 2;
    lds -= old_gz - size_t(MTW) * 16 * A.PS * sizeof(T);
  }
  if (lds < size_t(3) * 64 * 4 * sizeof(T)) lds = size_t(3) * 64 * 4 * sizeof(T);      // the waves' exchange at the end
  int rc2;
  if (MTW == 1) rc2 = launch_wgrad_sites<T, 1>(A, ntiles, lds, grid, stream);
  else if (MTW == 2) rc2 = launch_wgrad_sites<T, 2>(A, ntiles, lds, grid, stream);
  else rc2 = launch_wgrad_sites<T, 3>(A, ntiles, lds, grid, stream);
  if (rc2) return rc2;
  hipLaunchKernelGGL((wgrad_sites_reduce_kernel<T>), dim3(unsigned((n + 15) / 16)), dim3(256), 0, stream,
                     static_cast<const T *>(workspace), static_cast<T *>(gw), grid, n);
  return check_launch("conv wgrad (sites) reduce kernel");
}

// d(activation)/d(pre-activation) expressed through the activation's OUTPUT y
template <typename T> __global__ __launch_bounds__(kBlock) void act_vjp_kernel(const T *__restrict__ gout,
                                                                                 const T *__restrict__ y,
                                                                                 T *__restrict__ gz, int64_t n, int act) {
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kBlock) {
    const T v = y[i];
    T d = T(1);
    switch (act) {
      case kActTanh: d = T(1) - v * v; break;
      case kActRelu: d = v > T(0) ? T(1) : T(0); break;
      case kActLeakyRelu: d = v > T(0) ? T(1) : T(0.01); break;
      case kActSoftplus: d = T(1) - (sizeof(T) == 4 ? T(expf(-float(v))) : T(exp(-double(v)))); break;
      case kActSigmoid: d = v * (T(1) - v); break;
      default: break;
    }
    gz[i] = gout[i] * d;
  }
}

}  // namespace nf

extern "C" int nf_conv_wgrad_cols(int cin, int ntaps) { return (((cin * ntaps + 1) + 15) >> 4) << 4; }

extern "C" int nf_conv_wgrad(const void *in, const void *gz, void *gw, int64_t B, const int32_t *lattice,
                             const int32_t *ksize, int cin, int cout, int dtype, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == NF_F32) return run_wgrad<float>(in, gz, gw, B, lattice, ksize, cin, cout, s);
  if (dtype == NF_F64) return run_wgrad<double>(in, gz, gw, B, lattice, ksize, cin, cout, s);
  set_error("nf_conv_wgrad: unsupported dtype %d", dtype);
  return NF_EINVAL;
}

// 1 if nf_conv_wgrad_sites takes a layer with this kernel and input-channel count (taps x cin + 1 columns in <= 14 tiles:
// every 1-, 2- and 3-D 3-tap layer of up to 8 input channels)
extern "C" int nf_conv_wgrad_sites_supported(const int32_t *lattice, const int32_t *ksize, int cin, int cout, int dtype) {
  if (!lattice || !ksize || cin < 1 || cout < 1 || cout > 48 || (dtype != NF_F32 && dtype != NF_F64)) return 0;
  const int nt = wgrad_sites_tiles(ksize, cin);
  if (nt < 1 || nt > kSitesMaxTiles) return 0;
  if (dtype == NF_F64 && cout > 32 && nt > 6) return 0;      // (3 x 14 fp64 accumulator tiles do not fit the register file)
  int64_t V = 1;
  for (int mu = 0; mu < 4; ++mu) {
    if (lattice[mu] < 1) return 0;
    V *= lattice[mu];
  }
  if (V >= (int64_t(1) << 31)) return 0;
  WgArgs A{};
  int ntiles = 0;
  size_t lds = 0;
  const int dummy = 0;
  const int rc = dtype == NF_F32 ? wgrad_setup<float>(A, &dummy, &dummy, const_cast<int *>(&dummy), 1, lattice, ksize, cin, cout, &ntiles, &lds)
                                 : wgrad_setup<double>(A, &dummy, &dummy, const_cast<int *>(&dummy), 1, lattice, ksize, cin, cout, &ntiles, &lds);
  if (rc) return 0;
  return int64_t(A.hal[0]) * A.hal[1] * A.hal[2] * A.hal[3] <= int64_t(kSitesStage) * kBlock;
}

extern "C" size_t nf_conv_wgrad_sites_workspace(const int32_t *ksize, int cin, int cout, int dtype) {
  if (!ksize || cin < 1 || cout < 1 || cout > 48 || wgrad_sites_tiles(ksize, cin) > kSitesMaxTiles) return 0;
  return size_t(kSitesMaxGrid) * size_t((cout + 15) / 16 * 16) * size_t(wgrad_sites_tiles(ksize, cin) * 16) * (dtype == NF_F64 ? 8 : 4);
}

extern "C" int nf_conv_wgrad_sites(const void *in, const void *gz, void *gw, int64_t B, const int32_t *lattice,
                                   const int32_t *ksize, int cin, int cout, int compact_parity, void *workspace,
                                   size_t workspace_bytes, int dtype, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  NF_REQUIRE(lattice && ksize && nf_conv_wgrad_sites_supported(lattice, ksize, cin, cout, dtype), "nf_conv_wgrad_sites: layer not supported (nf_conv_wgrad_sites_supported)");
  if (dtype == NF_F32) return run_wgrad_sites<float>(in, gz, gw, B, lattice, ksize, cin, cout, compact_parity, workspace, workspace_bytes, s);
  if (dtype == NF_F64) return run_wgrad_sites<double>(in, gz, gw, B, lattice, ksize, cin, cout, compact_parity, workspace, workspace_bytes, s);
  set_error("nf_conv_wgrad_sites: unsupported dtype %d", dtype);
  return NF_EINVAL;
}

// out[i] = index[i] < nsrc ? src[index[i]] : 0 -- a weight tensor re-arranged into a kernel's fragment layout (two-site
// expansion, fragment order, zero padding) in one launch; the index map is built once per layer shape on the host side
namespace nf {
template <typename T>
__global__ __launch_bounds__(256) void gather_pad_kernel(const T *__restrict__ src, const int32_t *__restrict__ index, T *__restrict__ out,
                                                         int64_t n, int64_t nsrc) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t k = index[i];
  out[i] = k >= 0 && k < nsrc ? src[k] : T(0);
}
}  // namespace nf

extern "C" int nf_gather_pad(const void *src, const int32_t *index, void *out, int64_t n, int64_t nsrc, int elem_bytes, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  NF_REQUIRE(src && index && out && n >= 0 && nsrc >= 0, "nf_gather_pad: bad arguments");
  NF_REQUIRE(elem_bytes == 2 || elem_bytes == 4 || elem_bytes == 8, "nf_gather_pad: elements of 2, 4 or 8 bytes");
  if (n == 0) return NF_OK;
  const dim3 grid(unsigned((n + 255) / 256));
  if (elem_bytes == 4)
    hipLaunchKernelGGL((gather_pad_kernel<uint32_t>), grid, dim3(256), 0, s, static_cast<const uint32_t *>(src), index, static_cast<uint32_t *>(out), n, nsrc);
  else if (elem_bytes == 8)
    hipLaunchKernelGGL((gather_pad_kernel<uint64_t>), grid, dim3(256), 0, s, static_cast<const uint64_t *>(src), index, static_cast<uint64_t *>(out), n, nsrc);
  else
    hipLaunchKernelGGL((gather_pad_kernel<uint16_t>), grid, dim3(256), 0, s, static_cast<const uint16_t *>(src), index, static_cast<uint16_t *>(out), n, nsrc);
  return check_launch("gather kernel");
}

extern "C" int nf_act_vjp(const void *grad_out, const void *y, void *grad_pre, int64_t n, int act, int dtype,
                          void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  NF_REQUIRE(grad_out && y && grad_pre && n >= 0, "nf_act_vjp: bad arguments");
  NF_REQUIRE(act >= kActNone && act <= kActSigmoid && act != kActAbs, "nf_act_vjp: activation %d has no output-only derivative", act);
  if (n == 0) return NF_OK;
  const int64_t want = (n + kBlock - 1) / kBlock;
  const unsigned grid = unsigned(want < 8192 ? want : 8192);
  if (dtype == NF_F32)
    hipLaunchKernelGGL((act_vjp_kernel<float>), dim3(grid), dim3(kBlock), 0, s, static_cast<const float *>(grad_out),
                       static_cast<const float *>(y), static_cast<float *>(grad_pre), n, act);
  else if (dtype == NF_F64)
    hipLaunchKernelGGL((act_vjp_kernel<double>), dim3(grid), dim3(kBlock), 0, s, static_cast<const double *>(grad_out),
                       static_cast<const double *>(y), static_cast<double *>(grad_pre), n, act);
  else {
    set_error("nf_act_vjp: unsupported dtype %d", dtype);
    return NF_EINVAL;
  }
  return check_launch("act vjp kernel");
}
