// nf_conv_core.h -- device code shared by the convolution kernels (nf_conv.hip: one box per
// workgroup; nf_conv_pipe.hip: persistent workgroups with staging overlapped with the MFMAs).
#pragma once
#include <hip/hip_fp16.h>
#include "nf_rqs_core.h"

namespace nf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// The 16x16x4 matrix-core instruction per element type.  A: lane l holds A[row l&15][k l>>4];
// B: B[k l>>4][col l&15]; C/D: col = l&15 and, for f32, rows 4(l>>4)+r, for f64 rows (l>>4)+4r
// (cdna_hip_programming.md section 3: "f64 MFMA does NOT use these maps").
template <typename T> struct Mma;
template <> struct Mma<float> {
  typedef f32x4 vec4;
  static constexpr bool kStridedRows = false;
  static __device__ __forceinline__ vec4 mma(float a, float b, vec4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<double> {
  typedef f64x4 vec4;
  static constexpr bool kStridedRows = true;
  static __device__ __forceinline__ vec4 mma(double a, double b, vec4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
};

enum { kActNone = 0, kActTanh = 1, kActRelu = 2, kActLeakyRelu = 3, kActSoftplus = 4, kActAbs = 5,
       kActSigmoid = 6 };

// Timing ablations (kernels that SKIP work) and clock stamps exist only in diagnostic builds (`make DIAG=1` defines
// NF_DIAG); in the product library the tests below are compile-time false and the environment is never consulted.
#ifdef NF_DIAG
#define NF_DBG(A, bits) ((((A).dbg) & (bits)) != 0)
#define NF_STAMPS(A) ((A).stamps != nullptr)
#define NF_DIAG_ENV_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define NF_DBG(A, bits) false
#define NF_STAMPS(A) false
#define NF_DIAG_ENV_INT(name, dflt) (dflt)
#endif

struct ConvArgs {
  const void *in;       // (B, Cin, V) of T
  const void *wfrag;    // [tap][kq][ntile][64] fragment-ordered, zero padded, of T
  const void *bias;     // (Cout) of T, or null
  void *out;            // (B, Cout, V) or (B, Cout, V/2) of T
  int64_t V;
  int L[4], k[4], box[4], lbox[4], nbox[4], hal[4];
  int S;                // LDS plane stride (dwords)
  int cin, cin_pad, cout, kq, nt_total, nt0;
  int kt3;              // taps walked along the fastest axis: k[3], or k[3]+1 in two-site mode
  int sh2;              // two-site column packing (cout <= 8): cols 0-7 -> site 2p, cols 8-15 -> site 2p+1
  int cchunk, kq_total;  // channels staged per pass of the K loop (multiple of 4); cin_pad / 4
  int act, compact, parity;
  // fused coupling epilogue (nf_conv_rqs): the logits never leave the CU
  const float *xact;    // (B, V) field, active sites are transformed
  float *yout;          // (B, V) out: value at active sites, 0 at frozen sites
  double *partial;      // (B, gridDim.x) per-workgroup log-det partials
  RqsParams P;
  int packed, ns;       // packed: K = (tap, ci) flattened, 4 per step, ns steps (multiple of 4)
  unsigned long long *stamps;   // diagnostic build only (NF_CONV_STAMPS): 8 clock stamps per workgroup
  int dbg;              // DIAGNOSTIC builds only (make DIAG=1; NF_CONV_DBG / NF_CONVG_DBG timing ablations that skip work): always 0 otherwise
  int field16;          // fused coupling epilogue: xact / yout are IEEE half (NF_CONV_FIELD_F16: fp16 field storage, fp32 arithmetic)
  int in_split16;       // nf_conv_rqs: `in` is the (B, V, 16) fp16 (hi, lo) pair tensor a previous layer wrote (NF_CONV_SPLIT16_INPUT)
  int64_t nitems;       // nf_conv_pipe.hip: (sample, box) items in the launch, boxes per sample
  int nboxes;
  int out_split16;          // two-site layers with 8 output channels: store (hi, lo) fp16 pairs, channel-last, 32 bytes per site
  int wide_no, wide_llpr;   // nf_conv_pipe.hip wide staging: row blocks per wave (0 = narrow), log2(lanes per row)
  const unsigned *gscale_bits;   // nf_conv_dgrad_split16: bits of max |cotangent| the input pair tensor was scaled by (or null)
  int accumulate;                // ... add to the output planes instead of overwriting them
  const float *gyout;            // nf_conv_rqs_vjp: cotangent of the coupling's value, (B, V) like xact
  const float *glogj;            // ... and of its log-det, (B)
  int seg_lo, seg_n;             // nf_conv_g.hip: the launch covers segments seg_lo .. seg_lo + seg_n - 1 of every row (0, 0 = all)
};

// The power of two that brings a tensor whose largest magnitude has the bits *absmax to [2^12, 2^13): cotangents of a mean
// over a batch are far below fp16's normal range (nf_conv_w.hip, nf_conv_dgrad_split16).
__device__ __forceinline__ float pow2_scale_for(const unsigned *absmax) {
  if (!absmax) return 1.f;
  const float mx = __uint_as_float(*absmax);
  if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.f;
  int e;
  (void)frexpf(mx, &e);                       // mx = f 2^e, f in [0.5, 1)
  return ldexpf(1.f, 13 - e);
}

// The fp16 (hi, lo) pair tensor exchanged by the split-fp16 kernels (include/normflow_hip.h, NF_OUT_SPLIT16) is row-major:
// a lattice row (L3 sites of the fastest axis) is L3*32 bytes = [hi | lo][even sites | odd sites][L3/2 slots][8 channels]
// halfs; the even block holds site 2s in slot s, the odd block site 2s-1 (mod L3) in slot s (the one-slot rotation keeps
// the four lane groups of a ds_read_b128 on different LDS banks when a row image is read tap by tap).  Byte offset of a
// site's hi entry inside its row; its lo entry is L3*16 bytes further.
__host__ __device__ __forceinline__ int pair_row_offset(int x3, int L3) {
  const int par = x3 & 1;
  int slot = (x3 + par) >> 1;
  slot = slot >= (L3 >> 1) ? slot - (L3 >> 1) : slot;
  return par * (L3 * 8) + slot * 16;
}

// The field of a fused coupling epilogue: a pair (sites 2h, 2h+1) as fp32 or -- NF_CONV_FIELD_F16 -- IEEE half storage
// (BASELINE config 5: fp16 fields / fp32 log-det; the arithmetic is fp32 either way).
#include <hip/hip_fp16.h>
__device__ __forceinline__ float2 load_field_pair(const ConvArgs &A, int64_t pair) {
  if (A.field16) return __half22float2(reinterpret_cast<const __half2 *>(A.xact)[pair]);
  return reinterpret_cast<const float2 *>(A.xact)[pair];
}
__device__ __forceinline__ void store_field_pair(const ConvArgs &A, int64_t pair, float2 v) {
  if (A.field16) reinterpret_cast<__half2 *>(A.yout)[pair] = __float22half2_rn(v);
  else reinterpret_cast<float2 *>(A.yout)[pair] = v;
}

// One LDS-DMA piece: 64 lanes x 16 bytes from sbase + voff (per lane) to the 1 KiB at LDS byte address `lds` (wave-uniform).
// M0 is written in the same statement that reads it and is not restored: nothing else in this kernel uses it (checked in the
// ISA), and a save / restore pair per piece bought nothing.
__device__ __forceinline__ void dma_row(const void *sbase, unsigned voff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// tanh on the hardware exp/rcp: (1 - t) / (1 + t), t = exp(-2|v|); absolute error ~1e-7 (the fp32 rounding of an
// O(1) activation), ~8 instructions where ocml's tanhf takes ~40 -- it was most of the 8->8 layer's epilogue.
__device__ __forceinline__ float fast_tanh(float v) {
  const float t = __expf(-2.f * fabsf(v));
  return copysignf((1.f - t) * __frcp_rn(1.f + t), v);
}

__device__ __forceinline__ float activate(float v, int act) {
  switch (act) {
    case kActTanh: return fast_tanh(v);
    case kActRelu: return v > 0.f ? v : 0.f;
    case kActLeakyRelu: return v > 0.f ? v : 0.01f * v;
    case kActSoftplus: return v > 20.f ? v : log1pf(expf(v));
    case kActAbs: return fabsf(v);
    case kActSigmoid: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}
__device__ __forceinline__ double activate(double v, int act) {
  switch (act) {
    case kActTanh: return tanh(v);
    case kActRelu: return v > 0. ? v : 0.;
    case kActLeakyRelu: return v > 0. ? v : 0.01 * v;
    case kActSoftplus: return v > 20. ? v : log1p(exp(v));
    case kActAbs: return fabs(v);
    case kActSigmoid: return 1. / (1. + exp(-v));
    default: return v;
  }
}

__device__ __forceinline__ int wrap(int v, int L) {
  v %= L;
  return v < 0 ? v + L : v;
}

// Tap counters (uniform): row-major walk over the kernel window; `off` is the LDS offset of
// the tap relative to a unit's own position in the staged tile.
struct TapWalk {
  int j1, j2, j3, off, tap;
  __device__ __forceinline__ void next(const ConvArgs &A, int ntaps) {
    if (tap + 1 >= ntaps) return;               // clamp at the last tap (harmless re-read)
    ++tap;
    const int h3 = A.hal[3], h2 = A.hal[2], h1 = A.hal[1];
    ++off;
    if (++j3 == A.kt3) {
      j3 = 0;
      off += h3 - A.kt3;
      if (++j2 == A.k[2]) {
        j2 = 0;
        off += (h2 - A.k[2]) * h3;
        if (++j1 == A.k[1]) {
          j1 = 0;
          off += (h1 - A.k[1]) * h2 * h3;
        }
      }
    }
  }
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter
// (s_waitcnt vmcnt(0)): every output store and every staging load still in flight would have to complete at each
// barrier, which serialises the persistent kernels on memory latency (K5c: same time with and without a two-item
// prefetch lag until this replaced the barriers).  Global memory written here is never read back by the workgroup.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void stamp([[maybe_unused]] const ConvArgs &A, [[maybe_unused]] int slot) {
#ifdef NF_DIAG
  if (NF_STAMPS(A) && threadIdx.x == 0) {
    const unsigned id = blockIdx.y * gridDim.x + blockIdx.x;
    if (id < 4096u) A.stamps[id * 8 + slot] = __builtin_readcyclecounter();
  }
#endif
}

// Epilogue of one box: accumulators -> (bias, activation) -> output planes, or, FUSE > 0, the
// coupling map on the logits (see nf_conv.hip).  `tile` is LDS scratch (>= 48*(UNITS+4) floats when
// FUSE), dead as input at this point; `pidx` is the slot of this box's log-det partial.
template <typename T, int MT, int NT, bool COMPACT, int FUSE>
__device__ __forceinline__ void conv_epilogue(const ConvArgs &A, const int (&o)[4], int b, int64_t pidx,
                                              typename Mma<T>::vec4 (&acc)[MT][NT], T *tile, double *red,
                                              int wave, int lane) {
  typedef typename Mma<T>::vec4 acc_t;
  const int g = lane >> 4;
  const bool sh2 = !COMPACT && A.sh2;
  const int lb3 = (COMPACT || sh2) ? A.lbox[3] - 1 : A.lbox[3];
  // ---- fused coupling epilogue (FUSE = 1 forward, 2 inverse): accumulators (+bias) -> LDS as
  // [channel][unit] -> one lane per ACTIVE site runs the RQ-spline map on its 3M-2 logits ->
  // y pair store + per-workgroup log-det partial.  The (B, C, V/2) logit tensor is never
  // written to or read from HBM.
  if constexpr (FUSE > 0) {
    constexpr int M = NT == 1 ? 4 : (NT == 2 ? 8 : 16);
    constexpr int C = 3 * M - 2;
    constexpr int UNITS = (kBlock / kWave) * MT * 16;
    constexpr int PU = UNITS + 4;                       // row pitch: 16-B aligned rows
    lds_barrier();                                    // the input tile is dead from here on
    float *pt = reinterpret_cast<float *>(tile);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co = (nt << 4) + (lane & 15);
        const float bv = (A.bias && co < A.cout) ? static_cast<const float *>(A.bias)[co] : 0.f;
        acc_t v = acc[mt][nt];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv;
        *reinterpret_cast<acc_t *>(reinterpret_cast<T *>(pt) + co * PU + ((wave * MT + mt) << 4) + (g << 2)) = v;
      }
    lds_barrier();
    double lacc = 0.0;
    if (threadIdx.x < UNITS) {
      int u = threadIdx.x;
      const int p3 = u & ((1 << lb3) - 1);
      u >>= lb3;
      const int z2 = u & (A.box[2] - 1);
      u >>= A.lbox[2];
      const int z1 = u & (A.box[1] - 1);
      u >>= A.lbox[1];
      const int z0 = u;
      const int x0 = o[0] + z0, x1 = o[1] + z1, x2 = o[2] + z2, x3p = o[3] / 2 + p3;
      if (x0 < A.L[0] && x1 < A.L[1] && x2 < A.L[2] && x3p < A.L[3] / 2) {
        const int off = (A.parity + x0 + x1 + x2) & 1;  // which site of the pair is active
        const int64_t pair = int64_t(b) * (A.V / 2) + ((int64_t(x0) * A.L[1] + x1) * A.L[2] + x2) * (A.L[3] / 2) + x3p;
        const float2 xv = load_field_pair(A, pair);
        RegCol<float, C> a;
#pragma unroll
        for (int c = 0; c < C; ++c) a[c] = pt[c * PU + threadIdx.x];
        float val, logd;
        rqs_site<float, M, FUSE == 2>(a, A.P, off ? xv.y : xv.x, val, logd);
        float2 ov;
        ov.x = off ? 0.f : val;
        ov.y = off ? val : 0.f;
        store_field_pair(A, pair, ov);
        lacc = double(logd);
      }
    }
    const double tot = block_sum(lacc, red);
    if (threadIdx.x == 0) A.partial[pidx] = tot;
    return;
  }

  // ---- epilogue: C/D layout  col = lane&15 (channel); rows (sites) 4*(lane>>4)+r for f32,
  // (lane>>4)+4r for f64
  const int64_t Vout = COMPACT ? A.V / 2 : A.V;
  T *__restrict__ out_b = static_cast<T *>(A.out) + int64_t(b) * A.cout * Vout;
  const int L3u = COMPACT ? A.L[3] / 2 : A.L[3];
  auto unit_base = [&](int u, int &x3u, bool &ok) -> int64_t {   // unit -> offset in an output plane
    const int p3 = u & ((1 << lb3) - 1);
    u >>= lb3;
    const int z2 = u & (A.box[2] - 1);
    u >>= A.lbox[2];
    const int z1 = u & (A.box[1] - 1);
    u >>= A.lbox[1];
    const int z0 = u;
    const int x0 = o[0] + z0, x1 = o[1] + z1, x2 = o[2] + z2;
    x3u = (COMPACT ? o[3] / 2 : o[3]) + p3;
    ok = x0 < A.L[0] && x1 < A.L[1] && x2 < A.L[2];
    return ((int64_t(x0) * A.L[1] + x1) * A.L[2] + x2) * L3u + x3u;
  };
  if (sh2) {
    // columns 0-7: channel co at site 2p; columns 8-15: channel co at site 2p + 1
    const int co = lane & 7, shift = (lane >> 3) & 1;
    const T bv = (A.bias && co < A.cout) ? static_cast<const T *>(A.bias)[co] : T(0);
    if constexpr (!Mma<T>::kStridedRows) {
      if ((A.L[3] & 3) == 0) {
        // Through LDS, so that the box leaves as whole 16-byte pieces of lattice rows (8 complete rows per store
        // instruction) instead of 8-byte pairs scattered over 32 rows: ot[co][row of the box][x3].
        constexpr int UNITS = (kBlock / kWave) * MT * 16;
        const int b3 = 2 << lb3, lrows_units = lb3;              // sites per box row; units per row = 1 << lb3
        const int rows = UNITS >> lrows_units;
        const int CS = rows * b3 + 8;                            // channel stride: +8 floats spreads the channels over banks
        lds_barrier();                                         // every wave is done with the input tile
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int u = ((wave * MT + mt) << 4) + (g << 2) + r;
            const int p3 = u & ((1 << lb3) - 1), zr = u >> lb3;
            tile[co * CS + zr * b3 + 2 * p3 + shift] = activate(acc[mt][0][r] + bv, A.act);
          }
        lds_barrier();
        if (A.out_split16) {
          // the consumer is the split-fp16 kernel (nf_conv_h.hip): hand it every site's 8 channels already split
          // into fp16 (hi, lo) pairs in the row-major pair layout (pair_row_offset) -- one conversion per site here
          // instead of one per halo copy there
          typedef _Float16 h8 __attribute__((ext_vector_type(8)));
          unsigned char *ob = reinterpret_cast<unsigned char *>(out_b);
          for (int t = threadIdx.x; t < rows * b3; t += kBlock) {
            int zr = t / b3;
            const int x3 = o[3] + (t - zr * b3);
            h8 hi, lo;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              const float v = tile[c * CS + t];
              const _Float16 hh = static_cast<_Float16>(v);
              hi[c] = hh;
              lo[c] = static_cast<_Float16>(v - static_cast<float>(hh));
            }
            const int z2 = zr & (A.box[2] - 1);
            zr >>= A.lbox[2];
            const int z1 = zr & (A.box[1] - 1);
            zr >>= A.lbox[1];
            const int x0 = o[0] + zr, x1 = o[1] + z1, x2 = o[2] + z2;
            if (x0 < A.L[0] && x1 < A.L[1] && x2 < A.L[2] && x3 < A.L[3]) {
              unsigned char *d = ob + ((int64_t(x0) * A.L[1] + x1) * A.L[2] + x2) * (A.L[3] * 32) + pair_row_offset(x3, A.L[3]);
              *reinterpret_cast<h8 *>(d) = hi;
              *reinterpret_cast<h8 *>(d + A.L[3] * 16) = lo;
            }
          }
          return;
        }
        const int lq = lb3 - 1;                                  // log2(16-byte pieces per row) = log2(b3 / 4)
        const int per_ch = rows << lq;
        for (int q = threadIdx.x; q < 8 * per_ch; q += kBlock) {
          const int c = q / per_ch, rem = q - c * per_ch;
          int zr = rem >> lq;
          const int c4 = rem & ((1 << lq) - 1);
          const acc_t v = *reinterpret_cast<const acc_t *>(tile + c * CS + zr * b3 + 4 * c4);
          const int z2 = zr & (A.box[2] - 1);
          zr >>= A.lbox[2];
          const int z1 = zr & (A.box[1] - 1);
          zr >>= A.lbox[1];
          const int x0 = o[0] + zr, x1 = o[1] + z1, x2 = o[2] + z2, x3 = o[3] + 4 * c4;
          if (c < A.cout && x0 < A.L[0] && x1 < A.L[1] && x2 < A.L[2] && x3 < A.L[3])      // L3 % 4 == 0: whole piece inside
            *reinterpret_cast<acc_t *>(out_b + int64_t(c) * Vout + ((int64_t(x0) * A.L[1] + x1) * A.L[2] + x2) * A.L[3] + x3) = v;
        }
        return;
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int u = ((wave * MT + mt) << 4) + (Mma<T>::kStridedRows ? g + (r << 2) : (g << 2) + r);
        const int p3 = u & ((1 << lb3) - 1);
        u >>= lb3;
        const int z2 = u & (A.box[2] - 1);
        u >>= A.lbox[2];
        const int z1 = u & (A.box[1] - 1);
        u >>= A.lbox[1];
        const int x0 = o[0] + u, x1 = o[1] + z1, x2 = o[2] + z2, x3 = o[3] + 2 * p3 + shift;
        if (co < A.cout && x0 < A.L[0] && x1 < A.L[1] && x2 < A.L[2] && x3 < A.L[3])
          out_b[int64_t(co) * Vout + ((int64_t(x0) * A.L[1] + x1) * A.L[2] + x2) * A.L[3] + x3] =
              activate(acc[mt][0][r] + bv, A.act);
      }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if constexpr (!Mma<T>::kStridedRows) {
      // the 4 units of this lane are consecutive along the fastest box axis (box3 units >= 4 is
      // guaranteed by the launcher): they share z0..z2 and form one 16-byte store
      int x3u;
      bool row_ok;
      const int64_t base = unit_base(((wave * MT + mt) << 4) + (g << 2), x3u, row_ok);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co = ((A.nt0 + nt) << 4) + (lane & 15);
        if (!row_ok || co >= A.cout) continue;
        const T bv = A.bias ? static_cast<const T *>(A.bias)[co] : T(0);
        acc_t v = acc[mt][nt];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = activate(v[r] + bv, A.act);
        T *dst = out_b + int64_t(co) * Vout + base;
        if (x3u + 3 < L3u && ((Vout | base) & 3) == 0) {
          *reinterpret_cast<acc_t *>(dst) = v;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (x3u + r < L3u) dst[r] = v[r];
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int x3u;
        bool row_ok;
        const int64_t base = unit_base(((wave * MT + mt) << 4) + g + (r << 2), x3u, row_ok);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int co = ((A.nt0 + nt) << 4) + (lane & 15);
          if (!row_ok || x3u >= L3u || co >= A.cout) continue;
          const T bv = A.bias ? static_cast<const T *>(A.bias)[co] : T(0);
          out_b[int64_t(co) * Vout + base] = activate(acc[mt][nt][r] + bv, A.act);
        }
      }
    }
  }
}

// nf_conv_pipe.hip: 1 = launched (dry: would launch), 0 = layer not eligible (use the one-box kernel), < 0 = error
int conv_h_eligible(const ConvArgs &A, int fuse, int64_t *nboxes);
int launch_conv_h(const ConvArgs &A0, int64_t B, int fuse, hipStream_t stream, bool dry);
int launch_conv_c1(const ConvArgs &A0, int MT, int64_t B, int64_t nboxes, hipStream_t stream);
int launch_conv_pipe(const ConvArgs &A0, int64_t B, int64_t nboxes, int fuse, hipStream_t stream, bool dry);

}  // namespace nf
