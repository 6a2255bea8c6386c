// nf_conv_pipe.hip -- K5p: the fp32 circular convolution of nf_conv.hip as PERSISTENT workgroups
// whose staging is overlapped with their own MFMAs.
//
// Why: in-kernel clock stamps on the one-box-per-workgroup kernel showed a
// workgroup spending ~48 % of its life staging its input box (latency-bound 136-byte row loads)
// and the other ~52 % issuing MFMAs, with too few co-resident workgroups to cover one phase with
// another's.  Here a workgroup walks a list of (sample, box) items; the LDS holds TWO input
// buffers of 4 channel planes; while the waves multiply out of one buffer they carry the loads of
// the next (item, channel chunk) along: every iteration of the MFMA loop (two kernel rows, 2*K3*MT*NT
// MFMAs) first issues a few global row loads into registers and, after its MFMAs, writes them into
// the other buffer.  The load latency is hidden behind the matrix work of the same wave, the
// row -> address arithmetic is scalar (rows are wave-uniform) and nothing is staged synchronously
// except the very first chunk of a workgroup.
//
// Same math, operand layouts, weight fragments and epilogues as nf_conv.hip (reference:
// src/nn/scalar/modules.py:120-145, src/nn/scalar/convNd.py:86-126); used by nf_conv_fwd /
// nf_conv_rqs whenever the layer is eligible (fp32, cin % 4 == 0, kernel extent 3 along the
// fastest axis, halo row <= 64 sites), otherwise the one-box kernel runs.
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include "nf_conv_core.h"

namespace nf {

constexpr int kGPI = 2;        // row groups (x 4 channels) a wave loads per MFMA-loop iteration
constexpr int kSlack = 64;     // dwords of slack at the end of every LDS plane: target of masked-off stores

__device__ __forceinline__ constexpr int ilog2_c(int v) { return v <= 1 ? 0 : 1 + ilog2_c(v >> 1); }

__device__ __forceinline__ int wrap1(int v, int L) {      // v in [-L, 2L)
  return v < 0 ? v + L : (v >= L ? v - L : v);
}

// The NV = K3*NT weight values a lane needs for one kernel row and channel quad: whole 16-byte words plus a
// remainder loaded at its own width.  (Loading the zero padding of the row record as well looks harmless, but the
// compiler treats the padding registers as dead, reuses them at once as temporaries and must then WAIT for the load
// that is still going to overwrite them: a full L2 latency exposed in every iteration.)
template <int NV> struct BFrag {
  static constexpr int N4 = NV / 4, NR = NV % 4;
  f32x4 v4[N4 > 0 ? N4 : 1];
  float r[NR > 0 ? NR : 1];
  __device__ __forceinline__ void load(const float *__restrict__ p) {
#pragma unroll
    for (int i = 0; i < N4; ++i) v4[i] = *reinterpret_cast<const f32x4 *>(p + 4 * i);
    if constexpr (NR == 1) {
      r[0] = p[4 * N4];
    } else if constexpr (NR == 2) {
      const float2 t = *reinterpret_cast<const float2 *>(p + 4 * N4);
      r[0] = t.x; r[1] = t.y;
    } else if constexpr (NR == 3) {
      const float2 t = *reinterpret_cast<const float2 *>(p + 4 * N4);
      r[0] = t.x; r[1] = t.y; r[2] = p[4 * N4 + 2];
    }
  }
  __device__ __forceinline__ float get(int idx) const { return idx < 4 * N4 ? v4[idx >> 2][idx & 3] : r[idx - 4 * N4]; }
};

template <int MT, int NT, int K3, bool COMPACT, int FUSE, bool WIDE, bool HOT>
__global__ __launch_bounds__(kBlock, 2) void conv_pipe_kernel(ConvArgs A) {
  static_assert(!HOT || WIDE, "the unrolled schedule is written for wide staging");
  typedef float T;
  typedef f32x4 acc_t;
  extern __shared__ __align__(16) unsigned char smem_pipe[];
  __shared__ double red[kBlock / kWave];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int nwaves = kBlock / kWave;
  const int g = lane >> 4;
  const int r0 = A.k[0] >> 1, r1 = A.k[1] >> 1, r2 = A.k[2] >> 1, r3 = A.k[3] >> 1;
  const int h1 = A.hal[1], h2 = A.hal[2], h3 = A.hal[3];
  const int R = A.hal[0] * h1 * h2;
  const int bufsz = 4 * A.S;
  T *buf = reinterpret_cast<T *>(smem_pipe);
  auto rowz_of = [&](int t) {                               // halo row -> z0 | z1 << 8 | z2 << 16
    const int z0 = t / (h1 * h2), rem = t - z0 * (h1 * h2);
    const int z1 = rem / h2, z2 = rem - z1 * h2;
    return z0 | (z1 << 8) | (z2 << 16);
  };

  // ---- the items of this workgroup.  Workgroups are dealt round-robin to the 8 XCDs; the remap
  // gives each XCD a contiguous run of boxes at every step, so that neighbouring boxes (which share
  // halo rows) meet in the same L2.
  const int nb = gridDim.x;                                 // multiple of 8 (launcher)
  const int vb = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);
  if (vb >= A.nitems) return;
  const int n_my = int((A.nitems - vb + nb - 1) / nb);
  const int nchunk = A.cin_pad >> 2;
  const int gpw = (R + nwaves - 1) / nwaves;               // row groups per wave and phase
  auto decode = [&](int it, int &b, int (&o)[4]) {          // nitems < 2^31 (launcher)
    b = it / A.nboxes;
    int bid = it - b * A.nboxes;
#pragma unroll
    for (int mu = 3; mu >= 0; --mu) {
      o[mu] = (bid % A.nbox[mu]) * A.box[mu];
      bid /= A.nbox[mu];
    }
  };

  // ---- staging context of the NEXT phase: the wave's share of the halo rows x 4 channel planes, cut into
  // "groups" that issue() loads into registers and commit() writes to the other LDS buffer.
  //  * WIDE (the box spans whole lattice rows, L3 in {8,16,32,64}): a group = RPI = 256/L3 rows of ONE channel; lane
  //    (rr, k) loads the 16 bytes k of row rr (one global_load_dwordx4 covers RPI complete, aligned rows) and stores
  //    them as two ds_write2_b32; the two halo sites of a row are its own end elements (periodic wrap), written by
  //    the row's edge lanes.  4 instructions move what takes 2*RPI in the narrow scheme -- the vector-memory
  //    instruction count, not the bytes, is what the MFMA loop feels (tools/mfma_probe2.hip).
  //  * narrow (any box): a group = one row x 4 channels, one dword per lane; lane l keeps the source / LDS row
  //    offsets of group l and issue() fetches them with v_readlane.
  // Row offsets in the input are recomputed only when the next phase belongs to a new box.
  const T *__restrict__ nsrc = nullptr;
  int nreal = 4;                                            // real channels in the next phase's quad (cin is zero-padded to a multiple of 4)
  T *nbuf = buf;
  constexpr int kG = WIDE ? 1 : kGPI;                       // groups per MFMA-loop iteration
  // narrow state
  int nx3 = 0, myoff = 0;
  const int myrow = wave + nwaves * lane;
  const bool myvalid = lane < gpw && myrow < R;
  const int mypz = myvalid ? rowz_of(myrow) : 0;
  const int mydst = myvalid ? myrow * h3 : -1;
  T sv[kGPI][4];
  int sdst[kGPI];
  // wide state
  const int llpr = A.wide_llpr, lpr = 1 << llpr, rpi = 64 >> llpr;
  const int wk = lane & (lpr - 1);
  int wpz[4], wdst[4], whalo[4], woff[4] = {0, 0, 0, 0};
  if constexpr (WIDE) {
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      int row = rpi * (nwaves * o + wave) + (lane >> llpr);
      if (row >= R) row = (lane >> llpr) < R ? (lane >> llpr) : 0;     // spare lanes repeat an early row (same data, same place)
      wpz[o] = rowz_of(row);
      wdst[o] = row * h3 + 1 + 4 * wk;
      whalo[o] = wk == 0 ? row * h3 + h3 - 1 : (wk == lpr - 1 ? row * h3 : A.S - kSlack + lane);
    }
  }
  f32x4 svw = {0.f, 0.f, 0.f, 0.f};
  int sc = 0, so = 0;
  auto pick = [&](const int (&arr)[4], int o) {
    int v = arr[0];
    if (o == 1) v = arr[1];
    if (o == 2) v = arr[2];
    if (o == 3) v = arr[3];
    return v;
  };
  const int ngroups = WIDE ? 4 * A.wide_no : gpw;
  auto issue = [&](int i) {
    if constexpr (WIDE) {
      so = i >> 2;
      sc = i & 3;
      svw = *reinterpret_cast<const f32x4 *>(nsrc + int64_t(sc < nreal ? sc : 0) * A.V + pick(woff, so));
    } else {
#pragma unroll
      for (int j = 0; j < kGPI; ++j) {
        const int gi = i * kGPI + j;                          // < 64 (launcher: R <= 256)
        const int off = __builtin_amdgcn_readlane(myoff, gi) + nx3;
        const int drow = __builtin_amdgcn_readlane(mydst, gi);
#pragma unroll
        for (int c = 0; c < 4; ++c) sv[j][c] = nsrc[int64_t(c < nreal ? c : 0) * A.V + off];
        sdst[j] = (drow >= 0 && lane < h3) ? drow + lane : A.S - kSlack + lane;
      }
    }
  };
  auto commit = [&]() {
    if constexpr (WIDE) {
      T *pl = nbuf + sc * A.S;
      const int d = pick(wdst, so);
      const f32x4 v = sc < nreal ? svw : f32x4{0.f, 0.f, 0.f, 0.f};     // padded channel planes are zero (selected HERE:
      pl[d] = v[0]; pl[d + 1] = v[1]; pl[d + 2] = v[2]; pl[d + 3] = v[3];  //  touching svw at issue time would wait for the load)
      pl[pick(whalo, so)] = wk == 0 ? v[0] : v[3];
    } else {
#pragma unroll
      for (int j = 0; j < kGPI; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) nbuf[c * A.S + sdst[j]] = c < nreal ? sv[j][c] : 0.f;
    }
  };
  auto set_next = [&](int b, const int (&o)[4], int q, T *dstbuf, bool new_box) {
    nsrc = static_cast<const T *>(A.in) + (int64_t(b) * A.cin + 4 * q) * A.V;
    nreal = A.cin - 4 * q < 4 ? A.cin - 4 * q : 4;
    nbuf = dstbuf;
    if (new_box) {
      if constexpr (WIDE) {
#pragma unroll
        for (int oo = 0; oo < 4; ++oo) {
          const int x0 = wrap1(o[0] + (wpz[oo] & 255) - r0, A.L[0]);
          const int x1 = wrap1(o[1] + ((wpz[oo] >> 8) & 255) - r1, A.L[1]);
          const int x2 = wrap1(o[2] + (wpz[oo] >> 16) - r2, A.L[2]);
          woff[oo] = ((x0 * A.L[1] + x1) * A.L[2] + x2) * A.L[3] + 4 * wk;
        }
      } else {
        nx3 = lane < h3 ? wrap1(o[3] + lane - r3, A.L[3]) : 0;
        const int x0 = wrap1(o[0] + (mypz & 255) - r0, A.L[0]);
        const int x1 = wrap1(o[1] + ((mypz >> 8) & 255) - r1, A.L[1]);
        const int x2 = wrap1(o[2] + (mypz >> 16) - r2, A.L[2]);
        myoff = ((x0 * A.L[1] + x1) * A.L[2] + x2) * A.L[3];  // lanes without a row: some valid offset, stored to the slack
      }
    }
  };

  // ---- per-lane A-fragment bases (as in nf_conv.hip); depend on the box only through the parity of its origin
  const bool sh2 = !COMPACT && A.sh2;
  const int lb3 = (COMPACT || sh2) ? A.lbox[3] - 1 : A.lbox[3];
  int abase[MT];
  auto set_abase = [&](const int (&o)[4]) {
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
      abase[mt] = ((z0 * h1 + z1) * h2 + z2) * h3 + z3 + g * A.S;
    }
  };

  // ---- the MFMA loop of one phase: kernel rows in pairs (nf_conv.hip, mma_rows), the first `nst`
  // iterations carrying the next phase's loads
  const int nrows = A.k[0] * A.k[1] * A.k[2];
  // weights: row-packed layout (include/normflow_hip.h, NF_WLAYOUT_ROWPACK): all K3*NT values a lane needs for
  // one kernel row and channel quad are adjacent, so a row costs NV4 16-byte loads instead of K3*NT 4-byte ones
  // (probe: tools/mfma_probe2.hip -- the per-fragment dword loads, not the MFMAs, capped the rate)
  constexpr int NV4 = (K3 * NT + 3) / 4;
  const float *__restrict__ wrow = static_cast<const float *>(A.wfrag) + lane * (4 * NV4);
  typedef BFrag<K3 * NT> bfrag_t;
  acc_t acc[MT][NT];
  auto phase_mma = [&](const T *tile, int kq0, int nst) {
    T a0[K3][MT], a1[K3][MT];
    bfrag_t b0, b1;
    int j1 = 0, j2 = 0, off = 0, row = 0;
    auto next = [&]() {
      if (row + 1 >= nrows) return;
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
    auto request = [&](T (&a)[K3][MT], bfrag_t &b) {
      b.load(wrow + (int64_t(row) * A.kq_total + kq0) * (256 * NV4));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const T *pa = tile + abase[mt] + off;
#pragma unroll
        for (int j3 = 0; j3 < K3; ++j3) a[j3][mt] = pa[j3];
      }
    };
    auto multiply = [&](const T (&a)[K3][MT], const bfrag_t &b) {
#pragma unroll
      for (int j3 = 0; j3 < K3; ++j3)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = Mma<T>::mma(a[j3][mt], b.get(j3 * NT + nt), acc[mt][nt]);
    };
    request(a0, b0);
    int r = 0, i = 0;
    for (; i < nst && r + 1 < nrows; ++i, r += 2) {      // iterations that carry staging
      next();
      request(a1, b1);
      issue(i);
      __builtin_amdgcn_sched_barrier(0);
      multiply(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      next();
      request(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      multiply(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      commit();
      __builtin_amdgcn_sched_barrier(0);
    }
    for (; r + 1 < nrows; r += 2) {
      next();
      request(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      multiply(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      next();
      request(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      multiply(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (r < nrows) multiply(a0, b0);
    for (; i < nst; ++i) {                                // groups the MFMA loop had no iterations for
      issue(i);
      commit();
    }
  };

  // ---- HOT: the 3x3x3(x3) kernel with 8 staging groups per wave and phase (the 4-d ConvAct layers on a box of
  // 2x2x2 full rows).  The 13 row pairs are unrolled, so the row walk is constant-folded and -- the point --
  // staging group g is issued in pair g and committed in pair g + kLag: the load has kLag pairs (~3.5k cycles of
  // MFMA work) to land, where the rolled loop commits in the pair it issues in and stalls on every L2/HBM miss
  // (measured: 12-18 % of the kernel).  The block's last phase re-stages its own chunk into the idle buffer so
  // that the schedule has no conditionals.
  constexpr int kLag = 3;
  auto phase_mma_hot = [&](const T *tile, int kq0) {
    T a0[K3][MT], a1[K3][MT];
    bfrag_t b0, b1;
    f32x4 sh[8];
    auto rowoff = [&](int row) { return (((row / 9) * h1 + (row / 3) % 3) * h2 + row % 3) * h3; };
    auto request = [&](T (&a)[K3][MT], bfrag_t &b, int row) {
      b.load(wrow + (int64_t(row) * A.kq_total + kq0) * (256 * NV4));
      const int off = rowoff(row);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const T *pa = tile + abase[mt] + off;
#pragma unroll
        for (int j3 = 0; j3 < K3; ++j3) a[j3][mt] = pa[j3];
      }
    };
    auto multiply = [&](const T (&a)[K3][MT], const bfrag_t &b) {
#pragma unroll
      for (int j3 = 0; j3 < K3; ++j3)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = Mma<T>::mma(a[j3][mt], b.get(j3 * NT + nt), acc[mt][nt]);
    };
    request(a0, b0, 0);
#pragma unroll
    for (int t = 0; t < 13; ++t) {
      request(a1, b1, 2 * t + 1);
      if (t < 8) {
        sh[t] = *reinterpret_cast<const f32x4 *>(nsrc + int64_t((t & 3) < nreal ? (t & 3) : 0) * A.V + woff[t >> 2]);
      }
      __builtin_amdgcn_sched_barrier(0);
      multiply(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      request(a0, b0, 2 * t + 2);
      __builtin_amdgcn_sched_barrier(0);
      multiply(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      if (t >= kLag && t - kLag < 8) {
        constexpr int dummy = 0; (void)dummy;
        const int g8 = t - kLag;
        T *pl = nbuf + (g8 & 3) * A.S;
        const int d = wdst[g8 >> 2];
        const f32x4 v = (g8 & 3) < nreal ? sh[g8] : f32x4{0.f, 0.f, 0.f, 0.f};
        pl[d] = v[0]; pl[d + 1] = v[1]; pl[d + 2] = v[2]; pl[d + 3] = v[3];
        pl[whalo[g8 >> 2]] = wk == 0 ? v[0] : v[3];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    multiply(a0, b0);
  };

  // ---- prologue: the first chunk of the first item is staged synchronously
  const int nst_full = (ngroups + kG - 1) / kG;
  int cb, co[4];
  decode(vb, cb, co);
  set_next(cb, co, 0, buf, true);
  for (int i = 0; i < nst_full; ++i) {
    issue(i);
    commit();
  }
  lds_barrier();

  const int P = n_my * nchunk;
  int q = 0, m = 0;                                       // chunk within the item, item counter
  for (int p = 0; p < P; ++p) {
    T *cur = buf + (p & 1) * bufsz;
    if (q == 0) {
      set_abase(co);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc_t{0.f, 0.f, 0.f, 0.f};
    }
    // what the next phase needs
    int nb_ = cb, no_[4] = {co[0], co[1], co[2], co[3]};
    const bool last_chunk = q + 1 == nchunk;
    if (last_chunk && p + 1 < P) decode(vb + (m + 1) * nb, nb_, no_);
    set_next(nb_, no_, last_chunk ? 0 : q + 1, buf + ((p + 1) & 1) * bufsz, last_chunk);
    if constexpr (HOT) phase_mma_hot(cur, q);
    else phase_mma(cur, q, (p + 1 < P && !NF_DBG(A, 16)) ? nst_full : 0);     // dbg 16: timing ablation, no staging
    lds_barrier();      // everyone is done reading `cur`; the next phase's planes are complete
    if (last_chunk) {
      if (!NF_DBG(A, 32))                                    // dbg 32: timing ablation, no epilogue
      conv_epilogue<T, MT, NT, COMPACT, FUSE>(A, co, cb, int64_t(vb) + int64_t(m) * nb, acc, cur, red, wave, lane);
      lds_barrier();                                    // `cur` may have served as scratch; it is staged into next
      cb = nb_;
#pragma unroll
      for (int mu = 0; mu < 4; ++mu) co[mu] = no_[mu];
      q = 0;
      ++m;
    } else {
      ++q;
    }
  }
}

template <int MT, int NT, int K3, bool COMPACT, int FUSE, bool WIDE, bool HOT>
static int launch_pipe_w(const ConvArgs &A, size_t lds, hipStream_t stream) {
  const void *fn = reinterpret_cast<const void *>(&conv_pipe_kernel<MT, NT, K3, COMPACT, FUSE, WIDE, HOT>);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    ncu = prop.multiProcessorCount;
  }
  if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess) return -1;
  int blocks_per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, fn, kBlock, lds) != hipSuccess || blocks_per_cu < 1)
    blocks_per_cu = 1;
  int64_t grid = int64_t(blocks_per_cu) * ncu;
  if (grid > A.nitems) grid = A.nitems;
  grid = (grid + 7) & ~int64_t(7);
  hipLaunchKernelGGL((conv_pipe_kernel<MT, NT, K3, COMPACT, FUSE, WIDE, HOT>), dim3(unsigned(grid)), dim3(kBlock), lds, stream, A);
  return 1;
}

template <int MT, int NT, int K3, bool COMPACT, int FUSE>
static int launch_pipe_one(const ConvArgs &A, size_t lds, hipStream_t stream) {
  static const int hot_off = NF_DIAG_ENV_INT("NF_CONV_PIPE_ROLLED", 0);      // A/B knob
  const bool hot = !hot_off && A.wide_no == 2 && A.k[0] == 3 && A.k[1] == 3 && A.k[2] == 3 && !NF_DBG(A, 16);
  if (hot) return launch_pipe_w<MT, NT, K3, COMPACT, FUSE, true, true>(A, lds, stream);
  return A.wide_no ? launch_pipe_w<MT, NT, K3, COMPACT, FUSE, true, false>(A, lds, stream)
                   : launch_pipe_w<MT, NT, K3, COMPACT, FUSE, false, false>(A, lds, stream);
}

// Returns 1 when the layer was launched here, 0 when it is not eligible (the caller falls back to
// the one-box kernel), < 0 on error.  `A0` is the fully planned argument block of nf_conv.hip
// (MT = 2 boxes); only the plane stride is re-planned (slack for masked stores).
int launch_conv_pipe(const ConvArgs &A0, int64_t B, int64_t nboxes, int fuse, hipStream_t stream, bool dry) {
  const int off = !option(NF_OPT_PIPE);
  if (off) return 0;
  ConvArgs A = A0;
  if (A.packed || A.k[3] != 3 || NF_DBG(A, 15) || NF_STAMPS(A)) return 0;    // cin % 4 != 0 (>= 8): channel planes zero-padded in LDS
  A.wide_no = 0; A.wide_llpr = 0;
  {
    static const int wide_off = NF_DIAG_ENV_INT("NF_CONV_PIPE_NARROW", 0);     // A/B knob
    const int L3 = A.L[3];
    int64_t rows3 = 1;
    for (int mu = 0; mu < 3; ++mu) rows3 *= A.hal[mu];
    if (!wide_off && A.nbox[3] == 1 && A.box[3] == L3 && (L3 == 8 || L3 == 16 || L3 == 32 || L3 == 64)) {
      int llpr = 1;
      while ((4 << llpr) < L3) ++llpr;
      const int rpi = 64 >> llpr;
      const int64_t no = (rows3 + rpi * 4 - 1) / (rpi * 4);
      if (no <= 4) { A.wide_no = int(no); A.wide_llpr = llpr; }
    }
  }
  if (!A.wide_no && A.hal[3] > 64) return 0;
  if (A.nt_total - A.nt0 > 3 || A.nt0 != 0) return 0;
  const int nrows = A.k[0] * A.k[1] * A.k[2];
  if (nrows < 2) return 0;
  for (int mu = 0; mu < 4; ++mu) {
    const int r = A.k[mu] >> 1;
    if (r > A.L[mu] || A.nbox[mu] * A.box[mu] + r - 1 >= 2 * A.L[mu]) return 0;    // wrap1's range
    if (A.hal[mu] > 255) return 0;
  }
  int64_t halvol = 1, rows = 1;
  for (int mu = 0; mu < 4; ++mu) halvol *= A.hal[mu];
  for (int mu = 0; mu < 3; ++mu) rows *= A.hal[mu];
  int S = int(halvol) + kSlack;
  if (A.compact || A.sh2) S |= 1; else S = ((S + 15) & ~31) + 16;
  A.S = S;
  A.cchunk = 4;
  A.nitems = B * nboxes;
  if (!dry && A.nitems >= (int64_t(1) << 31) - 4096) return -2;
  A.nboxes = int(nboxes);
  const size_t lds = size_t(2) * 4 * S * sizeof(float);
  if (rows > 256) return 0;                                 // one lane per row group of a wave (v_readlane)
  if (lds > 160 * 1024) return 0;
  if (fuse && size_t(48) * ((kBlock / kWave) * 2 * 16 + 4) * sizeof(float) > size_t(4) * S * sizeof(float)) return 0;
  if (A.sh2 && size_t(8) * (2 * (kBlock / kWave) * 2 * 16 + 8) > size_t(4) * S) return 0;   // epilogue scratch = one buffer
  if (dry) return 1;
  const int n = A.nt_total;
  if (fuse) {
    if (n == 3) return fuse == 1 ? launch_pipe_one<2, 3, 3, true, 1>(A, lds, stream) : launch_pipe_one<2, 3, 3, true, 2>(A, lds, stream);
    if (n == 2) return fuse == 1 ? launch_pipe_one<2, 2, 3, true, 1>(A, lds, stream) : launch_pipe_one<2, 2, 3, true, 2>(A, lds, stream);
    return fuse == 1 ? launch_pipe_one<2, 1, 3, true, 1>(A, lds, stream) : launch_pipe_one<2, 1, 3, true, 2>(A, lds, stream);
  }
  if (A.sh2) return launch_pipe_one<2, 1, 4, false, 0>(A, lds, stream);
  if (A.compact) {
    if (n == 3) return launch_pipe_one<2, 3, 3, true, 0>(A, lds, stream);
    if (n == 2) return launch_pipe_one<2, 2, 3, true, 0>(A, lds, stream);
    return launch_pipe_one<2, 1, 3, true, 0>(A, lds, stream);
  }
  if (n == 3) return launch_pipe_one<2, 3, 3, false, 0>(A, lds, stream);
  if (n == 2) return launch_pipe_one<2, 2, 3, false, 0>(A, lds, stream);
  return launch_pipe_one<2, 1, 3, false, 0>(A, lds, stream);
}


// ---------------------------------------------------------------------------------------------------------
// K5c: the first ConvAct layer (ONE input channel, <= 8 output channels, kernel extent 3 on the fastest axis).
// With two-site column packing (columns 0-7 = the layer at site 2p, columns 8-15 = at site 2p+1) the four taps
// -1..+2 along the fastest axis ARE the K = 4 of one MFMA: A[site pair][k] = in[row + 2p + k - 1], so a kernel row
// costs one MFMA per 16 site pairs and the whole layer 27 -- the layer is pure data movement (4 B in, 32 B out
// per site).  Hence: persistent workgroups, the NROWS weight fragments live in registers for the whole launch,
// one input plane double-buffered in LDS (wide 16-byte row staging, issued before and committed after the item's
// arithmetic), and the 8 output planes leave through an LDS transpose as whole 16-byte row pieces.
// Weights: the K-packed layout of nf_conv.hip for cin = 1 (step = kernel row, k = tap): fragment `row` is the
// 64 floats at wfrag + 64*row.
template <int MT, int NROWS>
__global__ __launch_bounds__(kBlock, 2) void conv_c1_kernel(ConvArgs A) {
  typedef float T;
  typedef f32x4 acc_t;
  extern __shared__ __align__(16) unsigned char smem_c1[];
  constexpr int nwaves = kBlock / kWave;
  constexpr int UNITS = nwaves * MT * 16;                   // site pairs per box
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4;
  const int r0 = A.k[0] >> 1, r1 = A.k[1] >> 1, r2 = A.k[2] >> 1;
  const int h1 = A.hal[1], h2 = A.hal[2], h3 = A.hal[3];
  const int R = A.hal[0] * h1 * h2;
  T *buf = reinterpret_cast<T *>(smem_c1);                  // ring of 3 input planes (S floats each), then the output transpose
  T *ot = buf + 3 * A.S;
  const int lb3 = A.lbox[3] - 1;
  const int b3 = 2 << lb3;
  const int CS = 2 * UNITS + 8;

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
  auto rowz_of = [&](int t) {
    const int z0 = t / (h1 * h2), rem = t - z0 * (h1 * h2);
    const int z1 = rem / h2, z2 = rem - z1 * h2;
    return z0 | (z1 << 8) | (z2 << 16);
  };

  // weights: one register per kernel row
  T wreg[NROWS];
#pragma unroll
  for (int r = 0; r < NROWS; ++r) wreg[r] = static_cast<const T *>(A.wfrag)[r * 64 + lane];
  const int co = lane & 7, shift = (lane >> 3) & 1;
  const T bv = (A.bias && co < A.cout) ? static_cast<const T *>(A.bias)[co] : T(0);

  // wide staging state (see conv_pipe_kernel)
  const int llpr = A.wide_llpr, lpr = 1 << llpr, rpi = 64 >> llpr;
  const int wk = lane & (lpr - 1);
  int wpz[4], wdst[4], whalo[4], woff[4] = {0, 0, 0, 0};
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    int row = rpi * (nwaves * o + wave) + (lane >> llpr);
    if (row >= R) row = (lane >> llpr) < R ? (lane >> llpr) : 0;
    wpz[o] = rowz_of(row);
    wdst[o] = row * h3 + 1 + 4 * wk;
    whalo[o] = wk == 0 ? row * h3 + h3 - 1 : (wk == lpr - 1 ? row * h3 : A.S - kSlack + lane);
  }
  const T *__restrict__ nsrc = nullptr;
  auto set_next = [&](int b, const int (&o)[4]) {
    nsrc = static_cast<const T *>(A.in) + int64_t(b) * A.V;
#pragma unroll
    for (int oo = 0; oo < 4; ++oo) {
      if (oo >= A.wide_no) break;
      const int x0 = wrap1(o[0] + (wpz[oo] & 255) - r0, A.L[0]);
      const int x1 = wrap1(o[1] + ((wpz[oo] >> 8) & 255) - r1, A.L[1]);
      const int x2 = wrap1(o[2] + (wpz[oo] >> 16) - r2, A.L[2]);
      woff[oo] = ((x0 * A.L[1] + x1) * A.L[2] + x2) * A.L[3] + 4 * wk;
    }
  };
  // Two register sets: a plane is issued at the start of item m and committed to the LDS ring at the end of item
  // m + 1, two items before it is read -- the loads get a whole item of arithmetic and stores to land (committing
  // in the item that issued them left the layer latency-bound at ~1 TB/s).
  f32x4 shA[4], shB[4];
  auto issue_all = [&](f32x4 (&sh)[4]) {
#pragma unroll
    for (int oo = 0; oo < 4; ++oo)
      if (oo < A.wide_no) sh[oo] = *reinterpret_cast<const f32x4 *>(nsrc + woff[oo]);
  };
  auto commit_all = [&](const f32x4 (&sh)[4], T *pl) {
#pragma unroll
    for (int oo = 0; oo < 4; ++oo)
      if (oo < A.wide_no) {
        const int d = wdst[oo];
        pl[d] = sh[oo][0]; pl[d + 1] = sh[oo][1]; pl[d + 2] = sh[oo][2]; pl[d + 3] = sh[oo][3];
        pl[whalo[oo]] = wk == 0 ? sh[oo][0] : sh[oo][3];
      }
  };

  // A-fragment bases: lane (pair p3 of row z, k-group g) reads in[row + 2*p3 + g]
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
    abase[mt] = ((u * h1 + z1) * h2 + z2) * h3 + 2 * p3 + g;
  }
  auto rowoff = [&](int row) {
    const int j2 = row % A.k[2], t = row / A.k[2];
    return (((t / A.k[1]) * h1 + t % A.k[1]) * h2 + j2) * h3;
  };
  int roff[NROWS];
#pragma unroll
  for (int r = 0; r < NROWS; ++r) roff[r] = rowoff(r);       // uniform; SGPRs / constant-folded adds

  // the block's items are nb apart: walk (sample, box coordinates) with mixed-radix counters, no divisions
  int sb, sc[4];                                            // digits of the step nb
  {
    int so_[4];
    decode(nb, sb, so_);
#pragma unroll
    for (int mu = 0; mu < 4; ++mu) sc[mu] = so_[mu];        // already multiplied by the box extent
  }
  auto advance = [&](int &b, int (&o)[4]) {
    int carry = 0;
#pragma unroll
    for (int mu = 3; mu >= 0; --mu) {
      o[mu] += sc[mu] + carry * A.box[mu];
      const int lim = A.nbox[mu] * A.box[mu];
      carry = o[mu] >= lim ? 1 : 0;
      o[mu] -= carry ? lim : 0;
    }
    b += sb + carry;
  };
  int cb, co4[4], b1, o1[4], b2, o2[4];                     // items m, m + 1, m + 2 (past the end: the last one again)
  const int last = vb + (n_my - 1) * nb;
  decode(vb, cb, co4);
  set_next(cb, co4);
  issue_all(shA);
  commit_all(shA, buf);                                      // item 0 -> plane 0
  b1 = cb;
#pragma unroll
  for (int mu = 0; mu < 4; ++mu) o1[mu] = co4[mu];
  if (n_my > 1) advance(b1, o1);
  b2 = b1;
#pragma unroll
  for (int mu = 0; mu < 4; ++mu) o2[mu] = o1[mu];
  (void)last;
  set_next(b1, o1);
  issue_all(shA);                                            // item 1 -> set A (committed at the end of item 0)
  lds_barrier();

  // split16 output (the bench's path): the transpose is double-buffered and item m-1 is converted and stored between
  // the MFMA groups of item m
  constexpr int NT = (2 * UNITS + kBlock - 1) / kBlock;     // sites per thread
  const bool wov = A.out_split16 != 0;
  int pcb = 0, pco4[4] = {0, 0, 0, 0};
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  auto store_site16 = [&](const float (&v)[8], int b, const int (&o)[4], int t) {
    if (t >= 2 * UNITS) return;
    int zr = t / b3;
    const int x3 = o[3] + (t - zr * b3);
    h8 hi, lo;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const _Float16 hh = static_cast<_Float16>(v[c]);
      hi[c] = hh;
      lo[c] = static_cast<_Float16>(v[c] - static_cast<float>(hh));
    }
    const int z2 = zr & (A.box[2] - 1);
    zr >>= A.lbox[2];
    const int z1 = zr & (A.box[1] - 1);
    zr >>= A.lbox[1];
    const int x0 = o[0] + zr, x1 = o[1] + z1, x2 = o[2] + z2;
    if (x0 < A.L[0] && x1 < A.L[1] && x2 < A.L[2] && x3 < A.L[3]) {
      unsigned char *d = static_cast<unsigned char *>(A.out) + int64_t(b) * A.V * 32 +
                         ((int64_t(x0) * A.L[1] + x1) * A.L[2] + x2) * (A.L[3] * 32) + pair_row_offset(x3, A.L[3]);
      *reinterpret_cast<h8 *>(d) = hi;
      *reinterpret_cast<h8 *>(d + A.L[3] * 16) = lo;
    }
  };
#ifdef NF_C1_TIMING
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define NF_TICK(k) { const unsigned long long tn = __builtin_readcyclecounter(); tacc[k] += tn - tprev; tprev = tn; }
#else
#define NF_TICK(k)
#endif
  auto do_item = [&](int m, f32x4 (&sh_issue)[4], const f32x4 (&sh_commit)[4]) {
    const T *cur = buf + (m % 3) * A.S;
    if (m + 2 < n_my) advance(b2, o2);
    set_next(b2, o2);
    if (!NF_DBG(A, 256)) issue_all(sh_issue);                 // item m + 2 -> registers (dbg 256: timing ablation)
    acc_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = acc_t{0.f, 0.f, 0.f, 0.f};
    if (!NF_DBG(A, 128)) {                                    // dbg 128: timing ablation
      // A values are read three kernel rows ahead into named buffers; left to itself the compiler serialises
      // ds_read -> wait -> MFMA through one register and exposes the LDS latency 27*MT times per item
      constexpr int NG3 = NROWS / 3;
      T a0[3][MT], a1[3][MT];
      auto fetch = [&](T (&a)[3][MT], int g3) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) a[j][mt] = cur[abase[mt] + roff[3 * g3 + j]];
      };
      auto mult = [&](const T (&a)[3][MT], int g3) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt] = Mma<T>::mma(a[j][mt], wreg[3 * g3 + j], acc[mt]);
      };
      const T *otp = ot + ((m + 1) & 1) * (8 * CS);            // item m - 1's transpose
      const bool outp = wov && m > 0;
      float ov[8];
      fetch(a0, 0);
#pragma unroll
      for (int g3 = 0; g3 < NG3; g3 += 2) {
        if (g3 + 1 < NG3) fetch(a1, g3 + 1);
        if (g3 / 2 < NT && outp) {
#pragma unroll
          for (int c = 0; c < 8; ++c) ov[c] = otp[c * CS + ((threadIdx.x + kBlock * (g3 / 2)) < 2 * UNITS ? threadIdx.x + kBlock * (g3 / 2) : 0)];
        }
        __builtin_amdgcn_sched_barrier(0);
        mult(a0, g3);
        __builtin_amdgcn_sched_barrier(0);
        if (g3 + 1 < NG3) {
          if (g3 + 2 < NG3) fetch(a0, g3 + 2);
          __builtin_amdgcn_sched_barrier(0);
          mult(a1, g3 + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (g3 / 2 < NT && outp) store_site16(ov, pcb, pco4, threadIdx.x + kBlock * (g3 / 2));
      }
    } else if (wov && m > 0) {
      const T *otp = ot + ((m + 1) & 1) * (8 * CS);
      for (int k = 0; k < NT; ++k) {
        float ov[8];
        const int t = threadIdx.x + kBlock * k;
#pragma unroll
        for (int c = 0; c < 8; ++c) ov[c] = otp[c * CS + (t < 2 * UNITS ? t : 0)];
        store_site16(ov, pcb, pco4, t);
      }
    }
    NF_TICK(0)      // issue + MFMA loop
    // ---- epilogue: bias + activation -> ot[co][box row][x3] -> 16-byte row pieces
    // (the common activation gets its own straight-line copy: sixteen inlined runtime switches over every
    //  activation made the item body ~40 KB of code, and the kernel instruction-fetch bound)
    if (A.act == kActTanh) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mt][r] = fast_tanh(acc[mt][r] + bv);
    } else {
#pragma unroll 1
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll 1
        for (int r = 0; r < 4; ++r) acc[mt][r] = activate(acc[mt][r] + bv, A.act);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int u = ((wave * MT + mt) << 4) + (g << 2) + r;
        const int p3 = u & ((1 << lb3) - 1), zr = u >> lb3;
        ot[(wov ? (m & 1) * (8 * CS) : 0) + co * CS + zr * b3 + 2 * p3 + shift] = acc[mt][r];
      }
    NF_TICK(1)      // tanh + ot writes
    commit_all(sh_commit, buf + ((m + 1) % 3) * A.S);        // item m + 1 (issued one item ago) -> its plane
    NF_TICK(2)      // commit
    lds_barrier();                                         // ot complete; plane m + 1 complete
    NF_TICK(3)      // barrier
    if (wov) {
      // (stored during the next item's MFMA groups; the last item's after the loop)
    } else {
      T *__restrict__ out_b = static_cast<T *>(A.out) + int64_t(cb) * A.cout * A.V;
      const int lq = lb3 - 1;
      const int lpc = (ilog2_c(UNITS) - lb3) + lq;          // log2(16-byte pieces per channel)
      for (int q = threadIdx.x; q < (8 << lpc); q += kBlock) {
        const int c = q >> lpc, rem = q & ((1 << lpc) - 1);
        int zr = rem >> lq;
        const int c4 = rem & ((1 << lq) - 1);
        const acc_t v = *reinterpret_cast<const acc_t *>(ot + c * CS + zr * b3 + 4 * c4);
        const int z2 = zr & (A.box[2] - 1);
        zr >>= A.lbox[2];
        const int z1 = zr & (A.box[1] - 1);
        zr >>= A.lbox[1];
        const int x0 = co4[0] + zr, x1 = co4[1] + z1, x2 = co4[2] + z2, x3 = co4[3] + 4 * c4;
        if (c < A.cout && x0 < A.L[0] && x1 < A.L[1] && x2 < A.L[2] && x3 < A.L[3] && !NF_DBG(A, 64))   // dbg 64: timing ablation
          *reinterpret_cast<acc_t *>(out_b + int64_t(c) * A.V + ((int64_t(x0) * A.L[1] + x1) * A.L[2] + x2) * A.L[3] + x3) = v;
      }
    }
    NF_TICK(4)      // output
    if (!wov) lds_barrier();                               // ot is free again (split16: the other buffer is written next)
    NF_TICK(5)      // barrier
    pcb = cb;
#pragma unroll
    for (int mu = 0; mu < 4; ++mu) pco4[mu] = co4[mu];
    cb = b1;
    b1 = b2;
#pragma unroll
    for (int mu = 0; mu < 4; ++mu) { co4[mu] = o1[mu]; o1[mu] = o2[mu]; }
  };
  for (int m = 0; m < n_my; m += 2) {
    do_item(m, shB, shA);
    if (m + 1 < n_my) do_item(m + 1, shA, shB);
  }
  if (wov) {                                                 // the last item's output
    const T *otp = ot + ((n_my - 1) & 1) * (8 * CS);
    for (int k = 0; k < NT; ++k) {
      float ov[8];
      const int t = threadIdx.x + kBlock * k;
#pragma unroll
      for (int c = 0; c < 8; ++c) ov[c] = otp[c * CS + (t < 2 * UNITS ? t : 0)];
      store_site16(ov, pcb, pco4, t);
    }
  }
#ifdef NF_C1_TIMING
  if (blockIdx.x == 8 && threadIdx.x == 0 && n_my > 100)
    printf("[c1 timing] items %d | cycles per item: mma %.0f  epilogue %.0f  commit %.0f  barrier %.0f  output %.0f  barrier %.0f\n", n_my,
           double(tacc[0]) / n_my, double(tacc[1]) / n_my, double(tacc[2]) / n_my, double(tacc[3]) / n_my, double(tacc[4]) / n_my, double(tacc[5]) / n_my);
#endif
}

template <int MT, int NROWS>
static int launch_c1(const ConvArgs &A, size_t lds, hipStream_t stream) {
  const void *fn = reinterpret_cast<const void *>(&conv_c1_kernel<MT, NROWS>);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    ncu = prop.multiProcessorCount;
  }
  if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess) return -1;
  int blocks_per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, fn, kBlock, lds) != hipSuccess || blocks_per_cu < 1)
    blocks_per_cu = 1;
  int64_t grid = int64_t(blocks_per_cu) * ncu;
  if (grid > A.nitems) grid = A.nitems;
  grid = (grid + 7) & ~int64_t(7);
  hipLaunchKernelGGL((conv_c1_kernel<MT, NROWS>), dim3(unsigned(grid)), dim3(kBlock), lds, stream, A);
  return 1;
}

// 1 = launched, 0 = not this kernel's layer, < 0 = error.  `MT` is the box size nf_conv.hip planned (2 or 4).
int launch_conv_c1(const ConvArgs &A0, int MT, int64_t B, int64_t nboxes, hipStream_t stream) {
  const int off = !option(NF_OPT_PIPE);
  if (off) return 0;
  ConvArgs A = A0;
  if (A.cin != 1 || !A.sh2 || A.k[3] != 3 || NF_DBG(A, 15) || NF_STAMPS(A) || A.compact) return 0;
  if (A.out_split16 && A.cout != 8) return 0;
  const int nrows = A.k[0] * A.k[1] * A.k[2];
  if (nrows != 27 && nrows != 9 && nrows != 3) return 0;
  const int L3 = A.L[3];
  if (!(A.nbox[3] == 1 && A.box[3] == L3 && (L3 == 8 || L3 == 16 || L3 == 32 || L3 == 64))) return 0;
  for (int mu = 0; mu < 4; ++mu) {
    const int r = A.k[mu] >> 1;
    if (r > A.L[mu] || A.nbox[mu] * A.box[mu] + r - 1 >= 2 * A.L[mu] || A.hal[mu] > 255) return 0;
  }
  int64_t halvol = 1, rows = 1;
  for (int mu = 0; mu < 4; ++mu) halvol *= A.hal[mu];
  for (int mu = 0; mu < 3; ++mu) rows *= A.hal[mu];
  int llpr = 1;
  while ((4 << llpr) < L3) ++llpr;
  const int rpi = 64 >> llpr;
  const int64_t no = (rows + rpi * 4 - 1) / (rpi * 4);
  if (no > 4) return 0;
  A.wide_no = int(no); A.wide_llpr = llpr;
  A.S = (int(halvol) + kSlack + 3) & ~3;
  A.nitems = B * nboxes;
  A.nboxes = int(nboxes);
  if (A.nitems >= (int64_t(1) << 31) - 4096) return -2;
  // three input planes + the output transpose (two of them for the split16 output: item m-1 leaves while item m multiplies)
  const size_t lds = (size_t(3) * A.S + size_t(A.out_split16 ? 2 : 1) * 8 * (2 * (kBlock / kWave) * MT * 16 + 8)) * sizeof(float);
  if (lds > 160 * 1024) return 0;
  if (MT == 4) return nrows == 27 ? launch_c1<4, 27>(A, lds, stream) : (nrows == 9 ? launch_c1<4, 9>(A, lds, stream) : launch_c1<4, 3>(A, lds, stream));
  return nrows == 27 ? launch_c1<2, 27>(A, lds, stream) : (nrows == 9 ? launch_c1<2, 9>(A, lds, stream) : launch_c1<2, 3>(A, lds, stream));
}

}  // namespace nf
