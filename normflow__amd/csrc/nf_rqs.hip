// nf_rqs.hip -- K2/K3: fused rational-quadratic-spline coupling kernels for gfx950.
//
// One lane = one ACTIVE lattice site.  Per site the kernel reads the C = 3m-2 raw
// logits the parameter net produced (channel-strided, so every channel read of a
// wave is one coalesced 256-B / 512-B row), builds the knots in registers (static
// m) or in an LDS column (runtime m), finds the bin by a predicated scan, evaluates
// or inverts the rational-quadratic segment, writes the value and reduces log|J|
// per sample (lane partials -> wave shuffle -> LDS -> one double per workgroup ->
// finalize kernel; no atomics, bitwise reproducible).
//
// Maths restated from the reference (file:line relative to the reference root):
//   knots      src/nn/scalar/couplings_.py:211-262
//   boundary   src/lib/spline/spline.py:458-532 ('linear' = tangent-line tails,
//              'anti' = point reflection through the end knot, done here by
//              reflecting the argument instead of materialising mirrored knots)
//   bin search src/lib/spline/spline.py:154-172 (left-bisect + clamp == number of
//              interior knots strictly below the value)
//   evaluate   src/lib/spline/spline.py:185-220 ; invert :222-287 (stable root)
//   log-det    src/nn/scalar/couplings_.py:186-188, src/nn/_core.py:38-42
#include <type_traits>
#include "nf_rqs_core.h"

namespace nf {

enum { kFwd = 0, kInv = 1 };

struct RqsArgs {
  const void *x;
  const void *params;
  const uint8_t *mask;
  void *y;
  double *partial;          // (B, gridDim.x) or null when the VJP runs
  const void *grad_out;     // VJP only
  const void *grad_logj;    // VJP only
  void *grad_in;            // VJP only
  void *grad_params;        // VJP only
  void *site_out;           // optional (B,V): the derivative (site_mode 2) or its log (1) of every site, 0 at frozen sites
  int site_mode;
  int64_t V;                // sites per sample
  int64_t Vp;               // site extent of params: V (full) or V/2 (pair)
  int64_t units;            // work units per sample: V (full) or V/2 (pair)
  int64_t x_bs, y_bs, p_bs; // batch strides (elements)
  RqsParams P;              // limits, boundary rules, knots_len, fixed knots
  int layout, iters, C;
};

// ------------------------------------------------------------------ kernels
// Unit -> (site, parameter column).  PAIR: unit h covers sites 2h, 2h+1 and the
// active one is read from the mask; FULL: unit = site.
template <typename S> __device__ __forceinline__ S ld_stream(const S *p) { return __builtin_nontemporal_load(p); }
template <> __device__ __forceinline__ __half ld_stream<__half>(const __half *p) {
  return __ushort_as_half(__builtin_nontemporal_load(reinterpret_cast<const unsigned short *>(p)));
}

// T = arithmetic type, S = storage type of x, params and y (S = T, or S = __half with T = float: BASELINE config 5,
// "fp16 params / fp32 log-det accumulate" -- half the HBM bytes per site, log-det partials in double as always).
template <typename T, typename S, int MT, int MODE, bool PAIR>
__global__ __launch_bounds__(kBlock) void rqs_kernel(RqsArgs A) {
  constexpr int C = MT > 0 ? 3 * MT - 2 : 1;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __shared__ double red[kBlock / kWave];
  const int b = blockIdx.y;
  const int C_rt = A.C;
  const S *__restrict__ xin = static_cast<const S *>(A.x) + int64_t(b) * A.x_bs;
  const S *__restrict__ par = static_cast<const S *>(A.params) + int64_t(b) * A.p_bs;
  S *__restrict__ yout = static_cast<S *>(A.y) + int64_t(b) * A.y_bs;
  double acc = 0.0;
  const int64_t base = int64_t(blockIdx.x) * blockDim.x * A.iters + threadIdx.x;
  for (int it = 0; it < A.iters; ++it) {
    const int64_t u = base + int64_t(it) * blockDim.x;
    if (u >= A.units) break;
    T v, val = T(0), logd = T(0);
    bool active;
    int off = 0;
    if (PAIR) {
      const uint16_t mk = reinterpret_cast<const uint16_t *>(A.mask)[u];
      off = (mk & 0xff) ? 0 : 1;
      active = true;
      const typename Pair2<S>::type xv = reinterpret_cast<const typename Pair2<S>::type *>(xin)[u];
      v = T(off ? xv.y : xv.x);
    } else {
      active = A.mask[u] != 0;
      v = T(xin[u]);
    }
    if (active) {
      if constexpr (MT > 0) {
        RegCol<T, C> a;
#pragma unroll
        for (int c = 0; c < C; ++c) a[c] = T(ld_stream(&par[int64_t(c) * A.Vp + u]));   // logits are read once: nontemporal (+8 % on the same box)
        rqs_site<T, MT, MODE == kInv>(a, A.P, v, val, logd);
      } else {
        LdsCol<T> a{reinterpret_cast<T *>(smem_raw) + threadIdx.x, int(blockDim.x)};
        for (int c = 0; c < C_rt; ++c) a[c] = T(par[int64_t(c) * A.Vp + u]);
        rqs_site<T, 0, MODE == kInv>(a, A.P, v, val, logd);
      }
    }
    if (PAIR) {
      // write the pair: transformed value at the active site, 0 at the frozen one
      typename Pair2<S>::type o;
      o.x = S(off ? T(0) : val);
      o.y = S(off ? val : T(0));
      reinterpret_cast<typename Pair2<S>::type *>(yout)[u] = o;
    } else {
      yout[u] = S(val);
    }
    if (A.site_out) {         // what the reference's spline object returns with grad=True (spline.py:87-123), per site
      const T sv = active ? (A.site_mode == 2 ? nf_exp(logd) : logd) : T(0);
      S *so = static_cast<S *>(A.site_out) + int64_t(b) * A.V;
      if (PAIR) {
        typename Pair2<S>::type o;
        o.x = S(off ? T(0) : sv);
        o.y = S(off ? sv : T(0));
        reinterpret_cast<typename Pair2<S>::type *>(so)[u] = o;
      } else {
        so[u] = S(sv);
      }
    }
    acc += double(logd);
  }
  const double tot = block_sum(acc, red);
  if (threadIdx.x == 0) A.partial[int64_t(b) * gridDim.x + blockIdx.x] = tot;
}

template <typename T, int MT, int MODE, bool PAIR>
__global__ __launch_bounds__(kBlock) void rqs_vjp_kernel(RqsArgs A) {
  constexpr int C = MT > 0 ? 3 * MT - 2 : 1;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int b = blockIdx.y;
  const int C_rt = A.C;
  const T *__restrict__ xin = static_cast<const T *>(A.x) + int64_t(b) * A.x_bs;
  const T *__restrict__ par = static_cast<const T *>(A.params) + int64_t(b) * A.p_bs;
  const T *__restrict__ gout = static_cast<const T *>(A.grad_out) + int64_t(b) * A.y_bs;
  T *__restrict__ gin = static_cast<T *>(A.grad_in) + int64_t(b) * A.x_bs;
  T *__restrict__ gpar = static_cast<T *>(A.grad_params) + int64_t(b) * A.p_bs;
  const T glog = static_cast<const T *>(A.grad_logj)[b];
  const int64_t base = int64_t(blockIdx.x) * blockDim.x * A.iters + threadIdx.x;
  for (int it = 0; it < A.iters; ++it) {
    const int64_t u = base + int64_t(it) * blockDim.x;
    if (u >= A.units) break;
    T v, go, gi = T(0);
    bool active;
    int off = 0;
    if (PAIR) {
      const uint16_t mk = reinterpret_cast<const uint16_t *>(A.mask)[u];
      off = (mk & 0xff) ? 0 : 1;
      active = true;
      const typename Pair2<T>::type xv = reinterpret_cast<const typename Pair2<T>::type *>(xin)[u];
      const typename Pair2<T>::type gv = reinterpret_cast<const typename Pair2<T>::type *>(gout)[u];
      v = off ? xv.y : xv.x;
      go = off ? gv.y : gv.x;
    } else {
      active = A.mask[u] != 0;
      v = xin[u];
      go = gout[u];
    }
    if constexpr (MT > 0) {
      RegCol<T, C> a;
      if (active) {
#pragma unroll
        for (int c = 0; c < C; ++c) a[c] = par[int64_t(c) * A.Vp + u];
        gi = rqs_site_vjp<T, MT, MODE == kInv>(a, A.P, v, go, glog);
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) a[c] = T(0);
      }
#pragma unroll
      for (int c = 0; c < C; ++c) gpar[int64_t(c) * A.Vp + u] = a[c];
    } else {
      LdsCol<T> a{reinterpret_cast<T *>(smem_raw) + threadIdx.x, int(blockDim.x)};
      if (active) {
        for (int c = 0; c < C_rt; ++c) a[c] = par[int64_t(c) * A.Vp + u];
        gi = rqs_site_vjp<T, 0, MODE == kInv>(a, A.P, v, go, glog);
        for (int c = 0; c < C_rt; ++c) gpar[int64_t(c) * A.Vp + u] = a[c];
      } else {
        for (int c = 0; c < C_rt; ++c) gpar[int64_t(c) * A.Vp + u] = T(0);
      }
    }
    if (PAIR) {
      typename Pair2<T>::type o;
      o.x = off ? T(0) : gi;
      o.y = off ? gi : T(0);
      reinterpret_cast<typename Pair2<T>::type *>(gin)[u] = o;
    } else {
      gin[u] = gi;
    }
  }
}

// The knots a site's logits stand for -- the tensors `RQSplineCoupling_.make_spline` hands to `RQSpline`
// (couplings_.py:211-262), before the boundary augmentation: out[b][0..m) = knots_x, [m..2m) = knots_y, [2m..3m) = knots_d,
// each a plane of V sites.  Same arithmetic, in the same order, as scan_bins above: these ARE the knots the coupling
// kernels evaluate.  An inspection path (three passes over the logits, no registers arrays): not tuned.
template <typename T>
__global__ __launch_bounds__(kBlock) void rqs_knots_kernel(RqsArgs A) {
  const int b = blockIdx.y;
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= A.V) return;
  const int m = A.P.m, nb = m - 1;
  const T *fx = static_cast<const T *>(A.P.fx), *fy = static_cast<const T *>(A.P.fy);
  const ChanMap cm = chan_map(m, fx != nullptr, fy != nullptr);
  const T *__restrict__ par = static_cast<const T *>(A.params) + int64_t(b) * A.p_bs + u;
  T *__restrict__ out = static_cast<T *>(A.y) + int64_t(b) * 3 * m * A.V + u;
  for (int blk = 0; blk < 2; ++blk) {
    const T *fixed = blk ? fy : fx;
    const T lo = T(blk ? A.P.ylo : A.P.xlo), width = T(blk ? A.P.yhi : A.P.xhi) - lo;
    T *o = out + int64_t(blk) * m * A.V;
    if (fixed) {
      for (int k = 0; k < m; ++k) o[int64_t(k) * A.V] = fixed[k];
      continue;
    }
    const T *a = par + int64_t(blk ? cm.oy : cm.ox) * A.V;
    T amax = a[0];
    for (int k = 1; k < nb; ++k) amax = Num<T>::max(amax, a[int64_t(k) * A.V]);
    T sum = T(0);
    for (int k = 0; k < nb; ++k) sum += Num<T>::exp2((a[int64_t(k) * A.V] - amax) * Num<T>::kLog2e);
    const T w = width / sum;
    T c = lo;
    o[0] = c;
    for (int k = 0; k < nb; ++k) {
      c += Num<T>::exp2((a[int64_t(k) * A.V] - amax) * Num<T>::kLog2e) * w;
      o[int64_t(k + 1) * A.V] = c;
    }
  }
  for (int k = 0; k < m; ++k) out[int64_t(2 * m + k) * A.V] = softplus2(par[int64_t(cm.od + k) * A.V]);
}

// ------------------------------------------------------------------ launchers
static int fill_args(RqsArgs &A, int64_t B, int64_t V, const nf_rqs_opts *o, const nf_strides *st,
                     const uint8_t *mask) {
  NF_REQUIRE(o != nullptr, "nf_rqs: opts is NULL");
  NF_REQUIRE(mask != nullptr, "nf_rqs: mask is NULL");
  NF_REQUIRE(B >= 0 && V >= 0, "nf_rqs: negative size");
  NF_REQUIRE(B <= 65535, "nf_rqs: batch %lld > 65535 (split the batch)", (long long)B);
  NF_REQUIRE(o->m >= 2, "nf_rqs: knots_len m=%d < 2", o->m);
  NF_REQUIRE(o->xhi > o->xlo && o->yhi > o->ylo, "nf_rqs: empty xlim/ylim");
  for (int e : {o->extrap_left, o->extrap_right})
    NF_REQUIRE(e == NF_EXTRAP_NONE || e == NF_EXTRAP_LINEAR || e == NF_EXTRAP_ANTI,
               "nf_rqs: unsupported extrapolation code %d", e);
  NF_REQUIRE(o->layout == NF_LAYOUT_FULL || o->layout == NF_LAYOUT_PAIR, "nf_rqs: bad layout");
  if (o->layout == NF_LAYOUT_PAIR) {
    NF_REQUIRE(V % 2 == 0, "nf_rqs: pair layout needs even V");
    NF_REQUIRE(!st || (st->x_batch % 2 == 0 && st->y_batch % 2 == 0), "nf_rqs: pair layout needs even batch strides");
  }
  const int C = (o->fixed_knots_x ? 0 : o->m - 1) + (o->fixed_knots_y ? 0 : o->m - 1) + o->m;
  A.P.fx = o->fixed_knots_x; A.P.fy = o->fixed_knots_y; A.C = C;
  A.mask = mask;
  A.V = V;
  A.Vp = o->layout == NF_LAYOUT_PAIR ? V / 2 : V;
  A.units = A.Vp;
  A.x_bs = (st && st->x_batch) ? st->x_batch : V;
  A.y_bs = (st && st->y_batch) ? st->y_batch : V;
  A.p_bs = (st && st->params_batch) ? st->params_batch : int64_t(C) * A.Vp;
  A.P.xlo = o->xlo; A.P.xhi = o->xhi; A.P.ylo = o->ylo; A.P.yhi = o->yhi;
  A.P.m = o->m; A.P.el = o->extrap_left; A.P.er = o->extrap_right; A.layout = o->layout;
  return NF_OK;
}

// knots_len values with a register-resident specialisation; everything else takes
// the LDS-column kernel (any m that fits 64 KiB of LDS per workgroup).
#define NF_STATIC_M(X) X(4) X(8) X(16)

static bool has_static_kernel(int m) {
#define NF_IS(MV) if (m == MV) return true;
  NF_STATIC_M(NF_IS)
#undef NF_IS
  return false;
}

// Workgroup size: 256 for the register kernels; the LDS-column kernel keeps one column of
// C logits per lane, so it shrinks the workgroup until the tile fits 64 KiB and opts in to
// the CU's full 160 KiB only for very long splines.  Returns 0 if even 64 lanes do not fit.
template <typename T> static int pick_block(const RqsArgs &A) {
  if (has_static_kernel(A.P.m) && !A.P.fx && !A.P.fy) return kBlock;
  const size_t col = size_t(A.C) * sizeof(T);
  int block = kBlock;
  while (block > kWave && col * block > 64 * 1024) block >>= 1;
  return col * block <= 160 * 1024 ? block : 0;
}

template <typename T, typename S, int MODE, bool VJP>
static int dispatch(const RqsArgs &A, dim3 grid, int block, hipStream_t stream) {
  const bool pair = A.layout == NF_LAYOUT_PAIR;
#define NF_CASE(MV)                                                                         \
  if (A.P.m == MV && !A.P.fx && !A.P.fy) {                                                                          \
    if (VJP) {                                                                              \
      if (pair) hipLaunchKernelGGL((rqs_vjp_kernel<T, MV, MODE, true>), grid, dim3(kBlock), 0, stream, A);  \
      else hipLaunchKernelGGL((rqs_vjp_kernel<T, MV, MODE, false>), grid, dim3(kBlock), 0, stream, A);      \
    } else {                                                                                \
      if (pair) hipLaunchKernelGGL((rqs_kernel<T, S, MV, MODE, true>), grid, dim3(kBlock), 0, stream, A);   \
      else hipLaunchKernelGGL((rqs_kernel<T, S, MV, MODE, false>), grid, dim3(kBlock), 0, stream, A);       \
    }                                                                                       \
    return check_launch("rqs kernel");                                                      \
  }
  NF_STATIC_M(NF_CASE)
#undef NF_CASE
  if constexpr (!std::is_same<T, S>::value) {
    set_error("nf_rqs: fp16 storage is built for knots_len 4, 8 and 16 without fixed knots");
    return NF_EINVAL;
  } else {
  const size_t lds = size_t(A.C) * sizeof(T) * block;
#define NF_LDS_LAUNCH(KERNEL)                                                                     \
  do {                                                                                            \
    if (lds > 64 * 1024)                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL),                          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));            \
    hipLaunchKernelGGL(KERNEL, grid, dim3(block), lds, stream, A);                               \
  } while (0)
  if (VJP) {
    if (pair) NF_LDS_LAUNCH((rqs_vjp_kernel<T, 0, MODE, true>));
    else NF_LDS_LAUNCH((rqs_vjp_kernel<T, 0, MODE, false>));
  } else {
    if (pair) NF_LDS_LAUNCH((rqs_kernel<T, T, 0, MODE, true>));
    else NF_LDS_LAUNCH((rqs_kernel<T, T, 0, MODE, false>));
  }
#undef NF_LDS_LAUNCH
  return check_launch("rqs kernel (lds)");
  }
}

template <typename T, typename S, int MODE>
static int run_map(const void *in, const void *params, const uint8_t *mask, const void *log0, void *out,
                   void *logj, int64_t B, int64_t V, const nf_rqs_opts *o, const nf_strides *st,
                   void *ws, size_t ws_bytes, hipStream_t stream, void *site_out = nullptr, int site_mode = 0) {
  RqsArgs A{};
  int rc = fill_args(A, B, V, o, st, mask);
  if (rc) return rc;
  A.site_out = site_out;
  A.site_mode = site_mode;
  NF_REQUIRE(in && params && out && logj, "nf_rqs: NULL tensor pointer");
  if (B == 0) return NF_OK;
  const int block = pick_block<T>(A);
  NF_REQUIRE(block > 0, "nf_rqs: knots_len m=%d does not fit the 160 KiB of LDS of one CU", A.P.m);
  const Tiling t = make_tiling(A.units, B, block);
  NF_REQUIRE(t.blocks_x <= kMaxBlocksX, "nf_rqs: lattice too large for one launch");
  const size_t need = size_t(B) * size_t(t.blocks_x > 0 ? t.blocks_x : 1) * sizeof(double);
  if (ws == nullptr || ws_bytes < need) {
    set_error("nf_rqs: workspace %zu B < %zu B needed", ws_bytes, need);
    return NF_EWORKSPACE;
  }
  A.x = in; A.params = params; A.y = out; A.partial = static_cast<double *>(ws);
  A.iters = t.iters;
  if (t.blocks_x > 0) {
    rc = dispatch<T, S, MODE, false>(A, dim3(unsigned(t.blocks_x), unsigned(B)), block, stream);
    if (rc) return rc;
  }
  return launch_finalize<T>(A.partial, t.blocks_x, log0, logj, B, stream);
}

template <typename T, int MODE>
static int run_vjp(const void *x, const void *params, const uint8_t *mask, const void *grad_out,
                   const void *grad_logj, void *grad_in, void *grad_params, int64_t B, int64_t V,
                   const nf_rqs_opts *o, const nf_strides *st, hipStream_t stream) {
  RqsArgs A{};
  int rc = fill_args(A, B, V, o, st, mask);
  if (rc) return rc;
  NF_REQUIRE(x && params && grad_out && grad_logj && grad_in && grad_params, "nf_rqs_vjp: NULL tensor pointer");
  if (B == 0 || A.units == 0) return NF_OK;
  const int block = pick_block<T>(A);
  NF_REQUIRE(block > 0, "nf_rqs: knots_len m=%d does not fit the 160 KiB of LDS of one CU", A.P.m);
  const Tiling t = make_tiling(A.units, B, block);
  A.x = x; A.params = params; A.grad_out = grad_out; A.grad_logj = grad_logj;
  A.grad_in = grad_in; A.grad_params = grad_params; A.iters = t.iters;
  return dispatch<T, T, MODE, true>(A, dim3(unsigned(t.blocks_x), unsigned(B)), block, stream);
}

}  // namespace nf

using namespace nf;

extern "C" int nf_rqs_fwd(const void *x, const void *params, const uint8_t *mask, const void *log0,
                          void *y, void *logj, int64_t B, int64_t V, const nf_rqs_opts *opts,
                          const nf_strides *strides, void *workspace, size_t workspace_bytes,
                          int dtype, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == NF_F32) return run_map<float, float, kFwd>(x, params, mask, log0, y, logj, B, V, opts, strides, workspace, workspace_bytes, s);
  if (dtype == NF_F64) return run_map<double, double, kFwd>(x, params, mask, log0, y, logj, B, V, opts, strides, workspace, workspace_bytes, s);
  if (dtype == NF_F16) return run_map<float, __half, kFwd>(x, params, mask, log0, y, logj, B, V, opts, strides, workspace, workspace_bytes, s);
  set_error("nf_rqs_fwd: unsupported dtype %d", dtype);
  return NF_EINVAL;
}

extern "C" int nf_rqs_inv(const void *y, const void *params, const uint8_t *mask, const void *log0,
                          void *x, void *logj, int64_t B, int64_t V, const nf_rqs_opts *opts,
                          const nf_strides *strides, void *workspace, size_t workspace_bytes,
                          int dtype, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == NF_F32) return run_map<float, float, kInv>(y, params, mask, log0, x, logj, B, V, opts, strides, workspace, workspace_bytes, s);
  if (dtype == NF_F64) return run_map<double, double, kInv>(y, params, mask, log0, x, logj, B, V, opts, strides, workspace, workspace_bytes, s);
  if (dtype == NF_F16) return run_map<float, __half, kInv>(y, params, mask, log0, x, logj, B, V, opts, strides, workspace, workspace_bytes, s);
  set_error("nf_rqs_inv: unsupported dtype %d", dtype);
  return NF_EINVAL;
}

extern "C" int nf_rqs_fwd_vjp(const void *x, const void *params, const uint8_t *mask,
                              const void *grad_out, const void *grad_logj, void *grad_in,
                              void *grad_params, int64_t B, int64_t V, const nf_rqs_opts *opts,
                              const nf_strides *strides, int dtype, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == NF_F32) return run_vjp<float, kFwd>(x, params, mask, grad_out, grad_logj, grad_in, grad_params, B, V, opts, strides, s);
  if (dtype == NF_F64) return run_vjp<double, kFwd>(x, params, mask, grad_out, grad_logj, grad_in, grad_params, B, V, opts, strides, s);
  set_error("nf_rqs_fwd_vjp: unsupported dtype %d", dtype);
  return NF_EINVAL;
}

extern "C" int nf_rqs_inv_vjp(const void *x, const void *params, const uint8_t *mask,
                              const void *grad_out, const void *grad_logj, void *grad_in,
                              void *grad_params, int64_t B, int64_t V, const nf_rqs_opts *opts,
                              const nf_strides *strides, int dtype, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == NF_F32) return run_vjp<float, kInv>(x, params, mask, grad_out, grad_logj, grad_in, grad_params, B, V, opts, strides, s);
  if (dtype == NF_F64) return run_vjp<double, kInv>(x, params, mask, grad_out, grad_logj, grad_in, grad_params, B, V, opts, strides, s);
  set_error("nf_rqs_inv_vjp: unsupported dtype %d", dtype);
  return NF_EINVAL;
}

// Per-site derivatives beside the summed log-det: what the reference's spline object returns with grad=True
// (src/lib/spline/spline.py:87-123) and what Module_.sum_density passes through when propagate_density is set
// (src/nn/_core.py:38-42).
static int rqs_sites(int inverse, const void *in, const void *params, const uint8_t *mask, const void *log0, void *out,
                     void *logj, void *site_out, int site_mode, int64_t B, int64_t V, const nf_rqs_opts *opts,
                     const nf_strides *strides, void *workspace, size_t workspace_bytes, int dtype, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  NF_REQUIRE(site_out != nullptr, "nf_rqs_sites: site_out is NULL");
  NF_REQUIRE(site_mode == NF_SITES_LOG || site_mode == NF_SITES_DERIVATIVE, "nf_rqs_sites: site_mode %d", site_mode);
  if (dtype == NF_F32)
    return inverse ? run_map<float, float, kInv>(in, params, mask, log0, out, logj, B, V, opts, strides, workspace, workspace_bytes, s, site_out, site_mode)
                   : run_map<float, float, kFwd>(in, params, mask, log0, out, logj, B, V, opts, strides, workspace, workspace_bytes, s, site_out, site_mode);
  if (dtype == NF_F64)
    return inverse ? run_map<double, double, kInv>(in, params, mask, log0, out, logj, B, V, opts, strides, workspace, workspace_bytes, s, site_out, site_mode)
                   : run_map<double, double, kFwd>(in, params, mask, log0, out, logj, B, V, opts, strides, workspace, workspace_bytes, s, site_out, site_mode);
  set_error("nf_rqs_sites: unsupported dtype %d", dtype);
  return NF_EINVAL;
}

extern "C" int nf_rqs_fwd_sites(const void *x, const void *params, const uint8_t *mask, const void *log0, void *y,
                                void *logj, void *site_out, int site_mode, int64_t B, int64_t V,
                                const nf_rqs_opts *opts, const nf_strides *strides, void *workspace,
                                size_t workspace_bytes, int dtype, void *stream) {
  return rqs_sites(0, x, params, mask, log0, y, logj, site_out, site_mode, B, V, opts, strides, workspace, workspace_bytes, dtype, stream);
}

extern "C" int nf_rqs_inv_sites(const void *y, const void *params, const uint8_t *mask, const void *log0, void *x,
                                void *logj, void *site_out, int site_mode, int64_t B, int64_t V,
                                const nf_rqs_opts *opts, const nf_strides *strides, void *workspace,
                                size_t workspace_bytes, int dtype, void *stream) {
  return rqs_sites(1, y, params, mask, log0, x, logj, site_out, site_mode, B, V, opts, strides, workspace, workspace_bytes, dtype, stream);
}

extern "C" int nf_rqs_knots(const void *params, void *knots, int64_t B, int64_t V, const nf_rqs_opts *opts, int dtype,
                            void *stream) {
  NF_REQUIRE(opts != nullptr, "nf_rqs_knots: opts is NULL");
  NF_REQUIRE(params && knots, "nf_rqs_knots: NULL tensor pointer");
  NF_REQUIRE(opts->layout == NF_LAYOUT_FULL, "nf_rqs_knots: params must be in the full layout (B, C, V)");
  NF_REQUIRE(dtype == NF_F32 || dtype == NF_F64, "nf_rqs_knots: unsupported dtype %d", dtype);
  static const uint8_t unused_mask = 1;
  RqsArgs A{};
  int rc = fill_args(A, B, V, opts, nullptr, &unused_mask);
  if (rc) return rc;
  if (B == 0 || V == 0) return NF_OK;
  A.params = params;
  A.y = knots;
  const dim3 grid(unsigned((V + kBlock - 1) / kBlock), unsigned(B));
  if (dtype == NF_F32) hipLaunchKernelGGL(rqs_knots_kernel<float>, grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), A);
  else hipLaunchKernelGGL(rqs_knots_kernel<double>, grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), A);
  return check_launch("rqs knots kernel");
}
