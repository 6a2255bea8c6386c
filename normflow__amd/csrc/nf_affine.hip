// nf_affine.hip -- K1: affine / shift coupling, value + per-sample log-det, and VJP.
// Restates src/nn/scalar/couplings_.py:123-139 (affine) and :110-116 (shift):
//   fwd  y = t + x e^{-|s|},  logJ -= sum|s|      inv  x = (y - t) e^{|s|},  logJ += sum|s|
// with t, s read only at active sites (the reference's two `purify` passes) and the
// per-sample sum (src/nn/_core.py:38-42) done by wave shuffles + one double per workgroup.
#include <hip/hip_fp16.h>
#include "nf_internal.h"

namespace nf {

struct AffArgs {
  const void *v;
  const void *params;
  const uint8_t *mask;
  void *out;
  double *partial;
  const void *grad_out;
  const void *grad_logj;
  void *grad_in;
  void *grad_params;
  void *site_out;            // optional (B,V): the log-derivative of every site (-|s| / +|s|), 0 at frozen sites
  int64_t V, Vp, units;
  int n_ch, iters;
};

template <typename T> struct Pair2a;
template <> struct Pair2a<float> { typedef float2 type; };
template <> struct Pair2a<double> { typedef double2 type; };

// Storage adaptors: T = arithmetic type; SX = storage of the field (x, y), SP = storage of the parameters.  SX = SP = T for
// fp32 / fp64; SX = __half (and SP = __half or float) for BASELINE config 5's fp16 storage with fp32 arithmetic and log-det.
template <typename T, typename S> struct Store {
  static __device__ __forceinline__ T ld(const S *p, int64_t i) { return T(p[i]); }
  static __device__ __forceinline__ void ld2(const S *p, int64_t u, T &a, T &b) {
    const typename Pair2a<T>::type v = reinterpret_cast<const typename Pair2a<T>::type *>(p)[u];
    a = v.x; b = v.y;
  }
  static __device__ __forceinline__ void st(S *p, int64_t i, T v) { p[i] = S(v); }
  static __device__ __forceinline__ void st2(S *p, int64_t u, T a, T b) {
    typename Pair2a<T>::type o;
    o.x = a; o.y = b;
    reinterpret_cast<typename Pair2a<T>::type *>(p)[u] = o;
  }
};
template <> struct Store<float, __half> {
  static __device__ __forceinline__ float ld(const __half *p, int64_t i) { return __half2float(p[i]); }
  static __device__ __forceinline__ void ld2(const __half *p, int64_t u, float &a, float &b) {
    const float2 v = __half22float2(reinterpret_cast<const __half2 *>(p)[u]);
    a = v.x; b = v.y;
  }
  static __device__ __forceinline__ void st(__half *p, int64_t i, float v) { p[i] = __float2half_rn(v); }
  static __device__ __forceinline__ void st2(__half *p, int64_t u, float a, float b) {
    reinterpret_cast<__half2 *>(p)[u] = __float22half2_rn(float2{a, b});
  }
};

template <typename T, typename SX, typename SP, bool INV, bool PAIR>
__global__ __launch_bounds__(kBlock) void affine_kernel(AffArgs A) {
  __shared__ double red[kBlock / kWave];
  const int b = blockIdx.y;
  const SX *__restrict__ vin = static_cast<const SX *>(A.v) + int64_t(b) * A.V;
  const SP *__restrict__ par = static_cast<const SP *>(A.params) + int64_t(b) * A.n_ch * A.Vp;
  SX *__restrict__ out = static_cast<SX *>(A.out) + int64_t(b) * A.V;
  double acc = 0.0;
  const int64_t base = int64_t(blockIdx.x) * kBlock * A.iters + threadIdx.x;
  for (int it = 0; it < A.iters; ++it) {
    const int64_t u = base + int64_t(it) * kBlock;
    if (u >= A.units) break;
    T v, val = T(0), ls = T(0);
    bool active = true;
    int off = 0;
    if (PAIR) {
      off = (reinterpret_cast<const uint16_t *>(A.mask)[u] & 0xff) ? 0 : 1;
      T xa, xb;
      Store<T, SX>::ld2(vin, u, xa, xb);
      v = off ? xb : xa;
    } else {
      active = A.mask[u] != 0;
      v = Store<T, SX>::ld(vin, u);
    }
    if (active) {
      const T t = Store<T, SP>::ld(par, u);
      const T s = A.n_ch > 1 ? Num<T>::abs(Store<T, SP>::ld(par, A.Vp + u)) : T(0);
      val = INV ? (v - t) * nf_exp(s) : t + v * nf_exp(-s);
      ls = INV ? s : -s;
    }
    if (PAIR) Store<T, SX>::st2(out, u, off ? T(0) : val, off ? val : T(0));
    else Store<T, SX>::st(out, u, val);
    if (A.site_out) {         // propagate_density (src/nn/_core.py:19,38-42): nothing summed
      SX *so = static_cast<SX *>(A.site_out) + int64_t(b) * A.V;
      if (PAIR) Store<T, SX>::st2(so, u, off ? T(0) : ls, off ? ls : T(0));
      else Store<T, SX>::st(so, u, ls);
    }
    acc += double(ls);
  }
  const double tot = block_sum(acc, red);
  if (threadIdx.x == 0) A.partial[int64_t(b) * gridDim.x + blockIdx.x] = tot;
}

template <typename T, bool INV, bool PAIR>
__global__ __launch_bounds__(kBlock) void affine_vjp_kernel(AffArgs A) {
  typedef typename Pair2a<T>::type P2;
  const int b = blockIdx.y;
  const T *__restrict__ vin = static_cast<const T *>(A.v) + int64_t(b) * A.V;
  const T *__restrict__ par = static_cast<const T *>(A.params) + int64_t(b) * A.n_ch * A.Vp;
  const T *__restrict__ gout = static_cast<const T *>(A.grad_out) + int64_t(b) * A.V;
  T *__restrict__ gin = static_cast<T *>(A.grad_in) + int64_t(b) * A.V;
  T *__restrict__ gpar = static_cast<T *>(A.grad_params) + int64_t(b) * A.n_ch * A.Vp;
  const T gl = static_cast<const T *>(A.grad_logj)[b];
  const int64_t base = int64_t(blockIdx.x) * kBlock * A.iters + threadIdx.x;
  for (int it = 0; it < A.iters; ++it) {
    const int64_t u = base + int64_t(it) * kBlock;
    if (u >= A.units) break;
    T v, go, gi = T(0), gt = T(0), gs = T(0);
    bool active = true;
    int off = 0;
    if (PAIR) {
      off = (reinterpret_cast<const uint16_t *>(A.mask)[u] & 0xff) ? 0 : 1;
      const P2 xv = reinterpret_cast<const P2 *>(vin)[u];
      const P2 gv = reinterpret_cast<const P2 *>(gout)[u];
      v = off ? xv.y : xv.x;
      go = off ? gv.y : gv.x;
    } else {
      active = A.mask[u] != 0;
      v = vin[u];
      go = gout[u];
    }
    if (active) {
      const T t = par[u];
      const T sr = A.n_ch > 1 ? par[A.Vp + u] : T(0);
      const T s = Num<T>::abs(sr);
      const T sg = sr > T(0) ? T(1) : (sr < T(0) ? T(-1) : T(0));   // d|s|/ds, 0 at 0 like torch.abs
      if (!INV) {
        const T e = nf_exp(-s);
        gi = go * e;
        gt = go;
        gs = sg * (-go * v * e - gl);
      } else {
        const T e = nf_exp(s);
        gi = go * e;
        gt = -go * e;
        gs = sg * (go * (v - t) * e + gl);
      }
    }
    gpar[u] = gt;
    if (A.n_ch > 1) gpar[A.Vp + u] = gs;
    if (PAIR) {
      P2 o;
      o.x = off ? T(0) : gi;
      o.y = off ? gi : T(0);
      reinterpret_cast<P2 *>(gin)[u] = o;
    } else {
      gin[u] = gi;
    }
  }
}

static int fill(AffArgs &A, int64_t B, int64_t V, int n_ch, int layout, const uint8_t *mask) {
  NF_REQUIRE(mask != nullptr, "nf_affine: mask is NULL");
  NF_REQUIRE(B >= 0 && V >= 0, "nf_affine: negative size");
  NF_REQUIRE(B <= 65535, "nf_affine: batch %lld > 65535 (split the batch)", (long long)B);
  NF_REQUIRE(n_ch == 1 || n_ch == 2, "nf_affine: n_ch must be 2 (affine) or 1 (shift), got %d", n_ch);
  NF_REQUIRE(layout == NF_LAYOUT_FULL || layout == NF_LAYOUT_PAIR, "nf_affine: bad layout");
  if (layout == NF_LAYOUT_PAIR) NF_REQUIRE(V % 2 == 0, "nf_affine: pair layout needs even V");
  A.mask = mask;
  A.V = V;
  A.Vp = layout == NF_LAYOUT_PAIR ? V / 2 : V;
  A.units = A.Vp;
  A.n_ch = n_ch;
  return NF_OK;
}

template <typename T, bool INV, typename SX = T, typename SP = T>
static int run_map(const void *v, const void *params, const uint8_t *mask, const void *log0, void *out,
                   void *logj, int64_t B, int64_t V, int n_ch, int layout, void *ws, size_t ws_bytes,
                   hipStream_t stream, void *site_out = nullptr) {
  AffArgs A{};
  int rc = fill(A, B, V, n_ch, layout, mask);
  if (rc) return rc;
  A.site_out = site_out;
  NF_REQUIRE(v && params && out && logj, "nf_affine: NULL tensor pointer");
  if (B == 0) return NF_OK;
  const Tiling t = make_tiling(A.units, B);
  const size_t need = size_t(B) * size_t(t.blocks_x > 0 ? t.blocks_x : 1) * sizeof(double);
  if (ws == nullptr || ws_bytes < need) {
    set_error("nf_affine: workspace %zu B < %zu B needed", ws_bytes, need);
    return NF_EWORKSPACE;
  }
  A.v = v; A.params = params; A.out = out; A.partial = static_cast<double *>(ws); A.iters = t.iters;
  if (t.blocks_x > 0) {
    const dim3 grid(unsigned(t.blocks_x), unsigned(B));
    if (layout == NF_LAYOUT_PAIR) hipLaunchKernelGGL((affine_kernel<T, SX, SP, INV, true>), grid, dim3(kBlock), 0, stream, A);
    else hipLaunchKernelGGL((affine_kernel<T, SX, SP, INV, false>), grid, dim3(kBlock), 0, stream, A);
    rc = check_launch("affine kernel");
    if (rc) return rc;
  }
  return launch_finalize<T>(A.partial, t.blocks_x, log0, logj, B, stream);
}

template <typename T, bool INV>
static int run_vjp(const void *v, const void *params, const uint8_t *mask, const void *grad_out,
                   const void *grad_logj, void *grad_in, void *grad_params, int64_t B, int64_t V, int n_ch,
                   int layout, hipStream_t stream) {
  AffArgs A{};
  int rc = fill(A, B, V, n_ch, layout, mask);
  if (rc) return rc;
  NF_REQUIRE(v && params && grad_out && grad_logj && grad_in && grad_params, "nf_affine_vjp: NULL tensor pointer");
  if (B == 0 || A.units == 0) return NF_OK;
  const Tiling t = make_tiling(A.units, B);
  A.v = v; A.params = params; A.grad_out = grad_out; A.grad_logj = grad_logj;
  A.grad_in = grad_in; A.grad_params = grad_params; A.iters = t.iters;
  const dim3 grid(unsigned(t.blocks_x), unsigned(B));
  if (layout == NF_LAYOUT_PAIR) hipLaunchKernelGGL((affine_vjp_kernel<T, INV, true>), grid, dim3(kBlock), 0, stream, A);
  else hipLaunchKernelGGL((affine_vjp_kernel<T, INV, false>), grid, dim3(kBlock), 0, stream, A);
  return check_launch("affine vjp kernel");
}

}  // namespace nf

using namespace nf;

extern "C" int nf_affine_fwd(const void *x, const void *params, const uint8_t *mask, const void *log0,
                             void *y, void *logj, int64_t B, int64_t V, int n_ch, int layout,
                             void *workspace, size_t workspace_bytes, int dtype, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == NF_F32) return run_map<float, false>(x, params, mask, log0, y, logj, B, V, n_ch, layout, workspace, workspace_bytes, s);
  if (dtype == NF_F64) return run_map<double, false>(x, params, mask, log0, y, logj, B, V, n_ch, layout, workspace, workspace_bytes, s);
  if (dtype == NF_F16) return run_map<float, false, __half, __half>(x, params, mask, log0, y, logj, B, V, n_ch, layout, workspace, workspace_bytes, s);
  if (dtype == NF_F16_FIELD) return run_map<float, false, __half, float>(x, params, mask, log0, y, logj, B, V, n_ch, layout, workspace, workspace_bytes, s);
  set_error("nf_affine_fwd: unsupported dtype %d", dtype);
  return NF_EINVAL;
}

extern "C" int nf_affine_inv(const void *y, const void *params, const uint8_t *mask, const void *log0,
                             void *x, void *logj, int64_t B, int64_t V, int n_ch, int layout,
                             void *workspace, size_t workspace_bytes, int dtype, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == NF_F32) return run_map<float, true>(y, params, mask, log0, x, logj, B, V, n_ch, layout, workspace, workspace_bytes, s);
  if (dtype == NF_F64) return run_map<double, true>(y, params, mask, log0, x, logj, B, V, n_ch, layout, workspace, workspace_bytes, s);
  if (dtype == NF_F16) return run_map<float, true, __half, __half>(y, params, mask, log0, x, logj, B, V, n_ch, layout, workspace, workspace_bytes, s);
  if (dtype == NF_F16_FIELD) return run_map<float, true, __half, float>(y, params, mask, log0, x, logj, B, V, n_ch, layout, workspace, workspace_bytes, s);
  set_error("nf_affine_inv: unsupported dtype %d", dtype);
  return NF_EINVAL;
}

extern "C" int nf_affine_vjp(const void *v, const void *params, const uint8_t *mask, const void *grad_out,
                             const void *grad_logj, void *grad_in, void *grad_params, int64_t B, int64_t V,
                             int n_ch, int layout, int inverse, int dtype, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == NF_F32) {
    return inverse ? run_vjp<float, true>(v, params, mask, grad_out, grad_logj, grad_in, grad_params, B, V, n_ch, layout, s)
                   : run_vjp<float, false>(v, params, mask, grad_out, grad_logj, grad_in, grad_params, B, V, n_ch, layout, s);
  }
  if (dtype == NF_F64) {
    return inverse ? run_vjp<double, true>(v, params, mask, grad_out, grad_logj, grad_in, grad_params, B, V, n_ch, layout, s)
                   : run_vjp<double, false>(v, params, mask, grad_out, grad_logj, grad_in, grad_params, B, V, n_ch, layout, s);
  }
  set_error("nf_affine_vjp: unsupported dtype %d", dtype);
  return NF_EINVAL;
}

// nf_affine_fwd / nf_affine_inv with the log-derivative of every site written beside the sum (fp32 / fp64).
extern "C" int nf_affine_sites(const void *v, const void *params, const uint8_t *mask, const void *log0, void *out,
                               void *logj, void *site_out, int64_t B, int64_t V, int n_ch, int layout, int inverse,
                               void *workspace, size_t workspace_bytes, int dtype, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  NF_REQUIRE(site_out != nullptr, "nf_affine_sites: site_out is NULL");
  if (dtype == NF_F32)
    return inverse ? run_map<float, true>(v, params, mask, log0, out, logj, B, V, n_ch, layout, workspace, workspace_bytes, s, site_out)
                   : run_map<float, false>(v, params, mask, log0, out, logj, B, V, n_ch, layout, workspace, workspace_bytes, s, site_out);
  if (dtype == NF_F64)
    return inverse ? run_map<double, true>(v, params, mask, log0, out, logj, B, V, n_ch, layout, workspace, workspace_bytes, s, site_out)
                   : run_map<double, false>(v, params, mask, log0, out, logj, B, V, n_ch, layout, workspace, workspace_bytes, s, site_out);
  set_error("nf_affine_sites: unsupported dtype %d", dtype);
  return NF_EINVAL;
}
