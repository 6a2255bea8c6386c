// nf_distconv.hip -- K4: DistConvertor_ = Expit_ -> SplineNet_ (ONE spline shared by
// all sites) -> Logit_, any subset of the three stages, in one pass over the field.
//
// The K (already boundary-augmented) knots are staged in LDS once per workgroup;
// every lane binary-searches them (spline.py:154-172: left-bisect + clamp == number
// of interior knots strictly below the value), evaluates / inverts the segment
// (spline.py:185-220 / 222-287, stable root) and the per-sample log-det is reduced by
// wave shuffles -> LDS -> one double per workgroup -> finalize kernel.
//
// Restates src/nn/scalar/modules_.py:93-102 (Expit_: y = 1/(1+e^-x), logJ = sum(-x +
// 2 log y)), :105-114 (Logit_: y = log(x/(1-x)), logJ = -sum log(x(1-x))), :277-302
// (SplineNet_) and the list order of :333-358; the inverse chain is the reversed list
// of inverted stages (src/nn/_core.py:69-72), which is again expit -> spline^-1 -> logit.
#include "nf_internal.h"

namespace nf {

constexpr int kMaxSharedKnots = 512;
constexpr int kVjpBlocks = 512;

struct DcArgs {
  const void *v;
  const void *knots;        // 3*K values of T: x | y | d
  void *out;
  double *partial;
  const void *grad_out;
  const void *grad_logj;
  void *grad_in;
  double *knot_partial;     // (gridDim.x * gridDim.y, 3K) doubles
  int64_t V;
  int K, pre_expit, spline, post_logit, iters;
};

template <typename T> struct Seg { T x0, x1, y0, y1, d0, d1; int j; };

template <typename T>
__device__ __forceinline__ Seg<T> find_segment(const T *__restrict__ key, const T *kx, const T *ky,
                                               const T *kd, int K, T v) {
  int lo = 0, hi = K - 2;
  while (lo < hi) {                      // largest j in [0, K-2] with j == 0 or key[j] < v
    const int mid = (lo + hi + 1) >> 1;
    if (key[mid] < v) lo = mid; else hi = mid - 1;
  }
  Seg<T> s;
  s.j = lo;
  s.x0 = kx[lo]; s.x1 = kx[lo + 1];
  s.y0 = ky[lo]; s.y1 = ky[lo + 1];
  s.d0 = kd[lo]; s.d1 = kd[lo + 1];
  return s;
}

// -log(1 + e^-a) for a >= 0, accurate for large a
template <typename T> __device__ __forceinline__ T neg_log1p_exp_neg(T e) {
  // e = exp(-a) in (0, 1]
  return e < Num<T>::tiny_log1p_cut() ? -e * (T(1) - T(0.5) * e) : -nf_log(T(1) + e);
}

template <typename T, bool INV>
__global__ __launch_bounds__(kBlock) void distconv_kernel(DcArgs A) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __shared__ double red[kBlock / kWave];
  T *kx = reinterpret_cast<T *>(smem_raw), *ky = kx + A.K, *kd = ky + A.K;
  if (A.spline)
    for (int i = threadIdx.x; i < 3 * A.K; i += kBlock) kx[i] = static_cast<const T *>(A.knots)[i];
  __syncthreads();
  const int b = blockIdx.y;
  const T *__restrict__ vin = static_cast<const T *>(A.v) + int64_t(b) * A.V;
  T *__restrict__ out = static_cast<T *>(A.out) + int64_t(b) * A.V;
  double acc = 0.0;
  const int64_t base = int64_t(blockIdx.x) * kBlock * A.iters + threadIdx.x;
  for (int it = 0; it < A.iters; ++it) {
    const int64_t i = base + int64_t(it) * kBlock;
    if (i >= A.V) break;
    T u = vin[i], lg = T(0);
    if (A.pre_expit) {
      const T e = nf_exp(-Num<T>::abs(u));           // in (0,1]
      lg += -Num<T>::abs(u) + T(2) * neg_log1p_exp_neg(e);   // log(u(1-u)) = -x + 2 log expit(x)
      const T p = e / (T(1) + e);                    // the smaller of (expit, 1 - expit)
      u = u > T(0) ? T(1) - p : p;
    }
    if (A.spline) {
      const Seg<T> s = find_segment<T>(INV ? ky : kx, kx, ky, kd, A.K, u);
      const T bw = s.x1 - s.x0, bh = s.y1 - s.y0;
      const T sl = bh / bw, curv = s.d0 + s.d1 - T(2) * sl;
      T th;
      if (!INV) {
        th = (u - s.x0) / bw;
      } else {
        const T eta = (u - s.y0) / bh;
        const T a2 = -curv * eta + s.d0 - sl, bb = a2 + sl, a0 = sl * eta;
        const T disc = Num<T>::sqrt(Num<T>::max(bb * bb - T(4) * a0 * a2, T(0)));
        th = (bb >= T(0)) ? T(2) * a0 / (bb + disc) : (bb - disc) / (T(2) * a2);
      }
      const T om = T(1) - th, t1 = th * om;
      const T den = sl + curv * t1;
      const T P = s.d1 * th * th + T(2) * sl * t1 + s.d0 * om * om;
      const T lgs = nf_log(sl * sl * P / (den * den));
      if (!INV) {
        u = s.y0 + bh * (sl * th * th + s.d0 * t1) / den;
        lg += lgs;
      } else {
        u = s.x0 + bw * th;
        lg -= lgs;
      }
    }
    if (A.post_logit) {
      lg -= nf_log(u * (T(1) - u));
      u = nf_log(u / (T(1) - u));
    }
    out[i] = u;
    acc += double(lg);
  }
  const double tot = block_sum(acc, red);
  if (threadIdx.x == 0) A.partial[int64_t(b) * gridDim.x + blockIdx.x] = tot;
}

// VJP.  `v` is the x-side end point of the chain (forward input / inverse output), so
// the chain is always re-run in its forward direction; no root is recomputed.
template <typename T, bool INV>
__global__ __launch_bounds__(kBlock) void distconv_vjp_kernel(DcArgs A, int64_t B) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double *gacc = reinterpret_cast<double *>(smem_raw);          // 3K doubles
  T *kx = reinterpret_cast<T *>(gacc + 3 * A.K), *ky = kx + A.K, *kd = ky + A.K;
  for (int i = threadIdx.x; i < 3 * A.K; i += kBlock) {
    gacc[i] = 0.0;
    if (A.spline) kx[i] = static_cast<const T *>(A.knots)[i];
  }
  __syncthreads();
  const int64_t n = B * A.V;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kBlock) {
    const int64_t b = i / A.V;
    const T p = static_cast<const T *>(A.v)[i];
    const T gout = static_cast<const T *>(A.grad_out)[i];
    const T glog = static_cast<const T *>(A.grad_logj)[b];
    // stage 1: expit
    T u = p, du_dp = T(1), l1p = T(0);
    if (A.pre_expit) {
      const T e = nf_exp(-Num<T>::abs(p));
      const T q = e / (T(1) + e);
      u = p > T(0) ? T(1) - q : q;
      du_dp = q * (T(1) - q);
      l1p = p > T(0) ? T(2) * q - T(1) : T(1) - T(2) * q;   // 1 - 2u
    }
    // stage 2: spline at u
    T w = u, g = T(1), Lu = T(0);
    Seg<T> s{};
    T bw = T(1), bh = T(1), sl = T(1), curv = T(0), th = T(0), om = T(1), t1 = T(0), den = T(1), num = T(0),
      P = T(1), Lth = T(0);
    if (A.spline) {
      s = find_segment<T>(kx, kx, ky, kd, A.K, u);
      bw = s.x1 - s.x0; bh = s.y1 - s.y0;
      sl = bh / bw; curv = s.d0 + s.d1 - T(2) * sl;
      th = (u - s.x0) / bw; om = T(1) - th; t1 = th * om;
      den = sl + curv * t1;
      num = sl * th * th + s.d0 * t1;
      P = s.d1 * th * th + T(2) * sl * t1 + s.d0 * om * om;
      w = s.y0 + bh * num / den;
      g = sl * sl * P / (den * den);
      const T Pp = T(2) * (s.d1 * th + sl * (T(1) - T(2) * th) - s.d0 * om);
      Lth = Pp / P - T(2) * curv * (T(1) - T(2) * th) / den;
      Lu = Lth / bw;
    }
    // stage 3: logit at w
    T dq_dw = T(1), l3w = T(0);
    if (A.post_logit) {
      const T ww = w * (T(1) - w);
      dq_dw = T(1) / ww;
      l3w = -(T(1) - T(2) * w) / ww;
    }
    const T Fp = du_dp * g * dq_dw;                      // d(chain)/dp
    const T Lp = l1p + du_dp * (Lu + g * l3w);           // d(log-det)/dp
    T gq, gl, gin;
    if (!INV) {
      gq = gout; gl = glog;
      gin = gq * Fp + gl * Lp;
    } else {
      gin = (gout - glog * Lp) / Fp;
      gq = -gin; gl = -glog;
    }
    static_cast<T *>(A.grad_in)[i] = gin;
    if (A.spline) {
      const T gw = gq * dq_dw + gl * l3w;                // cotangent on the spline value
      const T iden = T(1) / den, i2 = iden * iden, iP = T(1) / P, ibw = T(1) / bw;
      const T thb = gw * g * bw + gl * Lth;
      const T slb = gw * bh * (th * th * den - num * (T(1) - T(2) * t1)) * i2 +
                    gl * (T(2) / sl + T(2) * t1 * iP - T(2) * (T(1) - T(2) * t1) * iden);
      const T d0b = gw * bh * t1 * (den - num) * i2 + gl * (om * om * iP - T(2) * t1 * iden);
      const T d1b = -gw * bh * num * t1 * i2 + gl * (th * th * iP - T(2) * t1 * iden);
      const T hb = gw * num * iden + slb * ibw;
      const T wb = -(thb * th + slb * sl) * ibw;
      const T x0b = -thb * ibw - wb, y0b = gw - hb;
      atomicAdd(&gacc[s.j], double(x0b));
      atomicAdd(&gacc[s.j + 1], double(wb));
      atomicAdd(&gacc[A.K + s.j], double(y0b));
      atomicAdd(&gacc[A.K + s.j + 1], double(hb));
      atomicAdd(&gacc[2 * A.K + s.j], double(d0b));
      atomicAdd(&gacc[2 * A.K + s.j + 1], double(d1b));
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * A.K; i += kBlock) A.knot_partial[int64_t(blockIdx.x) * 3 * A.K + i] = gacc[i];
}

__global__ __launch_bounds__(kBlock) void knot_reduce_kernel(const double *__restrict__ part, int nblocks, int n,
                                                             double *__restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
  for (int k = 0; k < nblocks; ++k) acc += part[int64_t(k) * n + i];
  out[i] = acc;
}

static int fill(DcArgs &A, int K, int64_t B, int64_t V, int stages, int inverse) {
  NF_REQUIRE(B >= 0 && V >= 0, "nf_distconv: negative size");
  NF_REQUIRE(B <= 65535, "nf_distconv: batch %lld > 65535 (split the batch)", (long long)B);
  NF_REQUIRE(stages >= 1 && stages <= 7, "nf_distconv: stages must be a non-empty subset of {1,2,4}");
  A.spline = (stages & 2) != 0;
  if (A.spline) NF_REQUIRE(K >= 2 && K <= kMaxSharedKnots, "nf_distconv: K=%d outside [2, %d]", K, kMaxSharedKnots);
  A.pre_expit = inverse ? (stages & 4) != 0 : (stages & 1) != 0;
  A.post_logit = inverse ? (stages & 1) != 0 : (stages & 4) != 0;
  A.K = A.spline ? K : 0;
  A.V = V;
  return NF_OK;
}

template <typename T>
static int run_map(const void *v, const void *knots, int K, const void *log0, void *out, void *logj, int64_t B,
                   int64_t V, int stages, int inverse, void *ws, size_t ws_bytes, hipStream_t stream) {
  DcArgs A{};
  int rc = fill(A, K, B, V, stages, inverse);
  if (rc) return rc;
  NF_REQUIRE(v && out && logj && (knots || !A.spline), "nf_distconv: NULL tensor pointer");
  if (B == 0) return NF_OK;
  const Tiling t = make_tiling(V, B);
  const size_t need = size_t(B) * size_t(t.blocks_x > 0 ? t.blocks_x : 1) * sizeof(double);
  if (ws == nullptr || ws_bytes < need) {
    set_error("nf_distconv: workspace %zu B < %zu B needed", ws_bytes, need);
    return NF_EWORKSPACE;
  }
  A.v = v; A.knots = knots; A.out = out; A.partial = static_cast<double *>(ws); A.iters = t.iters;
  if (t.blocks_x > 0) {
    const dim3 grid(unsigned(t.blocks_x), unsigned(B));
    const size_t lds = size_t(3) * A.K * sizeof(T);
    if (inverse) hipLaunchKernelGGL((distconv_kernel<T, true>), grid, dim3(kBlock), lds, stream, A);
    else hipLaunchKernelGGL((distconv_kernel<T, false>), grid, dim3(kBlock), lds, stream, A);
    rc = check_launch("distconv kernel");
    if (rc) return rc;
  }
  return launch_finalize<T>(A.partial, t.blocks_x, log0, logj, B, stream);
}

template <typename T>
static int run_vjp(const void *v, const void *knots, int K, const void *grad_out, const void *grad_logj,
                   void *grad_in, double *grad_knots, int64_t B, int64_t V, int stages, int inverse, void *ws,
                   size_t ws_bytes, hipStream_t stream) {
  DcArgs A{};
  int rc = fill(A, K, B, V, stages, inverse);
  if (rc) return rc;
  NF_REQUIRE(v && grad_out && grad_logj && grad_in && (!A.spline || (knots && grad_knots)),
             "nf_distconv_vjp: NULL tensor pointer");
  const int64_t n = B * V;
  const int n3 = 3 * A.K;
  int blocks = int((n + kBlock - 1) / kBlock);
  if (blocks > kVjpBlocks) blocks = kVjpBlocks;
  if (blocks == 0) {
    if (n3) (void)hipMemsetAsync(grad_knots, 0, size_t(n3) * sizeof(double), stream);
    return NF_OK;
  }
  const size_t need = size_t(blocks) * size_t(n3 > 0 ? n3 : 1) * sizeof(double);
  if (ws == nullptr || ws_bytes < need) {
    set_error("nf_distconv_vjp: workspace %zu B < %zu B needed", ws_bytes, need);
    return NF_EWORKSPACE;
  }
  A.v = v; A.knots = knots; A.grad_out = grad_out; A.grad_logj = grad_logj; A.grad_in = grad_in;
  A.knot_partial = static_cast<double *>(ws);
  const size_t lds = size_t(n3) * (sizeof(double) + sizeof(T));
  if (inverse) hipLaunchKernelGGL((distconv_vjp_kernel<T, true>), dim3(blocks), dim3(kBlock), lds, stream, A, B);
  else hipLaunchKernelGGL((distconv_vjp_kernel<T, false>), dim3(blocks), dim3(kBlock), lds, stream, A, B);
  rc = check_launch("distconv vjp kernel");
  if (rc || n3 == 0) return rc;
  hipLaunchKernelGGL(knot_reduce_kernel, dim3((n3 + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                     A.knot_partial, blocks, n3, grad_knots);
  return check_launch("knot reduce kernel");
}

}  // namespace nf

using namespace nf;

extern "C" int nf_distconv(const void *x, const void *knots, int K, const void *log0, void *y, void *logj,
                           int64_t B, int64_t V, int stages, int inverse, void *workspace,
                           size_t workspace_bytes, int dtype, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == NF_F32) return run_map<float>(x, knots, K, log0, y, logj, B, V, stages, inverse, workspace, workspace_bytes, s);
  if (dtype == NF_F64) return run_map<double>(x, knots, K, log0, y, logj, B, V, stages, inverse, workspace, workspace_bytes, s);
  set_error("nf_distconv: unsupported dtype %d", dtype);
  return NF_EINVAL;
}

extern "C" int nf_distconv_vjp(const void *v, const void *knots, int K, const void *grad_out,
                               const void *grad_logj, void *grad_in, double *grad_knots, int64_t B, int64_t V,
                               int stages, int inverse, void *workspace, size_t workspace_bytes, int dtype,
                               void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == NF_F32) return run_vjp<float>(v, knots, K, grad_out, grad_logj, grad_in, grad_knots, B, V, stages, inverse, workspace, workspace_bytes, s);
  if (dtype == NF_F64) return run_vjp<double>(v, knots, K, grad_out, grad_logj, grad_in, grad_knots, B, V, stages, inverse, workspace, workspace_bytes, s);
  set_error("nf_distconv_vjp: unsupported dtype %d", dtype);
  return NF_EINVAL;
}
