// nf_spline.hip -- a rational-quadratic spline given by EXPLICIT knot tensors, evaluated at every site:
// the reference's generic spline object `RQSpline(knots_x, knots_y, knots_d)` = `Pade22Spline`
// (src/lib/spline/spline.py:39-68 constructor, :87-123 forward / backward, :154-172 searchsorted + clamp,
// :185-220 segment function, :222-287 inverse).  The coupling kernels (nf_rqs.hip) never materialise knots; this
// kernel is for callers that hold them (a spline built by hand, or inspected and modified after make_spline).
//
// One lane = one site.  Knot tensors are (B, K, V) planes (lane-coalesced reads along V) or shared 1-D vectors of K
// entries (the reference's 1-D knots_x / knots_y case, spline.py:191-194).  The knots are taken as given -- already
// augmented for extrapolation (the host mirrors AugmentKnots as a layout operation) -- and values outside the knot
// range reuse the first / last segment, as the reference's clamp does (:171-172).  HBM-bound: 3K reads per site.
#include "nf_internal.h"

namespace nf {

struct SplineArgs {
  const void *v, *kx, *ky, *kd;
  void *out, *deriv;
  int64_t V;
  int K, sx, sy, sd, inverse;
};

template <typename T> __global__ __launch_bounds__(kBlock) void spline_eval_kernel(SplineArgs A) {
  const int64_t site = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (site >= A.V) return;
  const int64_t b = blockIdx.y;
  const int K = A.K;
  const int64_t V = A.V;
  const T *kx = static_cast<const T *>(A.kx), *ky = static_cast<const T *>(A.ky), *kd = static_cast<const T *>(A.kd);
  // element k of a knot tensor at this site: shared vectors are indexed by k alone
  const int64_t ox = A.sx ? 0 : b * K * V + site, oy = A.sy ? 0 : b * K * V + site, od = A.sd ? 0 : b * K * V + site;
  const int64_t tx = A.sx ? 1 : V, ty = A.sy ? 1 : V, td = A.sd ? 1 : V;
  const T val = static_cast<const T *>(A.v)[b * V + site];
  // searchsorted(left) then clamp(idx, 1, K-1) - 1 = the number of INTERIOR knots strictly below the value
  // (the knots are sorted: walk up while the next interior knot is still below)
  T x0 = kx[ox], y0 = ky[oy], d0 = kd[od];
  T x1 = kx[ox + tx], y1 = ky[oy + ty], d1 = kd[od + td];
  for (int k = 1; k < K - 1; ++k) {          // (x1, y1, d1) holds knot k here
    if (!((A.inverse ? y1 : x1) < val)) break;
    x0 = x1; y0 = y1; d0 = d1;
    x1 = kx[ox + (k + 1) * tx]; y1 = ky[oy + (k + 1) * ty]; d1 = kd[od + (k + 1) * td];
  }
  const T bw = x1 - x0, bh = y1 - y0;
  const T sl = bh / bw;
  const T curv = d0 + d1 - T(2) * sl;
  T th, res;
  if (!A.inverse) {
    th = (val - x0) / bw;
    const T t1 = th * (T(1) - th);
    res = y0 + bh * th * (sl * th + d0 * (T(1) - th)) / (sl + curv * t1);
  } else {
    const T eta = (val - y0) / bh;
    const T a2 = -curv * eta + d0 - sl;
    const T bb = a2 + sl;
    const T a0 = sl * eta;
    const T disc = Num<T>::sqrt(Num<T>::max(bb * bb - T(4) * a0 * a2, T(0)));
    th = (bb >= T(0)) ? T(2) * a0 / (bb + disc) : (bb - disc) / (T(2) * a2);   // neither branch cancels
    res = x0 + bw * th;
  }
  static_cast<T *>(A.out)[b * V + site] = res;
  if (A.deriv) {
    const T den = sl + curv * th * (T(1) - th);
    const T g = sl * sl * (d0 + T(2) * (sl - d0) * th + curv * th * th) / (den * den);
    static_cast<T *>(A.deriv)[b * V + site] = A.inverse ? T(1) / g : g;
  }
}

}  // namespace nf

extern "C" int nf_spline_eval(const void *v, const void *knots_x, const void *knots_y, const void *knots_d, void *out,
                              void *deriv, int64_t B, int64_t V, int K, int shared_x, int shared_y, int shared_d,
                              int inverse, int dtype, void *stream) {
  using namespace nf;
  NF_REQUIRE(B >= 0 && V >= 0, "nf_spline_eval: negative size (B=%lld, V=%lld)", (long long)B, (long long)V);
  NF_REQUIRE(K >= 2, "nf_spline_eval: a spline needs at least 2 knots, got %d", K);
  NF_REQUIRE(dtype == NF_F32 || dtype == NF_F64, "nf_spline_eval: unsupported dtype %d", dtype);
  NF_REQUIRE(B <= 65535, "nf_spline_eval: batch %lld exceeds the grid's y extent; slab it", (long long)B);
  if (B == 0 || V == 0) return NF_OK;
  NF_REQUIRE(v && knots_x && knots_y && knots_d && out, "nf_spline_eval: NULL tensor pointer");
  SplineArgs A{v, knots_x, knots_y, knots_d, out, deriv, V, K, shared_x != 0, shared_y != 0, shared_d != 0, inverse != 0};
  const dim3 grid(unsigned((V + kBlock - 1) / kBlock), unsigned(B));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == NF_F32) hipLaunchKernelGGL(spline_eval_kernel<float>, grid, dim3(kBlock), 0, s, A);
  else hipLaunchKernelGGL(spline_eval_kernel<double>, grid, dim3(kBlock), 0, s, A);
  return check_launch("spline eval kernel");
}
