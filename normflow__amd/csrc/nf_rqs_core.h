// nf_rqs_core.h -- device-side core of the rational-quadratic-spline coupling: logits of one
// site (in registers or in an LDS column) -> knots -> bin -> value, log|derivative| and VJP.
// Shared by the stand-alone coupling kernels (nf_rqs.hip) and by the conv kernel's fused
// epilogue (nf_conv.hip).  Reference lines restated: see the header of nf_rqs.hip.
#pragma once
#include <hip/hip_fp16.h>
#include "nf_internal.h"

namespace nf {

// The options of one spline family as the device code sees them.
struct RqsParams {
  double xlo, xhi, ylo, yhi;
  const void *fx, *fy;       // optional fixed knot coordinates (m values of T), LDS-column kernels only
  int m, el, er;
};

// ------------------------------------------------------------ parameter columns
template <typename T, int C> struct RegCol {   // static m: logits live in VGPRs
  T v[C];
  __device__ __forceinline__ T &operator[](int i) { return v[i]; }
};
template <typename T> struct LdsCol {          // runtime m: one LDS column per lane
  T *p;                                        // row stride = blockDim.x (a multiple of 64):
  int stride;                                  // bank = lane % 32 for every row, conflict-free
  __device__ __forceinline__ T &operator[](int i) const { return p[i * stride]; }
};

template <typename T> struct Pair2;   // two adjacent sites as one 8/16-byte access
template <> struct Pair2<float> { typedef float2 type; };
template <> struct Pair2<double> { typedef double2 type; };
template <> struct Pair2<__half> { typedef __half2 type; };

template <typename T> struct Site {   // what the scan selects for one site
  T x0, y0, bw, bh, c0, c1, xe, ye;
  int j;
};

// Channel layout of the logits: [x widths (m-1) | y heights (m-1) | derivatives (m)], where
// the x (y) block is absent when knots_x (knots_y) is fixed (couplings_.py:236-262).
struct ChanMap { int ox, oy, od; };
__device__ __forceinline__ ChanMap chan_map(int m, bool fixx, bool fixy) {
  const int nb = m - 1;
  ChanMap c;
  c.ox = 0;
  c.oy = fixx ? 0 : nb;
  c.od = (fixx ? 0 : nb) + (fixy ? 0 : nb);
  return c;
}

// Softmax numerators in place, then the predicated bin scan.  On return the x and y logit
// blocks of `a` hold exp(logit - max); sa/sb their sums.  With fixed knot coordinates (only
// reachable in the LDS-column kernel, MT == 0) bin widths come from the fixed array.
template <typename T, int MT, bool ON_Y, typename Col>
__device__ __forceinline__ Site<T> scan_bins(Col &a, const RqsParams &A, T v, T xlo, T W, T ylo, T H, T &sa,
                                             T &sb) {
  const int m = MT > 0 ? MT : A.m;
  const int nb = m - 1;
  const T *fx = MT > 0 ? nullptr : static_cast<const T *>(A.fx);
  const T *fy = MT > 0 ? nullptr : static_cast<const T *>(A.fy);
  const ChanMap cm = chan_map(m, fx != nullptr, fy != nullptr);
  sa = T(1);
  sb = T(1);
  if (!fx) {
    T amax = a[cm.ox];
#pragma unroll
    for (int k = 1; k < nb; ++k) amax = Num<T>::max(amax, a[cm.ox + k]);
    sa = T(0);
#pragma unroll
    for (int k = 0; k < nb; ++k) {
      const T e = Num<T>::exp2((a[cm.ox + k] - amax) * Num<T>::kLog2e);
      a[cm.ox + k] = e;
      sa += e;
    }
  }
  if (!fy) {
    T bmax = a[cm.oy];
#pragma unroll
    for (int k = 1; k < nb; ++k) bmax = Num<T>::max(bmax, a[cm.oy + k]);
    sb = T(0);
#pragma unroll
    for (int k = 0; k < nb; ++k) {
      const T e = Num<T>::exp2((a[cm.oy + k] - bmax) * Num<T>::kLog2e);
      a[cm.oy + k] = e;
      sb += e;
    }
  }
  const T wx = W / sa, wy = H / sb;
  Site<T> s;
  T cx = xlo, cy = ylo;
  s.x0 = xlo; s.y0 = ylo;
  s.bw = fx ? fx[1] - fx[0] : a[cm.ox] * wx;
  s.bh = fy ? fy[1] - fy[0] : a[cm.oy] * wy;
  s.c0 = a[cm.od]; s.c1 = a[cm.od + 1]; s.j = 0;
  cx = fx ? fx[1] : cx + s.bw;
  cy = fy ? fy[1] : cy + s.bh;
#pragma unroll
  for (int k = 1; k < nb; ++k) {
    const T wk = fx ? fx[k + 1] - fx[k] : a[cm.ox + k] * wx;
    const T hk = fy ? fy[k + 1] - fy[k] : a[cm.oy + k] * wy;
    const bool sel = (ON_Y ? cy : cx) < v;   // knot k strictly below the value
    s.x0 = sel ? cx : s.x0;
    s.y0 = sel ? cy : s.y0;
    s.bw = sel ? wk : s.bw;
    s.bh = sel ? hk : s.bh;
    s.c0 = sel ? a[cm.od + k] : s.c0;
    s.c1 = sel ? a[cm.od + k + 1] : s.c1;
    s.j = sel ? k : s.j;
    cx = fx ? fx[k + 1] : cx + wk;
    cy = fy ? fy[k + 1] : cy + hk;
  }
  s.xe = cx; s.ye = cy;   // last knot as accumulated (the reference's cumsum end)
  return s;
}

// Value and log|derivative| of the map at one site.  INV=false: v is x, returns y
// and log(dy/dx).  INV=true: v is y, returns x and log(dx/dy) = -log g.
template <typename T, int MT, bool INV, typename Col>
__device__ __forceinline__ void rqs_site(Col &a, const RqsParams &A, T v, T &val, T &logd) {
  const T xlo = T(A.xlo), W = T(A.xhi) - T(A.xlo), ylo = T(A.ylo), H = T(A.yhi) - T(A.ylo);
  const T in_lo = INV ? ylo : xlo, in_hi = INV ? ylo + H : xlo + W;
  const T out_lo = INV ? xlo : ylo, out_hi = INV ? xlo + W : ylo + H;
  const bool refl_l = (A.el == NF_EXTRAP_ANTI) && (v < in_lo);
  const bool refl_r = (A.er == NF_EXTRAP_ANTI) && (v > in_hi);
  v = refl_l ? T(2) * in_lo - v : (refl_r ? T(2) * in_hi - v : v);
  T sa, sb;
  const Site<T> s = scan_bins<T, MT, INV>(a, A, v, xlo, W, ylo, H, sa, sb);
  const bool tail_l = (A.el == NF_EXTRAP_LINEAR) && !(in_lo < v);
  const bool tail_r = (A.er == NF_EXTRAP_LINEAR) && ((INV ? s.ye : s.xe) < v);
  const T d0 = softplus2(s.c0), d1 = softplus2(s.c1);
  const T sl = s.bh / s.bw;            // segment slope
  const T curv = d0 + d1 - T(2) * sl;
  T th, g;
  if (!INV) {
    th = (v - s.x0) / s.bw;
    const T t1 = th * (T(1) - th);
    const T den = sl + curv * t1;
    val = s.y0 + s.bh * (sl * th * th + d0 * t1) / den;
    const T P = d1 * th * th + T(2) * sl * t1 + d0 * (T(1) - th) * (T(1) - th);
    g = sl * sl * P / (den * den);
    val = tail_l ? ylo + d0 * (v - xlo) : (tail_r ? s.ye + d1 * (v - s.xe) : val);
    g = tail_l ? d0 : (tail_r ? d1 : g);
    logd = nf_log(g);
  } else {
    const T eta = (v - s.y0) / s.bh;
    const T a2 = -curv * eta + d0 - sl;
    const T bb = a2 + sl;              // = -a1
    const T a0 = sl * eta;
    const T disc = Num<T>::sqrt(Num<T>::max(bb * bb - T(4) * a0 * a2, T(0)));
    // the root in [0,1], written so that neither branch cancels
    th = (bb >= T(0)) ? T(2) * a0 / (bb + disc) : (bb - disc) / (T(2) * a2);
    const T t1 = th * (T(1) - th);
    const T den = sl + curv * t1;
    const T P = d1 * th * th + T(2) * sl * t1 + d0 * (T(1) - th) * (T(1) - th);
    g = sl * sl * P / (den * den);
    val = s.x0 + s.bw * th;
    val = tail_l ? xlo + (v - ylo) / d0 : (tail_r ? s.xe + (v - s.ye) / d1 : val);
    g = tail_l ? d0 : (tail_r ? d1 : g);
    logd = -nf_log(g);
  }
  val = refl_l ? T(2) * out_lo - val : (refl_r ? T(2) * out_hi - val : val);
}

// VJP at one site.  `x` is the point on the x axis (forward input, or inverse
// output).  gout / glog are the cotangents of (value, log-det) of the map selected
// by INV.  Writes the C parameter cotangents back into `a` and returns grad_in.
template <typename T, int MT, bool INV, typename Col>
__device__ __forceinline__ T rqs_site_vjp(Col &a, const RqsParams &A, T x, T gout, T glog) {
  const int m = MT > 0 ? MT : A.m;
  const int nb = m - 1;
  const T xlo = T(A.xlo), W = T(A.xhi) - T(A.xlo), ylo = T(A.ylo), H = T(A.yhi) - T(A.ylo);
  const bool refl_l = (A.el == NF_EXTRAP_ANTI) && (x < xlo);
  const bool refl_r = (A.er == NF_EXTRAP_ANTI) && (x > xlo + W);
  const T sgn = (refl_l || refl_r) ? T(-1) : T(1);
  const T v = refl_l ? T(2) * xlo - x : (refl_r ? T(2) * (xlo + W) - x : x);
  T sa, sb;
  const Site<T> s = scan_bins<T, MT, false>(a, A, v, xlo, W, ylo, H, sa, sb);
  const bool tail_l = (A.el == NF_EXTRAP_LINEAR) && !(xlo < v);
  const bool tail_r = (A.er == NF_EXTRAP_LINEAR) && (s.xe < v);
  const bool tail = tail_l || tail_r;
  T sg0, sg1;
  const T d0 = softplus2(s.c0, &sg0), d1 = softplus2(s.c1, &sg1);
  const T ibw = T(1) / s.bw;
  const T sl = s.bh * ibw;
  const T curv = d0 + d1 - T(2) * sl;
  const T th = (v - s.x0) * ibw;
  const T om = T(1) - th;
  const T t1 = th * om;
  const T den = sl + curv * t1, iden = T(1) / den;
  const T num = sl * th * th + d0 * t1;
  const T P = d1 * th * th + T(2) * sl * t1 + d0 * om * om;
  const T iP = T(1) / P;
  T g = sl * sl * P * iden * iden;
  g = tail_l ? d0 : (tail_r ? d1 : g);
  // dL/dtheta, L = log g (0 on the linear tails)
  const T Pp = T(2) * (d1 * th + sl * (T(1) - T(2) * th) - d0 * om);
  const T Lth = tail ? T(0) : (Pp * iP - T(2) * curv * (T(1) - T(2) * th) * iden);
  // cotangents (gy on the value of the forward map in the actual frame, gl on log g)
  T gy, gl, grad_in;
  if (!INV) {
    gy = gout; gl = glog;
    grad_in = gy * g + gl * sgn * Lth * ibw;
  } else {
    // inverse outputs (x, -L):  dx = (dy - f_p dp)/g ,  d(-L) = -(L_x dx + L_p dp)
    // => cotangent of y: A1 = (gout - glog L_x)/g ; of p: -A1 f_p - glog L_p
    const T Lx = sgn * Lth * ibw;
    const T A1 = (gout - glog * Lx) / g;
    grad_in = A1;
    gy = -A1; gl = -glog;
  }
  const T gyF = sgn * gy;   // cotangent on F's value in the unreflected frame
  T d0b, d1b, x0b, wb, y0b, hb;
  if (tail) {
    d0b = tail_l ? gyF * (v - xlo) + gl / d0 : T(0);
    d1b = tail_r ? gyF * (v - s.xe) + gl / d1 : T(0);
    x0b = wb = y0b = hb = T(0);
  } else {
    const T thb = gyF * g * s.bw + gl * Lth;
    const T i2 = iden * iden;
    const T slb = gyF * s.bh * (th * th * den - num * (T(1) - T(2) * t1)) * i2 +
                  gl * (T(2) / sl + T(2) * t1 * iP - T(2) * (T(1) - T(2) * t1) * iden);
    d0b = gyF * s.bh * t1 * (den - num) * i2 + gl * (om * om * iP - T(2) * t1 * iden);
    d1b = -gyF * s.bh * num * t1 * i2 + gl * (th * th * iP - T(2) * t1 * iden);
    hb = gyF * num * iden + slb * ibw;
    y0b = gyF;
    x0b = -thb * ibw;
    wb = -(thb * th + slb * sl) * ibw;
  }
  // back through softmax / cumsum (the free x / y blocks of a[] hold the softmax numerators)
  const bool fixx = MT == 0 && A.fx != nullptr, fixy = MT == 0 && A.fy != nullptr;
  const ChanMap cm = chan_map(m, fixx, fixy);
  const T gc0 = d0b * sg0, gc1 = d1b * sg1;
  if (!fixx) {
    const T Sx = x0b * (s.x0 - xlo) + wb * s.bw;
    const T isa = T(1) / sa;
#pragma unroll
    for (int k = 0; k < nb; ++k) {
      const T lead_x = (k < s.j) ? x0b : ((k == s.j) ? wb : T(0));
      a[cm.ox + k] = a[cm.ox + k] * isa * (W * lead_x - Sx);
    }
  }
  if (!fixy) {
    const T Sy = y0b * (s.y0 - ylo) + hb * s.bh;
    const T isb = T(1) / sb;
#pragma unroll
    for (int k = 0; k < nb; ++k) {
      const T lead_y = (k < s.j) ? y0b : ((k == s.j) ? hb : T(0));
      a[cm.oy + k] = a[cm.oy + k] * isb * (H * lead_y - Sy);
    }
  }
#pragma unroll
  for (int k = 0; k < m; ++k) a[cm.od + k] = (k == s.j) ? gc0 : ((k == s.j + 1) ? gc1 : T(0));
  return grad_in;
}


}  // namespace nf
