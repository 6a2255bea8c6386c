// nf_endpoints.hip -- the two per-sample reductions on either side of the flow:
//   * ScalarPhi4Action.action   (reference: src/action/scalar_action.py:24-46)
//       S[b] = sum_x ( w2 phi^2 + w4 phi^4 ) - w0 sum_mu sum_x phi(x) phi(x - mu)
//     the reference makes d+1 full passes (pow, pow, sum, d x (roll, mul, sum)); here one pass:
//     every lane owns a site, reads its d backward neighbours (L2 hits), and the per-sample
//     sum is wave shuffle -> LDS -> one double per workgroup -> finalize kernel;
//   * Normal prior log-density  (reference: src/prior/prior.py:30-36 with
//     torch.distributions.Normal.log_prob): sum_x [ -(x-loc)^2/(2 s^2) - log s - log sqrt(2 pi) ].
// Both come with their VJPs (Fitter.step differentiates the action, src/_normflowcore.py:285-288).
#include "nf_internal.h"

namespace nf {

struct EpArgs {
  const void *x;
  const void *loc, *scale;    // prior only; (V) or null
  void *out;                  // VJP: gradient field (B, V)
  const void *gout;           // VJP: cotangent of the per-sample scalar (B)
  double *partial;
  int64_t V;
  int L[4];
  double w0, w2, w4;
  int iters;
};

template <typename T, bool GRAD>
__global__ __launch_bounds__(kBlock) void phi4_kernel(EpArgs A) {
  __shared__ double red[kBlock / kWave];
  const int b = blockIdx.y;
  const T *__restrict__ phi = static_cast<const T *>(A.x) + int64_t(b) * A.V;
  const T w0 = T(A.w0), w2 = T(A.w2), w4 = T(A.w4);
  const int s3 = 1, s2 = A.L[3], s1 = A.L[3] * A.L[2], s0 = A.L[3] * A.L[2] * A.L[1];
  double acc = 0.0;
  const int64_t base = int64_t(blockIdx.x) * kBlock * A.iters + threadIdx.x;
  // coordinates of this lane's first site (one decomposition per lane), then advanced by the
  // mixed-radix digits of the 256-site stride with carries: no division per site
  int x3, x2, x1, x0;
  {
    int r = int(base < A.V ? base : 0);
    x3 = r % A.L[3]; r /= A.L[3];
    x2 = r % A.L[2]; r /= A.L[2];
    x1 = r % A.L[1];
    x0 = r / A.L[1];
  }
  int d3, d2, d1, d0;
  {
    int r = kBlock;
    d3 = r % A.L[3]; r /= A.L[3];
    d2 = r % A.L[2]; r /= A.L[2];
    d1 = r % A.L[1];
    d0 = r / A.L[1];
  }
  for (int it = 0; it < A.iters; ++it) {
    const int64_t i = base + int64_t(it) * kBlock;
    if (i >= A.V) break;
    if (it) {
      x3 += d3; if (x3 >= A.L[3]) { x3 -= A.L[3]; ++x2; }
      x2 += d2; if (x2 >= A.L[2]) { x2 -= A.L[2]; ++x1; }
      x1 += d1; if (x1 >= A.L[1]) { x1 -= A.L[1]; ++x0; }
      x0 += d0;
    }
    const T p = phi[i];
    // backward neighbours: roll(cfgs, 1, mu)[x] = cfgs[x - mu]
    const T n0 = A.L[0] > 1 ? phi[i + (x0 ? -s0 : s0 * (A.L[0] - 1))] : T(0);
    const T n1 = A.L[1] > 1 ? phi[i + (x1 ? -s1 : s1 * (A.L[1] - 1))] : T(0);
    const T n2 = A.L[2] > 1 ? phi[i + (x2 ? -s2 : s2 * (A.L[2] - 1))] : T(0);
    const T n3 = A.L[3] > 1 ? phi[i + (x3 ? -s3 : s3 * (A.L[3] - 1))] : T(0);
    if (!GRAD) {
      const T p2 = p * p;
      acc += double((w2 + w4 * p2) * p2 - w0 * p * (n0 + n1 + n2 + n3));
    } else {
      const T f0 = A.L[0] > 1 ? phi[i + (x0 + 1 < A.L[0] ? s0 : -s0 * (A.L[0] - 1))] : T(0);
      const T f1 = A.L[1] > 1 ? phi[i + (x1 + 1 < A.L[1] ? s1 : -s1 * (A.L[1] - 1))] : T(0);
      const T f2 = A.L[2] > 1 ? phi[i + (x2 + 1 < A.L[2] ? s2 : -s2 * (A.L[2] - 1))] : T(0);
      const T f3 = A.L[3] > 1 ? phi[i + (x3 + 1 < A.L[3] ? s3 : -s3 * (A.L[3] - 1))] : T(0);
      const T g = static_cast<const T *>(A.gout)[b];
      static_cast<T *>(A.out)[int64_t(b) * A.V + i] =
          g * (T(2) * w2 * p + T(4) * w4 * p * p * p - w0 * (n0 + n1 + n2 + n3 + f0 + f1 + f2 + f3));
    }
  }
  if (!GRAD) {
    const double tot = block_sum(acc, red);
    if (threadIdx.x == 0) A.partial[int64_t(b) * gridDim.x + blockIdx.x] = tot;
  }
}

template <typename T, bool GRAD>
__global__ __launch_bounds__(kBlock) void normal_kernel(EpArgs A) {
  __shared__ double red[kBlock / kWave];
  const int b = blockIdx.y;
  const T *__restrict__ x = static_cast<const T *>(A.x) + int64_t(b) * A.V;
  const T *loc = static_cast<const T *>(A.loc), *sc = static_cast<const T *>(A.scale);
  const T klog = T(0.91893853320467274178);   // log sqrt(2 pi)
  double acc = 0.0;
  const int64_t base = int64_t(blockIdx.x) * kBlock * A.iters + threadIdx.x;
  for (int it = 0; it < A.iters; ++it) {
    const int64_t i = base + int64_t(it) * kBlock;
    if (i >= A.V) break;
    const T s = sc ? sc[i] : T(1);
    const T z = (x[i] - (loc ? loc[i] : T(0))) / s;
    if (!GRAD) acc += double(T(-0.5) * z * z - (sc ? nf_log(s) : T(0)) - klog);
    else static_cast<T *>(A.out)[int64_t(b) * A.V + i] = -static_cast<const T *>(A.gout)[b] * z / s;
  }
  if (!GRAD) {
    const double tot = block_sum(acc, red);
    if (threadIdx.x == 0) A.partial[int64_t(b) * gridDim.x + blockIdx.x] = tot;
  }
}

template <typename T, bool PHI4, bool GRAD>
static int run(EpArgs &A, void *out_b, int64_t B, void *ws, size_t ws_bytes, hipStream_t stream) {
  if (B == 0) return NF_OK;
  const Tiling t = make_tiling(A.V, B);
  A.iters = t.iters;
  if (!GRAD) {
    const size_t need = size_t(B) * size_t(t.blocks_x > 0 ? t.blocks_x : 1) * sizeof(double);
    if (ws == nullptr || ws_bytes < need) {
      set_error("nf endpoint kernel: workspace %zu B < %zu B needed", ws_bytes, need);
      return NF_EWORKSPACE;
    }
    A.partial = static_cast<double *>(ws);
  }
  if (t.blocks_x > 0) {
    const dim3 grid(unsigned(t.blocks_x), unsigned(B));
    if (PHI4) hipLaunchKernelGGL((phi4_kernel<T, GRAD>), grid, dim3(kBlock), 0, stream, A);
    else hipLaunchKernelGGL((normal_kernel<T, GRAD>), grid, dim3(kBlock), 0, stream, A);
    const int rc = check_launch("endpoint kernel");
    if (rc) return rc;
  }
  return GRAD ? NF_OK : launch_finalize<T>(A.partial, t.blocks_x, nullptr, out_b, B, stream);
}


// ---------------------------------------------------------------------------------------------------------------
// NormalPrior.sample_ (reference: src/prior/prior.py:26-29 + :30-36 = torch.distributions.Normal.sample, then a second
// pass for log_prob and its per-sample sum): ONE kernel that draws x = loc + scale z and accumulates
// logr[b] = sum_x [-z^2/2 - log scale - log sqrt(2 pi)] from the z it has in registers -- the field is written once and
// never read back.  Generator: Philox4x32-10 (counter-based: any lane computes its numbers from (seed, offset, index),
// no state in memory); layout of the draws: include/normflow_hip.h, restated by oracle/nf_oracle.py::normal_prior_sample.
struct SampleArgs {
  void *x;
  const void *loc, *scale;
  double *partial;
  int64_t V, ngroups;
  uint32_t k0, k1, o0, o1;
  int iters;
};

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void normal_sample_kernel(SampleArgs A) {
  __shared__ double red[kBlock / kWave];
  constexpr int PER = sizeof(T) == 4 ? 4 : 2;
  const int b = blockIdx.y;
  T *__restrict__ xo = static_cast<T *>(A.x) + int64_t(b) * A.V;
  const T *loc = static_cast<const T *>(A.loc), *sc = static_cast<const T *>(A.scale);
  const bool vec = (A.V % PER) == 0;               // then every sample starts 16-byte aligned
  double acc = 0.0;
  const int64_t base = int64_t(blockIdx.x) * kBlock * A.iters + threadIdx.x;
  for (int it = 0; it < A.iters; ++it) {
    const int64_t q = base + int64_t(it) * kBlock;
    if (q >= A.ngroups) break;
    const uint64_t g = uint64_t(b) * uint64_t(A.ngroups) + uint64_t(q);
    uint32_t c[4] = {uint32_t(g), uint32_t(g >> 32), A.o0, A.o1};
    philox4x32_10(c, A.k0, A.k1);
    T z[PER];
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float u1 = (float(c[2 * h] >> 8) + (float(c[2 * h] & 255u) + 1.0f) * 0.00390625f) * 5.9604644775390625e-08f;   // (r + 1) 2^-32, no rounding to 0 or above 1
        const float u2 = float(c[2 * h + 1]) * 2.3283064365386963e-10f;
        const float rho = __builtin_sqrtf(-2.0f * logf(u1 > 1.0f ? 1.0f : u1));
        float sn, cs;
        sincospif(2.0f * u2, &sn, &cs);        // exact range reduction (the argument is in half-turns); the kernel stays near its HBM floor
        z[2 * h] = rho * cs;
        z[2 * h + 1] = rho * sn;
      }
    } else {
      const uint64_t a = (uint64_t(c[0]) << 21) ^ (uint64_t(c[1]) >> 11);
      const uint64_t d = (uint64_t(c[2]) << 21) ^ (uint64_t(c[3]) >> 11);
      const double u1 = (double(a) + 1.0) * 1.1102230246251565e-16, u2 = double(d) * 1.1102230246251565e-16;
      const double rho = ::sqrt(-2.0 * ::log(u1));
      double sn, cs;
      ::sincos(6.283185307179586 * u2, &sn, &cs);
      z[0] = rho * cs;
      z[1] = rho * sn;
    }
    const int64_t i0 = q * PER;
    T v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int64_t i = i0 + j;
      const bool in = i < A.V;
      const T s = (sc && in) ? sc[i] : T(1);
      v[j] = ((loc && in) ? loc[i] : T(0)) + s * z[j];
      if (in) acc += double(T(-0.5) * z[j] * z[j] - (sc ? nf_log(s) : T(0)));
    }
    if (vec) {
      if constexpr (sizeof(T) == 4) *reinterpret_cast<float4 *>(xo + i0) = float4{v[0], v[1], v[2], v[3]};
      else *reinterpret_cast<double2 *>(xo + i0) = double2{v[0], v[1]};
    } else {
#pragma unroll
      for (int j = 0; j < PER; ++j)
        if (i0 + j < A.V) xo[i0 + j] = v[j];
    }
  }
  const double tot = block_sum(acc, red);
  if (threadIdx.x == 0) A.partial[int64_t(b) * gridDim.x + blockIdx.x] = tot - (blockIdx.x == 0 ? 0.91893853320467274178 * double(A.V) : 0.0);
}

template <typename T>
static int run_sample(SampleArgs &A, void *logr, int64_t B, void *ws, size_t ws_bytes, hipStream_t stream) {
  if (B == 0) return NF_OK;
  constexpr int PER = sizeof(T) == 4 ? 4 : 2;
  A.ngroups = (A.V + PER - 1) / PER;
  const Tiling t = make_tiling(A.ngroups > 0 ? A.ngroups : 1, B);
  A.iters = t.iters;
  const size_t need = size_t(B) * size_t(t.blocks_x) * sizeof(double);
  if (ws == nullptr || ws_bytes < need) {
    set_error("nf_normal_sample: workspace %zu B < %zu B needed", ws_bytes, need);
    return NF_EWORKSPACE;
  }
  A.partial = static_cast<double *>(ws);
  hipLaunchKernelGGL((normal_sample_kernel<T>), dim3(unsigned(t.blocks_x), unsigned(B)), dim3(kBlock), 0, stream, A);
  const int rc = check_launch("normal sample kernel");
  if (rc) return rc;
  return launch_finalize<T>(A.partial, t.blocks_x, nullptr, logr, B, stream);
}

static int fill_lattice(EpArgs &A, const int32_t *lattice) {
  NF_REQUIRE(lattice != nullptr, "nf_phi4: lattice is NULL");
  A.V = 1;
  for (int mu = 0; mu < 4; ++mu) {
    NF_REQUIRE(lattice[mu] >= 1, "nf_phi4: lattice extents must be >= 1");
    A.L[mu] = lattice[mu];
    A.V *= lattice[mu];
  }
  NF_REQUIRE(A.V < (int64_t(1) << 31), "nf_phi4: lattice volume must be < 2^31");
  return NF_OK;
}

}  // namespace nf

using namespace nf;

extern "C" int nf_phi4_action(const void *cfgs, void *action, int64_t B, const int32_t *lattice, double w0,
                              double w2, double w4, void *workspace, size_t workspace_bytes, int dtype,
                              void *stream) {
  EpArgs A{};
  int rc = fill_lattice(A, lattice);
  if (rc) return rc;
  NF_REQUIRE(cfgs && action && B >= 0 && B <= 65535, "nf_phi4_action: bad arguments");
  A.x = cfgs; A.w0 = w0; A.w2 = w2; A.w4 = w4;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == NF_F32) return run<float, true, false>(A, action, B, workspace, workspace_bytes, s);
  if (dtype == NF_F64) return run<double, true, false>(A, action, B, workspace, workspace_bytes, s);
  set_error("nf_phi4_action: unsupported dtype %d", dtype);
  return NF_EINVAL;
}

extern "C" int nf_phi4_action_vjp(const void *cfgs, const void *grad_action, void *grad_cfgs, int64_t B,
                                  const int32_t *lattice, double w0, double w2, double w4, int dtype,
                                  void *stream) {
  EpArgs A{};
  int rc = fill_lattice(A, lattice);
  if (rc) return rc;
  NF_REQUIRE(cfgs && grad_action && grad_cfgs && B >= 0 && B <= 65535, "nf_phi4_action_vjp: bad arguments");
  A.x = cfgs; A.gout = grad_action; A.out = grad_cfgs; A.w0 = w0; A.w2 = w2; A.w4 = w4;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == NF_F32) return run<float, true, true>(A, nullptr, B, nullptr, 0, s);
  if (dtype == NF_F64) return run<double, true, true>(A, nullptr, B, nullptr, 0, s);
  set_error("nf_phi4_action_vjp: unsupported dtype %d", dtype);
  return NF_EINVAL;
}

extern "C" int nf_normal_logprob(const void *x, const void *loc, const void *scale, void *logp, int64_t B,
                                 int64_t V, void *workspace, size_t workspace_bytes, int dtype, void *stream) {
  EpArgs A{};
  NF_REQUIRE(x && logp && B >= 0 && B <= 65535 && V >= 0, "nf_normal_logprob: bad arguments");
  A.x = x; A.loc = loc; A.scale = scale; A.V = V;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == NF_F32) return run<float, false, false>(A, logp, B, workspace, workspace_bytes, s);
  if (dtype == NF_F64) return run<double, false, false>(A, logp, B, workspace, workspace_bytes, s);
  set_error("nf_normal_logprob: unsupported dtype %d", dtype);
  return NF_EINVAL;
}

extern "C" int nf_normal_logprob_vjp(const void *x, const void *loc, const void *scale, const void *grad_logp,
                                     void *grad_x, int64_t B, int64_t V, int dtype, void *stream) {
  EpArgs A{};
  NF_REQUIRE(x && grad_logp && grad_x && B >= 0 && B <= 65535 && V >= 0, "nf_normal_logprob_vjp: bad arguments");
  A.x = x; A.loc = loc; A.scale = scale; A.gout = grad_logp; A.out = grad_x; A.V = V;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == NF_F32) return run<float, false, true>(A, nullptr, B, nullptr, 0, s);
  if (dtype == NF_F64) return run<double, false, true>(A, nullptr, B, nullptr, 0, s);
  set_error("nf_normal_logprob_vjp: unsupported dtype %d", dtype);
  return NF_EINVAL;
}

extern "C" int nf_normal_sample(void *x, void *logr, const void *loc, const void *scale, int64_t B, int64_t V,
                                uint64_t seed, uint64_t offset, void *workspace, size_t workspace_bytes, int dtype,
                                void *stream) {
  NF_REQUIRE(x && logr && B >= 0 && B <= 65535 && V >= 0, "nf_normal_sample: bad arguments");
  SampleArgs A{};
  A.x = x; A.loc = loc; A.scale = scale; A.V = V;
  // key = torch's seed with a fixed constant folded into its high word: torch's own Philox kernels key on the bare seed
  // with the counter words transposed ((offset, subsequence) against this kernel's (group, offset)), so without the
  // constant one of their threads could replay the raw words of one of this kernel's groups (same 128-bit space)
  A.k0 = uint32_t(seed); A.k1 = uint32_t(seed >> 32) ^ NF_PHILOX_KEY_DOMAIN; A.o0 = uint32_t(offset); A.o1 = uint32_t(offset >> 32);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == NF_F32) return run_sample<float>(A, logr, B, workspace, workspace_bytes, s);
  if (dtype == NF_F64) return run_sample<double>(A, logr, B, workspace, workspace_bytes, s);
  set_error("nf_normal_sample: unsupported dtype %d", dtype);
  return NF_EINVAL;
}
