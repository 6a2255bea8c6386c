// nf_common.hip -- error channel, version, workspace sizing, stage-2 log-det reduction.
#include <cstdarg>
#include <cstdio>
#include "nf_internal.h"

namespace nf {

static thread_local char g_err[512] = "";

// ---- process-wide kernel-selection options (nf_set_option / nf_get_option).  Plain ints written by the host thread that
// configures the library; kernels never read them, only the launch planners do.
static int g_options[NF_OPT_COUNT_] = {1, 1, 1};
int option(int which) { return (which >= 0 && which < NF_OPT_COUNT_) ? g_options[which] : 0; }

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char *what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return NF_ELAUNCH;
  }
  return NF_OK;
}

// One wave per sample: logj[b] = log0[b] + sum_i partial[b, i], summed in double
// in a fixed order (bitwise reproducible).
template <typename T>
__global__ __launch_bounds__(kBlock) void finalize_kernel(const double *__restrict__ partial, int64_t n_part,
                                                          const T *__restrict__ log0, T *__restrict__ logj,
                                                          int64_t B) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t b = int64_t(blockIdx.x) * (kBlock / kWave) + threadIdx.x / kWave;
  if (b >= B) return;
  double acc = 0.0;
  for (int64_t i = lane; i < n_part; i += kWave) acc += partial[b * n_part + i];
  acc = wave_sum(acc);
  if (lane == 0) logj[b] = T((log0 ? double(log0[b]) : 0.0) + acc);
}

template <typename T>
int launch_finalize(const double *partial, int64_t n_part, const void *log0, void *logj, int64_t B,
                    hipStream_t stream) {
  if (B == 0) return NF_OK;
  const int per = kBlock / kWave;
  hipLaunchKernelGGL((finalize_kernel<T>), dim3(unsigned((B + per - 1) / per)), dim3(kBlock), 0, stream,
                     partial, n_part, static_cast<const T *>(log0), static_cast<T *>(logj), B);
  return check_launch("finalize kernel");
}
template int launch_finalize<float>(const double *, int64_t, const void *, void *, int64_t, hipStream_t);
template int launch_finalize<double>(const double *, int64_t, const void *, void *, int64_t, hipStream_t);

}  // namespace nf

extern "C" int nf_version(void) { return NF_VERSION; }
extern "C" int nf_set_option(int which, int value) {
  if (which < 0 || which >= NF_OPT_COUNT_) {
    nf::set_error("nf_set_option: unknown option %d", which);
    return NF_EINVAL;
  }
  const int old = nf::g_options[which];
  nf::g_options[which] = value;
  return old;
}
extern "C" int nf_get_option(int which) { return nf::option(which); }
extern "C" const char *nf_last_error_string(void) { return nf::g_err; }
extern "C" size_t nf_workspace_bytes(int64_t B, int64_t V) {
  if (B < 0 || V < 0) return 0;
  // worst case of make_tiling: one double per 64-unit workgroup per sample (the LDS-column
  // kernel may shrink its workgroup to one wave), plus room for the 3K-per-workgroup
  // knot-cotangent partials of nf_distconv_vjp
  const int64_t blocks = (V + nf::kWave - 1) / nf::kWave + 1;
  return size_t(B) * size_t(blocks) * sizeof(double) + (size_t(8) << 20);
}
