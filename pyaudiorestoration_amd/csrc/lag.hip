// K_lag / K_lerp -- the two np.interp call sites of resampling.run, restated with numpy's operation order: the lag
// curve -> fractional read positions (pytapesynch branch) and the "Linear" resampling mode.
//
// Semantics (reference util/resampling.py:189-206): sample_at = np.interp(arange(num_out), xp, fp) with
// xp = sampletimes, fp = sampletimes - lags; cut at the first sample_at >= len_signal (find_cutoff :265-270);
// clip at 0.  np.interp's arithmetic (numpy compiled_base.c arr_interp) is restated operation by operation:
// slope = (fp[j+1]-fp[j])/(xp[j+1]-xp[j]); y = slope*(x - xp[j]) + fp[j], no fused multiply-add (this file is
// built with -ffp-contract=off), so positions are bit-identical to the reference's.
#include "par_common.h"
#include <math.h>

namespace par {

__device__ __forceinline__ double interp_one(double x, const double* __restrict__ xp, const double* __restrict__ fp, int64_t m) {
  if (x > xp[m - 1]) return fp[m - 1];          // right = fp[-1]
  if (x < xp[0]) return fp[0];                  // left = fp[0]
  // largest j with xp[j] <= x  (binary_search_with_guess)
  int64_t lo = 0, hi = m - 1;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (xp[mid] <= x) lo = mid; else hi = mid;
  }
  int64_t j = (xp[hi] <= x) ? hi : lo;
  if (j == m - 1) return fp[j];
  if (xp[j] == x) return fp[j];
  const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
  double r = slope * (x - xp[j]) + fp[j];
  if (isnan(r)) {
    r = slope * (x - xp[j + 1]) + fp[j + 1];
    if (isnan(r) && fp[j] == fp[j + 1]) r = fp[j];
  }
  return r;
}

__global__ void __launch_bounds__(256) k_lag_pos(const double* __restrict__ xp, const double* __restrict__ fp, int64_t m,
                                                 int64_t num_out, double len_signal, double* __restrict__ pos,
                                                 unsigned long long* __restrict__ cutoff) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  long long first = 0x7fffffffffffffffLL;
  if (i < num_out) {
    const double v = interp_one((double)i, xp, fp, m);
    if (v >= len_signal) first = i;
    pos[i] = v < 0.0 ? 0.0 : v;                 // np.clip(sample_at, 0, None); NaN propagates like numpy
  }
  first = wave_min_ll(first);
  if ((threadIdx.x & (kWave - 1)) == 0 && first != 0x7fffffffffffffffLL)
    atomicMin(cutoff, (unsigned long long)first);
}

__global__ void k_lag_init(unsigned long long* cutoff, unsigned long long v) { *cutoff = v; }

// "Linear" mode: np.interp(sample_at, arange(len_in), signal, left=0, right=0)  (util/resampling.py:229)
__global__ __launch_bounds__(256) void k_lerp(const double* __restrict__ pos, int64_t len_out,
                                               const float* __restrict__ sig, int64_t sig_stride, int64_t len_in,
                                               float* __restrict__ out, int64_t out_stride) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= len_out) return;
  const double p = pos[j];
  float v = 0.0f;
  if (p != p) v = __builtin_nanf("");                 // np.interp passes a NaN abscissa through
  if (p >= 0.0 && p <= (double)(len_in - 1)) {
    long long i = (long long)p;                       // floor, p >= 0
    if (i >= (long long)len_in - 1) {
      v = sig[(len_in - 1) * sig_stride];
    } else {
      const double y0 = (double)sig[i * sig_stride], y1 = (double)sig[(i + 1) * sig_stride];
      // numpy's interp kernel: slope*(x - x0) + y0 with slope = (y1-y0)/(x1-x0), x1-x0 == 1
      v = (float)((y1 - y0) * (p - (double)i) + y0);   // separate roundings: this file is built with -ffp-contract=off
    }
  }
  out[j * out_stride] = v;
}


}  // namespace par

extern "C" int par_lag_to_pos_f64(int device, const double* xp, const double* fp, int64_t m, int64_t num_out,
                                  int64_t len_signal, double* pos, void* work, int64_t* len_out, int* trimmed,
                                  void* stream) {
  using namespace par;
  PAR_REQUIRE(xp && fp && pos && work && len_out, PAR_ERR_ARG, "par_lag_to_pos_f64: null pointer");
  PAR_REQUIRE(m >= 2 && num_out >= 0 && len_signal >= 0, PAR_ERR_ARG, "par_lag_to_pos_f64: need m >= 2 curve points (got %lld)",
              (long long)m);
  *len_out = num_out;
  if (trimmed) *trimmed = 0;
  if (num_out == 0) return PAR_OK;
  PAR_HIP_CHECK(hipSetDevice(device));
  hipStream_t s = as_stream(stream);
  unsigned long long* cut = reinterpret_cast<unsigned long long*>(work);
  hipLaunchKernelGGL(k_lag_init, dim3(1), dim3(1), 0, s, cut, (unsigned long long)num_out);
  hipLaunchKernelGGL(k_lag_pos, dim3((unsigned)ceil_div(num_out, 256)), dim3(256), 0, s, xp, fp, m, num_out,
                     (double)len_signal, pos, cut);
  PAR_HIP_CHECK(hipGetLastError());
  unsigned long long h = 0;
  PAR_HIP_CHECK(hipMemcpyAsync(&h, cut, sizeof(h), hipMemcpyDeviceToHost, s));
  PAR_HIP_CHECK(hipStreamSynchronize(s));
  if ((int64_t)h < num_out) {
    *len_out = (int64_t)h;
    if (trimmed) *trimmed = 1;
  }
  return PAR_OK;
}

extern "C" int par_linear_resample_f32(int device, const double* pos, int64_t len_out, const float* sig, int64_t sig_stride,
                            int64_t len_in, float* out, int64_t out_stride, void* stream) {
  using namespace par;
  PAR_REQUIRE(pos && sig && out && len_out >= 0 && len_in >= 1, PAR_ERR_ARG, "par_linear_resample_f32: bad args");
  if (len_out == 0) return PAR_OK;
  PAR_HIP_CHECK(hipSetDevice(device));
  hipLaunchKernelGGL(k_lerp, dim3((unsigned)ceil_div(len_out, 256)), dim3(256), 0, as_stream(stream), pos, len_out, sig,
                     sig_stride, len_in, out, out_stride);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}
