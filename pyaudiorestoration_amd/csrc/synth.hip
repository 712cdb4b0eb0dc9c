// Synthetic workload generators so that benchmark inputs are born in HBM.
// Bit-faithful to tests/inputs.py up to the last ulp of float64 sin() before the float32 cast.
#include "par_common.h"
#include <math.h>

namespace par {

__device__ __forceinline__ double splitmix_uniform(uint64_t idx, uint64_t seed) {
  uint64_t z = (idx ^ seed) + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

// x[n] = 0.25 sin(2pi 1000 n/sr) + 0.25 sin(2pi (0.45 sr/2) n/sr) + 0.1 u(n)
__global__ void k_synth_signal(float* __restrict__ out, int64_t start, int64_t count, double sr, uint64_t seed) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const double n = (double)(start + i);
  const double w1 = 2.0 * M_PI * 1000.0, w2 = 2.0 * M_PI * (0.45 * sr / 2.0);
  const double v = 0.25 * sin(w1 * n / sr) + 0.25 * sin(w2 * n / sr) + 0.1 * splitmix_uniform((uint64_t)(start + i), seed);
  out[i] = (float)v;
}

// t = linspace(0, dur, m); sampletimes = t*sr; speed = 1 + depth*sin(2pi rate t + phase)
__global__ void k_synth_curve(double* __restrict__ st, double* __restrict__ sp, int64_t m, double dur, double sr,
                              double depth, double rate, double phase) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const double step = dur / (double)(m - 1);
  const double t = (i == m - 1) ? dur : (double)i * step;
  st[i] = t * sr;
  sp[i] = 1.0 + depth * sin(2.0 * M_PI * rate * t + phase);
}

}  // namespace par

extern "C" {

int par_synth_signal_f32(int device, float* out, int64_t start, int64_t count, double sr, uint64_t seed, void* stream) {
  using namespace par;
  PAR_REQUIRE(out && count >= 0 && sr > 0, PAR_ERR_ARG, "par_synth_signal_f32: bad args");
  if (count == 0) return PAR_OK;
  PAR_HIP_CHECK(hipSetDevice(device));
  hipLaunchKernelGGL(k_synth_signal, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0, as_stream(stream), out, start,
                     count, sr, seed);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

int par_synth_speed_curve_f64(int device, double* sampletimes, double* speeds, int64_t m, double duration_s, double sr,
                              double depth, double rate_hz, double phase, void* stream) {
  using namespace par;
  PAR_REQUIRE(sampletimes && speeds && m >= 2 && sr > 0, PAR_ERR_ARG, "par_synth_speed_curve_f64: bad args");
  PAR_HIP_CHECK(hipSetDevice(device));
  hipLaunchKernelGGL(k_synth_curve, dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, as_stream(stream), sampletimes, speeds,
                     m, duration_s, sr, depth, rate_hz, phase);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

}  // extern "C"
