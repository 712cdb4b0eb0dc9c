// K_heal -- dropout healer kernels on a device-resident, FRAME-MAJOR spectrogram (what K_stft writes).
//
// Semantics (reference dropout_healer_gui.py):
//   k_inpaint_gain : Canvas.resample_files :135-159 -- per marker: mean dB of the `fs` frames before and
//                    after the box per bin, linear fill across the box (RegularGridInterpolator on a
//                    2-frame grid evaluated at linspace(frame_b, frame_a, nf)), gain = target - dB,
//                    np.clip(gain, previous_gain, 255).
//   k_band_mean_db : Canvas.on_mouse_release :195-203 -- to_dB(magnitude) averaged over a band of bins,
//                    one value per frame (the volume curve the valley detector runs on).
// dB arithmetic is float64 like the reference (util/units.py:24-25 on a float64/complex128 array).
//
// The reference applies markers one after the other; since the mask starts at 0 and every update is
// min(max(g, previous), 255), the final mask is min(255, max(0, max_k g_k)) -- independent of the order.
// That is what lets all markers of a batch run concurrently with an atomic max on the (non-negative)
// float mask.
#include "par_common.h"
#include <math.h>

namespace par {

constexpr double kMagEps = .0000001;      // to_mag: abs + 1e-7 (util/fourier.py:23-24)

__device__ __forceinline__ double spec_db(float2 z) {
  return 20.0 * log10(hypot((double)z.x, (double)z.y) + kMagEps);
}

// markers: int32 [n][5] = frame_b, frame_a, fs (surrounding frames), bin_l, bin_u
// A marker whose box or surrounding frames leave the spectrogram is skipped here (the host mirror refuses it
// before the launch; the reference would index with a negative slice start and average an empty slice).
__global__ void __launch_bounds__(256) k_inpaint_gain(const float2* __restrict__ spec, int64_t n_frames, int64_t bins,
                                                      const int32_t* __restrict__ markers, float* __restrict__ gain) {
  const int32_t* mk = markers + (int64_t)blockIdx.x * 5;
  const int64_t frame_b = mk[0], frame_a = mk[1], fs = mk[2];
  const int bin_l = mk[3], bin_u = mk[4];
  const int64_t nf = frame_a - frame_b;
  if (fs < 1 || nf < 1 || frame_b - fs < 0 || frame_a + fs > n_frames || bin_l < 0 || bin_u > bins) return;
  for (int b = bin_l + (int)threadIdx.x; b < bin_u; b += (int)blockDim.x) {
    double before = 0.0, after = 0.0;
    for (int64_t f = frame_b - fs; f < frame_b; ++f) before += spec_db(spec[f * bins + b]);
    for (int64_t f = frame_a; f < frame_a + fs; ++f) after += spec_db(spec[f * bins + b]);
    before /= (double)fs;
    after /= (double)fs;
    for (int64_t i = 0; i < nf; ++i) {
      // x_i = linspace(frame_b, frame_a, nf)[i]; normalised distance on the 2-point frame grid
      const double t = nf > 1 ? (double)i / (double)(nf - 1) : 0.0;
      const double target = before * (1.0 - t) + after * t;
      double g = target - spec_db(spec[(frame_b + i) * bins + b]);
      g = g < 255.0 ? g : 255.0;
      const float gf = (float)g;
      if (gf > 0.0f) atomicMax(reinterpret_cast<int*>(gain + (frame_b + i) * bins + b), __float_as_int(gf));
    }
  }
}

// Apply-and-clear over the marker boxes only: spec *= 10^(gain/20) (util/units.py:28-29 to_fac) for every masked
// bin, taking the mask value with an atomic exchange so that a bin inside several overlapping boxes is scaled
// exactly once; the mask is all zeros again afterwards (no dense zero-fill / dense apply pass per file).
__global__ void __launch_bounds__(256) k_apply_gain_boxes(float2* __restrict__ spec, int64_t n_frames, int64_t bins,
                                                          const int32_t* __restrict__ markers, float* __restrict__ gain) {
  const int32_t* mk = markers + (int64_t)blockIdx.x * 5;
  const int64_t frame_b = mk[0], frame_a = mk[1], fs = mk[2];
  const int bin_l = mk[3], bin_u = mk[4];
  if (fs < 1 || frame_a - frame_b < 1 || frame_b - fs < 0 || frame_a + fs > n_frames || bin_l < 0 || bin_u > bins) return;
  const int nb = bin_u - bin_l;
  if (nb < 1) return;
  const int64_t total = (frame_a - frame_b) * nb;
  for (int64_t e = threadIdx.x; e < total; e += blockDim.x) {
    const int64_t idx = (frame_b + e / nb) * bins + bin_l + (e % nb);
    const float g = __int_as_float(atomicExch(reinterpret_cast<int*>(gain + idx), 0));
    if (g == 0.0f) continue;
    const float f = exp2f(g * 0.16609640474436813f);     // 10^(g/20) = 2^(g*log2(10)/20)
    float2 v = spec[idx];
    v.x *= f;
    v.y *= f;
    spec[idx] = v;
  }
}

// one wave per frame, lanes stride over the band; mag is the float32 magnitude (already + 1e-7)
__global__ void __launch_bounds__(256) k_band_mean_db(const float* __restrict__ mag, int64_t bins /* row pitch */, int bin_l, int bin_u,
                                                      int64_t frame_b, int64_t count, double* __restrict__ out) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t w = (int64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
  if (w >= count) return;
  const float* row = mag + (frame_b + w) * bins;
  double acc = 0.0;
  for (int b = bin_l + lane; b < bin_u; b += kWave) acc += 20.0 * log10((double)row[b]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
  if (lane == 0) out[w] = acc / (double)(bin_u - bin_l);
}

}  // namespace par

// ---- sparse healing (r03): only the frames a marker can influence go through STFT -> inpaint -> ISTFT ------------------
// Outside the boxes (plus their `surrounding` frames and one window of overlap on either side) STFT -> ISTFT is the
// identity to 1.6e-8, so the healed signal equals the input there.  The host cuts the padded signal into the
// sample ranges ("segments") whose frames matter, lines them up in ONE short pseudo-signal on the original frame grid,
// runs the dense kernels on that, and copies the valid interior of every segment back over a copy of the input.
// One kernel serves both directions: dst[dst_start[k] + i] = src(src_start[k] + i), i < len[k]; with `padded` the source
// is the reference's fix_length(signal, n + n_fft/2) (zeros behind sample n_valid) under np.pad(.., n_fft/2, 'reflect').
__device__ __forceinline__ long long heal_reflect(long long q, long long n) {
  if (n == 1) return 0;
  const long long P = 2 * (n - 1);
  q %= P;
  if (q < 0) q += P;
  return q < n ? q : P - q;
}
__global__ void __launch_bounds__(256) k_copy_segments(const float* __restrict__ src, int64_t src_stride, int64_t n_valid,
                                                       int64_t n_padded, int padded, const int64_t* __restrict__ src_start,
                                                       const int64_t* __restrict__ dst_start, const int64_t* __restrict__ len,
                                                       const int64_t* __restrict__ run_start, int64_t n_seg, int64_t total,
                                                       float* __restrict__ dst, int64_t dst_stride) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // e-th copied sample over all segments
  if (e >= total) return;
  // the block's first sample finds its segment by bisection once (wave-uniform: scalar loads); a thread then walks on from
  // there -- segments are thousands of samples long, so that is 0 or 1 step
  const int64_t e0 = (int64_t)blockIdx.x * blockDim.x;
  int64_t lo = 0, hi = n_seg - 1;                                           // last k with run_start[k] <= e0
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) >> 1;
    if (run_start[mid] <= e0) lo = mid; else hi = mid - 1;
  }
  while (lo + 1 < n_seg && run_start[lo + 1] <= e) ++lo;
  const int64_t i = e - run_start[lo];
  if (i >= len[lo]) return;
  long long q = src_start[lo] + i;
  float v;
  if (padded) {
    if (q < 0 || q >= n_padded) q = heal_reflect(q, n_padded);          // only at the two ends of the file
    v = q < n_valid ? src[q * src_stride] : 0.0f;
  } else {
    v = src[q * src_stride];
  }
  dst[(dst_start[lo] + i) * dst_stride] = v;
}

extern "C" int par_copy_segments_f32(int device, const float* src, int64_t src_stride, int64_t n_valid, int64_t n_padded,
                                     int padded, const int64_t* src_start, const int64_t* dst_start, const int64_t* len,
                                     const int64_t* run_start, int64_t n_seg, int64_t total, float* dst, int64_t dst_stride,
                                     void* stream) {
  using namespace par;
  PAR_REQUIRE(src && dst && src_start && dst_start && len && run_start, PAR_ERR_ARG, "par_copy_segments_f32: null pointer");
  PAR_REQUIRE(n_seg >= 0 && total >= 0 && src_stride >= 1 && dst_stride >= 1 && (!padded || (n_padded >= 1 && n_valid >= 0)),
              PAR_ERR_ARG, "par_copy_segments_f32: bad sizes");
  if (n_seg == 0 || total == 0) return PAR_OK;
  PAR_HIP_CHECK(hipSetDevice(device));
  hipLaunchKernelGGL(k_copy_segments, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream), src, src_stride, n_valid,
                     n_padded, padded, src_start, dst_start, len, run_start, n_seg, total, dst, dst_stride);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

extern "C" int par_inpaint_gain_db_c64(int device, const float* spec, int64_t n_frames, int64_t bins,
                                       const int32_t* markers, int64_t n_markers, float* gain_db, void* stream) {
  using namespace par;
  PAR_REQUIRE(spec && markers && gain_db && n_frames > 0 && bins > 0 && n_markers >= 0, PAR_ERR_ARG,
              "par_inpaint_gain_db_c64: bad args");
  if (n_markers == 0) return PAR_OK;
  PAR_REQUIRE(n_markers <= 0x7fffffff, PAR_ERR_ARG, "par_inpaint_gain_db_c64: too many markers");
  PAR_HIP_CHECK(hipSetDevice(device));
  hipLaunchKernelGGL(k_inpaint_gain, dim3((unsigned)n_markers), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float2*>(spec), n_frames, bins, markers, gain_db);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

extern "C" int par_spec_apply_gain_boxes_c64(int device, float* spec, int64_t n_frames, int64_t bins, const int32_t* markers,
                                             int64_t n_markers, float* gain_db, void* stream) {
  using namespace par;
  PAR_REQUIRE(spec && markers && gain_db && n_frames > 0 && bins > 0 && n_markers >= 0, PAR_ERR_ARG,
              "par_spec_apply_gain_boxes_c64: bad args");
  if (n_markers == 0) return PAR_OK;
  PAR_REQUIRE(n_markers <= 0x7fffffff, PAR_ERR_ARG, "par_spec_apply_gain_boxes_c64: too many markers");
  PAR_HIP_CHECK(hipSetDevice(device));
  hipLaunchKernelGGL(k_apply_gain_boxes, dim3((unsigned)n_markers), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<float2*>(spec), n_frames, bins, markers, gain_db);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

extern "C" int par_band_mean_db_f32(int device, const float* mag, int64_t n_frames, int64_t bins, int64_t mag_pitch, int bin_l,
                                    int bin_u, int64_t frame_b, int64_t frame_a, double* out, void* stream) {
  using namespace par;
  PAR_REQUIRE(mag && out, PAR_ERR_ARG, "par_band_mean_db_f32: null pointer");
  PAR_REQUIRE(mag_pitch == 0 || mag_pitch >= bins, PAR_ERR_ARG, "par_band_mean_db_f32: mag_pitch < bins");
  PAR_REQUIRE(0 <= bin_l && bin_l < bin_u && bin_u <= bins, PAR_ERR_ARG, "par_band_mean_db_f32: empty or out-of-range band [%d, %d) of %lld bins",
              bin_l, bin_u, (long long)bins);
  PAR_REQUIRE(0 <= frame_b && frame_b <= frame_a && frame_a <= n_frames, PAR_ERR_ARG,
              "par_band_mean_db_f32: frame range [%lld, %lld) outside %lld frames", (long long)frame_b, (long long)frame_a,
              (long long)n_frames);
  const int64_t count = frame_a - frame_b;
  if (count == 0) return PAR_OK;
  PAR_HIP_CHECK(hipSetDevice(device));
  hipLaunchKernelGGL(k_band_mean_db, dim3((unsigned)ceil_div(count, 4)), dim3(256), 0, as_stream(stream), mag,
                     mag_pitch ? mag_pitch : bins, bin_l, bin_u, frame_b, count, out);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}
