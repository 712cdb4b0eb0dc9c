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

// log10 of a positive, finite, normal float64 (every magnitude here is >= 1e-7): exponent + 2 atanh((f - 1) / (f + 1)) on
// f in [1/sqrt 2, sqrt 2), nine odd terms; |error| <= 1e-15 against numpy's log10 over 1e-7 .. 1e6 (about a third of the
// instructions of the library routine, which is what k_inpaint_gain spent 60 % of its time in).  Anything else -- NaN, Inf,
// the reference's poison values -- goes to the library.
__device__ __forceinline__ double log10_pos(double t) {
  if (!(t >= 0x1p-1000 && t <= 0x1p1000)) return log10(t);
  const long long bits = __double_as_longlong(t);
  int e = (int)(bits >> 52) - 1023;
  double f = __longlong_as_double((bits & 0x000fffffffffffffll) | 0x3ff0000000000000ll);      // [1, 2)
  if (f > 1.4142135623730951) {
    f *= 0.5;
    e += 1;
  }
  const double num = f - 1.0, den = f + 1.0;                       // den in (1.7, 2.42)
  double r = __builtin_amdgcn_rcp(den);
  r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
  double sq = num * r;
  sq = __builtin_fma(__builtin_fma(-den, sq, num), r, sq);         // (f - 1) / (f + 1) to the last bit or so
  const double z = sq * sq;
  double p = 1.0 / 17.0;
  p = __builtin_fma(p, z, 1.0 / 15.0);
  p = __builtin_fma(p, z, 1.0 / 13.0);
  p = __builtin_fma(p, z, 1.0 / 11.0);
  p = __builtin_fma(p, z, 1.0 / 9.0);
  p = __builtin_fma(p, z, 1.0 / 7.0);
  p = __builtin_fma(p, z, 1.0 / 5.0);
  p = __builtin_fma(p, z, 1.0 / 3.0);
  const double two_s = sq + sq;
  const double lnf = __builtin_fma(two_s * z, p, two_s);
  return __builtin_fma((double)e, 0.30102999566398120, lnf * 0.43429448190325182);
}
__device__ __forceinline__ double spec_db(float2 z) {
  const double x = (double)z.x, y = (double)z.y;
  // |z| as sqrt(x^2 + y^2): both squares are exact in float64 (24-bit operands) and cannot overflow, so this is hypot to an ulp
  return 20.0 * log10_pos(sqrt(x * x + y * y) + kMagEps);
}

// markers: int32 [n][5] = frame_b, frame_a, fs (surrounding frames), bin_l, bin_u
// A marker whose box or surrounding frames leave the spectrogram is skipped here (the host mirror refuses it
// before the launch; the reference would index with a negative slice start and average an empty slice).
// One workgroup per (marker, chunk of kGainBins bins).  A band of nb bins leaves 256 / nb "frame lanes" per bin: thread
// (bin, lane) takes every P-th frame, so a 99-bin band keeps 198 of 256 lanes busy (one thread per bin walking all frames
// serially kept 99: the float64 log10 / hypot of util/units.py is what this kernel spends its time on).  Partial sums meet in
// LDS in a fixed order: the result does not depend on the launch.
constexpr int kGainBins = 256;
__global__ void __launch_bounds__(256) k_inpaint_gain(const float2* __restrict__ spec, int64_t n_frames, int64_t bins,
                                                      const int32_t* __restrict__ markers, float* __restrict__ gain) {
  __shared__ double s_before[256], s_after[256];
  const int32_t* mk = markers + (int64_t)blockIdx.x * 5;
  const int64_t frame_b = mk[0], frame_a = mk[1], fs = mk[2];
  const int bin_l = mk[3], bin_u = mk[4];
  const int64_t nf = frame_a - frame_b;
  if (fs < 1 || nf < 1 || frame_b - fs < 0 || frame_a + fs > n_frames || bin_l < 0 || bin_u > bins) return;
  const int b0 = bin_l + (int)blockIdx.y * kGainBins;
  if (b0 >= bin_u) return;                                   // workgroup-uniform, like everything above
  const int nbc = bin_u - b0 < kGainBins ? bin_u - b0 : kGainBins;
  const int P = 256 / nbc;
  const int t = (int)threadIdx.x, bi = t % nbc, fl = t / nbc;
  const bool act = fl < P;
  const float2* col = spec + b0 + bi;
  double before = 0.0, after = 0.0;
  if (act) {
    for (int64_t f = frame_b - fs + fl; f < frame_b; f += P) before += spec_db(col[f * bins]);
    for (int64_t f = frame_a + fl; f < frame_a + fs; f += P) after += spec_db(col[f * bins]);
  }
  s_before[t] = before;
  s_after[t] = after;
  __syncthreads();
  if (!act) return;
  before = after = 0.0;
  for (int p = 0; p < P; ++p) {
    before += s_before[bi + p * nbc];
    after += s_after[bi + p * nbc];
  }
  before /= (double)fs;
  after /= (double)fs;
  for (int64_t i = fl; i < nf; i += P) {
    // x_i = linspace(frame_b, frame_a, nf)[i]; normalised distance on the 2-point frame grid
    const double tt = nf > 1 ? (double)i / (double)(nf - 1) : 0.0;
    const double target = before * (1.0 - tt) + after * tt;
    double g = target - spec_db(col[(frame_b + i) * bins]);
    g = g < 255.0 ? g : 255.0;
    const float gf = (float)g;
    if (gf > 0.0f) atomicMax(reinterpret_cast<int*>(gain + (frame_b + i) * bins + b0 + bi), __float_as_int(gf));
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
  // thread (bin, frame lane) as in k_inpaint_gain: no division per element, rows read in runs of nb consecutive bins
  const int P = nb < 256 ? 256 / nb : 1;
  const int t = (int)threadIdx.x, fl = nb < 256 ? t / nb : 0;
  if (fl >= P) return;
  for (int bi = nb < 256 ? t % nb : t; bi < nb; bi += 256)
  for (int64_t f = frame_b + fl; f < frame_a; f += P) {
    const int64_t idx = f * bins + bin_l + bi;
    const float g = __int_as_float(atomicExch(reinterpret_cast<int*>(gain + idx), 0));
    if (g == 0.0f) continue;
    const float fac = exp2f(g * 0.16609640474436813f);     // 10^(g/20) = 2^(g*log2(10)/20)
    float2 v = spec[idx];
    v.x *= fac;
    v.y *= fac;
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
  for (int b = bin_l + lane; b < bin_u; b += kWave) acc += 20.0 * log10_pos((double)row[b]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
  if (lane == 0) out[w] = acc / (double)(bin_u - bin_l);
}


// ---- heuristic dropout repair (dropouts_gui.py:314-321): the two O(n) passes around the band-pass filter ---------------------
// out[c][i] = sig[i][c] * np.interp(np.linspace(0, 1, n)[i], np.linspace(0, 1, frames), fac[c])   (float64, like numpy's product)
__global__ void k_curve_scale(const float* __restrict__ sig, int64_t sig_stride, int64_t n, const double* __restrict__ fac,
                              int64_t frames, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = blockIdx.y;
  const double* fp = fac + (int64_t)c * frames;
  // np.linspace(0, 1, k)[j] = j * (1 / (k - 1)), its last element set to 1 exactly
  const double sx = n > 1 ? 1.0 / (double)(n - 1) : 0.0, sp = frames > 1 ? 1.0 / (double)(frames - 1) : 0.0;
  const double x = i == n - 1 && n > 1 ? 1.0 : (double)i * sx;
  double v;
  if (frames == 1) {
    v = fp[0];
  } else {
    long long j = (long long)(x * (double)(frames - 1));
    j = j < 0 ? 0 : (j > frames - 2 ? frames - 2 : j);
    auto xp = [&](long long q) { return q == frames - 1 ? 1.0 : (double)q * sp; };
    if (xp(j) > x && j > 0) --j;                    // the float product can land one interval off
    else if (xp(j + 1) <= x && j < frames - 2) ++j;
    const double x0 = xp(j), x1 = xp(j + 1);
    v = x >= 1.0 ? fp[frames - 1] : (fp[j + 1] - fp[j]) / (x1 - x0) * (x - x0) + fp[j];      // np.interp's own form
  }
  out[(int64_t)c * n + i] = (double)sig[i * sig_stride + c] * v;
}
// sig[i][c] = float32(float64(sig[i][c]) + y[c][i])       (numpy's  float32_column += float64_array)
__global__ void k_accumulate(float* __restrict__ sig, int64_t sig_stride, int64_t n, const double* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = blockIdx.y;
  float* p = sig + i * sig_stride + c;
  *p = (float)((double)*p + y[(int64_t)c * n + i]);
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
constexpr int kCopySpan = 2048;     // copied samples per workgroup: one bisection of the segment table serves eight rounds
__global__ void __launch_bounds__(256) k_copy_segments(const float* __restrict__ src, int64_t src_stride, int64_t n_valid,
                                                       int64_t n_padded, int padded, const int64_t* __restrict__ src_start,
                                                       const int64_t* __restrict__ dst_start, const int64_t* __restrict__ len,
                                                       const int64_t* __restrict__ run_start, int64_t n_seg, int64_t total,
                                                       float* __restrict__ dst, int64_t dst_stride) {
  // the workgroup's first sample finds its segment by bisection once (uniform: scalar loads -- a dozen dependent loads,
  // which at one bisection per 256 samples was most of the kernel); a thread then walks on from there -- segments are
  // thousands of samples long, so that is a step or two over the whole span
  const int64_t e0 = (int64_t)blockIdx.x * kCopySpan;
  int64_t lo = 0, hi = n_seg - 1;                                           // last k with run_start[k] <= e0
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) >> 1;
    if (run_start[mid] <= e0) lo = mid; else hi = mid - 1;
  }
  int64_t next = lo + 1 < n_seg ? run_start[lo + 1] : INT64_MAX;
  int64_t run = run_start[lo], s0 = src_start[lo], d0 = dst_start[lo], ln = len[lo];
  for (int j = 0; j < kCopySpan / 256; ++j) {
    const int64_t e = e0 + j * 256 + threadIdx.x;                           // e-th copied sample over all segments
    if (e >= total) return;
    if (e >= next) {
      do { ++lo; } while (lo + 1 < n_seg && run_start[lo + 1] <= e);
      next = lo + 1 < n_seg ? run_start[lo + 1] : INT64_MAX;
      run = run_start[lo], s0 = src_start[lo], d0 = dst_start[lo], ln = len[lo];
    }
    const int64_t i = e - run;
    if (i >= ln) continue;
    long long q = s0 + i;
    float v;
    if (padded) {
      if (q < 0 || q >= n_padded) q = heal_reflect(q, n_padded);          // only at the two ends of the file
      v = q < n_valid ? src[q * src_stride] : 0.0f;
    } else {
      v = src[q * src_stride];
    }
    dst[(d0 + i) * dst_stride] = v;
  }
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
  hipLaunchKernelGGL(k_copy_segments, dim3((unsigned)ceil_div(total, (int64_t)kCopySpan)), dim3(256), 0, as_stream(stream), src, src_stride, n_valid,
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
  hipLaunchKernelGGL(k_inpaint_gain, dim3((unsigned)n_markers, (unsigned)ceil_div(bins, (int64_t)kGainBins)), dim3(256), 0, as_stream(stream),
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

extern "C" int par_curve_scale_f64(int device, const float* sig, int64_t sig_stride, int n_ch, int64_t n, const double* fac,
                                   int64_t frames, double* out, void* stream) {
  using namespace par;
  PAR_REQUIRE(sig && fac && out && n >= 1 && frames >= 1 && n_ch >= 1 && n_ch <= 65535 && sig_stride >= n_ch, PAR_ERR_ARG,
              "par_curve_scale_f64: bad args");
  PAR_HIP_CHECK(hipSetDevice(device));
  hipLaunchKernelGGL(k_curve_scale, dim3((unsigned)ceil_div(n, 256), (unsigned)n_ch), dim3(256), 0, as_stream(stream), sig, sig_stride, n,
                     fac, frames, out);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

extern "C" int par_accumulate_f64_f32(int device, float* sig, int64_t sig_stride, int n_ch, int64_t n, const double* y, void* stream) {
  using namespace par;
  PAR_REQUIRE(sig && y && n >= 1 && n_ch >= 1 && n_ch <= 65535 && sig_stride >= n_ch, PAR_ERR_ARG, "par_accumulate_f64_f32: bad args");
  PAR_HIP_CHECK(hipSetDevice(device));
  hipLaunchKernelGGL(k_accumulate, dim3((unsigned)ceil_div(n, 256), (unsigned)n_ch), dim3(256), 0, as_stream(stream), sig, sig_stride, n, y);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}
