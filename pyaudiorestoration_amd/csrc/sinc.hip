// K_sinc -- per-sample time-varying Hann-windowed sinc interpolation (the north-star kernel).
//
// Semantics: resampling.sinc_core (reference util/resampling.py:51-90) as driven by
// sinc_wrapper (:21-27):  for output i at fractional read position p = sample_at[i]
//     ind = rint(p); shift = p - ind; fc = min(1 / max(1e-12, p[i+1]-p[i]), 1)
//     out[i] = sum_{k=0}^{2NT-1} signal[ind-NT+k] * sinc((k-NT-shift)*fc)*fc * hanning(2NT+1)[k]
// with the window clipped at the signal ends exactly like `signal[lower:upper]` (:71-72, incl. the
// bug-compatible mis-aligned leading edge).
//
// CDNA4 mapping (no MFMA: per-lane transcendental weights, not a shared-operand contraction):
//  * one 256-thread workgroup = TILE consecutive outputs; their input footprint
//    [min(ind)-NT, max(ind)+NT) is staged ONCE into LDS with coalesced loads, so HBM sees each
//    input sample once (+ a 2NT halo per tile): algorithmic 4 B in + 4 B out (+8 B position).
//  * lane = output sample, so the 64 lanes of a wave read 64 consecutive LDS words per tap
//    (conflict-free ds_read_b32).
//  * the weight  sinc(x*fc)*fc*win_k = sin(pi*fc*x) / (pi*x/win_k),  x = k-NT-shift, is evaluated
//    without any per-tap sin: the numerator obeys the 3-term recurrence u[n+1] = 2cos(theta)u[n]-u[n-1]
//    (theta = pi*fc), seeded at the window centre and run outwards in both directions (its error
//    grows ~n while the weight decays ~1/n); when a whole wave has fc == 1 (speed <= 1) the numerator
//    collapses to (-1)^(n+1) sin(pi*shift) and is factored out of the sum.  The denominator is one
//    v_fma_f32 (x folded with pi/win_k from a wave-uniform table held in SGPRs) + one v_rcp_f32.
//  * positions stay float64 end to end (a 345.6 M-sample index does not fit float32); only the
//    sub-sample shift in [-0.5, 0.5] and fc drop to float32.
#include "par_common.h"
#include <math.h>
#include <map>
#include <vector>

namespace par {

constexpr int kSincBlock = 256;
constexpr int kSincR = 4;                         // outputs per thread
constexpr int kSincTile = kSincBlock * kSincR;    // outputs per workgroup
constexpr int kSincCap = 6144;                    // LDS floats for the staged input span (24 KiB)

// sin(pi*x), cos(pi*x) on [-0.5, 0.5]; Taylor in (pi*x), abs error < 1e-7 at the interval ends.
__device__ __forceinline__ float sinpi_half(float x) {
  const float z = x * x;
  float p = -0.00737043094f;               // -pi^11/11!
  p = fmaf(p, z, 0.0821458866f);           //  pi^9/9!
  p = fmaf(p, z, -0.599264529f);           // -pi^7/7!
  p = fmaf(p, z, 2.55016404f);             //  pi^5/5!
  p = fmaf(p, z, -5.16771278f);            // -pi^3/3!
  p = fmaf(p, z, 3.14159265f);             //  pi
  return p * x;
}
__device__ __forceinline__ float cospi_half(float x) {
  const float z = x * x;
  float p = 0.00192957431f;                //  pi^12/12!
  p = fmaf(p, z, -0.0258068914f);          // -pi^10/10!
  p = fmaf(p, z, 0.235330630f);            //  pi^8/8!
  p = fmaf(p, z, -1.33526277f);            // -pi^6/6!
  p = fmaf(p, z, 4.05871213f);             //  pi^4/4!
  p = fmaf(p, z, -4.93480220f);            // -pi^2/2!
  p = fmaf(p, z, 1.0f);
  return p;
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Fully general float64 evaluation of ONE output straight from global memory.  Used for the
// leading-edge outputs (ind < NT), for tiles whose input span does not fit LDS, and as the
// in-library cross-check of the fast path.  Follows util/resampling.py:66-90 line by line.
__device__ float sinc_one_f64(double p, double dp, const float* __restrict__ sig, int64_t sig_stride,
                              int64_t len_in, int NT) {
  const long long ind = llrint(p);
  const long long lower = ind - NT > 0 ? ind - NT : 0;
  const long long upper = ind + NT < (long long)len_in ? ind + NT : (long long)len_in;
  const long long L = upper - lower;
  if (L <= 0) return 0.0f;
  const double period = dp > 1e-12 ? dp : 1e-12;
  const double inv = 1.0 / period;
  const double fc = inv < 1.0 ? inv : 1.0;
  const double shift = p - (double)ind;
  double acc = 0.0;
  for (long long k = 0; k < L; ++k) {
    double x = ((double)(k - NT) - shift) * fc;
    double y = M_PI * (x == 0.0 ? 1e-20 : x);           // np.sinc
    double si = sin(y) / y * fc;
    float win = (float)(0.5 + 0.5 * cos(M_PI * (double)(k - NT) / (double)NT));   // np.hanning(2NT+1)[k] as f32
    acc += (double)sig[(lower + k) * sig_stride] * si * (double)win;
  }
  return (float)acc;
}

// fc == 1 for every lane of the wave: numerator (-1)^(n+1) sin(pi*s) factored out.
template <int R>
__device__ __forceinline__ void taps_unity(const float* __restrict__ tile, const int (&c)[R], const float (&s)[R],
                                           int NT, const float* __restrict__ tabA, const float* __restrict__ tabB,
                                           float (&res)[R]) {
  float accE[R], accO[R];
  const float b0 = tabB[0];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    accE[r] = tile[c[r]] * fast_rcp(s[r] * b0);
    accO[r] = 0.0f;
  }
  int n = 1;
  for (; n + 1 < NT; n += 2) {
    const float a1 = tabA[n], b1 = tabB[n], a2 = tabA[n + 1], b2 = tabB[n + 1];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float* t = tile + c[r];
      accO[r] = fmaf(t[n], fast_rcp(fmaf(s[r], b1, a1)), accO[r]);
      accO[r] = fmaf(t[-n], fast_rcp(fmaf(s[r], b1, -a1)), accO[r]);
      accE[r] = fmaf(t[n + 1], fast_rcp(fmaf(s[r], b2, a2)), accE[r]);
      accE[r] = fmaf(t[-n - 1], fast_rcp(fmaf(s[r], b2, -a2)), accE[r]);
    }
  }
  if (n < NT) {   // n is odd here
    const float a1 = tabA[n], b1 = tabB[n];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float* t = tile + c[r];
      accO[r] = fmaf(t[n], fast_rcp(fmaf(s[r], b1, a1)), accO[r]);
      accO[r] = fmaf(t[-n], fast_rcp(fmaf(s[r], b1, -a1)), accO[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) res[r] = -sinpi_half(s[r]) * (accE[r] - accO[r]);
}

// general fc in (0, 1]: numerator by the two-sided 3-term recurrence seeded at the centre.
template <int R>
__device__ __forceinline__ void taps_general(const float* __restrict__ tile, const int (&c)[R], const float (&s)[R],
                                             const float (&fc)[R], const float (&dd)[R], int NT,
                                             const float* __restrict__ tabA, const float* __restrict__ tabB,
                                             float (&res)[R]) {
  float acc[R], up[R], upp[R], um[R], umm[R], c2[R];
  const float b0 = tabB[0];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float h = fc[r] * s[r];                    // phi / pi, |h| <= 0.5
    const float sphi = sinpi_half(h), cphi = cospi_half(h);
    float sth, cth;                                   // sin/cos(theta), theta = pi*fc = pi - pi*dd
    if (dd[r] <= 0.5f) {
      sth = sinpi_half(dd[r]);
      cth = -cospi_half(dd[r]);
    } else {
      sth = sinpi_half(fc[r]);
      cth = cospi_half(fc[r]);
    }
    const float u0 = -sphi;
    up[r] = fmaf(sth, cphi, -cth * sphi);             // u[+1]
    um[r] = -fmaf(sth, cphi, cth * sphi);             // u[-1]
    upp[r] = u0;
    umm[r] = u0;
    c2[r] = 2.0f * cth;
    acc[r] = tile[c[r]] * (u0 * fast_rcp(s[r] * b0));
  }
  for (int n = 1; n < NT; ++n) {
    const float a = tabA[n], b = tabB[n];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float* t = tile + c[r];
      acc[r] = fmaf(t[n], up[r] * fast_rcp(fmaf(s[r], b, a)), acc[r]);
      acc[r] = fmaf(t[-n], um[r] * fast_rcp(fmaf(s[r], b, -a)), acc[r]);
      const float un = fmaf(c2[r], up[r], -upp[r]);
      upp[r] = up[r];
      up[r] = un;
      const float vn = fmaf(c2[r], um[r], -umm[r]);
      umm[r] = um[r];
      um[r] = vn;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) res[r] = acc[r];
}

__global__ __launch_bounds__(kSincBlock) void k_sinc(const double* __restrict__ pos, int64_t len_out,
                                                      const float* __restrict__ sig, int64_t sig_stride,
                                                      int64_t len_in, int NT, const float* __restrict__ tabA,
                                                      const float* __restrict__ tabB, float* __restrict__ out,
                                                      int64_t out_stride) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  __shared__ long long red[2 * (kSincBlock / kWave)];
  const int t = threadIdx.x;
  const int64_t j0 = (int64_t)blockIdx.x * kSincTile;

  double p[kSincR], dp[kSincR];
  long long ind[kSincR];
  bool valid[kSincR];
  long long mn = INT64_MAX, mx = INT64_MIN;
#pragma unroll
  for (int r = 0; r < kSincR; ++r) {
    const int64_t j = j0 + t + (int64_t)r * kSincBlock;
    valid[r] = j < len_out;
    p[r] = 0.0;
    dp[r] = 1.0;
    ind[r] = 0;
    if (valid[r]) {
      p[r] = pos[j];
      // last output reuses the previous period (util/resampling.py:76-77)
      dp[r] = (j + 1 < len_out) ? pos[j + 1] - p[r] : p[r] - pos[j - 1];
      ind[r] = llrint(p[r]);
      mn = ind[r] < mn ? ind[r] : mn;
      mx = ind[r] > mx ? ind[r] : mx;
    }
  }
  mn = wave_min_ll(mn);
  mx = wave_max_ll(mx);
  if ((t & (kWave - 1)) == 0) {
    red[t / kWave] = mn;
    red[kSincBlock / kWave + t / kWave] = mx;
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < kSincBlock / kWave; ++w) {
    mn = red[w] < mn ? red[w] : mn;
    mx = red[kSincBlock / kWave + w] > mx ? red[kSincBlock / kWave + w] : mx;
  }
  const long long lo = mn - NT;
  const long long span = mx + NT - lo;             // <= kSincCap for the LDS path
  const bool staged = span <= kSincCap;
  if (staged) {
    for (long long q = t; q < span; q += kSincBlock) {
      const long long g = lo + q;
      tile[q] = (g >= 0 && g < (long long)len_in) ? sig[g * sig_stride] : 0.0f;
    }
  }
  __syncthreads();

  float res[kSincR];
  int c[kSincR];
  float s[kSincR], fc[kSincR], dd[kSincR];
  bool fastlane[kSincR];
  bool unity = true, anyfast = false;
#pragma unroll
  for (int r = 0; r < kSincR; ++r) {
    // leading-edge outputs (ind < NT) keep the reference's mis-aligned taps: float64 path.
    fastlane[r] = valid[r] && staged && ind[r] >= NT;
    c[r] = fastlane[r] ? (int)(ind[r] - lo) : NT;          // harmless in-range index for idle lanes
    float sh = (float)(p[r] - (double)ind[r]);
    s[r] = (sh == 0.0f) ? 1e-20f : sh;                      // np.sinc's own 0 -> 1e-20 substitution
    const bool one = !(dp[r] > 1.0);                        // fc == 1 (also catches the 1e-12 floor)
    const float dpf = (float)(dp[r] > 1e-12 ? dp[r] : 1e-12);
    const float inv = fast_rcp(dpf);
    fc[r] = one ? 1.0f : inv;
    dd[r] = one ? 0.0f : (float)(dp[r] - 1.0) * inv;        // 1 - fc without cancellation
    unity = unity && (one || !fastlane[r]);
    anyfast = anyfast || fastlane[r];
    res[r] = 0.0f;
  }
  if (__any(anyfast)) {
    if (__all(unity)) taps_unity<kSincR>(tile, c, s, NT, tabA, tabB, res);
    else taps_general<kSincR>(tile, c, s, fc, dd, NT, tabA, tabB, res);
  }
#pragma unroll
  for (int r = 0; r < kSincR; ++r) {
    if (!valid[r]) continue;
    const int64_t j = j0 + t + (int64_t)r * kSincBlock;
    float v = fastlane[r] ? res[r] : sinc_one_f64(p[r], dp[r], sig, sig_stride, len_in, NT);
    out[j * out_stride] = v;
  }
}

#pragma clang fp contract(off)   // numpy's interp kernel is slope*(x-x0)+y0 with separate roundings
// "Linear" mode: np.interp(sample_at, arange(len_in), signal, left=0, right=0)  (util/resampling.py:229)
__global__ __launch_bounds__(256) void k_lerp(const double* __restrict__ pos, int64_t len_out,
                                               const float* __restrict__ sig, int64_t sig_stride, int64_t len_in,
                                               float* __restrict__ out, int64_t out_stride) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= len_out) return;
  const double p = pos[j];
  float v = 0.0f;
  if (p >= 0.0 && p <= (double)(len_in - 1)) {
    long long i = (long long)p;                       // floor, p >= 0
    if (i >= (long long)len_in - 1) {
      v = sig[(len_in - 1) * sig_stride];
    } else {
      const double y0 = (double)sig[i * sig_stride], y1 = (double)sig[(i + 1) * sig_stride];
      // numpy's interp kernel: slope*(x - x0) + y0 with slope = (y1-y0)/(x1-x0), x1-x0 == 1
      v = (float)__dadd_rn(__dmul_rn(y1 - y0, p - (double)i), y0);   // no FMA contraction
    }
  }
  out[j * out_stride] = v;
}

// ---- host side: per-(device, NT) tap tables -------------------------------------------------------
struct SincTable {
  float* a = nullptr;
  float* b = nullptr;
};
static std::mutex g_tab_mu;
static std::map<std::pair<int, int>, SincTable> g_tabs;

static int get_sinc_table(int device, int NT, SincTable* out) {
  std::lock_guard<std::mutex> lk(g_tab_mu);
  auto key = std::make_pair(device, NT);
  auto it = g_tabs.find(key);
  if (it != g_tabs.end()) {
    *out = it->second;
    return PAR_OK;
  }
  // a[n] = pi*n/win, b[n] = -pi/win with win = float32(np.hanning(2NT+1)[NT+n]), n = 0..NT-1
  std::vector<float> a(NT), b(NT);
  for (int n = 0; n < NT; ++n) {
    float win = (float)(0.5 + 0.5 * cos(M_PI * (double)n / (double)NT));
    a[n] = (float)(M_PI * (double)n / (double)win);
    b[n] = (float)(-M_PI / (double)win);
  }
  SincTable t;
  PAR_HIP_CHECK(hipMalloc(&t.a, NT * sizeof(float)));
  PAR_HIP_CHECK(hipMalloc(&t.b, NT * sizeof(float)));
  PAR_HIP_CHECK(hipMemcpy(t.a, a.data(), NT * sizeof(float), hipMemcpyHostToDevice));
  PAR_HIP_CHECK(hipMemcpy(t.b, b.data(), NT * sizeof(float), hipMemcpyHostToDevice));
  g_tabs[key] = t;
  *out = t;
  return PAR_OK;
}

}  // namespace par

extern "C" {

int par_sinc_resample_f32(int device, const double* pos, int64_t len_out, const float* sig, int64_t sig_stride,
                          int64_t len_in, int NT, float* out, int64_t out_stride, void* stream) {
  using namespace par;
  PAR_REQUIRE(pos && sig && out, PAR_ERR_ARG, "par_sinc_resample_f32: null pointer");
  PAR_REQUIRE(len_out >= 2, PAR_ERR_ARG, "par_sinc_resample_f32: len_out=%lld < 2 (reference raises UnboundLocalError)",
              (long long)len_out);
  PAR_REQUIRE(NT >= 1 && NT <= 512, PAR_ERR_ARG, "par_sinc_resample_f32: NT=%d outside [1,512]", NT);
  PAR_REQUIRE(len_in >= 1 && sig_stride >= 1 && out_stride >= 1, PAR_ERR_ARG, "par_sinc_resample_f32: bad sizes");
  PAR_HIP_CHECK(hipSetDevice(device));
  SincTable tab;
  int rc = get_sinc_table(device, NT, &tab);
  if (rc != PAR_OK) return rc;
  const int64_t blocks = ceil_div(len_out, kSincTile);
  hipLaunchKernelGGL(k_sinc, dim3((unsigned)blocks), dim3(kSincBlock), kSincCap * sizeof(float), as_stream(stream), pos,
                     len_out, sig, sig_stride, len_in, NT, tab.a, tab.b, out, out_stride);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

int par_linear_resample_f32(int device, const double* pos, int64_t len_out, const float* sig, int64_t sig_stride,
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

}  // extern "C"
