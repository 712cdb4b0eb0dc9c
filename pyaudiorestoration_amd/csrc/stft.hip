// K_stft / K_istft -- overlapped-window real STFT (+ fused magnitude) and least-squares ISTFT.
//
// Semantics (reference util/fourier.py): stft :37-75 with the numpy/pyfftw framing --
// estimate_and_center :78-82 (reflect pad n_fft/2, n_frames = (len_pad-n_fft)//hop+1),
// segment_array :160-166 (frame = window * slice, zero-extended at the END to n_fft*zeropad),
// rfft, / sqrt(n_fft) (:157); to_mag :23-24 (|X| + 1e-7) fused as mode 1.
// istft :314-437 with window_sumsquare :492-546 and __overlap_add :677-687.
//
// CDNA4 mapping: no frame matrix is ever materialised.  A 256-thread workgroup owns FR consecutive
// frames: it gathers them straight from the (possibly channel-strided) signal with the reflect
// boundary folded into the index, multiplies by the window, packs the real frame as an M/2-point
// complex sequence in LDS, runs a Stockham autosort FFT (radix-4 stages + one radix-2 when needed,
// ping-pong LDS buffers, no bit reversal), untangles the half-size spectrum into the M/2+1 real-FFT
// bins and writes them FRAME-MAJOR (bins contiguous -> coalesced stores).  HBM traffic is the
// algorithmic minimum: each input sample is fetched from HBM once (neighbouring frames hit L2) and
// each output bin is written once; magnitude never round-trips a complex spectrogram.
#include "par_common.h"
#include <math.h>
#include <map>
#include <vector>

namespace par {

struct Twiddles {
  float2* w = nullptr;    // exp(-2*pi*i*t/H), t = 0..H-1          (complex FFT of size H = M/2)
  float2* post = nullptr; // exp(-2*pi*i*k/M), k = 0..H            (real-FFT untangling)
};
static std::mutex g_tw_mu;
static std::map<std::pair<int, int>, Twiddles> g_tw;

static int get_twiddles(int device, int M, Twiddles* out) {
  std::lock_guard<std::mutex> lk(g_tw_mu);
  auto key = std::make_pair(device, M);
  auto it = g_tw.find(key);
  if (it != g_tw.end()) {
    *out = it->second;
    return PAR_OK;
  }
  const int H = M / 2;
  std::vector<float2> w(H), post(H + 1);
  for (int t = 0; t < H; ++t) {
    const double a = -2.0 * M_PI * (double)t / (double)H;
    w[t] = make_float2((float)cos(a), (float)sin(a));
  }
  for (int k = 0; k <= H; ++k) {
    const double a = -2.0 * M_PI * (double)k / (double)M;
    post[k] = make_float2((float)cos(a), (float)sin(a));
  }
  Twiddles t;
  PAR_HIP_CHECK(hipMalloc(&t.w, H * sizeof(float2)));
  PAR_HIP_CHECK(hipMalloc(&t.post, (H + 1) * sizeof(float2)));
  PAR_HIP_CHECK(hipMemcpy(t.w, w.data(), H * sizeof(float2), hipMemcpyHostToDevice));
  PAR_HIP_CHECK(hipMemcpy(t.post, post.data(), (H + 1) * sizeof(float2), hipMemcpyHostToDevice));
  g_tw[key] = t;
  *out = t;
  return PAR_OK;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// multiply by -i (forward) : (x,y) -> (y,-x)
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }

// np.pad(x, pad, mode="reflect") index: position q relative to x[0], any q (multiple reflections).
__device__ __forceinline__ long long reflect_index(long long q, long long n) {
  if (n == 1) return 0;
  const long long P = 2 * (n - 1);
  q %= P;
  if (q < 0) q += P;
  return q < n ? q : P - q;
}

// In-LDS Stockham autosort FFT of FR independent H-point complex sequences.
// buf holds 2*FR*H float2: [ping | pong]; returns which half holds the result.
__device__ int lds_fft(float2* buf, int FR, int H, int logH, const float2* __restrict__ tw, int tid, int nthreads) {
  int cur = 0;
  int Ns = 1;
  int remaining = logH;
  while (remaining > 0) {
    const float2* in = buf + cur * FR * H;
    float2* out = buf + (cur ^ 1) * FR * H;
    if (remaining >= 2) {
      const int Q = H >> 2;                           // butterflies per frame
      const int twstep = H / (Ns * 4);
      for (int b = tid; b < FR * Q; b += nthreads) {
        const int f = b / Q, j = b - f * Q;
        const int k = j & (Ns - 1);
        const float2* x = in + f * H;
        float2 v0 = x[j], v1 = x[j + Q], v2 = x[j + 2 * Q], v3 = x[j + 3 * Q];
        if (Ns > 1) {
          const float2 w1 = tw[k * twstep], w2 = tw[2 * k * twstep], w3 = tw[3 * k * twstep];
          v1 = cmul(v1, w1);
          v2 = cmul(v2, w2);
          v3 = cmul(v3, w3);
        }
        const float2 a0 = cadd(v0, v2), a1 = csub(v0, v2), a2 = cadd(v1, v3), a3 = mul_mi(csub(v1, v3));
        float2* y = out + f * H + ((j - k) << 2) + k;   // (j/Ns)*Ns*4 + k
        y[0] = cadd(a0, a2);
        y[Ns] = cadd(a1, a3);
        y[2 * Ns] = csub(a0, a2);
        y[3 * Ns] = csub(a1, a3);
      }
      Ns <<= 2;
      remaining -= 2;
    } else {
      const int Q = H >> 1;
      const int twstep = H / (Ns * 2);
      for (int b = tid; b < FR * Q; b += nthreads) {
        const int f = b / Q, j = b - f * Q;
        const int k = j & (Ns - 1);
        const float2* x = in + f * H;
        float2 v0 = x[j], v1 = x[j + Q];
        if (Ns > 1) v1 = cmul(v1, tw[k * twstep]);
        float2* y = out + f * H + ((j - k) << 1) + k;
        y[0] = cadd(v0, v1);
        y[Ns] = csub(v0, v1);
      }
      Ns <<= 1;
      remaining -= 1;
    }
    cur ^= 1;
    __syncthreads();
  }
  return cur;
}

constexpr int kStftThreads = 256;

__global__ __launch_bounds__(kStftThreads) void k_stft(const float* __restrict__ x, int64_t n, int64_t x_stride,
                                                        int n_fft, int hop, int M, int logH, int FR,
                                                        const float* __restrict__ window, const float2* __restrict__ tw,
                                                        const float2* __restrict__ post, float* __restrict__ out,
                                                        int64_t n_frames, int mode, float scale) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const int H = M >> 1;
  const int tid = threadIdx.x;
  const int64_t f0 = (int64_t)blockIdx.x * FR;
  const int half = n_fft >> 1;
  // gather + window + pack: z[j] = (xw[2j], xw[2j+1])
  for (int e = tid; e < FR * H; e += kStftThreads) {
    const int f = e / H, j = e - f * H;
    float2 z = make_float2(0.0f, 0.0f);
    const int64_t fr = f0 + f;
    if (fr < n_frames) {
      const int t0 = 2 * j;
      const long long base = (long long)fr * hop - half;
      if (t0 < n_fft) z.x = window[t0] * x[reflect_index(base + t0, n) * x_stride];
      if (t0 + 1 < n_fft) z.y = window[t0 + 1] * x[reflect_index(base + t0 + 1, n) * x_stride];
    }
    lds[e] = z;
  }
  __syncthreads();
  const int cur = lds_fft(lds, FR, H, logH, tw, tid, kStftThreads);
  const float2* Z = lds + cur * FR * H;
  const int bins = H + 1;
  for (int e = tid; e < FR * bins; e += kStftThreads) {
    const int f = e / bins, k = e - f * bins;
    const int64_t fr = f0 + f;
    if (fr >= n_frames) continue;
    const float2 zk = Z[f * H + (k & (H - 1))];
    const float2 zc = cconj(Z[f * H + ((H - k) & (H - 1))]);
    const float2 ev = cadd(zk, zc);            // 2*E[k]
    const float2 od = csub(zk, zc);            // 2*i*O[k]... untangled below
    const float2 t = cmul(post[k], od);        // W^k * (Z[k] - conj(Z[H-k]))
    // X[k] = 0.5*(ev) - 0.5*i*t
    const float re = 0.5f * (ev.x + t.y) * scale;
    const float im = 0.5f * (ev.y - t.x) * scale;
    if (mode == 0) {
      reinterpret_cast<float2*>(out)[fr * bins + k] = make_float2(re, im);
    } else {
      out[fr * bins + k] = sqrtf(re * re + im * im) + 1e-7f;
    }
  }
}

// ISTFT stage 1: frame f -> window * irfft(spec[f] * sqrt(n_fft))   (util/fourier.py:359, :401)
// irfft of H+1 bins via an H-point complex inverse FFT (conjugate trick on the forward core).
__global__ __launch_bounds__(kStftThreads) void k_istft_frames(const float2* __restrict__ spec, int64_t n_frames,
                                                                int n_fft, int logH, int FR,
                                                                const float* __restrict__ window,
                                                                const float2* __restrict__ tw,
                                                                const float2* __restrict__ post,
                                                                float* __restrict__ frames, float scale) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const int H = n_fft >> 1;
  const int bins = H + 1;
  const int tid = threadIdx.x;
  const int64_t f0 = (int64_t)blockIdx.x * FR;
  // Build conj(Z[k]) with Z[k] = E[k] + i*O[k],  E = (X[k]+conj(X[H-k]))/2,  O = conj(W^k)*(X[k]-conj(X[H-k]))/2
  // (numpy's irfft ignores the imaginary parts of the DC and Nyquist bins.)
  for (int e = tid; e < FR * H; e += kStftThreads) {
    const int f = e / H, k = e - f * H;
    const int64_t fr = f0 + f;
    float2 z = make_float2(0.0f, 0.0f);
    if (fr < n_frames) {
      float2 a = spec[fr * bins + k];
      float2 b = spec[fr * bins + (H - k)];
      if (k == 0) {
        a.y = 0.0f;
        b.y = 0.0f;
      }
      b = cconj(b);
      const float2 ev = cadd(a, b);
      const float2 od = cmul(cconj(post[k]), csub(a, b));
      // Z = 0.5*(ev + i*od)
      z = make_float2(0.5f * (ev.x - od.y), 0.5f * (ev.y + od.x));
      z = cconj(z);
    }
    lds[e] = z;
  }
  __syncthreads();
  const int cur = lds_fft(lds, FR, H, logH, tw, tid, kStftThreads);
  const float2* Z = lds + cur * FR * H;
  // z_time[j] = conj(FFT(conj(Z)))[j] / H ; y[2j] = re, y[2j+1] = im
  for (int e = tid; e < FR * H; e += kStftThreads) {
    const int f = e / H, j = e - f * H;
    const int64_t fr = f0 + f;
    if (fr >= n_frames) continue;
    const float2 v = Z[f * H + j];
    float2 o;
    o.x = v.x * scale * window[2 * j];
    o.y = -v.y * scale * window[2 * j + 1];
    reinterpret_cast<float2*>(frames)[fr * H + j] = o;
  }
}

// ISTFT stage 2: gather-form overlap-add + window-sumsquare normalisation (no atomics):
// y[t] = sum_f frames[f][T - f*hop] / sum_f win^2[T - f*hop],  T = t + skip   (:405-416)
__global__ __launch_bounds__(256) void k_istft_ola(const float* __restrict__ frames, int64_t n_frames, int n_fft, int hop,
                                                    const float* __restrict__ window, float* __restrict__ y,
                                                    int64_t y_len, int64_t skip) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= y_len) return;
  const int64_t T = t + skip;
  const int64_t ola_len = (int64_t)n_fft + (int64_t)hop * (n_frames - 1);
  float acc = 0.0f, env = 0.0f;
  if (T < ola_len) {
    int64_t f_hi = T / hop;
    if (f_hi > n_frames - 1) f_hi = n_frames - 1;
    int64_t f_lo = (T - n_fft + hop) / hop;         // ceil((T - n_fft + 1)/hop)
    if (T - n_fft + 1 <= 0) f_lo = 0;
    for (int64_t f = f_lo; f <= f_hi; ++f) {
      const int64_t off = T - f * hop;
      if (off < 0 || off >= n_fft) continue;
      const float w = window[off];
      acc += frames[f * n_fft + off];
      env += w * w;
    }
    if (env > 1.17549435e-38f) acc /= env;          // > tiny(float32)  (:414-415)
  }
  y[t] = acc;
}

}  // namespace par

extern "C" {

int64_t par_stft_frames(int64_t n, int n_fft, int hop) {
  if (n < 1 || n_fft < 1 || hop < 1) return 0;
  return (n + 2 * (int64_t)(n_fft / 2) - n_fft) / hop + 1;
}

static int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

int par_stft_f32(int device, const float* x, int64_t n, int64_t x_stride, int n_fft, int hop, int zeropad,
                 const float* window, float* out, int mode, void* stream) {
  using namespace par;
  PAR_REQUIRE(x && window && out, PAR_ERR_ARG, "par_stft_f32: null pointer");
  PAR_REQUIRE(n >= 1 && x_stride >= 1 && hop >= 1 && zeropad >= 1 && n_fft >= 2, PAR_ERR_ARG, "par_stft_f32: bad sizes");
  PAR_REQUIRE(mode == 0 || mode == 1, PAR_ERR_ARG, "par_stft_f32: mode must be 0 (complex) or 1 (magnitude)");
  const int64_t M64 = (int64_t)n_fft * zeropad;
  PAR_REQUIRE(M64 >= 16 && M64 <= 8192 && (M64 & (M64 - 1)) == 0 && (n_fft % 2) == 0, PAR_ERR_UNSUPPORTED,
              "par_stft_f32: n_fft*zeropad=%lld is not a power of two in [16, 8192]", (long long)M64);
  const int M = (int)M64, H = M / 2;
  PAR_HIP_CHECK(hipSetDevice(device));
  Twiddles tw;
  int rc = get_twiddles(device, M, &tw);
  if (rc != PAR_OK) return rc;
  const int64_t n_frames = par_stft_frames(n, n_fft, hop);
  int FR = 2048 / H;
  if (FR < 1) FR = 1;
  if (FR > 8) FR = 8;
  const size_t lds = (size_t)2 * FR * H * sizeof(float2);
  const float scale = (float)(1.0 / sqrt((double)n_fft));
  hipLaunchKernelGGL(k_stft, dim3((unsigned)ceil_div(n_frames, FR)), dim3(kStftThreads), lds, as_stream(stream), x, n,
                     x_stride, n_fft, hop, M, ilog2(H), FR, window, tw.w, tw.post, out, n_frames, mode, scale);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

int par_istft_f32(int device, const float* spec, int64_t n_frames, int n_fft, int hop, const float* window,
                  float* frames, float* y, int64_t y_len, int64_t skip, void* stream) {
  using namespace par;
  PAR_REQUIRE(spec && window && frames && y, PAR_ERR_ARG, "par_istft_f32: null pointer");
  PAR_REQUIRE(n_frames >= 1 && hop >= 1 && y_len >= 0 && skip >= 0, PAR_ERR_ARG, "par_istft_f32: bad sizes");
  PAR_REQUIRE(n_fft >= 16 && n_fft <= 8192 && (n_fft & (n_fft - 1)) == 0, PAR_ERR_UNSUPPORTED,
              "par_istft_f32: n_fft=%d is not a power of two in [16, 8192]", n_fft);
  PAR_HIP_CHECK(hipSetDevice(device));
  Twiddles tw;
  int rc = get_twiddles(device, n_fft, &tw);
  if (rc != PAR_OK) return rc;
  const int H = n_fft / 2;
  int FR = 2048 / H;
  if (FR < 1) FR = 1;
  if (FR > 8) FR = 8;
  const size_t lds = (size_t)2 * FR * H * sizeof(float2);
  // spec * sqrt(n_fft) (:359), irfft's 1/n_fft, and the conj-trick's 1/H fold into one factor:
  // irfft(X)[t] = (1/n_fft) * sum ...; the H-point complex inverse carries 1/H with a factor 2 from packing.
  const float scale = (float)(sqrt((double)n_fft) / (double)H);
  hipLaunchKernelGGL(k_istft_frames, dim3((unsigned)ceil_div(n_frames, FR)), dim3(kStftThreads), lds, as_stream(stream),
                     reinterpret_cast<const float2*>(spec), n_frames, n_fft, ilog2(H), FR, window, tw.w, tw.post, frames,
                     scale);
  PAR_HIP_CHECK(hipGetLastError());
  if (y_len > 0) {
    hipLaunchKernelGGL(k_istft_ola, dim3((unsigned)ceil_div(y_len, 256)), dim3(256), 0, as_stream(stream), frames,
                       n_frames, n_fft, hop, window, y, y_len, skip);
    PAR_HIP_CHECK(hipGetLastError());
  }
  return PAR_OK;
}

}  // extern "C"

// ---- spectral gain mask (config 4, dropout_healer_gui.py:161-162): S *= 10^(gain_db/20) ----------------
namespace par {
__global__ void k_apply_gain_db(float2* __restrict__ spec, const float* __restrict__ gain_db, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float g = gain_db[i];
  if (g == 0.0f) return;                               // np.power(10, 0/20) == 1 exactly
  const float f = exp2f(g * 0.16609640474436813f);     // 10^(g/20) = 2^(g*log2(10)/20)
  float2 v = spec[i];
  v.x *= f;
  v.y *= f;
  spec[i] = v;
}
}  // namespace par

extern "C" int par_spec_apply_gain_db_c64(int device, float* spec, const float* gain_db, int64_t count, void* stream) {
  using namespace par;
  PAR_REQUIRE(spec && gain_db && count >= 0, PAR_ERR_ARG, "par_spec_apply_gain_db_c64: bad args");
  if (count == 0) return PAR_OK;
  PAR_HIP_CHECK(hipSetDevice(device));
  hipLaunchKernelGGL(k_apply_gain_db, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<float2*>(spec), gain_db, count);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}
