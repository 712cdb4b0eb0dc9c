// K_stft / K_istft -- overlapped-window real STFT (+ fused magnitude) and least-squares ISTFT.
//
// Semantics (reference util/fourier.py): stft :37-75 with the numpy/pyfftw framing --
// estimate_and_center :78-82 (reflect pad n_fft/2, n_frames = (len_pad-n_fft)//hop+1),
// segment_array :160-166 (frame = window * slice, zero-extended at the END to n_fft*zeropad),
// rfft, / sqrt(n_fft) (:157); to_mag :23-24 (|X| + 1e-7) fused as mode 1.
// istft :314-437 with window_sumsquare :492-546 and __overlap_add :677-687.
//
// CDNA4 mapping: no frame matrix is ever materialised.  A workgroup owns a few consecutive frames,
// H/8 lanes each (one wave per frame at n_fft = 1024): a lane gathers 8 complex points straight from
// the (possibly channel-strided) signal with the reflect boundary folded into the index, multiplies
// by the window (the real frame is packed as an M/2-point complex sequence), runs a Stockham autosort
// FFT with radix-8 butterflies in registers (one LDS exchange per radix-8 stage, no bit reversal),
// untangles the half-size spectrum into the M/2+1 real-FFT bins and writes them FRAME-MAJOR (bins
// contiguous -> coalesced stores).  HBM traffic is the
// algorithmic minimum: each input sample is fetched from HBM once (neighbouring frames hit L2) and
// each output bin is written once; magnitude never round-trips a complex spectrogram.
#include "par_common.h"
#include <math.h>
#include <map>
#include <set>
#include <vector>

#ifndef PAR_STFT_TW
#define PAR_STFT_TW 0
#endif
#ifndef PAR_STFT_STORE
#define PAR_STFT_STORE 0
#endif

namespace par {

struct Twiddles {
  float2* w = nullptr;    // exp(-2*pi*i*t/H), t = 0..H-1          (complex FFT of size H = M/2)
  float2* post = nullptr; // exp(-2*pi*i*k/M), k = 0..H            (real-FFT untangling)
};
static std::mutex g_tw_mu;

// Dynamic LDS above the 64 KB a kernel gets by default is raised per (kernel, device) once: the attribute belongs to the
// function object of the CURRENT device.
static int raise_dynamic_lds(const void* fn, int bytes, int device) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  std::lock_guard<std::mutex> lk(mu);
  if (bytes <= 65536 || done.count({fn, device})) return PAR_OK;
  PAR_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.insert({fn, device});
  return PAR_OK;
}
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

// ---- FFT core ------------------------------------------------------------------------------------------
// H-point complex Stockham autosort FFT with 8 points per lane held in REGISTERS: T = H/8 lanes per frame
// (H = 512 -> exactly one wave per frame).  Every stage consumes x[j + q*T], q = 0..7 (conflict-free
// LDS reads, and for the first stage coalesced global reads), does radix-8 butterflies in registers and
// scatters to the autosort positions; a final radix-2/4 stage covers log2(H) not divisible by 3.  LDS is
// touched log8(H) times instead of log2..log4(H) times, and indices are padded e -> e + e/8 so the
// scattered 64-bit writes are bank-conflict free.
__device__ __forceinline__ int lpad(int e) { return e + (e >> 3); }

__device__ __forceinline__ void radix2(float2& a, float2& b) {
  const float2 t = a;
  a = cadd(t, b);
  b = csub(t, b);
}
// forward 4-point DFT in place: (v0,v1,v2,v3) -> (X0,X1,X2,X3)
__device__ __forceinline__ void radix4(float2& v0, float2& v1, float2& v2, float2& v3) {
  const float2 a0 = cadd(v0, v2), a1 = csub(v0, v2), a2 = cadd(v1, v3), a3 = mul_mi(csub(v1, v3));
  v0 = cadd(a0, a2);
  v1 = cadd(a1, a3);
  v2 = csub(a0, a2);
  v3 = csub(a1, a3);
}
// forward 8-point DFT in place, natural order in and out
__device__ __forceinline__ void radix8(float2 (&v)[8]) {
  // split into even (0,2,4,6) and odd (1,3,5,7) 4-point DFTs
  float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
  float2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  radix4(e0, e1, e2, e3);
  radix4(o0, o1, o2, o3);
  const float h = 0.70710678118654752f;
  // odd outputs times W8^k: W8^1 = (1-i)/sqrt2, W8^2 = -i, W8^3 = (-1-i)/sqrt2
  o1 = make_float2((o1.x + o1.y) * h, (o1.y - o1.x) * h);
  o2 = mul_mi(o2);
  o3 = make_float2((o3.y - o3.x) * h, -(o3.x + o3.y) * h);
  v[0] = cadd(e0, o0);
  v[4] = csub(e0, o0);
  v[1] = cadd(e1, o1);
  v[5] = csub(e1, o1);
  v[2] = cadd(e2, o2);
  v[6] = csub(e2, o2);
  v[3] = cadd(e3, o3);
  v[7] = csub(e3, o3);
}

// The lanes of one frame exchange data through LDS.  Up to 64 lanes per frame they all sit in ONE wave, whose
// LDS operations execute in order: a compiler-level fence is enough and the waves of a workgroup never wait
// for each other; larger frames need the workgroup barrier.
template <int T>
__device__ __forceinline__ void frame_sync() {
  if constexpr (T <= kWave) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
}

// v[q] = x[j + q*T] on entry (natural order); on exit X (padded LDS frame) holds the transform in natural order.
// All lanes of the frame (for T > 64: of the workgroup) must call this together.
template <int LOGH>
__device__ __forceinline__ void fft_core(float2 (&v)[8], float2* __restrict__ X, int j, const float2* __restrict__ tw) {
  constexpr int H = 1 << LOGH, T = H / 8;
  constexpr int N8 = LOGH / 3, REM = LOGH % 3;
  // the gathered twiddle of every later stage depends only on the lane: fetch them all up front so their
  // latency hides under the first butterflies instead of sitting between the LDS exchanges
  float2 w1s[N8 > 1 ? N8 : 1];
#pragma unroll
  for (int st = 1; st < N8; ++st) {
    const int Nst = 1 << (3 * st);
    w1s[st] = tw[(j & (Nst - 1)) * (H / (Nst * 8))];
  }
  int Ns = 1;
#pragma unroll
  for (int st = 0; st < N8; ++st) {
    if (st > 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = X[lpad(j + q * T)];
    }
    const int k = j & (Ns - 1);
    if (st > 0) {
      // one gathered twiddle load per stage; the other six are its powers (3 multiply levels, ~2 ulp)
      const float2 w1 = w1s[st];
#if PAR_STFT_TW == 1          // accuracy experiment: all seven from the table (correctly rounded)
      const int ti = (j & (Ns - 1)) * (H / (Ns * 8));
      const float2 w2 = tw[2 * ti], w3 = tw[3 * ti], w4 = tw[4 * ti], w5 = tw[5 * ti], w6 = tw[6 * ti], w7 = tw[7 * ti];
#elif PAR_STFT_TW == 2        // accuracy experiment: w1, w2, w4 from the table, the rest one multiply away
      const int ti = (j & (Ns - 1)) * (H / (Ns * 8));
      const float2 w2 = tw[2 * ti], w4 = tw[4 * ti];
      const float2 w3 = cmul(w2, w1), w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
#else
      const float2 w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
      const float2 w5 = cmul(w4, w1), w6 = cmul(w3, w3), w7 = cmul(w4, w3);
#endif
      v[1] = cmul(v[1], w1);
      v[2] = cmul(v[2], w2);
      v[3] = cmul(v[3], w3);
      v[4] = cmul(v[4], w4);
      v[5] = cmul(v[5], w5);
      v[6] = cmul(v[6], w6);
      v[7] = cmul(v[7], w7);
    }
    radix8(v);
    frame_sync<T>();                       // everybody has finished reading X
    const int base = ((j - k) << 3) + k;
#pragma unroll
    for (int r = 0; r < 8; ++r) X[lpad(base + r * Ns)] = v[r];
    frame_sync<T>();
    Ns <<= 3;
  }
  if (REM) {
    constexpr int Rl = 1 << REM, U = 8 / Rl;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = X[lpad(j + q * T)];
    const int step = H / (Ns * Rl);
    int bk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = j + u * T;
      const int k = b & (Ns - 1);
      bk[u] = ((b - k) * Rl) + k;
#pragma unroll
      for (int r = 1; r < Rl; ++r) v[u + r * U] = cmul(v[u + r * U], tw[r * k * step]);
      if (Rl == 2) radix2(v[u], v[u + U]);
      else radix4(v[u], v[u + U], v[u + 2 * U], v[u + 3 * U]);
    }
    frame_sync<T>();
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int r = 0; r < Rl; ++r) X[lpad(bk[u] + r * Ns)] = v[u + r * U];
    frame_sync<T>();
  }
}

// Workgroups are dispatched round-robin over the 8 XCDs, each with its own L2.  Overlapping frames share most of
// their input, so consecutive frame groups should meet in ONE L2: logical group g = (b % 8) * ceil(n/8) + b / 8
// gives every XCD a contiguous range of frames (groups past the end exit).
constexpr int kXcds = 8;
__device__ __forceinline__ int64_t xcd_contiguous_block(int64_t n_blocks) {
  const int64_t per = (n_blocks + kXcds - 1) / kXcds;
  return (int64_t)(blockIdx.x % kXcds) * per + blockIdx.x / kXcds;
}

template <int LOGH>
struct FftGeom {
  static constexpr int H = 1 << LOGH;
  static constexpr int T = H / 8;                               // lanes per frame
  static constexpr int Threads = T > 256 ? T : 256;
  static constexpr int Frames = Threads / T;                    // frames per workgroup
  static constexpr int FrameLds = H + H / 8 + 8;                // padded float2 slots per frame
};

// MODE 0: complex spectrum, 1: magnitude |X| + 1e-7.  UNIT: the signal is contiguous (x_stride == 1): no 64-bit stride
// multiply per sample and the two samples of a packed point arrive in one 8-byte load.
template <int LOGH, int MODE, bool UNIT>
__global__ __launch_bounds__(FftGeom<LOGH>::Threads) void k_stft(const float* __restrict__ x, int64_t n, int64_t x_stride_arg,
                                                                  int n_fft, int hop, const float* __restrict__ window,
                                                                  const float2* __restrict__ tw,
                                                                  const float2* __restrict__ post, float* __restrict__ out,
                                                                  int64_t n_frames, float scale, int64_t pitch) {
  const int64_t x_stride = UNIT ? 1 : x_stride_arg;
  constexpr int mode = MODE;
  using G = FftGeom<LOGH>;
  constexpr int H = G::H, T = G::T;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const int f = threadIdx.x / T, j = threadIdx.x - f * T;
  float2* X = lds + f * G::FrameLds;
  const int64_t fr = xcd_contiguous_block((n_frames + G::Frames - 1) / G::Frames) * G::Frames + f;
  const bool live = fr < n_frames;
  // gather + window + pack: z[i] = (xw[2i], xw[2i+1]); lanes read consecutive float pairs (coalesced)
  float2 v[8];
  const long long base = (long long)fr * hop - (n_fft >> 1);
  // interior frames (the vast majority) index the signal directly; only frames that overhang an end of
  // the signal pay for the reflect fold (a 64-bit modulo per sample)
  const bool interior = live && base >= 0 && base + n_fft <= (long long)n;
  if (interior && n_fft == 2 * H) {
    // no zero padding (the usual case): every packed pair lies inside the frame, no compare per pair
    const float* xs = x + base * x_stride;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int t0 = 2 * (j + q * T);
      const float2 w = *reinterpret_cast<const float2*>(window + t0);
      v[q] = make_float2(w.x * xs[(int64_t)t0 * x_stride], w.y * xs[(int64_t)(t0 + 1) * x_stride]);
    }
  } else if (interior) {
    const float* xs = x + base * x_stride;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int t0 = 2 * (j + q * T);
      float2 z = make_float2(0.0f, 0.0f);
      if (t0 + 1 < n_fft) {
        const float2 w = *reinterpret_cast<const float2*>(window + t0);
        z.x = w.x * xs[(int64_t)t0 * x_stride];
        z.y = w.y * xs[(int64_t)(t0 + 1) * x_stride];
      } else if (t0 < n_fft) {
        z.x = window[t0] * xs[(int64_t)t0 * x_stride];
      }
      v[q] = z;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int t0 = 2 * (j + q * T);
      float2 z = make_float2(0.0f, 0.0f);
      if (live) {
        if (t0 < n_fft) z.x = window[t0] * x[reflect_index(base + t0, n) * x_stride];
        if (t0 + 1 < n_fft) z.y = window[t0 + 1] * x[reflect_index(base + t0 + 1, n) * x_stride];
      }
      v[q] = z;
    }
  }
  // untangling twiddles W^k of this lane's bins: fetched before the FFT so they are in registers when the last
  // exchange completes.  Bins come in pairs (k, H-k) that share everything but two signs, so a lane owns
  // k = j, j+T, .. below H/2 and writes both members (lane 0 adds the self-paired bin H/2).
  constexpr int bins = H + 1, P = (H / 2) / T;
  float2 pw[P + 1];
#pragma unroll
  for (int i = 0; i < P; ++i) pw[i] = post[j + i * T];
  pw[P] = post[H / 2];
  fft_core<LOGH>(v, X, j, tw);
  if (!live) return;
  const float hs = 0.5f * scale;
  // PAR_STFT_STORE (experiment switch): 0 streaming (nontemporal) stores straight from the registers; 1 plain stores;
  // 3 one-wave-per-frame sizes stage the row in the wave's own LDS and write it with 16-byte aligned stores
  constexpr bool kRowStage = (PAR_STFT_STORE == 3) && T == kWave;
  float* Mg = reinterpret_cast<float*>(lds + G::Frames * G::FrameLds) + f * (bins + 3);      // this frame's row (mode 1)
  // (re, im) arrive UNSCALED (twice the bin, before the 1/sqrt(n_fft)): the complex form scales both parts, the magnitude
  // form scales once behind the square root (|hs z| = hs |z|: two multiplies per bin saved)
  auto emit = [&](int k, float re, float im) {
    // The magnitude spectrogram is consumed sparsely (tracker bands): streaming stores (get_mag 0.39 -> 0.365 ms).
    // The complex one is re-read right away by the inpaint / ISTFT kernels: regular stores (streaming ones cost
    // 2 % on the config-4 chain).
    if constexpr (mode == 0) {
      reinterpret_cast<float2*>(out)[fr * pitch + k] = make_float2(re * hs, im * hs);
    } else {
      // v_sqrt_f32 itself (1 ulp): the correctly rounded sqrtf costs 16 instructions per bin, mostly compares and selects
      const float mag = fmaf(__builtin_amdgcn_sqrtf(fmaf(re, re, im * im)), hs, 1e-7f);
      // streaming stores pay when a frame's lanes write runs of at least 64 bytes (T >= 16: n_fft >= 256); shorter runs only
      // complete their lines together with the neighbouring frames', in L2, and went to HBM piecemeal as streaming stores
      // (n_fft = 64, hop 16 on 57.6 M samples: 0.81 ms streaming, 0.39 ms plain; n_fft = 256: 0.26 against 0.31)
      if (kRowStage) Mg[k] = mag;
      else if (PAR_STFT_STORE == 1 || T < 16) out[fr * pitch + k] = mag;
      else __builtin_nontemporal_store(mag, out + fr * pitch + k);
    }
  };
  // LDS slots of the pair (k, H - k), k = j + i T: with T a multiple of 8 the padding is linear in i, so both are one
  // base per lane plus an immediate (the wrap H - 0 -> 0 only exists for lane 0's first pair)
  constexpr bool kLinearPad = (T % 8) == 0;
  constexpr int kPadStep = T + T / 8;
  const int lp_k0 = lpad(j), lp_m0 = lpad(H - j);
#pragma unroll
  for (int i = 0; i <= P; ++i) {
    const int k = (i < P) ? j + i * T : H / 2;
    if (i == P && j != 0) break;
    int sk = lpad(k), sm = lpad((H - k) & (H - 1));
    if (kLinearPad && i < P) {
      sk = lp_k0 + i * kPadStep;
      sm = (i == 0 && j == 0) ? 0 : lp_m0 - i * kPadStep;
    }
    const float2 zk = X[sk];
    const float2 zc = cconj(X[sm]);
    const float2 ev = cadd(zk, zc);                     // Z[k] + conj(Z[H-k])
    const float2 t = cmul(pw[i], csub(zk, zc));         // W^k * (Z[k] - conj(Z[H-k]))
    emit(k, ev.x + t.y, ev.y - t.x);                    // X[k]   = (ev - i*t)/2 * scale
    if (i < P) emit(H - k, ev.x - t.y, -ev.y - t.x);    // X[H-k] = (conj(ev) - i*conj(t))/2 * scale
  }
  if (kRowStage && mode == 1) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    typedef float vf4 __attribute__((ext_vector_type(4)));
    const int64_t e0 = fr * pitch, e1 = e0 + bins;
    const int64_t a0 = (e0 + 3) & ~3ll, a1 = e1 & ~3ll;
    for (int64_t e = a0 + 4 * j; e < a1; e += 4 * T) {
      const float* m = Mg + (e - e0);
      const vf4 q = {m[0], m[1], m[2], m[3]};
      __builtin_nontemporal_store(q, reinterpret_cast<vf4*>(out + e));
    }
    if (j < a0 - e0) __builtin_nontemporal_store(Mg[j], out + e0 + j);
    if (j < e1 - a1) __builtin_nontemporal_store(Mg[a1 - e0 + j], out + a1 + j);
  }
}

// ---- transforms above 8192 points: four-step FFT ----------------------------------------------------------
// The GUI offers FFT sizes up to 2^20 (util/widgets.py:333-349).  A frame of M = n_fft*zeropad real points is an
// H = M/2-point complex sequence; H = N1 N2 is transformed in two passes over HBM around the register/LDS core:
//   pass A  N2 column transforms of N1 points (input index n2 + N2 n1), times W_H^(n2 k1), stored as A[k1][n2];
//           the gather (reflect boundary), the window and the real->complex packing are fused into its loads
//   pass B  N1 row transforms of N2 points, stored at k1 + N1 k2
//   untangle to the H+1 real-FFT bins (complex, or |X| + 1e-7)
// A workgroup owns C = 8 adjacent columns (rows): both its loads and its stores then move 64-byte runs (loads and
// stores go through one LDS transposition each), and the C transforms run side by side in the core.
struct BigTw {
  float2* lo = nullptr;     // exp(-2 pi i m / H),        m = 0 .. kBigR-1
  float2* hi = nullptr;     // exp(-2 pi i m kBigR / H),  m = 0 .. H/kBigR-1
};
constexpr int kBigR = 1024;
constexpr int kBigC = 8;
static std::map<std::pair<int, int>, BigTw> g_bigtw;

static int get_big_twiddles(int device, int H, BigTw* out) {
  std::lock_guard<std::mutex> lk(g_tw_mu);
  auto key = std::make_pair(device, H);
  auto it = g_bigtw.find(key);
  if (it != g_bigtw.end()) {
    *out = it->second;
    return PAR_OK;
  }
  const int nh = H / kBigR > 1 ? H / kBigR : 1;
  std::vector<float2> lo(kBigR), hi(nh);
  for (int m = 0; m < kBigR; ++m) {
    const double a = -2.0 * M_PI * (double)m / (double)H;
    lo[m] = make_float2((float)cos(a), (float)sin(a));
  }
  for (int m = 0; m < nh; ++m) {
    const double a = -2.0 * M_PI * (double)m * (double)kBigR / (double)H;
    hi[m] = make_float2((float)cos(a), (float)sin(a));
  }
  BigTw t;
  PAR_HIP_CHECK(hipMalloc(&t.lo, lo.size() * sizeof(float2)));
  PAR_HIP_CHECK(hipMalloc(&t.hi, hi.size() * sizeof(float2)));
  PAR_HIP_CHECK(hipMemcpy(t.lo, lo.data(), lo.size() * sizeof(float2), hipMemcpyHostToDevice));
  PAR_HIP_CHECK(hipMemcpy(t.hi, hi.data(), hi.size() * sizeof(float2), hipMemcpyHostToDevice));
  g_bigtw[key] = t;
  *out = t;
  return PAR_OK;
}

// PASS 0: columns of the packed windowed frame -> A;  PASS 1: rows of A -> Z.
// grid: round_up(groups * n_batch, 8) workgroups in ONE dimension, group = kBigC adjacent transforms of one frame.  The
// hardware deals workgroups round-robin to the 8 XCDs, so neighbouring column groups -- whose 64-byte runs are the two
// halves of the same 128-byte lines, on the load and on the store side -- used to meet in eight different L2s and reach
// HBM as scattered half-line accesses (1.4 GB moved in 0.79 ms).  xcd_contiguous_block() hands every XCD a contiguous
// range of (frame, group) pairs instead, so the halves merge in one L2.
__host__ __device__ constexpr int64_t round_up8(int64_t v) { return (v + 7) & ~7ll; }
template <int LOGS, int PASS, int SRC = 0>
__global__ __launch_bounds__(1 << LOGS) void k_bigfft(const float* __restrict__ x, int64_t n, int64_t x_stride, int n_fft, int hop,
                                                      const float* __restrict__ window, const float2* __restrict__ tw,
                                                      const float2* __restrict__ tlo, const float2* __restrict__ thi,
                                                      float2* __restrict__ A, float2* __restrict__ Z, int64_t f_first,
                                                      int logN1, int logN2, int64_t n_batch) {
  static_assert(LOGS >= 6 && LOGS <= 10, "one lane per 8 points, kBigC transforms side by side: blockDim.x == 1 << LOGS");
  constexpr int S = 1 << LOGS, T = S / 8;
  constexpr int FrameLds = S + S / 8 + 8;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const int N1 = 1 << logN1, N2 = 1 << logN2;
  const int64_t H = (int64_t)N1 * N2;
  const int tid = threadIdx.x;
  const int groups = (PASS == 0 ? N2 : N1) / kBigC;
  const int64_t w = xcd_contiguous_block(groups * n_batch);
  if (w >= groups * n_batch) return;
  const int64_t fb = w / groups;                           // frame of this batch
  const int g0 = (int)(w - fb * groups) * kBigC;           // first column (pass 0) / row (pass 1) of the group
  float2* Ab = A + fb * H;
  // every loop below runs kBigC times with compile-time bounds: all of a thread's loads are in flight together
  if (PASS == 0 && SRC == 1) {                             // plain complex input, transformed in place
#pragma unroll
    for (int it = 0; it < kBigC; ++it) {
      const int idx = tid + it * S, n1 = idx / kBigC, f = idx % kBigC;
      lds[f * FrameLds + lpad(n1)] = Ab[(int64_t)(g0 + f) + (int64_t)N2 * n1];
    }
  } else if (PASS == 0) {
    const long long base = (long long)(f_first + fb) * hop - (n_fft >> 1);
    // a frame that lies inside the signal (all but the first and last few) indexes it directly; the reflect fold is a
    // 64-bit modulo per sample
    if (base >= 0 && base + n_fft <= (long long)n) {
      const float* xs = x + base * x_stride;
      // contiguous signal, frame starting on an 8-byte boundary (even hop and n_fft / 2): the two samples of a packed
      // point arrive in one load
      const bool paired = x_stride == 1 && (reinterpret_cast<uintptr_t>(xs) & 7) == 0;
      // branch-free: n_fft is even, so a packed pair lies entirely inside the frame or entirely in the zero padding; a
      // padded pair loads the frame's last pair instead and is zeroed afterwards (a branch per pair made every load wait
      // for the one before it)
      float2 z[kBigC], wv[kBigC];
      int t0[kBigC];
#pragma unroll
      for (int it = 0; it < kBigC; ++it) {
        const int idx = tid + it * S, n1 = idx / kBigC, f = idx % kBigC;
        t0[it] = 2 * (g0 + f + N2 * n1);                   // first real sample of the packed pair (< n_fft * zeropad <= 2^21)
        const int tc = t0[it] < n_fft ? t0[it] : n_fft - 2;
        wv[it] = *reinterpret_cast<const float2*>(window + tc);
        t0[it] = t0[it] < n_fft ? tc : -1;
      }
      if (paired) {
#pragma unroll
        for (int it = 0; it < kBigC; ++it) z[it] = *reinterpret_cast<const float2*>(xs + (t0[it] < 0 ? n_fft - 2 : t0[it]));
      } else {
#pragma unroll
        for (int it = 0; it < kBigC; ++it) {
          const int64_t tc = t0[it] < 0 ? n_fft - 2 : t0[it];
          z[it] = make_float2(xs[tc * x_stride], xs[(tc + 1) * x_stride]);
        }
      }
#pragma unroll
      for (int it = 0; it < kBigC; ++it) {
        z[it].x = t0[it] < 0 ? 0.0f : wv[it].x * z[it].x;
        z[it].y = t0[it] < 0 ? 0.0f : wv[it].y * z[it].y;
      }
#pragma unroll
      for (int it = 0; it < kBigC; ++it) {
        const int idx = tid + it * S, n1 = idx / kBigC, f = idx % kBigC;
        lds[f * FrameLds + lpad(n1)] = z[it];
      }
    } else {
      for (int it = 0; it < kBigC; ++it) {
        const int idx = tid + it * S, n1 = idx / kBigC, f = idx % kBigC;
        const long long t0 = 2ll * ((long long)(g0 + f) + (long long)N2 * n1);
        float2 z = make_float2(0.0f, 0.0f);
        if (t0 < n_fft) z.x = window[t0] * x[reflect_index(base + t0, n) * x_stride];
        if (t0 + 1 < n_fft) z.y = window[t0 + 1] * x[reflect_index(base + t0 + 1, n) * x_stride];
        lds[f * FrameLds + lpad(n1)] = z;
      }
    }
  } else {
#pragma unroll
    for (int f = 0; f < kBigC; ++f) lds[f * FrameLds + lpad(tid)] = Ab[(int64_t)(g0 + f) * N2 + tid];
  }
  __syncthreads();
  // kBigC transforms side by side: thread -> (transform f, lane j)
  const int f = tid / T, j = tid - f * T;
  float2* X = lds + f * FrameLds;
  float2 v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = X[lpad(j + q * T)];
  fft_core<LOGS>(v, X, j, tw);
  __syncthreads();
  if (PASS == 0) {
    float2 wv[kBigC];
#pragma unroll
    for (int it = 0; it < kBigC; ++it) {
      const int idx = tid + it * S, k1 = idx / kBigC, ff = idx % kBigC;
      const unsigned m = (unsigned)(g0 + ff) * (unsigned)k1;       // < H <= 2^20
      wv[it] = cmul(thi[m / kBigR], tlo[m % kBigR]);
    }
#pragma unroll
    for (int it = 0; it < kBigC; ++it) {
      const int idx = tid + it * S, k1 = idx / kBigC, ff = idx % kBigC;
      Ab[(int64_t)k1 * N2 + g0 + ff] = cmul(lds[ff * FrameLds + lpad(k1)], wv[it]);
    }
  } else {
    float2* Zb = Z + fb * H;
#pragma unroll
    for (int it = 0; it < kBigC; ++it) {
      const int idx = tid + it * S, k2 = idx / kBigC, ff = idx % kBigC;
      Zb[(int64_t)(g0 + ff) + (int64_t)N1 * k2] = lds[ff * FrameLds + lpad(k2)];
    }
  }
}

// Row pass of the STFT, fused with the untangle to the H+1 real-FFT bins
//   X[k] = (Z[k] + conj Z[H-k])/2 - i W_2H^k (Z[k] - conj Z[H-k])/2,      k = k1 + N1 k2,  H - k = (N1 - k1) + N1 (N2 - 1 - k2):
// a bin's partner lives in row N1 - k1, so a workgroup transforms the kBigC rows k1 = 8m+1 .. 8m+8 AND their partners
// N1-8m-8 .. N1-8m-1 (two rounds through the core, all 16 spectra staying in LDS) and writes both output runs itself:
// the H-point spectrum Z never travels to HBM (it used to be written, then read back by a separate untangle kernel:
// 2 x 8 H bytes per frame of the pass's 3 x 8 H).  Row N1/2 pairs with itself and rides on both sides of the last
// regular group; row 0 pairs with itself under k2 -> (N2 - k2) mod N2 and has the extra group blockIdx.x == N1/16.
// tlo2/thi2: the two-level table of the 2H-point circle.  MODE 0: complex bins, 1: |X| + 1e-7.
template <int LOGS, int MODE>
__global__ __launch_bounds__(1 << LOGS) void k_bigfft_rows_out(const float2* __restrict__ A, const float2* __restrict__ tw,
                                                               const float2* __restrict__ tlo2, const float2* __restrict__ thi2,
                                                               float* __restrict__ out, int64_t f_first, int logN1, float scale,
                                                               int64_t n_batch) {
  static_assert(LOGS >= 6 && LOGS <= 10, "one lane per 8 points, 8 transforms per round");
  constexpr int S = 1 << LOGS, T = S / 8, N2 = S;
  constexpr int FrameLds = S + S / 8 + 8;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const int N1 = 1 << logN1;
  const int64_t H = (int64_t)N1 * N2, bins = H + 1;
  const int tid = threadIdx.x;                               // blockDim.x == S
  // one grid dimension, XCD-contiguous like k_bigfft: neighbouring groups write neighbouring 32/64-byte runs
  const int groups = N1 / (2 * kBigC) + 1;
  const int64_t w = xcd_contiguous_block(groups * n_batch);
  if (w >= groups * n_batch) return;
  const int64_t fb = w / groups, fr = f_first + fb;
  const float2* Ab = A + fb * H;
  const int m = (int)(w - fb * groups);
  const bool row0 = m == N1 / 16;
  const int p0 = 8 * m + 1, q0 = N1 - 8 * m - 8;             // first row of the group, first row of its partners
  if (row0) {
    lds[lpad(tid)] = Ab[tid];
#pragma unroll
    for (int f = 1; f < kBigC; ++f) lds[f * FrameLds + lpad(tid)] = make_float2(0.0f, 0.0f);
  } else {
#pragma unroll
    for (int f = 0; f < 2 * kBigC; ++f) {
      const int row = f < kBigC ? p0 + f : q0 + (f - kBigC);
      lds[f * FrameLds + lpad(tid)] = Ab[(int64_t)row * N2 + tid];
    }
  }
  __syncthreads();
  const int f = tid / T, j = tid - f * T;
  for (int round = 0; round < (row0 ? 1 : 2); ++round) {
    float2* X = lds + (round * kBigC + f) * FrameLds;
    float2 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = X[lpad(j + q * T)];
    fft_core<LOGS>(v, X, j, tw);
  }
  __syncthreads();
  const float hs = 0.5f * scale;
  auto emit = [&](int64_t kk, float re, float im) {
    // plain stores: a group's runs are 32 (magnitudes) or 64 bytes long and complete their 128-byte lines only together with
    // the neighbouring groups' -- in L2.  Streaming (nontemporal) stores, right for K_stft's whole rows, sent every run to
    // HBM on its own: 0.85 ms for this kernel against 0.32 ms (0.25 ms with the stores removed altogether).
    if constexpr (MODE == 0) reinterpret_cast<float2*>(out)[fr * bins + kk] = make_float2(re * hs, im * hs);
    else out[fr * bins + kk] = fmaf(__builtin_amdgcn_sqrtf(fmaf(re, re, im * im)), hs, 1e-7f);
  };
  auto half_step = [&](int64_t k) { return cmul(thi2[(uint64_t)k / kBigR], tlo2[(uint64_t)k % kBigR]); };
  if (row0) {
    if (tid <= N2 / 2) {
      const int k2 = tid;
      const int64_t k = (int64_t)N1 * k2;
      const float2 zk = lds[lpad(k2)];
      const float2 zc = cconj(lds[lpad((N2 - k2) & (N2 - 1))]);
      const float2 ev = cadd(zk, zc);
      const float2 t = cmul(half_step(k), csub(zk, zc));
      emit(k, ev.x + t.y, ev.y - t.x);
      if (k2 > 0 && k2 < N2 / 2) emit(H - k, ev.x - t.y, -ev.y - t.x);
      if (k2 == 0) emit(H, ev.x - t.y, -ev.y - t.x);
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < kBigC; ++it) {
    const int idx = tid + it * S;
    const int k2 = idx / kBigC, i = idx % kBigC;
    const int k1 = p0 + i;
    if (k1 == N1 / 2 && k2 >= N2 / 2) continue;              // the self-paired row: each pair once
    const int64_t k = k1 + (int64_t)N1 * k2;
    const float2 zk = lds[i * FrameLds + lpad(k2)];
    const float2 zc = cconj(lds[(2 * kBigC - 1 - i) * FrameLds + lpad(N2 - 1 - k2)]);
    const float2 ev = cadd(zk, zc);
    const float2 t = cmul(half_step(k), csub(zk, zc));
    emit(k, ev.x + t.y, ev.y - t.x);
    emit(H - k, ev.x - t.y, -ev.y - t.x);
  }
}

// H-point complex FFT (H a power of two in [8192, 2^20]) of `batch` arrays: A is overwritten, the spectrum lands in Z
static int big_fft_c2c(int device, float2* A, float2* Z, int64_t H, int64_t batch, hipStream_t s) {
  const int L = [&] { int l = 0; while ((1ll << l) < H) ++l; return l; }(), l1 = (L + 1) / 2, l2 = L - l1;
  PAR_REQUIRE(H >= 8192 && H <= (1ll << 20) && (1ll << L) == H, PAR_ERR_UNSUPPORTED, "big_fft_c2c: H=%lld", (long long)H);
  Twiddles t1, t2;
  BigTw bt;
  int rc = get_twiddles(device, 2 << l1, &t1);
  if (rc == PAR_OK) rc = get_twiddles(device, 2 << l2, &t2);
  if (rc == PAR_OK) rc = get_big_twiddles(device, (int)H, &bt);
  if (rc != PAR_OK) return rc;
#define PAR_BIG_C2C(LS, PASS, NG, TW)                                                                                     \
  if (int rc_lds = raise_dynamic_lds(reinterpret_cast<const void*>(&k_bigfft<LS, PASS, 1>),                               \
                                     kBigC * ((1 << LS) + (1 << LS) / 8 + 8) * (int)sizeof(float2), device))              \
    return rc_lds;                                                                                                        \
  hipLaunchKernelGGL((k_bigfft<LS, PASS, 1>), dim3((unsigned)round_up8((int64_t)((NG) / kBigC) * batch)), dim3(1 << LS),   \
                     (size_t)kBigC * ((1 << LS) + (1 << LS) / 8 + 8) * sizeof(float2), s, (const float*)nullptr, (int64_t)0, \
                     (int64_t)1, 0, 1, (const float*)nullptr, TW, bt.lo, bt.hi, A, Z, (int64_t)0, l1, l2, batch)
  switch (l1) {
    case 7: PAR_BIG_C2C(7, 0, 1 << l2, t1.w); break;
    case 8: PAR_BIG_C2C(8, 0, 1 << l2, t1.w); break;
    case 9: PAR_BIG_C2C(9, 0, 1 << l2, t1.w); break;
    case 10: PAR_BIG_C2C(10, 0, 1 << l2, t1.w); break;
    default: PAR_REQUIRE(false, PAR_ERR_UNSUPPORTED, "big_fft_c2c: unsupported size");
  }
  switch (l2) {
    case 6: PAR_BIG_C2C(6, 1, 1 << l1, t2.w); break;
    case 7: PAR_BIG_C2C(7, 1, 1 << l1, t2.w); break;
    case 8: PAR_BIG_C2C(8, 1, 1 << l1, t2.w); break;
    case 9: PAR_BIG_C2C(9, 1, 1 << l1, t2.w); break;
    case 10: PAR_BIG_C2C(10, 1, 1 << l1, t2.w); break;
    default: PAR_REQUIRE(false, PAR_ERR_UNSUPPORTED, "big_fft_c2c: unsupported size");
  }
#undef PAR_BIG_C2C
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

// ---- transforms of 2^22 .. 2^24 points (r03): one more radix step around the four-step transform ------------------------
// The GUI offers FFT sizes to 2^20 with zero-padding to 16 (util/widgets.py:334-351): frames of up to 2^24 real points, an
// H = 2^23-point complex sequence z.  The four-step core holds rows / columns of <= 1024 points (H <= 2^20), so the
// sequence is decimated in time by R = H / 2^20 (2, 4 or 8): z_r[m] = z[R m + r] are R sequences of 2^20 points,
// transformed as a batch by big_fft_c2c, and
//     Z[k' + H' q] = sum_r W_R^(r q) (W_H^(r k') Z_r[k'])          k' < H' = 2^20, q < R
// is formed bin by bin in the kernel that also untangles the real transform and writes |X| or X.  Five passes over an
// H-point array per frame instead of the four-step's three, for sizes the reference's CPU chain needs seconds per frame for.
//   k_huge_gather   z_r[m] = (xw[2 (R m + r)], xw[2 (R m + r) + 1]): window, reflect boundary, zero padding
//   k_huge_out      R-point recombination with float64 twiddles, untangle, magnitude
__global__ __launch_bounds__(256) void k_huge_gather(const float* __restrict__ x, int64_t n, int64_t x_stride, int n_fft, int hop,
                                                     const float* __restrict__ window, int64_t frame, int logR, int64_t Hp,
                                                     float2* __restrict__ z) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // t = r Hp + m
  if (t >= (Hp << logR)) return;
  const int64_t r = t / Hp, m = t - r * Hp;
  const int64_t i = (m << logR) + r;                                        // packed point of the frame
  const long long base = (long long)frame * hop - (n_fft >> 1);
  float2 v = make_float2(0.0f, 0.0f);
  const int64_t t0 = 2 * i;
  if (t0 < n_fft) v.x = window[t0] * x[reflect_index(base + t0, n) * x_stride];
  if (t0 + 1 < n_fft) v.y = window[t0 + 1] * x[reflect_index(base + t0 + 1, n) * x_stride];
  z[t] = v;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_huge_out(const float2* __restrict__ Zr, int logR, int64_t Hp, float* __restrict__ out,
                                                  int64_t row, float scale) {
  const int64_t kp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // k' < H'
  if (kp >= Hp) return;
  const int R = 1 << logR;
  const int64_t H = Hp << logR;
  // the R bins k' + H' q of this thread and their partners H - k = (H' - k') + H' (R - 1 - q)  (k' = 0: H' ((R - q) mod R))
  const int64_t kc = kp == 0 ? 0 : Hp - kp;
  double ar[8], ai[8], br[8], bi[8];                                        // W_H^(r k') Z_r[k'] and the same at k'' = kc
  double s1, c1, s2, c2;
  sincospi(-2.0 * (double)kp / (double)H, &s1, &c1);
  sincospi(-2.0 * (double)kc / (double)H, &s2, &c2);
  double w1r = 1.0, w1i = 0.0, w2r = 1.0, w2i = 0.0;
  for (int r = 0; r < R; ++r) {
    const float2 a = Zr[(int64_t)r * Hp + kp], b = Zr[(int64_t)r * Hp + kc];
    ar[r] = (double)a.x * w1r - (double)a.y * w1i;
    ai[r] = (double)a.x * w1i + (double)a.y * w1r;
    br[r] = (double)b.x * w2r - (double)b.y * w2i;
    bi[r] = (double)b.x * w2i + (double)b.y * w2r;
    const double t1 = w1r * c1 - w1i * s1, t2 = w2r * c2 - w2i * s2;
    w1i = w1r * s1 + w1i * c1;
    w1r = t1;
    w2i = w2r * s2 + w2i * c2;
    w2r = t2;
  }
  const double hs = 0.5 * (double)scale;
  for (int q = 0; q < R; ++q) {
    const int qp = kp == 0 ? (R - q) & (R - 1) : R - 1 - q;                // the partner's q
    double zr = 0.0, zi = 0.0, pr = 0.0, pi_ = 0.0;
    for (int r = 0; r < R; ++r) {                                           // W_R^(r q): exact multiples of 1/R turns
      double sq, cq, sp, cp;
      sincospi(-2.0 * (double)((r * q) & (R - 1)) / (double)R, &sq, &cq);
      sincospi(-2.0 * (double)((r * qp) & (R - 1)) / (double)R, &sp, &cp);
      zr += ar[r] * cq - ai[r] * sq;
      zi += ar[r] * sq + ai[r] * cq;
      pr += br[r] * cp - bi[r] * sp;
      pi_ += br[r] * sp + bi[r] * cp;
    }
    // X[k] = (Z[k] + conj Z[H-k])/2 - i W_2H^k (Z[k] - conj Z[H-k])/2
    const int64_t k = kp + Hp * q;
    double sk, ck;
    sincospi(-(double)k / (double)H, &sk, &ck);
    const double evr = zr + pr, evi = zi - pi_, dr = zr - pr, di = zi + pi_;
    const double tr = ck * dr - sk * di, ti = ck * di + sk * dr;            // W^k (Z[k] - conj Z[H-k])
    const double xr = (evr + ti) * hs, xi = (evi - tr) * hs;
    if (MODE == 0) reinterpret_cast<float2*>(out)[row + k] = make_float2((float)xr, (float)xi);
    else out[row + k] = (float)(sqrt(xr * xr + xi * xi) + 1e-7);
    if (k == 0) {                                                           // bin H: X[H] = Re Z[0] - Im Z[0]
      const double xh = (zr - zi) * 2.0 * hs;
      if (MODE == 0) reinterpret_cast<float2*>(out)[row + H] = make_float2((float)xh, 0.0f);
      else out[row + H] = (float)(fabs(xh) + 1e-7);
    }
  }
}

// ---- X2: normalised cross-correlation and delay search (util/correlation.py:6-39) ---------------------------
// scipy.signal.correlate(a/|a|, b/|b|, 'full') through ONE complex transform of z = a + i b (both zero-padded to
// N >= len a + len b - 1): A_k = (Z_k + conj Z_{N-k})/2, B_k = (Z_k - conj Z_{N-k})/(2i), cross spectrum A conj(B),
// inverse transform by the conjugate trick.  The float32 transform places the peak (error ~1e-6 of it); find_delay
// then re-evaluates the few lags around it as float64 dot products, so the parabola through the peak -- the delay
// the tape-sync tool uses -- is exact.
__global__ __launch_bounds__(256) void k_xc_pack(const double* __restrict__ a, int64_t na, const double* __restrict__ b,
                                                 int64_t nb, float2* __restrict__ Z, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  Z[i] = make_float2(i < na ? (float)a[i] : 0.0f, i < nb ? (float)b[i] : 0.0f);
}
// conj(A_k conj(B_k)) for the second (forward) transform
__global__ __launch_bounds__(256) void k_xc_cross(const float2* __restrict__ Z, float2* __restrict__ C, int64_t N) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= N) return;
  const float2 zk = Z[k], zc = cconj(Z[(N - k) & (N - 1)]);
  const float2 A = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
  const float2 d = csub(zk, zc);                            // 2i B_k  ->  B_k = (d.y, -d.x)/2
  const float2 B = make_float2(0.5f * d.y, -0.5f * d.x);
  const float2 c = cmul(A, cconj(B));
  C[k] = cconj(c);
}
// full[j] = circ[(j - (nb-1)) mod N] / (N |a| |b|), j = 0 .. na + nb - 2
__global__ __launch_bounds__(256) void k_xc_unpack(const float2* __restrict__ Y, int64_t N, int64_t nb, int64_t n_full,
                                                   const double* __restrict__ norms, double* __restrict__ full) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n_full) return;
  full[j] = (double)Y[(j - (nb - 1)) & (N - 1)].x / ((double)N * norms[0] * norms[1]);
}
// sectioned form: full[j] += circ[(j - (nbq-1)) mod N] / (N |a| |b|) for the n_local lags of one section pair
__global__ __launch_bounds__(256) void k_xc_unpack_acc(const float2* __restrict__ Y, int64_t N, int64_t nbq, int64_t n_local,
                                                       const double* __restrict__ norms, double* __restrict__ full) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n_local) return;
  full[j] += (double)Y[(j - (nbq - 1)) & (N - 1)].x / ((double)N * norms[0] * norms[1]);
}
// norms[0] = |a|, norms[1] = |b| (float64, one workgroup each)
__global__ __launch_bounds__(1024) void k_xc_norms(const double* __restrict__ a, int64_t na, const double* __restrict__ b,
                                                   int64_t nb, double* __restrict__ norms) {
  __shared__ double part[16];
  const double* v = blockIdx.x ? b : a;
  const int64_t n = blockIdx.x ? nb : na;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) acc += v[i] * v[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += part[w];
    norms[blockIdx.x] = sqrt(t);
  }
}
// first index of the maximum of v[lo .. lo+n) (|v| when ignore_phase), one workgroup
__global__ __launch_bounds__(1024) void k_xc_argmax(const double* __restrict__ v, int64_t n, int use_abs, long long* __restrict__ arg) {
  __shared__ double bv[1024];
  __shared__ long long bi[1024];
  double best = -INFINITY;
  long long at = 0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double x = use_abs ? fabs(v[i]) : v[i];
    if (x > best) {
      best = x;
      at = i;
    }
  }
  bv[threadIdx.x] = best;
  bi[threadIdx.x] = at;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const double x = bv[threadIdx.x + o];
      const long long j = bi[threadIdx.x + o];
      if (x > bv[threadIdx.x] || (x == bv[threadIdx.x] && j < bi[threadIdx.x])) {
        bv[threadIdx.x] = x;
        bi[threadIdx.x] = j;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *arg = bi[0];
}
constexpr int kXcCand = 64;                        // rival peaks find_delay re-evaluates exactly
// Rival peaks: the float32 transform finds the peak to ~1e-6 of its height; a periodic tone, or the +/- lobes
// under ignore_phase, can put another local maximum within that distance of the top more than two lags away, and the
// reference's float64 argmax may pick that one.  Every local maximum (of |v| under use_abs) within `tol` of the
// transform's top, more than two lags from it, is listed (up to kXcCand of them) for exact re-evaluation.
__global__ __launch_bounds__(1024) void k_xc_candidates(const double* __restrict__ v, int64_t n, int use_abs,
                                                        const long long* __restrict__ arg, double tol,
                                                        long long* __restrict__ cand, int* __restrict__ n_cand) {
  const long long top = *arg;
  const double m = use_abs ? fabs(v[top]) : v[top];
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    if (i >= top - 2 && i <= top + 2) continue;
    const double x = use_abs ? fabs(v[i]) : v[i];
    if (!(x >= m - tol)) continue;
    const double xl = i > 0 ? (use_abs ? fabs(v[i - 1]) : v[i - 1]) : -INFINITY;
    const double xr = i + 1 < n ? (use_abs ? fabs(v[i + 1]) : v[i + 1]) : -INFINITY;
    if (x >= xl && x >= xr) {
      const int slot = atomicAdd(n_cand, 1);
      if (slot < kXcCand) cand[slot] = i;
    }
  }
}
// exact 'same' correlation value at every listed lag (one workgroup per entry, float64)
__global__ __launch_bounds__(256) void k_xc_exact_list(const double* __restrict__ a, int64_t na, const double* __restrict__ b,
                                                       int64_t nb, const double* __restrict__ norms,
                                                       const long long* __restrict__ list, double* __restrict__ vals) {
  __shared__ double part[4];
  const int64_t j = list[blockIdx.x];
  double acc = 0.0;
  if (j >= 0 && j < na) {
    const int64_t sh = j + (nb - 1) / 2 - (nb - 1);
    for (int64_t t = threadIdx.x; t < nb; t += 256) {
      const int64_t ia = t + sh;
      if (ia >= 0 && ia < na) acc += a[ia] * b[t];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) vals[blockIdx.x] = (part[0] + part[1] + part[2] + part[3]) / (norms[0] * norms[1]);
}
// exact 'same' correlation values at lags c0 .. c0+count-1 (one workgroup per lag, float64):
// same[j] = sum_t a[t + j + (nb-1)//2 - (nb-1)] b[t] / (|a| |b|)
__global__ __launch_bounds__(256) void k_xc_exact(const double* __restrict__ a, int64_t na, const double* __restrict__ b,
                                                  int64_t nb, const double* __restrict__ norms, const long long* __restrict__ arg,
                                                  int64_t rel0, double* __restrict__ vals) {
  __shared__ double part[4];
  const int64_t j = *arg + rel0 + blockIdx.x;
  double acc = 0.0;
  if (j >= 0 && j < na) {
    const int64_t sh = j + (nb - 1) / 2 - (nb - 1);
    for (int64_t t = threadIdx.x; t < nb; t += 256) {
      const int64_t ia = t + sh;
      if (ia >= 0 && ia < na) acc += a[ia] * b[t];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) vals[blockIdx.x] = (part[0] + part[1] + part[2] + part[3]) / (norms[0] * norms[1]);
}

// ISTFT stage 1: frame f -> window * irfft(spec[f] * sqrt(n_fft))   (util/fourier.py:359, :401)
// irfft of H+1 bins via an H-point complex inverse FFT (conjugate trick on the forward core).
template <int LOGH>
__global__ __launch_bounds__(FftGeom<LOGH>::Threads) void k_istft_frames(const float2* __restrict__ spec, int64_t n_frames,
                                                                          const float* __restrict__ window,
                                                                          const float2* __restrict__ tw,
                                                                          const float2* __restrict__ post,
                                                                          float* __restrict__ frames, float scale) {
  using G = FftGeom<LOGH>;
  constexpr int H = G::H, T = G::T, bins = H + 1;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const int f = threadIdx.x / T, j = threadIdx.x - f * T;
  float2* X = lds + f * G::FrameLds;
  const int64_t fr = (int64_t)blockIdx.x * G::Frames + f;
  const bool live = fr < n_frames;
  // conj(Z[k]) with Z[k] = E[k] + i*O[k],  E = (X[k]+conj(X[H-k]))/2,  O = conj(W^k)*(X[k]-conj(X[H-k]))/2
  // (numpy's irfft ignores the imaginary parts of the DC and Nyquist bins.)
  float2 v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int k = j + q * T;
    float2 z = make_float2(0.0f, 0.0f);
    if (live) {
      float2 a = spec[fr * bins + k];
      float2 b = spec[fr * bins + (H - k)];
      if (k == 0) {
        a.y = 0.0f;
        b.y = 0.0f;
      }
      b = cconj(b);
      const float2 ev = cadd(a, b);
      const float2 od = cmul(cconj(post[k]), csub(a, b));
      z = make_float2(0.5f * (ev.x - od.y), -0.5f * (ev.y + od.x));     // conj(0.5*(ev + i*od))
    }
    v[q] = z;
  }
  fft_core<LOGH>(v, X, j, tw);
  if (!live) return;
  // z_time[i] = conj(FFT(conj(Z)))[i] / H ; y[2i] = re, y[2i+1] = im
  for (int i = j; i < H; i += T) {
    const float2 r = X[lpad(i)];
    float2 o;
    o.x = r.x * scale * window[2 * i];
    o.y = -r.y * scale * window[2 * i + 1];
    reinterpret_cast<float2*>(frames)[fr * H + i] = o;
  }
}

// ISTFT of frames above 8192 points (r04): the irfft of one frame through the four-step complex transform K_stft uses for
// frames of that size (big_fft_c2c), `batch` frames at a time.
//   k_ibig_pack    A[b][k] = conj(Z[k]),  Z[k] = ((X[k] + conj X[H-k]) + i e^(i pi k / H) (X[k] - conj X[H-k])) / 2   (k < H)
//   big_fft_c2c    V = FFT_H(A);  z = conj(V) / H  is the packed frame: y[2 i] = Re z[i], y[2 i + 1] = Im z[i]
//   k_ibig_unpack  frames[f][n] = window[n] y[n] sqrt(n_fft)      (util/fourier.py:359, :401), then k_istft_ola as below
__global__ __launch_bounds__(256) void k_ibig_pack(const float2* __restrict__ spec, int64_t frame0, int64_t batch, int64_t H,
                                                   float2* __restrict__ A) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= batch * H) return;
  const int64_t b = t / H, k = t - b * H;
  const float2* X = spec + (frame0 + b) * (H + 1);
  float2 a = X[k], c = X[H - k];
  if (k == 0) {                                    // numpy's irfft ignores the imaginary parts of the DC and Nyquist bins
    a.y = 0.0f;
    c.y = 0.0f;
  }
  c = cconj(c);
  double sn, cs;
  sincospi((double)k / (double)H, &sn, &cs);       // e^(+i pi k / H)
  const float2 ev = cadd(a, c), df = csub(a, c);
  const float2 od = make_float2((float)(cs * df.x - sn * df.y), (float)(cs * df.y + sn * df.x));
  A[t] = make_float2(0.5f * (ev.x - od.y), -0.5f * (ev.y + od.x));          // conj((ev + i od) / 2)
}

__global__ __launch_bounds__(256) void k_ibig_unpack(const float2* __restrict__ V, int64_t frame0, int64_t batch, int64_t H,
                                                     const float* __restrict__ window, float* __restrict__ frames, float scale) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= batch * H) return;
  const int64_t b = t / H, i = t - b * H;
  const float2 r = V[t];
  reinterpret_cast<float2*>(frames)[(frame0 + b) * H + i] = make_float2(r.x * scale * window[2 * i], -r.y * scale * window[2 * i + 1]);
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

// Fused ISTFT: the [n_frames][n_fft] time-frame array never exists in HBM (at hop = n_fft/16 it is 16x the signal,
// written once and read once by the two-kernel path).  A workgroup owns NF = Frames*s consecutive frames,
// s = ceil(n_fft/hop) being the spacing at which frames stop overlapping.  Round r = 0..s-1 inverse-transforms the
// Frames frames F0 + r + u*s: they are pairwise disjoint in time, so each frame's own lanes add it straight into
// the LDS overlap-add accumulator with plain read-modify-writes; one workgroup barrier per round orders the
// rounds.  The summation order per sample is fixed (by round), so results are reproducible run to run.
// The workgroup writes the (NF - s + 1)*hop samples that only its own frames cover; the s-1 frames on either side
// are transformed by the neighbour as well ((s-1)/NF extra work: 12 % at 512/32).
__host__ __device__ constexpr bool istft_tables_in_lds(int logh) { return logh <= 9; }

template <int LOGH>
__global__ __launch_bounds__(FftGeom<LOGH>::Threads) __attribute__((amdgpu_waves_per_eu(LOGH <= 9 ? 3 : 1))) void k_istft_fused(const float2* __restrict__ spec, int64_t n_frames,
                                                                         int hop, int s, const float* __restrict__ window,
                                                                         const float2* __restrict__ tw,
                                                                         const float2* __restrict__ post,
                                                                         float* __restrict__ y, int64_t y_len, int64_t skip,
                                                                         float scale) {
  using G = FftGeom<LOGH>;
  constexpr int H = G::H, T = G::T, bins = H + 1, n_fft = 2 * H;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  float* acc = reinterpret_cast<float*>(lds + G::Frames * G::FrameLds);
  const int tid = threadIdx.x;
  const int u = tid / T, j = tid - u * T;
  float2* X = lds + u * G::FrameLds;
  const int NF = G::Frames * s;
  const int out_frames = NF - s + 1;
  const int acc_len = (NF - 1) * hop + n_fft;
  const int64_t F0 = (int64_t)blockIdx.x * out_frames - (s - 1);      // first frame of the workgroup (may be < 0)
  for (int i = tid; i < acc_len; i += G::Threads) acc[i] = 0.0f;
  // window taps and untangle twiddles are the same for every frame: they live in LDS, not in 32 registers per lane
  // (n_fft <= 1024; the larger transforms keep their LDS for the overlap-add span and read the tables from memory)
  constexpr bool kTab = istft_tables_in_lds(LOGH);
  float* wl = acc + acc_len;                                           // [n_fft]
  float2* pl = reinterpret_cast<float2*>(wl + n_fft);                  // [H] conj(post)
  if (kTab) {
    for (int i = tid; i < n_fft; i += G::Threads) wl[i] = window[i];
    for (int i = tid; i < H; i += G::Threads) pl[i] = cconj(post[i]);
    __syncthreads();
  }
  // Occupancy hides the spectrum fetch: with the tables in LDS and no register double buffer the kernel fits 3 waves
  // per SIMD (146 VGPRs at n_fft = 512; the double-buffered form needed 216 = 2 waves and was 16 % slower, measured).
  float2 pa[8], pb[8], pk[8], wq[8];        // pk / wq: the tables in registers, large transforms only
  if (!kTab) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      pk[q] = cconj(post[j + q * T]);
      wq[q] = *reinterpret_cast<const float2*>(window + 2 * (j + q * T));
    }
  }
  // A frame outside [0, n_frames) is transformed like any other and dropped at the overlap-add (`live` below): its rows
  // only have to be readable, so the row index is clamped instead of selecting zeros element by element (32 v_cndmask per
  // round).  Whatever the clamped row holds -- NaNs included -- never reaches the accumulator.
  auto fetch = [&](int r) {
    if (r >= s) return;
    int64_t fr = F0 + r + u * s;
    fr = fr < 0 ? 0 : (fr >= n_frames ? n_frames - 1 : fr);
    const float2* row = spec + fr * bins;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = j + q * T;
      pa[q] = row[k];
      pb[q] = row[H - k];
    }
  };
  if (!kTab) fetch(0);                       // large transforms run at 2 waves per SIMD either way: keep their register
  for (int r = 0; r < s; ++r) {              // double buffer (round r+1 in flight under round r's butterflies)
    if (kTab) fetch(r);
    const int lf = r + u * s;                                         // frame index inside the workgroup
    const int64_t fr = F0 + lf;
    const bool live = fr >= 0 && fr < n_frames;
    float2 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = j + q * T;
      float2 a = pa[q], b = pb[q];
      if (k == 0) {
        a.y = 0.0f;
        b.y = 0.0f;
      }
      b = cconj(b);
      const float2 ev = cadd(a, b);
      const float2 od = cmul(kTab ? pl[k] : pk[q], csub(a, b));
      v[q] = make_float2(0.5f * (ev.x - od.y), -0.5f * (ev.y + od.x));   // zeros stay zeros for a dead frame
    }
    if (!kTab) fetch(r + 1);
    fft_core<LOGH>(v, X, j, tw);
    __syncthreads();                                                  // the previous round's adds (and the zeroing) are done
    if (live) {
      float* dst = acc + lf * hop;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = j + q * T;
        const float2 z = X[lpad(i)];                                  // conj(FFT(conj Z)): y[2i] = re, y[2i+1] = -im
        const float2 wv = kTab ? *reinterpret_cast<const float2*>(wl + 2 * i) : wq[q];
        const float a0 = z.x * scale * wv.x, a1 = -z.y * scale * wv.y;
        if ((hop & 1) == 0) {
          float2 t = *reinterpret_cast<float2*>(dst + 2 * i);
          t.x += a0;
          t.y += a1;
          *reinterpret_cast<float2*>(dst + 2 * i) = t;
        } else {
          dst[2 * i] += a0;
          dst[2 * i + 1] += a1;
        }
      }
    }
  }
  __syncthreads();
  if (!kTab) {
    wl = reinterpret_cast<float*>(lds);                                // frame slots are free now: stage the window there
    for (int i = tid; i < n_fft; i += G::Threads) wl[i] = window[i];
    __syncthreads();
  }
  const int64_t ola_len = (int64_t)n_fft + (int64_t)hop * (n_frames - 1);
  const int out_len = out_frames * hop;
  const int64_t TT0 = (int64_t)blockIdx.x * out_len;                  // overlap-add coordinate of the first owned sample
  // Window-sumsquare envelope.  Away from the two ends of the signal every frame that can cover a sample exists, and the
  // envelope depends on the sample's phase TT mod hop alone: one table of hop sums per workgroup (same order of terms
  // as the general loop below: frames ascending = window offsets descending) replaces two 64-bit divisions and a
  // 16-term loop PER SAMPLE (they were a third of the kernel's instructions at 512/32).
  float* env_tab = wl + n_fft;          // kTab: behind the window (pl is dead now); else behind the staged window in the frame slots
  const bool use_tab = kTab || n_fft + hop <= 2 * G::Frames * G::FrameLds;
  if (use_tab) {
    for (int ph = tid; ph < hop; ph += G::Threads) {
      float env = 0.0f;
      for (int o = ph + ((n_fft - 1 - ph) / hop) * hop; o >= 0; o -= hop) env += wl[o] * wl[o];
      env_tab[ph] = env;
    }
    __syncthreads();
  }
  const int64_t interior_lo = n_fft - 1, interior_hi = (int64_t)hop * (n_frames - 1);     // all covering frames exist in [lo, hi]
  const int step_ph = G::Threads % hop;
  int ph = tid % hop;                                                  // TT0 is a multiple of hop: phase of TT = phase of p
  for (int p = tid; p < out_len; p += G::Threads, ph = ph + step_ph >= hop ? ph + step_ph - hop : ph + step_ph) {
    const int64_t TT = TT0 + p, t = TT - skip;
    if (t < 0) continue;
    if (t >= y_len) break;
    float a = 0.0f;
    if (TT < ola_len) {
      float env;
      if (use_tab && TT >= interior_lo && TT <= interior_hi) {
        env = env_tab[ph];
      } else {
        int64_t e_hi = TT / hop;
        if (e_hi > n_frames - 1) e_hi = n_frames - 1;
        const int64_t e_lo = (TT - n_fft + 1 <= 0) ? 0 : (TT - n_fft + hop) / hop;
        env = 0.0f;
        for (int64_t e = e_lo; e <= e_hi; ++e) {
          const float w = wl[TT - e * hop];
          env += w * w;
        }
      }
      a = acc[p + (s - 1) * hop];
      if (env > 1.17549435e-38f) a /= env;                           // > tiny(float32)  (:414-415)
    }
    y[t] = a;
  }
}

// bytes of LDS the fused ISTFT needs for (n_fft, hop); it is used when they fit 64 KB
template <int LOGH>
static inline size_t istft_fused_lds(int hop) {
  using G = FftGeom<LOGH>;
  const int64_t n_fft = 2 * G::H, s = (n_fft + hop - 1) / hop;
  return (size_t)G::Frames * G::FrameLds * sizeof(float2) + (size_t)((G::Frames * s - 1) * hop + n_fft) * sizeof(float) +
         (istft_tables_in_lds(LOGH) ? (size_t)n_fft * sizeof(float) + (size_t)G::H * sizeof(float2) : 0);   // + the tables
}
static inline size_t istft_fused_lds_any(int n_fft, int hop) {
  switch (n_fft) {
    case 16: return istft_fused_lds<3>(hop);
    case 32: return istft_fused_lds<4>(hop);
    case 64: return istft_fused_lds<5>(hop);
    case 128: return istft_fused_lds<6>(hop);
    case 256: return istft_fused_lds<7>(hop);
    case 512: return istft_fused_lds<8>(hop);
    case 1024: return istft_fused_lds<9>(hop);
    case 2048: return istft_fused_lds<10>(hop);
    case 4096: return istft_fused_lds<11>(hop);
    case 8192: return istft_fused_lds<12>(hop);
  }
  return (size_t)-1;
}
// Largest n_fft that takes the fused kernel.  A workgroup re-transforms s - 1 = n_fft/hop - 1 frames of its neighbour; with
// 8 or 4 frames side by side (n_fft <= 1024) that is 13-23 % extra work for never storing the frames, at 2048 (2 frames)
// the two forms are within 9 %, and a 4096-point frame fills the workgroup alone: s frames transformed per hop of
// output (tools/bench_istft_sweep.py, 23 M samples: 4096 at hop n/4, n/8, n/16 took 0.78 / 2.44 / 8.65 ms fused against
// 0.37 / 0.68 / 1.27 ms through the frame array).
#ifndef PAR_ISTFT_FUSE_MAX
#define PAR_ISTFT_FUSE_MAX 2048
#endif
static inline bool istft_is_fused(int n_fft, int hop) {
  return hop >= 1 && n_fft <= PAR_ISTFT_FUSE_MAX && istft_fused_lds_any(n_fft, hop) <= 64 * 1024;
}

}  // namespace par

extern "C" {

// frames per four-step batch of the big ISTFT: two complex arrays of batch x H points beside the frame array
static int64_t istft_big_batch(int64_t n_frames, int n_fft) {
  const int64_t H = n_fft / 2, want = (1ll << 22) / H;
  const int64_t b = want < 1 ? 1 : want;
  return b < n_frames ? b : n_frames;
}

int64_t par_istft_scratch_floats(int64_t n_frames, int n_fft, int hop) {
  if (n_frames < 1 || n_fft < 1 || hop < 1) return 0;
  if (n_fft > 8192) return n_frames * (int64_t)n_fft + 4 * istft_big_batch(n_frames, n_fft) * (int64_t)(n_fft / 2) + 16;
  return par::istft_is_fused(n_fft, hop) ? 0 : n_frames * (int64_t)n_fft;
}

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
                 const float* window, float* out, int mode, int64_t out_pitch, void* stream) {
  using namespace par;
  PAR_REQUIRE(x && window && out, PAR_ERR_ARG, "par_stft_f32: null pointer");
  const int64_t pitch = out_pitch ? out_pitch : (int64_t)n_fft * zeropad / 2 + 1;     // elements (float or float2) between rows
  PAR_REQUIRE(pitch >= (int64_t)n_fft * zeropad / 2 + 1, PAR_ERR_ARG, "par_stft_f32: out_pitch %lld < bins", (long long)out_pitch);
  PAR_REQUIRE(n >= 1 && x_stride >= 1 && hop >= 1 && zeropad >= 1 && n_fft >= 2, PAR_ERR_ARG, "par_stft_f32: bad sizes");
  PAR_REQUIRE(mode == 0 || mode == 1, PAR_ERR_ARG, "par_stft_f32: mode must be 0 (complex) or 1 (magnitude)");
  const int64_t M64 = (int64_t)n_fft * zeropad;
  PAR_REQUIRE(M64 >= 16 && M64 <= 16384 && (M64 & (M64 - 1)) == 0 && (n_fft % 2) == 0, PAR_ERR_UNSUPPORTED,
              "par_stft_f32: n_fft*zeropad=%lld is not a power of two in [16, 16384]", (long long)M64);
  const int M = (int)M64, H = M / 2;
  PAR_HIP_CHECK(hipSetDevice(device));
  Twiddles tw;
  int rc = get_twiddles(device, M, &tw);
  if (rc != PAR_OK) return rc;
  const int64_t n_frames = par_stft_frames(n, n_fft, hop);
  const float scale = (float)(1.0 / sqrt((double)n_fft));
#define PAR_STFT_LAUNCH_MU(LH, MD, UN)                                                                              \
  hipLaunchKernelGGL((k_stft<LH, MD, UN>), dim3((unsigned)(ceil_div(ceil_div(n_frames, FftGeom<LH>::Frames), 8) * 8)), dim3(FftGeom<LH>::Threads),  \
                     (size_t)FftGeom<LH>::Frames * FftGeom<LH>::FrameLds * sizeof(float2) +                              \
                         (PAR_STFT_STORE == 3 ? (size_t)FftGeom<LH>::Frames * ((1 << LH) + 4) * sizeof(float) : 0),      \
                     as_stream(stream), x, n, x_stride, n_fft, hop, window, tw.w, tw.post, out, n_frames, scale, pitch)
#define PAR_STFT_LAUNCH(LH)                                                                                          \
  do {                                                                                                               \
    if (mode == 1 && x_stride == 1) PAR_STFT_LAUNCH_MU(LH, 1, true);                                                 \
    else if (mode == 1) PAR_STFT_LAUNCH_MU(LH, 1, false);                                                            \
    else if (x_stride == 1) PAR_STFT_LAUNCH_MU(LH, 0, true);                                                         \
    else PAR_STFT_LAUNCH_MU(LH, 0, false);                                                                           \
  } while (0)
  switch (ilog2(H)) {
    case 3: PAR_STFT_LAUNCH(3); break;
    case 4: PAR_STFT_LAUNCH(4); break;
    case 5: PAR_STFT_LAUNCH(5); break;
    case 6: PAR_STFT_LAUNCH(6); break;
    case 7: PAR_STFT_LAUNCH(7); break;
    case 8: PAR_STFT_LAUNCH(8); break;
    case 9: PAR_STFT_LAUNCH(9); break;
    case 10: PAR_STFT_LAUNCH(10); break;
    case 11: PAR_STFT_LAUNCH(11); break;
    case 12: PAR_STFT_LAUNCH(12); break;
    case 13: {
      // 16384 points (the GUI's 4096 x zero-padding 4): one 1024-lane workgroup per frame, 72 KB of the CU's 160 KB of LDS
      // (above the 64 KB a kernel gets by default: raised per kernel once)
      constexpr int kLds13 = FftGeom<13>::Frames * FftGeom<13>::FrameLds * (int)sizeof(float2) + (PAR_STFT_STORE == 3 ? (8192 + 4) * 4 : 0);
      for (const void* fn : {reinterpret_cast<const void*>(&k_stft<13, 0, true>), reinterpret_cast<const void*>(&k_stft<13, 0, false>),
                             reinterpret_cast<const void*>(&k_stft<13, 1, true>), reinterpret_cast<const void*>(&k_stft<13, 1, false>)}) {
        const int rc = raise_dynamic_lds(fn, kLds13, device);
        if (rc != PAR_OK) return rc;
      }
      PAR_STFT_LAUNCH(13);
      break;
    }
    default: PAR_REQUIRE(false, PAR_ERR_UNSUPPORTED, "par_stft_f32: unsupported size");
  }
#undef PAR_STFT_LAUNCH
#undef PAR_STFT_LAUNCH_MU
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

// Frames of more than 8192 points (up to 2^21): four-step transform through a caller-owned scratch of
// par_stft_big_scratch_bytes() (one H-point complex array per frame of a batch; a batch fills up to 1 GiB).
size_t par_stft_big_scratch_bytes(int64_t n, int n_fft, int hop, int zeropad) {
  const int64_t M = (int64_t)n_fft * zeropad;
  if (M > (1ll << 21) && M <= (1ll << 24)) return (size_t)(2 * (M / 2) * (int64_t)sizeof(float2));   // z_r and Z_r of ONE frame
  if (M <= 8192 || M > (1ll << 21)) return 0;
  // frames per batch: as many as fit 1 GiB of scratch (at least 16, at most what one grid dimension takes): 16-frame
  // batches made the 16384-point transform launch-bound (880 batches of three small launches: 16 ms for a 10-minute file)
  int64_t frames = par_stft_frames(n, n_fft, hop);
  const int64_t per_frame = (M / 2) * (int64_t)sizeof(float2);       // A[k1][n2]; the row pass writes the bins itself
  int64_t cap = (1ll << 30) / per_frame;
  cap = cap < 16 ? 16 : (cap > 32768 ? 32768 : cap);
  if (frames > cap) frames = cap;
  return (size_t)(frames * per_frame);
}

int par_stft_big_f32(int device, const float* x, int64_t n, int64_t x_stride, int n_fft, int hop, int zeropad,
                     const float* window, float* out, int mode, void* scratch, size_t scratch_bytes, void* stream) {
  using namespace par;
  PAR_REQUIRE(x && window && out && scratch, PAR_ERR_ARG, "par_stft_big_f32: null pointer");
  PAR_REQUIRE(n >= 1 && x_stride >= 1 && hop >= 1 && zeropad >= 1 && n_fft >= 2, PAR_ERR_ARG, "par_stft_big_f32: bad sizes");
  PAR_REQUIRE(mode == 0 || mode == 1, PAR_ERR_ARG, "par_stft_big_f32: mode must be 0 (complex) or 1 (magnitude)");
  const int64_t M64 = (int64_t)n_fft * zeropad;
  PAR_REQUIRE(M64 > 8192 && M64 <= (1ll << 24) && (M64 & (M64 - 1)) == 0 && (n_fft % 2) == 0, PAR_ERR_UNSUPPORTED,
              "par_stft_big_f32: n_fft*zeropad=%lld is not a power of two in (8192, 2^24]", (long long)M64);
  PAR_REQUIRE(scratch_bytes >= par_stft_big_scratch_bytes(n, n_fft, hop, zeropad), PAR_ERR_WORKSPACE,
              "par_stft_big_f32: scratch %zu < %zu", scratch_bytes, par_stft_big_scratch_bytes(n, n_fft, hop, zeropad));
  PAR_HIP_CHECK(hipSetDevice(device));
  hipStream_t s = as_stream(stream);
  const int64_t H = M64 / 2;
  if (M64 > (1ll << 21)) {
    // 2^22 .. 2^24 points: R = H / 2^20 decimated sequences of 2^20 points through the four-step transform, recombined and
    // untangled bin by bin (k_huge_gather / k_huge_out), a frame at a time
    const int64_t Hp = 1ll << 20;
    int logR = 0;
    while ((Hp << logR) < H) ++logR;
    float2* z = static_cast<float2*>(scratch);
    float2* Z = z + H;
    const int64_t frames = par_stft_frames(n, n_fft, hop);
    const float scale_h = (float)(1.0 / sqrt((double)n_fft));
    for (int64_t f = 0; f < frames; ++f) {
      hipLaunchKernelGGL(k_huge_gather, dim3((unsigned)ceil_div(H, 256)), dim3(256), 0, s, x, n, x_stride, n_fft, hop, window, f, logR,
                         Hp, z);
      const int rc = big_fft_c2c(device, z, Z, Hp, (int64_t)1 << logR, s);
      if (rc != PAR_OK) return rc;
      if (mode == 0)
        hipLaunchKernelGGL(k_huge_out<0>, dim3((unsigned)ceil_div(Hp, 256)), dim3(256), 0, s, (const float2*)Z, logR, Hp, out,
                           f * (H + 1), scale_h);
      else
        hipLaunchKernelGGL(k_huge_out<1>, dim3((unsigned)ceil_div(Hp, 256)), dim3(256), 0, s, (const float2*)Z, logR, Hp, out,
                           f * (H + 1), scale_h);
    }
    PAR_HIP_CHECK(hipGetLastError());
    return PAR_OK;
  }
  // N1 >= N2, both in [64, 1024]: the balanced split, except 2^15 = 512 x 64 -- both pure radix-8 sizes (five radix-8 stages
  // instead of 256 x 128's four plus a radix-4 and a radix-2 stage) -- and 2^14 = 256 x 64 (one remainder stage instead of two)
  const int L = ilog2((int)H), l1 = L == 15 ? 9 : (L == 14 ? 8 : (L + 1) / 2), l2 = L - l1;
  Twiddles t1, t2;
  BigTw bt;
  int rc = get_twiddles(device, 2 << l1, &t1);
  if (rc == PAR_OK) rc = get_twiddles(device, 2 << l2, &t2);
  BigTw bt2;                                                              // the untangle step's half-step circle
  if (rc == PAR_OK) rc = get_big_twiddles(device, (int)H, &bt);
  if (rc == PAR_OK) rc = get_big_twiddles(device, (int)(2 * H), &bt2);
  if (rc != PAR_OK) return rc;
  const int64_t n_frames = par_stft_frames(n, n_fft, hop);
  const float scale = (float)(1.0 / sqrt((double)n_fft));
  float2* A = static_cast<float2*>(scratch);
  // frames the scratch holds: one H-point array per frame
  int64_t batch = (int64_t)(scratch_bytes / (size_t)(H * (int64_t)sizeof(float2)));
  batch = batch > 32768 ? 32768 : batch;
  for (int64_t f0 = 0; f0 < n_frames; f0 += batch) {
    const int64_t nb = n_frames - f0 < batch ? n_frames - f0 : batch;
#define PAR_BIG_COLS(LS, NG, TW)                                                                                         \
  if (int rc_lds = raise_dynamic_lds(reinterpret_cast<const void*>(&k_bigfft<LS, 0>),                                    \
                                     kBigC * ((1 << LS) + (1 << LS) / 8 + 8) * (int)sizeof(float2), device))             \
    return rc_lds;                                                                                                       \
  hipLaunchKernelGGL((k_bigfft<LS, 0>), dim3((unsigned)round_up8((int64_t)((NG) / kBigC) * nb)), dim3(1 << LS),           \
                     (size_t)kBigC * ((1 << LS) + (1 << LS) / 8 + 8) * sizeof(float2), s, x, n, x_stride, n_fft, hop,     \
                     window, TW, bt.lo, bt.hi, A, (float2*)nullptr, f0, l1, l2, nb)
    switch (l1) {
      case 7: PAR_BIG_COLS(7, 1 << l2, t1.w); break;
      case 8: PAR_BIG_COLS(8, 1 << l2, t1.w); break;
      case 9: PAR_BIG_COLS(9, 1 << l2, t1.w); break;
      case 10: PAR_BIG_COLS(10, 1 << l2, t1.w); break;
      default: PAR_REQUIRE(false, PAR_ERR_UNSUPPORTED, "par_stft_big_f32: unsupported size");
    }
#undef PAR_BIG_COLS
    // rows + untangle: 16 padded spectra in LDS (148 KB at 1024 points: above the 64 KB default)
#define PAR_BIG_ROWS(LS)                                                                                                 \
  {                                                                                                                      \
    constexpr int kLds = 2 * kBigC * ((1 << LS) + (1 << LS) / 8 + 8) * (int)sizeof(float2);                              \
    int rc_lds = raise_dynamic_lds(reinterpret_cast<const void*>(&k_bigfft_rows_out<LS, 0>), kLds, device);              \
    if (rc_lds == PAR_OK) rc_lds = raise_dynamic_lds(reinterpret_cast<const void*>(&k_bigfft_rows_out<LS, 1>), kLds, device); \
    if (rc_lds != PAR_OK) return rc_lds;                                                                                 \
    const dim3 grid((unsigned)round_up8((int64_t)((1 << l1) / (2 * kBigC) + 1) * nb));                                   \
    if (mode == 0)                                                                                                       \
      hipLaunchKernelGGL((k_bigfft_rows_out<LS, 0>), grid, dim3(1 << LS), kLds, s, (const float2*)A, t2.w, bt2.lo, bt2.hi, out, \
                         f0, l1, scale, nb);                                                                               \
    else                                                                                                                 \
      hipLaunchKernelGGL((k_bigfft_rows_out<LS, 1>), grid, dim3(1 << LS), kLds, s, (const float2*)A, t2.w, bt2.lo, bt2.hi, out, \
                         f0, l1, scale, nb);                                                                               \
  }
    switch (l2) {
      case 6: PAR_BIG_ROWS(6); break;
      case 7: PAR_BIG_ROWS(7); break;
      case 8: PAR_BIG_ROWS(8); break;
      case 9: PAR_BIG_ROWS(9); break;
      case 10: PAR_BIG_ROWS(10); break;
      default: PAR_REQUIRE(false, PAR_ERR_UNSUPPORTED, "par_stft_big_f32: unsupported size");
    }
#undef PAR_BIG_ROWS
    PAR_HIP_CHECK(hipGetLastError());
  }
  return PAR_OK;
}

// X2: xcorr(a, b, 'full') of two float64 device signals (util/correlation.py:6-13): full[na + nb - 1] float64 out.
// scratch: par_xcorr_scratch_bytes(na, nb) device bytes.  The values carry the float32 transform's error (~1e-6 of the
// peak); par_find_delay_f64 refines what the tape-sync tool needs exactly.
constexpr int64_t kXcMaxFft = 1ll << 20;         // the four-step transform's largest size
constexpr int64_t kXcSection = kXcMaxFft / 2;    // longer signals are correlated section pair by section pair
constexpr int64_t kXcMaxLen = 1ll << 25;         // per signal (4096 section pairs of two 2^20-point transforms each)
static int64_t xc_fft_len(int64_t na, int64_t nb) {
  int64_t N = 8192;
  while (N < na + nb - 1 && N < kXcMaxFft) N <<= 1;
  return N;
}
size_t par_xcorr_scratch_bytes(int64_t na, int64_t nb) {
  if (na < 1 || nb < 1) return 0;
  return (size_t)(2 * xc_fft_len(na, nb) * sizeof(float2) + 64 + (size_t)(na + nb + 16 + 2 * par::kXcCand + 8) * sizeof(double));
}

static int xcorr_full(int device, const double* a, int64_t na, const double* b, int64_t nb, void* scratch, double* full,
                      double** norms_out, hipStream_t s) {
  using namespace par;
  const int64_t N = xc_fft_len(na, nb);
  PAR_REQUIRE(na <= kXcMaxLen && nb <= kXcMaxLen, PAR_ERR_UNSUPPORTED, "xcorr: %lld / %lld samples (limit 2^25 per signal)",
              (long long)na, (long long)nb);
  float2* Z = static_cast<float2*>(scratch);
  float2* Y = Z + N;
  double* norms = reinterpret_cast<double*>(Y + N);
  hipLaunchKernelGGL(k_xc_norms, dim3(2), dim3(1024), 0, s, a, na, b, nb, norms);
  if (na + nb - 1 > N) {
    // Windows of more than ~2.7 s at 192 kHz: correlation is bilinear, so the signals are cut into sections of 2^19
    // samples and every pair (p, q) adds its 2^20-point correlation into full at the lag offset p - q sections.  The
    // float32 transforms only have to FIND the peak; par_find_delay_f64 re-evaluates the lags around it exactly.
    PAR_HIP_CHECK(hipMemsetAsync(full, 0, (size_t)(na + nb - 1) * sizeof(double), s));
    for (int64_t pa = 0; pa < na; pa += kXcSection) {
      const int64_t la = na - pa < kXcSection ? na - pa : kXcSection;
      for (int64_t qb = 0; qb < nb; qb += kXcSection) {
        const int64_t lb = nb - qb < kXcSection ? nb - qb : kXcSection;
        hipLaunchKernelGGL(k_xc_pack, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, s, a + pa, la, b + qb, lb, Z, N);
        int rc = big_fft_c2c(device, Z, Y, N, 1, s);
        if (rc != PAR_OK) return rc;
        hipLaunchKernelGGL(k_xc_cross, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, s, (const float2*)Y, Z, N);
        rc = big_fft_c2c(device, Z, Y, N, 1, s);
        if (rc != PAR_OK) return rc;
        // local index jl = lag + lb - 1; global index = lag + (pa - qb) + nb - 1
        const int64_t joff = pa - qb + nb - lb;
        hipLaunchKernelGGL(k_xc_unpack_acc, dim3((unsigned)ceil_div(la + lb - 1, 256)), dim3(256), 0, s, (const float2*)Y, N, lb,
                           la + lb - 1, (const double*)norms, full + joff);
      }
    }
    PAR_HIP_CHECK(hipGetLastError());
    *norms_out = norms;
    return PAR_OK;
  }
  hipLaunchKernelGGL(k_xc_pack, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, s, a, na, b, nb, Z, N);
  int rc = big_fft_c2c(device, Z, Y, N, 1, s);                       // Y = FFT(a + i b)
  if (rc != PAR_OK) return rc;
  hipLaunchKernelGGL(k_xc_cross, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, s, (const float2*)Y, Z, N);
  rc = big_fft_c2c(device, Z, Y, N, 1, s);                           // Y = conj(N * circular correlation)
  if (rc != PAR_OK) return rc;
  hipLaunchKernelGGL(k_xc_unpack, dim3((unsigned)ceil_div(na + nb - 1, 256)), dim3(256), 0, s, (const float2*)Y, N, nb,
                     na + nb - 1, (const double*)norms, full);
  PAR_HIP_CHECK(hipGetLastError());
  *norms_out = norms;
  return PAR_OK;
}

int par_xcorr_f64(int device, const double* a, int64_t na, const double* b, int64_t nb, void* scratch, size_t scratch_bytes,
                  double* full, void* stream) {
  using namespace par;
  PAR_REQUIRE(a && b && scratch && full && na >= 1 && nb >= 1, PAR_ERR_ARG, "par_xcorr_f64: bad args");
  PAR_REQUIRE(scratch_bytes >= par_xcorr_scratch_bytes(na, nb), PAR_ERR_WORKSPACE, "par_xcorr_f64: scratch too small");
  PAR_HIP_CHECK(hipSetDevice(device));
  double* norms;
  return xcorr_full(device, a, na, b, nb, scratch, full, &norms, as_stream(stream));
}

// find_delay(a, b, ignore_phase) of util/correlation.py:16-39 (windows are applied by the caller, in place, like the
// reference): *delay = parabola-refined peak of the 'same' correlation minus len//2, *corr = its height.  The peak is
// located by the float32 transform and the lags around it are re-evaluated as float64 dot products.  A peak on the last
// lag is the reference's IndexError (PAR_ERR_INDEX).  Synchronises.
int par_find_delay_f64(int device, const double* a, int64_t na, const double* b, int64_t nb, int ignore_phase, void* scratch,
                       size_t scratch_bytes, double* delay, double* corr, void* stream) {
  using namespace par;
  PAR_REQUIRE(a && b && scratch && delay && corr && na >= 3 && nb >= 1, PAR_ERR_ARG, "par_find_delay_f64: bad args");
  PAR_REQUIRE(scratch_bytes >= par_xcorr_scratch_bytes(na, nb), PAR_ERR_WORKSPACE, "par_find_delay_f64: scratch too small");
  PAR_HIP_CHECK(hipSetDevice(device));
  hipStream_t s = as_stream(stream);
  const int64_t N = xc_fft_len(na, nb);
  double* full = reinterpret_cast<double*>(static_cast<char*>(scratch) + 2 * N * sizeof(float2) + 64);
  double* norms;
  int rc = xcorr_full(device, a, na, b, nb, scratch, full, &norms, s);
  if (rc != PAR_OK) return rc;
  long long* arg = reinterpret_cast<long long*>(norms + 2);
  double* vals = full + (na + nb - 1);                                 // 7 values behind the correlation (scratch holds 16 spare)
  const double* same = full + (nb - 1) / 2;                            // 'same': na values centred on the full output
  hipLaunchKernelGGL(k_xc_argmax, dim3(1), dim3(1024), 0, s, same, na, ignore_phase, arg);
  {
    // rival peaks within the float32 transform's error of the top: exact values decide, the lowest lag wins a tie
    // (np.argmax returns the first maximum)
    long long* cand = reinterpret_cast<long long*>(vals + 8);
    double* cvals = vals + 8 + kXcCand;
    int* n_cand = reinterpret_cast<int*>(cvals + kXcCand);
    PAR_HIP_CHECK(hipMemsetAsync(n_cand, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_xc_candidates, dim3(1), dim3(1024), 0, s, same, na, ignore_phase, (const long long*)arg, 4.0e-6, cand, n_cand);
    int nc = 0;
    long long top = 0;
    PAR_HIP_CHECK(hipMemcpyAsync(&nc, n_cand, sizeof(int), hipMemcpyDeviceToHost, s));
    PAR_HIP_CHECK(hipMemcpyAsync(&top, arg, sizeof(top), hipMemcpyDeviceToHost, s));
    PAR_HIP_CHECK(hipStreamSynchronize(s));
    if (nc > 0 && nc < kXcCand) {                        // the top goes into slot nc: nc + 1 entries must fit the kXcCand slots;
                                                         // as many rivals as slots or more: a plateau, the transform's top stands
      long long lags[kXcCand + 1];
      double ex[kXcCand + 1];
      PAR_HIP_CHECK(hipMemcpyAsync(lags, cand, nc * sizeof(long long), hipMemcpyDeviceToHost, s));
      PAR_HIP_CHECK(hipStreamSynchronize(s));
      lags[nc] = top;
      PAR_HIP_CHECK(hipMemcpyAsync(cand, lags, (nc + 1) * sizeof(long long), hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(k_xc_exact_list, dim3((unsigned)(nc + 1)), dim3(256), 0, s, a, na, b, nb, (const double*)norms,
                         (const long long*)cand, cvals);
      PAR_HIP_CHECK(hipMemcpyAsync(ex, cvals, (nc + 1) * sizeof(double), hipMemcpyDeviceToHost, s));
      PAR_HIP_CHECK(hipStreamSynchronize(s));
      int best = nc;
      for (int c = 0; c < nc; ++c) {
        const double x = ignore_phase ? fabs(ex[c]) : ex[c], y = ignore_phase ? fabs(ex[best]) : ex[best];
        if (x > y || (x == y && lags[c] < lags[best])) best = c;
      }
      if (lags[best] != top) PAR_HIP_CHECK(hipMemcpyAsync(arg, &lags[best], sizeof(long long), hipMemcpyHostToDevice, s));
    }
  }
  // exact values at arg-3 .. arg+3; the final peak is the best of arg-2 .. arg+2
  hipLaunchKernelGGL(k_xc_exact, dim3(7), dim3(256), 0, s, a, na, b, nb, (const double*)norms, (const long long*)arg, (int64_t)-3,
                     vals);
  PAR_HIP_CHECK(hipGetLastError());
  long long p0 = 0;
  double v[7];
  PAR_HIP_CHECK(hipMemcpyAsync(&p0, arg, sizeof(p0), hipMemcpyDeviceToHost, s));
  PAR_HIP_CHECK(hipMemcpyAsync(v, vals, sizeof(v), hipMemcpyDeviceToHost, s));
  PAR_HIP_CHECK(hipStreamSynchronize(s));
  int best = 3;
  for (int c = 1; c <= 5; ++c) {
    const long long j = p0 - 3 + c;
    if (j < 0 || j >= na) continue;
    const double x = ignore_phase ? fabs(v[c]) : v[c], y = ignore_phase ? fabs(v[best]) : v[best];
    if (x > y || (x == y && c < best)) best = c;
  }
  const long long peak = p0 - 3 + best;
  PAR_REQUIRE(peak != na - 1, PAR_ERR_INDEX, "par_find_delay_f64: correlation peak on the last lag (index %lld is out of bounds "
              "for axis 0 with size %lld in the reference's parabolic())", (long long)na, (long long)na);
  double fm = v[best - 1], f0 = v[best], fp = v[best + 1];
  if (peak == 0) {                                                     // f[-1]: numpy wraps to the last element
    hipLaunchKernelGGL(k_xc_exact, dim3(1), dim3(256), 0, s, a, na, b, nb, (const double*)norms, (const long long*)arg,
                       (int64_t)(na - 1 - p0), vals);
    PAR_HIP_CHECK(hipMemcpyAsync(&fm, vals, sizeof(double), hipMemcpyDeviceToHost, s));
    PAR_HIP_CHECK(hipStreamSynchronize(s));
  }
  const double xv = 0.5 * (fm - fp) / (fm - 2.0 * f0 + fp) + (double)peak;
  *corr = f0 - 0.25 * (fm - fp) * (xv - (double)peak);
  *delay = xv - (double)(na / 2);
  return PAR_OK;
}

int par_istft_f32(int device, const float* spec, int64_t n_frames, int n_fft, int hop, const float* window,
                  float* frames, float* y, int64_t y_len, int64_t skip, void* stream) {
  using namespace par;
  PAR_REQUIRE(spec && window && y, PAR_ERR_ARG, "par_istft_f32: null pointer");
  PAR_REQUIRE(n_frames >= 1 && hop >= 1 && y_len >= 0 && skip >= 0, PAR_ERR_ARG, "par_istft_f32: bad sizes");
  PAR_REQUIRE(n_fft >= 16 && n_fft <= (1 << 21) && (n_fft & (n_fft - 1)) == 0, PAR_ERR_UNSUPPORTED,
              "par_istft_f32: n_fft=%d is not a power of two in [16, 2^21]", n_fft);
  PAR_HIP_CHECK(hipSetDevice(device));
  if (n_fft > 8192) {                              // frames of the four-step size class (the GUI's FFT sizes go to 2^20)
    PAR_REQUIRE(frames, PAR_ERR_ARG, "par_istft_f32: n_fft=%d needs the scratch of par_istft_scratch_floats", n_fft);
    const int64_t H = n_fft / 2, B = istft_big_batch(n_frames, n_fft);
    float2* A = reinterpret_cast<float2*>(frames + ((n_frames * (int64_t)n_fft + 3) & ~3ll));
    float2* V = A + B * H;
    const float scale = (float)(sqrt((double)n_fft) / (double)H);
    for (int64_t f0 = 0; f0 < n_frames; f0 += B) {
      const int64_t nb = n_frames - f0 < B ? n_frames - f0 : B;
      hipLaunchKernelGGL(k_ibig_pack, dim3((unsigned)ceil_div(nb * H, 256)), dim3(256), 0, as_stream(stream),
                         reinterpret_cast<const float2*>(spec), f0, nb, H, A);
      int rc2 = big_fft_c2c(device, A, V, H, nb, as_stream(stream));
      if (rc2 != PAR_OK) return rc2;
      hipLaunchKernelGGL(k_ibig_unpack, dim3((unsigned)ceil_div(nb * H, 256)), dim3(256), 0, as_stream(stream), (const float2*)V,
                         f0, nb, H, window, frames, scale);
    }
    PAR_HIP_CHECK(hipGetLastError());
    if (y_len > 0) {
      hipLaunchKernelGGL(k_istft_ola, dim3((unsigned)ceil_div(y_len, 256)), dim3(256), 0, as_stream(stream), frames, n_frames,
                         n_fft, hop, window, y, y_len, skip);
      PAR_HIP_CHECK(hipGetLastError());
    }
    return PAR_OK;
  }
  Twiddles tw;
  int rc = get_twiddles(device, n_fft, &tw);
  if (rc != PAR_OK) return rc;
  const int H = n_fft / 2;
  // spec * sqrt(n_fft) (:359) and the 1/H of the H-point complex inverse fold into one factor
  const float scale = (float)(sqrt((double)n_fft) / (double)H);
  const float2* sp2 = reinterpret_cast<const float2*>(spec);
  if (istft_is_fused(n_fft, hop)) {
    if (y_len == 0) return PAR_OK;
    const int s = (n_fft + hop - 1) / hop;
#define PAR_ISTFT_FUSED(LH)                                                                                             \
  {                                                                                                                     \
    const int64_t out_len = (int64_t)(FftGeom<LH>::Frames * s - s + 1) * hop;                                             \
    hipLaunchKernelGGL(k_istft_fused<LH>, dim3((unsigned)ceil_div(y_len + skip, out_len)), dim3(FftGeom<LH>::Threads),    \
                       istft_fused_lds<LH>(hop), as_stream(stream), sp2, n_frames, hop, s, window, tw.w, tw.post, y,     \
                       y_len, skip, scale);                                                                              \
  }
    switch (ilog2(H)) {
      case 3: PAR_ISTFT_FUSED(3); break;
      case 4: PAR_ISTFT_FUSED(4); break;
      case 5: PAR_ISTFT_FUSED(5); break;
      case 6: PAR_ISTFT_FUSED(6); break;
      case 7: PAR_ISTFT_FUSED(7); break;
      case 8: PAR_ISTFT_FUSED(8); break;
      case 9: PAR_ISTFT_FUSED(9); break;
      case 10: PAR_ISTFT_FUSED(10); break;
      case 11: PAR_ISTFT_FUSED(11); break;
      case 12: PAR_ISTFT_FUSED(12); break;
      default: PAR_REQUIRE(false, PAR_ERR_UNSUPPORTED, "par_istft_f32: unsupported size");
    }
#undef PAR_ISTFT_FUSED
    PAR_HIP_CHECK(hipGetLastError());
    return PAR_OK;
  }
  PAR_REQUIRE(frames, PAR_ERR_ARG, "par_istft_f32: n_fft=%d hop=%d needs the frames scratch (par_istft_scratch_floats)", n_fft, hop);
#define PAR_ISTFT_LAUNCH(LH)                                                                                            \
  hipLaunchKernelGGL(k_istft_frames<LH>, dim3((unsigned)ceil_div(n_frames, FftGeom<LH>::Frames)),                         \
                     dim3(FftGeom<LH>::Threads), (size_t)FftGeom<LH>::Frames * FftGeom<LH>::FrameLds * sizeof(float2),      \
                     as_stream(stream), sp2, n_frames, window, tw.w, tw.post, frames, scale)
  switch (ilog2(H)) {
    case 3: PAR_ISTFT_LAUNCH(3); break;
    case 4: PAR_ISTFT_LAUNCH(4); break;
    case 5: PAR_ISTFT_LAUNCH(5); break;
    case 6: PAR_ISTFT_LAUNCH(6); break;
    case 7: PAR_ISTFT_LAUNCH(7); break;
    case 8: PAR_ISTFT_LAUNCH(8); break;
    case 9: PAR_ISTFT_LAUNCH(9); break;
    case 10: PAR_ISTFT_LAUNCH(10); break;
    case 11: PAR_ISTFT_LAUNCH(11); break;
    case 12: PAR_ISTFT_LAUNCH(12); break;
    default: PAR_REQUIRE(false, PAR_ERR_UNSUPPORTED, "par_istft_f32: unsupported size");
  }
#undef PAR_ISTFT_LAUNCH
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
