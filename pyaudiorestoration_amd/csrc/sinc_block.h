// Device code of the block kernels (K_sinc; the kernels, the tile lists and the host side are in sinc.hip): tap loops, placement
// from the plan, the float64 slow path, and fused_wave -- one wave's share of a fused launch -- which the streaming kernel
// (sinc2.hip) also calls for the file's end tiles.  See the head of sinc.hip for the arithmetic.
#pragma once
#include "par_common.h"
#include "pos_plan.h"
#include "sinc_taps_gen.h"
#include "sinc_common.h"
#include <atomic>
#include <utility>
#include <limits.h>
#include <math.h>
#include <map>
#include <vector>

// PAR_SINC_EXP (tools/build_variant.py): phase-timing builds, never shipped.  Bit mask: 1 tap loops skipped, 2 placement
// replaced by identity positions, 4 (with 2) fc = 0.995 instead of 1, 8 anchor taken from the output index (no dependent
// scalar loads at the top), 16 staging loads skipped, 32 stores skipped.
#ifndef PAR_SINC_EXP
#define PAR_SINC_EXP 0
#endif
#ifndef PAR_SINC_HOT
#define PAR_SINC_HOT 1          // experiment knob: 0 = every wave takes the general (masked, strided) path
#endif
#ifndef PAR_SINC_WAVES
#define PAR_SINC_WAVES 6      // waves per SIMD the fused kernel is built for (register budget 512 / this)
#endif
#ifndef PAR_FARROW_BARRIER_EARLY
#define PAR_FARROW_BARRIER_EARLY 0   // 1: the workgroup barrier behind the constant fragments sits in front of the span DMA instead of behind it (A/B knob: equal on an all-fast tape, the benchmark mix prefers 0)
#endif
#ifndef PAR_MFMA_EXP
#define PAR_MFMA_EXP 0        // timing experiments on the bank (never shipped): 1 no MFMAs, 2 no fragment reads, 4 no conversion, 8 no bank / gather
#endif
#ifndef PAR_FARROW_WAVES
#define PAR_FARROW_WAVES 4    // waves per workgroup of the kernel that keeps the Farrow constants in LDS (experiment knob: 8, 12, 16)
#endif
#ifndef PAR_SINC_MFMA
#define PAR_SINC_MFMA 1       // unity path of the mono NT = 32 kernel: taps n >= 5 as a Farrow bank on the matrix cores (0: VALU loops)
#endif

namespace par {

#if PAR_SINC_EXP & 64
static __device__ unsigned int* g_sinc_phase;           // [wave][8] cycle counts, one row per wave of the launch (no atomics)
#define PAR_PHASE_MARK(k)                                                                   \
  do {                                                                                      \
    const unsigned long long now_ = __builtin_readcyclecounter();                           \
    if ((threadIdx.x & 63) == 0)                                                            \
      g_sinc_phase[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + (k)] = (unsigned)(now_ - phase_t_); \
    phase_t_ = now_;                                                                        \
  } while (0)
#define PAR_PHASE_BEGIN() unsigned long long phase_t_ = __builtin_readcyclecounter()
#else
#define PAR_PHASE_MARK(k) do { } while (0)
#define PAR_PHASE_BEGIN() do { } while (0)
#endif

#ifndef PAR_SINC_PRIO
#define PAR_SINC_PRIO 0                  // s_setprio level of the fused kernels' waves (0: the default level, as every other kernel)
#endif
constexpr int kSincBlock = 256;
constexpr int kSincR = 4;                         // outputs per thread
constexpr int kSincTile = kSincBlock * kSincR;    // outputs per workgroup
constexpr int kSincCap = 4096;                    // LDS floats for the staged input span (16 KiB; speeds up to ~3.7)

// sin(pi x) in float64 without the library's range reduction (a float64 sin costs ~250 instructions here, and the slow path
// below calls two transcendentals per tap: the file's leading edge alone kept one wave of k_sinc_fused_list busy for 50 us
// behind every launch of the streaming kernel, r05): x - rint(x) is exact, the Taylor polynomial of sin(pi r) to r^21 is good to
// 3e-16 on [-1/2, 1/2].
__device__ __forceinline__ double sinpi_f64(double x) {
  const double n = rint(x);
  const double r = x - n;                                  // exact
  const double z = r * r;
  double p = 5.392664662608125e-10;
  p = fma(p, z, -2.2948428997269856e-08);
  p = fma(p, z, 7.952054001475508e-07);
  p = fma(p, z, -2.1915353447830204e-05);
  p = fma(p, z, 0.00046630280576761234);
  p = fma(p, z, -0.007370430945714348);
  p = fma(p, z, 0.08214588661112819);
  p = fma(p, z, -0.5992645293207919);
  p = fma(p, z, 2.550164039877345);
  p = fma(p, z, -5.167712780049969);
  p = fma(p, z, 3.141592653589793);
  const double v = p * r;
  return ((long long)n & 1) ? -v : v;
}

#ifndef PAR_SLOW_CHUNK
#define PAR_SLOW_CHUNK 4         // (8: 80 registers and scratch, slower)
#endif

#ifndef PAR_SLOW_WAVE
#define PAR_SLOW_WAVE 1
#endif
constexpr int kSlowChunk = PAR_SLOW_CHUNK;         // taps whose samples the float64 slow path fetches together

// Fully general float64 evaluation of ONE output straight from global memory.  Used for the
// leading-edge outputs (ind < NT), for tiles whose input span does not fit LDS, and as the
// in-library cross-check of the fast path.  Follows util/resampling.py:66-90 line by line.
__device__ __noinline__ float sinc_one_f64(double p, double dp, const float* __restrict__ sig, int64_t sig_stride,
                              int64_t len_in, int NT) {
  // Python's int(round(p)) has no range limit: a position beyond +-2^63 selects an EMPTY slice of the signal
  // (sum 0.0).  Caught here before the 64-bit index arithmetic below could wrap (found by tools/fuzz_operator_slot.py).
  if (!(fabs(p) < 9.0e18)) return 0.0f;
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
  // kSlowChunk taps at a time, their samples fetched together: one load per trip of the loop cost a memory round trip per TAP (the
  // file's first and last tiles kept the tile list behind the streaming kernel busy for 55-80 us, r05); the sum runs in tap order
  for (long long k0 = 0; k0 < L; k0 += kSlowChunk) {
    float xs[kSlowChunk];
#pragma unroll
    for (int u = 0; u < kSlowChunk; ++u) xs[u] = k0 + u < L ? sig[(lower + k0 + u) * sig_stride] : 0.0f;
#pragma unroll
    for (int u = 0; u < kSlowChunk; ++u) {
      const long long k = k0 + u;
      if (k >= L) break;
      double x = ((double)(k - NT) - shift) * fc;
      x = x == 0.0 ? 1e-20 : x;                            // np.sinc
      // sin(pi x) / (pi x): |x| <= NT + 1 here (|k - NT| <= NT, |shift| <= 1/2, fc <= 1).  (A library sin() kept as a fallback for
      // huge |x| -- dead code -- was inlined into the loop and, under the kernels' 80-register cap, put the polynomial's constants
      // into scratch: six reloads per tap.)
      double si = sinpi_f64(x) / (M_PI * x) * fc;
      // np.hanning(2NT+1)[k] as f32: cos(pi u) = sin(pi (u + 1/2)), u = (k - NT) / NT in [-1, 1]
      float win = (float)(0.5 + 0.5 * sinpi_f64((double)(k - NT) / (double)NT + 0.5));
      acc += (double)xs[u] * si * (double)win;
    }
  }
  return (float)acc;
}

// The two channels of one output of a stereo file: the same arithmetic per channel (bit for bit), the tap weights worked out
// once.  (The stereo tile list behind the streaming kernel spent 79 us per launch here, twice the mono list's time, r05.)
__device__ __noinline__ float2 sinc_two_f64(double p, double dp, const float* __restrict__ sig, const float* __restrict__ sig1,
                                            int64_t sig_stride, int64_t len_in, int NT) {
  if (!(fabs(p) < 9.0e18)) return make_float2(0.0f, 0.0f);
  const long long ind = llrint(p);
  const long long lower = ind - NT > 0 ? ind - NT : 0;
  const long long upper = ind + NT < (long long)len_in ? ind + NT : (long long)len_in;
  const long long L = upper - lower;
  if (L <= 0) return make_float2(0.0f, 0.0f);
  const double period = dp > 1e-12 ? dp : 1e-12;
  const double inv = 1.0 / period;
  const double fc = inv < 1.0 ? inv : 1.0;
  const double shift = p - (double)ind;
  double acc0 = 0.0, acc1 = 0.0;
  for (long long k0 = 0; k0 < L; k0 += kSlowChunk) {
    float xs[kSlowChunk], ys[kSlowChunk];
#pragma unroll
    for (int u = 0; u < kSlowChunk; ++u) {
      xs[u] = k0 + u < L ? sig[(lower + k0 + u) * sig_stride] : 0.0f;
      ys[u] = k0 + u < L ? sig1[(lower + k0 + u) * sig_stride] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < kSlowChunk; ++u) {
      const long long k = k0 + u;
      if (k >= L) break;
      double x = ((double)(k - NT) - shift) * fc;
      x = x == 0.0 ? 1e-20 : x;
      const double si = sinpi_f64(x) / (M_PI * x) * fc;
      const float win = (float)(0.5 + 0.5 * sinpi_f64((double)(k - NT) / (double)NT + 0.5));
      acc0 += (double)xs[u] * si * (double)win;
      acc1 += (double)ys[u] * si * (double)win;
    }
  }
  return make_float2((float)acc0, (float)acc1);
}

// The float64 slow path by a whole wave (the fused kernels' epilogue).  A lane on its own spends ~0.65 us per tap of an output
// -- a float64 division, two polynomials and a memory round trip, nothing beside them to hide their latency -- and the few
// slow outputs of a wave (a rounding tie; the file's first and last 32 outputs) left the other lanes idle meanwhile: the
// first and last tiles kept the tile list behind the streaming kernel busy for 45-80 us per launch (r05).  Here the n slow
// outputs of the row are compacted and each gets f = 64 / n (a power of two) lanes, which share its taps k = sub, sub + f, ...;
// partial sums meet in a butterfly over the f lanes.  Same per-tap arithmetic as sinc_one_f64 (util/resampling.py:66-90); the
// order of the float64 sum differs (1e-16 of the result, which is then rounded to float32).
// want: this lane has a slow output (p, dp); returns its two channels' values (.y unused for NCH = 1) in the lanes that want.
template <int NCH>
__device__ __noinline__ float2 sinc_slow_wave(bool want, double p, double dp, const float* __restrict__ sig,
                                              const float* __restrict__ sig1, int64_t sig_stride, int64_t len_in, int NT, int l) {
  const unsigned long long mask = __ballot(want);
  const int n = __popcll(mask);
  int sh = 0;                                     // f = 1 << sh lanes per output
  while ((n << (sh + 1)) <= kWave) ++sh;
  const int f = 1 << sh, g = l >> sh, sub = l & (f - 1);
  unsigned long long rest = mask;                 // source lane of group g: the g-th lane that wants
  for (int i = 0; i < g && rest; ++i) rest &= rest - 1;
  const bool active = g < n;
  const int src = rest ? __builtin_ctzll(rest) : 0;
  const double pg = __shfl(p, src, kWave), dpg = __shfl(dp, src, kWave);
  double acc0 = 0.0, acc1 = 0.0;
  if (active && fabs(pg) < 9.0e18) {              // (beyond: Python's empty slice, see sinc_one_f64)
    const long long ind = llrint(pg);
    const long long lower = ind - NT > 0 ? ind - NT : 0;
    const long long upper = ind + NT < (long long)len_in ? ind + NT : (long long)len_in;
    const long long L = upper - lower;
    const double period = dpg > 1e-12 ? dpg : 1e-12;
    const double inv = 1.0 / period;
    const double fc = inv < 1.0 ? inv : 1.0;
    const double shift = pg - (double)ind;
    for (long long k = sub; k < L; k += f) {
      double x = ((double)(k - NT) - shift) * fc;
      x = x == 0.0 ? 1e-20 : x;
      const double si = sinpi_f64(x) / (M_PI * x) * fc;
      const float win = (float)(0.5 + 0.5 * sinpi_f64((double)(k - NT) / (double)NT + 0.5));
      acc0 += (double)sig[(lower + k) * sig_stride] * si * (double)win;
      if (NCH == 2) acc1 += (double)sig1[(lower + k) * sig_stride] * si * (double)win;
    }
  }
  for (int o = f >> 1; o > 0; o >>= 1) {
    acc0 += __shfl_xor(acc0, o, kWave);
    if (NCH == 2) acc1 += __shfl_xor(acc1, o, kWave);
  }
  // back to the lanes that asked: the lane of rank q reads group q's first lane
  const int rank = __popcll(mask & ((1ull << l) - 1ull));
  const int from = want ? (rank << sh) : l;
  const double r0 = __shfl(acc0, from, kWave), r1 = NCH == 2 ? __shfl(acc1, from, kWave) : 0.0;
  return make_float2((float)r0, (float)r1);
}

// ---- tap loops ------------------------------------------------------------------------------------
// Taps +n and -n share one reciprocal:  with q = s^2 and R_n = (win_n/pi)/(n^2 - q) = rcp(q*b_n + a2_n)
//   sig[+n]*w(+n) + sig[-n]*w(-n) = R_n * ( n*(G + H) + s*(G - H) ),   G = sig[+n]*U_n,  H = sig[-n]*V_n
// where U_n = sin(theta*(n-s)), V_n = sin(theta*(n+s)) (theta = pi*fc) are the sinc numerators.
// The n loop runs in chunks of kChunk taps so that LDS offsets inside a chunk are instruction immediates,
// the chunk's table entries arrive in one scalar load, and the (-1)^n sign is a free operand modifier.
constexpr int kChunk = 4;
// LDS pointers carry their address space in the type so that, kept live across the chunk loop, they stay
// ds_read base registers with immediate offsets (generic pointers degrade to flat loads, indices to a
// shift+add per access).
typedef __attribute__((address_space(3))) const float lds_cfloat;
// A window base the compiler cannot look into.  Left visible, `workgroup LDS base + constant area + lane part` is re-associated
// so that the constant area's offset (10 KB in the mono NT = 32 kernel) rides on every tap's immediate, which then no longer
// fits the 8-bit offset fields of ds_read2_b32: one v_add_u32 with a literal per read -- a VALU instruction per tap pair
// (r05, found in the listing: 526 of them in k_sinc_fused<1, 32, 4>).  Opaque, every tap is base + immediate.
#ifndef PAR_SINC_OPAQUE
#define PAR_SINC_OPAQUE 1
#endif
// ONLY where a constant area sits in front of the spans (the mono NT = 32 kernel: OPQ = fused_is_farrow): everywhere else the
// immediates fit anyway and the opaque base costs the scheduler its view of the addresses -- the interleaved stereo kernel
// ran the fc = 1 tape in 0.92 instead of 0.72 ms with it, NT = 50 stereo 1.34 instead of 0.94 (r05, tools/exp/stereo_only.py).
template <bool OPQ>
__device__ __forceinline__ lds_cfloat* opaque_lds(lds_cfloat* p) {
  if (!PAR_SINC_OPAQUE || !OPQ) return p;
  unsigned a = (unsigned)(uintptr_t)p;
  asm("" : "+v"(a));
  return (lds_cfloat*)(uintptr_t)a;
}

// How R_n(q) = (win_n/pi)/(n^2 - q), q = shift^2 <= 1/4, is evaluated for the taps of one chunk.  Only the four
// innermost pairs pay for a v_rcp_f32 (quarter rate).  Further out q/n^2 <= 0.01 and R_n is a short polynomial in q
// against a wave-uniform table that arrives by scalar loads and stays in SGPRs: a 2-term Taylor series (n = 5..),
// then the linear minimax fit over [0, 1/4], then a constant (the mid-range value: e and d then accumulate straight
// against SGPR constants, 4 VALU per tap pair).  Where each form starts is decided per NT on the host from worst-case
// error budgets (get_sinc_table: every tap pair's approximation error times its largest possible contribution,
// summed over the pairs that use the form, stays below 3e-7 for the linear and 1.5e-6 for the constant form, against
// the 1e-5 the reference is matched to); the table rows change meaning accordingly.
enum { kRcp = 0, kPoly2 = 1, kPoly1 = 2, kPoly0 = 3 };
constexpr int kPoly2From = 5;     // rows n >= 5: (A_n, B_n, n, C_n) with A = win/(pi n^2), B = A/n^2, C = B/n^2
struct TapModes {
  int p1_from;                    // rows n >= p1_from: (A_n, B_n, n, -) linear minimax;  p1_from = 1 (mod kChunk), >= 5
  int p0_from;                    // rows n >= p0_from: (A_n, n A_n, n, -) constant;       p0_from = 1 (mod kChunk), >= p1_from
};
template <int MODE>
__device__ __forceinline__ float tap_R(float q, const float4& t) {
  if (MODE == kRcp) return fast_rcp(fmaf(q, t.y, t.x));            // row = (pi n^2/win, -pi/win, n, -)
  if (MODE == kPoly2) return fmaf(fmaf(t.w, q, t.y), q, t.x);
  return fmaf(t.y, q, t.x);
}

// fc == 1 for every lane of the wave: U_n = -(-1)^n sin(pi s), V_n = +(-1)^n sin(pi s) -> factored out.
// Accumulates e = sum (-1)^n (sig[+n]+sig[-n]) R_n  and  d = sum (-1)^n n (sig[+n]-sig[-n]) R_n.
// LAST: the chunk that reaches n = NT.  The reference's window is offsets -NT .. NT-1: tap -NT is in it (with the
// Hann endpoint weight 0, so a NaN/Inf sample there still poisons the sum as 0*NaN), tap +NT and the padding
// taps beyond are not -- their samples are replaced by 0 so that non-finite input spreads exactly as far as it
// does in the reference.
template <int MODE, bool LAST, int R>
__device__ __forceinline__ void unity_chunk(lds_cfloat* (&tp)[R], lds_cfloat* (&tm)[R], const float (&q)[R],
                                            float (&e)[R], float (&d)[R], const float4* __restrict__ tab, int n0,
                                            int NT) {
  float4 ab[kChunk];                         // wave-uniform: one s_load_dwordx16, operands stay in SGPRs
#pragma unroll
  for (int k = 0; k < kChunk; ++k) ab[k] = tab[n0 + k];
#pragma unroll
  for (int k = 0; k < kChunk; ++k) {
    const float fn = ab[k].z;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float sp = tp[r][k], sm = tm[r][kChunk - 1 - k];
      if (LAST) {
        if (n0 + k >= NT) sp = 0.0f;
        if (n0 + k > NT) sm = 0.0f;
      }
      const float D = sp - sm, E = sp + sm;
      if (MODE == kPoly0) {                    // R_n constant: straight against the SGPR pair (A_n, n A_n)
        if (k & 1) {                           // n0 is odd, so odd k is an even n: +
          e[r] = fmaf(E, ab[k].x, e[r]);
          d[r] = fmaf(D, ab[k].y, d[r]);
        } else {                               // odd n: -
          e[r] = fmaf(-E, ab[k].x, e[r]);
          d[r] = fmaf(-D, ab[k].y, d[r]);
        }
      } else {
        const float Rn = tap_R<MODE>(q[r], ab[k]);
        const float DR = D * Rn;
        if (k & 1) {
          e[r] = fmaf(E, Rn, e[r]);
          d[r] = fmaf(DR, fn, d[r]);
        } else {
          e[r] = fmaf(-E, Rn, e[r]);
          d[r] = fmaf(-DR, fn, d[r]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    tp[r] += kChunk;
    tm[r] -= kChunk;
  }
}

template <int R>
__device__ __forceinline__ void taps_unity(const float* __restrict__ tile, const int (&c)[R], const float (&s)[R],
                                           int NT, const float4* __restrict__ tab, const TapModes tmd, float (&res)[R]) {
  float q[R], e[R], d[R];
  lds_cfloat* tp[R];
  lds_cfloat* tm[R];
  lds_cfloat* tl = (lds_cfloat*)tile;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    q[r] = s[r] * s[r];
    e[r] = d[r] = 0.0f;
    tp[r] = tl + c[r] + 1;                   // -> t[+n0]
    tm[r] = tl + c[r] - kChunk;              // -> t[-(n0 + kChunk - 1)]
  }
  // chunks n0 = 1, 5, .. while n0 <= NT; the one containing n = NT is the LAST instance (padded table rows
  // n >= NT have R_n == 0)
  int n0 = 1;
  if (n0 + kChunk <= NT) {
    unity_chunk<kRcp, false, R>(tp, tm, q, e, d, tab, n0, NT);
    n0 += kChunk;
#pragma unroll 1
    for (; n0 + kChunk <= NT && n0 < tmd.p1_from; n0 += kChunk) unity_chunk<kPoly2, false, R>(tp, tm, q, e, d, tab, n0, NT);
#pragma unroll 1
    for (; n0 + kChunk <= NT && n0 < tmd.p0_from; n0 += kChunk) unity_chunk<kPoly1, false, R>(tp, tm, q, e, d, tab, n0, NT);
#pragma unroll 1
    for (; n0 + kChunk <= NT; n0 += kChunk) unity_chunk<kPoly0, false, R>(tp, tm, q, e, d, tab, n0, NT);
  }
  if (n0 == 1) unity_chunk<kRcp, true, R>(tp, tm, q, e, d, tab, n0, NT);
  else if (n0 < tmd.p1_from) unity_chunk<kPoly2, true, R>(tp, tm, q, e, d, tab, n0, NT);
  else if (n0 < tmd.p0_from) unity_chunk<kPoly1, true, R>(tp, tm, q, e, d, tab, n0, NT);
  else unity_chunk<kPoly0, true, R>(tp, tm, q, e, d, tab, n0, NT);
  const float b0 = tab[0].y;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float centre = tile[c[r]] * fast_rcp(s[r] * b0);
    res[r] = -sinpi_half(s[r]) * (centre + fmaf(s[r], e[r], d[r]));
  }
}

// general fc in (0, 1]: numerators by 3-term recurrences seeded at the centre and run outwards.
template <int R>
struct GenState {
  float q[R], accP[R], accM[R], U[R], Up[R], V[R], Vp[R], c2[R];
};
template <int MODE, bool LAST, int R>
__device__ __forceinline__ void general_chunk(lds_cfloat* (&tp)[R], lds_cfloat* (&tm)[R], GenState<R>& g,
                                              const float4* __restrict__ tab, int n0, int NT) {
  float4 ab[kChunk];
#pragma unroll
  for (int k = 0; k < kChunk; ++k) ab[k] = tab[n0 + k];
#pragma unroll
  for (int k = 0; k < kChunk; ++k) {
    const float fn = ab[k].z;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float sp = tp[r][k], sm = tm[r][kChunk - 1 - k];
      if (LAST) {
        if (n0 + k >= NT) sp = 0.0f;
        if (n0 + k > NT) sm = 0.0f;
      }
      const float G = sp * g.U[r], H = sm * g.V[r];
      if (MODE == kPoly0) {
        g.accM[r] = fmaf(G - H, ab[k].x, g.accM[r]);
        g.accP[r] = fmaf(G + H, ab[k].y, g.accP[r]);
      } else {
        const float Rn = tap_R<MODE>(g.q[r], ab[k]);
        g.accM[r] = fmaf(G - H, Rn, g.accM[r]);
        g.accP[r] = fmaf((G + H) * Rn, fn, g.accP[r]);
      }
      const float un = fmaf(g.c2[r], g.U[r], -g.Up[r]);
      g.Up[r] = g.U[r];
      g.U[r] = un;
      const float vn = fmaf(g.c2[r], g.V[r], -g.Vp[r]);
      g.Vp[r] = g.V[r];
      g.V[r] = vn;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    tp[r] += kChunk;
    tm[r] -= kChunk;
  }
}

template <int R>
__device__ __forceinline__ void taps_general(const float* __restrict__ tile, const int (&c)[R], const float (&s)[R],
                                             const float (&fc)[R], const float (&dd)[R], int NT,
                                             const float4* __restrict__ tab, const TapModes tmd, float (&res)[R]) {
  GenState<R> g;
  float centre[R];
  lds_cfloat* tp[R];
  lds_cfloat* tm[R];
  lds_cfloat* tl = (lds_cfloat*)tile;
  const float b0 = tab[0].y;
  bool gentle = true;                                 // see taps_general_ct
#pragma unroll
  for (int r = 0; r < R; ++r) gentle = gentle && dd[r] <= 0.03125f;
  gentle = __all(gentle);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float h = fc[r] * s[r];                    // phi / pi, |h| <= 0.5
    const float sphi = sinpi_half(h);
    if (gentle) {
      const float z = dd[r] * dd[r];
      g.U[r] = sinpi_half(dd[r] + h);                 // U_1 = sin(theta - phi)
      g.V[r] = sinpi_half(dd[r] - h);                 // V_1 = sin(theta + phi)
      g.c2[r] = fmaf(z, fmaf(z, -8.11742426f, 9.86960440f), -2.0f);
    } else {
      const float cphi = cospi_half(h);
      float sth, cth;                                 // sin/cos(theta), theta = pi*fc = pi - pi*dd
      if (dd[r] <= 0.5f) {
        sth = sinpi_half(dd[r]);
        cth = -cospi_half(dd[r]);
      } else {
        sth = sinpi_half(fc[r]);
        cth = cospi_half(fc[r]);
      }
      g.U[r] = fmaf(sth, cphi, -cth * sphi);
      g.V[r] = fmaf(sth, cphi, cth * sphi);
      g.c2[r] = 2.0f * cth;
    }
    g.Up[r] = -sphi;                                  // U_0 = sin(-phi)
    g.Vp[r] = sphi;                                   // V_0 = sin(+phi)
    g.q[r] = s[r] * s[r];
    centre[r] = tile[c[r]] * (g.Up[r] * fast_rcp(s[r] * b0));
    g.accP[r] = g.accM[r] = 0.0f;
    tp[r] = tl + c[r] + 1;
    tm[r] = tl + c[r] - kChunk;
  }
  int n0 = 1;
  if (n0 + kChunk <= NT) {
    general_chunk<kRcp, false, R>(tp, tm, g, tab, n0, NT);
    n0 += kChunk;
#pragma unroll 1
    for (; n0 + kChunk <= NT && n0 < tmd.p1_from; n0 += kChunk) general_chunk<kPoly2, false, R>(tp, tm, g, tab, n0, NT);
#pragma unroll 1
    for (; n0 + kChunk <= NT && n0 < tmd.p0_from; n0 += kChunk) general_chunk<kPoly1, false, R>(tp, tm, g, tab, n0, NT);
#pragma unroll 1
    for (; n0 + kChunk <= NT; n0 += kChunk) general_chunk<kPoly0, false, R>(tp, tm, g, tab, n0, NT);
  }
  if (n0 == 1) general_chunk<kRcp, true, R>(tp, tm, g, tab, n0, NT);
  else if (n0 < tmd.p1_from) general_chunk<kPoly2, true, R>(tp, tm, g, tab, n0, NT);
  else if (n0 < tmd.p0_from) general_chunk<kPoly1, true, R>(tp, tm, g, tab, n0, NT);
  else general_chunk<kPoly0, true, R>(tp, tm, g, tab, n0, NT);
#pragma unroll
  for (int r = 0; r < R; ++r) res[r] = centre[r] + fmaf(s[r], g.accM[r], g.accP[r]);
}

// ---- NT-specialised tap loops ---------------------------------------------------------------------------
// Measured on gfx950 (tools/ubench2.hip): a VALU instruction with an SGPR source operand (or the same VGPR twice, or
// a compare / convert / select / DPP / any float64 operation) issues in ~4 cycles per wave, one whose sources are
// distinct VGPRs or an instruction literal in ~2.  The generic loops above keep their table in SGPRs (a third of their
// instructions are therefore slow).  For the tap counts that matter (TapTab<NT>: NT = 32, the benchmark's 64 taps, and
// NT = 50, the GUI default) the loops are fully unrolled instead: every coefficient is an instruction literal, LDS
// offsets are immediates off ONE base pointer per output, and the polynomial forms are accumulated coefficient by
// coefficient (e = e0 + q e1 + q^2 e2 is assembled once at the end) so that no per-tap weight is ever formed:
//   constant form 4, linear 6, 2-term series 8 VALU per tap pair and output (unity path), all fast.
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// LS: LDS stride of a sample.  1: one channel per LDS array; 2: the wave's span holds interleaved stereo samples and the R = 2
// slots of a call are the two channels of ONE output (c[1] == c[0] + 1): their taps are adjacent words of one base.
template <int NT, int R, int LS = 1, bool OPQ = false>
__device__ __forceinline__ void taps_unity_ct(const float* __restrict__ tile, const int (&c)[R], const float (&s)[R],
                                              const int nt_rt, float (&res)[R]) {
  static_assert(LS == 1 || (LS == 2 && R == 2), "interleaved spans: the two channel slots of one output");
  using T = TapTab<NT>;
  float q[R], e0[R], e1[R], e2[R], d0[R], d1[R], d2[R];
  lds_cfloat* base[R];
  lds_cfloat* tl = (lds_cfloat*)tile;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    q[r] = s[r] * s[r];
    e0[r] = e1[r] = e2[r] = d0[r] = d1[r] = d2[r] = 0.0f;
    base[r] = (LS == 2 && r == 1) ? base[0] + 1 : opaque_lds<OPQ>(tl + c[r] - NT * LS);    // tap +n at [(NT + n) LS], tap -n at [(NT - n) LS]: immediates
  }
  // One basic block per chunk of kChunk taps: the (always true, but opaque to the compiler) test on the run-time NT keeps
  // the instruction selector from interleaving the whole unrolled sequence -- left as ONE block it runs loads and
  // recurrences dozens of taps ahead and spills hundreds of registers (measured twice with loop unrolling, once here).
  static_for<(NT + kChunk - 1) / kChunk>([&](auto cidx) {
   constexpr int n0 = decltype(cidx)::value * kChunk + 1;
   if (nt_rt >= n0) static_for<(n0 + kChunk - 1 <= NT ? kChunk : NT - n0 + 1)>([&](auto idx) {
    constexpr int n = n0 + decltype(idx)::value;          // 1 .. NT
    constexpr int mode = T::mode[n];
    constexpr float fn = (float)n;
    if constexpr (n == T::p1_from && T::p1_from > kPoly2From) {   // the 2-term rows are behind us: fold their q^2 sums
#pragma unroll
      for (int r = 0; r < R; ++r) {
        e1[r] = fmaf(q[r], e2[r], e1[r]);
        d1[r] = fmaf(q[r], d2[r], d1[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if constexpr (n == NT) {
        // the reference's window is offsets -NT .. NT-1: tap -NT is in it with the Hann endpoint weight 0 (a NaN/Inf
        // sample there still poisons the sum as 0 * NaN), tap +NT is not
        e0[r] = fmaf(base[r][0], 0.0f, e0[r]);
      } else {
        const float sp = base[r][(NT + n) * LS], sm = base[r][(NT - n) * LS];
        const float D = sp - sm, E = sp + sm;
        if constexpr (mode == kRcp) {
#pragma clang fp contract(off)
          const float x = q[r] * T::B[n] + T::A[n];                 // two literals: a multiply and an add, both fast
          const float Rn = fast_rcp(x);
          const float DR = D * Rn;
          if constexpr (n & 1) {
            e0[r] = fmaf(-E, Rn, e0[r]);
            d0[r] = fmaf(DR, -fn, d0[r]);
          } else {
            e0[r] = fmaf(E, Rn, e0[r]);
            d0[r] = fmaf(DR, fn, d0[r]);
          }
        } else if constexpr (mode == kPoly2) {
          e0[r] = fmaf(E, T::A[n], e0[r]);
          e1[r] = fmaf(E, T::B[n], e1[r]);
          e2[r] = fmaf(E, T::C[n], e2[r]);
          d0[r] = fmaf(D, fn * T::A[n], d0[r]);
          d1[r] = fmaf(D, fn * T::B[n], d1[r]);
          d2[r] = fmaf(D, fn * T::C[n], d2[r]);
        } else if constexpr (mode == kPoly1) {
          e0[r] = fmaf(E, T::A[n], e0[r]);
          e1[r] = fmaf(E, T::B[n], e1[r]);
          d0[r] = fmaf(D, fn * T::A[n], d0[r]);
          d1[r] = fmaf(D, fn * T::B[n], d1[r]);
        } else {
          e0[r] = fmaf(E, T::A[n], e0[r]);
          d0[r] = fmaf(D, fn * T::A[n], d0[r]);
        }
      }
    }
   });
  });
  constexpr float b0 = T::B[0];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float e = fmaf(q[r], e1[r], e0[r]), d = fmaf(q[r], d1[r], d0[r]);
    const float centre = tile[c[r]] * fast_rcp(s[r] * b0);
    res[r] = -sinpi_half(s[r]) * (centre + fmaf(s[r], e, d));
  }
}

template <int NT, int R, int LS = 1, bool OPQ = false>
__device__ __forceinline__ void taps_general_ct(const float* __restrict__ tile, const int (&c)[R], const float (&s)[R],
                                                const float (&fc)[R], const float (&dd)[R], const int nt_rt,
                                                float (&res)[R]) {
  static_assert(LS == 1 || (LS == 2 && R == 2), "interleaved spans: the two channel slots of one output");
  using T = TapTab<NT>;
  float q[R], U[R], Up[R], V[R], Vp[R], c2[R], M0[R], M1[R], M2[R], P0[R], P1[R], P2[R], centre[R];
  lds_cfloat* base[R];
  lds_cfloat* tl = (lds_cfloat*)tile;
  constexpr float b0 = T::B[0];
  // Gentle speed-ups (1 - fc <= 1/32 in every lane: any real wow / flutter curve) seed the recurrences directly:
  //   U_1 = sin(theta - phi) = sin(pi (1 - (dd + h))) = sinpi(dd + h),  V_1 = sinpi(dd - h)   (|dd +- h| <= 0.516: the
  //   Taylor polynomial of sinpi_half is still good to 1e-7 there), 2 cos(theta) = -2 cos(pi dd) by three terms in dd^2
  // instead of four full-range polynomials and the angle addition (18 VALU per output less, and more accurate).
  bool gentle = true;
#pragma unroll
  for (int r = 0; r < R; ++r) gentle = gentle && dd[r] <= 0.03125f;
  gentle = __all(gentle);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float h = fc[r] * s[r];                    // phi / pi, |h| <= 0.5
    const float sphi = sinpi_half(h);
    if (gentle) {
      const float z = dd[r] * dd[r];
      U[r] = sinpi_half(dd[r] + h);                   // U_1 = sin(theta - phi)
      V[r] = sinpi_half(dd[r] - h);                   // V_1 = sin(theta + phi)
      c2[r] = fmaf(z, fmaf(z, -8.11742426f, 9.86960440f), -2.0f);      // -2 cos(pi dd): -2 + pi^2 dd^2 - pi^4 dd^4 / 12
    } else {
      const float cphi = cospi_half(h);
      float sth, cth;                                 // sin/cos(theta), theta = pi*fc = pi - pi*dd
      if (dd[r] <= 0.5f) {
        sth = sinpi_half(dd[r]);
        cth = -cospi_half(dd[r]);
      } else {
        sth = sinpi_half(fc[r]);
        cth = cospi_half(fc[r]);
      }
      U[r] = fmaf(sth, cphi, -cth * sphi);
      V[r] = fmaf(sth, cphi, cth * sphi);
      c2[r] = 2.0f * cth;
    }
    Up[r] = -sphi;                                    // U_0 = sin(-phi)
    Vp[r] = sphi;                                     // V_0 = sin(+phi)
    q[r] = s[r] * s[r];
    centre[r] = tile[c[r]] * (Up[r] * fast_rcp(s[r] * b0));
    M0[r] = M1[r] = M2[r] = P0[r] = P1[r] = P2[r] = 0.0f;
    base[r] = (LS == 2 && r == 1) ? base[0] + 1 : opaque_lds<OPQ>(tl + c[r] - NT * LS);
  }
  static_for<(NT + kChunk - 1) / kChunk>([&](auto cidx) {
   constexpr int n0 = decltype(cidx)::value * kChunk + 1;
   if (nt_rt >= n0) static_for<(n0 + kChunk - 1 <= NT ? kChunk : NT - n0 + 1)>([&](auto idx) {
    constexpr int n = n0 + decltype(idx)::value;
    constexpr int mode = T::mode[n];
    constexpr float fn = (float)n;
    constexpr float sg = (n & 1) ? -1.0f : 1.0f;      // the numerators carry (-1)^n themselves: undo the table's sign
    if constexpr (n == T::p1_from && T::p1_from > kPoly2From) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        M1[r] = fmaf(q[r], M2[r], M1[r]);
        P1[r] = fmaf(q[r], P2[r], P1[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if constexpr (n == NT) {
        M0[r] = fmaf(base[r][0] * V[r], 0.0f, M0[r]);             // tap -NT: weight 0, poison kept (see taps_unity_ct)
      } else {
        const float sp = base[r][(NT + n) * LS], sm = base[r][(NT - n) * LS];
        const float G = sp * U[r], H = sm * V[r];
        const float t1 = G - H, t2 = G + H;
        if constexpr (mode == kRcp) {
#pragma clang fp contract(off)
          const float x = q[r] * T::B[n] + T::A[n];
          const float Rn = fast_rcp(x);
          M0[r] = fmaf(t1, Rn, M0[r]);
          P0[r] = fmaf(t2 * Rn, fn, P0[r]);
        } else if constexpr (mode == kPoly2) {
          M0[r] = fmaf(t1, sg * T::A[n], M0[r]);
          M1[r] = fmaf(t1, sg * T::B[n], M1[r]);
          M2[r] = fmaf(t1, sg * T::C[n], M2[r]);
          P0[r] = fmaf(t2, sg * fn * T::A[n], P0[r]);
          P1[r] = fmaf(t2, sg * fn * T::B[n], P1[r]);
          P2[r] = fmaf(t2, sg * fn * T::C[n], P2[r]);
        } else if constexpr (mode == kPoly1) {
          M0[r] = fmaf(t1, sg * T::A[n], M0[r]);
          M1[r] = fmaf(t1, sg * T::B[n], M1[r]);
          P0[r] = fmaf(t2, sg * fn * T::A[n], P0[r]);
          P1[r] = fmaf(t2, sg * fn * T::B[n], P1[r]);
        } else {
          M0[r] = fmaf(t1, sg * T::A[n], M0[r]);
          P0[r] = fmaf(t2, sg * fn * T::A[n], P0[r]);
        }
        const float vn = fmaf(c2[r], V[r], -Vp[r]);
        Vp[r] = V[r];
        V[r] = vn;
        if constexpr (n + 1 < NT) {                   // U_NT is never used (tap +NT is outside the window)
          const float un = fmaf(c2[r], U[r], -Up[r]);
          Up[r] = U[r];
          U[r] = un;
        }
      }
    }
   });
  });
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float accM = fmaf(q[r], M1[r], M0[r]), accP = fmaf(q[r], P1[r], P0[r]);
    res[r] = centre[r] + fmaf(s[r], accM, accP);
  }
}

// ---- unity path, far taps on the matrix cores (r03; NT = 32) --------------------------------------------------------------
// On the fc = 1 path the weights of the taps n >= 5 are polynomials in q = shift^2 with FIXED coefficients (above): the
// six sums e0 e1 e2 d0 d1 d2 are fixed 63-tap FIR filters evaluated at the window centre -- a Farrow bank on the input
// grid -- and a dense contraction after all:
//     D[(filter f, position i)][block b] += A[(f, i)][k] B[k][b]          v_mfma_f32_16x16x32_f16
//     B[k][b] = x16[8 b + k]      the signal itself: a PAIR of rows (128 outputs, centres p0 .. p0 + 127) re-based so that
//                                 block b = 8 centres starts on a multiple of 8 halves: 16-byte aligned ds_read_b128 (any
//                                 other alignment costs 256 cycles instead of 30, tools/exp/lds_pattern.hip)
//     A[(f, i)][k]                coefficient of tap n = k - 31 - i of filter f: ten constant fragments (sinc_taps_gen.h,
//                                 10 KB at the front of the workgroup's LDS)
// float16 carries 11 bits, so the signal and the dominant filter pair are split hi + lo 2^-12 (three products where they
// matter, the lo x lo one is 2^-24): 15 MFMAs per 128 outputs, accumulating in float32.  A lane ends up with four
// consecutive positions of one filter: the bank [6][128] goes to LDS (over the float16 image it was made from), every
// output gathers the six values of its window centre and finishes with 5 FMAs.  Same accuracy as the literal-FMA loops
// (tools/ubench_farrow_mfma.hip, profiles/r03_farrow_mfma.txt: <= 1.7e-6 of the peak on the three stress signals, 1.43x on
// the unity tap path); the reciprocal taps n = 1..4, the centre tap and tap -NT (weight 0, NaN poison) stay on the VALU.
// Waves whose samples do not suit float16 (non-finite or >= 6e4: overflow, and the exact NaN footprint of the reference's
// window; or all below 2^-10: the lo part would go subnormal) are told so by the return value and take the VALU loops.
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
// Geometry of the bank by NT (r06: NT = 50, the reference's default quality, mono): K slices of 32 taps, where the fragments of
// (e1 d1), the lo parts of (e0 d0) and (e2 d2) start in the table and which slices they cover (tools/gen_sinc_taps.py).
template <int NT>
struct FarGeom;
template <>
struct FarGeom<32> {
  static constexpr int kFrags = kFarrowFrags, kSlices = 3, kImage = 224;       // image: 128 centres + 63 taps, padded to the slices
  static constexpr int kE1Frag = 3, kE1First = 0, kE1Count = 2, kLoFrag = 5, kE2Frag = 8, kE2First = 0, kE2Count = 2;
  static constexpr float kScaleInv = kFarrowScaleInv;
  __device__ static const unsigned int* table() { return kFarrowFrags32; }
};
template <>
struct FarGeom<50> {
  static constexpr int kFrags = kFarrowFrags50, kSlices = 4, kImage = 256;     // 128 centres + 99 taps
  static constexpr int kE1Frag = 4, kE1First = 0, kE1Count = 3, kLoFrag = 7, kE2Frag = 11, kE2First = 1, kE2Count = 2;
  static constexpr float kScaleInv = kFarrowScaleInv50;
  __device__ static const unsigned int* table() { return kFarrowFrags50_32; }
};
__host__ __device__ constexpr int far_const_bytes(int NTC) { return (NTC == 50 ? kFarrowFrags50 : kFarrowFrags) * 1024; }
constexpr int kFarSpanMax = 384;                       // floats of a wave's LDS piece the span may use on this path (the rest: image / bank)
constexpr int kFarSpanMax2 = 448;                      // ... of the stereo kernel's piece (1280 floats): 128 outputs x 2 interleaved channels + halo
constexpr int kFarConstBytes = far_const_bytes(32);
static_assert(kFarSpanMax * 4 + 4 * 128 * 4 + 2 * 128 * 2 <= 1024 * 4, "span + bank fit the wave's LDS piece");
static_assert(2 * FarGeom<32>::kImage * 2 <= 4 * 128 * 4 && 2 * FarGeom<50>::kImage * 2 <= 4 * 128 * 4, "the bank overwrites the image");
static_assert(8 * 15 + 32 * FarGeom<50>::kSlices <= FarGeom<50>::kImage + 8 && 8 * 15 + 32 * FarGeom<32>::kSlices <= FarGeom<32>::kImage + 8,
              "the last lane's last fragment ends inside the image (+ its 8 trailing halves of slack)");
static_assert(kFarSpanMax2 * 4 + 4 * 128 * 4 + 2 * 128 * 2 <= 1280 * 4, "span + bank fit the stereo wave's LDS piece");

__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// LS = 1: mono, slots (0, 1) and (2, 3) are the two row pairs (256 consecutive outputs).  LS = 2: an interleaved stereo span
// (sample i of channel ch at word 2 i + ch), slots (0, 2) are the row pair of channel 0, (1, 3) of channel 1; c[] are WORD
// indices, the image of a pair takes every LS-th word from its first centre's.
template <int NT, int kOut, int LS>
__device__ __forceinline__ void unity_far_mfma(const float* __restrict__ tile, const int nlim, float* __restrict__ scratch,
                                               const unsigned consts_addr, const int l, const int (&c)[kOut],
                                               const float (&s)[kOut], const float (&q)[kOut], float (&far)[kOut]) {
  static_assert(kOut == 4 && (LS == 1 || LS == 2), "two row pairs per wave");
  using G = FarGeom<NT>;
  constexpr int kFarImage = G::kImage;
  _Float16* rb = reinterpret_cast<_Float16*>(scratch);             // [2][kFarImage]: hi, lo x 4096
  float* bank = scratch;                                            // [4][128] float32: e0 d0 e1 d1
  _Float16* bankh = reinterpret_cast<_Float16*>(scratch + 4 * 128); // [2][128] float16: e2 d2 (1e-4 of the sum: 11 bits are plenty)
  const unsigned rb_addr = lds_addr_of(scratch);
  const unsigned ca = consts_addr + (unsigned)l * 16u;
  const int bb = l & 15, g = l >> 4;
#pragma unroll
  for (int rp = 0; rp < 2; ++rp) {
    const int r0 = LS == 1 ? 2 * rp : rp, rstep = LS == 1 ? 1 : 2;   // the pair's slots: r0, r0 + rstep
    const int p0 = __builtin_amdgcn_readlane(c[r0], 0);             // first centre of the pair (LDS index); centres p0 .. p0 + 127 (samples)
    // float16 image of the samples p0 - (NT - 1) .. (zero behind the staged span: only zero coefficients meet those)
    for (int i2 = l; i2 < ((PAR_MFMA_EXP & 4) ? 0 : kFarImage / 2); i2 += kWave) {
      const int ti = p0 + (2 * i2 - (NT - 1)) * LS;
      const float x0 = ti < nlim ? tile[ti] : 0.0f, x1 = ti + LS < nlim ? tile[ti + LS] : 0.0f;
      const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
      const half2v hv = {h0, h1};
      const half2v lv = {(_Float16)((x0 - (float)h0) * 4096.0f), (_Float16)((x1 - (float)h1) * 4096.0f)};
      *reinterpret_cast<half2v*>(rb + 2 * i2) = hv;
      *reinterpret_cast<half2v*>(rb + kFarImage + 2 * i2) = lv;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // fragment element k = 32 ks + 8 g + j  ->  x16[8 bb + k]
    const unsigned off = rb_addr + (unsigned)(8 * bb + 8 * g) * 2u;
    float4v a_e0 = {0.0f, 0.0f, 0.0f, 0.0f}, a_lo = a_e0, a_e1 = a_e0, a_x1 = a_e0, a_e2 = a_e0;
    // constants (NT = 32): fragments 0-2 (e0 d0)h slices 0-2; 3-4 (e1 d1)h slices 0-1; 5-7 (e0 d0)lo slices 0-2; 8-9 (e2 d2)h slices 0-1
    // (NT = 50: FarGeom<50>)
#pragma unroll
    for (int ks = 0; ks < G::kSlices; ++ks) {
      // four fragments in flight at a time (16 VGPRs): the kernel lives in 80 registers
      half8v xh, xl, ca_, cb_;
      if (PAR_MFMA_EXP & 2) {
        xh = xl = ca_ = cb_ = half8v{(_Float16)(float)off, 0, 0, 0, 0, 0, 0, 0};
      } else {
      asm volatile("ds_read_b128 %0, %1" : "=v"(xh) : "v"(off + (unsigned)(64 * ks)));
      asm volatile("ds_read_b128 %0, %1" : "=v"(xl) : "v"(off + (unsigned)(64 * ks + 2 * kFarImage)));
      asm volatile("ds_read_b128 %0, %1" : "=v"(ca_) : "v"(ca + (unsigned)(ks * 1024)));
      asm volatile("ds_read_b128 %0, %1" : "=v"(cb_) : "v"(ca + (unsigned)((G::kLoFrag + ks) * 1024)));
      // the fragments are the asm's outputs: the wait has to name them or the MFMAs may be scheduled above it
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xh), "+v"(xl), "+v"(ca_), "+v"(cb_)::"memory");
      }
      if (!(PAR_MFMA_EXP & 1)) {
      a_e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ca_, xh, a_e0, 0, 0, 0);
      a_lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(cb_, xh, a_lo, 0, 0, 0);
      a_lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(ca_, xl, a_lo, 0, 0, 0);
      } else {
        a_e0[0] += (float)xh[0] + (float)cb_[1];
        a_lo[0] += (float)xl[0] + (float)ca_[1];
      }
      const bool has_e1 = ks >= G::kE1First && ks < G::kE1First + G::kE1Count;       // (compile-time: the loop is unrolled)
      const bool has_e2 = ks >= G::kE2First && ks < G::kE2First + G::kE2Count;
      if (has_e1 || has_e2) {
        if (!(PAR_MFMA_EXP & 2)) {
        if (has_e1) asm volatile("ds_read_b128 %0, %1" : "=v"(ca_) : "v"(ca + (unsigned)((G::kE1Frag + ks - G::kE1First) * 1024)));
        if (has_e2) asm volatile("ds_read_b128 %0, %1" : "=v"(cb_) : "v"(ca + (unsigned)((G::kE2Frag + ks - G::kE2First) * 1024)));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ca_), "+v"(cb_)::"memory");
        }
        if (!(PAR_MFMA_EXP & 1)) {
        if (has_e1) {
          a_e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ca_, xh, a_e1, 0, 0, 0);
          a_x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ca_, xl, a_x1, 0, 0, 0);
        }
        if (has_e2) a_e2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(cb_, xh, a_e2, 0, 0, 0);
        } else {
          a_e1[0] += (float)ca_[0];
          a_x1[0] += (float)cb_[0];
        }
      }
    }
    // D[row = 4 g + reg][col = bb], row m = (filter m >> 3, position m & 7): g = 0, 1 hold the e filter's positions
    // 4 (g & 1) + reg of block bb, g = 2, 3 the d filter's
    const float4v v0 = a_e0 + a_lo * kFarrowLoInv, v1 = a_e1 + a_x1 * kFarrowLoInv;
    const half4v v2 = {(_Float16)a_e2[0], (_Float16)a_e2[1], (_Float16)a_e2[2], (_Float16)a_e2[3]};
    const int fsel = g >> 1, pos4 = 8 * bb + 4 * (g & 1);
    if (PAR_MFMA_EXP & 8) {
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) far[r0 + rstep * rr] = v0[rr] + v1[rr] + (float)v2[rr];
      continue;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");           // every lane has read the image the bank overwrites
    __builtin_amdgcn_wave_barrier();
    *reinterpret_cast<float4v*>(bank + fsel * 128 + pos4) = v0;                 // e0 | d0
    *reinterpret_cast<float4v*>(bank + (2 + fsel) * 128 + pos4) = v1;           // e1 | d1
    *reinterpret_cast<half4v*>(bankh + fsel * 128 + pos4) = v2;                 // e2 | d2
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = r0 + rstep * rr;
      const int pi = (c[r] - p0) / LS;                               // 0 .. 127 (positions advance by at most one per output)
      const float e = fmaf(q[r], fmaf(q[r], (float)bankh[pi], bank[2 * 128 + pi]), bank[pi]);
      const float d = fmaf(q[r], fmaf(q[r], (float)bankh[128 + pi], bank[3 * 128 + pi]), bank[128 + pi]);
      far[r] = fmaf(s[r], e, d);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");           // the image is rewritten by the next pair
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// float16 suits a wave's span when every sample is finite and below 32768 and the loudest one is at least 2^-10 (wave-uniform)
__device__ __forceinline__ bool span_suits_f16(const float* __restrict__ tile, const int nlim, const int l) {
  float vmax = 0.0f;
  bool fin = true;
  for (int i = l; i < nlim; i += kWave) {
    const float a = fabsf(tile[i]);
    // below 32768: from there float16's spacing is 32 and the residual x - hi can reach 16, which x 4096 leaves float16's range
    // (found again by a 2e4 x noise test in r04).  False for NaN and Inf too.
    fin = fin && a < 32768.0f;
    vmax = fmaxf(vmax, a);
  }
  return __all(fin) && __any(vmax >= 0.0009765625f);
}

// the taps the bank leaves out: the reciprocal rows n = 1 .. 4, the centre tap and tap -NT (weight 0: it only carries the
// reference's NaN poison); `far` is the bank's sum, still scaled by 32
template <int NT, int R, int LS = 1, bool OPQ = false>
__device__ __forceinline__ void taps_unity_near(const float* __restrict__ tile, const int (&c)[R], const float (&s)[R],
                                                const float (&q)[R], const float (&far)[R], float (&res)[R]) {
  using T = TapTab<NT>;
  lds_cfloat* tl = (lds_cfloat*)tile;
  constexpr float b0 = T::B[0];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    lds_cfloat* base = opaque_lds<OPQ>(tl + c[r] - NT * LS);
    float e0 = fmaf(base[0], 0.0f, 0.0f), d0 = 0.0f;
    static_for<kPoly2From - 1>([&](auto idx) {
#pragma clang fp contract(off)
      constexpr int n = 1 + decltype(idx)::value;
      static_assert(T::mode[n] == kRcp, "reciprocal rows");
      constexpr float fn = (float)n;
      const float sp = base[(NT + n) * LS], sm = base[(NT - n) * LS];
      const float D = sp - sm, E = sp + sm;
      const float x = q[r] * T::B[n] + T::A[n];                 // two literals: a multiply and an add, both fast
      const float Rn = fast_rcp(x);
      const float DR = D * Rn;
      if constexpr (n & 1) {
        e0 = fmaf(-E, Rn, e0);
        d0 = fmaf(DR, -fn, d0);
      } else {
        e0 = fmaf(E, Rn, e0);
        d0 = fmaf(DR, fn, d0);
      }
    });
    const float centre = tile[c[r]] * fast_rcp(s[r] * b0);
    res[r] = -sinpi_half(s[r]) * (centre + fmaf(far[r], FarGeom<NT == 50 ? 50 : 32>::kScaleInv, fmaf(s[r], e0, d0)));
  }
}

// ---- output placement -------------------------------------------------------------------------------
// What the tap loops need per output: the integer window centre (relative to a block-uniform EVEN anchor, so that
// indices are int32 and round-half-even ties equal rint(p)), the sub-sample shift, fc and 1 - fc.

// From a float64 position and the distance to the next one: util/resampling.py:66-79 (the position-array form, and
// every output the closed form below hands over).
__device__ __forceinline__ void place_from_pos(double p, double dp, double anchor_d, int& c, float& s, float& fc, float& dd,
                                               bool& lowfc, bool& wild) {
  const double rel = p - anchor_d;             // exact to ~1e-13: anchor is within a tile's span of p
  const double rf = rint(rel);
  wild = !(fabs(rel) < 1.0e9);
  c = wild ? 0 : (int)rf;
  const float sh = (float)(rel - rf);          // = p - rint(p)
  s = (sh == 0.0f) ? 1e-20f : sh;              // np.sinc's own 0 -> 1e-20 substitution
  const bool one = !(dp > 1.0);                // fc == 1 (also catches the 1e-12 floor)
  // fc < 1/8 (an 8x slow-down of the read head and more): the output is a long average, small against the
  // signal, and float32 tap arithmetic (abs. error ~1e-6 of the signal level) would exceed 1e-5 of the OUTPUT
  // peak -- those lanes take the float64 path (not an audio-restoration regime; found by tools/fuzz_resampler.py)
  lowfc = dp > 8.0;
  const float inv = fast_rcp((float)(dp > 1e-12 ? dp : 1e-12));
  fc = one ? 1.0f : inv;
  dd = one ? 0.0f : (float)(dp - 1.0) * inv;   // 1 - fc without cancellation
}

// EXACT position of output j of segment i and the period to its successor, as numpy produces them
// (util/resampling.py:120-126): the cumsum restarts from the checkpoint below j and repeats the reference's own
// sequential float64 adds.  Slow (up to kCk + 1 correctly rounded reciprocals per output); only outputs the closed
// form cannot vouch for come here.  (Pointers by value, result by value: a noinline function taking references would
// force the kernel's argument struct into scratch memory.)
struct PosDp {
  double p, dp;
};
// A lazy plan (pos_plan.h) holds no checkpoints (ckp == nullptr): the walk starts at the segment's first step (<= kLazyMaxN
// of them; ~1 output in 10^6 comes here).
__device__ __noinline__ PosDp place_exact(const double* __restrict__ speeds, const int64_t* __restrict__ seg_start,
                                          const double* __restrict__ seg_off, const double* __restrict__ ckp, long long i,
                                          long long j, long long len_out) {
#pragma clang fp contract(off)
  const long long start = seg_start[i], n = seg_start[i + 1] - start;
  const long long k = j - start, b = ckp ? k / kCk : 0;
  const Ramp r = make_ramp(speeds[i], speeds[i + 1], n);
  const double off = seg_off[i];
  double c = b ? ckp[ck_slot0(start, i) + b] : 0.0;
  double cprev = c;
  // eight reciprocals at a time -- they do not depend on one another -- then the adds in the reference's order: a division per
  // trip of the loop made the lazy plans' walk (up to kLazyMaxN steps) ~150 cycles per step, and the file's LAST output, which
  // always comes here, held the tile list behind the streaming kernel for 50 us per launch (r05)
  long long v = b * kCk;
  for (; v + 8 <= k + 1; v += 8) {
    double rr[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) rr[u] = ramp_recip((double)(v + u), r);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      cprev = c;
      c = c + rr[u];
    }
  }
  for (; v <= k; ++v) {
    cprev = c;
    c = c + ramp_recip((double)v, r);
  }
  PosDp o;
  o.p = c + off;
  if (j + 1 < len_out) {
    double pn;
    if (k + 1 < n) {
      pn = (c + ramp_recip((double)(k + 1), r)) + off;
    } else {                                             // first output of the next segment (cumsum restarts at 0)
      const long long n2 = seg_start[i + 2] - seg_start[i + 1];
      const Ramp r2 = make_ramp(speeds[i + 1], speeds[i + 2], n2);
      pn = (0.0 + ramp_recip(0.0, r2)) + seg_off[i + 1];
    }
    o.dp = pn - o.p;
  } else {                                               // last output reuses the previous period (:76-77)
    const double pp = k > 0 ? cprev + off : off;         // k == 0: the offset IS the previous segment's last position
    o.dp = o.p - pp;
  }
  return o;
}

// Closed-form placement (the fast path of the fused kernel).  The reference's position is
//     p_j = fl( off_i + c_k ),   c_k = sum_{v <= k} 1/speed_v   (sequential float64 adds)
// and the plan stores c at every kCk-th step.  Behind the checkpoint lie w = (k mod kCk) + 1 reciprocals of a LINEAR
// ramp, whose sum is  w r_m (1 + (r_m step)^2 (w^2 - 1)/12)  up to a fourth-order remainder (r_m = reciprocal at the
// midpoint step; the plan marks a segment `fast` only where that remainder is < 2e-10 samples).  Everything is kept
// relative to the anchor and to rint(off_i), so the float64 operands are small and the result is accurate to ~1e-10 --
// MORE accurate than the reference's own p_j, which carries the rounding of its last add (half an ulp of p: 6e-8 at
// 7e8).  Consequences: shift differs from the reference's by <= ulp(p)/2 (output: ~2e-7 relative at worst), and
// rint(p) can differ only when p lies within that distance of a half-integer -- such outputs are flagged (`exact`) and
// recomputed with the reference's own arithmetic, so every window centre is the reference's.
// The period to the next position is 1/speed at the next step, so fc = min(1, speed_next) needs no division at all.
__device__ __forceinline__ void place_fast(const FusedArgs& fa, long long i, long long j, long long len_out, long long anchor,
                                           double tol, int& c, float& s, float& fc, float& dd, bool& lowfc, bool& wild,
                                           bool& exact, const bool lazy) {
  const long long start = fa.seg_start[i];
  const SegFast sf = fa.seg_fast[i];
  const double s0 = fa.speeds[i];
  const int k = (int)(j - start);
  // lazy plans: no checkpoint, all k + 1 reciprocals in closed form (the plan vouches for every segment: pos_plan.h)
  const int u = lazy ? k : (k & (kCk - 1));              // steps between the checkpoint and this output
  const long long b = lazy ? 0 : (k >> 3);
  static_assert(kCk == 8, "k >> 3");
  const double ckv = b ? fa.ck[ck_slot0(start, i) + b] : 0.0;
  const long long dA = sf.A - anchor;
  wild = !(dA > -2000000000ll && dA < 2000000000ll);
  const double base = ((double)(int)dA + sf.foff) + ckv;
  const double tm = (double)k - 0.5 * (double)u;         // midpoint of steps k-u .. k
  const double bsm = fma(sf.step, tm, s0);
  double r = __builtin_amdgcn_rcp(bsm);                  // ~2^-26; one Newton step -> ~1e-15
  r = fma(r, fma(-bsm, r, 1.0), r);
  if (lazy) r = fma(r, fma(-bsm, r, 1.0), r);            // the sum of up to kLazyMaxN terms hangs on this one reciprocal
  const double z = r * sf.step;
  const double cw = (double)(u * (u + 2)) * (1.0 / 12.0);   // (w^2 - 1)/12 with w = u + 1
  const double w2 = (double)(u + 1) * (double)(u + 1), zz = z * z;
  // (the fourth-order term only counts over the hundreds of steps of a lazy plan's closed form: lazy_prefix)
  const double prel = fma(r * (double)(u + 1), fma(zz * cw, lazy ? fma(zz * 0.05, fma(3.0, w2, -7.0), 1.0) : 1.0, 1.0), base);
  const double rf = rint(prel);
  const double shd = prel - rf;
  if (!(fabs(prel) < 1.0e9)) wild = true;
  c = wild ? 0 : (int)rf;
  const float sh = (float)shd;
  s = (sh == 0.0f) ? 1e-20f : sh;
  // rint(p) is the reference's when p is farther from a half-integer than the reference's own roundings: half an ulp of
  // p (tol) plus, per add behind the checkpoint, half an ulp of the running sum (matters for segments of > 10^7 outputs)
  // (lazy: the reference's k + 1 sequential adds against the closed form, lazy_bound)
  const double smin_ = s0 < s0 + sf.step * (double)(sf.n - 1) ? s0 : s0 + sf.step * (double)(sf.n - 1);
  exact = !sf.fast || !(fabs(fabs(shd) - 0.5) > (lazy ? tol + lazy_bound((double)(k + 1), smin_ * 0.999) : fma(ckv, 9.6e-16, tol)));
  if (lazy && exact && sf.fast && !wild) {
    // A near-tie of a lazy plan, settled WITHOUT walking the segment's cumsum: the reference's position is p = fl(C_k + off)
    // (util/resampling.py:125) with C_k within lazy_bound of the closed form.  TwoSum gives the exact error of that one add:
    // unless the true sum lies within the bound of a rounding midpoint of p's grid (1.2e-7 at 7e8 against a bound of ~4e-12:
    // 1 near-tie in 10^4), fl() lands on the same float64 whichever C_k numpy had -- p bit for bit, and rint(p) with it.
    // (Walking took ~10 us per near-tie on one lane: most of k_sinc_fused_list's time behind the streaming kernel.)
    const double Ck = r * (double)(u + 1) * fma(zz * cw, fma(zz * 0.05, fma(3.0, w2, -7.0), 1.0), 1.0);
    const double off = fa.seg_off[i];
    const double sum = Ck + off;
    const double bb = sum - Ck;
    const double err = (Ck - (sum - bb)) + (off - bb);                     // Ck + off = sum + err exactly
    const int e2 = (int)((__double_as_longlong(sum) >> 52) & 0x7ff);        // ulp(sum) = 2^(e2 - 1075)
    const double half_ulp = ldexp(1.0, e2 - 1076);
    const double B = lazy_bound((double)(k + 1), smin_ * 0.999) + 8.0 * 0x1p-52 * Ck;
    const double rel = sum - (double)anchor;                                // exact: anchor is an integer within a tile of sum
    if (e2 > 1 && e2 < 2046 && half_ulp - fabs(err) > B && fabs(rel) < 1.0e9) {
      const double rr = rint(rel);
      c = (int)rr;
      const float sh2 = (float)(rel - rr);
      s = (sh2 == 0.0f) ? 1e-20f : sh2;
      exact = false;
    }
  }
  int kn = k + 1 < sf.n ? k + 1 : sf.n - 1;              // the last output of a segment looks at the next segment's
  if (j + 1 >= len_out) kn = k;                          // first speed = this ramp's end; the global last one back
  const double bsn = fma(sf.step, (double)kn, s0);
  const bool one = !(bsn < 1.0);
  fc = one ? 1.0f : (float)bsn;
  dd = one ? 0.0f : (float)(1.0 - bsn);
  lowfc = bsn < 0.125;
}

// The tap loops of one lane: NS (output, channel) slots, two at a time where the loops carry 6-10 live values per slot
// (one pass over four spilled 48 B/lane = as much HBM write traffic as the output).  `all_unity`: fc == 1 for every lane
// of the wave.
template <int NTC, int NS, int LS = 1, bool OPQ = false>
__device__ __forceinline__ void run_taps(const float* __restrict__ tile, const int (&cs)[NS], const float (&ss)[NS],
                                         const float (&fcs)[NS], const float (&dds)[NS], const bool all_unity,
                                         const int NT, const float4* __restrict__ tab, const TapModes tmd,
                                         float (&res)[NS]) {
  static_assert(NS == 2 || NS == 4, "2 or 4 slots per lane");
  if constexpr (NTC == 0) {
    if (all_unity) {
      taps_unity<NS>(tile, cs, ss, NT, tab, tmd, res);
      return;
    }
  }
#pragma unroll
  for (int h = 0; h < NS / 2; ++h) {
    const int ca[2] = {cs[2 * h], cs[2 * h + 1]};
    const float sa[2] = {ss[2 * h], ss[2 * h + 1]};
    float ra[2];
    if (all_unity) {
      if constexpr (NTC > 0) taps_unity_ct<NTC, 2, LS, OPQ>(tile, ca, sa, NT, ra);
    } else {
      const float fa_[2] = {fcs[2 * h], fcs[2 * h + 1]};
      const float da[2] = {dds[2 * h], dds[2 * h + 1]};
      if constexpr (NTC > 0) taps_general_ct<NTC, 2, LS, OPQ>(tile, ca, sa, fa_, da, NT, ra);
      else taps_general<2>(tile, ca, sa, fa_, da, NT, tab, tmd, ra);
    }
    res[2 * h] = ra[0];
    res[2 * h + 1] = ra[1];
  }
}

// ---- the tile body of the position-array form --------------------------------------------------------
// Stages the tile's input span once, runs the tap loops, writes the outputs.  slow_pos(r, p, dp) yields the float64
// position of the lane's r-th output for the (rare) lanes that leave the float32 path.
// NCH = 2: two channels of one file (same positions) in one launch.  A lane then owns 2 outputs x 2 channels instead of
// 4 outputs x 1: the register state and the per-lane ILP are those of the mono kernel, the workgroup has 512 threads
// for the same 1024-output tile, and everything that depends only on the POSITION -- placement, window-centre search,
// and (because both channel slots of an output carry the very same shift / fc values) the tap weights themselves -- is
// computed once for both channels.
template <int NCH, int NTC, class SlowPos>
__device__ __forceinline__ void sinc_tile_body(float* __restrict__ tile, int* __restrict__ red, const int t, const int64_t j0,
                                               const int64_t j_end, const long long anchor, int (&c)[kSincR / NCH],
                                               const float (&s)[kSincR / NCH], const float (&fc)[kSincR / NCH],
                                               const float (&dd)[kSincR / NCH], const bool (&valid)[kSincR / NCH],
                                               const bool (&lowfc)[kSincR / NCH], const bool unity_in, const bool wild,
                                               const float* __restrict__ sig, const float* __restrict__ sig1,
                                               const int64_t sig_stride, const int64_t len_in, const int NT,
                                               const float4* __restrict__ tab, const TapModes tmd, float* __restrict__ out,
                                               float* __restrict__ out1, const int64_t out_stride, SlowPos slow_pos) {
  constexpr int kBlk = kSincBlock * NCH;        // threads per workgroup
  constexpr int kOut = kSincR / NCH;            // outputs per lane; kOut * NCH = kSincR (output, channel) slots
  constexpr int cap = kSincCap;
  float res[kSincR];                            // one per (output, channel) slot: slot = output * NCH + channel
  bool fastlane[kOut];
  int mn = INT_MAX, mx = INT_MIN;
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    if (valid[r]) {
      mn = c[r] < mn ? c[r] : mn;
      mx = c[r] > mx ? c[r] : mx;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int a = __shfl_xor(mn, o, kWave), b = __shfl_xor(mx, o, kWave);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if (__any(wild)) mn = INT_MIN;                    // poisons the span test below for the whole block
  if ((t & (kWave - 1)) == 0) {
    red[t / kWave] = mn;
    red[kBlk / kWave + t / kWave] = mx;
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < kBlk / kWave; ++w) {
    mn = red[w] < mn ? red[w] : mn;
    mx = red[kBlk / kWave + w] > mx ? red[kBlk / kWave + w] : mx;
  }
  // the tap loops run in chunks of kChunk and may touch up to kChunk-1 taps beyond +-(NT-1); those
  // carry an exactly-zero weight (R_n = rcp(inf)) but must read finite data: stage a kChunk margin.
  const int margin = NT + kChunk;
  const long long span = (long long)mx - (long long)mn + 2ll * margin;     // <= kSincCap for the LDS path
  const bool staged = mn != INT_MIN && span <= cap;
  const long long lo = anchor + mn - margin;        // signal index of tile[0]
  if (staged) {
    for (int q = t; q < (int)span; q += kBlk) {
      const long long g = lo + q;
      const bool inside = g >= 0 && g < (long long)len_in;
      if (PAR_SINC_EXP & 16) {
        tile[q] = 0.5f;
        if (NCH == 2) tile[cap + q] = 0.25f;
        continue;
      }
      tile[q] = inside ? sig[g * sig_stride] : 0.0f;
      if (NCH == 2) tile[cap + q] = inside ? sig1[g * sig_stride] : 0.0f;      // channel 1 right behind channel 0
    }
  }
  __syncthreads();

  // leading-edge outputs (ind < NT) keep the reference's mis-aligned taps: float64 slow path.
  bool anyfast = false, unity = unity_in;
  const long long edge = (long long)NT - anchor;    // ind >= NT  <=>  rel index >= edge
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    fastlane[r] = valid[r] && staged && (long long)c[r] >= edge && !lowfc[r];
    c[r] = fastlane[r] ? c[r] - mn + margin : margin;      // LDS index of the window centre (idle lanes: harmless)
    anyfast = anyfast || fastlane[r];
  }
  // (output, channel) slots: channel ch of an output reads the tile `ch * cap` floats further on; shift, fc and
  // 1 - fc are the SAME values for both slots of an output, so the compiler evaluates their tap weights once
  int cs[kSincR];
  float ss[kSincR], fcs[kSincR], dds[kSincR];
#pragma unroll
  for (int sl = 0; sl < kSincR; ++sl) {
    cs[sl] = c[sl / NCH] + (sl % NCH) * cap;
    ss[sl] = s[sl / NCH];
    fcs[sl] = fc[sl / NCH];
    dds[sl] = dd[sl / NCH];
    res[sl] = 0.0f;
  }
  if (!(PAR_SINC_EXP & 1) && __any(anyfast)) run_taps<NTC, kSincR>(tile, cs, ss, fcs, dds, __all(unity), NT, tab, tmd, res);
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    const int64_t j = j0 + t + (int64_t)r * kBlk;
    if (j >= j_end) continue;
    double pj = 0.0, dpj = 1.0;
    if (!fastlane[r]) slow_pos(r, pj, dpj);
    float2 two = make_float2(0.0f, 0.0f);
    if (NCH == 2 && !fastlane[r]) two = sinc_two_f64(pj, dpj, sig, sig1, sig_stride, len_in, NT);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      float v = res[r * NCH + ch];
      if (!fastlane[r]) v = NCH == 2 ? (ch ? two.y : two.x) : sinc_one_f64(pj, dpj, sig, sig_stride, len_in, NT);
      if ((PAR_SINC_EXP & 32) && v != 12345.678f) continue;
      (ch ? out1 : out)[j * out_stride] = v;
    }
  }
}

// 6 waves/SIMD (80 VGPRs) measured best: 4 -> 1.39 ms, 5 -> 1.29, 6 -> 1.25, 7 -> 1.32, 8 -> 1.59 (spills) per
// 115 M outputs.  Fully unrolling the tap loop (compile-time NT) was tried twice and spills badly.
//
// Position-array form (operator slot #2, sinc_wrapper): reads the caller's float64 sample_at.
template <int NTC>
__global__ __launch_bounds__(kSincBlock, 6) void k_sinc_pos(const double* __restrict__ pos, int64_t len_out,
                                                             const float* __restrict__ sig, int64_t sig_stride,
                                                             int64_t len_in, int NT, const float4* __restrict__ tab,
                                                             TapModes tmd, float* __restrict__ out, int64_t out_stride,
                                                             int64_t j_begin, int64_t j_end) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  __shared__ int red[2 * (kSincBlock / kWave)];
  const int t = threadIdx.x;
  const int64_t j0 = j_begin + (int64_t)blockIdx.x * kSincTile;     // this launch covers outputs [j_begin, j_end)
  const double p0 = pos[j0];
  const long long anchor = (fabs(p0) < 4.0e18) ? (llrint(p0) & ~1ll) : 0ll;
  const double anchor_d = (double)anchor;
  int c[kSincR];
  float s[kSincR], fc[kSincR], dd[kSincR];
  bool valid[kSincR], lowfc[kSincR];
  bool unity = true, wild = false;
#pragma unroll
  for (int r = 0; r < kSincR; ++r) {
    const int64_t j = j0 + t + (int64_t)r * kSincBlock;
    valid[r] = j < j_end;
    lowfc[r] = false;
    c[r] = 0;
    s[r] = 0.25f;
    fc[r] = 1.0f;
    dd[r] = 0.0f;
    if (valid[r]) {
      const double p = pos[j];
      // last output reuses the previous period (util/resampling.py:76-77)
      const double dp = (j + 1 < len_out) ? pos[j + 1] - p : p - pos[j - 1];
      bool w;
      place_from_pos(p, dp, anchor_d, c[r], s[r], fc[r], dd[r], lowfc[r], w);
      wild = wild || w;
      unity = unity && fc[r] == 1.0f;
    }
  }
  sinc_tile_body<1, NTC>(tile, red, t, j0, j_end, anchor, c, s, fc, dd, valid, lowfc, unity, wild, sig, (const float*)nullptr,
                    sig_stride, len_in, NT, tab, tmd, out, (float*)nullptr, out_stride,
                    [&](int r, double& p, double& dp) {
                      const int64_t j = j0 + t + (int64_t)r * kSincBlock;
                      p = pos[j];
                      dp = (j + 1 < len_out) ? pos[j + 1] - p : p - pos[j - 1];
                    });
}

// Waves per workgroup of the fused kernel.  Waves never meet, so any number works.  The mono NT = 32 kernel fetches its
// 10 KB of Farrow constants once per workgroup (load + the kernel's only barrier: with both removed an all-fast tape runs
// 8 % faster, PAR_MFMA_EXP=16), but spreading that over 8 / 12 / 16 waves per workgroup measured 0.72 / 0.73 / 0.88 ms
// against 0.72 at 4 (fewer, larger workgroups fill the CUs less evenly): 4 it stays.
// The stereo NT = 32 kernel can keep them too (PAR_SINC_MFMA_STEREO=1, r04): its waves own 128 outputs x 2 channels, so a
// tile is EIGHT waves -- one workgroup, 40 KB of spans + 10 KB of constants, three of them per CU (24 waves, as before).
// Measured on a 300-s interleaved file (tools/exp/stereo_only.py, Gsamples/s over both channels, vector loops -> bank):
// fc = 1 tape 166.0 -> 174.6, fc < 1 tape 109.8 -> 105.5, the benchmark's tape 150.7 -> 148.5.  The two channels of an
// output share one set of tap weights, so the vector loops already cost half of mono's per sample and the bank has little
// left to win, while every tile pays the constants and the barrier: off.
#ifndef PAR_SINC_MFMA_STEREO
#define PAR_SINC_MFMA_STEREO 0
#endif
__host__ __device__ constexpr bool fused_is_farrow(int NCH, int NTC, int NS) {
  return ((NCH == 1 && (NTC == 32 || NTC == 50)) || (NCH == 2 && NTC == 32 && PAR_SINC_MFMA_STEREO)) && NS == 4 && PAR_SINC_MFMA;
}
__host__ __device__ constexpr int fused_waves(int NCH, int NTC, int NS) {
  return fused_is_farrow(NCH, NTC, NS) ? (NCH == 1 ? PAR_FARROW_WAVES : kSincTile / (kWave * NS / NCH)) : kSincBlock / kWave;
}
// LDS floats per wave and channel: room for the span of 64 NS/NCH outputs at speeds up to ~3.7 plus the halo
__host__ __device__ constexpr int fused_capw(int NS, int NCH) { return (kWave * NS / NCH) * 4 >= 1024 ? 1024 : (NS / NCH == 2 ? 640 : 448); }
// One wave's share of the fused kernel.  HOT: a full wave (all kWaveOut outputs exist) on unit-stride signal and output --
// the case every wave but a file's last one is in: no validity masks, no index clamps, stride-free addresses, and the
// span goes to LDS by direct loads.
template <int NCH, int NTC, int NS, bool HOT>
__device__ __forceinline__ void fused_wave(const int64_t len_out, const float* __restrict__ sig, const float* __restrict__ sig1,
                                           const int64_t sig_stride, const int64_t len_in, const int NT,
                                           const float4* __restrict__ tab, const TapModes tmd, float* __restrict__ out,
                                           float* __restrict__ out1, const int64_t out_stride, const FusedArgs& fa,
                                           float* __restrict__ tile, const int l, const int64_t jw, const int nrem,
                                           float* __restrict__ lds_front) {
  constexpr int kOut = NS / NCH;                    // outputs per lane
  constexpr int kWaveOut = kWave * kOut;            // outputs per wave: 256 / 128 / 64
  constexpr int capw = fused_capw(NS, NCH);         // floats of one channel's span a wave may stage
  // HOT stereo = an interleaved file (sig1 == sig + 1, stride 2, same for the output): the span is staged as it lies in
  // memory, sample i of channel ch at LDS word 2 i + ch, and the two channel slots of an output read adjacent words
  constexpr int LS = (HOT && NCH == 2) ? 2 : 1;     // LDS words per sample
  constexpr int CHO = LS == 2 ? 1 : capw;           // LDS offset of channel 1
  const int t = threadIdx.x;
  const int64_t T = jw / kSincTile;
  PAR_PHASE_BEGIN();
  // 1. records of this lane's outputs: block (jw >> 5) + (l >> 5) + 2 r, u = l & 31 for every r.  Unconditional 16-byte
  // loads off a wave-uniform base (indices past the file's last block are clamped to it); the 32 lanes of a block read
  // the same 16 bytes
  const int gmax = nrem > 0 ? (nrem - 1) >> kRecShift : 0;
  const uint4* rp = reinterpret_cast<const uint4*>(fa.rec + (jw >> kRecShift));
  uint4 ra[kOut];
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    const int gr = (l >> kRecShift) + (kWave / kRec) * r;
    ra[r] = rp[HOT ? gr : (gr < gmax ? gr : gmax)];
  }
  // 2. tile header: the anchor all window centres are relative to
  const TileHdr hd = fa.hdr[T];
  const long long anchor = hd.anchor;
  const bool lazy = (hd.flags & kTileLazy) != 0;     // the plan holds no cumsum checkpoints (pos_plan.h)
  // 2b. mono NT = 32 kernel: the constant fragments of the unity path's Farrow bank, in workgroups whose tile may hold fc = 1
  // outputs (TileHdr hint, written by the plan; a workgroup of that kernel IS one tile, so the choice is workgroup-uniform).
  // The ten 1-KB fragments go from L2 straight into the front of the workgroup's LDS, asynchronously and BEHIND the record
  // loads: wave w moves fragments w, w + 4, w + 8 with one 16-byte DMA per lane each; they are waited for behind the
  // placement, where the kernel's only workgroup barrier sits.  (Copied through registers in front of everything they cost
  // 9 % on an all-fast tape and 1 % on tiles that never use them, tools/exp/unity_only.py.)
  constexpr bool kFarrow = fused_is_farrow(NCH, NTC, NS);
  const unsigned far_consts = lds_addr_of(lds_front);
  bool farrow_wg = false;
  if constexpr (kFarrow && !(PAR_MFMA_EXP & 16)) {
    static_assert(!kFarrow || fused_waves(NCH, NTC, NS) * kWaveOut == kSincTile, "one tile per workgroup");
    farrow_wg = (hd.flags & kTileMayUnity) != 0;
    if (farrow_wg && !(PAR_MFMA_EXP & 32)) {
      constexpr int kWaves = fused_waves(NCH, NTC, NS);
      const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
      using FG = FarGeom<NTC == 50 ? 50 : 32>;
      const char* src = reinterpret_cast<const char*>(FG::table()) + l * 16;
#pragma unroll
      for (int k = 0; k < (FG::kFrags + kWaves - 1) / kWaves; ++k) {
        const int f = wv + k * kWaves;
        if (f < FG::kFrags)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f * 1024),
                                           (__attribute__((address_space(3))) void*)(lds_front + f * 256), 16, 0, 0);
      }
    }
  }
#if PAR_SINC_EXP & 256               // timing experiment: the span DMA issued NOW from a guessed start (6 rows), not after placement
  if constexpr (HOT && NCH == 1) {
    long long lo_g = anchor + (jw - T * kSincTile) - 40;
    lo_g = lo_g < 0 ? 0 : (lo_g > (long long)len_in - 512 ? (long long)len_in - 512 : lo_g);
    const float* gp = sig + lo_g + l;
    static_for<6>([&](auto qi) {
      constexpr int q = decltype(qi)::value;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, (__attribute__((address_space(3))) void*)tile, 4,
                                       q * kWave * 4, 0);
    });
  }
#endif
  PAR_PHASE_MARK(0);                 // issue of the record loads, arrival of the header
  // 3. placement
  const double tol = (fabs((double)anchor) + 4.2e6) * 1.2e-16 + 2.0e-10;     // the reference's own rounding of p (half an ulp)
  const float tolf = (float)tol + 2.0e-7f;                                    // + float32 evaluation of the block polynomial
  const int u = l & (kRec - 1), uc = u - kRec / 2;                            // centred block variable u' = -16 .. 15
  const float uf = (float)uc, u2f = uf * uf, tw1 = 2.0f * uf + 1.0f, tw0 = 2.0f * uf - 1.0f;
  int c[kOut];
  float s[kOut], ep[kOut];           // ep = max(period - 1, 0): fc = 1 / (1 + ep) and 1 - fc = ep fc are formed where the general tap path needs them
  bool valid[kOut], lowfc[kOut], redo[kOut], slow[kOut], second[kOut], last[kOut];
  bool unity = true, wild = (hd.flags & 1) != 0, anyredo = false, anysecond = false, anycubic = false;
  int Irel[kOut];
  float F[kOut], e1[kOut], e2[kOut], frac[kOut], e[kOut];
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    valid[r] = HOT || l + kWave * r < nrem;
    second[r] = (unsigned)u > (ra[r].x & 31u);           // u >= ustar, ustar - 1 in bits 0-4
    anysecond = anysecond || second[r];
    Irel[r] = (int)ra[r].x >> 16;
    F[r] = __uint_as_float(ra[r].y);
    e1[r] = __uint_as_float(ra[r].z);
    e2[r] = __uint_as_float(ra[r].w);
  }
  if (__any(anysecond)) {            // a segment starts inside some lane's block: those lanes take its second piece
    const uint4* rp2 = reinterpret_cast<const uint4*>(fa.rec2 + (jw >> kRecShift));
#pragma unroll
    for (int r = 0; r < kOut; ++r) {
      if (second[r]) {
        const int gr = (l >> kRecShift) + (kWave / kRec) * r;
        const uint4 q = rp2[HOT ? gr : (gr < gmax ? gr : gmax)];
        Irel[r] = (int)q.x >> 16;
        F[r] = __uint_as_float(q.y);
        e1[r] = __uint_as_float(q.z);
        e2[r] = __uint_as_float(q.w);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    const unsigned m = ra[r].x;
    // the segment's last output (its period is the PREVIOUS increment): u = ustar - 1 where E0 says so, or the block's
    // last output of a second piece whose segment ends there (E1)
    last[r] = second[r] ? (u == kRec - 1 && (m & kRecE1)) : ((unsigned)u == (m & 31u) && (m & kRecE0));
    slow[r] = valid[r] && ((second[r] ? (m & kRecSlow1) : (m & kRecSlow0)) != 0u);
    anycubic = anycubic || (m & kRecCubic);
    frac[r] = fmaf(u2f, e2[r], fmaf(uf, e1[r], F[r]));
    e[r] = fmaf(e2[r], last[r] ? tw0 : tw1, e1[r]);      // period to the next position, minus 1
  }
  if (__any(anycubic)) {             // steeper ramps: the cubic term of the block polynomial, e3 = (4/3) e2^2 / (1 + e1)
    const float u3f = u2f * uf, tc1 = fmaf(3.0f, u2f, fmaf(3.0f, uf, 1.0f)), tc0 = fmaf(3.0f, u2f, fmaf(-3.0f, uf, 1.0f));
#pragma unroll
    for (int r = 0; r < kOut; ++r) {
      if (ra[r].x & kRecCubic) {
        const float e3 = 1.33333333f * e2[r] * e2[r] * (1.0f - e1[r]);
        frac[r] = fmaf(u3f, e3, frac[r]);
        e[r] = fmaf(e3, last[r] ? tc0 : tc1, e[r]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    const float ri = rintf(frac[r]);
    const float sh = frac[r] - ri;
    c[r] = Irel[r] + uc + (int)ri;
    s[r] = (sh == 0.0f) ? 1e-20f : sh;              // np.sinc's own 0 -> 1e-20 substitution
    // fc = min(1, 1/period) and 1 - fc without a compare and two selects: a period <= 1 (or NaN) clamps its excess to 0,
    // and v_rcp_f32(1.0f) is exactly 1.0f: the wave takes the fc == 1 tap path when 1 + ep rounds to 1 in every lane
    ep[r] = fmaxf(e[r], 0.0f);
    // fc < 1/8 cannot come out of a record: the plan flags a piece `slow` unless its period is within 1/32 of 1
    // (k_block_rec: |a1m1| <= 0.03125, second pieces 0.971 <= speed <= 1.031); redone outputs set it in place_fast
    lowfc[r] = false;
    redo[r] = slow[r] || (valid[r] && !(fabsf(fabsf(sh) - 0.5f) > tolf));
    anyredo = anyredo || redo[r];
    if (PAR_SINC_EXP & 2) {
      c[r] = (int)(jw + l + kWave * r - anchor) + 64;
      s[r] = 0.3f - 1e-4f * (float)(t & 63);
      ep[r] = (PAR_SINC_EXP & 4) ? 0.005f / 0.995f : 0.0f;
      lowfc[r] = false;
      redo[r] = false;
      anyredo = false;
    }
  }
  if (__any(anyredo)) {      // rare: a rounding tie to settle, or a block outside the record model
#pragma unroll
    for (int r = 0; r < kOut; ++r) {
      if (redo[r]) {
        const long long j = jw + l + (int64_t)r * kWave;
        long long i = hd.iT;
        while (i + 1 < fa.nseg && fa.seg_start[i + 1] <= j) ++i;
        bool ex = !slow[r], w = false;
        float fcr = 1.0f, ddr = 0.0f;
        // (lazy plans: near-ties go through place_fast too -- it settles all but 1 in 10^4 of them without the walk)
        if (slow[r] || lazy) place_fast(fa, i, j, len_out, anchor, tol, c[r], s[r], fcr, ddr, lowfc[r], w, ex, lazy);
        if (ex) {
          const PosDp e = place_exact(fa.speeds, fa.seg_start, fa.seg_off, lazy ? nullptr : fa.ck, i, j, len_out);
          place_from_pos(e.p, e.dp, (double)anchor, c[r], s[r], fcr, ddr, lowfc[r], w);
        }
        ep[r] = ddr / fcr;                                 // (1 - fc) / fc = period - 1 (0 when fc == 1)
        wild = wild || w;
      }
    }
  }
  int cmin = INT_MAX, cmax = INT_MIN;
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    unity = unity && (1.0f + ep[r] == 1.0f || !valid[r]);
    if (valid[r]) {
      cmin = c[r] < cmin ? c[r] : cmin;
      cmax = c[r] > cmax ? c[r] : cmax;
    }
  }
  PAR_PHASE_MARK(1);                 // placement (waits for the records)
  if constexpr (fused_is_farrow(NCH, NTC, NS) && PAR_FARROW_BARRIER_EARLY) {
    if (farrow_wg && !(PAR_MFMA_EXP & 64)) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
  }
  // 4. the wave's input span.  Positions increase with the output index: the first centre is lane 0's first output,
  // the last one the last valid lane's last output (one sample of slack: a centre redone exactly may move by one).
  // the tap loops run in chunks of kChunk and may touch up to kChunk-1 taps beyond +-(NT-1); those carry an
  // exactly-zero weight but must read finite data: stage a kChunk margin
  const int margin = NT + kChunk + 1;
  const unsigned long long vmask = HOT ? ~0ull : __ballot(valid[0]);
  const int lastl = HOT ? kWave - 1 : (vmask ? 63 - __builtin_clzll(vmask) : 0);
  const int mn = __builtin_amdgcn_readlane(cmin, 0);
  const int mx = __builtin_amdgcn_readlane(cmax, lastl);
  const long long span = (long long)mx - (long long)mn + 2ll * margin;     // <= capw for the LDS path
  const bool usable = !__any(wild) && vmask != 0 && span <= capw && span > 0;
  const long long lo = anchor + mn - margin;                               // signal index of tile[0]
  const int nspan = usable ? (int)span : 0;
  // HOT (full wave, unit strides) and the span, rounded up to whole 64-sample rows, inside the signal: the rows go from
  // HBM straight into LDS (global_load_lds_dword: LDS address = M0 base + instruction offset + 4 lane, the same offset
  // advances the global address) -- one address per lane, no VGPR round trip, no address or bounds arithmetic per row
  bool dma = false;
  if constexpr (HOT && !(PAR_SINC_EXP & 16)) {
    const int nwords = nspan * LS;                  // float words of the span as it lies in memory (interleaved: 2 per sample)
    dma = lo >= 0 && (lo * LS + (long long)((nwords + kWave - 1) & ~(kWave - 1))) <= (long long)len_in * LS;
    if (dma) {
      const float* gp = sig + lo * LS + l;
      // the instruction offset (13 bits, signed) advances the global and the LDS address alike: 16 rows per base
      if (!((PAR_SINC_EXP & 256) && NCH == 1)) static_for<(capw * NCH / kWave + 15) / 16>([&](auto bi) {
        constexpr int b = decltype(bi)::value;
        static_for<16>([&](auto qi) {
          constexpr int q = b * 16 + decltype(qi)::value;
          if (q < capw * NCH / kWave && q * kWave < nwords)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + b * 16 * kWave),
                                             (__attribute__((address_space(3))) void*)(tile + b * 16 * kWave), 4,
                                             (q - b * 16) * kWave * 4, 0);
        });
      });
      __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the rows are in LDS
    }
  }
  if (!dma) {
    for (int q = l; q < nspan; q += kWave) {
      const long long g = lo + q;
      const bool inside = g >= 0 && g < (long long)len_in;
      tile[q * LS] = (inside && !(PAR_SINC_EXP & 16)) ? sig[g * sig_stride] : 0.0f;
      if (NCH == 2) tile[q * LS + CHO] = (inside && !(PAR_SINC_EXP & 16)) ? sig1[g * sig_stride] : 0.0f;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the wave's own LDS writes before its LDS reads
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if constexpr (fused_is_farrow(NCH, NTC, NS) && !PAR_FARROW_BARRIER_EARLY) {
    if (farrow_wg) {                   // workgroup-uniform: the constant fragments of ALL four waves are in LDS behind this
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
  }
  PAR_PHASE_MARK(2);                 // span in LDS (waits for the signal loads)
  // 5. taps
  // leading-edge outputs (ind < NT) keep the reference's mis-aligned taps: float64 slow path.
  bool fastlane[kOut];
  bool anyfast = false;
  const long long edge64 = (long long)NT - anchor;    // ind >= NT  <=>  rel index >= edge
  const int edge = edge64 < -2000000000ll ? -2000000000 : (edge64 > 2000000000ll ? 2000000000 : (int)edge64);
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    fastlane[r] = valid[r] && usable && c[r] >= edge && !lowfc[r] && c[r] >= mn - 1 && c[r] <= mx + 1;
    c[r] = fastlane[r] ? c[r] - mn + margin : margin;      // LDS index of the window centre (idle lanes: harmless)
    anyfast = anyfast || fastlane[r];
  }
  // (output, channel) slots: channel ch of an output reads the tile `ch * capw` floats further on; shift, fc and
  // 1 - fc are the SAME values for both slots of an output, so the compiler evaluates their tap weights once
  int cs[NS];
  float ss[NS], eps[NS], res[NS];
#pragma unroll
  for (int sl = 0; sl < NS; ++sl) {
    cs[sl] = c[sl / NCH] * LS + (sl % NCH) * CHO;
    ss[sl] = s[sl / NCH];
    // lanes past the end of the file were placed from whatever the clamped record slot holds: they must not take part in
    // the wave-wide choice of tap path / seed form (results of the valid lanes would depend on stale memory)
    const bool vo = HOT || valid[sl / NCH];
    eps[sl] = vo ? ep[sl / NCH] : 0.0f;
    res[sl] = 0.0f;
  }
  bool taps_done = false;
  if constexpr (HOT && fused_is_farrow(NCH, NTC, NS) && !(PAR_SINC_EXP & 1)) {
    // unity path with every lane on the fast path and the span inside the first kFarSpanMax floats of the piece: the taps
    // n >= 5 come from the Farrow bank on the matrix cores (unity_far_mfma); it declines waves float16 does not suit
    bool allfast = true;
#pragma unroll
    for (int r = 0; r < kOut; ++r) allfast = allfast && fastlane[r];
    const int nlim = dma ? ((nspan * LS + kWave - 1) & ~(kWave - 1)) : nspan * LS;     // staged words
    // the bank of a row pair covers the centres p0 .. p0 + 127: a period that rounds to 1 in float32 may still be 1 + 6e-8,
    // and 128 of those can reach p0 + 128 -- such waves take the vector loops
    // (stereo: the pair of a channel is its two slots, slot 2 the later output; word distance 2 per sample)
    constexpr int kSpanMax = NCH == 1 ? kFarSpanMax : kFarSpanMax2;
    const bool in_bank = NCH == 1 ? (cs[1] - __builtin_amdgcn_readlane(cs[0], 0) <= 127 && cs[3] - __builtin_amdgcn_readlane(cs[2], 0) <= 127)
                                  : (cs[2] - __builtin_amdgcn_readlane(cs[0], 0) <= 127 * LS);
    if (farrow_wg && __all(unity) && __all(allfast) && __all(in_bank) && nlim <= kSpanMax && span_suits_f16(tile, nlim, l)) {
      float qs[NS], far[NS];
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) qs[sl] = ss[sl] * ss[sl];
      unity_far_mfma<NTC == 50 ? 50 : 32, NS, LS>(tile, nlim, tile + kSpanMax, far_consts, l, cs, ss, qs, far);
      taps_unity_near<NTC == 50 ? 50 : 32, NS, LS, true>(tile, cs, ss, qs, far, res);
      taps_done = true;
    }
  }
  if (!taps_done && !(PAR_SINC_EXP & 1) && __any(anyfast)) {
    float fcs[NS], dds[NS];            // formed here, not at placement: the fc == 1 paths never need them
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
      fcs[sl] = fast_rcp(1.0f + eps[sl]);
      dds[sl] = eps[sl] * fcs[sl];
    }
    run_taps<NTC, NS, LS, fused_is_farrow(NCH, NTC, NS)>(tile, cs, ss, fcs, dds, __all(unity), NT, tab, tmd, res);
  }
#if PAR_SINC_EXP & 128
  {                                   // the tap loops a second time (what does ONE more pass cost?)
    float res2[NS];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) ss[sl] += 1e-3f * res[sl];
    float fcs[NS], dds[NS];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
      fcs[sl] = fast_rcp(1.0f + eps[sl]);
      dds[sl] = eps[sl] * fcs[sl];
    }
    if (__any(anyfast)) run_taps<NTC, NS, LS, fused_is_farrow(NCH, NTC, NS)>(tile, cs, ss, fcs, dds, __all(unity), NT, tab, tmd, res2);
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) res[sl] += 1e-9f * res2[sl];
  }
#endif
  PAR_PHASE_MARK(4);                 // taps
#if PAR_SINC_EXP & 64
  if (g_sinc_phase && l == 0) {        // (timing builds: why a wave's rows went to the slow path)
    int nslow = 0;
    for (int r = 0; r < kOut; ++r) nslow += __popcll(__ballot(valid[r] && !fastlane[r]));
    g_sinc_phase[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + 6] =
        (unsigned)usable | ((unsigned)(__any(wild) != 0) << 1) | ((unsigned)(span > capw) << 2) | ((unsigned)(span <= 0) << 3) | ((unsigned)nslow << 8);
    g_sinc_phase[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + 7] = (unsigned)span;
  }
#endif
  // 6. stores
  const int64_t ostr = HOT ? (int64_t)LS : out_stride;
  float* const op0 = out + (jw + l) * ostr;
  float* const op1 = NCH == 2 ? out1 + (jw + l) * ostr : nullptr;
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    const bool slowr = valid[r] && !fastlane[r];
    PosDp e{0.0, 1.0};
    if (slowr) {
      const long long j = jw + l + (int64_t)r * kWave;
      long long i = hd.iT;
      while (i + 1 < fa.nseg && fa.seg_start[i + 1] <= j) ++i;
      e = place_exact(fa.speeds, fa.seg_start, fa.seg_off, lazy ? nullptr : fa.ck, i, j, len_out);
    }
    // the float64 taps of the row's slow outputs, by the whole wave (wave-uniform call)
    float2 two = make_float2(0.0f, 0.0f);
#if PAR_SLOW_WAVE
    if (__any(slowr)) two = sinc_slow_wave<NCH>(slowr, e.p, e.dp, sig, sig1, sig_stride, len_in, NT, l);
#else
    if (slowr) {
      if constexpr (NCH == 2) two = sinc_two_f64(e.p, e.dp, sig, sig1, sig_stride, len_in, NT);
      else two.x = sinc_one_f64(e.p, e.dp, sig, sig_stride, len_in, NT);
    }
#endif
    if (!valid[r]) continue;
    float vch[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) vch[ch] = slowr ? (ch ? two.y : two.x) : res[r * NCH + ch];
    if constexpr (LS == 2) {           // interleaved output: both channels of the output in one 8-byte store
      if (!((PAR_SINC_EXP & 32) && vch[0] != 12345.678f))
        *reinterpret_cast<float2*>(op0 + (int64_t)(r * kWave) * 2) = make_float2(vch[0], vch[NCH - 1]);
    } else {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        if ((PAR_SINC_EXP & 32) && vch[ch] != 12345.678f) continue;
        (ch ? op1 : op0)[(int64_t)(r * kWave) * ostr] = vch[ch];
      }
    }
  }
  PAR_PHASE_MARK(5);                 // stores issued
}

// FUSED form: there is no position array in HBM.  The plan leaves a 32-byte record per block of 8 outputs (BlockRec:
// the block's positions as a quadratic in u) and a header per tile (anchor, first and last window centre), both at
// addresses that follow from the output index alone.  A workgroup therefore issues ALL its loads up front -- the
// records of its outputs, the header, then the signal span the header names -- instead of walking
// tile map -> segment -> checkpoint -> positions -> span (five dependent HBM round trips per tile, which 6 waves per
// SIMD could not cover: measured, phases were additive).  Per output the placement is ~20 float32 / integer
// instructions; outputs within the reference's own rounding of a half-integer position, and blocks the record model
// does not cover, are redone through place_fast / place_exact, so every window centre rint(p) is the reference's.
// Every WAVE is on its own: it owns 64 kOut consecutive outputs of the tile, places them, stages just their input span
// into its own quarter of the workgroup's LDS and runs the tap loops -- no workgroup barrier anywhere (measured with
// the per-phase wave clock, tools/phase_clock.py: with one span per workgroup a wave spent 28 % of its life waiting at
// the barrier for the slowest of its three siblings).  The halo (2 NT + margin samples per 64 kOut outputs) is fetched
// by neighbouring waves too; they sit on the same CU, so the repeats are L1/L2 hits.
template <int NCH, int NTC, int NS>
__global__ __launch_bounds__(fused_waves(NCH, NTC, NS) * kWave, PAR_SINC_WAVES) void k_sinc_fused(int64_t len_out, const float* __restrict__ sig,
                                                                  const float* __restrict__ sig1, int64_t sig_stride,
                                                                  int64_t len_in, int NT, const float4* __restrict__ tab,
                                                                  TapModes tmd, float* __restrict__ out,
                                                                  float* __restrict__ out1, int64_t out_stride,
                                                                  FusedArgs fa) {
#if PAR_SINC_PRIO
  __builtin_amdgcn_s_setprio(PAR_SINC_PRIO);        // experiment: K_sinc's waves ahead of the plan's in the issue arbitration
#endif
  // NS (output, channel) slots per lane: a wave owns kWaveOut = 64 NS / NCH consecutive outputs and its own piece of LDS;
  // workgroups are 4 waves whatever NS is (the plan's tiles, 1024 outputs, hold a whole number of waves)
  constexpr int kOut = NS / NCH;                    // outputs per lane
  constexpr int kWaveOut = kWave * kOut;            // outputs per wave: 256 / 128 / 64
  constexpr int kWaves = fused_waves(NCH, NTC, NS);
  constexpr int capw = fused_capw(NS, NCH);         // floats of one channel's span a wave may stage
  static_assert(NCH == 1 || NCH == 2, "mono or stereo");
  static_assert(kSincTile % kWaveOut == 0, "tiles hold whole waves");
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  const int t = threadIdx.x;
  const int l = t & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  // the mono NT = 32 kernel keeps the constant fragments of the unity path's Farrow bank at the front of its LDS
  constexpr bool kFarrow = fused_is_farrow(NCH, NTC, NS);
  constexpr int kConstFloats = kFarrow ? far_const_bytes(NTC) / 4 : 0;
  // ... in workgroups whose tile may hold fc = 1 outputs (fused_wave fetches them behind its record loads)
  float* tile = lds_all + kConstFloats + wv * (capw * NCH);        // this wave's span: channel 0, then channel 1 `capw` floats on
  const int64_t jw = ((int64_t)blockIdx.x * kWaves + wv) * kWaveOut;   // the wave's outputs: jw + l + 64 r, r < kOut
  const int nrem = (int)(len_out - jw < (int64_t)kWaveOut ? (len_out - jw > 0 ? len_out - jw : 0) : kWaveOut);   // valid outputs of the wave
  // hot waves: full, and either mono on unit strides or an interleaved stereo file (8-byte aligned output pairs)
  const bool hot_layout = NCH == 1 ? (sig_stride == 1 && out_stride == 1)
                                   : (NTC > 0 && sig_stride == 2 && out_stride == 2 && sig1 == sig + 1 && out1 == out + 1 &&
                                      (reinterpret_cast<uintptr_t>(out) & 7) == 0);
  // (Persistent workgroups -- one constant fetch per workgroup instead of per tile -- were measured and rejected: the tile
  // loop keeps every kernel argument live, 83 SGPRs and 35 VGPRs spill, 5.16 -> 6.74 ms.)
  if (PAR_SINC_HOT && nrem == kWaveOut && hot_layout)
    fused_wave<NCH, NTC, NS, true>(len_out, sig, sig1, sig_stride, len_in, NT, tab, tmd, out, out1, out_stride, fa, tile, l, jw, nrem,
                                   lds_all);
  else
    fused_wave<NCH, NTC, NS, false>(len_out, sig, sig1, sig_stride, len_in, NT, tab, tmd, out, out1, out_stride, fa, tile, l, jw, nrem,
                                    lds_all);
}

}  // namespace par
