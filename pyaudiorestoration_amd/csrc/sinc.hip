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
//    collapses to (-1)^(n+1) sin(pi*shift) and is factored out of the sum.  Taps +n and -n share one
//    denominator R_n = (win_n/pi)/(n^2 - shift^2): a v_rcp_f32 (quarter rate on gfx950) only for n = 1..4, further
//    out a short polynomial in shift^2 (2-term series, linear minimax fit, constant: see TapModes) against a
//    wave-uniform table that arrives by scalar loads and stays in SGPRs.  The kernel is VALU-bound, not HBM-bound.
//  * everything the float32 fast path is not built for runs lane-wise in float64, line by line like the
//    reference (sinc_one_f64): the bug-compatible leading edge, tiles too wide for LDS, positions beyond the
//    int32 offset range, and fc < 1/8 (long averages whose output is small against the signal).
//  * FUSED form (k_sinc_fused): no position array in HBM and no float64 cumsum in the kernel: every output is placed
//    in closed form from the plan's per-segment record and the cumsum checkpoint below it (place_fast), accurate to
//    ~1e-10 samples; outputs whose position lies within the reference's own rounding of a half-integer are redone with
//    the reference's sequential float64 adds (place_exact), so every window centre rint(p) is the reference's.
//  * positions stay float64 end to end (a 345.6 M-sample index does not fit float32); only the
//    sub-sample shift in [-0.5, 0.5] and fc drop to float32.
#include "par_common.h"
#include "pos_plan.h"
#include "sinc_taps_gen.h"
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

namespace par {

#if PAR_SINC_EXP & 64
__device__ unsigned int* g_sinc_phase;           // [wave][8] cycle counts, one row per wave of the launch (no atomics)
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

constexpr int kSincBlock = 256;
constexpr int kSincR = 4;                         // outputs per thread
constexpr int kSincTile = kSincBlock * kSincR;    // outputs per workgroup
constexpr int kSincCap = 4096;                    // LDS floats for the staged input span (16 KiB; speeds up to ~3.7)

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
  for (long long k = 0; k < L; ++k) {
    double x = ((double)(k - NT) - shift) * fc;
    double y = M_PI * (x == 0.0 ? 1e-20 : x);           // np.sinc
    double si = sin(y) / y * fc;
    float win = (float)(0.5 + 0.5 * cos(M_PI * (double)(k - NT) / (double)NT));   // np.hanning(2NT+1)[k] as f32
    acc += (double)sig[(lower + k) * sig_stride] * si * (double)win;
  }
  return (float)acc;
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
    g.Up[r] = -sphi;                                  // U_0 = sin(-phi)
    g.Vp[r] = sphi;                                   // V_0 = sin(+phi)
    g.U[r] = fmaf(sth, cphi, -cth * sphi);            // U_1 = sin(theta - phi)
    g.V[r] = fmaf(sth, cphi, cth * sphi);             // V_1 = sin(theta + phi)
    g.c2[r] = 2.0f * cth;
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
template <int NT, int R, int LS = 1>
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
    base[r] = (LS == 2 && r == 1) ? base[0] + 1 : tl + c[r] - NT * LS;    // tap +n at [(NT + n) LS], tap -n at [(NT - n) LS]: immediates
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

template <int NT, int R, int LS = 1>
__device__ __forceinline__ void taps_general_ct(const float* __restrict__ tile, const int (&c)[R], const float (&s)[R],
                                                const float (&fc)[R], const float (&dd)[R], const int nt_rt,
                                                float (&res)[R]) {
  static_assert(LS == 1 || (LS == 2 && R == 2), "interleaved spans: the two channel slots of one output");
  using T = TapTab<NT>;
  float q[R], U[R], Up[R], V[R], Vp[R], c2[R], M0[R], M1[R], M2[R], P0[R], P1[R], P2[R], centre[R];
  lds_cfloat* base[R];
  lds_cfloat* tl = (lds_cfloat*)tile;
  constexpr float b0 = T::B[0];
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
    Up[r] = -sphi;                                    // U_0 = sin(-phi)
    Vp[r] = sphi;                                     // V_0 = sin(+phi)
    U[r] = fmaf(sth, cphi, -cth * sphi);              // U_1 = sin(theta - phi)
    V[r] = fmaf(sth, cphi, cth * sphi);               // V_1 = sin(theta + phi)
    c2[r] = 2.0f * cth;
    q[r] = s[r] * s[r];
    centre[r] = tile[c[r]] * (Up[r] * fast_rcp(s[r] * b0));
    M0[r] = M1[r] = M2[r] = P0[r] = P1[r] = P2[r] = 0.0f;
    base[r] = (LS == 2 && r == 1) ? base[0] + 1 : tl + c[r] - NT * LS;
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

// ---- output placement -------------------------------------------------------------------------------
// What the tap loops need per output: the integer window centre (relative to a block-uniform EVEN anchor, so that
// indices are int32 and round-half-even ties equal rint(p)), the sub-sample shift, fc and 1 - fc.
struct FusedArgs {
  const double* speeds;
  const int64_t* seg_start;
  const double* seg_off;
  const double* ck;
  const int64_t* tile_seg;
  const SegFast* seg_fast;
  const TileHdr* hdr;
  const BlockRec* rec;
  const BlockRec2* rec2;
  int64_t nseg;
};

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
__device__ __noinline__ PosDp place_exact(const double* __restrict__ speeds, const int64_t* __restrict__ seg_start,
                                          const double* __restrict__ seg_off, const double* __restrict__ ckp, long long i,
                                          long long j, long long len_out) {
#pragma clang fp contract(off)
  const long long start = seg_start[i], n = seg_start[i + 1] - start;
  const long long k = j - start, b = k / kCk;
  const Ramp r = make_ramp(speeds[i], speeds[i + 1], n);
  const double off = seg_off[i];
  double c = b ? ckp[ck_slot0(start, i) + b] : 0.0;
  double cprev = c;
  for (long long v = b * kCk; v <= k; ++v) {
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
                                           bool& exact) {
  const long long start = fa.seg_start[i];
  const SegFast sf = fa.seg_fast[i];
  const double s0 = fa.speeds[i];
  const int k = (int)(j - start);
  const int u = k & (kCk - 1);                           // steps between the checkpoint and this output
  const long long b = k >> 3;
  static_assert(kCk == 8, "k >> 3");
  const double ckv = b ? fa.ck[ck_slot0(start, i) + b] : 0.0;
  const long long dA = sf.A - anchor;
  wild = !(dA > -2000000000ll && dA < 2000000000ll);
  const double base = ((double)(int)dA + sf.foff) + ckv;
  const double tm = (double)k - 0.5 * (double)u;         // midpoint of steps k-u .. k
  const double bsm = fma(sf.step, tm, s0);
  double r = __builtin_amdgcn_rcp(bsm);                  // ~2^-26; one Newton step -> ~1e-15
  r = fma(r, fma(-bsm, r, 1.0), r);
  const double z = r * sf.step;
  const double cw = (double)(u * (u + 2)) * (1.0 / 12.0);   // (w^2 - 1)/12 with w = u + 1
  const double prel = fma(r * (double)(u + 1), fma(z * z, cw, 1.0), base);
  const double rf = rint(prel);
  const double shd = prel - rf;
  if (!(fabs(prel) < 1.0e9)) wild = true;
  c = wild ? 0 : (int)rf;
  const float sh = (float)shd;
  s = (sh == 0.0f) ? 1e-20f : sh;
  // rint(p) is the reference's when p is farther from a half-integer than the reference's own roundings: half an ulp of
  // p (tol) plus, per add behind the checkpoint, half an ulp of the running sum (matters for segments of > 10^7 outputs)
  exact = !sf.fast || !(fabs(fabs(shd) - 0.5) > fma(ckv, 9.6e-16, tol));
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
template <int NTC, int NS, int LS = 1>
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
      if constexpr (NTC > 0) taps_unity_ct<NTC, 2, LS>(tile, ca, sa, NT, ra);
    } else {
      const float fa_[2] = {fcs[2 * h], fcs[2 * h + 1]};
      const float da[2] = {dds[2 * h], dds[2 * h + 1]};
      if constexpr (NTC > 0) taps_general_ct<NTC, 2, LS>(tile, ca, sa, fa_, da, NT, ra);
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
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      float v = res[r * NCH + ch];
      if (!fastlane[r]) v = sinc_one_f64(pj, dpj, ch ? sig1 : sig, sig_stride, len_in, NT);
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
                                           float* __restrict__ tile, const int l, const int64_t jw, const int nrem) {
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
#ifdef PAR_SINC_PRIO
  __builtin_amdgcn_s_setprio(PAR_SINC_PRIO);     // a young wave gets its loads out ahead of its neighbours' tap loops
#endif
  // 1. records of this lane's outputs: block (jw >> 3) + (l >> 3) + 8 r, u = l & 7 for every r.  Unconditional 16-byte
  // loads off a wave-uniform base (indices past the file's last block are clamped to it)
  const int gmax = nrem > 0 ? (nrem - 1) >> 3 : 0;
  const uint4* rp = reinterpret_cast<const uint4*>(fa.rec + (jw >> 3) * kRecStride);
  uint4 ra[kOut];
#if PAR_REC_BOTH
  uint4 rb[kOut];                   // the block's second piece (garbage unless a segment starts inside the block)
  const uint4* rp2 = PAR_REC_INTERLEAVE == 1 ? rp + 1 : reinterpret_cast<const uint4*>(fa.rec2 + (jw >> 3));
#endif
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    const int gr = (l >> 3) + 8 * r;
    const int gi = (HOT ? gr : (gr < gmax ? gr : gmax)) * kRecStride;
    ra[r] = rp[gi];
#if PAR_REC_BOTH
    rb[r] = rp2[gi];
#endif
  }
  // 2. tile header: the anchor all window centres are relative to
  const TileHdr hd = fa.hdr[T];
  const long long anchor = hd.anchor;
  PAR_PHASE_MARK(0);                 // issue of the record loads, arrival of the header
  // 3. placement
  const double tol = (fabs((double)anchor) + 4.2e6) * 1.2e-16 + 2.0e-10;     // the reference's own rounding of p (half an ulp)
  const float tolf = (float)tol + 1.5e-7f;                                    // + float32 evaluation of the block quadratic
  const unsigned alo = (unsigned)(unsigned long long)anchor;
  const int u = l & 7;
  const float uf = (float)u, u2f = uf * uf, tw1 = 2.0f * uf + 1.0f, tw0 = 2.0f * uf - 1.0f;
  int c[kOut];
  float s[kOut], fc[kOut], dd[kOut];
  bool valid[kOut], lowfc[kOut], redo[kOut], slow[kOut], second[kOut];
  bool unity = true, wild = (hd.flags & 1) != 0, anyredo = false, anysecond = false;
  unsigned I[kOut];
  float F[kOut], e1[kOut];
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    valid[r] = HOT || l + kWave * r < nrem;
    second[r] = (unsigned)u > (ra[r].w & 7u);            // u >= ustar, ustar - 1 in bits 0-2
    anysecond = anysecond || second[r];
    I[r] = ra[r].x;
    F[r] = __uint_as_float(ra[r].y);
    e1[r] = __uint_as_float(ra[r].z);
  }
#if PAR_REC_BOTH
#pragma unroll
  for (int r = 0; r < kOut; ++r) {   // a segment starts inside the lane's block at or before u: its second piece
    I[r] = second[r] ? rb[r].x : I[r];
    F[r] = second[r] ? __uint_as_float(rb[r].y) : F[r];
    e1[r] = second[r] ? __uint_as_float(rb[r].z) : e1[r];
  }
#else
  if (__any(anysecond)) {            // a segment starts inside some lane's block: those lanes take its second piece
    const uint4* rp2 = reinterpret_cast<const uint4*>(fa.rec2 + (jw >> 3));
#pragma unroll
    for (int r = 0; r < kOut; ++r) {
      if (second[r]) {
        const int gr = (l >> 3) + 8 * r;
        const uint4 q = rp2[HOT ? gr : (gr < gmax ? gr : gmax)];
        I[r] = q.x;
        F[r] = __uint_as_float(q.y);
        e1[r] = __uint_as_float(q.z);
      }
    }
  }
#endif
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    const unsigned m = ra[r].w;
    const float e2 = __uint_as_float(m & ~kRecFlagBits);
    const unsigned ustar1 = m & 7u;                      // ustar - 1
    // the segment's last output (period = previous increment): u = ustar - 1 in a boundary block, else u = 7 with end0/end1
    const bool plateau = second[r] ? (u == 7 && (m & 16u)) : (ustar1 < 7u ? (unsigned)u == ustar1 : (u == 7 && (m & 8u)));
    slow[r] = valid[r] && ((second[r] ? (m & 64u) : (m & 32u)) != 0u);
    const float frac = fmaf(u2f, e2, fmaf(uf, e1[r], F[r]));
    const float ri = rintf(frac);
    const float sh = frac - ri;
    c[r] = (int)(I[r] - alo) + u + (int)ri;
    s[r] = (sh == 0.0f) ? 1e-20f : sh;              // np.sinc's own 0 -> 1e-20 substitution
    const float e = fmaf(e2, plateau ? tw0 : tw1, e1[r]);                    // period to the next position, minus 1
    const bool one = !(e > 0.0f);
    const float inv = fast_rcp(1.0f + e);
    fc[r] = one ? 1.0f : inv;
    dd[r] = one ? 0.0f : e * inv;
    lowfc[r] = e > 7.0f;
    redo[r] = slow[r] || (valid[r] && !(fabsf(fabsf(sh) - 0.5f) > tolf));
    anyredo = anyredo || redo[r];
    if (PAR_SINC_EXP & 2) {
      c[r] = (int)(jw + l + kWave * r - anchor) + 64;
      s[r] = 0.3f - 1e-4f * (float)(t & 63);
      fc[r] = (PAR_SINC_EXP & 4) ? 0.995f : 1.0f;
      dd[r] = (PAR_SINC_EXP & 4) ? 0.005f : 0.0f;
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
        if (slow[r]) place_fast(fa, i, j, len_out, anchor, tol, c[r], s[r], fc[r], dd[r], lowfc[r], w, ex);
        if (ex) {
          const PosDp e = place_exact(fa.speeds, fa.seg_start, fa.seg_off, fa.ck, i, j, len_out);
          place_from_pos(e.p, e.dp, (double)anchor, c[r], s[r], fc[r], dd[r], lowfc[r], w);
        }
        wild = wild || w;
      }
    }
  }
  int cmin = INT_MAX, cmax = INT_MIN;
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    unity = unity && (fc[r] == 1.0f || !valid[r]);
    if (valid[r]) {
      cmin = c[r] < cmin ? c[r] : cmin;
      cmax = c[r] > cmax ? c[r] : cmax;
    }
  }
  PAR_PHASE_MARK(1);                 // placement (waits for the records)
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
      static_for<(capw * NCH / kWave + 15) / 16>([&](auto bi) {
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
  float ss[NS], fcs[NS], dds[NS], res[NS];
#pragma unroll
  for (int sl = 0; sl < NS; ++sl) {
    cs[sl] = c[sl / NCH] * LS + (sl % NCH) * CHO;
    ss[sl] = s[sl / NCH];
    fcs[sl] = fc[sl / NCH];
    dds[sl] = dd[sl / NCH];
    res[sl] = 0.0f;
  }
#ifdef PAR_SINC_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  if (!(PAR_SINC_EXP & 1) && __any(anyfast)) run_taps<NTC, NS, LS>(tile, cs, ss, fcs, dds, __all(unity), NT, tab, tmd, res);
#ifdef PAR_SINC_PRIO
  __builtin_amdgcn_s_setprio(PAR_SINC_PRIO_OUT);
#endif
#if PAR_SINC_EXP & 128
  {                                   // the tap loops a second time (what does ONE more pass cost?)
    float res2[NS];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) ss[sl] += 1e-3f * res[sl];
    if (__any(anyfast)) run_taps<NTC, NS, LS>(tile, cs, ss, fcs, dds, __all(unity), NT, tab, tmd, res2);
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) res[sl] += 1e-9f * res2[sl];
  }
#endif
  PAR_PHASE_MARK(4);                 // taps
  // 6. stores
  const int64_t ostr = HOT ? (int64_t)LS : out_stride;
  float* const op0 = out + (jw + l) * ostr;
  float* const op1 = NCH == 2 ? out1 + (jw + l) * ostr : nullptr;
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    if (!valid[r]) continue;
    PosDp e{0.0, 1.0};
    if (!fastlane[r]) {
      const long long j = jw + l + (int64_t)r * kWave;
      long long i = hd.iT;
      while (i + 1 < fa.nseg && fa.seg_start[i + 1] <= j) ++i;
      e = place_exact(fa.speeds, fa.seg_start, fa.seg_off, fa.ck, i, j, len_out);
    }
    float vch[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      vch[ch] = res[r * NCH + ch];
      if (!fastlane[r]) vch[ch] = sinc_one_f64(e.p, e.dp, ch ? sig1 : sig, sig_stride, len_in, NT);
    }
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
__global__ __launch_bounds__(kSincBlock, PAR_SINC_WAVES) void k_sinc_fused(int64_t len_out, const float* __restrict__ sig,
                                                                  const float* __restrict__ sig1, int64_t sig_stride,
                                                                  int64_t len_in, int NT, const float4* __restrict__ tab,
                                                                  TapModes tmd, float* __restrict__ out,
                                                                  float* __restrict__ out1, int64_t out_stride,
                                                                  FusedArgs fa) {
  // NS (output, channel) slots per lane: a wave owns kWaveOut = 64 NS / NCH consecutive outputs and its own piece of LDS;
  // workgroups are 4 waves whatever NS is (the plan's tiles, 1024 outputs, hold a whole number of waves)
  constexpr int kOut = NS / NCH;                    // outputs per lane
  constexpr int kWaveOut = kWave * kOut;            // outputs per wave: 256 / 128 / 64
  constexpr int kWaves = kSincBlock / kWave;
  constexpr int capw = fused_capw(NS, NCH);         // floats of one channel's span a wave may stage
  static_assert(NCH == 1 || NCH == 2, "mono or stereo");
  static_assert(kSincTile % kWaveOut == 0, "tiles hold whole waves");
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  const int t = threadIdx.x;
  const int l = t & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  float* tile = lds_all + wv * (capw * NCH);        // this wave's span: channel 0, then channel 1 `capw` floats on
  const int64_t jw = ((int64_t)blockIdx.x * kWaves + wv) * kWaveOut;   // the wave's outputs: jw + l + 64 r, r < kOut
  const int nrem = (int)(len_out - jw < (int64_t)kWaveOut ? (len_out - jw > 0 ? len_out - jw : 0) : kWaveOut);   // valid outputs of the wave
  // hot waves: full, and either mono on unit strides or an interleaved stereo file (8-byte aligned output pairs)
  const bool hot_layout = NCH == 1 ? (sig_stride == 1 && out_stride == 1)
                                   : (NTC > 0 && sig_stride == 2 && out_stride == 2 && sig1 == sig + 1 && out1 == out + 1 &&
                                      (reinterpret_cast<uintptr_t>(out) & 7) == 0);
#ifdef PAR_SINC_PRIO_ALL
  __builtin_amdgcn_s_setprio(PAR_SINC_PRIO_ALL);   // experiment: the whole kernel above the plan kernels that share its SIMDs
#endif
  if (PAR_SINC_HOT && nrem == kWaveOut && hot_layout)
    fused_wave<NCH, NTC, NS, true>(len_out, sig, sig1, sig_stride, len_in, NT, tab, tmd, out, out1, out_stride, fa, tile, l, jw, nrem);
  else
    fused_wave<NCH, NTC, NS, false>(len_out, sig, sig1, sig_stride, len_in, NT, tab, tmd, out, out1, out_stride, fa, tile, l, jw, nrem);
}

// ---- host side: per-(device, NT) tap tables -------------------------------------------------------
struct SincTable {
  float4* ab = nullptr;      // per-tap rows, see tap_R
  TapModes modes{kPoly2From, kPoly2From};
};
static std::mutex g_tab_mu;
static std::map<std::pair<int, int>, SincTable> g_tabs;

// Which polynomial form each tap pair uses (see TapModes): worst-case error of a pair = approximation error of R_n
// over q in [0, 1/4] times the largest the bracket n(G+H) + s(G-H) can get for unit-peak input, 2n + 1.  The forms
// start at the first chunk boundary from which the errors of ALL pairs further out sum to less than the budget.
static void choose_tap_modes(int NT, TapModes* tm, std::vector<double>* lin_a, std::vector<double>* lin_b,
                             std::vector<double>* cst) {
  std::vector<double> e0(NT + kChunk, 0.0), e1(NT + kChunk, 0.0);
  lin_a->assign(NT + kChunk, 0.0);
  lin_b->assign(NT + kChunk, 0.0);
  cst->assign(NT + kChunk, 0.0);
  for (int n = 1; n < NT; ++n) {
    const double win = (double)(float)(0.5 + 0.5 * cos(M_PI * (double)n / (double)NT));
    const double K = win / M_PI, n2 = (double)n * (double)n;
    if (K <= 0.0) continue;
    const double f0 = K / n2, f1 = K / (n2 - 0.25);
    (*cst)[n] = 0.5 * (f0 + f1);                          // constant minimax: mid-range
    e0[n] = 0.5 * (f1 - f0) * (2.0 * n + 1.0);
    // linear minimax of the convex f(q) = K/(n^2 - q) on [0, 1/4]: chord slope, intercept halfway between the chord
    // and the parallel tangent (touching at q* with f'(q*) = slope)
    const double slope = (f1 - f0) / 0.25;
    const double qs = n2 - sqrt(K / slope);
    const double gap = (f0 + slope * qs) - K / (n2 - qs);
    (*lin_b)[n] = slope;
    (*lin_a)[n] = f0 - 0.5 * gap;
    e1[n] = 0.5 * gap * (2.0 * n + 1.0);
  }
  auto first_from = [&](const std::vector<double>& e, double budget) {
    int from = ((NT + kChunk) / kChunk) * kChunk + 1;     // beyond every chunk: form unused
    double tail = 0.0;
    for (int n0 = ((NT - 1) / kChunk) * kChunk + 1; n0 >= kPoly2From; n0 -= kChunk) {
      for (int n = n0; n < n0 + kChunk && n < NT; ++n) tail += e[n];
      if (tail > budget) break;
      from = n0;
    }
    return from;
  };
  tm->p1_from = first_from(e1, 3.0e-7);
  tm->p0_from = first_from(e0, 1.5e-6);
  if (tm->p0_from < tm->p1_from) tm->p0_from = tm->p1_from;
}

static int get_sinc_table(int device, int NT, SincTable* out) {
  std::lock_guard<std::mutex> lk(g_tab_mu);
  auto key = std::make_pair(device, NT);
  auto it = g_tabs.find(key);
  if (it != g_tabs.end()) {
    *out = it->second;
    return PAR_OK;
  }
  SincTable t;
  std::vector<double> lin_a, lin_b, cst;
  choose_tap_modes(NT, &t.modes, &lin_a, &lin_b, &cst);
  // a2[n] = pi*n^2/win, b[n] = -pi/win with win = float32(np.hanning(2NT+1)[NT+n]), n = 0..NT-1
  std::vector<float4> ab(NT + kChunk);
  for (int n = 0; n < NT + kChunk; ++n) {
    const double win = n < NT ? (double)(float)(0.5 + 0.5 * cos(M_PI * (double)n / (double)NT)) : 0.0;
    const double n2 = (double)n * (double)n;
    if (n < kPoly2From) {            // reciprocal rows; padded taps: rcp(q*b + inf) == 0
      ab[n] = n < NT ? make_float4((float)(M_PI * n2 / win), (float)(-M_PI / win), (float)n, 0.0f)
                     : make_float4(INFINITY, (float)(-M_PI), (float)n, 0.0f);
    } else if (n < t.modes.p1_from) {   // 2-term series rows; padded taps: all-zero coefficients
      const double A = win / (M_PI * n2);
      ab[n] = make_float4((float)A, (float)(A / n2), (float)n, (float)(A / (n2 * n2)));
    } else if (n < t.modes.p0_from) {   // linear minimax rows
      ab[n] = n < NT ? make_float4((float)lin_a[n], (float)lin_b[n], (float)n, 0.0f) : make_float4(0.0f, 0.0f, (float)n, 0.0f);
    } else {                            // constant rows: (A, n A)
      ab[n] = n < NT ? make_float4((float)cst[n], (float)(cst[n] * n), (float)n, 0.0f) : make_float4(0.0f, 0.0f, (float)n, 0.0f);
    }
  }
  PAR_HIP_CHECK(hipMalloc(&t.ab, ab.size() * sizeof(float4)));
  PAR_HIP_CHECK(hipMemcpy(t.ab, ab.data(), ab.size() * sizeof(float4), hipMemcpyHostToDevice));
  g_tabs[key] = t;
  *out = t;
  return PAR_OK;
}

static_assert(kSincTile == kSincTileOutputs, "chunk alignment constant out of sync");

// outputs [j_begin, j_begin+count) of a len_out-long position array (j_begin must be tile aligned)
int launch_sinc(int device, const double* pos, int64_t len_out, int64_t j_begin, int64_t count, const float* sig,
                int64_t sig_stride, int64_t len_in, int NT, float* out, int64_t out_stride, hipStream_t s) {
  if (count <= 0) return PAR_OK;
  SincTable tab;
  int rc = get_sinc_table(device, NT, &tab);
  if (rc != PAR_OK) return rc;
  const int64_t blocks = ceil_div(count, kSincTile);
#define PAR_LAUNCH_POS(NTC)                                                                                              \
  hipLaunchKernelGGL((k_sinc_pos<NTC>), dim3((unsigned)blocks), dim3(kSincBlock), kSincCap * sizeof(float), s, pos, len_out, \
                     sig, sig_stride, len_in, NT, tab.ab, tab.modes, out, out_stride, j_begin, j_begin + count)
  if (NT == 32) PAR_LAUNCH_POS(32);
  else if (NT == 50) PAR_LAUNCH_POS(50);
  else PAR_LAUNCH_POS(0);
#undef PAR_LAUNCH_POS
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

// whole output range, every output placed in-kernel from the plan + checkpoints (no position array)
// sig1 / out1 != nullptr: second channel of the same file (same strides), resampled in the same launch
int launch_sinc_fused(int device, const double* speeds, int64_t m, const void* work, const void* aux, int64_t max_out,
                      int64_t len_out, const float* sig, const float* sig1, int64_t sig_stride, int64_t len_in, int NT,
                      float* out, float* out1, int64_t out_stride, hipStream_t s) {
  SincTable tab;
  int rc = get_sinc_table(device, NT, &tab);
  if (rc != PAR_OK) return rc;
  PlanView pv = plan_view(const_cast<void*>(work), m);
  // No header read-back here (it would cost a stream sync per channel): the caller vouches, through the
  // fused_ok flag of par_speed_to_pos_plan_fused, that aux holds this plan's checkpoints for max_out.
  const FusedAux av = fused_aux_view(const_cast<void*>(aux), max_out, m);
  FusedArgs fa;
  fa.speeds = speeds;
  fa.seg_start = pv.seg_start;
  fa.seg_off = pv.seg_off;
  fa.ck = av.ck;
  fa.tile_seg = av.tile_seg;
  fa.seg_fast = av.seg_fast;
  fa.hdr = av.hdr;
  fa.rec = av.rec;
  fa.rec2 = av.rec2;
  fa.nseg = m - 1;
  // slots per lane (PAR_SINC_SLOTS_*: experiment knobs; 4 = 256 mono / 128 stereo outputs per wave)
#ifndef PAR_SINC_SLOTS_MONO
#define PAR_SINC_SLOTS_MONO 4
#endif
#ifndef PAR_SINC_SLOTS_STEREO
#define PAR_SINC_SLOTS_STEREO 4
#endif
  // experiment knob: extra (unused) LDS per workgroup lowers K_sinc's occupancy and leaves wave slots to a concurrent plan
  static const size_t lds_pad = getenv("PAR_SINC_LDS_PAD") ? (size_t)atoi(getenv("PAR_SINC_LDS_PAD")) : 0;
#define PAR_LAUNCH_FUSED(NCH, NTC, NS)                                                                                     \
  hipLaunchKernelGGL((k_sinc_fused<NCH, NTC, NS>),                                                                         \
                     dim3((unsigned)ceil_div(len_out, (int64_t)(kSincBlock / kWave) * kWave * (NS) / (NCH))),             \
                     dim3(kSincBlock), (kSincBlock / kWave) * fused_capw(NS, NCH) * (NCH) * sizeof(float) + lds_pad, s,    \
                     len_out, sig, sig1, sig_stride, len_in, NT, tab.ab, tab.modes, out, out1, out_stride, fa)
  if (sig1 && out1) {
    if (NT == 32) PAR_LAUNCH_FUSED(2, 32, PAR_SINC_SLOTS_STEREO);
    else if (NT == 50) PAR_LAUNCH_FUSED(2, 50, PAR_SINC_SLOTS_STEREO);
    else PAR_LAUNCH_FUSED(2, 0, PAR_SINC_SLOTS_STEREO);
  } else {
    sig1 = nullptr;
    out1 = nullptr;
    if (NT == 32) PAR_LAUNCH_FUSED(1, 32, PAR_SINC_SLOTS_MONO);
    else if (NT == 50) PAR_LAUNCH_FUSED(1, 50, PAR_SINC_SLOTS_MONO);
    else PAR_LAUNCH_FUSED(1, 0, PAR_SINC_SLOTS_MONO);
  }
#undef PAR_LAUNCH_FUSED
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

}  // namespace par

extern "C" {

#if PAR_SINC_EXP & 64
// experiment builds (tools/phase_clock.py): set the per-wave phase buffer ([waves][8] uint32)
int par_debug_sinc_phase_buffer(unsigned int* dev_buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(par::g_sinc_phase), &dev_buf, sizeof(dev_buf)) == hipSuccess ? 0 : 1;
}
#endif

int par_sinc_resample_f32(int device, const double* pos, int64_t len_out, const float* sig, int64_t sig_stride,
                          int64_t len_in, int NT, float* out, int64_t out_stride, void* stream) {
  using namespace par;
  PAR_REQUIRE(pos && sig && out, PAR_ERR_ARG, "par_sinc_resample_f32: null pointer");
  PAR_REQUIRE(len_out >= 2, PAR_ERR_ARG, "par_sinc_resample_f32: len_out=%lld < 2 (reference raises UnboundLocalError)",
              (long long)len_out);
  PAR_REQUIRE(NT >= 1 && NT <= 512, PAR_ERR_ARG, "par_sinc_resample_f32: NT=%d outside [1,512]", NT);
  PAR_REQUIRE(len_in >= 1 && sig_stride >= 1 && out_stride >= 1, PAR_ERR_ARG, "par_sinc_resample_f32: bad sizes");
  PAR_HIP_CHECK(hipSetDevice(device));
  return launch_sinc(device, pos, len_out, 0, len_out, sig, sig_stride, len_in, NT, out, out_stride, as_stream(stream));
}

}  // extern "C"
