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
//    denominator R_n = (win_n/pi)/(n^2 - shift^2): a v_rcp_f32 (quarter rate on gfx950) only for n = 1..4, from
//    n = 5 on a 2-term and from n = 13 on a 1-term series in shift^2 against a wave-uniform table that arrives by
//    scalar loads and stays in SGPRs.  The kernel is VALU-bound (~80 % issue utilisation), not HBM-bound.
//  * everything the float32 fast path is not built for runs lane-wise in float64, line by line like the
//    reference (sinc_one_f64): the bug-compatible leading edge, tiles too wide for LDS, positions beyond the
//    int32 offset range, and fc < 1/8 (long averages whose output is small against the signal).
//  * FUSED form (k_sinc<true>): no position array in HBM; the tile's float64 positions are regenerated in LDS
//    from the plan's cumsum checkpoints with the same sequential adds numpy's cumsum does (bit-identical).
//  * positions stay float64 end to end (a 345.6 M-sample index does not fit float32); only the
//    sub-sample shift in [-0.5, 0.5] and fc drop to float32.
#include "par_common.h"
#include "pos_plan.h"
#include <limits.h>
#include <math.h>
#include <map>
#include <vector>

namespace par {

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

// How R_n(q) = (win_n/pi)/(n^2 - q), q = shift^2 <= 1/4, is evaluated for the taps of one chunk.  Only the
// four innermost pairs pay for a v_rcp_f32 (quarter rate); from n = 5 on q/n^2 <= 0.01 and the geometric
// series A_n*(1 + q/n^2 + q^2/n^4) is exact to 1e-6 relative (weight <= 0.06: 6e-8 absolute), from
// n = 13 on one term suffices (2e-6 * weight 0.017).  The table rows change meaning accordingly.
enum { kRcp = 0, kPoly2 = 1, kPoly1 = 2 };
constexpr int kPoly2From = 5;     // rows n >= 5 : (A_n, B_n, n, C_n) with A = win/(pi n^2), B = A/n^2, C = B/n^2
constexpr int kPoly1From = 13;    // rows n >= 13: only (A_n, B_n) are used
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
      const float Rn = tap_R<MODE>(q[r], ab[k]);
      const float DR = D * Rn;
      if (k & 1) {                           // n0 is odd, so odd k is an even n: +
        e[r] = fmaf(E, Rn, e[r]);
        d[r] = fmaf(DR, fn, d[r]);
      } else {                               // odd n: -
        e[r] = fmaf(-E, Rn, e[r]);
        d[r] = fmaf(-DR, fn, d[r]);
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
                                           int NT, const float4* __restrict__ tab, float (&res)[R]) {
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
    for (; n0 + kChunk <= NT && n0 < kPoly1From; n0 += kChunk) unity_chunk<kPoly2, false, R>(tp, tm, q, e, d, tab, n0, NT);
#pragma unroll 1
    for (; n0 + kChunk <= NT; n0 += kChunk) unity_chunk<kPoly1, false, R>(tp, tm, q, e, d, tab, n0, NT);
  }
  if (n0 == 1) unity_chunk<kRcp, true, R>(tp, tm, q, e, d, tab, n0, NT);
  else if (n0 < kPoly1From) unity_chunk<kPoly2, true, R>(tp, tm, q, e, d, tab, n0, NT);
  else unity_chunk<kPoly1, true, R>(tp, tm, q, e, d, tab, n0, NT);
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
      const float Rn = tap_R<MODE>(g.q[r], ab[k]);
      g.accM[r] = fmaf(G - H, Rn, g.accM[r]);
      g.accP[r] = fmaf((G + H) * Rn, fn, g.accP[r]);
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
                                             const float4* __restrict__ tab, float (&res)[R]) {
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
    for (; n0 + kChunk <= NT && n0 < kPoly1From; n0 += kChunk) general_chunk<kPoly2, false, R>(tp, tm, g, tab, n0, NT);
#pragma unroll 1
    for (; n0 + kChunk <= NT; n0 += kChunk) general_chunk<kPoly1, false, R>(tp, tm, g, tab, n0, NT);
  }
  if (n0 == 1) general_chunk<kRcp, true, R>(tp, tm, g, tab, n0, NT);
  else if (n0 < kPoly1From) general_chunk<kPoly2, true, R>(tp, tm, g, tab, n0, NT);
  else general_chunk<kPoly1, true, R>(tp, tm, g, tab, n0, NT);
#pragma unroll
  for (int r = 0; r < R; ++r) res[r] = centre[r] + fmaf(s[r], g.accM[r], g.accP[r]);
}

// 6 waves/SIMD (80 VGPRs) measured best: 4 -> 1.39 ms, 5 -> 1.29, 6 -> 1.25, 7 -> 1.32, 8 -> 1.59 (spills) per
// 115 M outputs.  Fully unrolling the tap loop (compile-time NT) was tried twice and spills badly.
//
// FUSED = true: there is no position array in HBM.  The workgroup regenerates the float64 positions of its
// tile into LDS from the plan (segment starts / offsets / speeds) and the per-segment cumsum checkpoints:
// one lane per checkpoint block advances kCk sequential float64 adds (bit-identical to numpy's cumsum) and
// scatters pos = cumsum + offset into P[]; everything downstream is the same code as the position-array path.
struct FusedArgs {
  const double* speeds;
  const int64_t* seg_start;
  const double* seg_off;
  const double* ck;
  const int64_t* tile_seg;
  int64_t nseg;
};
constexpr int kPosLds = kSincTile + 2;                 // positions jlo .. jhi of a tile (fused mode)
constexpr int kPosLdsFloats = 2 * kPosLds + 2;         // float slots they occupy (keeps the tile 16-B aligned)
constexpr int kSincCapFused = kSincCap;                 // same staging limit in both forms: identical fast/edge-path
                                                        // decisions, hence bit-identical outputs

__device__ __forceinline__ void generate_tile_positions(const FusedArgs& fa, int64_t j0, int64_t len_out, int64_t jlo,
                                                        int64_t jhi, double* __restrict__ P, int t, int n_threads) {
  const int64_t T = j0 / kSincTile;
  long long i0 = fa.tile_seg[T];
  if (jlo < fa.seg_start[i0]) i0 -= 1;                                    // j0 opens a segment: j0-1 is in the previous one
  const int64_t n_tiles = (len_out + kSincTile - 1) / kSincTile;
  const long long i1 = (jhi == (T + 1) * kSincTile) ? fa.tile_seg[T + 1] : fa.tile_seg[n_tiles];   // segment holding jhi
  const long long s0 = fa.seg_start[i0], s1 = fa.seg_start[i1];
  const long long g_lo = ck_slot0(s0, i0) + (jlo - s0) / kCk;
  const long long g_hi = ck_slot0(s1, i1) + (jhi - s1) / kCk;
  for (long long g = g_lo + t; g <= g_hi; g += n_threads) {
    // slot -> (segment, block): slot0 is monotone in the segment index
    long long lo = i0, hi = i1;
    if (hi - lo <= 16) {
      while (lo < hi && ck_slot0(fa.seg_start[lo + 1], lo + 1) <= g) ++lo;
    } else {
      while (lo < hi) {
        const long long mid = (lo + hi + 1) >> 1;
        if (ck_slot0(fa.seg_start[mid], mid) <= g) lo = mid; else hi = mid - 1;
      }
    }
    const long long i = lo;
    const long long start = fa.seg_start[i];
    const long long n = fa.seg_start[i + 1] - start;
    const long long b = g - ck_slot0(start, i);
    const long long k0 = b * kCk;
    if (k0 >= n) continue;                                                // unused gap slot
    const Ramp r = make_ramp(fa.speeds[i], fa.speeds[i + 1], n);
    const double off = fa.seg_off[i];
    double c = b ? fa.ck[g] : 0.0;
    double rr[kCk];
    const double a0 = (double)k0;
#pragma unroll
    for (int u = 0; u < kCk; ++u) rr[u] = ramp_recip(a0 + (double)u, r);
#pragma unroll
    for (int u = 0; u < kCk; ++u) {
#pragma clang fp contract(off)
      c = c + rr[u];
      const long long jj = start + k0 + u;
      if (k0 + u < n && jj >= jlo && jj <= jhi) P[jj - jlo] = c + off;
    }
  }
}

// NCH = 2: two channels of one file (same positions) in one launch.  A lane then owns 2 outputs x 2 channels instead of
// 4 outputs x 1: the register state and the per-lane ILP are those of the mono kernel, the workgroup has 512 threads
// for the same 1024-output tile, and everything that depends only on the POSITION -- regeneration from the plan,
// prologue, window-centre search, and (because both channel slots of an output carry the very same shift / fc
// values) the tap weights themselves -- is computed once for both channels.
template <bool FUSED, int NCH>
__global__ __launch_bounds__(kSincBlock * NCH, 6) void k_sinc(const double* __restrict__ pos, int64_t len_out,
                                                            const float* __restrict__ sig, const float* __restrict__ sig1,
                                                            int64_t sig_stride, int64_t len_in, int NT,
                                                            const float4* __restrict__ tab, float* __restrict__ out,
                                                            float* __restrict__ out1, int64_t out_stride, int64_t j_begin,
                                                            int64_t j_end, FusedArgs fa) {
  constexpr int kBlk = kSincBlock * NCH;        // threads per workgroup
  constexpr int kOut = kSincR / NCH;            // outputs per lane; kOut * NCH = kSincR (output, channel) slots
  static_assert(NCH == 1 || NCH == 2, "mono or stereo");
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  __shared__ int red[2 * (kBlk / kWave)];
  const int t = threadIdx.x;
  const int64_t j0 = j_begin + (int64_t)blockIdx.x * kSincTile;     // this launch covers outputs [j_begin, j_end)
  float* tile = FUSED ? lds_raw + kPosLdsFloats : lds_raw;
  const int cap = FUSED ? kSincCapFused : kSincCap;
  // position source: the caller's array, or the tile's positions regenerated into LDS
  double* P = reinterpret_cast<double*>(lds_raw);
  const int64_t jlo = j0 > 0 ? j0 - 1 : 0;
  if (FUSED) {
    const int64_t jhi = (j0 + kSincTile < len_out) ? j0 + kSincTile : len_out - 1;
    generate_tile_positions(fa, j0, len_out, jlo, jhi, P, t, kBlk);
    __syncthreads();
  }
  const double* psrc = FUSED ? P - jlo : pos;     // psrc[j] is the position of output j in both modes

  // Block-uniform EVEN integer anchor: indices are handled as int32 offsets from it (no 64-bit integer
  // math per output); an even anchor keeps round-half-even ties identical to rint(p).
  const double p0 = psrc[j0];
  const long long anchor = (fabs(p0) < 4.0e18) ? (llrint(p0) & ~1ll) : 0ll;
  const double anchor_d = (double)anchor;
  float res[kSincR];                               // one per (output, channel) slot: slot = output * NCH + channel
  int c[kOut];                                     // first: index relative to the anchor, later: LDS index
  float s[kOut], fc[kOut], dd[kOut];
  bool valid[kOut], fastlane[kOut], lowfc[kOut];
  bool unity = true, wild = false;
  int mn = INT_MAX, mx = INT_MIN;
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    const int64_t j = j0 + t + (int64_t)r * kBlk;
    valid[r] = j < j_end;
    lowfc[r] = false;
    c[r] = 0;
    s[r] = 0.25f;
    fc[r] = 1.0f;
    dd[r] = 0.0f;
    if (valid[r]) {
      const double p = psrc[j];
      // last output reuses the previous period (util/resampling.py:76-77)
      const double dp = (j + 1 < len_out) ? psrc[j + 1] - p : p - psrc[j - 1];
      const double rel = p - anchor_d;             // exact to ~1e-13: anchor is within a tile's span of p
      const double rf = rint(rel);
      if (fabs(rel) < 1.0e9) c[r] = (int)rf; else wild = true;
      const float sh = (float)(rel - rf);          // = p - rint(p)
      s[r] = (sh == 0.0f) ? 1e-20f : sh;           // np.sinc's own 0 -> 1e-20 substitution
      const bool one = !(dp > 1.0);                // fc == 1 (also catches the 1e-12 floor)
      // fc < 1/8 (an 8x slow-down of the read head and more): the output is a long average, small against the
      // signal, and float32 tap arithmetic (abs. error ~1e-6 of the signal level) would exceed 1e-5 of the OUTPUT
      // peak -- those lanes take the float64 path (not an audio-restoration regime; found by tools/fuzz_resampler.py)
      if (dp > 8.0) lowfc[r] = true;
      const float inv = fast_rcp((float)(dp > 1e-12 ? dp : 1e-12));
      fc[r] = one ? 1.0f : inv;
      dd[r] = one ? 0.0f : (float)(dp - 1.0) * inv;   // 1 - fc without cancellation
      unity = unity && one;
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
      tile[q] = inside ? sig[g * sig_stride] : 0.0f;
      if (NCH == 2) tile[cap + q] = inside ? sig1[g * sig_stride] : 0.0f;      // channel 1 right behind channel 0
    }
  }
  __syncthreads();

  // leading-edge outputs (ind < NT) keep the reference's mis-aligned taps: float64 slow path.
  bool anyfast = false;
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
  if (__any(anyfast)) {
    if (__all(unity)) {
      taps_unity<kSincR>(tile, cs, ss, NT, tab, res);
    } else {
      // the general path carries 10 live values per slot: two passes over half of the lane's slots keep
      // it inside the 80-VGPR budget (one pass spilled 48 B/lane = as much HBM write traffic as the output)
      static_assert(kSincR == 4, "split assumes 4 slots per lane");
      const int ca[2] = {cs[0], cs[1]}, cb[2] = {cs[2], cs[3]};
      const float sa[2] = {ss[0], ss[1]}, sb[2] = {ss[2], ss[3]};
      const float fa_[2] = {fcs[0], fcs[1]}, fb_[2] = {fcs[2], fcs[3]};
      const float da[2] = {dds[0], dds[1]}, db[2] = {dds[2], dds[3]};
      float ra[2], rb[2];
      taps_general<2>(tile, ca, sa, fa_, da, NT, tab, ra);
      taps_general<2>(tile, cb, sb, fb_, db, NT, tab, rb);
      res[0] = ra[0];
      res[1] = ra[1];
      res[2] = rb[0];
      res[3] = rb[1];
    }
  }
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    const int64_t j = j0 + t + (int64_t)r * kBlk;
    if (j >= j_end) continue;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      float v = res[r * NCH + ch];
      if (!fastlane[r]) {
        const double pj = psrc[j];
        const double dpj = (j + 1 < len_out) ? psrc[j + 1] - pj : pj - psrc[j - 1];
        v = sinc_one_f64(pj, dpj, ch ? sig1 : sig, sig_stride, len_in, NT);
      }
      (ch ? out1 : out)[j * out_stride] = v;
    }
  }
}

// ---- host side: per-(device, NT) tap tables -------------------------------------------------------
struct SincTable {
  float4* ab = nullptr;      // per-tap rows, see tap_R
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
  // a2[n] = pi*n^2/win, b[n] = -pi/win with win = float32(np.hanning(2NT+1)[NT+n]), n = 0..NT-1
  std::vector<float4> ab(NT + kChunk);
  for (int n = 0; n < NT + kChunk; ++n) {
    const double win = n < NT ? (double)(float)(0.5 + 0.5 * cos(M_PI * (double)n / (double)NT)) : 0.0;
    const double n2 = (double)n * (double)n;
    if (n < kPoly2From) {            // reciprocal rows; padded taps: rcp(q*b + inf) == 0
      ab[n] = n < NT ? make_float4((float)(M_PI * n2 / win), (float)(-M_PI / win), (float)n, 0.0f)
                     : make_float4(INFINITY, (float)(-M_PI), (float)n, 0.0f);
    } else {                         // series rows; padded taps: all-zero coefficients
      const double A = win / (M_PI * n2);
      ab[n] = make_float4((float)A, (float)(A / n2), (float)n, (float)(A / (n2 * n2)));
    }
  }
  SincTable t;
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
  hipLaunchKernelGGL((k_sinc<false, 1>), dim3((unsigned)blocks), dim3(kSincBlock), kSincCap * sizeof(float), s, pos, len_out,
                     sig, (const float*)nullptr, sig_stride, len_in, NT, tab.ab, out, (float*)nullptr, out_stride, j_begin,
                     j_begin + count, FusedArgs{});
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

// whole output range, positions regenerated in-kernel from the plan + checkpoints (no position array)
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
  const int64_t ck_len = max_out / kCk + m + 16;
  FusedArgs fa;
  fa.speeds = speeds;
  fa.seg_start = pv.seg_start;
  fa.seg_off = pv.seg_off;
  fa.ck = static_cast<const double*>(aux);
  fa.tile_seg = reinterpret_cast<const int64_t*>(fa.ck + ck_len);
  fa.nseg = m - 1;
  const int64_t blocks = ceil_div(len_out, kSincTile);
  if (sig1 && out1) {
    hipLaunchKernelGGL((k_sinc<true, 2>), dim3((unsigned)blocks), dim3(2 * kSincBlock),
                       (kPosLdsFloats + 2 * kSincCapFused) * sizeof(float), s, (const double*)nullptr, len_out, sig, sig1,
                       sig_stride, len_in, NT, tab.ab, out, out1, out_stride, (int64_t)0, len_out, fa);
  } else {
    hipLaunchKernelGGL((k_sinc<true, 1>), dim3((unsigned)blocks), dim3(kSincBlock),
                       (kPosLdsFloats + kSincCapFused) * sizeof(float), s, (const double*)nullptr, len_out, sig,
                       (const float*)nullptr, sig_stride, len_in, NT, tab.ab, out, (float*)nullptr, out_stride, (int64_t)0,
                       len_out, fa);
  }
  PAR_HIP_CHECK(hipGetLastError());
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
  return launch_sinc(device, pos, len_out, 0, len_out, sig, sig_stride, len_in, NT, out, out_stride, as_stream(stream));
}

}  // extern "C"
