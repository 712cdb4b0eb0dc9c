// K_pos -- speed curve -> float64 fractional read positions, bit-identical to the reference.
//
// Semantics: resampling.speed_to_pos (reference util/resampling.py:93-137).  For segment i between
// speed samples i and i+1:
//     a_i   = (st[i+1]-st[i]) * mean(speeds[i:i+2])                   (:103,:111)
//     n_i   = round-half-even(a_i + err), err carried                 (:113-118)  error diffusion
//     bs_k  = k/(n_i-1) * (s[i+1]-s[i]) + s[i]                        (:120)      speed ramp
//     pos   = cumsum(1/bs) + offset;  offset = pos[-1]                (:125-126)
//     trim at the first segment whose [first,last] straddles num_input_samples (:129-135)
//
// Everything is float64 in numpy's operation order with NO FMA contraction (build.py compiles this
// file with -ffp-contract=off; hip's __dmul_rn/__dadd_rn are plain inline operators that DO get fused
// under the default -ffp-contract=fast -- measured: positions off by 1 ulp).
//
// The reference walks the segments serially; two of its chains are order-dependent in floating point.
// Both are evaluated on the device by parallel scans that reproduce the serial result exactly:
//   (1) segment lengths.  n_i = N_i - N_{i-1} with N_i = round(sum_{t<=i} a_t): the running sum is
//       scanned in 128-bit fixed point (64 fractional bits, exact for a_i >= 2^-11).  The reference's
//       float64 chain deviates from exact arithmetic by at most a few 1e-11 (one 2^-45 rounding now and
//       then), so any cumulative sum closer than 2^-32 to a rounding tie is flagged and the whole plan is
//       redone by the serial host path (probability ~1e-3 per hour-long file).
//   (2) segment offsets  x_{i+1} = fl(x_i + S_i).  Inside one binade of x the float64 add is an integer
//       translation in units of ulp(x): X' = X + floor(S/u) + round-bit, where an exact half rounds to
//       even, i.e. depends only on the parity of X.  Such "parity-dependent translations" (c_even, c_odd)
//       are closed under composition, so each binade run is one segmented scan; the <= ~60 steps that
//       cross a binade (predicted from a plain float64 scan, then verified) are evaluated directly by a
//       single thread that stitches the runs together.  Any verification failure -> serial host path.
// The O(len_out) work (one IEEE division per output sample, twice: reciprocal sums, then the fill) is
// one lane per segment; k/(n-1) uses Markstein's correctly-rounded quotient from y = RN(1/(n-1)).
#include "par_common.h"
#include "pos_plan.h"
#include <math.h>
#include <vector>

#pragma clang fp contract(off)

namespace par {

// ------------------------------------------------------------------------------ generic scans
struct AddU128 {
  using T = U128;
  __device__ static T identity() { return T{0ull, 0ull}; }
  __device__ static T combine(T a, T b) {          // a then b
    T r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ull : 0ull);
    return r;
  }
};
struct AddF64 {
  using T = double;
  __device__ static T identity() { return 0.0; }
  __device__ static T combine(T a, T b) { return a + b; }
};
struct ComposeP {
  using T = PElem;
  __device__ static T identity() { return T{0, 0, 0, 0}; }
  __device__ static T combine(T a, T b) {          // apply a first, then b; b.head restarts
    if (b.head) return b;
    T r;
    r.c0 = a.c0 + ((a.c0 & 1) ? b.c1 : b.c0);
    r.c1 = a.c1 + (((1 + a.c1) & 1) ? b.c1 : b.c0);
    r.head = a.head;
    r.pad = 0;
    return r;
  }
};

template <typename T>
__device__ __forceinline__ T shfl_up_any(T v, int delta) {
  static_assert(sizeof(T) % 8 == 0, "scan element must be a multiple of 8 bytes");
  union {
    T t;
    unsigned long long w[sizeof(T) / 8];
  } u;
  u.t = v;
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 8; ++i) u.w[i] = __shfl_up(u.w[i], delta, kWave);
  return u.t;
}

constexpr int kScanThreads = 256;
constexpr int kScanItems = 4;
constexpr int kScanTile = kScanThreads * kScanItems;   // 1024

// scan of the per-thread totals across the block; returns the EXCLUSIVE prefix of this thread and the
// block total (valid in every thread).
template <typename Op>
__device__ typename Op::T block_exclusive(typename Op::T mine, typename Op::T* smem, typename Op::T* block_total) {
  using T = typename Op::T;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  T inc = mine;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    T up = shfl_up_any(inc, d);
    if (lane >= d) inc = Op::combine(up, inc);
  }
  if (lane == kWave - 1) smem[wave] = inc;
  __syncthreads();
  T wave_prefix = Op::identity();
  T total = Op::identity();
#pragma unroll
  for (int w = 0; w < kScanThreads / kWave; ++w) {
    if (w < wave) wave_prefix = Op::combine(wave_prefix, smem[w]);
    total = Op::combine(total, smem[w]);
  }
  __syncthreads();
  T excl = shfl_up_any(inc, 1);
  if (lane == 0) excl = Op::identity();
  *block_total = total;
  return Op::combine(wave_prefix, excl);
}

template <typename Op>
__global__ __launch_bounds__(kScanThreads) void k_scan_reduce(const typename Op::T* __restrict__ data, int64_t n,
                                                               typename Op::T* __restrict__ bsum) {
  using T = typename Op::T;
  __shared__ T smem[kScanThreads / kWave];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  T acc = Op::identity();
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (base + k < n) acc = Op::combine(acc, data[base + k]);
  T total;
  block_exclusive<Op>(acc, smem, &total);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// single block: in-place EXCLUSIVE scan of the block sums (loops over tiles with a carry)
template <typename Op>
__global__ __launch_bounds__(kScanThreads) void k_scan_top(typename Op::T* __restrict__ bsum, int64_t nb) {
  using T = typename Op::T;
  __shared__ T smem[kScanThreads / kWave];
  T carry = Op::identity();
  for (int64_t t0 = 0; t0 < nb; t0 += kScanTile) {
    const int64_t base = t0 + (int64_t)threadIdx.x * kScanItems;
    T v[kScanItems];
    T acc = Op::identity();
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      v[k] = (base + k < nb) ? bsum[base + k] : Op::identity();
      acc = Op::combine(acc, v[k]);
    }
    T total;
    T run = Op::combine(carry, block_exclusive<Op>(acc, smem, &total));
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      if (base + k < nb) bsum[base + k] = run;
      run = Op::combine(run, v[k]);
    }
    carry = Op::combine(carry, total);
  }
}

// in-place INCLUSIVE scan with the block prefixes applied
template <typename Op>
__global__ __launch_bounds__(kScanThreads) void k_scan_apply(typename Op::T* __restrict__ data, int64_t n,
                                                              const typename Op::T* __restrict__ bsum) {
  using T = typename Op::T;
  __shared__ T smem[kScanThreads / kWave];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  T v[kScanItems];
  T acc = Op::identity();
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = (base + k < n) ? data[base + k] : Op::identity();
    acc = Op::combine(acc, v[k]);
  }
  T total;
  T run = Op::combine(bsum[blockIdx.x], block_exclusive<Op>(acc, smem, &total));
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    run = Op::combine(run, v[k]);
    if (base + k < n) data[base + k] = run;
  }
}

template <typename Op>
static int inclusive_scan(typename Op::T* data, int64_t n, typename Op::T* bsum, hipStream_t s) {
  const int64_t nb = ceil_div(n, kScanTile);
  hipLaunchKernelGGL(k_scan_reduce<Op>, dim3((unsigned)nb), dim3(kScanThreads), 0, s, data, n, bsum);
  hipLaunchKernelGGL(k_scan_top<Op>, dim3(1), dim3(kScanThreads), 0, s, bsum, nb);
  hipLaunchKernelGGL(k_scan_apply<Op>, dim3((unsigned)nb), dim3(kScanThreads), 0, s, data, n, bsum);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

// ------------------------------------------------------------------------- stage kernels
__global__ void k_init_header(PlanHeader* h, int64_t m, int lazy) {
  h->m = m;
  h->lazy = lazy;
  h->lazy_fail = 0;
  h->n_cand = 0;
  h->pad4 = 0;
  h->len_out = 0;
  h->total_written = 0;
  h->trim_seg = kNoTrim;
  h->cap = 0;
  h->trimmed = 0;
  h->flags = 0;
  h->n_direct = 0;
  h->n_runs = 0;
  h->speed_sum = 0.0;
  h->ck_len = 0;
  h->ck_valid = 0;
  h->pad2 = 0;
  h->n_long = 0;
  h->pad3 = 0;
  h->written = 0;
  h->first_bad = kNoTrim;
}

// a_i (util/resampling.py:103,:111) in numpy's order, converted EXACTLY to 64.64 fixed point.
__global__ void k_seg_want(const double* __restrict__ st, const double* __restrict__ sp, int64_t nseg,
                           U128* __restrict__ a_fix, PlanHeader* __restrict__ h) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nseg) return;
  const double period = st[i + 1] - st[i];
  const double mean = (sp[i] + sp[i + 1]) / 2.0;
  const double a = period * mean;
  U128 e{0ull, 0ull};
  if (!(a >= 0x1p-11 && a < 0x1p62)) {
    atomicOr(&h->flags, kFlagRange);
  } else {
    const double ip = floor(a);
    e.hi = (unsigned long long)ip;
    e.lo = (unsigned long long)((a - ip) * 0x1p64);     // exact: < 2^64, at most 64 fractional bits
  }
  a_fix[i] = e;
}

__device__ __forceinline__ unsigned long long round_fixed(U128 A, bool* ambiguous) {
  const unsigned long long half = 1ull << 63, zone = 1ull << 32;
  const unsigned long long d = A.lo > half ? A.lo - half : half - A.lo;
  if (d < zone) *ambiguous = true;
  return A.hi + (A.lo > half ? 1ull : 0ull);
}

// n_i = round(A_i) - round(A_{i-1});  seg_start[i] = round(A_{i-1})
__global__ void k_seg_lengths(const U128* __restrict__ A, int64_t nseg, int64_t* __restrict__ seg_start,
                              PlanHeader* __restrict__ h) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nseg) return;
  bool amb = false;
  const unsigned long long Ni = round_fixed(A[i], &amb);
  const unsigned long long Np = i > 0 ? round_fixed(A[i - 1], &amb) : 0ull;
  seg_start[i] = (int64_t)Np;
  if (i == nseg - 1) {
    seg_start[nseg] = (int64_t)Ni;
    h->total_written = (int64_t)Ni;
  }
  int f = 0;
  if (amb) f |= kFlagAmbiguous;
  // n_i < 2: the reference divides by zero / indexes an empty array -- if it ever gets there.  Downstream the segment
  // counts as empty (S_i = 0, cannot straddle n_in); k_trim raises the flag unless the trim fires in front of it.
  if (Ni < Np + 2ull) atomicMin(&h->first_bad, (unsigned long long)i);
  if (f) atomicOr(&h->flags, f);
}

// sum(speeds) for the reference's buffer bound int(mean(speeds) * span * 1.01) (:108).  numpy sums pairwise; any other
// order may differ in the last bits, so this sum is made ACCURATE instead (compensated per thread, tree across lanes,
// compensated + tree again over the per-wave partials in the last workgroup: <= ~18 ulp-roundings, 2e-15 relative)
// and k_trim sends the plan to the serial path -- which restates numpy's order -- in the rare case where that
// uncertainty could move int().
__global__ void __launch_bounds__(256) k_speed_sum(const double* __restrict__ sp, int64_t m, double* __restrict__ partial,
                                                   PlanHeader* __restrict__ h) {
  __shared__ double red[256];
  __shared__ int is_last;
  double s = 0.0, c = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const double y = sp[i] - c;          // Kahan: pos.hip is built with -ffp-contract=off and no fast-math
    const double t = s + y;
    c = (t - s) - y;
    s = t;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, kWave);
  const int waves = blockDim.x / kWave;
  if ((threadIdx.x & (kWave - 1)) == 0) partial[blockIdx.x * waves + threadIdx.x / kWave] = s;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) is_last = atomicAdd(&h->pad2, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  const int64_t n_part = (int64_t)gridDim.x * waves;
  s = 0.0;
  c = 0.0;
  for (int64_t i = threadIdx.x; i < n_part; i += blockDim.x) {
    const double y = __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - c;
    const double t = s + y;
    c = (t - s) - y;
    s = t;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    h->speed_sum = red[0];
    h->pad2 = 0;                                 // the counter serves k_off_apply_trim next
  }
}

// S_i = last element of np.cumsum(1/block_speeds): strictly sequential float64 adds, one lane per segment.
// With ck != nullptr the running sum is also checkpointed every kCk steps (slot layout: pos_plan.h) for the
// fused resampler, which then regenerates the positions of a tile from the nearest checkpoint.
//
// Checkpoint stores go through a 64 x 16 LDS transpose: written lane-by-lane they would be 64 scattered 8-byte
// stores per instruction (one segment ~ 33 slots apart per lane: a 64-byte DRAM sector per 8 useful bytes);
// transposed, a quarter wave writes 16 consecutive slots of ONE segment (128-byte runs).
#ifndef PAR_SEG_SUM_PREDICT
#define PAR_SEG_SUM_PREDICT 1     // k_seg_sum: divisions of gentle ramps start from the previous block's reciprocals (A/B knob)
#endif
constexpr int kCkRound = 16;       // checkpoints per lane per transpose round (= 128 cumsum steps; 8.7 KB LDS per wave)
__device__ __forceinline__ int f64_exponent(double x) { return (int)((__double_as_longlong(x) >> 52) & 0x7ff); }

// ---- long segments (sparse speed curves) ------------------------------------------------------------------
// One lane per segment is fine while segments are a few hundred samples long.  A curve with a handful of points
// (a constant speed correction is TWO points) makes segments of 10^7..10^9 samples, and the sequential float64
// cumsum of such a segment on one lane takes seconds.  Those segments are cut into chunks of 256 steps and their
// cumsum chain c' = fl(c + r_k) is evaluated EXACTLY in parallel with the parity-translation algebra of the offset
// chain (see the header comment): per chunk the 256 single-add maps are composed into one map (valid while the
// running sum stays in one binade), chunk starts come from a scan of the maps, the ~30 chunks in which the sum
// crosses a power of two are stepped through sequentially, and a final pass recomputes every chunk from its exact
// start, writes the checkpoints and VERIFIES that it lands bit-exactly on the next chunk's start.
constexpr long long kLongSeg = 32768;                   // longer segments take the chunked path (needs the ck buffer); below,
                                                        // one lane per segment is faster (measured crossover 3e4..5e4)
constexpr int kLongChunk = 256;                         // least steps per chunk; the last chunk of a segment takes the remainder
constexpr int kLongSlots = kLongChunk / kCk;            // checkpoint slots a chunk owns: its scratch lives there first
enum { kLsA = 8, kLsC0 = 9, kLsC1 = 10, kLsExp = 11, kLsStart = 12, kLsEnd = 13, kLsApprox = 14 };

__host__ __device__ inline long long chunk_slot0(long long seg_start, long long i) { return seg_start / kLongChunk + i; }
__device__ __forceinline__ bool long_segment(long long n, long long start, long long i, const double* ck, int64_t ck_len) {
  return ck != nullptr && n > kLongSeg && ck_slot0(start, i) + (n + kCk - 1) / kCk <= ck_len;
}

#ifndef PAR_PLAN_VGPR_CAP
#define PAR_PLAN_VGPR_CAP 0          // experiment: cap the VGPRs of the plan's two widest kernels (k_seg_sum 44, k_scan_top 40)
#endif
#if PAR_PLAN_VGPR_CAP
#define PLAN_VGPR_CAP __attribute__((amdgpu_num_vgpr(PAR_PLAN_VGPR_CAP)))
#else
#define PLAN_VGPR_CAP
#endif
__global__ __launch_bounds__(64) PLAN_VGPR_CAP void k_seg_sum(const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                                                 int64_t nseg, double* __restrict__ S, double* __restrict__ ck,
                                                 int64_t ck_len, PlanHeader* __restrict__ h, int count_long) {
  __shared__ double T[kWave][kCkRound + 1];
  __shared__ long long slot_of[kWave];
  __shared__ int n_ck[kWave];
  const int lane = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + lane;
  long long n = 0, slot0 = 0;
  Ramp r = make_ramp(1.0, 1.0, 2);
  bool ck_ok = false, is_long = false;
  if (i < nseg) {
    const long long start = seg_start[i];
    n = seg_start[i + 1] - start;
    if (n >= 2) {
      r = make_ramp(sp[i], sp[i + 1], n);
      slot0 = ck_slot0(start, i);
      // segments whose slots do not fit are skipped here; whether any of them is actually needed (starts
      // before len_out) is decided by k_tile_seg once the trim is known
      ck_ok = ck != nullptr && slot0 + (n + kCk - 1) / kCk <= ck_len;
      if (long_segment(n, start, i, ck, ck_len)) {      // the k_long_* kernels own this segment (S[i] included)
        is_long = true;
        n = 0;
        if (count_long) {
          atomicAdd(&h->n_long, 1);
          S[i] = 0.0;
          // first go of a sparse curve (mode 1): the plan is made again with the chunked cumsum, so the rest of THIS pass is void --
          // the kernels behind see the mark and leave at once instead of scanning undefined sums
          if (count_long == 1) atomicOr(&h->lazy_fail, 4);
        }
      }
    } else {
      n = 0;
    }
  }
  // checkpoints b = 1 .. (n-1)/kCk hold the cumsum after step kCk*b - 1 (written only when steps follow)
  const long long my_ck = ck_ok ? (n - 1) / kCk : 0;
  slot_of[lane] = slot0;
  const long long n_blocks = n / kCk;                       // full kCk-step blocks
  const long long max_blocks = wave_max_ll(n_blocks);
  double c = 0.0;
  // Gentle ramps (every lane of the wave: the speed changes by <= 1e-5 of itself over kCk steps) start each division from
  // the reciprocal kCk steps back, 1/(b + d) = x - x^2 d + O(x^3 d^2) <= 1e-10 off, instead of from v_rcp_f64's 2^-27 or so:
  // ONE Newton step then lands within half an ulp and the residual correction of recip_unscaled rounds it correctly
  // (Markstein: any start within one ulp does) -- the same bits for a quarter-rate v_rcp_f64 and two fmas less per step.
  const double smin_ = fabs(r.s0) < fabs(r.s0 + r.ds) ? fabs(r.s0) : fabs(r.s0 + r.ds);
  const double d8 = r.ds * r.y * (double)kCk;                 // b_{k + kCk} - b_k (to a rounding)
  const bool pred = PAR_SEG_SUM_PREDICT && __all(r.fast && fabs(d8) <= 1.0e-5 * smin_);
  double xp[kCk];                                             // the reciprocals of the previous block
#pragma unroll
  for (int u = 0; u < kCk; ++u) xp[u] = 0.0;
  for (long long b0 = 0; b0 < max_blocks; b0 += kCkRound) {
    int filled = 0;
    for (int w = 0; w < kCkRound; ++w) {
      const long long b = b0 + w;
      if (b >= n_blocks) break;
      const double a0 = (double)(b * kCk);
      double rr[kCk];                                         // kCk independent divisions in flight per block
      if (pred && b > 0) {
#pragma unroll
        for (int u = 0; u < kCk; ++u) {
          const double bs = ramp_value(a0 + (double)u, r);
          const double x0 = __builtin_fma(-xp[u], xp[u] * d8, xp[u]);
          const double x1 = __builtin_fma(x0, __builtin_fma(-bs, x0, 1.0), x0);
          rr[u] = __builtin_fma(__builtin_fma(-bs, x1, 1.0), x1, x1);
        }
      } else {
#pragma unroll
        for (int u = 0; u < kCk; ++u) rr[u] = ramp_recip(a0 + (double)u, r);
      }
#pragma unroll
      for (int u = 0; u < kCk; ++u) xp[u] = rr[u];
#pragma unroll
      for (int u = 0; u < kCk; ++u) c = c + rr[u];
      T[lane][w] = c;                                         // checkpoint b + 1
      filled = w + 1;
    }
    if (ck != nullptr) {
      long long lim = my_ck - b0;                             // checkpoints of this round that are to be stored
      if (lim > filled) lim = filled;
      n_ck[lane] = lim > 0 ? (int)lim : 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int w = lane & (kCkRound - 1), half = lane / kCkRound;
      for (int sgm = half; sgm < kWave; sgm += kWave / kCkRound)
        if (w < n_ck[sgm]) ck[slot_of[sgm] + b0 + w + 1] = T[sgm][w];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  for (long long k = n_blocks * kCk; k < n; ++k) c = c + ramp_recip((double)k, r);
  if (i < nseg && !is_long) S[i] = c;
}

__global__ void k_count_long(const int64_t* __restrict__ seg_start, int64_t nseg, const double* __restrict__ ck,
                             int64_t ck_len, PlanHeader* __restrict__ h) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nseg) return;
  const long long start = seg_start[i];
  if (long_segment(seg_start[i + 1] - start, start, i, ck, ck_len)) atomicAdd(&h->n_long, 1);
}

struct ChunkRef {
  long long i, j, J, n, k0, k1, base, slot0;
  bool ok;
};
// global chunk slot g = chunk_slot0(start_i, i) + j is unique and monotone in (i, j) without any prefix sum
__device__ __forceinline__ ChunkRef find_chunk(long long g, const int64_t* __restrict__ seg_start, int64_t nseg,
                                               const double* ck, int64_t ck_len) {
  long long lo = 0, hi = nseg - 1;
  while (lo < hi) {
    const long long mid = (lo + hi + 1) >> 1;
    if (chunk_slot0(seg_start[mid], mid) <= g) lo = mid; else hi = mid - 1;
  }
  ChunkRef c;
  c.i = lo;
  const long long start = seg_start[lo];
  c.n = seg_start[lo + 1] - start;
  c.j = g - chunk_slot0(start, lo);
  c.J = c.n / kLongChunk;
  c.ok = c.j >= 0 && c.j < c.J && long_segment(c.n, start, lo, ck, ck_len);
  c.k0 = c.j * kLongChunk;
  c.k1 = (c.j == c.J - 1) ? c.n : c.k0 + kLongChunk;
  c.slot0 = ck_slot0(start, lo);
  c.base = c.slot0 + c.j * kLongSlots;
  return c;
}

// pass A: plain float64 sum of the chunk's reciprocals (only used to PREDICT the binade of the running sum)
__global__ __launch_bounds__(256) void k_long_approx(const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                                                     int64_t nseg, double* __restrict__ ck, int64_t ck_len, long long G,
                                                     const PlanHeader* __restrict__ h) {
  if (h->n_long == 0) return;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const ChunkRef c = find_chunk(g, seg_start, nseg, ck, ck_len);
  if (!c.ok) return;
  const Ramp r = make_ramp(sp[c.i], sp[c.i + 1], c.n);
  double acc[kCk] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long k = c.k0;
  for (; k + kCk <= c.k1; k += kCk) {
    const double a0 = (double)k;
#pragma unroll
    for (int u = 0; u < kCk; ++u) acc[u] += ramp_recip(a0 + (double)u, r);
  }
  double A = 0.0;
#pragma unroll
  for (int u = 0; u < kCk; ++u) A += acc[u];
  for (; k < c.k1; ++k) A += ramp_recip((double)k, r);
  ck[c.base + kLsA] = A;
}

// The chunk records of a segment are reduced in WINDOWS of 256 * kWinC chunks, three phases per reduction: every window
// on its own workgroup (a thread owns kWinC consecutive chunks, thread totals are scanned with wave shuffles), one
// workgroup per segment over the window totals (256 windows per step), every window again to hand the result down to
// its chunks.  A 7e8-sample segment is 2.7 M chunks = 660 windows: the only serial part left is 3 steps long.
// Window records live in free fields of the window's first chunk (a chunk owns 32 checkpoint slots).
constexpr int kWinC = 16;
constexpr int kWin = 256 * kWinC;
enum { kLsWSum = 16, kLsWPre = 17, kLsWC0 = 18, kLsWC1 = 19, kLsWClean = 20, kLsWStart = 21 };
static_assert(kLsWStart < kLongChunk / kCk, "window record must fit the chunk's own slots");

// Global window slot, again without a prefix sum: chunk slots of consecutive segments are at least J_i + 1 apart and a
// long segment has J_i >= 128 chunks, so floor(chunk_slot0 / 64) leaves floor((J_i + 1) / 64) >= ceil(J_i / kWin) + 1
// slots between a long segment and whatever follows it: windows w = 0.. of segment i sit at win_slot0(i) + w, and the
// LAST segment whose win_slot0 is <= a slot is its owner (short segments in front may share the value, none behind).
constexpr int kWinSlotDiv = 64;
static_assert(kLongSeg / kLongChunk >= 2 * kWinSlotDiv && kWin >= kWinSlotDiv, "window slots of long segments would collide");
__host__ __device__ inline long long win_slot0(long long seg_start, long long i) { return chunk_slot0(seg_start, i) / kWinSlotDiv; }
struct WinRef {
  long long i, n, J, slot0, j0, j1, rec;      // rec: slot of the window record (= first chunk of the window)
  bool ok;
};
__device__ __forceinline__ WinRef find_window(long long gw, const int64_t* __restrict__ seg_start, int64_t nseg,
                                              const double* ck, int64_t ck_len) {
  long long lo = 0, hi = nseg - 1;
  while (lo < hi) {
    const long long mid = (lo + hi + 1) >> 1;
    if (win_slot0(seg_start[mid], mid) <= gw) lo = mid; else hi = mid - 1;
  }
  WinRef r;
  r.i = lo;
  const long long start = seg_start[lo];
  r.n = seg_start[lo + 1] - start;
  r.J = r.n / kLongChunk;
  const long long w = gw - win_slot0(start, lo);
  r.ok = w >= 0 && w * kWin < r.J && long_segment(r.n, start, lo, ck, ck_len);
  r.slot0 = ck_slot0(start, lo);
  r.j0 = w * kWin;
  r.j1 = r.j0 + kWin < r.J ? r.j0 + kWin : r.J;
  r.rec = r.slot0 + r.j0 * kLongSlots;
  return r;
}

// exclusive prefix of `mine` over the 256 threads (+ the total), wave shuffles + one barrier; tot[] is 4 doubles of LDS
__device__ __forceinline__ double block_scan_sum(double mine, double* tot, double* total) {
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  double inc = mine;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const double up = __shfl_up(inc, o, kWave);
    if (lane >= o) inc += up;
  }
  if (lane == kWave - 1) tot[wv] = inc;
  __syncthreads();
  double before = 0.0, all = 0.0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wv) before += tot[w];
    all += tot[w];
  }
  *total = all;
  return before + (inc - mine);
}

// pass B1: approximate sum of every window
__global__ __launch_bounds__(256) void k_long_wsum(const int64_t* __restrict__ seg_start, int64_t nseg, double* __restrict__ ck,
                                                   int64_t ck_len, long long GW, const PlanHeader* __restrict__ h) {
  if (h->n_long == 0) return;
  __shared__ double tot[4];
  for (long long gw = blockIdx.x; gw < GW; gw += gridDim.x) {       // window slots, grid-strided (most are empty)
    const WinRef w = find_window(gw, seg_start, nseg, ck, ck_len);
    if (!w.ok) continue;
    const long long jt = w.j0 + (long long)threadIdx.x * kWinC;
    double v[kWinC];
#pragma unroll
    for (int u = 0; u < kWinC; ++u) {                                  // unconditional loads: all in flight at once
      const long long jc = jt + u < w.j1 ? jt + u : w.j1 - 1;
      v[u] = ck[w.slot0 + jc * kLongSlots + kLsA];
    }
    double mine = 0.0;
#pragma unroll
    for (int u = 0; u < kWinC; ++u) mine += jt + u < w.j1 ? v[u] : 0.0;
    double total;
    block_scan_sum(mine, tot, &total);
    if (threadIdx.x == 0) ck[w.rec + kLsWSum] = total;
    __syncthreads();                                                 // LDS scratch is reused by the next slot
  }
}

// pass B2: per long segment, exclusive prefix of the window sums (256 windows per step)
__global__ __launch_bounds__(256) void k_long_prefix(const int64_t* __restrict__ seg_start, int64_t nseg,
                                                     double* __restrict__ ck, int64_t ck_len,
                                                     const PlanHeader* __restrict__ h) {
  if (h->n_long == 0) return;
  __shared__ double tot[4];
  for (long long i = blockIdx.x; i < nseg; i += gridDim.x) {
    const long long start = seg_start[i], n = seg_start[i + 1] - start;
    if (!long_segment(n, start, i, ck, ck_len)) continue;            // uniform over the workgroup
    const long long J = n / kLongChunk, NW = (J + kWin - 1) / kWin, slot0 = ck_slot0(start, i);
    double carry = 0.0;                                              // same value in every thread
    for (long long w0 = 0; w0 < NW; w0 += 256) {
      const long long w = w0 + threadIdx.x;
      const long long rec = slot0 + (w < NW ? w : NW - 1) * kWin * kLongSlots;
      const double v = w < NW ? ck[rec + kLsWSum] : 0.0;
      double total;
      const double before = block_scan_sum(v, tot, &total);
      if (w < NW) ck[rec + kLsWPre] = carry + before;
      carry += total;
      __syncthreads();
    }
  }
}

// pass B3: approximate running sum at every chunk start
__global__ __launch_bounds__(256) void k_long_wprefix(const int64_t* __restrict__ seg_start, int64_t nseg, double* __restrict__ ck,
                                                      int64_t ck_len, long long GW, const PlanHeader* __restrict__ h) {
  if (h->n_long == 0) return;
  __shared__ double tot[4];
  for (long long gw = blockIdx.x; gw < GW; gw += gridDim.x) {       // window slots, grid-strided (most are empty)
    const WinRef w = find_window(gw, seg_start, nseg, ck, ck_len);
    if (!w.ok) continue;
    const long long jt = w.j0 + (long long)threadIdx.x * kWinC;
    const double pre = ck[w.rec + kLsWPre];
    double v[kWinC];
#pragma unroll
    for (int u = 0; u < kWinC; ++u) {
      const long long jc = jt + u < w.j1 ? jt + u : w.j1 - 1;
      v[u] = ck[w.slot0 + jc * kLongSlots + kLsA];
    }
    double mine = 0.0;
#pragma unroll
    for (int u = 0; u < kWinC; ++u) mine += jt + u < w.j1 ? v[u] : 0.0;
    double total;
    double run = pre + block_scan_sum(mine, tot, &total);
#pragma unroll
    for (int u = 0; u < kWinC; ++u) {
      if (jt + u < w.j1) ck[w.slot0 + (jt + u) * kLongSlots + kLsApprox] = run;
      run += v[u];
    }
    __syncthreads();                                                 // LDS scratch is reused by the next slot
  }
}

// one float64 add x -> fl(x + r) inside binade e as a parity-dependent integer translation (units of ulp)
__device__ __forceinline__ bool add_as_map(double r, int e, long long* d0, long long* d1) {
  const double t = ldexp(r, 1075 - e);
  if (!(t >= 0.0 && t < 0x1p62)) return false;
  const double fl = floor(t);
  const long long q = (long long)fl;
  const double fr = t - fl;                    // exact
  if (fr > 0.5) *d0 = *d1 = q + 1;
  else if (fr < 0.5) *d0 = *d1 = q;
  else {                                       // exact half: ties-to-even on the SUM's parity
    *d0 = q + (q & 1);
    *d1 = q + ((q + 1) & 1);
  }
  return true;
}

// pass C: compose the chunk's single-add maps when the running sum provably stays in one binade
__global__ __launch_bounds__(256) void k_long_map(const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                                                  int64_t nseg, double* __restrict__ ck, int64_t ck_len, long long G,
                                                  const PlanHeader* __restrict__ h) {
  if (h->n_long == 0) return;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const ChunkRef c = find_chunk(g, seg_start, nseg, ck, ck_len);
  if (!c.ok) return;
  const double xa = ck[c.base + kLsApprox], xb = xa + ck[c.base + kLsA];
  const double lo = 1.0 - 0x1p-30, hi = 1.0 + 0x1p-30;
  const int e = f64_exponent(xa);
  bool interior = c.j > 0 && xa > 0.0 && e > 64 && e < 2046 && f64_exponent(xa * lo) == e && f64_exponent(xa * hi) == e &&
                  f64_exponent(xb * lo) == e && f64_exponent(xb * hi) == e;
  long long c0 = 0, c1 = 0;
  if (interior) {
    const Ramp r = make_ramp(sp[c.i], sp[c.i + 1], c.n);
    for (long long k = c.k0; k < c.k1 && interior; k += kCk) {
      double rr[kCk];
#pragma unroll
      for (int u = 0; u < kCk; ++u) rr[u] = ramp_recip((double)(k + u), r);
#pragma unroll
      for (int u = 0; u < kCk; ++u) {
        if (k + u < c.k1) {
          long long d0, d1;
          if (!add_as_map(rr[u], e, &d0, &d1)) interior = false;
          else {
            const long long n0 = c0 + ((c0 & 1) ? d1 : d0);
            const long long n1 = c1 + (((1 + c1) & 1) ? d1 : d0);
            c0 = n0;
            c1 = n1;
          }
        }
      }
    }
  }
  ck[c.base + kLsC0] = __longlong_as_double(c0);
  ck[c.base + kLsC1] = __longlong_as_double(c1);
  ck[c.base + kLsExp] = interior ? (double)e : -1.0;
}

__device__ __forceinline__ double apply_map(long long c0, long long c1, double x) {
  const int e = f64_exponent(x);
  const long long X = (long long)ldexp(x, 1075 - e);      // exact integer in [2^52, 2^53)
  const long long Y = X + ((X & 1) ? c1 : c0);
  return ldexp((double)Y, e - 1075);
}

// (a then b) of two parity-translation maps
__device__ __forceinline__ void map_then(long long a0, long long a1, long long b0, long long b1, long long* r0,
                                         long long* r1) {
  *r0 = a0 + ((a0 & 1) ? b1 : b0);
  *r1 = a1 + (((1 + a1) & 1) ? b1 : b0);
}

// this thread's kWinC chunk maps of the chunk range [jt, jt + kWinC) clipped to j_end (identity beyond it and for
// direct chunks); first = offset (from jt0, the range start handed to thread 0) of my first direct chunk, else kWin
// per: chunks per thread in this pass (kWinC, or 1 while a marked window is searched for its next direct chunk); the
// thread's chunks are jt .. jt + per - 1
__device__ __forceinline__ void load_chunk_maps(const double* __restrict__ ck, long long slot0, long long jt, long long j_end,
                                                int per, long long (&m0)[kWinC], long long (&m1)[kWinC], int* first) {
  double ex[kWinC];
#pragma unroll
  for (int u = 0; u < kWinC; ++u) {                                  // unconditional loads: all in flight at once
    const long long jc = (u < per && jt + u < j_end) ? jt + u : j_end - 1;
    const long long base = slot0 + jc * kLongSlots;
    ex[u] = ck[base + kLsExp];
    m0[u] = __double_as_longlong(ck[base + kLsC0]);
    m1[u] = __double_as_longlong(ck[base + kLsC1]);
  }
  int f = kWin;
#pragma unroll
  for (int u = kWinC - 1; u >= 0; --u) {
    const bool live = u < per && jt + u < j_end, direct = live && ex[u] < 0.0;
    if (direct) f = (int)threadIdx.x * per + u;
    if (!live || direct) m0[u] = m1[u] = 0;
  }
  *first = f;
}

// minimum of v over the 256 threads (one barrier; buf is 4 ints of LDS)
__device__ __forceinline__ int block_min(int v, int* buf) {
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) {
    const int other = __shfl_xor(v, o, kWave);
    v = other < v ? other : v;
  }
  if (lane == 0) buf[wv] = v;
  __syncthreads();
  int r = buf[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) r = buf[w] < r ? buf[w] : r;
  return r;
}

// ordered composition over the 256 threads: (p0, p1) = everything in front of this thread, (all0, all1) = all of it;
// one barrier, w0/w1 are 4 long longs of LDS each
__device__ __forceinline__ void block_scan_maps(long long a0, long long a1, long long* w0, long long* w1, long long* p0,
                                                long long* p1, long long* all0, long long* all1) {
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  long long i0 = a0, i1 = a1;                                         // inclusive inside the wave
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const long long u0 = __shfl_up(i0, o, kWave), u1 = __shfl_up(i1, o, kWave);
    if (lane >= o) map_then(u0, u1, i0, i1, &i0, &i1);
  }
  if (lane == kWave - 1) {
    w0[wv] = i0;
    w1[wv] = i1;
  }
  long long e0 = __shfl_up(i0, 1, kWave), e1 = __shfl_up(i1, 1, kWave);          // exclusive inside the wave
  if (lane == 0) e0 = e1 = 0;
  __syncthreads();
  long long q0 = 0, q1 = 0, t0 = 0, t1 = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wv) map_then(q0, q1, w0[w], w1[w], &q0, &q1);
    map_then(t0, t1, w0[w], w1[w], &t0, &t1);
  }
  map_then(q0, q1, e0, e1, p0, p1);
  *all0 = t0;
  *all1 = t1;
}

// exact start / end of this thread's first `cnt` chunks from jt on, given the running sum x0 at the start of the range
// and the composed map (p0, p1) of everything in the range in front of them
__device__ __forceinline__ void write_chunk_bounds(double* __restrict__ ck, long long slot0, long long jt, long long cnt,
                                                   double x0, long long p0, long long p1, const long long (&m0)[kWinC],
                                                   const long long (&m1)[kWinC]) {
  if (cnt <= 0) return;
  const int e = f64_exponent(x0);
  long long X = (long long)ldexp(x0, 1075 - e);                       // exact integer in [2^52, 2^53)
  X += (X & 1) ? p1 : p0;
#pragma unroll
  for (int u = 0; u < kWinC; ++u) {
    if (u < cnt) {
      const long long base = slot0 + (jt + u) * kLongSlots;
      ck[base + kLsStart] = ldexp((double)X, e - 1075);
      X += (X & 1) ? m1[u] : m0[u];
      ck[base + kLsEnd] = ldexp((double)X, e - 1075);
    }
  }
}

// pass D1: every window composes the maps of its chunks; a window with a direct chunk (the sum crosses a power of two
// in it, or it is the segment's first) is marked and left to the per-segment pass
__global__ __launch_bounds__(256) void k_long_wmap(const int64_t* __restrict__ seg_start, int64_t nseg, double* __restrict__ ck,
                                                   int64_t ck_len, long long GW, const PlanHeader* __restrict__ h) {
  if (h->n_long == 0) return;
  __shared__ long long w0[4], w1[4];
  __shared__ int mn[4];
  for (long long gw = blockIdx.x; gw < GW; gw += gridDim.x) {       // window slots, grid-strided (most are empty)
    const WinRef w = find_window(gw, seg_start, nseg, ck, ck_len);
    if (!w.ok) continue;
    const long long jt = w.j0 + (long long)threadIdx.x * kWinC;
    long long m0[kWinC], m1[kWinC];
    int first;
    load_chunk_maps(ck, w.slot0, jt, w.j1, kWinC, m0, m1, &first);
    const int fd = block_min(first, mn);
    long long a0 = 0, a1 = 0;
#pragma unroll
    for (int u = 0; u < kWinC; ++u) map_then(a0, a1, m0[u], m1[u], &a0, &a1);
    long long p0, p1, all0, all1;
    block_scan_maps(a0, a1, w0, w1, &p0, &p1, &all0, &all1);
    if (threadIdx.x == 0) {
      ck[w.rec + kLsWC0] = __longlong_as_double(all0);
      ck[w.rec + kLsWC1] = __longlong_as_double(all1);
      ck[w.rec + kLsWClean] = fd == kWin ? 1.0 : 0.0;
    }
    __syncthreads();                                                 // LDS scratch is reused by the next slot
  }
}

// pass D2: per long segment, exact running sum at every WINDOW start.  Runs of clean windows are resolved by a scan of
// their maps (256 windows per step); a marked window is walked chunk by chunk right here: runs of single-binade chunks
// by a scan of the chunk maps, a direct chunk stepped through with real float64 adds by one thread.  The chunks of
// marked windows get their exact start / end here, those of clean windows in pass D3.
__global__ __launch_bounds__(256) void k_long_stitch(const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                                                     int64_t nseg, double* __restrict__ ck, int64_t ck_len,
                                                     const PlanHeader* __restrict__ h) {
  if (h->n_long == 0) return;
  __shared__ long long w0[4], w1[4];
  __shared__ int mn[4];
  __shared__ double x_run;
  const int t = threadIdx.x;
  for (long long i = blockIdx.x; i < nseg; i += gridDim.x) {
    const long long start = seg_start[i], n = seg_start[i + 1] - start;
    if (!long_segment(n, start, i, ck, ck_len)) continue;
    const long long J = n / kLongChunk, NW = (J + kWin - 1) / kWin, slot0 = ck_slot0(start, i);
    const Ramp r = make_ramp(sp[i], sp[i + 1], n);
    if (t == 0) x_run = 0.0;
    __syncthreads();
    long long w = 0;
    while (w < NW) {
      // ---- a run of clean windows from w on (at most 256 per step)
      const long long wt = w + t;
      const long long rec = slot0 + (wt < NW ? wt : NW - 1) * kWin * kLongSlots;
      const bool live = wt < NW;
      const bool clean = ck[rec + kLsWClean] > 0.5;
      long long a0 = __double_as_longlong(ck[rec + kLsWC0]), a1 = __double_as_longlong(ck[rec + kLsWC1]);
      const int fdw = block_min(live && !clean ? t : 256, mn);
      long long Lw = fdw;
      if (Lw > NW - w) Lw = NW - w;
      if (t >= Lw) a0 = a1 = 0;
      long long p0, p1, all0, all1;
      block_scan_maps(a0, a1, w0, w1, &p0, &p1, &all0, &all1);
      const double x0 = x_run;
      if (t < Lw) ck[rec + kLsWStart] = apply_map(p0, p1, x0);
      __syncthreads();
      if (t == 0 && Lw > 0) x_run = apply_map(all0, all1, x0);
      __syncthreads();
      w += Lw;
      if (w >= NW || fdw == 256) continue;
      // ---- the marked window w: chunk by chunk
      long long j = w * kWin;
      const long long j_end = j + kWin < J ? j + kWin : J;
      // Direct chunks cluster (the sum doubles at chunks 1, 3, 7, 15, ..): after one, look at 256 chunks only (one per
      // thread); a search that finds none widens to the full 16 per thread again.
      int per = 1;
      while (j < j_end) {
        const long long jt = j + (long long)t * per;
        long long m0[kWinC], m1[kWinC];
        int first;
        load_chunk_maps(ck, slot0, jt, j_end, per, m0, m1, &first);
        const int fd = block_min(first, mn);
        long long L = fd < kWin ? fd : 256ll * per;                   // interior chunks in front of the first direct one
        if (L > j_end - j) L = j_end - j;
        long long cnt = L - (long long)t * per;
        cnt = cnt < 0 ? 0 : (cnt > per ? per : cnt);
        long long c0 = 0, c1 = 0;
#pragma unroll
        for (int u = 0; u < kWinC; ++u)
          if (u < cnt) map_then(c0, c1, m0[u], m1[u], &c0, &c1);
        block_scan_maps(c0, c1, w0, w1, &p0, &p1, &all0, &all1);
        const double xs = x_run;
        write_chunk_bounds(ck, slot0, jt, cnt, xs, p0, p1, m0, m1);
        const bool has_direct = j + L < j_end && fd < kWin;
        __syncthreads();
        if (t == 0) {
          double x = L > 0 ? apply_map(all0, all1, xs) : xs;
          if (has_direct) {                                           // the direct chunk that ended the run
            const long long jd = j + L, bd = slot0 + jd * kLongSlots;
            const long long k0 = jd * kLongChunk, k1 = (jd == J - 1) ? n : k0 + kLongChunk;
            ck[bd + kLsStart] = x;
            double c = x;
            for (long long k = k0; k < k1; ++k) c = c + ramp_recip((double)k, r);
            ck[bd + kLsEnd] = c;
            x = c;
          }
          x_run = x;
        }
        __syncthreads();
        j += L + (has_direct ? 1 : 0);
        per = has_direct ? 1 : kWinC;
      }
      w += 1;
    }
  }
}

// pass D3: the chunks of every clean window get their exact start / end from the window's start
__global__ __launch_bounds__(256) void k_long_wapply(const int64_t* __restrict__ seg_start, int64_t nseg, double* __restrict__ ck,
                                                     int64_t ck_len, long long GW, const PlanHeader* __restrict__ h) {
  if (h->n_long == 0) return;
  __shared__ long long w0[4], w1[4];
  for (long long gw = blockIdx.x; gw < GW; gw += gridDim.x) {       // window slots, grid-strided (most are empty)
    const WinRef w = find_window(gw, seg_start, nseg, ck, ck_len);
    if (!w.ok) continue;
    if (!(ck[w.rec + kLsWClean] > 0.5)) continue;                         // uniform over the workgroup
    const double x0 = ck[w.rec + kLsWStart];
    const long long jt = w.j0 + (long long)threadIdx.x * kWinC;
    long long m0[kWinC], m1[kWinC];
    int first;
    load_chunk_maps(ck, w.slot0, jt, w.j1, kWinC, m0, m1, &first);
    long long a0 = 0, a1 = 0;
#pragma unroll
    for (int u = 0; u < kWinC; ++u) map_then(a0, a1, m0[u], m1[u], &a0, &a1);
    long long p0, p1, all0, all1;
    block_scan_maps(a0, a1, w0, w1, &p0, &p1, &all0, &all1);
    long long cnt = w.j1 - jt;
    cnt = cnt < 0 ? 0 : (cnt > kWinC ? kWinC : cnt);
    write_chunk_bounds(ck, w.slot0, jt, cnt, x0, p0, p1, m0, m1);
    __syncthreads();                                                 // LDS scratch is reused by the next slot
  }
}

// pass E: every chunk recomputed sequentially from its exact start: checkpoints out, S_i from the last chunk, and the
// proof -- the chunk must land bit-exactly on the value the scan promised to the next chunk.
__global__ __launch_bounds__(256) void k_long_final(const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                                                    int64_t nseg, double* __restrict__ S, double* __restrict__ ck,
                                                    int64_t ck_len, long long G, PlanHeader* __restrict__ h) {
  if (h->n_long == 0) return;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const ChunkRef c = find_chunk(g, seg_start, nseg, ck, ck_len);
  if (!c.ok) return;
  const double x_start = ck[c.base + kLsStart], x_end = ck[c.base + kLsEnd];    // read before this chunk's slots are reused
  const Ramp r = make_ramp(sp[c.i], sp[c.i + 1], c.n);
  double x = x_start;
  long long k = c.k0;
  for (; k + kCk <= c.k1; k += kCk) {
    double rr[kCk];
    const double a0 = (double)k;
#pragma unroll
    for (int u = 0; u < kCk; ++u) rr[u] = ramp_recip(a0 + (double)u, r);
#pragma unroll
    for (int u = 0; u < kCk; ++u) x = x + rr[u];
    if (k + kCk < c.n) ck[c.slot0 + k / kCk + 1] = x;               // cumsum after step k + kCk - 1 (steps follow)
  }
  for (; k < c.k1; ++k) x = x + ramp_recip((double)k, r);
  if (__double_as_longlong(x) != __double_as_longlong(x_end)) atomicOr(&h->flags, kFlagVerify);
  if (c.j == c.J - 1) S[c.i] = x;
}

// force_host == 2 (tests): a checkpoint verification failure as the chunked long-segment cumsum would report it
__global__ void k_inject_verify_fault(PlanHeader* __restrict__ h) { atomicOr(&h->flags, kFlagVerify); }

// after k_tile_seg: publish checkpoint validity in the header (device side, so the host needs one read-back)
__global__ void k_publish_ck(PlanHeader* __restrict__ h, int64_t ck_len) {
  h->ck_len = ck_len;
  // any flag still standing here (kFlagVerify from the chunked long-segment cumsum included) means some checkpoint
  // was not verified: the fused resampler must not regenerate positions from them
  // (kFlagCapAmbiguous only concerns the reference's buffer bound, which the host settles afterwards)
  h->ck_valid = (h->flags & ~kFlagCapAmbiguous) ? 0 : 1;
  h->flags &= ~kFlagCkOverflow;
}

// tile_seg[t] = segment that contains output t * kSincTileOutputs (tiles of the fused resampler)
// Thread x serves two roles: segment x checks that its checkpoints fit and writes its SegFast record, tile x looks its
// segment up (upper bound over seg_start: a tile per thread, not a segment per thread -- one segment can cover
// 10^5..10^6 tiles).
// closed-form placement record of segment x (sinc.hip place_fast).  `fast` bounds what the closed form leaves out: the
// fourth-order remainder of the <= kCk-term reciprocal sum is < 400 step^4 / smin^5, kept below 2e-10.
__device__ __forceinline__ SegFast seg_fast_record(const double* __restrict__ sp, const double* __restrict__ seg_off, long long x,
                                                   long long n) {
  const double s0 = sp[x], s1 = sp[x + 1], off = seg_off[x];
  SegFast f;
  const bool off_ok = fabs(off) < 4.0e18;            // also false for NaN
  const double ro = off_ok ? rint(off) : 0.0;
  f.foff = off_ok ? off - ro : 0.0;
  f.A = (long long)ro;
  f.n = n < 0x7fffffffll ? (int)n : 0x7fffffff;
  f.step = n >= 2 ? (s1 - s0) / (double)(n - 1) : 0.0;
  const double smin = s0 < s1 ? s0 : s1, smax = s0 < s1 ? s1 : s0;
  // |step| <= 6e-4 smin^(5/4), compared as fourth powers (two float64 square roots per segment were a third of this function)
  const double st2 = f.step * f.step, sm2 = smin * smin;
  f.fast = off_ok && n >= 2 && n < 0x7fffffffll && smin >= 0.0625 && smax <= 64.0 && st2 * st2 <= 1.296e-13 * sm2 * sm2 * smin;
  // level 2 (BlockRec with the cubic term): over the 32 centred steps of a block (|d| <= 16, up to 33 with the steps behind
  // the checkpoint) the quartic term of the reciprocal sum, z^3 d^4 / 4 with z = step / speed, stays < 1e-8 samples
  // level 3 (the block quadratic alone): the cubic term z^2 d^3 / 3 <= 1365 z^2 stays < 1e-8
  if (f.fast && fabs(f.step) <= 5.0e-5 * smin) f.fast = fabs(f.step) <= 2.7e-6 * smin ? 3 : 2;
  return f;
}

// entry x of the tile tables: tile x (x < n_tiles: its first output is `sample`) or the extra entry [n_tiles] (the segment
// that holds the LAST output); lo = the segment that contains `sample`
__device__ __forceinline__ void tile_entry(const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                                           const double* __restrict__ seg_off, int64_t nseg, const double* __restrict__ ck,
                                           int64_t ck_len, int64_t* __restrict__ tile_seg, long long* __restrict__ tile_st,
                                           TileHdr* __restrict__ hdr, const long long x, const long long n_tiles,
                                           const long long sample, const long long lo, const bool lazy) {
  tile_seg[x] = lo;
  // boundaries the tile's blocks may meet, for k_block_rec's lookup
  for (int q = 0; q < kTileStarts; ++q) tile_st[x * kTileStarts + q] = lo + q <= nseg ? seg_start[lo + q] : LLONG_MAX;
  // tile header: the anchor every window centre of the tile is relative to -- an even integer within ~8 samples of the
  // tile's first position (the checkpoint below it; block records keep a 16-bit offset from it)
  if (x < n_tiles) {
    const long long start = seg_start[lo], k = sample - start, b = k >> 3;
    static_assert(kCk == 8, "k >> 3");
    const double off = seg_off[lo];
    const bool off_ok = fabs(off) < 4.0e18;
    const double ro = off_ok ? rint(off) : 0.0;
    const long long slot = ck_slot0(start, lo) + b;
    // (lazy plans have no checkpoints: the closed-form position of the tile's first output serves as well)
    const double ckv = lazy ? (lazy_segment_ok(seg_start[lo + 1] - start, sp[lo], sp[lo + 1])
                                   ? lazy_prefix(sp[lo], (sp[lo + 1] - sp[lo]) / (double)(seg_start[lo + 1] - start - 1), (double)(k + 1))
                                   : 0.0)
                            : ((b && slot < ck_len) ? ck[slot] : 0.0);
    const double rel = (off_ok ? off - ro : 0.0) + ckv;
    const bool ok = off_ok && fabs(rel) < 1.0e15 && ro > -0x1p61 && ro < 0x1p61;
    TileHdr hd;
    hd.anchor = ok ? ((long long)ro + (long long)rint(rel)) & ~1ll : 0ll;
    hd.c_last = 0;
    hd.iT = lo;
    hd.mn_rel = 0;
    // bit 1: some segment under the tile (or the one behind it: the last output's period reaches there) touches
    // speed >= 1, i.e. the tile may hold fc = 1 outputs.  A hint only: K_sinc's matrix-core path is taken by workgroups
    // whose tile carries it, everything else computes the same numbers on the vector path.
    bool may_unity = false, may_slow = false;
    for (long long q = lo; q < nseg && seg_start[q] <= sample + kSincTileOutputs; ++q) {
      may_unity = may_unity || !(sp[q] < kUnityHintBelow) || !(sp[q + 1] < kUnityHintBelow);
      may_slow = may_slow || !(sp[q] >= kSlowHintBelow) || !(sp[q + 1] >= kSlowHintBelow);
    }
    hd.flags = (ok ? 0 : 1) | (may_unity ? kTileMayUnity : 0) | (may_slow ? kTileMaySlow : 0) | (lazy ? kTileLazy : 0);
    hdr[x] = hd;
  }
}

__device__ __forceinline__ void tile_seg_body(const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                           const double* __restrict__ seg_off, int64_t nseg, const double* __restrict__ ck, int64_t ck_len,
                           int64_t max_tiles, int64_t* __restrict__ tile_seg, SegFast* __restrict__ seg_fast,
                           long long* __restrict__ tile_st, TileHdr* __restrict__ hdr, PlanHeader* __restrict__ h, const int64_t x,
                           const bool lazy = false) {
  const long long len_out = h->len_out;                // written by k_trim / the host path earlier on this stream
  const long long n_tiles = (len_out + kSincTileOutputs - 1) / kSincTileOutputs;
  if (n_tiles + 1 > max_tiles) {
    if (x == 0 && len_out > 0) atomicOr(&h->flags, kFlagCkOverflow);
    return;
  }
  if (x < nseg) {
    const long long a = seg_start[x], b = seg_start[x + 1];
    if (!lazy && b > a && a < len_out && ck_slot0(a, x) + (b - a + kCk - 1) / kCk > ck_len)
      atomicOr(&h->flags, kFlagCkOverflow);             // a needed segment has no checkpoints: fused path refused
    seg_fast[x] = seg_fast_record(sp, seg_off, x, b - a);
  }
  if (x > n_tiles || len_out <= 0) return;
  // entry [n_tiles] is extra: the segment that holds the LAST output
  const long long sample = x < n_tiles ? x * kSincTileOutputs : len_out - 1;
  long long lo = 0, hi = nseg;                         // last i with seg_start[i] <= sample (its successor is larger)
  while (lo < hi) {
    const long long mid = (lo + hi + 1) >> 1;
    if (seg_start[mid] <= sample) lo = mid; else hi = mid - 1;
  }
  if (lo < nseg) tile_entry(sp, seg_start, seg_off, nseg, ck, ck_len, tile_seg, tile_st, hdr, x, n_tiles, sample, lo, lazy);
}

__global__ void k_tile_seg(const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                           const double* __restrict__ seg_off, int64_t nseg, const double* __restrict__ ck, int64_t ck_len,
                           int64_t max_tiles, int64_t* __restrict__ tile_seg, SegFast* __restrict__ seg_fast,
                           long long* __restrict__ tile_st, TileHdr* __restrict__ hdr, PlanHeader* __restrict__ h) {
  tile_seg_body(sp, seg_start, seg_off, nseg, ck, ck_len, max_tiles, tile_seg, seg_fast, tile_st, hdr, h,
                (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// ... with k_publish_ck's header update as the epilogue of the last block to finish (r04)
__global__ void k_tile_seg_publish(const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                                   const double* __restrict__ seg_off, int64_t nseg, const double* __restrict__ ck, int64_t ck_len,
                                   int64_t max_tiles, int64_t* __restrict__ tile_seg, SegFast* __restrict__ seg_fast,
                                   long long* __restrict__ tile_st, TileHdr* __restrict__ hdr, PlanHeader* __restrict__ h,
                                   int64_t n_items, int lazy) {
  __shared__ int is_last;
  if (h->lazy_fail) return;                             // the plan is being made again (eager, or with the chunked cumsum)
  for (int64_t x0 = (int64_t)blockIdx.x * blockDim.x; x0 < n_items; x0 += (int64_t)gridDim.x * blockDim.x)
    tile_seg_body(sp, seg_start, seg_off, nseg, ck, ck_len, max_tiles, tile_seg, seg_fast, tile_st, hdr, h, x0 + threadIdx.x,
                  lazy != 0);
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) is_last = atomicAdd(&h->pad3, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!is_last || threadIdx.x != 0) return;
  __threadfence();
  h->pad3 = 0;
  const int fl = __hip_atomic_load(&h->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  h->ck_len = ck_len;
  h->ck_valid = (fl & ~kFlagCapAmbiguous) ? 0 : (lazy ? 2 : 1);     // 2: K_sinc may run, but there are no checkpoints to fill from
  h->flags = fl & ~kFlagCkOverflow;
}

// Lazy plans: a thread per SEGMENT writes its SegFast record and the entries of the tiles whose first output it holds (n <= kLazyMaxN
// <= one tile: at most two of them) -- no bisection per tile (22 dependent loads each), no grid cap, no epilogue: k_trim_lazy has
// published ck_valid already.
__global__ __launch_bounds__(256) void k_tile_seg_lazy(const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                                                       const double* __restrict__ seg_off, int64_t nseg,
                                                       int64_t* __restrict__ tile_seg, SegFast* __restrict__ seg_fast,
                                                       long long* __restrict__ tile_st, TileHdr* __restrict__ hdr,
                                                       const PlanHeader* __restrict__ h) {
  if (h->ck_valid != 2) return;
  const long long x = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= nseg) return;
  const long long len_out = h->len_out;
  const long long n_tiles = (len_out + kSincTileOutputs - 1) / kSincTileOutputs;
  const long long a = seg_start[x], b = seg_start[x + 1];
  seg_fast[x] = seg_fast_record(sp, seg_off, x, b - a);
  if (b <= a || a >= len_out) return;
  for (long long T = (a + kSincTileOutputs - 1) / kSincTileOutputs; T * kSincTileOutputs < b && T < n_tiles; ++T)
    tile_entry(sp, seg_start, seg_off, nseg, nullptr, 0, tile_seg, tile_st, hdr, T, n_tiles, T * kSincTileOutputs, x, true);
  if (len_out - 1 < b) tile_entry(sp, seg_start, seg_off, nseg, nullptr, 0, tile_seg, tile_st, hdr, n_tiles, n_tiles, len_out - 1, x, true);
}

// ---- block records of the fused resampler (BlockRec, pos_plan.h) ---------------------------------------------------
// One lane per block of kRec = 32 consecutive outputs: finds the block's segment (the tile's boundary table, then a
// bisection), takes the cumsum checkpoint below the block's first output, adds the closed-form sum of the <= 7
// reciprocals in between, and expands the 32 positions of the block as a polynomial in the centred u' = u - 16.  A block
// that contains a segment boundary gets a second piece for the segment that starts inside it (k_block_rec2).  Positions
// are kept relative to rint(seg_off), so every float64 operand is small and the polynomial constant is accurate to
// ~1e-12.  ONE reciprocal per piece: every speed the piece needs is at most 24 steps from the block centre, so its
// reciprocal follows from the centre's by a short series in z d (|z d| < 1.3e-3 where the model applies).
struct BlockPoly {
  double a0;       // position of the centre output (u' = 0) relative to rint(seg_off) of the piece's segment
  double a1m1;     // first-order coefficient - 1
  double a2;
};
__device__ __forceinline__ double recip_nr(double b) {       // 1/b to ~1e-15 relative (no IEEE corner cases needed)
  double x = __builtin_amdgcn_rcp(b);
  x = __builtin_fma(x, __builtin_fma(-b, x, 1.0), x);
  return __builtin_fma(x, __builtin_fma(-b, x, 1.0), x);
}
// p(u') = base + sum_{d = d0}^{u'} r(d),  r(d) = 1 / (sc + step d) = rc (1 - z d + z^2 d^2 - ..),  rc = 1/sc, z = rc step:
//   sum 1 = u' - d0 + 1,  sum d = (u'^2 + u' - d0^2 + d0) / 2,  sum d^2 = (2 u'^3 + 3 u'^2 + u') / 6 - S2(d0 - 1)
// The pure cubic part rc z^2 u'^3 / 3 is what K_sinc adds itself where the record says `cubic` (it derives the
// coefficient from e2); its quadratic, linear and constant companions are folded in here.
__device__ __forceinline__ BlockPoly piece_poly(double base, double rc, double z, int d0) {
  const double rp = -(rc * z), cz = rc * z * z, dd0 = (double)d0;
  const double s2m = (dd0 - 1.0) * dd0 * (2.0 * dd0 - 1.0) * (1.0 / 6.0);
  BlockPoly q;
  q.a0 = ((base + rc * (1.0 - dd0)) + 0.5 * rp * (dd0 - dd0 * dd0)) - cz * s2m;
  q.a1m1 = ((rc - 1.0) + 0.5 * rp) + cz * (1.0 / 6.0);
  q.a2 = 0.5 * rp + 0.5 * cz;
  return q;
}
// first piece of a block: k = step index of the block's output u = 0 in its segment; uk = steps between the checkpoint
// and k; ckv = checkpoint (cumsum before step k - uk)
__device__ __forceinline__ BlockPoly block_poly(double foff, double step, double s0, double k, int uk, double ckv, bool fast) {
  const double rc = recip_nr(__builtin_fma(step, k + 16.0, s0));    // reciprocal speed at the block centre
  const double z = rc * step;
  double cprev = ckv;                                               // cumsum before step k
  if (uk) {
    if (fast) {                                                     // uk steps centred (uk + 1)/2 + 16 before the block centre
      const double x = z * (-0.5 * (double)(uk + 1) - 16.0);
      const double r = rc * __builtin_fma(x, __builtin_fma(x, -x, x) - 1.0, 1.0);   // rc (1 - x + x^2 - x^3)
      cprev += r * (double)uk * __builtin_fma(z * z, (double)(uk * uk - 1) * (1.0 / 12.0), 1.0);
    } else {
      for (int v = 0; v < uk; ++v) cprev += recip_nr(__builtin_fma(step, k - (double)(uk - v), s0));
    }
  }
  return piece_poly(foff + cprev, rc, z, -16);
}
__device__ __forceinline__ bool irel_ok(long long rel) { return rel > -32000 && rel < 32000; }

// second piece of the blocks that contain a segment start: one lane per SEGMENT (m lanes instead of a divergent branch in
// every wave of k_block_rec).  Runs after k_block_rec: a piece outside the 16-bit offset range flags its block slow1.
__global__ __launch_bounds__(256) void k_block_rec2(const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                                                    int64_t nseg, const SegFast* __restrict__ seg_fast,
                                                    const TileHdr* __restrict__ hdr, BlockRec* __restrict__ rec,
                                                    BlockRec2* __restrict__ rec2, const PlanHeader* __restrict__ h) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 1 || i >= nseg || !h->ck_valid) return;
  const long long start = seg_start[i];
  const int us = (int)(start & (kRec - 1));
  if (us == 0 || start >= h->len_out) return;
  const SegFast s1 = seg_fast[i];
  // the cumsum restarts at 0 with the segment's step 0 = the block's u = us, i.e. d0 = us - 16 steps from the centre
  const double rc = recip_nr(__builtin_fma(s1.step, (double)(16 - us), sp[i]));
  const BlockPoly b1 = piece_poly(s1.foff, rc, rc * s1.step, us - 16);
  const double r1 = rint(b1.a0);
  const long long rel = s1.A + (long long)r1 - hdr[start / kSincTileOutputs].anchor;
  BlockRec2 o2;
  o2.w0 = (unsigned)((int)rel << 16);
  o2.F = (float)(b1.a0 - r1);
  o2.e1 = (float)b1.a1m1;
  o2.e2 = (float)b1.a2;
  rec2[start >> kRecShift] = o2;
  if (!(fabs(b1.a0) < 1.0e9 && irel_ok(rel))) atomicOr(&rec[start >> kRecShift].w0, kRecSlow1);
}

__global__ __launch_bounds__(256) void k_block_rec(const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                                                   int64_t nseg, const double* __restrict__ ck,
                                                   const int64_t* __restrict__ tile_seg, const long long* __restrict__ tile_st,
                                                   const SegFast* __restrict__ seg_fast, const TileHdr* __restrict__ hdr,
                                                   BlockRec* __restrict__ rec, const PlanHeader* __restrict__ h) {
  const long long len_out = h->len_out;
  if (!h->ck_valid) return;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long jb = g * kRec;
  if (jb >= len_out) return;
  // The tile's first segment and the starts of the six segments from it on: everything below is 32-bit arithmetic
  // relative to the tile's first output; the lane counts the boundaries at or below its block.  Tiles with more segments
  // (shorter than ~170 outputs) bisect the rest in memory.  (Lanes of one tile read the same table: two tiles per wave.)
  const long long T = g / kBlocksPerTile;
  const long long j0 = T * kSincTileOutputs;
  const long long iT = tile_seg[T];
  int Sr[kTileStarts];                                        // segment starts relative to j0 (first one may be negative)
#pragma unroll
  for (int q = 0; q < kTileStarts; ++q) {
    const long long d = tile_st[T * kTileStarts + q] - j0;
    Sr[q] = d < -0x40000000ll ? -0x40000000 : (d > 0x40000000ll ? 0x40000000 : (int)d);
  }
  const int jr = (int)(jb - j0);                              // 0 .. 992
  int si = 0;
#pragma unroll
  for (int q = 1; q < kTileStarts; ++q) si += jr >= Sr[q];
  long long i = iT + si;
  long long k;                                                // step index of the block's first output in its segment
  double kd;                                                  // the same as a double (from 32 bits where the table serves)
  int rem;                                                    // outputs of segment i from jb on (>= 1), clamped to kRec + 1
  long long slot0;                                            // checkpoint slot of the segment's step 0
  if (si < kTileStarts - 1 && Sr[0] > -0x40000000) {
    int sr = Sr[0], nr = Sr[1];
#pragma unroll
    for (int q = 1; q < kTileStarts - 1; ++q) {
      if (si == q) {
        sr = Sr[q];
        nr = Sr[q + 1];
      }
    }
    k = jr - sr;
    kd = (double)(jr - sr);
    const int d = nr - jr;
    rem = d < kRec + 1 ? d : kRec + 1;
    slot0 = ck_slot0(j0 + sr, i);
  } else {                                                    // beyond the table (or a segment of > 10^9 outputs)
    long long hi = tile_seg[T + 1];                           // entry [n_tiles] is the segment of the last output
    while (i < hi) {
      const long long mid = (i + hi + 1) >> 1;
      if (seg_start[mid] <= jb) i = mid; else hi = mid - 1;
    }
    const long long start = seg_start[i], d = seg_start[i + 1] - jb;
    k = jb - start;
    kd = (double)k;
    rem = d < kRec + 1 ? (int)d : kRec + 1;
    slot0 = ck_slot0(start, i);
  }
  // everything the block needs, in one round of independent loads (the next segment's record speculatively: boundary
  // blocks need its flags)
  const long long i1 = i + 1 < nseg ? i + 1 : i;
  const SegFast sf = seg_fast[i];
  const SegFast s1 = seg_fast[i1];
  const double sp0 = sp[i], sp1 = sp[i1];
  const TileHdr hd = hdr[T];
  const long long b = k >> 3;
  const int uk = (int)(k & 7);
  static_assert(kCk == 8, "k >> 3");
  BlockPoly q0;
  if (hd.flags & kTileLazy) {        // no checkpoints: the cumsum in front of step k in closed form from the segment's first step
    const double rc = recip_nr(__builtin_fma(sf.step, kd + 16.0, sp0));
    q0 = piece_poly(sf.foff + lazy_prefix(sp0, sf.step, kd), rc, rc * sf.step, -16);
  } else {
    const double ckv = b ? ck[slot0 + b] : 0.0;
    q0 = block_poly(sf.foff, sf.step, sp0, kd, uk, ckv, sf.fast != 0);
  }
  const double r0 = rint(q0.a0);
  const long long rel0 = sf.A + (long long)(int)r0 - hd.anchor;   // |a0| >= 1e9 saturates: such a block is flagged and never placed from Irel
  const bool range0 = fabs(q0.a0) < 1.0e9 && sf.A > -(1ll << 61) && sf.A < (1ll << 61) && irel_ok(rel0) && !(hd.flags & 1);
  const unsigned ustar = rem < kRec ? (unsigned)rem : (unsigned)kRec;
  // the file's last output reuses the previous period: slow path
  const long long ul = j0 + kSincTileOutputs >= len_out ? len_out - 1 - jb : -1;
  const unsigned slow0 = !(sf.fast >= 2 && range0 && fabs(q0.a1m1) <= 0.03125) || (ul >= 0 && ul < (long long)ustar);
  unsigned slow1 = 0u, end1 = 0u, cubic = sf.fast == 2;
  if (ustar < (unsigned)kRec) {                               // a segment starts at u = ustar (its piece: k_block_rec2)
    const int need = kRec - (int)ustar;                       // outputs of the block that fall to segment i + 1
    end1 = s1.n == need;
    // the second piece has its own polynomial (r02 let its curvature ride on the first piece's, which the benchmark's
    // own curve missed at 72 % of its segment boundaries: 63 % of K_sinc's waves then ran the float64 redo path for a
    // handful of lanes, tools/rec_stats.py); its first-order term must be inside the float32 budget like the first's
    slow1 = !(i + 1 < nseg && s1.fast >= 2 && s1.n >= need && s1.A > -(1ll << 61) && s1.A < (1ll << 61) && sp1 >= 0.971 &&
              sp1 <= 1.031) ||
            (ul >= (long long)ustar && ul < kRec);
    cubic |= s1.fast == 2;
  }
  BlockRec o;
  const bool e0 = (ustar < (unsigned)kRec) || rem == kRec;
  // bits 10-15 (streaming kernel, sinc2.hip): u of the block's output that is the last one of its segment (63: none).  A
  // block with two of them (a segment of < 32 outputs that ends with the block) is left to the block kernel.
  if (end1) slow1 = 1u;
  o.w0 = ((unsigned)((int)rel0 << 16)) | (ustar - 1u) | (e0 ? kRecE0 : 0u) | (end1 ? kRecE1 : 0u) | (slow0 ? kRecSlow0 : 0u) |
         (slow1 ? kRecSlow1 : 0u) | (cubic ? kRecCubic : 0u) | ((e0 ? ustar - 1u : 63u) << kRecLastShift);
  o.F = (float)(q0.a0 - r0);
  o.e1 = (float)q0.a1m1;
  o.e2 = (float)q0.a2;
  rec[g] = o;
}

// Lazy plans: the records of the blocks whose first output lies in 64 consecutive segments, by ONE WAVE -- the segments' data
// staged in LDS, then a lane per block (the blocks of consecutive segments are consecutive: coalesced 16-byte stores), its
// segment found by a 6-step search of the staged starts.  The cumsum in front of a block in closed form from the segment's
// first step (no checkpoint, no dependent global load), and the second piece of the block a segment ends in (k_block_rec2's
// job) by the same lane.  Same records as k_block_rec + k_block_rec2 up to the closed form's 1e-12; flags identical.
// (A thread per segment looping over its ~8 blocks: 486 us for the 60-min curve -- 128-byte strides between the lanes of a
// store; a thread per block with the tile tables' lookup as in k_block_rec + k_block_rec2: 230 + 130 us.)
constexpr int kRecLazySegs = 64;
#ifndef PAR_REC_EXP
#define PAR_REC_EXP 0            // timing builds of k_block_rec_lazy, never shipped: 1 no record stores, 2 no tile-header loads
#endif
// (records need ~1e-10, not numpy's bits: this kernel's arithmetic may contract into FMAs, unlike the rest of the file)
__device__ __forceinline__ double recip_nr_c(double b) {
#pragma clang fp contract(fast)
  double x = __builtin_amdgcn_rcp(b);
  x = x + x * (1.0 - b * x);
  return x + x * (1.0 - b * x);
}
__device__ __forceinline__ BlockPoly piece_poly_c(double base, double rc, double z, int d0) {
#pragma clang fp contract(fast)
  const double rp = -(rc * z), cz = rc * z * z, dd0 = (double)d0;
  const double s2m = (dd0 - 1.0) * dd0 * (2.0 * dd0 - 1.0) * (1.0 / 6.0);
  BlockPoly q;
  q.a0 = ((base + rc * (1.0 - dd0)) + 0.5 * rp * (dd0 - dd0 * dd0)) - cz * s2m;
  q.a1m1 = ((rc - 1.0) + 0.5 * rp) + cz * (1.0 / 6.0);
  q.a2 = 0.5 * rp + 0.5 * cz;
  return q;
}
__device__ __forceinline__ double lazy_prefix_nr(double s0, double step, double K) {     // lazy_prefix to ~1e-15 relative
#pragma clang fp contract(fast)
  const double rc = recip_nr_c(__builtin_fma(step, 0.5 * (K - 1.0), s0));
  const double z = rc * step, t = z * z, K2 = K * K;
  return K * rc * (1.0 + t * (K2 - 1.0) * (1.0 / 12.0) * (1.0 + t * (3.0 * K2 - 7.0) * 0.05));
}
__global__ __launch_bounds__(256) void k_block_rec_lazy(const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                                                        int64_t nseg, const SegFast* __restrict__ seg_fast,
                                                        const TileHdr* __restrict__ hdr, BlockRec* __restrict__ rec,
                                                        BlockRec2* __restrict__ rec2, const PlanHeader* __restrict__ h) {
  __shared__ SegFast s_sf[4][kRecLazySegs + 1];
  __shared__ double s_sp[4][kRecLazySegs + 2];
  __shared__ int s_a[4][kRecLazySegs + 1];                       // segment starts relative to the wave's first one
  __shared__ unsigned char s_fl[4][kRecLazySegs];                // what a segment's END contributes to the block it ends in (bits below)
  __shared__ unsigned char s_seg[4][kRecLazySegs * (kLazyMaxN / kRec)];   // block (relative to the wave's first) -> staged segment
  static_assert(kLazyMaxN % kRec == 0 && kRecLazySegs <= 256, "a segment owns <= kLazyMaxN / kRec block starts; indices fit a byte");
  constexpr unsigned kF_Slow1 = 1u, kF_End1 = 2u, kF_Cubic1 = 4u;
  if (h->ck_valid != 2) return;
  const int l = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
  const long long i0 = ((long long)blockIdx.x * 4 + w) * kRecLazySegs;
  if (i0 >= nseg) return;
  const long long len_out = h->len_out;
  const long long a0 = seg_start[i0];
  if (a0 >= len_out) return;
  const int nsw = (int)(nseg - i0 < kRecLazySegs ? nseg - i0 : kRecLazySegs);      // segments of this wave
  const long long G0 = (a0 + kRec - 1) >> kRecShift;
  // ---- a lane per SEGMENT: stage it, name the blocks whose first output it holds, and do the second piece of the block it
  // ends in (k_block_rec2's job) -- once per segment instead of in a branch every lane of the block loop walks through
  {
    const long long il = i0 + l < nseg ? i0 + l : nseg;           // entries past the curve repeat the end (never selected)
    const long long al = seg_start[il];
    const long long d = al - a0;
    s_a[w][l] = d < 0x7fffffffll ? (int)d : 0x7fffffff;
    const SegFast sfl = seg_fast[il < nseg ? il : nseg - 1];
    s_sf[w][l] = sfl;
    const double spl = sp[il];
    s_sp[w][l] = spl;
    if (l == 0) {
      const long long ie = i0 + kRecLazySegs < nseg ? i0 + kRecLazySegs : nseg;
      const long long de = seg_start[ie] - a0;
      s_a[w][kRecLazySegs] = de < 0x7fffffffll ? (int)de : 0x7fffffff;
      s_sf[w][kRecLazySegs] = seg_fast[ie < nseg ? ie : nseg - 1];
      s_sp[w][kRecLazySegs] = sp[ie];
      s_sp[w][kRecLazySegs + 1] = sp[ie + 1 <= nseg ? ie + 1 : nseg];
    }
    unsigned fl = 0u;
    if (l < nsw) {
      const long long bl = seg_start[il + 1];
      // blocks g with al <= 32 g < min(bl, len_out): this segment holds their output u = 0
      const long long lim_l = bl < len_out ? bl : len_out;
      const long long g_lo = (al + kRec - 1) >> kRecShift, g_hi = (lim_l + kRec - 1) >> kRecShift;
      for (long long g = g_lo; g < g_hi; ++g) s_seg[w][g - G0] = (unsigned char)l;
      // the block this segment ENDS in (segment il + 1 starts at its u = us): that segment's piece of it
      const int us = (int)(bl & (kRec - 1));
      const bool has_next = il + 1 < nseg;
      if (us != 0 && bl > al) {
        const SegFast s1 = seg_fast[has_next ? il + 1 : il];
        const double sp1 = sp[has_next ? il + 1 : il];
        const int need = kRec - us;
        if (s1.n == need) fl |= kF_End1 | kF_Slow1;
        if (!(has_next && s1.fast >= 2 && s1.n >= need && s1.A > -(1ll << 61) && s1.A < (1ll << 61) && sp1 >= 0.971 && sp1 <= 1.031))
          fl |= kF_Slow1;
        if (s1.fast == 2) fl |= kF_Cubic1;
        if (has_next && bl < len_out) {
          const long long gb = bl >> kRecShift;
          const long long anchor = hdr[(gb << kRecShift) / kSincTileOutputs].anchor;
          const double rc1 = recip_nr_c(__builtin_fma(s1.step, (double)(16 - us), sp1));
          const BlockPoly b1 = piece_poly_c(s1.foff, rc1, rc1 * s1.step, us - 16);
          const double r1 = rint(b1.a0);
          const long long rel = s1.A + (long long)r1 - anchor;
          BlockRec2 o2;
          o2.w0 = (unsigned)((int)rel << 16);
          o2.F = (float)(b1.a0 - r1);
          o2.e1 = (float)b1.a1m1;
          o2.e2 = (float)b1.a2;
          rec2[gb] = o2;
          if (!(fabs(b1.a0) < 1.0e9 && irel_ok(rel))) fl |= kF_Slow1;
        }
      }
      s_fl[w][l] = (unsigned char)fl;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const long long b_end = a0 + s_a[w][nsw];                      // first output behind the wave's segments
  const long long lim = b_end < len_out ? b_end : len_out;
  const int nblk = (int)(((lim + kRec - 1) >> kRecShift) - G0);  // blocks of this wave
  const int jr0 = (int)((G0 << kRecShift) - a0);                 // first block's first output, relative to a0 (0 .. 31)
  const long long last_tile0 = ((len_out - 1) / kSincTileOutputs) * kSincTileOutputs;   // first output of the file's last tile
  long long Tc = -1;
  long long anchor = 0;
  int hflags = 0;
  // ---- a lane per BLOCK: the first piece and the flags (32-bit arithmetic relative to the wave's first output)
  for (int q = l; q < nblk; q += kWave) {
    const int lo = s_seg[w][q];
    const int jr = jr0 + (q << kRecShift);                        // the block's first output, relative to a0
    const long long g = G0 + q, jb = g << kRecShift;
    const int ar = s_a[w][lo], br = s_a[w][lo + 1];
    const SegFast sf = s_sf[w][lo];
    const double sp0 = s_sp[w][lo];
    const long long T = jb / kSincTileOutputs;
#if PAR_REC_EXP & 2
    anchor = jb;
#else
    if (T != Tc) {
      const TileHdr hd = hdr[T];
      anchor = hd.anchor;
      hflags = hd.flags;
      Tc = T;
    }
#endif
    const double kd = (double)(jr - ar);
    const int dseg = br - jr;
    const int rem = dseg < kRec + 1 ? dseg : kRec + 1;
    const double rc = recip_nr_c(__builtin_fma(sf.step, kd + 16.0, sp0));
    const BlockPoly q0 = piece_poly_c(sf.foff + lazy_prefix_nr(sp0, sf.step, kd), rc, rc * sf.step, -16);
    const double r0 = rint(q0.a0);
    const long long rel0 = sf.A + (long long)(int)r0 - anchor;
    const bool range0 = fabs(q0.a0) < 1.0e9 && sf.A > -(1ll << 61) && sf.A < (1ll << 61) && irel_ok(rel0) && !(hflags & 1);
    const unsigned ustar = rem < kRec ? (unsigned)rem : (unsigned)kRec;
    // the file's last output takes the slow path: its u in this block, or -1 (only blocks of the last tile look)
    const int ul = jb >= last_tile0 ? (int)(len_out - 1 - jb) : -1;
    const unsigned slow0 = !(sf.fast >= 2 && range0 && fabs(q0.a1m1) <= 0.03125) || (ul >= 0 && ul < (int)ustar);
    unsigned slow1 = 0u, end1 = 0u, cubic = sf.fast == 2;
    if (ustar < (unsigned)kRec) {                               // segment i + 1 starts at u = ustar: what its lane found
      const unsigned fl = s_fl[w][lo];
      end1 = (fl & kF_End1) != 0u;
      slow1 = ((fl & kF_Slow1) != 0u) || (ul >= (int)ustar && ul < kRec);
      cubic |= (fl & kF_Cubic1) != 0u;
    }
    const bool e0 = (ustar < (unsigned)kRec) || rem == kRec;
    BlockRec o;
    o.w0 = ((unsigned)((int)rel0 << 16)) | (ustar - 1u) | (e0 ? kRecE0 : 0u) | (end1 ? kRecE1 : 0u) | (slow0 ? kRecSlow0 : 0u) |
           (slow1 ? kRecSlow1 : 0u) | (cubic ? kRecCubic : 0u) | ((e0 ? ustar - 1u : 63u) << kRecLastShift);
    o.F = (float)(q0.a0 - r0);
    o.e1 = (float)q0.a1m1;
    o.e2 = (float)q0.a2;
#if PAR_REC_EXP & 1
    if (o.w0 == 0x12345678u)
#endif
#if PAR_REC_EXP & 4
    {
      typedef unsigned u4v __attribute__((ext_vector_type(4)));
      u4v v = {o.w0, __float_as_uint(o.F), __float_as_uint(o.e1), __float_as_uint(o.e2)};
      __builtin_nontemporal_store(v, reinterpret_cast<u4v*>(rec + g));
    }
#else
    rec[g] = o;
#endif
  }
}

__device__ __forceinline__ void mark_direct(long long i, long long* direct, PlanHeader* h) {
  const int slot = atomicAdd(&h->n_direct, 1);
  if (slot < kMaxDirect) direct[slot] = i;
  else atomicOr(&h->flags, kFlagDirectOverflow);
}

// xs[i] = inclusive plain-f64 scan of S  ->  approx offsets xa_i = st0 + xs[i-1].
// Build the parity-translation element of step i (x_i -> x_{i+1}) or mark it direct.
__global__ void k_off_prepare(const double* __restrict__ S, const double* __restrict__ xs, const double* __restrict__ st,
                              int64_t nseg, PElem* __restrict__ el, long long* __restrict__ direct,
                              PlanHeader* __restrict__ h) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nseg) return;
  const double st0 = st[0];
  const double xa = st0 + (i > 0 ? xs[i - 1] : 0.0);
  const double xb = st0 + xs[i];
  const double lo = 1.0 - 0x1p-30, hi = 1.0 + 0x1p-30;
  const int e = f64_exponent(xa);
  bool interior = xa > 0.0 && S[i] > 0.0 && e > 64 && e < 2046 && f64_exponent(xa * lo) == e &&
                  f64_exponent(xa * hi) == e && f64_exponent(xb * lo) == e && f64_exponent(xb * hi) == e;
  PElem p{0, 0, 0, 0};
  if (interior) {
    // u = ulp in binade e = 2^(e-1075); S/u is an exact power-of-two scaling
    const double t = ldexp(S[i], 1075 - e);
    if (t < 0x1p62) {
      const double fl = floor(t);
      const long long q = (long long)fl;
      const double fr = t - fl;                // exact
      if (fr > 0.5) p.c0 = p.c1 = q + 1;
      else if (fr < 0.5) p.c0 = p.c1 = q;
      else {                                   // exact half: ties-to-even on the SUM's parity
        p.c0 = q + (q & 1);
        p.c1 = q + ((q + 1) & 1);
      }
    } else {
      interior = false;
    }
  }
  if (!interior) mark_direct(i, direct, h);
  el[i] = p;
}

// head flags: a run starts at 0 and after every direct step
__global__ void k_off_heads(PElem* __restrict__ el, int64_t nseg, const long long* __restrict__ direct,
                            const PlanHeader* __restrict__ h) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  int nd = h->n_direct;
  if (nd > kMaxDirect) nd = kMaxDirect;
  if (j == 0) el[0].head = 1;
  if (j < nd) {
    const long long d = direct[j];
    if (d + 1 < nseg) el[d + 1].head = 1;
  }
}

__device__ __forceinline__ double apply_elem(const PElem& E, double xa) {
  // xa is the (bit-exact) offset at the start of the run; work in units of ulp(xa)
  const int e = f64_exponent(xa);
  const long long X = (long long)ldexp(xa, 1075 - e);      // exact integer in [2^52, 2^53)
  const long long Y = X + ((X & 1) ? E.c1 : E.c0);
  return ldexp((double)Y, e - 1075);
}

// one thread: sort the direct steps, walk the runs, evaluate the crossing steps with real float64 adds
__global__ void k_off_stitch(const double* __restrict__ S, const PElem* __restrict__ E, const double* __restrict__ st,
                             int64_t nseg, long long* __restrict__ direct, RunEntry* __restrict__ runs,
                             PlanHeader* __restrict__ h) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (h->lazy_fail) return;         // (a failed lazy pass may have listed thousands of direct steps)
  int nd = h->n_direct;
  if (nd > kMaxDirect) nd = kMaxDirect;
  for (int a = 1; a < nd; ++a) {               // insertion sort (nd is tiny)
    const long long v = direct[a];
    int b = a - 1;
    while (b >= 0 && direct[b] > v) {
      direct[b + 1] = direct[b];
      --b;
    }
    direct[b + 1] = v;
  }
  long long a = 0;
  double xa = st[0];
  int r = 0;
  for (int j = 0; j < nd; ++j) {
    const long long d = direct[j];
    runs[r].start = a;
    runs[r].x = xa;
    ++r;
    const double xd = (d == a) ? xa : apply_elem(E[d - 1], xa);
    xa = xd + S[d];                            // the direct step: a real float64 add
    a = d + 1;
  }
  runs[r].start = a;
  runs[r].x = xa;
  ++r;
  runs[r].start = INT64_MAX;                   // sentinel
  runs[r].x = 0.0;
  h->n_direct = nd;
  h->n_runs = r;
}

// ---- fused stages of the device plan (r04) ----------------------------------------------------------------------------
// Until r03 the plan was ~27 launches: every elementwise stage its own kernel around three-kernel scans, and a read-back
// in the middle.  The elementwise stages now ride inside the scans' own passes (a block's reduce pass only needs the
// block's elements, its apply pass has the running prefix at hand), single-thread epilogues run in the last block to
// finish, and the read-back of the long-segment count is gone (a plan that turns out to hold long segments is simply made
// again with the chunked-cumsum kernels in: sparse curves only, a few hundred segments).  Same arithmetic, bit for bit.
__device__ __forceinline__ U128 want_fixed(const double* __restrict__ st, const double* __restrict__ sp, int64_t i, bool* bad) {
  const double period = st[i + 1] - st[i];
  const double mean = (sp[i] + sp[i + 1]) / 2.0;
  const double a = period * mean;
  U128 e{0ull, 0ull};
  if (!(a >= 0x1p-11 && a < 0x1p62)) {
    *bad = true;
  } else {
    const double ip = floor(a);
    e.hi = (unsigned long long)ip;
    e.lo = (unsigned long long)((a - ip) * 0x1p64);     // exact: < 2^64, at most 64 fractional bits
  }
  return e;
}

// lengths, pass 1: a_i in 64.64 fixed point on the fly, block sums; the header is initialised here by block 0 before any
// flag can be raised?  No: flags are raised by any block, so the header is initialised by k_init_header in front.
// Also the per-wave partial sums of the speed samples for the reference's buffer bound (k_speed_sum's role).
__global__ __launch_bounds__(kScanThreads) void k_len_reduce(const double* __restrict__ st, const double* __restrict__ sp,
                                                              int64_t nseg, U128* __restrict__ bsum, double* __restrict__ partial,
                                                              PlanHeader* __restrict__ h) {
  __shared__ U128 smem[kScanThreads / kWave];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  U128 acc = AddU128::identity();
  bool bad = false;
  double s = 0.0, c = 0.0;                           // Kahan over this thread's speed samples (m = nseg + 1 of them)
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k < nseg) acc = AddU128::combine(acc, want_fixed(st, sp, base + k, &bad));
    if (base + k <= nseg) {
      const double y = sp[base + k] - c;
      const double t = s + y;
      c = (t - s) - y;
      s = t;
    }
  }
  if (bad) atomicOr(&h->flags, kFlagRange);
  U128 total;
  block_exclusive<AddU128>(acc, smem, &total);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) partial[blockIdx.x * (kScanThreads / kWave) + threadIdx.x / kWave] = s;
}

// lengths, pass 2 (one block): exclusive scan of the block sums; the speed sum from the per-wave partials
__global__ __launch_bounds__(kScanThreads) void k_len_top(U128* __restrict__ bsum, int64_t nb, const double* __restrict__ partial,
                                                           int64_t n_part, PlanHeader* __restrict__ h) {
  __shared__ U128 smem[kScanThreads / kWave];
  __shared__ double red[kScanThreads];
  U128 carry = AddU128::identity();
  for (int64_t t0 = 0; t0 < nb; t0 += kScanTile) {
    const int64_t base = t0 + (int64_t)threadIdx.x * kScanItems;
    U128 v[kScanItems];
    U128 acc = AddU128::identity();
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      v[k] = (base + k < nb) ? bsum[base + k] : AddU128::identity();
      acc = AddU128::combine(acc, v[k]);
    }
    U128 total;
    U128 run = AddU128::combine(carry, block_exclusive<AddU128>(acc, smem, &total));
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      if (base + k < nb) bsum[base + k] = run;
      run = AddU128::combine(run, v[k]);
    }
    carry = AddU128::combine(carry, total);
  }
  double s = 0.0, c = 0.0;
  for (int64_t i = threadIdx.x; i < n_part; i += blockDim.x) {
    const double y = partial[i] - c;
    const double t = s + y;
    c = (t - s) - y;
    s = t;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = kScanThreads / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) h->speed_sum = red[0];
}

// lengths, pass 3: the running 64.64 sum at every segment -> n_i = round(A_i) - round(A_{i-1}), seg_start (k_seg_lengths)
__global__ __launch_bounds__(kScanThreads) void k_len_apply(const double* __restrict__ st, const double* __restrict__ sp,
                                                             int64_t nseg, const U128* __restrict__ bsum,
                                                             int64_t* __restrict__ seg_start, PlanHeader* __restrict__ h) {
  __shared__ U128 smem[kScanThreads / kWave];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  U128 v[kScanItems];
  U128 acc = AddU128::identity();
  bool bad = false;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = (base + k < nseg) ? want_fixed(st, sp, base + k, &bad) : AddU128::identity();
    acc = AddU128::combine(acc, v[k]);
  }
  U128 total;
  U128 run = AddU128::combine(bsum[blockIdx.x], block_exclusive<AddU128>(acc, smem, &total));
  int f = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t i = base + k;
    const U128 prev = run;                          // A_{i-1}
    run = AddU128::combine(run, v[k]);              // A_i
    if (i >= nseg) continue;
    bool amb = false;
    const unsigned long long Ni = round_fixed(run, &amb);
    const unsigned long long Np = i > 0 ? round_fixed(prev, &amb) : 0ull;
    seg_start[i] = (int64_t)Np;
    if (i == nseg - 1) {
      seg_start[nseg] = (int64_t)Ni;
      h->total_written = (int64_t)Ni;
    }
    if (amb) f |= kFlagAmbiguous;
    if (Ni < Np + 2ull) atomicMin(&h->first_bad, (unsigned long long)i);
  }
  if (f) atomicOr(&h->flags, f);
}

// offsets, pass 1: block sums of the plain float64 scan of S (the prediction of the offsets' binades)
__global__ __launch_bounds__(kScanThreads) void k_offs_reduce(const double* __restrict__ S, int64_t nseg, double* __restrict__ bsum) {
  __shared__ double smem[kScanThreads / kWave];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (base + k < nseg) acc = acc + S[base + k];
  double total;
  block_exclusive<AddF64>(acc, smem, &total);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// the parity-translation element of step i from the approximate offsets around it (k_off_prepare's body)
__device__ __forceinline__ PElem off_element(double xa, double xb, double Si, bool* interior_out) {
  const double lo = 1.0 - 0x1p-30, hi = 1.0 + 0x1p-30;
  const int e = f64_exponent(xa);
  bool interior = xa > 0.0 && Si > 0.0 && e > 64 && e < 2046 && f64_exponent(xa * lo) == e &&
                  f64_exponent(xa * hi) == e && f64_exponent(xb * lo) == e && f64_exponent(xb * hi) == e;
  PElem p{0, 0, 0, 0};
  if (interior) {
    const double t = ldexp(Si, 1075 - e);
    if (t < 0x1p62) {
      const double fl = floor(t);
      const long long q = (long long)fl;
      const double fr = t - fl;                // exact
      if (fr > 0.5) p.c0 = p.c1 = q + 1;
      else if (fr < 0.5) p.c0 = p.c1 = q;
      else {                                   // exact half: ties-to-even on the SUM's parity
        p.c0 = q + (q & 1);
        p.c1 = q + ((q + 1) & 1);
      }
    } else {
      interior = false;
    }
  }
  *interior_out = interior;
  return p;
}

// offsets, pass 3 of the float64 scan fused with k_off_prepare: the running sum in front of and behind every step is the
// approximate offset there (a PREDICTION of its binade, verified later by k_off_apply: any accurate association of the
// sum serves), from which the step's parity-translation element follows -- or the step is listed `direct`.
// Lazy plans (cand != nullptr): S holds closed-form sums good to lazy_bound(); a step whose element that uncertainty could
// change -- S/ulp within the bound of a rounding boundary -- or that is not an interior step is listed for the exact sum.
__global__ __launch_bounds__(kScanThreads) void k_offs_elements(const double* __restrict__ S, const double* __restrict__ st,
                                                                 int64_t nseg, const double* __restrict__ bsum_f,
                                                                 PElem* __restrict__ el, long long* __restrict__ direct,
                                                                 PlanHeader* __restrict__ h, Cand* __restrict__ cand,
                                                                 const double* __restrict__ sp,
                                                                 const int64_t* __restrict__ seg_start) {
  __shared__ double smem_f[kScanThreads / kWave];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  double v[kScanItems];
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = (base + k < nseg) ? S[base + k] : 0.0;
    acc = acc + v[k];
  }
  double total;
  double run = bsum_f[blockIdx.x] + block_exclusive<AddF64>(acc, smem_f, &total);     // sum of the steps in front of `base`
  const double st0 = st[0];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t i = base + k;
    const double xa = st0 + run;
    run = run + v[k];
    if (i >= nseg) continue;
    bool interior;
    PElem p = off_element(xa, st0 + run, v[k], &interior);
    if (!interior) mark_direct(i, direct, h);
    el[i] = p;
    if (cand != nullptr) {
      bool listed = !interior;
      if (interior) {
        const int e = f64_exponent(xa);
        const double t = ldexp(v[k], 1075 - e);
        const double fr = t - floor(t);
        // the segment's own bound only where the largest one a lazy plan admits (K = kLazyMaxN, s_min = 1/16) does not settle it:
        // 2 % of the segments at 7e8, all of them while ulp(x) is small
        const double dist = fabs(fr - 0.5);
        if (!(dist > ldexp(lazy_bound((double)kLazyMaxN, 0.0625), 1075 - e))) {
          const double s0 = sp[i], s1 = sp[i + 1];
          const double bu = ldexp(lazy_bound((double)(seg_start[i + 1] - seg_start[i]), s0 < s1 ? s0 : s1), 1075 - e);
          listed = !(dist > bu);
        }
      }
      if (listed) {
        const int slot = atomicAdd(&h->n_cand, 1);
        if (slot < kMaxCand) cand[slot] = Cand{(long long)i, xa, st0 + run};
      }
    }
  }
}


// ---- lazy plans (pos_plan.h): closed-form segment sums, exact sums only where the offset chain's rounding needs them ------
// One thread per segment: S~_i by the closed form, or the verdict that the curve is not one for a lazy plan.
__global__ __launch_bounds__(256) void k_seg_sum_lazy(const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                                                      int64_t nseg, double* __restrict__ S, PlanHeader* __restrict__ h) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nseg) return;
  const long long n = seg_start[i + 1] - seg_start[i];
  const double s0 = sp[i], s1 = sp[i + 1];
  double v = 0.0;
  if (lazy_segment_ok(n, s0, s1)) {
    v = lazy_prefix(s0, (s1 - s0) / (double)(n - 1), (double)n);
  } else {
    // a degenerate segment (n < 2) behind the trim never exists for the reference; in front of it the plan is refused
    // anyway.  Either way the eager plan words the verdict.
    atomicOr(&h->lazy_fail, 1);
  }
  S[i] = v;
}

// One lane per candidate: the reference's own sequential sum, then the segment's chain element once more.
__global__ __launch_bounds__(64) void k_seg_exact_list(const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                                                       int64_t nseg, double* __restrict__ S, PElem* __restrict__ el,
                                                       const Cand* __restrict__ cand, long long* __restrict__ direct,
                                                       PlanHeader* __restrict__ h) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  int nc = h->n_cand;
  if (nc > kMaxCand) {                                   // list overflow: not a curve for a lazy plan after all
    if (slot == 0) atomicOr(&h->lazy_fail, 2);
    return;
  }
  if (slot >= nc || h->lazy_fail) return;
  const Cand c = cand[slot];
  const long long i = c.i;
  const long long n = seg_start[i + 1] - seg_start[i];
  const Ramp r = make_ramp(sp[i], sp[i + 1], n);
  // (eight reciprocals at a time -- they do not depend on one another -- then the adds in the reference's order: a division
  // per trip of the loop made this ~110 cycles per step on a lane with nothing else to issue)
  double sum = 0.0;
  long long k = 0;
  for (; k + 8 <= n; k += 8) {
    double rr[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) rr[u] = ramp_recip((double)(k + u), r);
#pragma unroll
    for (int u = 0; u < 8; ++u) sum = sum + rr[u];
  }
  for (; k < n; ++k) sum = sum + ramp_recip((double)k, r);
  bool was_interior, interior;
  (void)off_element(c.xa, c.xb, S[i], &was_interior);
  const PElem p = off_element(c.xa, c.xb, sum, &interior);
  if (!interior && was_interior) mark_direct(i, direct, h);
  S[i] = sum;
  el[i] = p;
}

// offsets for every segment, verification of the binade prediction, end-trim detection (:129)
__global__ void k_off_apply(const double* __restrict__ sp, const PElem* __restrict__ E, const RunEntry* __restrict__ runs,
                            int64_t nseg, double n_in, double* __restrict__ seg_off, PlanHeader* __restrict__ h) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nseg || h->lazy_fail) return;
  const int nr = h->n_runs;
  int lo = 0, hi = nr - 1;                     // last run with start <= i
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (runs[mid].start <= i) lo = mid; else hi = mid - 1;
  }
  const long long a = runs[lo].start;
  const double xa = runs[lo].x;
  const double xi = (i == a) ? xa : apply_elem(E[i - 1], xa);
  const bool is_direct = runs[lo + 1].start == i + 1;      // step i ends its run
  const double xn = is_direct ? runs[lo + 1].x : apply_elem(E[i], xa);
  seg_off[i] = xi;
  if (i == nseg - 1) seg_off[nseg] = xn;
  if (!is_direct) {
    // both ends of an interior step must sit in the run's binade, else the integer model was wrong
    const int e = f64_exponent(xa);
    if (f64_exponent(xi) != e || f64_exponent(xn) != e) atomicOr(&h->flags, kFlagVerify);
  }
  const double first = 1.0 / sp[i] + xi;       // k = 0: bs = 0/(n-1)*ds + s0 = s0
  if (first <= n_in && n_in <= xn) atomicMin(&h->trim_seg, (unsigned long long)i);
}

// np.argmin |pos - n_in| inside the trim segment (first occurrence), header finalisation (one thread)
__device__ __forceinline__ void trim_body(const double* __restrict__ st, const double* __restrict__ sp,
                                          const int64_t* __restrict__ seg_start, const double* __restrict__ seg_off, int64_t m,
                                          double n_in, const double* __restrict__ ck, int64_t ck_len, PlanHeader* __restrict__ h) {
  const int64_t nseg = m - 1;
  // int(np.mean(speeds) * (st[-1]-st[0]) * 1.01)  (:108)
  // speed_sum (k_speed_sum) and numpy's pairwise sum each sit within ~2.5e-15 (relative) of the true sum.  Only
  // when the product lies that close to an integer can int() differ from numpy's: such plans (1 file in ~10^5 at
  // hour length) go to the serial host path, which restates numpy's pairwise order.
  const double guess = (h->speed_sum / (double)m) * (st[m - 1] - st[0]) * 1.01;
  h->cap = (int64_t)guess;
  if (fabs(guess - rint(guess)) <= fabs(guess) * 1e-14 + 1e-12) atomicOr(&h->flags, kFlagCapAmbiguous);
  int64_t len = h->total_written;
  if (h->first_bad != kNoTrim && !(h->trim_seg != kNoTrim && h->trim_seg < h->first_bad)) atomicOr(&h->flags, kFlagBadLength);
  if (h->trim_seg != kNoTrim && h->trim_seg < (unsigned long long)nseg) {
    const int64_t i = (int64_t)h->trim_seg;
    const long long n = seg_start[i + 1] - seg_start[i];
    const Ramp r = make_ramp(sp[i], sp[i + 1], n);
    const double off = seg_off[i];
    double c = 0.0, best = INFINITY;
    long long arg = 0, k_from = 0, k_to = n;
    if (r.fast && long_segment(n, seg_start[i], i, ck, ck_len)) {
      // positions rise monotonically (positive speeds), so |pos - n_in| has its minimum where they cross n_in:
      // bisect the cumsum checkpoints for the last block that ends at or below n_in and search around it only
      const long long slot0 = ck_slot0(seg_start[i], i), nb = (n - 1) / kCk;      // checkpoints 1 .. nb exist
      long long lo = 0, hi = nb;                                                  // checkpoint 0 is the empty sum
      while (lo < hi) {
        const long long mid = (lo + hi + 1) >> 1;
        if (ck[slot0 + mid] + off <= n_in) lo = mid; else hi = mid - 1;
      }
      const long long b = lo > 0 ? lo - 1 : 0;                                    // one block of slack on the low side
      c = b ? ck[slot0 + b] : 0.0;
      k_from = b * kCk;
      k_to = (lo + 3) * kCk < n ? (lo + 3) * kCk : n;
      if (k_from > 0) best = fabs((c + off) - n_in), arg = k_from - 1;            // the sample the checkpoint stands for
    }
    long long k = k_from;
    for (; k + 8 <= k_to; k += 8) {              // (reciprocals in eights, adds and tests in order: see k_seg_exact_list)
      double rr[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) rr[u] = ramp_recip((double)(k + u), r);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c = c + rr[u];
        const double d = fabs((c + off) - n_in);
        if (d < best) {
          best = d;
          arg = k + u;
        }
      }
    }
    for (; k < k_to; ++k) {
      c = c + ramp_recip((double)k, r);
      const double d = fabs((c + off) - n_in);
      if (d < best) {
        best = d;
        arg = k;
      }
    }
    len = seg_start[i] + arg;
    h->trimmed = 1;
    h->written = seg_start[i + 1];        // the whole trim segment is written before the test (:127-129)
  } else {
    h->written = h->total_written;
  }
  h->len_out = len;
}

__global__ void k_trim(const double* __restrict__ st, const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                       const double* __restrict__ seg_off, int64_t m, double n_in, const double* __restrict__ ck,
                       int64_t ck_len, PlanHeader* __restrict__ h) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  trim_body(st, sp, seg_start, seg_off, m, n_in, ck, ck_len, h);
}

// Lazy plans: the trim (it walks the trim segment from its first step: <= kLazyMaxN steps) and, since nothing behind it can
// raise a flag any more, the verdict the record kernels and the host read: ck_valid = 2 (K_sinc may run; no checkpoints).
__global__ void k_trim_lazy(const double* __restrict__ st, const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                            const double* __restrict__ seg_off, int64_t m, double n_in, int64_t ck_len, int64_t max_tiles,
                            PlanHeader* __restrict__ h) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (h->lazy_fail) return;
  trim_body(st, sp, seg_start, seg_off, m, n_in, nullptr, 0, h);
  const int fl = __hip_atomic_load(&h->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const long long n_tiles = (h->len_out + kSincTileOutputs - 1) / kSincTileOutputs;
  h->ck_len = ck_len;
  h->ck_valid = ((fl & ~kFlagCapAmbiguous) || n_tiles + 1 > max_tiles) ? 0 : 2;
}

// k_off_apply with k_trim as the epilogue of the last block to finish (r04).  Grid-stride over the segments with at most 512
// blocks: every block ends with ONE atomic on the header's counter, and thousands of them serialise on that word (measured:
// 0.4 ms for the 10 548 blocks of a 2.7 M-segment curve).
__global__ __launch_bounds__(256) void k_off_apply_trim(const double* __restrict__ st, const double* __restrict__ sp,
                                                        const PElem* __restrict__ E, const RunEntry* __restrict__ runs,
                                                        const int64_t* __restrict__ seg_start, int64_t nseg, double n_in,
                                                        double* __restrict__ seg_off, const double* __restrict__ ck,
                                                        int64_t ck_len, PlanHeader* __restrict__ h) {
  __shared__ int is_last;
  if (h->lazy_fail) return;           // not a curve for a lazy plan: it is being made again the eager way (and the
                                                 // trim's walk over a segment without checkpoints could take seconds)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nseg; i += (int64_t)gridDim.x * blockDim.x) {
    const int nr = h->n_runs;
    int lo = 0, hi = nr - 1;                     // last run with start <= i
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (runs[mid].start <= i) lo = mid; else hi = mid - 1;
    }
    const long long a = runs[lo].start;
    const double xa = runs[lo].x;
    const double xi = (i == a) ? xa : apply_elem(E[i - 1], xa);
    const bool is_direct = runs[lo + 1].start == i + 1;      // step i ends its run
    const double xn = is_direct ? runs[lo + 1].x : apply_elem(E[i], xa);
    seg_off[i] = xi;
    if (i == nseg - 1) seg_off[nseg] = xn;
    if (!is_direct) {
      const int e = f64_exponent(xa);
      if (f64_exponent(xi) != e || f64_exponent(xn) != e) atomicOr(&h->flags, kFlagVerify);
    }
    const double first = 1.0 / sp[i] + xi;       // k = 0: bs = 0/(n-1)*ds + s0 = s0
    if (first <= n_in && n_in <= xn) atomicMin(&h->trim_seg, (unsigned long long)i);
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) is_last = atomicAdd(&h->pad2, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!is_last || threadIdx.x != 0) return;
  __threadfence();                               // acquire: the other blocks' seg_off and header atomics
  h->pad2 = 0;
  trim_body(st, sp, seg_start, seg_off, nseg + 1, n_in, ck, ck_len, h);
}

// pos[start_i + k] = cumsum_k + offset_i  (:125).  One lane per segment; each wave transposes 64 x 16
// blocks through LDS so that HBM sees contiguous 128-byte runs.
constexpr int kFillChunk = 32;
constexpr int kFillWaves = 4;
__global__ __launch_bounds__(kWave * kFillWaves) void k_pos_fill(const double* __restrict__ sp,
                                                                  const int64_t* __restrict__ seg_start,
                                                                  const double* __restrict__ seg_off, int64_t nseg,
                                                                  int64_t len_out, int64_t j_lo, int64_t j_hi,
                                                                  double* __restrict__ pos) {
  __shared__ double buf[kFillWaves][kWave][kFillChunk + 1];
  __shared__ long long s_start[kFillWaves][kWave];
  __shared__ long long s_n[kFillWaves][kWave];      // a sparse curve's segment can exceed 2^31 samples
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  const int64_t i = ((int64_t)blockIdx.x * kFillWaves + w) * kWave + lane;
  long long n = 0, start = 0;
  Ramp r = make_ramp(1.0, 1.0, 2);
  double off = 0.0;
  if (i < nseg) {
    start = seg_start[i];
    n = seg_start[i + 1] - start;
    if (n >= 2) r = make_ramp(sp[i], sp[i + 1], n);
    if (start >= (long long)len_out || start < (long long)j_lo || start >= (long long)j_hi) n = 0;   // not this chunk's
    else if (start + n > (long long)len_out) n = len_out - start;     // trimmed tail: fewer samples, same ramp
    off = seg_off[i];
  }
  s_start[w][lane] = start;
  s_n[w][lane] = n;
  long long nmax = n;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const long long t = __shfl_xor(nmax, o, kWave);
    nmax = t > nmax ? t : nmax;
  }
  double c = 0.0;
  for (long long k0 = 0; k0 < nmax; k0 += kFillChunk) {   // nmax == 0: the whole wave belongs to other chunks
    for (int kk = 0; kk < kFillChunk; kk += 4) {
      // four independent IEEE divisions in flight; only the running sum is serial
      const double ak = (double)(k0 + kk);
      const double r0 = ramp_recip(ak, r), r1 = ramp_recip(ak + 1.0, r);
      const double r2 = ramp_recip(ak + 2.0, r), r3 = ramp_recip(ak + 3.0, r);
      const double c0 = c + r0, c1 = c0 + r1, c2 = c1 + r2, c3 = c2 + r3;
      // rows past the segment end are never read back (the store side checks k < n)
      buf[w][lane][kk] = c0 + off;
      buf[w][lane][kk + 1] = c1 + off;
      buf[w][lane][kk + 2] = c2 + off;
      buf[w][lane][kk + 3] = c3 + off;
      c = c3;
    }
    // wave-synchronous transpose: rows = segments, kFillChunk consecutive outputs each; several rows per pass
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int col = lane & (kFillChunk - 1), half = lane / kFillChunk;
    for (int seg = 0; seg < kWave; seg += kWave / kFillChunk) {
      const int sg = seg + half;
      const long long k = k0 + col;
      if (k < s_n[w][sg]) pos[s_start[w][sg] + k] = buf[w][sg][col];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------ serial host path (exact, slow)
// Used when the device plan flags an ambiguity; follows the reference loop literally.
// numpy's float64 pairwise summation (np.add.reduce on a contiguous array): < 8 elements sequential, <= 128 eight
// interleaved accumulators combined as a balanced tree plus a sequential tail, else split at floor(n/2) rounded
// down to a multiple of 8.  np.mean(speeds) = this / m; the serial path sizes the reference's buffer with it.
static double np_pairwise_sum(const double* a, int64_t n) {
  if (n < 8) {
    double r = 0.0;
    for (int64_t i = 0; i < n; ++i) r += a[i];
    return r;
  }
  if (n <= 128) {
    double r[8];
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    int64_t i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int k = 0; k < 8; ++k) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int64_t n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

__global__ void k_set_total(PlanHeader* h, int64_t total, unsigned long long first_bad) {
  h->total_written = total;
  h->first_bad = first_bad;
}

// Segment lengths n_i by the reference's own recurrence (:111-118), serially on the host: inerr = n + err in float64,
// Python round (half to even), err carried.  O(m); used when the exact fixed-point scan on the device meets a sum too
// close to a rounding tie to call.  *ok = false when some n_i < 2 (the serial path then words the diagnosis).
static int host_lengths(const PlanView& pv, const double* d_st, const double* d_sp, int64_t m, hipStream_t s, bool* ok) {
  const int64_t nseg = m - 1;
  std::vector<double> sp(m), st(m);
  PAR_HIP_CHECK(hipMemcpyAsync(sp.data(), d_sp, m * sizeof(double), hipMemcpyDeviceToHost, s));
  PAR_HIP_CHECK(hipMemcpyAsync(st.data(), d_st, m * sizeof(double), hipMemcpyDeviceToHost, s));
  PAR_HIP_CHECK(hipStreamSynchronize(s));
  std::vector<int64_t> start(m);
  double err = 0.0;
  int64_t acc = 0;
  unsigned long long first_bad = kNoTrim;
  *ok = true;
  for (int64_t i = 0; i < nseg; ++i) {
    const double inerr = (st[i + 1] - st[i]) * ((sp[i] + sp[i + 1]) / 2.0) + err;
    const double rn = nearbyint(inerr);          // Python round(): half to even
    if (!(rn >= 0.0 && rn < 9.0e15)) {
      *ok = false;
      return PAR_OK;
    }
    if (rn < 2.0 && first_bad == kNoTrim) first_bad = (unsigned long long)i;   // harmless behind the trim: k_trim decides
    err = inerr - rn;
    start[i] = acc;
    acc += (int64_t)rn;
  }
  start[nseg] = acc;
  PAR_HIP_CHECK(hipMemcpyAsync(pv.seg_start, start.data(), m * sizeof(int64_t), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_set_total, dim3(1), dim3(1), 0, s, pv.hdr, acc, first_bad);
  PAR_HIP_CHECK(hipStreamSynchronize(s));        // `start` must outlive the copy
  return PAR_OK;
}

static int host_plan(const PlanView& pv, const double* d_st, const double* d_sp, int64_t m, int64_t n_in,
                     PlanHeader* out, hipStream_t s) {
  const int64_t nseg = m - 1;
  std::vector<double> sp(m), st(m);
  PAR_HIP_CHECK(hipMemcpyAsync(sp.data(), d_sp, m * sizeof(double), hipMemcpyDeviceToHost, s));
  PAR_HIP_CHECK(hipMemcpyAsync(st.data(), d_st, m * sizeof(double), hipMemcpyDeviceToHost, s));
  PAR_HIP_CHECK(hipStreamSynchronize(s));
  std::vector<int64_t> start(m);
  std::vector<double> off(m);
  double err = 0.0, offset = st[0];
  const double sum = np_pairwise_sum(sp.data(), m);
  const int64_t cap = (int64_t)((sum / (double)m) * (st[m - 1] - st[0]) * 1.01);
  int64_t acc = 0, out_len = -1;
  int trim = 0;
  for (int64_t i = 0; i < nseg; ++i) {
    const double a = (st[i + 1] - st[i]) * ((sp[i] + sp[i + 1]) / 2.0);
    const double inerr = a + err;
    const double rn = nearbyint(inerr);          // Python round(): half to even
    PAR_REQUIRE(rn >= 2.0 && rn < 9.0e15, PAR_ERR_ARG,
                "par_speed_to_pos_plan: segment %lld has n=%g samples (reference needs n >= 2)", (long long)i, rn);
    err = inerr - rn;
    const int64_t n = (int64_t)rn;
    start[i] = acc;
    off[i] = offset;
    PAR_REQUIRE(acc + n <= cap, PAR_ERR_ARG,
                "par_speed_to_pos_plan: positions overflow the reference's end_guess buffer (%lld > %lld); it raises here",
                (long long)(acc + n), (long long)cap);
    const double ds = sp[i + 1] - sp[i], nm1 = (double)(n - 1);
    double c = 0.0, first = 0.0;
    for (int64_t k = 0; k < n; ++k) {
      const double bs = ((double)k / nm1) * ds + sp[i];
      c += 1.0 / bs;
      if (k == 0) first = c + offset;
    }
    const double last = c + offset;
    acc += n;
    if (first <= (double)n_in && (double)n_in <= last) {
      double c2 = 0.0, best = INFINITY;
      int64_t arg = 0;
      for (int64_t k = 0; k < n; ++k) {
        const double bs = ((double)k / nm1) * ds + sp[i];
        c2 += 1.0 / bs;
        const double d = fabs((c2 + offset) - (double)n_in);
        if (d < best) {
          best = d;
          arg = k;
        }
      }
      out_len = start[i] + arg;
      trim = 1;
      // later segments are never produced by the reference; give them empty, consistent entries
      for (int64_t j = i + 1; j <= nseg; ++j) {
        start[j] = acc;
        off[j] = last;
      }
      break;
    }
    offset = last;
  }
  if (out_len < 0) {
    start[nseg] = acc;
    off[nseg] = offset;
    out_len = acc;
  }
  out->m = m;
  out->len_out = out_len;
  out->total_written = acc;
  out->trim_seg = kNoTrim;
  out->cap = cap;
  out->trimmed = trim;
  out->flags = 0;
  out->n_direct = 0;
  out->n_runs = 0;
  out->speed_sum = sum;
  out->ck_len = 0;
  out->ck_valid = 0;
  out->pad2 = 0;
  out->n_long = 0;          // counted afresh by k_count_long when the checkpoints are redone for this segmentation
  out->pad3 = 0;
  out->written = acc;
  out->first_bad = kNoTrim;
  PAR_HIP_CHECK(hipMemcpyAsync(pv.seg_start, start.data(), m * sizeof(int64_t), hipMemcpyHostToDevice, s));
  PAR_HIP_CHECK(hipMemcpyAsync(pv.seg_off, off.data(), m * sizeof(double), hipMemcpyHostToDevice, s));
  PAR_HIP_CHECK(hipMemcpyAsync(pv.hdr, out, sizeof(PlanHeader), hipMemcpyHostToDevice, s));
  PAR_HIP_CHECK(hipStreamSynchronize(s));
  return PAR_OK;
}



// Position fill from the cumsum checkpoints of a fused plan: one lane per 8-sample block restarts from the block's
// checkpoint (the same regeneration K_sinc<fused> does in LDS), so the work is spread over len_out/8 lanes whatever the
// segment lengths are -- the lane-per-segment fill below needs seconds for a segment of 10^8 samples.
__global__ __launch_bounds__(256) void k_pos_fill_ck(const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                                                     const double* __restrict__ seg_off, int64_t nseg,
                                                     const double* __restrict__ ck, int64_t n_slots, int64_t len_out,
                                                     double* __restrict__ pos, const PlanHeader* __restrict__ h) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_slots) return;
  // A LAZY plan (ck_valid 2, ABI 103+) or one whose checkpoints did not fit (0) left `ck` unwritten: a caller on the older
  // contract ("fused_ok != 0, so the fused fill is allowed") gets NaN positions, not garbage that looks like positions
  // (no host synchronisation needed: the header is read here)
  const bool have_ck = h->ck_valid == 1;
  long long lo = 0, hi = nseg - 1;                        // largest segment whose first slot is <= g
  while (lo < hi) {
    const long long mid = (lo + hi + 1) >> 1;
    if (ck_slot0(seg_start[mid], mid) <= g) lo = mid; else hi = mid - 1;
  }
  const long long i = lo, start = seg_start[i], n = seg_start[i + 1] - start;
  const long long b = g - ck_slot0(start, i), k0 = b * kCk;
  if (b < 0 || k0 >= n || start + k0 >= (long long)len_out) return;     // gap slot, or past the trim
  const Ramp r = make_ramp(sp[i], sp[i + 1], n);
  const double off = seg_off[i];
  double c = b ? ck[g] : 0.0;
  double rr[kCk];
  const double a0 = (double)k0;
#pragma unroll
  for (int u = 0; u < kCk; ++u) rr[u] = ramp_recip(a0 + (double)u, r);
#pragma unroll
  for (int u = 0; u < kCk; ++u) {
    c = c + rr[u];
    const long long jj = start + k0 + u;
    if (k0 + u < n && jj < (long long)len_out) pos[jj] = have_ck ? c + off : __longlong_as_double(0x7ff8000000000000ll);
  }
}

// positions of the segments whose first output index lies in [j_lo, j_hi)  (used whole or chunked)
int launch_pos_fill(const double* speeds, int64_t m, const void* work, double* pos, int64_t len_out, int64_t j_lo,
                    int64_t j_hi, hipStream_t s) {
  PlanView pv = plan_view(const_cast<void*>(work), m);
  const int64_t nseg = m - 1;
  hipLaunchKernelGGL(k_pos_fill, dim3((unsigned)ceil_div(nseg, kWave * kFillWaves)), dim3(kWave * kFillWaves), 0, s, speeds,
                     pv.seg_start, pv.seg_off, nseg, len_out, j_lo, j_hi, pos);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

}  // namespace par

extern "C" {

size_t par_speed_plan_bytes(int64_t m) { return par::plan_bytes(m < 2 ? 2 : m); }

// Per-segment reciprocal sums (+ checkpoints): one lane per ordinary segment, the chunked exact path for long ones.
// mode 0: count the long segments, read the count back and run the chunked-cumsum kernels if there are any (the serial-path
//         plans: a stream synchronisation in the middle);
// mode 1: k_seg_sum counts its long segments itself and nothing else happens here -- the caller finds the count in the
//         header it reads at the end of the plan and, if it is not zero, makes the plan again in mode 2;
// mode 2: the chunked-cumsum kernels unconditionally.
static int launch_seg_sums(const double* speeds, const par::PlanView& pv, int64_t nseg, double* ck, int64_t ck_len,
                            int64_t max_out, int64_t m, hipStream_t s, int mode = 0) {
  using namespace par;
  const unsigned g256 = (unsigned)ceil_div(nseg, 256);
  if (ck && mode == 0) hipLaunchKernelGGL(k_count_long, dim3(g256), dim3(256), 0, s, pv.seg_start, nseg, (const double*)ck, ck_len, pv.hdr);
  hipLaunchKernelGGL(k_seg_sum, dim3((unsigned)ceil_div(nseg, 64)), dim3(64), 0, s, speeds, pv.seg_start, nseg, pv.S, ck,
                     ck_len, pv.hdr, mode);
  if (!ck || mode == 1) return PAR_OK;
  if (mode == 2) goto long_kernels;
  // Curves without a long segment (every dense curve) skip the nine chunked-cumsum launches: one 4-byte read-back,
  // issued behind k_seg_sum so the GPU stays busy while the host waits.  (It is not only the launches: under a
  // concurrent K_sinc the side stream is served well for about a millisecond and then starves until K_sinc
  // drains -- measured --, so a plan that is to hide under the previous file's K_sinc has to be short.)
  {
    int n_long = 0;
    PAR_HIP_CHECK(hipMemcpyAsync(&n_long, &pv.hdr->n_long, sizeof(int), hipMemcpyDeviceToHost, s));
    PAR_HIP_CHECK(hipStreamSynchronize(s));
    if (n_long == 0) return PAR_OK;
  }
long_kernels:
  const long long G = max_out / kLongChunk + m + 8;                 // bound on the global chunk slots
  const long long GW = G / kWinSlotDiv + 8;                         // ... and on the global window slots
  const unsigned gc = (unsigned)ceil_div(G, 256), gs = (unsigned)(nseg < 2048 ? nseg : 2048),
                 gw = (unsigned)(GW < 8192 ? GW : 8192);
  hipLaunchKernelGGL(k_long_approx, dim3(gc), dim3(256), 0, s, speeds, pv.seg_start, nseg, ck, ck_len, G, pv.hdr);
  hipLaunchKernelGGL(k_long_wsum, dim3(gw), dim3(256), 0, s, pv.seg_start, nseg, ck, ck_len, GW, pv.hdr);
  hipLaunchKernelGGL(k_long_prefix, dim3(gs), dim3(256), 0, s, pv.seg_start, nseg, ck, ck_len, pv.hdr);
  hipLaunchKernelGGL(k_long_wprefix, dim3(gw), dim3(256), 0, s, pv.seg_start, nseg, ck, ck_len, GW, pv.hdr);
  hipLaunchKernelGGL(k_long_map, dim3(gc), dim3(256), 0, s, speeds, pv.seg_start, nseg, ck, ck_len, G, pv.hdr);
  hipLaunchKernelGGL(k_long_wmap, dim3(gw), dim3(256), 0, s, pv.seg_start, nseg, ck, ck_len, GW, pv.hdr);
  hipLaunchKernelGGL(k_long_stitch, dim3(gs), dim3(256), 0, s, speeds, pv.seg_start, nseg, ck, ck_len, pv.hdr);
  hipLaunchKernelGGL(k_long_wapply, dim3(gw), dim3(256), 0, s, pv.seg_start, nseg, ck, ck_len, GW, pv.hdr);
  hipLaunchKernelGGL(k_long_final, dim3(gc), dim3(256), 0, s, speeds, pv.seg_start, nseg, pv.S, ck, ck_len, G, pv.hdr);
  return PAR_OK;
}

// block records + tile headers for every block that may hold outputs (the kernel reads len_out on the device)
static void launch_block_rec(const double* speeds, const par::PlanView& pv, int64_t nseg, void* aux, int64_t max_out, int64_t m,
                             hipStream_t s) {
  using namespace par;
  const FusedAux a = fused_aux_view(aux, max_out, m);
  const int64_t blocks = (int64_t)fused_blocks(max_out);
  hipLaunchKernelGGL(k_block_rec, dim3((unsigned)ceil_div(blocks, 256)), dim3(256), 0, s, speeds, pv.seg_start, nseg,
                     (const double*)a.ck, (const int64_t*)a.tile_seg, (const long long*)a.tile_st, (const SegFast*)a.seg_fast,
                     (const TileHdr*)a.hdr, a.rec, (const PlanHeader*)pv.hdr);
  hipLaunchKernelGGL(k_block_rec2, dim3((unsigned)ceil_div(nseg, 256)), dim3(256), 0, s, speeds, pv.seg_start, nseg,
                     (const SegFast*)a.seg_fast, (const TileHdr*)a.hdr, a.rec, a.rec2, (const PlanHeader*)pv.hdr);
}

// Shared implementation.  aux (optional, device): cumsum checkpoints for the fused resampler.
static thread_local int g_last_plan_flags = 0;

static int plan_impl(int device, const double* sampletimes, const double* speeds, int64_t m, int64_t n_in, void* work,
                     size_t work_bytes, void* aux, size_t aux_bytes, int64_t max_out, int64_t* len_out, int* trimmed,
                     int force_host, int* path_used, int* fused_ok, void* stream) {
  using namespace par;
  PAR_REQUIRE(sampletimes && speeds && work && len_out && trimmed, PAR_ERR_ARG, "par_speed_to_pos_plan: null pointer");
  PAR_REQUIRE(m >= 2, PAR_ERR_ARG, "par_speed_to_pos_plan: need at least 2 speed samples (m=%lld)", (long long)m);
  PAR_REQUIRE(work_bytes >= plan_bytes(m), PAR_ERR_WORKSPACE, "par_speed_to_pos_plan: workspace %zu < %zu", work_bytes,
              plan_bytes(m));
  PAR_REQUIRE(!aux || aux_bytes >= fused_aux_bytes(max_out, m), PAR_ERR_WORKSPACE,
              "par_speed_to_pos_plan: aux buffer %zu < %zu", aux_bytes, aux ? fused_aux_bytes(max_out, m) : (size_t)0);
  PAR_HIP_CHECK(hipSetDevice(device));
  hipStream_t s = as_stream(stream);
  const int64_t nseg = m - 1;
  PlanView pv = plan_view(work, m);
  double* ck = static_cast<double*>(aux);
  const int64_t ck_len = aux ? (int64_t)fused_ck_len(max_out, m) : 0;
  const int64_t max_tiles = aux ? (int64_t)fused_tiles(max_out) : 0;
  PlanHeader h;
  memset(&h, 0, sizeof(h));
  bool need_host = (force_host & 3) != 0;
  // lazy plan (pos_plan.h): closed-form segment sums, exact ones only where a rounding decides -- for the fused resampler on
  // dense, gentle curves (every segment checked on the device; a curve that is not one is planned again the eager way).
  // force_host & 8 asks for the eager plan outright (callers that fill positions from the checkpoints).
  bool lazy = aux != nullptr && !need_host && !(force_host & 8) && n_in / (m - 1) <= kLazyMaxN;
  g_last_plan_flags = 0;
  // attempt 0: everything on the device.  attempt 1 (only after a near-tie in the segment lengths): the O(m) length
  // recurrence is redone serially on the host in the reference's own float64 order, everything else -- the O(len_out)
  // reciprocal sums, offsets, trim, checkpoints -- stays on the device.
  bool host_made_lengths = false;
  bool with_long = false;                       // the plan holds long segments: found out from the header of the first go
  for (int attempt = 0; attempt < 2 && !need_host; ++attempt) {
    const unsigned g256 = (unsigned)ceil_div(nseg, 256);
    const unsigned nbs = (unsigned)ceil_div(nseg, kScanTile), nbm = (unsigned)ceil_div(m, kScanTile);
    double* scratch_f = pv.xs;                  // per-wave speed partials, later the float64 scan's block sums (xs itself is gone)
    hipLaunchKernelGGL(k_init_header, dim3(1), dim3(1), 0, s, pv.hdr, m, lazy ? 1 : 0);
    if (aux) PAR_HIP_CHECK(hipMemsetAsync(fused_aux_view(aux, max_out, m).redo_count, 0, 64, s));   // K_sinc's tile list: empty until a launch fills it
    int rc;
    if (attempt == 0) {
      // lengths: the 64.64 fixed-point scan with k_seg_want / k_seg_lengths / k_speed_sum inside its passes (3 launches)
      U128* bsum = reinterpret_cast<U128*>(pv.bsum);
      hipLaunchKernelGGL(k_len_reduce, dim3(nbm), dim3(kScanThreads), 0, s, sampletimes, speeds, nseg, bsum, scratch_f, pv.hdr);
      hipLaunchKernelGGL(k_len_top, dim3(1), dim3(kScanThreads), 0, s, bsum, (int64_t)nbs, (const double*)scratch_f,
                         (int64_t)nbm * (kScanThreads / kWave), pv.hdr);
      hipLaunchKernelGGL(k_len_apply, dim3(nbs), dim3(kScanThreads), 0, s, sampletimes, speeds, nseg, (const U128*)bsum,
                         pv.seg_start, pv.hdr);
    } else {
      bool lengths_ok = false;
      rc = host_lengths(pv, sampletimes, speeds, m, s, &lengths_ok);
      if (rc != PAR_OK) return rc;
      if (!lengths_ok) {                 // a negative or absurd length: the serial path words the diagnosis
        need_host = true;
        break;
      }
      host_made_lengths = true;
      hipLaunchKernelGGL(k_speed_sum, dim3((unsigned)(m / 4096 + 1)), dim3(256), 0, s, speeds, m,
                         reinterpret_cast<double*>(pv.bsum), pv.hdr);
    }
    if (lazy) {
      hipLaunchKernelGGL(k_seg_sum_lazy, dim3(g256), dim3(256), 0, s, speeds, (const int64_t*)pv.seg_start, nseg, pv.S, pv.hdr);
    } else {
      rc = launch_seg_sums(speeds, pv, nseg, ck, ck_len, max_out, m, s, with_long ? 2 : 1);
      if (rc != PAR_OK) return rc;
    }
    // offsets: float64 scan of S (its apply pass makes the parity-translation elements), heads, ComposeP scan (its apply
    // pass ends with the stitch), then the offsets themselves with the trim as epilogue: 8 launches
    PElem* el = reinterpret_cast<PElem*>(pv.scan);
    hipLaunchKernelGGL(k_offs_reduce, dim3(nbs), dim3(kScanThreads), 0, s, (const double*)pv.S, nseg, scratch_f);
    hipLaunchKernelGGL(k_scan_top<AddF64>, dim3(1), dim3(kScanThreads), 0, s, scratch_f, (int64_t)nbs);
    hipLaunchKernelGGL(k_offs_elements, dim3(nbs), dim3(kScanThreads), 0, s, (const double*)pv.S, sampletimes, nseg,
                       (const double*)scratch_f, el, pv.direct, pv.hdr, lazy ? pv.cand : (Cand*)nullptr, speeds,
                       (const int64_t*)pv.seg_start);
    if (lazy)
      hipLaunchKernelGGL(k_seg_exact_list, dim3(kMaxCand / 64), dim3(64), 0, s, speeds, (const int64_t*)pv.seg_start, nseg, pv.S,
                         el, (const Cand*)pv.cand, pv.direct, pv.hdr);
    hipLaunchKernelGGL(k_off_heads, dim3((kMaxDirect + 255) / 256), dim3(256), 0, s, el, nseg, pv.direct, pv.hdr);
    hipLaunchKernelGGL(k_scan_reduce<ComposeP>, dim3(nbs), dim3(kScanThreads), 0, s, (const PElem*)el, nseg,
                       reinterpret_cast<PElem*>(pv.bsum));
    hipLaunchKernelGGL(k_scan_top<ComposeP>, dim3(1), dim3(kScanThreads), 0, s, reinterpret_cast<PElem*>(pv.bsum), (int64_t)nbs);
    hipLaunchKernelGGL(k_scan_apply<ComposeP>, dim3(nbs), dim3(kScanThreads), 0, s, el, nseg, (const PElem*)pv.bsum);
    hipLaunchKernelGGL(k_off_stitch, dim3(1), dim3(1), 0, s, pv.S, el, sampletimes, nseg, pv.direct, pv.runs, pv.hdr);
    if (lazy) {
      // full grids without the last-block epilogues (their one atomic per block on a header word is what capped those grids)
      hipLaunchKernelGGL(k_off_apply, dim3(g256), dim3(256), 0, s, speeds, (const PElem*)el, (const RunEntry*)pv.runs, nseg,
                         (double)n_in, pv.seg_off, pv.hdr);
      hipLaunchKernelGGL(k_trim_lazy, dim3(1), dim3(1), 0, s, sampletimes, speeds, (const int64_t*)pv.seg_start,
                         (const double*)pv.seg_off, m, (double)n_in, ck_len, max_tiles, pv.hdr);
      const FusedAux av = fused_aux_view(aux, max_out, m);
      hipLaunchKernelGGL(k_tile_seg_lazy, dim3(g256), dim3(256), 0, s, speeds, (const int64_t*)pv.seg_start,
                         (const double*)pv.seg_off, nseg, av.tile_seg, av.seg_fast, av.tile_st, av.hdr, (const PlanHeader*)pv.hdr);
      hipLaunchKernelGGL(k_block_rec_lazy, dim3((unsigned)ceil_div(nseg, 4 * kRecLazySegs)), dim3(256), 0, s, speeds, (const int64_t*)pv.seg_start, nseg,
                         (const SegFast*)av.seg_fast, (const TileHdr*)av.hdr, av.rec, av.rec2, (const PlanHeader*)pv.hdr);
    } else {
    hipLaunchKernelGGL(k_off_apply_trim, dim3(g256 < 512u ? g256 : 512u), dim3(256), 0, s, sampletimes, speeds, (const PElem*)el,
                       (const RunEntry*)pv.runs, (const int64_t*)pv.seg_start, nseg, (double)n_in, pv.seg_off,
                       (const double*)ck, ck_len, pv.hdr);
    }
    if (aux && !lazy) {
      const int64_t n_items = std::max<int64_t>(nseg, max_tiles);
      hipLaunchKernelGGL(k_tile_seg_publish, dim3((unsigned)std::min<int64_t>(ceil_div(n_items, 256), 512)), dim3(256), 0, s,
                         speeds, pv.seg_start, pv.seg_off, nseg, (const double*)ck, ck_len, max_tiles,
                         reinterpret_cast<int64_t*>(ck + ck_len), reinterpret_cast<SegFast*>(ck + ck_len + max_tiles),
                         fused_aux_view(aux, max_out, m).tile_st, fused_aux_view(aux, max_out, m).hdr, pv.hdr, n_items,
                         lazy ? 1 : 0);
      launch_block_rec(speeds, pv, nseg, aux, max_out, m, s);
    }
    PAR_HIP_CHECK(hipGetLastError());
    PAR_HIP_CHECK(hipMemcpyAsync(&h, pv.hdr, sizeof(h), hipMemcpyDeviceToHost, s));
    PAR_HIP_CHECK(hipStreamSynchronize(s));
    if (lazy && h.lazy_fail) {                  // not a curve for a lazy plan: once more, with the per-sample cumsum
      lazy = false;
      --attempt;
      continue;
    }
    if (h.n_long > 0 && !with_long) {           // sparse curve: once more, with the chunked exact cumsum of its long segments
      with_long = true;
      --attempt;
      continue;
    }
    if (h.flags & kFlagCapAmbiguous) {
      // Only the buffer bound is in doubt: settle int(mean * span * 1.01) with numpy's own pairwise order on the host
      // (one D2H of the speed samples); everything else the device computed stands.
      std::vector<double> sp_h(m);
      double ends[2];
      PAR_HIP_CHECK(hipMemcpyAsync(sp_h.data(), speeds, m * sizeof(double), hipMemcpyDeviceToHost, s));
      PAR_HIP_CHECK(hipMemcpyAsync(&ends[0], sampletimes, sizeof(double), hipMemcpyDeviceToHost, s));
      PAR_HIP_CHECK(hipMemcpyAsync(&ends[1], sampletimes + (m - 1), sizeof(double), hipMemcpyDeviceToHost, s));
      PAR_HIP_CHECK(hipStreamSynchronize(s));
      h.cap = (int64_t)((np_pairwise_sum(sp_h.data(), m) / (double)m) * (ends[1] - ends[0]) * 1.01);
      h.flags &= ~kFlagCapAmbiguous;
      PAR_HIP_CHECK(hipMemcpyAsync(pv.hdr, &h, sizeof(h), hipMemcpyHostToDevice, s));
      PAR_HIP_CHECK(hipStreamSynchronize(s));
    }
    if (h.flags == 0) break;
    g_last_plan_flags |= h.flags;
    // a near-tie in the lengths alone: one more round with host-made lengths; anything else (n_i < 2, range,
    // verification, too many crossings, or a second failure): the serial path decides -- it also produces the
    // reference's own diagnosis for genuinely bad curves.
    if (!(attempt == 0 && h.flags == kFlagAmbiguous)) need_host = true;
  }
  if (need_host) {
    int rc = host_plan(pv, sampletimes, speeds, m, n_in, &h, s);
    if (rc != PAR_OK) return rc;
    if (aux) {     // checkpoints + tile map for the serial path's segmentation: same exact device arithmetic
      rc = launch_seg_sums(speeds, pv, nseg, ck, ck_len, max_out, m, s);
      if (rc != PAR_OK) return rc;
      hipLaunchKernelGGL(k_tile_seg, dim3((unsigned)ceil_div(std::max<int64_t>(nseg, max_tiles), 256)), dim3(256), 0, s,
                         speeds, pv.seg_start, pv.seg_off, nseg, (const double*)ck, ck_len, max_tiles,
                         reinterpret_cast<int64_t*>(ck + ck_len), reinterpret_cast<SegFast*>(ck + ck_len + max_tiles),
                         fused_aux_view(aux, max_out, m).tile_st, fused_aux_view(aux, max_out, m).hdr, pv.hdr);
      if ((force_host & 3) == 2) hipLaunchKernelGGL(k_inject_verify_fault, dim3(1), dim3(1), 0, s, pv.hdr);
      hipLaunchKernelGGL(k_publish_ck, dim3(1), dim3(1), 0, s, pv.hdr, ck_len);
      launch_block_rec(speeds, pv, nseg, aux, max_out, m, s);
      PAR_HIP_CHECK(hipGetLastError());
      PAR_HIP_CHECK(hipMemcpyAsync(&h, pv.hdr, sizeof(h), hipMemcpyDeviceToHost, s));
      PAR_HIP_CHECK(hipStreamSynchronize(s));
      g_last_plan_flags |= h.flags;
    }
  } else {
    // the reference writes each segment into its end_guess-sized buffer BEFORE testing the trim (:127-129)
    PAR_REQUIRE(h.written <= h.cap, PAR_ERR_ARG,
                "par_speed_to_pos_plan: positions overflow the reference's end_guess buffer (%lld > %lld); it raises here",
                (long long)h.written, (long long)h.cap);
  }
  if (path_used) *path_used = need_host ? 1 : (host_made_lengths ? 2 : 0);
  if (fused_ok) *fused_ok = aux ? h.ck_valid : 0;
  *len_out = h.len_out;
  *trimmed = h.trimmed;
  return PAR_OK;
}

int par_last_plan_flags(void) { return g_last_plan_flags; }

// force_host != 0 exercises the serial host path (tests use it to cross-check the device scans);
// *path_used = 0 device scans, 1 serial host path, 2 segment lengths from the host (near-tie) and the rest on the device.
int par_speed_to_pos_plan_ex(int device, const double* sampletimes, const double* speeds, int64_t m, int64_t n_in,
                             void* work, size_t work_bytes, int64_t* len_out, int* trimmed, int force_host,
                             int* path_used, void* stream) {
  return plan_impl(device, sampletimes, speeds, m, n_in, work, work_bytes, nullptr, 0, 0, len_out, trimmed, force_host,
                   path_used, nullptr, stream);
}

size_t par_fused_aux_bytes(int64_t max_out, int64_t m) { return par::fused_aux_bytes(max_out, m < 2 ? 2 : m); }

// Plan + cumsum checkpoints (every 8th step of each segment) + tile map in `aux`, for par_varispeed_fused_f32.
// max_out bounds len_out (e.g. 1.02 * n_in); a larger result leaves the plan valid but the fused path refused.
int par_speed_to_pos_plan_fused(int device, const double* sampletimes, const double* speeds, int64_t m, int64_t n_in,
                                void* work, size_t work_bytes, void* aux, size_t aux_bytes, int64_t max_out,
                                int64_t* len_out, int* trimmed, int force_host, int* path_used, int* fused_ok,
                                void* stream) {
  PAR_REQUIRE(aux && max_out > 0, PAR_ERR_ARG, "par_speed_to_pos_plan_fused: aux buffer required");
  return plan_impl(device, sampletimes, speeds, m, n_in, work, work_bytes, aux, aux_bytes, max_out, len_out, trimmed,
                   force_host, path_used, fused_ok, stream);
}

int par_speed_to_pos_plan(int device, const double* sampletimes, const double* speeds, int64_t m, int64_t n_in,
                          void* work, size_t work_bytes, int64_t* len_out, int* trimmed, void* stream) {
  return par_speed_to_pos_plan_ex(device, sampletimes, speeds, m, n_in, work, work_bytes, len_out, trimmed, 0, nullptr,
                                  stream);
}

int par_speed_to_pos_fill(int device, const double* speeds, int64_t m, const void* work, double* pos, int64_t len_out,
                          void* stream) {
  using namespace par;
  PAR_REQUIRE(speeds && work && (pos || len_out == 0) && m >= 2, PAR_ERR_ARG, "par_speed_to_pos_fill: bad args");
  if (len_out == 0) return PAR_OK;
  PAR_HIP_CHECK(hipSetDevice(device));
  return launch_pos_fill(speeds, m, work, pos, len_out, 0, INT64_MAX, as_stream(stream));
}

// Fill from a FUSED plan (par_speed_to_pos_plan_fused with fused_ok): same positions, bit for bit, but parallel over
// 8-sample blocks instead of over segments -- the form to use when the curve has few points.
int par_speed_to_pos_fill_fused(int device, const double* speeds, int64_t m, const void* work, const void* aux,
                                int64_t max_out, double* pos, int64_t len_out, void* stream) {
  using namespace par;
  PAR_REQUIRE(speeds && work && aux && (pos || len_out == 0) && m >= 2 && len_out <= max_out, PAR_ERR_ARG,
              "par_speed_to_pos_fill_fused: bad args");
  if (len_out == 0) return PAR_OK;
  PAR_HIP_CHECK(hipSetDevice(device));
  PlanView pv = plan_view(const_cast<void*>(work), m);
  const int64_t n_slots = (int64_t)fused_ck_len(max_out, m);
  hipLaunchKernelGGL(k_pos_fill_ck, dim3((unsigned)ceil_div(n_slots, 256)), dim3(256), 0, as_stream(stream), speeds,
                     pv.seg_start, pv.seg_off, m - 1, static_cast<const double*>(aux), n_slots, len_out, pos, pv.hdr);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

}  // extern "C"
