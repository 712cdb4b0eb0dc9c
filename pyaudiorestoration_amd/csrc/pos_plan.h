// Plan layout and exact ramp arithmetic shared by K_pos (pos.hip) and the fused K_sinc (sinc.hip).
#pragma once
#include "par_common.h"

namespace par {

// ---------------------------------------------------------------------------------- plan layout
struct PlanHeader {
  int64_t m;
  int64_t len_out;        // trimmed length or total written
  int64_t total_written;  // sum n_i
  unsigned long long trim_seg;   // first segment whose [first,last] straddles n_in, or ~0
  int64_t cap;            // the reference's end_guess buffer size
  int32_t trimmed;
  int32_t flags;          // kFlag*
  int32_t n_direct;
  int32_t n_runs;
  double speed_sum;
  int64_t ck_len;         // checkpoint slots available in the caller's aux buffer (0: none)
  int32_t ck_valid;       // 1: the aux buffer holds this plan's cumsum checkpoints (fused resampler may run)
  int32_t pad2;
  int64_t written;        // outputs the reference has written into its buffer when it stops (end of the trim segment)
  int32_t n_long;         // segments handled by the chunked exact cumsum (sparse curves), see pos.hip
  int32_t pad3;
  unsigned long long first_bad;  // first segment with n_i < 2 (~0: none).  Harmless when the trim fires before it: the
                                 // reference stops there and never builds that segment (k_trim decides)
  int32_t lazy;           // 1: this plan was made WITHOUT the per-sample cumsum (closed-form segment sums, see "lazy plans")
  int32_t lazy_fail;      // != 0: this pass is void and the kernels behind the mark leave at once -- 1 / 2: some segment is outside
                          //       what the closed form vouches for / too many candidates (the caller makes the plan again the
                          //       eager way); 4: the first go of a sparse curve met a long segment (again, with the chunked cumsum)
  int32_t n_cand;         // segments whose sum was recomputed exactly (kMaxCand + 1: list overflow)
  int32_t pad4;
};
static_assert(sizeof(PlanHeader) <= 256, "PlanHeader must fit the reserved header bytes");
constexpr int kFlagAmbiguous = 1;   // a cumulative length is too close to a rounding tie
constexpr int kFlagBadLength = 2;   // some n_i < 2 (reference divides by zero / indexes an empty array)
constexpr int kFlagRange = 4;       // a_i outside the exactly-representable fixed-point range
constexpr int kFlagVerify = 8;      // offset-chain binade prediction failed verification
constexpr int kFlagDirectOverflow = 16;
constexpr int kFlagCapAmbiguous = 64; // end_guess within the device sum's error of an integer: the host path decides
constexpr int kFlagCkOverflow = 32;  // checkpoint buffer too small (plan stays valid; the fused path is refused)
constexpr int kMaxDirect = 2048;
constexpr int kMaxCand = 1 << 16;   // lazy plans: segments whose exact sum the offset chain needs after all (~5e3 per 60-min file)
constexpr unsigned long long kNoTrim = ~0ull;

struct U128 {
  unsigned long long hi, lo;       // value = hi + lo * 2^-64
};
struct PElem {                      // parity-dependent translation (+ segmented-scan head flag)
  long long c0, c1;                 // increment when the incoming integer is even / odd
  long long head;                   // 1: a run starts at this element (scan restarts here)
  long long pad;
};
struct RunEntry {
  long long start;                  // first segment of the run
  double x;                         // offset at that segment (bit-exact)
};

struct Cand {
  long long i;          // segment
  double xa, xb;        // the approximate offsets around it (binade prediction of its chain element)
};
static_assert(sizeof(Cand) == 24, "plan_bytes counts 24 bytes per candidate");

constexpr size_t kHdrBytes = 256;
struct PlanView {
  PlanHeader* hdr;
  int64_t* seg_start;   // [m]   seg_start[i] = outputs before segment i; [m-1] = total
  double* seg_off;      // [m]   offset chain; seg_off[i] = position offset of segment i; [m-1] = final
  double* S;            // [m]   per-segment reciprocal sums
  double* xs;           // [m]   approx offsets (plain f64 scan)
  char* scan;           // [m * 32] scan elements (U128 then PElem)
  char* bsum;           // block sums for the scans
  long long* direct;    // [kMaxDirect] indices of direct (binade-crossing) steps
  RunEntry* runs;       // [kMaxDirect + 2]
  Cand* cand;           // [kMaxCand] lazy plans: candidates for the exact sum
};
inline size_t scan_blocks(int64_t n) { return (size_t)((n + 1023) / 1024); }
inline size_t plan_bytes(int64_t m) {
  return kHdrBytes + (size_t)m * (8 + 8 + 8 + 8 + 32) + (scan_blocks(m) + 8) * 32 + kMaxDirect * 8 +
         (kMaxDirect + 2) * sizeof(RunEntry) + 256 + (size_t)kMaxCand * 24;
}
inline PlanView plan_view(void* work, int64_t m) {
  char* b = static_cast<char*>(work);
  PlanView v;
  v.hdr = reinterpret_cast<PlanHeader*>(b);
  b += kHdrBytes;
  v.seg_start = reinterpret_cast<int64_t*>(b);
  b += (size_t)m * 8;
  v.seg_off = reinterpret_cast<double*>(b);
  b += (size_t)m * 8;
  v.S = reinterpret_cast<double*>(b);
  b += (size_t)m * 8;
  v.xs = reinterpret_cast<double*>(b);
  b += (size_t)m * 8;
  v.scan = b;
  b += (size_t)m * 32;
  v.bsum = b;
  b += (scan_blocks(m) + 8) * 32;
  v.direct = reinterpret_cast<long long*>(b);
  b += kMaxDirect * 8;
  v.runs = reinterpret_cast<RunEntry*>(b);
  b += ((kMaxDirect + 2) * sizeof(RunEntry) + 255) / 256 * 256;
  v.cand = reinterpret_cast<Cand*>(b);
  return v;
}


// ---------------------------------------------------------------------------------- ramp arithmetic
struct Ramp {
  double s0, ds, nm1, y;     // y = RN(1/(n-1))
  bool fast;                 // both segment speeds in [2^-200, 2^200]: recip_unscaled is exact IEEE division
};

// RN(1/b) by the instruction sequence the compiler emits for an IEEE float64 division (v_rcp_f64, two Newton
// steps, residual correction) WITHOUT its v_div_scale / v_div_fmas scaling / v_div_fixup wrapper: for a normal b
// far from the exponent limits those three pass their operands through unchanged, so the result is
// bit-identical at 7 instead of 11 float64 instructions (measured: IEEE divide 64 cycles per wave).
__device__ __forceinline__ double recip_unscaled(double b) {
  double x = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, x, 1.0);
  x = __builtin_fma(x, e, x);
  e = __builtin_fma(-b, x, 1.0);
  x = __builtin_fma(x, e, x);
  const double r = __builtin_fma(-b, x, 1.0);      // numerator 1.0: the quotient estimate 1.0*x is x itself
  return __builtin_fma(r, x, x);
}
__device__ __forceinline__ Ramp make_ramp(double s0, double s1, long long n) {
#pragma clang fp contract(off)
  Ramp r;
  r.s0 = s0;
  r.ds = s1 - s0;
  r.nm1 = (double)(n - 1);
  r.y = 1.0 / r.nm1;
  // every ramp value lies between the two speeds (up to one rounding), so this range check covers them all
  const double lo = s0 < s1 ? s0 : s1, hi = s0 < s1 ? s1 : s0;
  r.fast = lo >= 0x1p-200 && hi <= 0x1p200;
  return r;
}
// 1 / (k/(n-1) * ds + s0), every operation individually rounded like numpy (:120, :125).
// k/(n-1) by Markstein's theorem: with y = RN(1/b), q0 = RN(a*y), r = a - b*q0 (exact, fma),
// q = RN(q0 + r*y) is the correctly rounded a/b  (checked exhaustively for n <= 6000 on the host).
// `a` is the step index k as a double (exact below 2^53); callers that walk consecutive k pass a0 + u with a0
// converted once, which replaces a 64-bit integer conversion per step by one exact float64 add.
__device__ __forceinline__ double ramp_value(double a, const Ramp& r) {
#pragma clang fp contract(off)   // block scope: also holds when included from files built with contraction on
  const double q0 = a * r.y;
  const double rem = __builtin_fma(-q0, r.nm1, a);
  const double q = __builtin_fma(rem, r.y, q0);
  return q * r.ds + r.s0;                     // not fused (-ffp-contract=off)
}
__device__ __forceinline__ double ramp_recip(double a, const Ramp& r) {
  const double bs = ramp_value(a, r);
  return r.fast ? recip_unscaled(bs) : 1.0 / bs;
}


// ---------------------------------------------------------------------------------- lazy plans (r05)
// The reference's position chain needs, per curve segment, S_i = the LAST element of np.cumsum(1 / block_speeds) -- n_i
// correctly rounded divisions and n_i sequential float64 adds (691 M of each for the 60-min file: the plan's k_seg_sum,
// 0.4 of the 0.9 ms the plan costs a concurrent K_sinc), plus a checkpoint of the running sum every 8 steps (0.7 GB written)
// so that K_sinc can redo an output with the reference's own arithmetic.  But the bits of S_i only matter where a rounding
// DECIDES something:
//   * the offset chain x_{i+1} = fl(x_i + S_i) rounds to a grid of ulp(x) ~ 1e-7 (at 7e8), S_i's own uncertainty is ~1e-11:
//     a closed-form S~_i with a rigorous error bound B_i settles the chain element of every segment whose S~_i/ulp lies
//     farther than B_i from a rounding boundary (k_offs_elements) -- all but ~5e3 of the 2.7 M segments of the 60-min
//     curve; those CANDIDATES (and the ~50 binade-crossing steps) get the exact sequential sum (k_seg_exact_list) and the
//     chain is bit-identical to numpy's;
//   * K_sinc needs rint(p) exact and the shift to ~1e-7: outputs whose closed-form position lies within the bound of a
//     half-integer (~1 in 10^6) walk the segment's cumsum from its first step with the reference's own arithmetic
//     (place_exact, sinc.hip); nobody else needs a checkpoint.
// A plan is lazy only if EVERY segment qualifies: 2 <= n <= kLazyMaxN outputs, speeds in [1/16, 64], |s1 - s0| <= 2^-9 of
// the smaller one (k_seg_sum_lazy); otherwise the caller makes it again the eager way (sparse, stepped or wild curves).
constexpr long long kLazyMaxN = 1024;
constexpr int kTileLazy = 4;         // TileHdr.flags: the plan holds no cumsum checkpoints
// sum_{k=0}^{K-1} 1 / (s0 + step k): expansion about the midpoint speed m (odd powers cancel),
//   (K/m) [1 + t (K^2-1)/12 (1 + t (3K^2-7)/20)],  t = (step/m)^2;  next term < K (step K / 2m)^6 / 7 < 1.5e-19 K here.
__device__ __forceinline__ double lazy_prefix(double s0, double step, double K) {
#pragma clang fp contract(off)
  const double m = __builtin_fma(step, 0.5 * (K - 1.0), s0);
  const double rc = 1.0 / m;
  const double z = rc * step, t = z * z, K2 = K * K;
  const double corr = t * (K2 - 1.0) * (1.0 / 12.0) * (1.0 + t * (3.0 * K2 - 7.0) * 0.05);
  return K * rc * (1.0 + corr);
}
// |lazy_prefix - numpy's sequentially rounded cumsum after K terms|: K adds each within 2^-53 of a running sum <= j / smin
// (0.5 K^2), every term within 2.01 * 2^-53 / smin of 1 / (s0 + step k) (three roundings of the ramp value, one of the
// reciprocal), ~6 ulp of the closed form's own evaluation (12 K); the rest is margin.
__device__ __forceinline__ double lazy_bound(double K, double smin) {
  return (0.55 * K * K + 16.0 * K + 64.0) * 0x1p-53 / smin;
}
__device__ __forceinline__ bool lazy_segment_ok(long long n, double s0, double s1) {
  const double lo = s0 < s1 ? s0 : s1, hi = s0 < s1 ? s1 : s0;
  return n >= 2 && n <= kLazyMaxN && lo >= 0.0625 && hi <= 64.0 && (hi - lo) <= 0x1p-9 * lo;    // false for NaN
}

// ---------------------------------------------------------------------------- cumsum checkpoints
// The fused resampler regenerates positions inside K_sinc from per-segment checkpoints of the running
// reciprocal sum: slot(i, b) holds cumsum after step kCk*b - 1 of segment i (b >= 1; b = 0 is 0.0).
// slot(i, b) = start_i/kCk + i + b is unique and monotone without any prefix sum, because a segment of n
// samples owns ceil(n/kCk) <= n/kCk + 1 slots.
constexpr int kCk = 8;
__host__ __device__ inline long long ck_slot0(long long seg_start, long long i) { return seg_start / kCk + i; }
// Per-segment record of a fused plan: what K_sinc needs to place an output of that segment WITHOUT walking the
// cumsum -- position = rint(seg_off) + [foff + checkpoint + closed-form sum of the <= kCk reciprocals behind it]
// (midpoint rule + second-order term; remainder < 2e-10 when `fast`), see place_fast in sinc.hip.
struct SegFast {
  double foff;              // seg_off - rint(seg_off), exact
  long long A;              // rint(seg_off)
  double step;              // (s1 - s0) / (n - 1): ramp increment per output
  int n;                    // outputs in the segment
  int fast;                 // 1: closed-form placement valid (ramp gentle enough, speeds in range, n < 2^31)
                            // 2: also the per-block polynomial (BlockRec) WITH the cubic term K_sinc derives from e2
                            // 3: the block quadratic alone is good to 1e-8 samples
};
static_assert(sizeof(SegFast) == 32, "SegFast is loaded as two 16-byte words");
// Block record of a fused plan: one per kRec = 32 consecutive outputs (absolute index j = 32 g + u).  Within the block the
// positions are a polynomial in the CENTRED variable u' = u - 16 (the speed ramp is linear, so the reciprocal increments
// are linear to second order):
//     p = anchor_T + Irel + F + u' (1 + e1) + u'^2 e2 [+ u'^3 e3]          u' = -16 .. 15
// with Irel the integer part of the centre output's position relative to the tile's anchor (16 bits: a tile spans ~10^3
// input samples where the model applies) and |F| <= 1/2.  Centred, |u' e1| <= 1/2 where the model applies (|e1| <= 1/32),
// so the float32 evaluation stays within ~1.2e-7 of the float64 polynomial.  e3 = (4/3) e2^2 / (1 + e1) follows from e2
// (both come from the one ramp slope) and is only evaluated where the record says so (`cubic`: steeper ramps, e.g.
// flutter at low sample rates); the plan folds the quadratic and linear parts of the cubic sum into e2 and e1.
// 16 bytes per 32 outputs (r02: per 8): K_sinc reads nothing else per output -- no segment lookup, no float64.
// Word 0: Irel in the high 16 bits (signed), flags in the low 16:
//   bits 0-4  ustar - 1: outputs u >= ustar belong to the NEXT segment and use the block's second piece (ustar = 32: none)
//   bit  5    E0: output u = ustar - 1 is the last one of its segment (always so in a boundary block; in an interior block
//             only when the segment ends with the block)
//   bit  6    E1: the second piece's segment ends with the block's last output
//   bit  7/8  slow0 / slow1: that piece is not covered by the model (steep ramp, speed far from 1, very short segments,
//             |p| out of range, the file's last output): its outputs are placed by place_fast / place_exact instead
//   bit  9    cubic: evaluate the e3 term (either piece)
//   bits 10-15 lastu: the u of the E0 output, 63 if none (what the streaming kernel reads instead of E0 / E1; a block with
//             an E1 output as well is flagged slow1)
// The last output of a segment matters because its period to the next position is the PREVIOUS increment (the next
// segment starts at this ramp's end speed).
// The second piece (same layout, its own polynomial in the same u') lives in a parallel array that only boundary
// blocks touch; its word 0 carries only Irel.
constexpr int kRec = 32;
constexpr int kRecShift = 5;
struct BlockRec {
  unsigned w0;              // Irel << 16 | flags
  float F, e1, e2;
};
static_assert(sizeof(BlockRec) == 16, "BlockRec is one 16-byte word");
constexpr int kRecLastShift = 10;     // bits 10-15: u of the output that ends a segment inside the block (63: none)
constexpr unsigned kRecE0 = 1u << 5, kRecE1 = 1u << 6, kRecSlow0 = 1u << 7, kRecSlow1 = 1u << 8, kRecCubic = 1u << 9;
typedef BlockRec BlockRec2;
// Tile header: everything K_sinc needs before it can stage a tile's input span, in ONE scalar load.
struct TileHdr {
  long long anchor;         // even integer within a few samples of the tile's first position (written by k_tile_seg)
  long long c_last;         // (unused)
  long long iT;             // segment of the tile's first output
  int mn_rel;               // (unused)
  int flags;                // 1: positions out of range (tile takes the float64 path); kTileMayUnity: see k_tile_seg
};
constexpr int kTileMayUnity = 2;
constexpr int kTileMaySlow = 8;      // some segment under the tile (or the one behind it) touches speed < 1 + 2e-7 (or is not a number): the
                                     // tile may hold fc < 1 outputs.  K_sinc's streaming launch sorts its streams by it (k_sinc_pipe's KIND)
constexpr double kSlowHintBelow = 1.0000002;     // float32 period - 1 stays <= 0 at and above this speed
constexpr double kUnityHintBelow = 0.9999998;    // float32 period - 1 rounds to 0 well above this speed
static_assert(sizeof(TileHdr) == 32, "TileHdr is one s_load_dwordx8");
constexpr int kBlocksPerTile = (int)(kSincTileOutputs / kRec);
constexpr int kTileStarts = 6;     // seg_start of the tile's first segment and the five behind it (k_block_rec's lookup)
// aux buffer of a fused plan: [ck_len checkpoints (f64)] [tile map (int64)] [m SegFast] [tiles TileHdr] [blocks BlockRec]
// [blocks BlockRec2 (sparse)] [tiles x kTileStarts int64] [16 x int32: redo count ...] [tiles x int32: redo list]
// The redo list is K_sinc's own scratch: tiles the streaming kernel (sinc2.hip) hands to the block kernel, rebuilt by every
// launch -- one K_sinc launch per plan at a time.
inline size_t fused_ck_len(int64_t max_out, int64_t m) { return (size_t)(max_out / kCk + m + 16); }
inline size_t fused_tiles(int64_t max_out) { return (size_t)(max_out / kSincTileOutputs + 4); }
inline size_t fused_blocks(int64_t max_out) { return fused_tiles(max_out) * kBlocksPerTile; }
inline size_t fused_aux_bytes(int64_t max_out, int64_t m) {
  return (fused_ck_len(max_out, m) + fused_tiles(max_out)) * 8 + (size_t)m * sizeof(SegFast) +
         fused_tiles(max_out) * (sizeof(TileHdr) + kTileStarts * 8) +
         fused_blocks(max_out) * (sizeof(BlockRec) + sizeof(BlockRec2)) + 64 + fused_tiles(max_out) * 4;
}
struct FusedAux {            // views into the aux buffer
  double* ck;
  int64_t* tile_seg;
  SegFast* seg_fast;
  TileHdr* hdr;
  BlockRec* rec;
  BlockRec2* rec2;
  long long* tile_st;
  int* redo_count;           // [16]: [0] tiles in the list
  int* redo_list;
};
inline FusedAux fused_aux_view(void* aux, int64_t max_out, int64_t m) {
  FusedAux v;
  v.ck = static_cast<double*>(aux);
  v.tile_seg = reinterpret_cast<int64_t*>(v.ck + fused_ck_len(max_out, m));
  v.seg_fast = reinterpret_cast<SegFast*>(v.tile_seg + fused_tiles(max_out));
  v.hdr = reinterpret_cast<TileHdr*>(v.seg_fast + m);
  v.rec = reinterpret_cast<BlockRec*>(v.hdr + fused_tiles(max_out));
  v.rec2 = reinterpret_cast<BlockRec2*>(v.rec + fused_blocks(max_out));
  v.tile_st = reinterpret_cast<long long*>(v.rec2 + fused_blocks(max_out));
  v.redo_count = reinterpret_cast<int*>(v.tile_st + fused_tiles(max_out) * kTileStarts);
  v.redo_list = v.redo_count + 16;
  return v;
}

}  // namespace par
