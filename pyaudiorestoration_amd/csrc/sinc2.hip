// K_sinc, streaming form -- the north-star kernel: NT = 32 files, mono on unit strides (k_sinc_pipe<1, 2> + <1, 1>: two kernels by
// stream kind, r06), the two channels of an interleaved stereo file (k_sinc_pipe<2, 0>, r05), one channel of such a file (<2, 3>,
// r06); k_sinc_pipe_n<...>: the same bodies over the streams of up to eight files in one launch (r06).
//
// Semantics: resampling.sinc_core (reference util/resampling.py:51-90), fused with the speed curve like k_sinc_fused (sinc.hip).
//
// Shape.  ONE WAVE = one worker that streams over 9-23 (an odd number of) consecutive 1024-output tiles of the file; waves never meet (no
// workgroup, no barrier).  A pass takes the next 128 outputs (two per lane), places them from the plan's block records, and
// evaluates their windows against ONE 128-centre stretch of the input grid [ws, ws + 128), ws = the first centre rounded down
// to 8; outputs whose centre lies beyond it (a handful: periods differ from 1 by <= 1.25 %) simply open the next pass.
//   * taps 3 <= |n| <= 31 at fc = 1: a Farrow bank in q = shift^2 (minimax polynomials of (win_n/pi)/(n^2 - q),
//     tools/sinc2_model.py) -- six fixed FIR filters on the input grid, evaluated for the pass's 128 centres on the matrix cores
//     (v_mfma_f32_16x16x32_f16, signal and dominant filter pair split float16 hi + lo 2^-12: 13 MFMAs since r06); the bank goes through
//     the wave's own LDS and every output gathers its centre's row;
//   * fc < 1 (read head slower than the output clock) = the fc = 1 result + a correction in g = 1 - fc whose taps are ENTIRE
//     functions of g (n - s): all 63 taps enter through seven fixed MOMENT filters m_i = sum_n (-1)^n win_n (n/32)^i x[c + n]
//     on the same image (18 more MFMAs; 15 to order 5 where every lane of the pass has g <= 0.0105, r06) and a complex Horner in 32 pi g per output (tools/sinc3_model.py; 1e-7 of the peak for
//     g <= 0.0101, 5e-7 to 0.0125; steeper tiles go to the block kernel).  One image, per-lane g exact, no restarts;
//   * the 25 constant fragments of both filter sets RESIDENT IN REGISTERS for the life of the wave (the fc = 1 kernel <1, 1>: its 10);
//   * taps |n| <= 2 on the vector units with the lane's exact shift (two reciprocals per output);
//   * the input streams through a ring per wave (8 chunks of 128 float32 samples, direct-to-LDS loads two passes ahead),
//     converted ONCE to float16 hi / lo images for the banks -- no halo is ever re-read or re-converted; block records of the
//     passes ahead are fetched while the current one computes.
// What the record model or float16 do not cover goes to the block kernel through a tile list (k_sinc_fused_list / _list2,
// sinc.hip): blocks flagged slow, outputs within the reference's own rounding of a half-integer position (window-centre ties),
// input that is non-finite / >= 32 / all but silent.  The file's END tiles (the ring would reach over the file's ends) are
// not streamed at all: the launch's first workgroups do them the block kernel's way (fused_wave, sinc_block.h) beside the
// streams.  Every window centre is the reference's rint(p) either way.
// Stereo (k_sinc_pipe<2>): the ring holds frames (left, right); a pass is placed ONCE and both channels are converted, banked
// and gathered for it -- in ONE set of bank rows, which the channels take turns in (the order of the loop's stages and the
// one-chunk lag of channel 1's image: see the loop) so that eight streams still fit a compute unit's LDS; outputs leave as
// 8-byte frames.  177 -> 133 vector instructions per channel-sample.
// History: r04 built this kernel in four forms (a pass as a chain of stages; the pipelined loop with the fc < 1 taps as two
// MODULATED images of the signal; an fc = 1-only kernel at three waves per SIMD; the moment form, also as workgroups of several
// streams sharing the constants through LDS).  NOTES r04 has their numbers; the product carries the one that won.
#define PAR_WANT_BANK2 1
#include "par_common.h"
#include "pos_plan.h"
#include "sinc_taps_gen.h"
#include "sinc_common.h"
#include "sinc_block.h"           // fused_wave: the file's end tiles are done the block kernel's way by the launch's first workgroups
#include <algorithm>
#include <atomic>
#include <limits.h>
#include <type_traits>

constexpr int kMaxTilesPerWave = 23;          // k_sinc_pipe's tiles per wave for long files (launch_sinc_stream picks 8 .. this)
#ifndef PAR_S2_EXP
#define PAR_S2_EXP 0            // timing builds, never shipped: 1 no MFMAs, 2 no near taps, 4 no stores, 8 no conversion, 16 unity maths on every pass
#endif

namespace par {

typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int kRing = 512;                       // samples per ring (float32 and float16 images alike)
constexpr int kPass = 128;                       // centres per pass = outputs tried per pass
// The float16 images hold the signal x kImgScale (r06).  A sample's hi part is zero -- and the sample then reaches only the
// filters that take the lo image -- below 6e-8 instead of 6e-5, i.e. practically never: passages 60 dB down lose nothing (their
// zero crossings used to), and the (e1 d1) filters no longer need the lo image at all (2 MFMAs per pass; without the scale a
// signal at 1e-3 came out 4.5e-5 wrong, tools/sinc2_model.py).  The price is the streaming kernel's range: |x| < 32 (beyond it the
// lo part of the scaled sample leaves float16; such tiles are the block kernel's).
constexpr float kImgScale = 1024.0f, kImgScaleInv = 0.0009765625f;
constexpr float kImgMax = 32.0f;                 // |x| the images can hold
// A chunk whose loudest sample is below 2^-13 (and not 0) is the block kernel's, as before the scale: a sample with a zero hi part
// (below 6e-8) loses its (e1 d1) and m2.. contributions, ~2e-3 of what it adds to an output -- 1.2e-10, which must stay below
// 1e-6 of the passage's own level (a file at 1e-6 came out 1.9e-4 wrong with the threshold scaled down, test_unity_path_matrix_core_bank)
constexpr float kQuiet = 0.0001220703125f;


#if PAR_S2_EXP & 64
__device__ unsigned long long* g_s2_phase;       // [waves][16] cycle sums per phase (timing builds only)
#define S2_MARK(k)                                              \
  do {                                                          \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    ph_[k] += now_ - pt_;                                       \
    pt_ = now_;                                                 \
  } while (0)
#else
#define S2_MARK(k) do { } while (0)
#endif

struct S2Args {
  int64_t len_out;
  const float* sig;
  int64_t len_in;
  float* out;
  const TileHdr* hdr;
  const BlockRec* rec;
  const BlockRec2* rec2;
  int* redo_count;
  int* redo_list;
  int tiles;                                     // tiles per wave (k_sinc_pipe; <= 48: a lane per tile header)
  int tiles_tail;                                // ... of the streams behind the first n_big (the launch's last round: short
  int64_t n_big;                                 // streams, so that the GPU does not idle behind a few long ones)
  int64_t n_full;                                // full tiles of the file
  int64_t n_tiles;                               // tiles with a header
  // The file's END tiles -- the first (its ring would reach in front of the file), the last two full ones (... behind it) and the
  // partial one -- are not streamed: the launch's first n_edge workgroups do them the block kernel's way (fused_wave, sinc_block.h),
  // 128 outputs each, BESIDE the streams.  (Through the tile list they cost 50 us behind every launch: cold code -- the masked
  // path of the last wave, the float64 slow path of the first outputs -- fetched by one wave while the GPU idles, r05.)
  int n_edge;
  int64_t out_stride;                            // k_sinc_pipe<2, 3> only (one channel of frames): elements between two outputs
  FusedArgs fa;
  const float4* tab;
  TapModes tmd;
};
constexpr int kEdgeWaveOut = 128;                                   // outputs per end-tile workgroup (one wave)
constexpr int kEdgeWavesPerTile = kSincTileOutputs / kEdgeWaveOut;
constexpr int kTileEdge = 0x100;                                    // (kernel-local header flag: an end tile, nobody pushes it)

__device__ __forceinline__ float sinpi_poly(float z) {       // sin(pi x) / x as a polynomial in z = x^2, |x| <= 0.52
  float p = -0.00737043094f;
  p = fmaf(p, z, 0.0821458866f);
  p = fmaf(p, z, -0.599264529f);
  p = fmaf(p, z, 2.55016404f);
  p = fmaf(p, z, -5.16771278f);
  return fmaf(p, z, 3.14159265f);
}

// hi part of the float16 split.  The matrix cores flush subnormal float16 operands to zero (measured: a modulated image lost
// its samples below 6.1e-5 near the modulator's zero crossings, 2e-5 of the peak): below float16's normal range the hi part is
// zero and the lo part (x 4096) carries the value.  (Switching the wave to flush float16 subnormals, MODE.FP_DENORM[3:2] = 0,
// would make the conversions do this for free -- tools/exp/denorm_mode.hip -- but the bank's packed e2 / d2 halves of a quiet
// passage ARE subnormal and must survive: 7.7e-5 of the block's peak on the 'loud next to quiet' test with the mode set.)
__device__ __forceinline__ _Float16 hi16(float x) { return (_Float16)(fabsf(x) < 6.103515625e-05f ? 0.0f : x); }
#define S2_HI(x) hi16(x)

__device__ __forceinline__ unsigned pack_h2(float a, float b) {
  const half2v h = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float h_lo(unsigned w) { return (float)__builtin_bit_cast(half2v, w)[0]; }
__device__ __forceinline__ float h_hi(unsigned w) { return (float)__builtin_bit_cast(half2v, w)[1]; }


// Direct-to-LDS loads as inline assembly: LDS address = M0 + 4 (or 16) x lane.  Through the compiler's builtin every later LDS
// read that might alias the destination gets an s_waitcnt vmcnt(0) in front of it -- the very latency the stream is built to
// hide.  Issued this way the compiler does not count them; the kernel waits for them itself (one vmcnt(0) at the head of a
// pass, when they are a whole pass old).
__device__ __forceinline__ void dma_dword(const float* gptr, const void* lds) {
  const unsigned la = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lds;
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" ::"v"(gptr), "s"(__builtin_amdgcn_readfirstlane(la)) : "memory", "m0");
}
__device__ __forceinline__ void dma_dwordx4(const void* gptr, const void* lds) {
  const unsigned la = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lds;
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(__builtin_amdgcn_readfirstlane(la)) : "memory", "m0");
}

// 128 consecutive floats from a wave-uniform address: two direct loads off a scalar base (per-lane byte offset 4 l; the instruction
// offset advances the global and the LDS address alike)
__device__ __forceinline__ void dma_chunk128(const float* base, unsigned lane_bytes, const void* lds) {
  const unsigned la = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lds;
  const unsigned long long b = (unsigned long long)(uintptr_t)base;
  const unsigned long long bs = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);       // (the value IS wave-uniform)
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dword %0, %1\n\tglobal_load_lds_dword %0, %1 offset:256"
               ::"v"(lane_bytes), "s"(bs), "s"(__builtin_amdgcn_readfirstlane(la)) : "memory", "m0");
}

// 128 stereo frames (256 consecutive floats): four direct loads
__device__ __forceinline__ void dma_chunk256(const float* base, unsigned lane_bytes, const void* lds) {
  const unsigned la = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lds;
  const unsigned long long b = (unsigned long long)(uintptr_t)base;
  const unsigned long long bs = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dword %0, %1\n\tglobal_load_lds_dword %0, %1 offset:256\n\t"
               "global_load_lds_dword %0, %1 offset:512\n\tglobal_load_lds_dword %0, %1 offset:768"
               ::"v"(lane_bytes), "s"(bs), "s"(__builtin_amdgcn_readfirstlane(la)) : "memory", "m0");
}

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Lane sets of a pass are PREFIXES of its 128 outputs (indices and window centres increase with the lane): they live as
// counts in scalar registers, a lane tests one with a single compare, and the set algebra is SALU work.
__device__ __forceinline__ int clamp64(int n) { return n < 0 ? 0 : (n > 64 ? 64 : n); }
__device__ __forceinline__ unsigned long long prefix(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }   // 0 <= n <= 64

// Placement of one row (64 consecutive outputs, lane l) from its block records (pos_plan.h BlockRec): window centre relative
// to A0, sub-sample shift, max(period - 1, 0); `bad` = the lane's block leaves the plain record model (a slow piece, the cubic
// term) or its centre lies within the reference's own rounding of a tie -- such tiles go to the block kernel.
struct S2Row {
  int c;
  float s, ep;
  bool bad;
};
template <bool SECOND = true>                    // false: the caller knows that no lane of the row takes its block's second piece
__device__ __forceinline__ S2Row s2_place_row(const uint4 ra, const uint4 rb, const int u, const int cbase, const float uf,
                                              const float u2f, const float tw1, const float tw0, const float tolf) {
  const unsigned m = ra.x;
  const bool second = SECOND && (unsigned)u > (m & 31u);
  const float F = __uint_as_float(second ? rb.y : ra.y), e1 = __uint_as_float(second ? rb.z : ra.z),
              e2 = __uint_as_float(second ? rb.w : ra.w);
  const int irel = (int)(second ? rb.x : ra.x) >> 16;
  const bool last = (unsigned)u == ((m >> kRecLastShift) & 63u);
  const float frac = fmaf(u2f, e2, fmaf(uf, e1, F));
  const float e = fmaf(e2, last ? tw0 : tw1, e1);               // period to the next position, minus 1
  const float ri = rintf(frac);
  S2Row o;
  o.s = frac - ri;
  o.c = cbase + irel + (int)ri;
  o.ep = __builtin_amdgcn_fmed3f(e, 0.0f, 3.0e38f);             // max(e, 0) in one instruction
  o.bad = (m & (kRecSlow0 | kRecSlow1 | kRecCubic)) != 0u || !(fabsf(fabsf(o.s) - 0.5f) > tolf);
  return o;
}


// ------------------------------------------------------------------------------------------------------------------------
// The loop.  Run as a chain -- records -> placement -> conversion -> bank -> gather, a wait for LDS or memory between every two
// links -- a pass took 4 360 cycles for ~350 instructions at the two waves per SIMD the registers allow (r04, first shape).
// Here the links of one iteration belong to DIFFERENT passes:
//     iteration k:   OUT(k)        gather + near taps + stores of the pass placed and banked in iteration k - 1
//                    PLACE(k + 1)  from records that set out in iteration k - 2
//                    BANK(k + 1)   over the centres PLACE(k + 1) just named, from image chunks converted in iterations <= k - 1
//                    CONV          the next 128-sample chunk of the ring (fetched in iteration k - 2) -> float16 images
//                    FETCH         chunk + 2 of the ring and the records of pass k + 3 set out (direct-to-LDS loads)
// so nothing an iteration reads from LDS was written in the same iteration: ONE wave-level fence per iteration, one
// s_waitcnt vmcnt(5) (everything but the previous iteration's five memory operations has landed), no branch in the body.
// The ring holds 8 chunks of float32 samples (near taps) and 4 of the float16 images; the bank is single-buffered (an
// iteration's gathers precede its bank writes in program order, LDS serves a wave's operations in order).
// Whatever is not the plain case -- the first pass of a run, a pass that meets a flagged tile or block, the end of the
// wave's range, a change of tap regime, a conversion window that has drifted out of its slack -- leaves the loop
// and is done by start_run(): the same stages, one after the other with full waits, which also primes the loop again.
constexpr int kRingF = 1024;                      // float32 samples in the ring (8 chunks)
constexpr int kRingH = 512;                       // samples per float16 image (4 chunks)

// A stream's LDS.  NCH = 2: an interleaved stereo file -- the ring holds frames (left, right), each channel has its own float16
// images, and the two channels take turns in ONE set of bank rows (see the loop of k_sinc_pipe).
// MOM = false: the stream of a kernel that only knows fc = 1 passes (k_sinc_pipe<1, 1>) -- no moment rows, the packed e2 | d2
// halves in an array of their own: 9.8 KB, sixteen streams per compute unit.
template <int NCH, bool MOM = true>
struct S3Lds {
  static constexpr bool kMoments = MOM;
  float ring_head[4 * NCH];                      // frames -2, -1 mirror frames 1022, 1023
  float ring[kRingF * NCH];
  float ring_tail[4 * NCH];                      // mirrors frames 0 .. 3
  _Float16 img[2 * NCH][kRingH];                 // x hi, lo x 4096 (stereo: [2 ch], [2 ch + 1])
  float4v qa[kPass];                             // bank rows {e0, d0, e1, d1}, slot = ci ^ ((ci >> 3) & 7)
  float4v qm0[MOM ? kPass : kPass / 4];          // moment rows {m0, m1, m2, e2|d2 (halves)} (same slots); !MOM: e2|d2 alone, a float per slot
  float4v qm1[MOM ? kPass : 1];                  // {m3, m4, m5, m6}
  __device__ __forceinline__ float& e2d2(int sl) {
    return MOM ? reinterpret_cast<float*>(&qm0[sl])[3] : reinterpret_cast<float*>(&qm0[0])[sl];
  }
  __device__ __forceinline__ const float& e2d2(int sl) const {
    return MOM ? reinterpret_cast<const float*>(&qm0[sl])[3] : reinterpret_cast<const float*>(&qm0[0])[sl];
  }
#ifdef PAR_S3_LDS_PAD
  uint4 pad[PAR_S3_LDS_PAD / 16];                // (occupancy experiments)
#endif
  uint4 recs[4][16];                             // block records of four passes: [0..7] first pieces, [8..15] second pieces
};
static_assert(offsetof(S3Lds<1>, img) % 16 == 0 && offsetof(S3Lds<1>, qa) % 16 == 0 && offsetof(S3Lds<1>, recs) % 16 == 0, "16-byte aligned");
static_assert(offsetof(S3Lds<2>, ring) % 16 == 0 && offsetof(S3Lds<2>, img) % 16 == 0 && offsetof(S3Lds<2>, qa) % 16 == 0 &&
              offsetof(S3Lds<2>, recs) % 16 == 0, "16-byte aligned");
static_assert(sizeof(S3Lds<2>) <= 20480, "eight stereo streams per compute unit (160 KB of LDS)");
using S3LdsUnity = S3Lds<1, false>;
static_assert(sizeof(S3LdsUnity) <= 10240 && offsetof(S3LdsUnity, recs) % 16 == 0 && offsetof(S3LdsUnity, qa) % 16 == 0,
              "sixteen fc = 1 streams per compute unit");

struct S3Pass {                                  // a placed pass: 128 candidate outputs j .. j + 127 (two per lane)
  int c[2];                                      // window centre relative to A0
  float s[2], ep[2];                             // shift, max(period - 1, 0)
  int nok[2];                                    // lanes of row r the pass finishes: l < nok[r]   (wave-uniform)
  int j, ws;                                     // first output (relative to Ja), first bank centre (wave-uniform)
};

// fc = 1 bank AND the seven moment filters of the fc < 1 correction over the same 128 centres, from the same signal fragments:
// 15 + 18 MFMAs.  fr: the fc = 1 bank's ten constant fragments (kBank2Frags32's first ten), fmr: the moment filters' fifteen
// (kBank3Frags32, sinc_taps_gen.h); all resident in the wave's registers.  MOMENTS = false: the fc = 1 bank alone.
constexpr int kCtabUnity = 10;
// SIX = false: without the moment of order 6 (passes with 1 - fc <= 0.0105: the series to order 5 is within 1.6e-6 there)
template <bool MOMENTS, bool SIX = true, class LDS>
__device__ __forceinline__ void bank_image3m(LDS& L, const half8v (&fr)[kBank2Frags], const half8v (&fmr)[kBank3Frags],
                                             const int offs, const int l, const int ch = 0) {
  const int bb = l & 15, g = l >> 4;
  const int i0 = offs + 8 * bb + 8 * g;
  half8v xh[3], xl[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    const int ix = (i0 + 32 * ks) & (kRingH - 1);
    xh[ks] = *reinterpret_cast<const half8v*>(&L.img[2 * ch][ix]);
    xl[ks] = *reinterpret_cast<const half8v*>(&L.img[2 * ch + 1][ix]);
  }
  auto frag = [&](int f) { return f < kCtabUnity ? fr[f] : fmr[f - kCtabUnity]; };
  const float4v z = {0.0f, 0.0f, 0.0f, 0.0f};
  float4v e0 = z, lo = z, e1 = z, e2 = z;
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    const half8v f0 = frag(ks);
    e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(f0, xh[ks], e0, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag(5 + ks), xh[ks], lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(f0, xl[ks], lo, 0, 0, 0);
    if (ks < 2) {
      const half8v f1 = frag(3 + ks);
      e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1, xh[ks], e1, 0, 0, 0);      // (no lo image here since the images are scaled, r06)
      e2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag(8 + ks), xh[ks], e2, 0, 0, 0);
    }
  }
  // (the factor in a vector register: a VALU instruction with an SGPR operand issues at 4.3 cycles instead of 2.4, tools/exp/valu_forms.hip)
  float lo_inv = kBank2LoInv;
  asm volatile("" : "+v"(lo_inv));
  const float4v v0 = e0 + lo * lo_inv, v1 = e1;
  float4v a01 = z, l01 = z, a23 = z, a45 = z, a6 = z;
  if (MOMENTS) {
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const half8v f01 = frag(kCtabUnity + ks);
      a01 = __builtin_amdgcn_mfma_f32_16x16x32_f16(f01, xh[ks], a01, 0, 0, 0);
      l01 = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag(kCtabUnity + 3 + ks), xh[ks], l01, 0, 0, 0);
      l01 = __builtin_amdgcn_mfma_f32_16x16x32_f16(f01, xl[ks], l01, 0, 0, 0);
      a23 = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag(kCtabUnity + 6 + ks), xh[ks], a23, 0, 0, 0);
      a45 = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag(kCtabUnity + 9 + ks), xh[ks], a45, 0, 0, 0);
      if constexpr (SIX) a6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag(kCtabUnity + 12 + ks), xh[ks], a6, 0, 0, 0);
    }
  }
  const float4v m01 = a01 + l01 * lo_inv;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int sl = (8 * bb + 2 * g + p) ^ (bb & 7);
    const float4v row = {v0[2 * p], v0[2 * p + 1], v1[2 * p], v1[2 * p + 1]};
    L.qa[sl] = row;
    const float e2d2 = __uint_as_float(pack_h2(e2[2 * p] * 0.015625f, e2[2 * p + 1] * 0.015625f));
    if constexpr (MOMENTS) {
      const float4v r0 = {m01[2 * p], m01[2 * p + 1], a23[2 * p], e2d2};
      L.qm0[sl] = r0;
      const float4v r1 = {a23[2 * p + 1], a45[2 * p], a45[2 * p + 1], a6[2 * p]};
      L.qm1[sl] = r1;
    } else {
      L.e2d2(sl) = e2d2;
    }
  }
}

// the outputs of one row of a placed pass (bank and ring in LDS): MODE 1 fc = 1, MODE 3 fc = 1 + the
// moment correction  -g (cos(pi s) Re Q - sin(pi s) Im Q),  Q = sum_i M_i (i 32 G)^i (alpha_i + i beta_i),  G = pi g, w = G s,
// alpha_i = 1/(i! (i+1)) - w^2 / (2 i! (i+3)),  beta_i = -w / (i! (i+2))   (tools/sinc3_model.py: 1e-7 for g <= 0.0101)
template <int MODE, int NCH = 1, class LDS>
__device__ __forceinline__ float s3_out_row(const LDS& L, const int ci, const float sr, const float epr, const int wsK, const int ch = 0) {
  using T32 = TapTab<32>;
  constexpr float kUs = kBank2ScaleInv * kImgScaleInv;      // bank rows -> signal units
  constexpr float kMs = kImgScaleInv;                       // moment rows -> signal units (rides in the Horner's constants)
  const int sl = ci ^ ((ci >> 3) & 7);
  const int rc = (wsK + ci) & (kRingF - 1);
  const float* xp = &L.ring[NCH * rc + ch];
  const float xm2 = xp[-2 * NCH], xm1 = xp[-NCH], x0 = xp[0], xp1 = xp[NCH], xp2 = xp[2 * NCH];
  const float q = sr * sr, q64 = 64.0f * q;
  const float R1 = fast_rcp(fmaf(q, T32::B[1], T32::A[1])), R2 = fast_rcp(fmaf(q, T32::B[2], T32::A[2]));
  {
    const float E1 = xp1 + xm1, D1 = xp1 - xm1, E2 = xp2 + xm2, D2 = xp2 - xm2;
    const float4v row = L.qa[sl];
    float4v M0 = {0.0f, 0.0f, 0.0f, 0.0f};
    unsigned w2;
    if constexpr (MODE != 1) {
      M0 = L.qm0[sl];
      w2 = __float_as_uint(M0[3]);
    } else {
      w2 = __float_as_uint(L.e2d2(sl));
    }
    const float e = fmaf(q, fmaf(q64, h_lo(w2), row[2]), row[0]), d = fmaf(q, fmaf(q64, h_hi(w2), row[3]), row[1]);
    const float en = fmaf(E2, R2, -(E1 * R1));
    const float dn = fmaf(D2 + D2, R2, -(D1 * R1));
    const float et = fmaf(e, kUs, en), dt = fmaf(d, kUs, dn);
    const float spq = sinpi_poly(q);                                  // sin(pi s) / s
    const float unity = spq * fmaf(-sr, fmaf(sr, et, dt), x0 * 0.318309886f);
    if constexpr (MODE == 1) return unity;
    else {
    const float4v Mh = L.qm1[sl];
    const float m3 = Mh[0], m4 = Mh[1], m5 = Mh[2], m6 = Mh[3];
    const float g = epr * fast_rcp(1.0f + epr);                        // 1 - fc
    const float G = 3.14159265f * g, w = G * sr, w2m = w * w, G32 = 32.0f * G;
    float re, im;
    if constexpr (MODE == 3) {
      // i = 6 (beta_6 and the w^2 terms of i >= 3 are below 1e-8 of the peak)
      re = m6 * (1.98412698e-4f * kMs);                                // 1 / (6! 7)
      im = 0.0f;
    } else {                                                           // MODE 2: the series to order 5 (1 - fc <= 0.0105)
      re = m5 * (1.38888889e-3f * kMs);                                // i = 5: 1/(5! 6), 1/(5! 7)
      im = m5 * (w * -(1.19047619e-3f * kMs));
    }
#define S3_MOM_STEP(Mi, A0, A1, B0)                                   \
    {                                                                 \
      const float al_ = (A1) != 0.0f ? fmaf(w2m, -((A1) * kMs), (A0) * kMs) : (A0) * kMs, be_ = w * -((B0) * kMs);      /* (A1 = 0: no fma with -0.0) */ \
      const float nre_ = fmaf(-G32, im, (Mi) * al_), nim_ = fmaf(G32, re, (Mi) * be_); \
      re = nre_;                                                      \
      im = nim_;                                                      \
    }
    if constexpr (MODE == 3) S3_MOM_STEP(m5, 1.38888889e-3f, 0.0f, 1.19047619e-3f)             // i = 5: 1/(5! 6), -, 1/(5! 7)
    S3_MOM_STEP(m4, 8.33333333e-3f, 0.0f, 6.94444444e-3f)             // i = 4: 1/(4! 5), -, 1/(4! 6)
    S3_MOM_STEP(m3, 4.16666667e-2f, 0.0f, 3.33333333e-2f)             // i = 3: 1/(3! 4), -, 1/(3! 5)
    S3_MOM_STEP(M0[2], 1.66666667e-1f, 5.0e-2f, 1.25e-1f)             // i = 2: 1/(2! 3), 1/(2 2! 5), 1/(2! 4)
    S3_MOM_STEP(M0[1], 0.5f, 0.125f, 3.33333333e-1f)                  // i = 1: 1/2, 1/(2 4), 1/3
    S3_MOM_STEP(M0[0], 1.0f, 1.66666667e-1f, 0.5f)                    // i = 0: 1, 1/(2 3), 1/2
#undef S3_MOM_STEP
    const float S = sr * spq, C = __builtin_amdgcn_cosf(0.5f * sr);
    return fmaf(-g, fmaf(C, re, -(S * im)), unity);
    }
  }
}

// one chunk of the ring -> float16 images (and the ring's mirrors); returns false when float16 does not suit the chunk
template <class LDS>
__device__ __forceinline__ bool s3_convert(LDS& L, const int chunk, const int l) {
  const int wi = chunk * kPass + 2 * l;
  const int ix = wi & (kRingF - 1), ih = wi & (kRingH - 1);
  const float2 xx = *reinterpret_cast<const float2*>(&L.ring[ix]);
  const float x0 = xx.x, x1 = xx.y;
  const float am = fmaxf(fabsf(x0), fabsf(x1));
  const bool ok = !(__ballot(!(fabsf(x0) < kImgMax) || !(fabsf(x1) < kImgMax)) != 0ull ||       // (also false for NaN)
                    (__ballot(am >= kQuiet) == 0ull && __ballot(am > 0.0f) != 0ull));
  if ((chunk & 7) == 0 && l < 2) *reinterpret_cast<float2*>(&L.ring_tail[ix]) = xx;
  if ((chunk & 7) == 7 && l == kWave - 1) *reinterpret_cast<float2*>(&L.ring_head[2]) = xx;
  const float s0 = x0 * kImgScale, s1 = x1 * kImgScale;
  const half2v h = {S2_HI(s0), S2_HI(s1)};
  const half2v lo = {(_Float16)((s0 - (float)h[0]) * 4096.0f), (_Float16)((s1 - (float)h[1]) * 4096.0f)};
  *reinterpret_cast<half2v*>(&L.img[0][ih]) = h;
  *reinterpret_cast<half2v*>(&L.img[1][ih]) = lo;
  return ok;
}

// Stereo ring (frames left, right): channel `ch` of one chunk -> that channel's float16 images.  `mirrors`: the call also keeps the
// ring's mirror frames (both channels': the caller converts channel 0 of a chunk first).
template <class LDS>
__device__ __forceinline__ bool s3_convert_ch(LDS& L, const int chunk, const int l, const int ch, const bool mirrors) {
  const int wi = chunk * kPass + 2 * l;
  const int ix = wi & (kRingF - 1), ih = wi & (kRingH - 1);
  const float4v xx = *reinterpret_cast<const float4v*>(&L.ring[2 * ix]);
  const float x0 = ch ? xx[1] : xx[0], x1 = ch ? xx[3] : xx[2];
  const float am = fmaxf(fabsf(x0), fabsf(x1));
  const bool ok = !(__ballot(!(fabsf(x0) < kImgMax) || !(fabsf(x1) < kImgMax)) != 0ull ||
                    (__ballot(am >= kQuiet) == 0ull && __ballot(am > 0.0f) != 0ull));
  if (mirrors) {
    if ((chunk & 7) == 0 && l < 2) *reinterpret_cast<float4v*>(&L.ring_tail[2 * ix]) = xx;
    if ((chunk & 7) == 7 && l == kWave - 1) *reinterpret_cast<float4v*>(&L.ring_head[4]) = xx;
  }
  const float s0 = x0 * kImgScale, s1 = x1 * kImgScale;
  const half2v h = {S2_HI(s0), S2_HI(s1)};
  const half2v lo = {(_Float16)((s0 - (float)h[0]) * 4096.0f), (_Float16)((s1 - (float)h[1]) * 4096.0f)};
  *reinterpret_cast<half2v*>(&L.img[2 * ch][ih]) = h;
  *reinterpret_cast<half2v*>(&L.img[2 * ch + 1][ih]) = lo;
  return ok;
}

// fc < 1 passes are fc = 1 + the moment correction (one image per channel, per-lane 1 - fc exact): valid for 1 - fc <= 0.0125,
// steeper tiles go to the block kernel's list.  (r04's other forms -- a pass as a chain of stages, the modulated-image form of
// the fc < 1 taps, an fc = 1-only kernel at three waves per SIMD, workgroups of several streams sharing the constants through
// LDS -- are in the repository's history and in NOTES r04 with their numbers; the product carries this one.)
constexpr float kEpMaxMom = 0.0125f / (1.0f - 0.0125f);      // period - 1 at 1 - fc = 0.0125
// Passes whose lanes all have 1 - fc <= 0.0105 take the moment series to order 5 (regime 2: no m6 -- three matrix-core
// instructions and a Horner step per pass; within 1.6e-6 of the peak on a Nyquist tone there, 5.5e-6 at 0.0125: tools/sinc3_model.py);
// the order-6 loop (regime 3) keeps a stream until its passes fall below 0.95 of that (no flapping between the loops)
constexpr float kEpMom5 = 0.0105f / (1.0f - 0.0105f), kEpMom5Lo = 0.95f * kEpMom5;
#ifndef PAR_S3_SHARE_OUT
#define PAR_S3_SHARE_OUT 1      // stereo: 0 = channel 1's row arithmetic worked out afresh (measured: 158 against 165 G on the benchmark's tape)
#endif
#ifndef PAR_S3_PIN_MONO
#define PAR_S3_PIN_MONO 0       // 1: the mono loop's row results pinned like the stereo loop's (234 instead of 250 registers, 1 % slower)
#endif
// KIND: which streams of the launch the kernel takes, and what it has to know for them (r06).  A wave issues an instruction
// every ~5 cycles whatever its kind, so two waves per SIMD -- what the 25 constant fragments of both filter sets leave room
// for -- cannot fill the SIMD's issue slots; a stream whose tiles hold fc = 1 outputs only needs the fc = 1 bank's ten:
//   0  every stream, both tap regimes (stereo: its LDS allows two streams per SIMD either way)
//   1  the streams WITHOUT a tile the plan marks kTileMaySlow: fc = 1 passes only, 10 fragments, no moment rows in LDS:
//      three waves per SIMD.  (A pass with an fc < 1 lane after all -- the hint is conservative, this does not happen -- sends
//      its tile to the block kernel's list.)
//   2  the streams with such a tile: both regimes, as KIND 0
//   3  (NCH = 2) ONE channel of an interleaved two-channel file -- the reference's use_channels, a strided column view
//      (util/resampling.py:211-227): the ring holds the frames as they lie in memory (a.sig = the wanted channel's first sample:
//      it is channel 0 of the frames that start there), conversion, bank and gather take that channel only, the loop is the
//      mono one; outputs a.out_stride elements apart.  Every stream, both regimes.
// A mono file is launched as KIND 2 and KIND 1 back to back over the same grid; a wave of the wrong kind leaves at once.
// (the body of the kernel: workgroup `bx` of one file's launch -- k_sinc_pipe has one file, k_sinc_pipe_n up to eight)
template <int NCH, int KIND>
__device__ __forceinline__ void sinc_pipe_body(const S2Args& a, const int bx, S3Lds<NCH, KIND != 1>& L) {
  static_assert(NCH == 1 || NCH == 2, "mono, or an interleaved stereo file");
  static_assert(KIND == 0 || KIND == 3 || NCH == 1, "the stereo form takes every stream");
  static_assert(KIND != 3 || NCH == 2, "one channel of FRAMES");
  constexpr bool kPick = KIND == 3;               // one channel of two-channel frames: stereo ring, mono loop
  constexpr bool kTwo = NCH == 2 && !kPick;       // both channels of a pass: the stereo loop
  constexpr bool kMom = KIND != 1;
  const int l = threadIdx.x & (kWave - 1);
  if (bx < a.n_edge) {                            // an end tile's wave: tile 0, then n_full - 2, n_full - 1 and the partial one
    if constexpr (KIND == 1) return;              // (done by the launch of the other kind)
    else {
    const int e = bx / kEdgeWavesPerTile, w = bx % kEdgeWavesPerTile;
    const int64_t T = e == 0 ? 0 : a.n_full - 3 + e;
    const int64_t jw = T * kSincTileOutputs + (int64_t)w * kEdgeWaveOut;
    const int nrem = (int)(a.len_out - jw < (int64_t)kEdgeWaveOut ? (a.len_out - jw > 0 ? a.len_out - jw : 0) : kEdgeWaveOut);
    float* const piece = reinterpret_cast<float*>(&L);
    static_assert(sizeof(L) >= fused_capw(2 * NCH, NCH) * NCH * sizeof(float), "the wave's span fits the stream's LDS");
    if constexpr (kPick) {                        // a strided view: the block kernel's general path, 128 outputs per wave too
      static_assert(sizeof(L) >= fused_capw(2, 1) * sizeof(float), "the wave's span fits the stream's LDS");
      if (nrem > 0)
        fused_wave<1, 32, 2, false>(a.len_out, a.sig, nullptr, 2, a.len_in, 32, a.tab, a.tmd, a.out, nullptr, a.out_stride, a.fa, piece, l,
                                    jw, nrem, piece);
      return;
    }
    if (nrem == kEdgeWaveOut)
      fused_wave<NCH, 32, 2 * NCH, true>(a.len_out, a.sig, a.sig + 1, NCH, a.len_in, 32, a.tab, a.tmd, a.out, a.out + 1, NCH, a.fa, piece, l,
                                         jw, nrem, piece);
    else if (nrem > 0)
      fused_wave<NCH, 32, 2 * NCH, false>(a.len_out, a.sig, a.sig + 1, NCH, a.len_in, 32, a.tab, a.tmd, a.out, a.out + 1, NCH, a.fa, piece, l,
                                          jw, nrem, piece);
    return;
    }
  }
  const int64_t stream_id = (int64_t)bx - a.n_edge;
  const int my_tiles = stream_id < a.n_big ? a.tiles : a.tiles_tail;
  const int64_t Ta = stream_id < a.n_big ? stream_id * a.tiles : a.n_big * a.tiles + (stream_id - a.n_big) * a.tiles_tail;
  if (Ta >= a.n_full) return;
  const int64_t Tb = Ta + my_tiles < a.n_full ? Ta + my_tiles : a.n_full;
  const int64_t Ja = Ta * kSincTileOutputs, Jb = Tb * kSincTileOutputs;
  long long A0;
  int hd_dA, hd_fl;
  {
    const int64_t Ti = Ta + l < a.n_tiles ? Ta + l : a.n_tiles - 1;
    const TileHdr h = a.hdr[l <= my_tiles ? Ti : Ta];
    // (lane 0's anchor through readfirstlane, not a shuffle: the compiler takes a shuffle's result for lane-variant, and with A0
    // everything derived from it -- the ring's source, the chunks inside the file, dma_bad, hence the loop's exit test and every
    // piece of state the loop carries -- sat in vector registers behind exec-masked branches, r06)
    A0 = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)h.anchor >> 32)) << 32) |
                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)h.anchor));
    const long long d = h.anchor - A0;
    hd_fl = h.flags | ((d > -0x40000000ll && d < 0x40000000ll) ? 0 : 1);
    if (a.n_edge > 0 && (Ta + l == 0 || Ta + l >= a.n_full - 2)) hd_fl |= 1 | kTileEdge;      // end tiles: not streamed, not pushed
    hd_dA = (int)d;
    if constexpr (KIND == 1 || KIND == 2) {       // is this stream the kernel's kind?
      const bool any_slow = __ballot(l < my_tiles && Ta + l < Tb && (h.flags & kTileMaySlow) != 0) != 0ull;
      if (any_slow != (KIND == 2)) return;
    }
  }
  half8v fr[kBank2Frags];                        // the fc = 1 bank's constant fragments
  {
    const uint4* src = reinterpret_cast<const uint4*>(kBank2Frags32) + l;
#pragma unroll
    for (int f = 0; f < kBank2Frags; ++f) fr[f] = __builtin_bit_cast(half8v, src[f * kWave]);
  }
  half8v fmr[kBank3Frags];                       // the moment filters' (kinds 0 and 2)
  if constexpr (kMom) {
    const uint4* src = reinterpret_cast<const uint4*>(kBank3Frags32) + l;
#pragma unroll
    for (int f = 0; f < kBank3Frags; ++f) fmr[f] = __builtin_bit_cast(half8v, src[f * kWave]);
  }
  {
    uint4* z = reinterpret_cast<uint4*>(&L.img[0][0]);
    const uint4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < (int)(sizeof(L.img) / 16 / kWave); ++i) z[i * kWave + l] = zero;
  }
  const float tolf = (float)((fabs((double)A0) + 2.0e7) * 1.2e-16 + 2.0e-10) + 2.0e-7f;
  const int nJ = (int)(Jb - Ja);
  float* const outW = a.out + (kPick ? a.out_stride : (int64_t)NCH) * Ja;
  const uint4* const recW = reinterpret_cast<const uint4*>(a.rec) + (Ja >> kRecShift);
  const uint4* const rec2W = reinterpret_cast<const uint4*>(a.rec2) + (Ja >> kRecShift);
  const int blk_max = (int)(((a.len_out + kRec - 1) >> kRecShift) - (Ja >> kRecShift)) - 1;      // last block with a record, relative

  // ---- stream state (wave-uniform) ----
  int j0 = 0;                                    // first output not yet placed into a finished pass
  int wbase = 0, conv_next = 0, conv_lo = 0, dma_next = 0, dma_bad = INT_MAX, mode = 0, pk = 0;
  int regime = 1;                                // the loop the current pass belongs to: 1 fc = 1, 2 / 3 fc = 1 + the moment correction to order 5 / 6
                                                 // (mode: 1 = the float16 images in LDS follow the ring, 0 = to be rebuilt)
  int rbA = 0, rbB = 0, rbC = 0;                 // first block (relative) of the record buffers of passes pk + 1, pk + 2, pk + 3

  // records: lanes 0-7 fetch first pieces, 8-15 second pieces, one 16-byte direct load each.  (The plan's record arrays end four
  // tiles behind the file's last block -- fused_blocks(), pos_plan.h -- so the 16 blocks a guess may reach past it are readable.)
  const uint4* const rec_lane = (l < 8 ? recW : rec2W) + (l & 7);
  auto fetch_records = [&](int buf, int blk0) {  // 8 first pieces + 8 second pieces from block blk0 on -> recs[buf]
    if (l < 16) dma_dwordx4(rec_lane + (blk0 < 0 ? 0 : blk0), &L.recs[buf][0]);
  };
  // ring chunks: chunk k = input samples [wbase + 128 k, + 128) relative to A0 -> ring slot k & 7.  Chunks dma_klo .. dma_khi lie
  // inside the file (set when the ring is restarted); the others are never used (dma_bad) and fetch some readable stretch instead.
  int dma_klo = 0, dma_khi = -1;
  const float* ring_src = a.sig;                 // &sig[A0 + wbase]
  auto ring_restart = [&]() {
    const long long o = A0 + wbase;
    ring_src = a.sig + NCH * o;
    // (one channel of frames: the last frame's second word may lie one float behind the caller's view -- never fetched)
    const long long klo = o >= 0 ? 0 : (-o + kPass - 1) / kPass, khi = ((long long)a.len_in - (kPick ? 1 : 0) - o) / kPass - 1;
    dma_klo = (int)(klo > 0x3fffffff ? 0x3fffffff : klo);
    dma_khi = (int)(khi > 0x3fffffff ? 0x3fffffff : (khi < -1 ? -1 : khi));
  };
  const unsigned lane_bytes = 4u * (unsigned)l;
  auto chunk_dma = [&](int k) {
    const bool inside = k >= dma_klo && k <= dma_khi;
    if (!inside) dma_bad = k < dma_bad ? k : dma_bad;
    const float* src = inside ? ring_src + (long long)(kPass * NCH) * k : a.sig;      // (a.sig: len_in >= 128 for every file this kernel is launched on)
    if constexpr (NCH == 2) dma_chunk256(src, lane_bytes, &L.ring[(k & 7) * (kPass * 2)]);
    else dma_chunk128(src, lane_bytes, &L.ring[(k & 7) * kPass]);
  };
  auto push_tile = [&](int64_t T) {
    if (l == 0) {
      const int slot = atomicAdd(a.redo_count, 1);
      a.redo_list[slot] = (int)T;
    }
  };
  if (stream_id == 0 && a.n_full < a.n_tiles && a.n_edge == 0) push_tile(a.n_full);

  // placement of the pass that starts at output j from record buffer `buf` (first block rb); the lanes' tile anchors
  struct Placed {
    S2Row R[2];
    int nt[2];
    int fl0, fl1, tend;
    int epm;                                     // the larger period - 1 of the lane's two outputs, as its bit pattern (ep >= 0: ordered like the floats)
  };
  int tc_T = -1, tc_dA0 = 0, tc_dA1 = 0, tc_fl0 = 0, tc_fl1 = 0;      // anchors and flags of the tile of j and of the one behind it (refreshed once per tile)
  auto place = [&](int j, int buf, int rb) {
    Placed P;
    const int T = j >> 10;
    P.tend = (T + 1) << 10;
    if (T != tc_T) {
      tc_T = T;
      tc_dA0 = __builtin_amdgcn_readlane(hd_dA, T);
      tc_dA1 = __builtin_amdgcn_readlane(hd_dA, T + 1);
      tc_fl0 = __builtin_amdgcn_readlane(hd_fl, T);
      tc_fl1 = __builtin_amdgcn_readlane(hd_fl, T + 1);
    }
    const int dA0 = tc_dA0, dA1 = tc_dA1;
    P.fl0 = tc_fl0;
    P.fl1 = tc_fl1;
    const int t5 = (j & (kRec - 1)) + l;
    const int u = t5 & (kRec - 1);
    const int bi = ((j >> kRecShift) - rb + (t5 >> kRecShift)) & 7;
    const uint4* rp = &L.recs[buf][0];
    const uint4 ra0 = rp[bi], rb0 = rp[8 + bi], ra1 = rp[(bi + 2) & 7], rb1 = rp[8 + ((bi + 2) & 7)];
    const int uc = u - kRec / 2;
    const float uf = (float)uc, u2f = uf * uf, tw1 = fmaf(2.0f, uf, 1.0f), tw0 = tw1 - 2.0f;
    P.nt[0] = clamp64(P.tend - j);
    P.nt[1] = clamp64(P.tend - j - 64);
    // a segment starts inside one block in eight: most passes have no second piece to select
    if (__ballot((unsigned)u > (ra0.x & 31u) || (unsigned)u > (ra1.x & 31u)) == 0ull) {
      P.R[0] = s2_place_row<false>(ra0, rb0, u, (l < P.nt[0] ? dA0 : dA1) + uc, uf, u2f, tw1, tw0, tolf);
      P.R[1] = s2_place_row<false>(ra1, rb1, u, (l < P.nt[1] ? dA0 : dA1) + uc, uf, u2f, tw1, tw0, tolf);
    } else {
      P.R[0] = s2_place_row(ra0, rb0, u, (l < P.nt[0] ? dA0 : dA1) + uc, uf, u2f, tw1, tw0, tolf);
      P.R[1] = s2_place_row(ra1, rb1, u, (l < P.nt[1] ? dA0 : dA1) + uc, uf, u2f, tw1, tw0, tolf);
    }
    // (one integer maximum serves the regime test and the moment correction's limit: two compares where there were a float
    // maximum with its quieting moves and three; ep is fmed3(e, 0, 3e38): never NaN, -0 sorts below everything)
    P.epm = max(__float_as_int(P.R[0].ep), __float_as_int(P.R[1].ep));
    return P;
  };

  S3Pass P;                                      // the pass OUT works on next (placed, its bank in LDS)
  P.nok[0] = P.nok[1] = 0;
  P.j = P.ws = 0;
  P.c[0] = P.c[1] = 0;
  P.s[0] = P.s[1] = P.ep[0] = P.ep[1] = 0.0f;

  auto out_pass = [&](auto mode_tag, const S3Pass& Q, float (&res)[2], const int ch = 0) {
    constexpr int MODE = decltype(mode_tag)::value;
    const int wsK = Q.ws - wbase;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      int ci = Q.c[r] - Q.ws;
      ci = ci < 0 ? 0 : (ci > kPass - 1 ? kPass - 1 : ci);
      float sr = Q.s[r], epr = Q.ep[r];
#if !PAR_S3_SHARE_OUT
      // (experiment: channel 1's row arithmetic hidden from the common-subexpression pass.  By default the compiler keeps the
      // channel-independent values of a row -- reciprocals, sin(pi s)/s, the moment step's coefficients -- from channel 0's turn)
      if (NCH == 2 && ch == 1) asm volatile("" : "+v"(sr), "+v"(epr));
#endif
      res[r] = s3_out_row<MODE, NCH>(L, ci, sr, epr, wsK, ch);
    }
  };
  auto store_pass = [&](const S3Pass& Q, const float (&res)[2]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
      if (l < Q.nok[r] && !(PAR_S2_EXP & 4)) {
        if constexpr (kPick) outW[(int64_t)((unsigned)Q.j + 64u * r + (unsigned)l) * a.out_stride] = res[r];
        else outW[(unsigned)Q.j + 64u * r + (unsigned)l] = res[r];
      }
  };
  auto store_pass2 = [&](const S3Pass& Q, const float (&res0)[2], const float (&res1)[2]) {      // stereo: a frame per lane and row
#pragma unroll
    for (int r = 0; r < 2; ++r)
      if (l < Q.nok[r] && !(PAR_S2_EXP & 4))
        reinterpret_cast<float2*>(outW)[(unsigned)Q.j + 64u * r + (unsigned)l] = make_float2(res0[r], res1[r]);
  };
  // one chunk of the ring -> float16 image(s); stereo: both channels of the chunk
  auto convert_chunk = [&](int chunk) -> bool {
    if constexpr (kPick) {
      return s3_convert_ch(L, chunk, l, 0, true);
    } else if constexpr (NCH == 2) {
      const bool ok0 = s3_convert_ch(L, chunk, l, 0, true);
      const bool ok1 = s3_convert_ch(L, chunk, l, 1, false);
      return ok0 && ok1;
    } else {
      return s3_convert(L, chunk, l);
    }
  };

  // ---- the cold path: places, converts for, banks and -- unless it is a full pass the loop can take -- finishes the pass
  // at j0, everything in order with full waits.  Returns true with P = a full pass ready for the loop (records of the two
  // passes behind it and the ring's next two chunks landed), false when the range is done.
#if PAR_S2_EXP & 128
#define S3_COUNT(k) do { if (l == 0) atomicAdd(a.redo_count + (k), 1); } while (0)
#else
#define S3_COUNT(k) do { } while (0)
#endif
  auto start_run = [&]() -> bool {
    for (;;) {
      if (j0 >= nJ) return false;
      S3_COUNT(1);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      wave_lds_fence();
      const int buf = pk & 3;
      fetch_records(buf, j0 >> kRecShift);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      wave_lds_fence();
      const Placed Q = place(j0, buf, j0 >> kRecShift);
      const int tend = Q.tend;
      // lane sets: valid lanes (inside the range), lanes of this tile
      int nv[2] = {clamp64(nJ - j0), clamp64(nJ - j0 - 64)};
      // the moment correction covers 1 - fc <= 0.0125
      const unsigned long long b0 = __ballot(Q.R[0].bad || !(Q.R[0].ep <= kEpMaxMom)) & prefix(nv[0]),
                               b1 = __ballot(Q.R[1].bad || !(Q.R[1].ep <= kEpMaxMom)) & prefix(nv[1]);
      const unsigned long long bad_here = (b0 & prefix(Q.nt[0])) | (b1 & prefix(Q.nt[1]));
      const bool bad_next = ((b0 & ~prefix(Q.nt[0])) | (b1 & ~prefix(Q.nt[1])) | (unsigned long long)((Q.fl1 & 1) && tend < nJ)) != 0ull;
      if (bad_next) {                             // the pass ends at the tile border
        nv[0] = nv[0] < Q.nt[0] ? nv[0] : Q.nt[0];
        nv[1] = nv[1] < Q.nt[1] ? nv[1] : Q.nt[1];
      }
      bool skip = (Q.fl0 & 1) || bad_here != 0ull;
      const unsigned long long gen = (__ballot(1.0f + Q.R[0].ep != 1.0f) & prefix(nv[0])) | (__ballot(1.0f + Q.R[1].ep != 1.0f) & prefix(nv[1]));
      const int ws = __builtin_amdgcn_readfirstlane(Q.R[0].c) & ~7;
      S3Pass N;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        N.c[r] = Q.R[r].c;
        N.s[r] = Q.R[r].s;
        N.ep[r] = Q.R[r].ep;
        const unsigned long long in = __ballot(Q.R[r].c - ws < kPass);
        N.nok[r] = __popcll(in & prefix(nv[r]));
      }
      N.j = j0;
      N.ws = ws;
      const unsigned long long steep = (__ballot(Q.R[0].ep > kEpMom5) & prefix(nv[0])) | (__ballot(Q.R[1].ep > kEpMom5) & prefix(nv[1]));
      const int want = gen == 0ull ? 1 : (steep == 0ull ? 2 : 3);      // the pass's regime
      if (!kMom && want != 1) skip = true;        // (a kernel of fc = 1 streams met an fc < 1 lane: the block kernel's tile)
      if (!skip) {
        bool rebuild = mode != 1;
        // the ring: restarted at the first pass of the wave and after a jump the fetched chunks do not cover
        const int lo_need = ws - 39 - wbase;      // window index of the first image sample the bank reads
        if (mode == 0 || lo_need < kPass * (dma_next - 8) + kPass || lo_need < 0 || lo_need > kPass * dma_next || ws - wbase > (1 << 20)) {
          __builtin_amdgcn_s_waitcnt(0x0F70);     // (nothing may still be on its way into the old ring)
          wbase = ws - 39;
          dma_next = 0;
          dma_bad = INT_MAX;
          rebuild = true;
          ring_restart();
        }
        const int wsK = ws - wbase;
        regime = want;
        if (rebuild) {
          mode = 1;
          conv_lo = conv_next = (wsK - 39) >> 7;
        }
        // slack of the conversion window.  The loop checks the pass BEHIND this one (its bank needs the image to reach
        // 161 samples beyond its first centre, one pass = 120 .. 136 centres further on) and converts one chunk per
        // iteration: fc = 1 passes advance by <= 128 centres (the converted stretch drifts ahead of them: start low),
        // fc < 1 passes by >= 128 (start as far ahead as the four-chunk image ring allows)
        const int conv_target = want == 1 ? (wsK + 300 + 127) >> 7 : (wsK + 465) >> 7;
        const int need_lo = (wsK - 39) >> 7;      // first image chunk the bank of this pass reads
        if (conv_next < need_lo) conv_lo = conv_next = need_lo;
        int conv_end = conv_target > conv_next ? conv_target : conv_next;
        if (need_lo < conv_lo || need_lo < conv_end - 4) {        // the image ring (4 chunks) has moved past it: converted again
          conv_lo = conv_next = need_lo;
          conv_end = conv_target > need_lo ? conv_target : need_lo;
        }
        while (dma_next < conv_end + 2) chunk_dma(dma_next++);
        // records of the two passes behind this one set out now (exact start of the next pass, a guess for the one behind)
        const int jn = j0 + N.nok[0] + N.nok[1];
        rbA = jn >> kRecShift;
        rbB = (jn + 120) >> kRecShift;
        fetch_records((pk + 1) & 3, rbA);
        fetch_records((pk + 2) & 3, rbB);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        wave_lds_fence();
        if (!skip) {
#pragma unroll 1
          while (conv_next < conv_end) {
            if (conv_next >= dma_bad) {
              skip = true;
              break;
            }
            const bool ok = convert_chunk(conv_next);
            if (!ok) {
              skip = true;
              break;
            }
            ++conv_next;
          }
        }
        if (!skip && conv_next + 1 >= dma_bad) skip = true;      // the loop would convert a chunk that was never fetched
      }
      if (skip) {
        if (!(Q.fl0 & kTileEdge)) push_tile(Ta + (j0 >> 10));
        j0 = tend;
        mode = 0;
        continue;
      }
      wave_lds_fence();
      const int offs = ws - wbase - 31;
      if (kMom && regime == 3) bank_image3m<kMom, true>(L, fr, fmr, offs, l);
      else if (kMom && regime == 2) bank_image3m<kMom, false>(L, fr, fmr, offs, l);
      else bank_image3m<false>(L, fr, fmr, offs, l);
      wave_lds_fence();
      j0 += N.nok[0] + N.nok[1];
      ++pk;
      rbC = (j0 + 240) >> kRecShift;
      const bool full = N.nok[1] >= 1 && N.nok[0] + N.nok[1] >= 120;
      if (full) {
        P = N;
        return true;
      }
      float res[2];
      if (kMom && regime == 3) out_pass(std::integral_constant<int, kMom ? 3 : 1>{}, N, res);
      else if (kMom && regime == 2) out_pass(std::integral_constant<int, kMom ? 2 : 1>{}, N, res);
      else out_pass(std::integral_constant<int, 1>{}, N, res);
      if constexpr (kTwo) {                       // the other channel through the same rows
        float res1[2];
        wave_lds_fence();
        if (regime == 3) bank_image3m<true, true>(L, fr, fmr, offs, l, 1);
        else if (regime == 2) bank_image3m<true, false>(L, fr, fmr, offs, l, 1);
        else bank_image3m<false>(L, fr, fmr, offs, l, 1);
        wave_lds_fence();
        if (regime == 3) out_pass(std::integral_constant<int, 3>{}, N, res1, 1);
        else if (regime == 2) out_pass(std::integral_constant<int, 2>{}, N, res1, 1);
        else out_pass(std::integral_constant<int, 1>{}, N, res1, 1);
        store_pass2(N, res, res1);
      } else {
        store_pass(N, res);
      }
    }
  };

#if PAR_S2_EXP & 64
  unsigned long long s3t_[4] = {0, 0, 0, 0}, s3last_ = __builtin_readcyclecounter();
  const unsigned long long s3start_ = s3last_;
#endif
  // ---- the loop: see the head of this kernel.  Leaves with P finished and j0 at a pass start_run() has to look at.
  auto hot = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    // (the cold path left the loop's state in vector registers: it is wave-uniform, and stays so through the loop)
    j0 = __builtin_amdgcn_readfirstlane(j0);
    wbase = __builtin_amdgcn_readfirstlane(wbase);
    conv_next = __builtin_amdgcn_readfirstlane(conv_next);
    dma_next = __builtin_amdgcn_readfirstlane(dma_next);
    dma_bad = __builtin_amdgcn_readfirstlane(dma_bad);
    pk = __builtin_amdgcn_readfirstlane(pk);
    rbA = __builtin_amdgcn_readfirstlane(rbA);
    rbB = __builtin_amdgcn_readfirstlane(rbB);
    rbC = __builtin_amdgcn_readfirstlane(rbC);
    for (;;) {
#if PAR_S2_EXP & 64
      const unsigned long long tA_ = __builtin_readcyclecounter();
#endif
      if constexpr (NCH == 2) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");      // (seven memory operations per iteration)
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
#if PAR_S2_EXP & 64
      const unsigned long long tB_ = __builtin_readcyclecounter();
      s3t_[0] += tB_ - tA_;
      s3t_[1] += tA_ - s3last_;
      s3last_ = tB_;
      s3t_[2] += 1;
#endif
      wave_lds_fence();
      // PLACE(pk): the pass behind P
      S3Pass N;
      bool ok;
      int wsK;
      auto place_next = [&]() {
      const Placed Q = place(j0, pk & 3, rbA);
      const unsigned long long bad = __ballot(Q.R[0].bad || Q.R[1].bad || Q.epm > __float_as_int(kEpMaxMom));
      // some lane with fc < 1: 1 + ep != 1 in float32, i.e. ep > 2^-24 (ep >= 0)
      const unsigned long long gen = __ballot(Q.epm > 0x33800000);          // 2^-24
      // order 5 while no lane is steeper than 1 - fc = 0.0105; order 6 while some lane is above 0.95 of that
      bool tier = true;
      if constexpr (MODE == 2) tier = __ballot(Q.epm > __float_as_int(kEpMom5)) == 0ull;
      if constexpr (MODE == 3) tier = __ballot(Q.epm > __float_as_int(kEpMom5Lo)) != 0ull;
      const int ws = __builtin_amdgcn_readfirstlane(Q.R[0].c) & ~7;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        N.c[r] = Q.R[r].c;
        N.s[r] = Q.R[r].s;
        N.ep[r] = Q.R[r].ep;
        N.nok[r] = __popcll(__ballot(Q.R[r].c - ws < kPass));
      }
      N.j = j0;
      N.ws = ws;
      wsK = ws - wbase;
      const int d = kPass * conv_next - wsK;      // image converted up to d samples beyond the first bank centre
      // (bitwise: one chain of scalar operations, no branch per term; >= 120 finished outputs imply a full first row)
      ok = (((Q.fl0 | Q.fl1) & 1) == 0) & (bad == 0ull) & (j0 + kPass <= nJ) & (N.nok[0] + N.nok[1] >= 120) &
           (MODE == 1 ? gen == 0ull : gen != 0ull) & tier & ((unsigned)(d - 161) <= 312u) & (conv_next + 1 < dma_bad) &
           ((unsigned)((j0 >> kRecShift) - rbA) <= 1u);
      };
      if constexpr (!kTwo) place_next();          // (stereo: behind OUT(P, 1), fewer registers live through the banks)
      // BANK(pk) over [ws, ws + 128): image samples converted in earlier iterations
      // OUT(P) first in program order: its gathers must precede the bank's row writes
      float res[2];
      out_pass(mode_tag, P, res);
      bool cok;
      if constexpr (kTwo) {
        // Stereo.  The bank rows in LDS hold ONE channel of one pass at a time: on entry channel 0 of P (banked by the previous
        // iteration or by the cold path).  OUT(P, 0) above has gathered them; now BANK(P, 1) -> OUT(P, 1) -> BANK(N, 0), each after
        // the reads of what it overwrites in program order.  Channel 1's image therefore runs one chunk behind channel 0's: its
        // bank of a pass sees the image exactly as channel 0's bank of that pass did one iteration earlier.
        float res1[2];
        // (pinned here: left alone the compiler sinks a row's arithmetic into the store's lane mask at the END of the iteration,
        // and the gathered rows and ring samples of both channels stay live through the banks: 33 registers in scratch)
        asm volatile("" : "+v"(res[0]), "+v"(res[1]));
        bank_image3m<MODE != 1, MODE == 3>(L, fr, fmr, P.ws - wbase - 31, l, 1);
        out_pass(mode_tag, P, res1, 1);
        asm volatile("" : "+v"(res1[0]), "+v"(res1[1]));
        place_next();
        const int offs = wsK - 31;
        bank_image3m<MODE != 1, MODE == 3>(L, fr, fmr, offs, l, 0);
        const bool c0 = s3_convert_ch(L, conv_next, l, 0, true);
        const bool c1 = s3_convert_ch(L, conv_next - 1, l, 1, false);
        cok = c0 && c1;
        fetch_records((pk + 2) & 3, rbC);
        chunk_dma(dma_next);
        store_pass2(P, res, res1);
      } else {
        const int offs = wsK - 31;
#if PAR_S3_PIN_MONO
        asm volatile("" : "+v"(res[0]), "+v"(res[1]));
#endif
        bank_image3m<MODE != 1, MODE == 3>(L, fr, fmr, offs, l);
        // CONV: one chunk per iteration
        cok = convert_chunk(conv_next);
        // FETCH + stores: records of pass pk + 2, chunk conv_next + 2, then P's outputs (five memory operations, in this order)
        fetch_records((pk + 2) & 3, rbC);
        chunk_dma(dma_next);
        store_pass(P, res);
      }
      wave_lds_fence();
      ++conv_next;
      ++dma_next;
      if (!cok) mode = 0;                         // (start_run converts again and sends the tile to the block kernel)
      S3_COUNT(2);
#if PAR_S2_EXP & 128
      if (!(ok && cok)) {
        if (((Q.fl0 | Q.fl1) & 1) != 0 || bad != 0ull) S3_COUNT(3);
        else if (!(j0 + kPass <= nJ)) S3_COUNT(4);
        else if (!(N.nok[1] >= 1 && N.nok[0] + N.nok[1] >= 120)) S3_COUNT(5);
        else if (!(MODE == 1 ? gen == 0ull : gen != 0ull)) S3_COUNT(6);
        else if (!(d >= 161 && d <= 473)) S3_COUNT(7);
        else if (!(conv_next < dma_bad)) S3_COUNT(9);
        else if (!((unsigned)((j0 >> kRecShift) - rbA) <= 1u)) S3_COUNT(11);
        else if (!cok) S3_COUNT(10);
        else S3_COUNT(8);
      }
#endif
      if (!(ok && cok)) {
        // stereo: channel 1's image catches up with channel 0's before the cold path takes over
        if constexpr (kTwo)
          if (mode != 0 && !s3_convert_ch(L, conv_next - 1, l, 1, false)) mode = 0;
        return;
      }
      // N becomes P
      P = N;
      j0 += N.nok[0] + N.nok[1];
      ++pk;
      rbA = rbB;
      rbB = rbC;
      rbC = (j0 + 240) >> kRecShift;
    }
  };

  while (start_run()) {
    if (kMom && regime == 3) hot(std::integral_constant<int, kMom ? 3 : 1>{});
    else if (kMom && regime == 2) hot(std::integral_constant<int, kMom ? 2 : 1>{});
    else hot(std::integral_constant<int, 1>{});
    // P has been finished by the loop; the pass at j0 needs the cold path
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
#if PAR_S2_EXP & 64
  if (l == 0) {
    unsigned long long* o = g_s2_phase + (size_t)stream_id * 16;
    o[0] = s3t_[0];                                // cycles in the loop's memory wait (includes two clock reads)
    o[1] = s3t_[1];                                // cycles between the end of one wait and the start of the next
    o[2] = s3t_[2];                                // iterations
    o[3] = __builtin_readcyclecounter() - s3start_;      // the wave's whole life behind its set-up
    o[6] = s3t_[2];
  }
#endif
}

template <int NCH, int KIND>
__global__ __launch_bounds__(kWave, KIND == 1 ? 3 : 2) void k_sinc_pipe(const S2Args a) {
  __shared__ S3Lds<NCH, KIND != 1> L;
  sinc_pipe_body<NCH, KIND>(a, (int)blockIdx.x, L);
}

// Several files in ONE launch (r06; the archive's 10-min files): a file's K_sinc is two or three kernels, each with a tail in
// which the GPU empties, plus the gaps between them -- ~0.13 ms per file, a sixth of a 10-min mono file's time.  Here the
// workgroups of file k + 1 follow those of file k inside the same grid: the tails and gaps are paid once per launch.
constexpr int kBatchFiles = kFusedBatchMax;
struct S2Batch {
  int n;
  int first[kBatchFiles + 1];                    // first workgroup of file k; first[n] = the grid
  S2Args f[kBatchFiles];
};
static_assert(sizeof(S2Batch) <= 3584, "kernel arguments fit the 4 KB segment");
template <int NCH, int KIND>
__global__ __launch_bounds__(kWave, KIND == 1 ? 3 : 2) void k_sinc_pipe_n(const S2Batch b) {
  __shared__ S3Lds<NCH, KIND != 1> L;
  const int bx = (int)blockIdx.x;
  int fi = 0;
#pragma unroll
  for (int k = 1; k < kBatchFiles; ++k) fi += (k < b.n && bx >= b.first[k]) ? 1 : 0;
  const S2Args* pa = &b.f[fi];                    // (kernel-argument memory: scalar loads at a wave-uniform offset)
  const S2Args a = *pa;
  sinc_pipe_body<NCH, KIND>(a, bx - b.first[fi], L);
}

// wave slots of the device for the two-waves-per-SIMD kernels (compute units x 4 SIMDs x 2; 2 048 on an MI355X): from the device's
// properties, so that a partitioned or CU-masked device cuts its streams for what it has
static int64_t stream_wave_slots(int device) {
  static std::atomic<int> cached[64];
  const int d = device >= 0 && device < 64 ? device : 0;
  int v = cached[d].load(std::memory_order_relaxed);
  if (v == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    v = cus * 8;
    cached[d].store(v, std::memory_order_relaxed);
  }
  return v;
}

// the launch's arguments and its grid for one file
static int64_t make_stream_args(int device, int64_t len_out, const float* sig, int64_t len_in, float* out, const FusedArgs& fa,
                                const float4* tab, const TapModes& tmd, int64_t pick_out_stride, S2Args& a) {
  const int64_t slots = stream_wave_slots(device);
  a.out_stride = pick_out_stride;
  a.len_out = len_out;
  a.sig = sig;
  a.len_in = len_in;
  a.out = out;
  a.hdr = fa.hdr;
  a.rec = fa.rec;
  a.rec2 = fa.rec2;
  a.redo_count = fa.redo_count;
  a.redo_list = fa.redo_list;
  a.fa = fa;
  a.tab = tab;
  a.tmd = tmd;
  a.n_full = len_out / kSincTileOutputs;
  a.n_tiles = ceil_div(len_out, kSincTileOutputs);
  // (launch_sinc_fused only comes here with >= 4 full tiles: the end tiles are distinct)
  a.n_edge = a.n_full >= 4 ? (int)(3 + (a.n_tiles - a.n_full)) * kEdgeWavesPerTile : 0;
  // Tiles per wave of the pipelined kernel: long streams amortise a wave's cold start and, beside a batch driver's plan kernels,
  // leave fewer wave boundaries for them to slip into -- 60-min file, ms per pipelined step: 4 tiles 5.03, 8: 4.84, 12: 4.70,
  // 16: 4.66, 24: 4.61, 32: 5.06 (10.3 rounds of the 2 048 wave slots: the last one nearly empty), 48: 4.67 (r05) -- while a
  // short file still has to fill the GPU's wave slots a few times over.
#ifdef PAR_EXPERIMENT            // (experiment builds only: the product reads no knob from the environment)
  static const int tiles_env = getenv("PAR_S2_TILES_RT") ? std::min(48, std::max(0, atoi(getenv("PAR_S2_TILES_RT")))) : 0;
  static const int tail_env = getenv("PAR_S2_TAIL") ? std::max(0, atoi(getenv("PAR_S2_TAIL"))) : 6;
  static const int tail_rounds = getenv("PAR_S2_TAIL_ROUNDS") ? std::max(0, atoi(getenv("PAR_S2_TAIL_ROUNDS"))) : 1;
#else
  constexpr int tiles_env = 0, tail_env = 6, tail_rounds = 1;      // tail_env: divisor of the last round's stream length
#endif
  const int64_t want = a.n_full / (4 * slots);
  // ... and an ODD number of them (r06): the concurrent streams start tiles x 4 KB (8 KB stereo) apart, and when that is a multiple
  // of 64 KB they camp on the same memory channels -- 60-min file, ms per step: 32 tiles 5.71, 16: 4.20, 48: 4.19 against
  // 4.09-4.18 for every odd count from 17 to 37 and 4.02-4.19 for 24 (what looked like rounds of the wave slots in r05 was this)
  a.tiles = tiles_env > 0 ? tiles_env : (int)(want < 8 ? 8 : (want > kMaxTilesPerWave ? kMaxTilesPerWave : want)) | 1;
  if (a.tiles > 48) a.tiles = 47;
  // ... and the launch's last round (2 048 wave slots' worth of tiles) as four rounds of quarter-length streams: the tail in which
  // the GPU empties behind the last long streams shrinks with them
  // (24 tiles, K_sinc alone: no short tail 4.37 ms, quarter-length 4.22, 1/8 4.20, 1/12 4.23; two rounds of them 4.23 / 4.30)
  a.tiles_tail = a.tiles >= 8 && tail_env > 1 ? std::max(2, a.tiles / tail_env) | 1 : a.tiles;
  a.n_big = (a.n_full - std::min<int64_t>(a.n_full, slots * tail_rounds * a.tiles * (a.tiles_tail < a.tiles ? 1 : 0))) / a.tiles;
  return a.n_big + ceil_div(a.n_full - a.n_big * a.tiles, (int64_t)a.tiles_tail) + a.n_edge;
}

int launch_sinc_stream(int device, int64_t len_out, const float* sig, int64_t len_in, float* out, const FusedArgs& fa,
                       const float4* tab, const TapModes& tmd, hipStream_t s, int nch, int64_t pick_out_stride) {
  S2Args a;
  const int64_t grid = make_stream_args(device, len_out, sig, len_in, out, fa, tab, tmd, pick_out_stride, a);
  if (grid > 0 && nch == 2 && pick_out_stride > 0) {
    hipLaunchKernelGGL((k_sinc_pipe<2, 3>), dim3((unsigned)grid), dim3(kWave), 0, s, a);
  } else if (grid > 0 && nch == 2) {
    hipLaunchKernelGGL((k_sinc_pipe<2, 0>), dim3((unsigned)grid), dim3(kWave), 0, s, a);
  } else if (grid > 0) {
    // mono: the streams with an fc < 1 tile (two waves per SIMD: all 25 constant fragments) and the end tiles, then the fc = 1
    // streams (three per SIMD), over the same grid -- a wave of the other kind leaves behind its tile headers
    hipLaunchKernelGGL((k_sinc_pipe<1, 2>), dim3((unsigned)grid), dim3(kWave), 0, s, a);
    hipLaunchKernelGGL((k_sinc_pipe<1, 1>), dim3((unsigned)grid), dim3(kWave), 0, s, a);
  }
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

// n <= kBatchFiles files of ONE form (nch 1: mono on unit strides; nch 2: interleaved stereo) in one launch per kernel kind
int launch_sinc_stream_batch(int device, int n, const StreamItem* items, const float4* tab, const TapModes& tmd, hipStream_t s, int nch) {
  if (n < 1 || n > kBatchFiles) return PAR_ERR_ARG;
  S2Batch b;
  b.n = n;
  int64_t at = 0;
  for (int k = 0; k < n; ++k) {
    b.first[k] = (int)at;
    at += make_stream_args(device, items[k].len_out, items[k].sig, items[k].len_in, items[k].out, items[k].fa, tab, tmd, 0, b.f[k]);
    if (at > INT_MAX) return PAR_ERR_ARG;
  }
  for (int k = n; k <= kBatchFiles; ++k) b.first[k] = (int)at;
  if (at <= 0) return PAR_OK;
  if (nch == 2) {
    hipLaunchKernelGGL((k_sinc_pipe_n<2, 0>), dim3((unsigned)at), dim3(kWave), 0, s, b);
  } else {
    hipLaunchKernelGGL((k_sinc_pipe_n<1, 2>), dim3((unsigned)at), dim3(kWave), 0, s, b);
    hipLaunchKernelGGL((k_sinc_pipe_n<1, 1>), dim3((unsigned)at), dim3(kWave), 0, s, b);
  }
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

}  // namespace par

#if PAR_S2_EXP & 64
extern "C" int par_debug_s2_phase_buffer(unsigned long long* dev_buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(par::g_s2_phase), &dev_buf, sizeof(dev_buf)) == hipSuccess ? 0 : 1;
}
#endif
