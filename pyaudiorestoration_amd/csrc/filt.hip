// K_sosfiltfilt -- zero-phase cascaded-biquad filtering of one long float64 signal.
//
// Semantics: scipy.signal.sosfiltfilt(sos, data) exactly as butter_bandpass_filter calls it
// (reference util/filters.py:24; scipy 1.15 _filter_design/_signaltools):
//     ext = odd_ext(x, padlen)            # 2*x[0]-x[padlen:0:-1] | x | 2*x[-1]-x[-2:-padlen-2:-1]
//     y, _ = sosfilt(sos, ext,     zi = sosfilt_zi(sos) * ext[0])
//     y, _ = sosfilt(sos, y[::-1], zi = sosfilt_zi(sos) * y[-1]);  return y[::-1][padlen:-padlen]
// with the transposed direct-form-II biquad  y = b0*x + z0; z0 = b1*x - a1*y + z1; z1 = b2*x - a2*y.
//
// An IIR is a serial recurrence, but a LINEAR one: state' = A*state + B*x.  So the signal is cut into
// blocks of kFiltBlock samples; every block (one lane each) computes its zero-state end state in
// parallel, a tiny serial pass chains  s[b+1] = A^blk * s[b] + r[b]  over the blocks, and every block
// is then re-run from its true initial state writing the outputs.  Results equal scipy's up to
// floating-point re-association of the state propagation (~1e-15 relative).
#include "par_common.h"
#include <math.h>
#include <vector>

namespace par {

constexpr int kFiltBlock = 256;

struct Biquad {
  double b0, b1, b2, a1, a2;
};

// element k of the pass (forward: k, backward: L-1-k)
__device__ __forceinline__ int64_t pass_index(int64_t k, int64_t L, int reverse) { return reverse ? L - 1 - k : k; }

__global__ void k_odd_ext(const double* __restrict__ x, int64_t n, int64_t pad, double* __restrict__ ext) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t L = n + 2 * pad;
  if (i >= L) return;
  double v;
  if (i < pad) v = 2.0 * x[0] - x[pad - i];
  else if (i < pad + n) v = x[i - pad];
  else v = 2.0 * x[n - 1] - x[n - 2 - (i - pad - n)];
  ext[i] = v;
}

// A lane owns one block of kFiltBlock consecutive samples, so the lanes of a wave sit 2 KB apart: read directly, every
// load touches 64 cache lines.  The wave therefore moves its 64 x kFiltBlock samples through LDS in tiles of
// kFiltTile columns: global accesses run along the blocks (two 256-byte runs per instruction), the recurrence reads its
// own row (row pitch kFiltTile + 1: conflict-free).  The block kernels went from 0.7 to ~3 TB/s of useful traffic.
// Tile width and the length from which the tiled form is used were set by tools/ab_filt.sh (order-3 band-pass, ms):
//            16M   24M   30M   40M   60M   80M  100M
//   direct  1.97  3.45  4.52
//   tile 32 2.27  2.93  3.30  5.17  6.12  8.19  9.13    (8.4 KB more LDS per wave: 9 waves per CU, a full round of the GPU
//   tile 16 2.26  3.11  3.42  3.83  4.88  7.18  8.37     is 37.7 M samples and 40 M took two)
//   tile  8 2.19  3.08  3.46  3.93  5.40  6.82  8.54
#ifndef PAR_FILT_TILE
#define PAR_FILT_TILE 16
#endif
#ifndef PAR_FILT_TILED_MIN
#define PAR_FILT_TILED_MIN 20000000ll
#endif
constexpr int kFiltTile = PAR_FILT_TILE;
template <bool WRITE>
__device__ __forceinline__ void sos_wave_blocks(const double* __restrict__ u, int64_t L, int reverse, const Biquad& q,
                                                int64_t nblk, double& z0, double& z1, double* __restrict__ y,
                                                double (*tile)[kFiltTile + 1]) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t bw = ((int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave) * kWave;   // the wave's first block
  const int64_t b = bw + lane;
  const int64_t kend = b < nblk ? ((b + 1) * kFiltBlock < L ? (b + 1) * kFiltBlock : L) : 0;        // this lane's block ends here
  const int col = lane & (kFiltTile - 1), half = lane / kFiltTile;                                   // loader role: column, row parity
  for (int c0 = 0; c0 < kFiltBlock; c0 += kFiltTile) {
    // load: rows = blocks bw .. bw+63, columns c0 .. c0+31 of each
#pragma unroll 4
    for (int row = half; row < kWave; row += kWave / kFiltTile) {
      const int64_t k = (bw + row) * kFiltBlock + c0 + col;
      tile[row][col] = k < L ? u[pass_index(k, L, reverse)] : 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int64_t k0 = b * kFiltBlock + c0;
#pragma unroll 4
    for (int t = 0; t < kFiltTile; ++t) {
      if (k0 + t < kend) {
        const double xv = tile[lane][t];
        const double yv = q.b0 * xv + z0;
        z0 = q.b1 * xv - q.a1 * yv + z1;
        z1 = q.b2 * xv - q.a2 * yv;
        if (WRITE) tile[lane][t] = yv;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (WRITE) {
#pragma unroll 4
      for (int row = half; row < kWave; row += kWave / kFiltTile) {
        const int64_t k = (bw + row) * kFiltBlock + c0 + col;
        if (k < L) y[pass_index(k, L, reverse)] = tile[row][col];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// zero-state response end state of each block
__global__ __launch_bounds__(kWave) void k_sos_block_zero(const double* __restrict__ u, int64_t L, int reverse, Biquad q,
                                                           int64_t nblk, double* __restrict__ r) {
  __shared__ double tile[kWave][kFiltTile + 1];
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double z0 = 0.0, z1 = 0.0;
  sos_wave_blocks<false>(u, L, reverse, q, nblk, z0, z1, nullptr, tile);
  if (b >= nblk) return;
  r[2 * b] = z0;
  r[2 * b + 1] = z1;
}

// zero-state response end state of each block, a lane reading its own block straight from memory (mid-sized inputs: more
// waves in flight than the LDS-tiled form allows)
__global__ void k_sos_block_zero_direct(const double* __restrict__ u, int64_t L, int reverse, Biquad q, int64_t nblk,
                                 double* __restrict__ r) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblk) return;
  const int64_t k0 = b * kFiltBlock;
  const int64_t k1 = k0 + kFiltBlock < L ? k0 + kFiltBlock : L;
  double z0 = 0.0, z1 = 0.0;
  for (int64_t k = k0; k < k1; ++k) {
    const double xv = u[pass_index(k, L, reverse)];
    const double y = q.b0 * xv + z0;
    z0 = q.b1 * xv - q.a1 * y + z1;
    z1 = q.b2 * xv - q.a2 * y;
  }
  r[2 * b] = z0;
  r[2 * b + 1] = z1;
}

// serial chain over block boundaries: s[b+1] = P * s[b] + r[b],  P = A^kFiltBlock (host-computed)
__global__ void k_sos_chain(const double* __restrict__ r, int64_t nblk, double p00, double p01, double p10, double p11,
                            double zi0, double zi1, const double* __restrict__ scale_sample, double* __restrict__ s) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double sc = *scale_sample;
  double z0 = zi0 * sc, z1 = zi1 * sc;
  for (int64_t b = 0; b < nblk; ++b) {
    s[2 * b] = z0;
    s[2 * b + 1] = z1;
    const double n0 = p00 * z0 + p01 * z1 + r[2 * b];
    const double n1 = p10 * z0 + p11 * z1 + r[2 * b + 1];
    z0 = n0;
    z1 = n1;
  }
}

// The chain over ~L/256 block boundaries is itself cut in two levels (a 10^8-sample signal has 4 10^5 blocks: one
// thread walking them all took 40 % of the filter's time): super-blocks of kFiltSuper blocks are chained from a zero
// state in parallel (k_sos_super_zero), one thread chains the few super-block boundaries with Q = P^kFiltSuper
// (k_sos_chain on the super-block records), and every super-block then re-walks its own blocks from its true state
// (k_sos_super_run).  Same affine maps, composed in a different order: float64 re-association only.
constexpr int kFiltSuper = 256;
__global__ void k_sos_super_zero(const double* __restrict__ r, int64_t nblk, double p00, double p01, double p10, double p11,
                                 int64_t nsup, double* __restrict__ R) {
  const int64_t B = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (B >= nsup) return;
  const int64_t b0 = B * kFiltSuper, b1 = b0 + kFiltSuper < nblk ? b0 + kFiltSuper : nblk;
  double z0 = 0.0, z1 = 0.0;
  for (int64_t b = b0; b < b1; ++b) {
    const double n0 = p00 * z0 + p01 * z1 + r[2 * b];
    const double n1 = p10 * z0 + p11 * z1 + r[2 * b + 1];
    z0 = n0;
    z1 = n1;
  }
  R[2 * B] = z0;
  R[2 * B + 1] = z1;
}
__global__ void k_sos_super_run(const double* __restrict__ r, int64_t nblk, double p00, double p01, double p10, double p11,
                                int64_t nsup, const double* __restrict__ S, double* __restrict__ s) {
  const int64_t B = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (B >= nsup) return;
  const int64_t b0 = B * kFiltSuper, b1 = b0 + kFiltSuper < nblk ? b0 + kFiltSuper : nblk;
  double z0 = S[2 * B], z1 = S[2 * B + 1];
  for (int64_t b = b0; b < b1; ++b) {
    s[2 * b] = z0;
    s[2 * b + 1] = z1;
    const double n0 = p00 * z0 + p01 * z1 + r[2 * b];
    const double n1 = p10 * z0 + p11 * z1 + r[2 * b + 1];
    z0 = n0;
    z1 = n1;
  }
}

__global__ __launch_bounds__(kWave) void k_sos_block_run(const double* __restrict__ u, int64_t L, int reverse, Biquad q,
                                                          int64_t nblk, const double* __restrict__ s, double* __restrict__ y) {
  __shared__ double tile[kWave][kFiltTile + 1];
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double z0 = b < nblk ? s[2 * b] : 0.0, z1 = b < nblk ? s[2 * b + 1] : 0.0;
  sos_wave_blocks<true>(u, L, reverse, q, nblk, z0, z1, y, tile);
}

__global__ void k_sos_block_run_direct(const double* __restrict__ u, int64_t L, int reverse, Biquad q, int64_t nblk,
                                const double* __restrict__ s, double* __restrict__ y) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblk) return;
  const int64_t k0 = b * kFiltBlock;
  const int64_t k1 = k0 + kFiltBlock < L ? k0 + kFiltBlock : L;
  double z0 = s[2 * b], z1 = s[2 * b + 1];
  for (int64_t k = k0; k < k1; ++k) {
    const int64_t idx = pass_index(k, L, reverse);
    const double xv = u[idx];
    const double yv = q.b0 * xv + z0;
    z0 = q.b1 * xv - q.a1 * yv + z1;
    z1 = q.b2 * xv - q.a2 * yv;
    y[idx] = yv;
  }
}

__global__ void k_copy_mid(const double* __restrict__ src, int64_t pad, int64_t n, double* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i + pad];
}


// ---- batched form (r05): B signals of one length, each with its own cascade (or all with one), per stage ONE launch -----------
// dropouts_gui.process_heuristic filters every channel of a file per band (dropouts_gui.py:314-321), a multi-band analysis
// filters one signal through many bands: a loop of par_sosfiltfilt_f64 calls costs ~20 latency-bound launches per signal
// (0.25 ms for 10^6 samples, of which the GPU is busy a few per cent).  Here the batch is the grid's y dimension.
struct FiltParam {                       // one (signal, section): coefficients, P = A^kFiltBlock, Q = P^kFiltSuper, zi
  double b0, b1, b2, a1, a2;
  double p00, p01, p10, p11;
  double q00, q01, q10, q11;
  double zi0, zi1, pad;
};
static_assert(sizeof(FiltParam) == 128, "FiltParam is uploaded as 16 doubles");

__global__ void k_odd_ext_b(const double* __restrict__ x, int64_t x_stride, int64_t n, int64_t pad, double* __restrict__ ext) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t L = n + 2 * pad;
  if (i >= L) return;
  const double* xs = x + (int64_t)blockIdx.y * x_stride;
  double v;
  if (i < pad) v = 2.0 * xs[0] - xs[pad - i];
  else if (i < pad + n) v = xs[i - pad];
  else v = 2.0 * xs[n - 1] - xs[n - 2 - (i - pad - n)];
  ext[(int64_t)blockIdx.y * L + i] = v;
}
// the sample that scales zi in this direction (first element of the cascade input), per signal
__global__ void k_latch_b(const double* __restrict__ cur, int64_t L, int reverse, int nsig, double* __restrict__ latch) {
  const int sg = blockIdx.x * blockDim.x + threadIdx.x;
  if (sg < nsig) latch[sg] = cur[(int64_t)sg * L + (reverse ? L - 1 : 0)];
}
__global__ void k_sos_block_zero_b(const double* __restrict__ u, int64_t L, int reverse, const FiltParam* __restrict__ fp,
                                   int fp_stride, int64_t nblk, double* __restrict__ r) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblk) return;
  const int sg = blockIdx.y;
  const FiltParam q = fp[(int64_t)sg * fp_stride];
  const double* us = u + (int64_t)sg * L;
  const int64_t k0 = b * kFiltBlock;
  const int64_t k1 = k0 + kFiltBlock < L ? k0 + kFiltBlock : L;
  double z0 = 0.0, z1 = 0.0;
  for (int64_t k = k0; k < k1; ++k) {
    const double xv = us[pass_index(k, L, reverse)];
    const double y = q.b0 * xv + z0;
    z0 = q.b1 * xv - q.a1 * y + z1;
    z1 = q.b2 * xv - q.a2 * y;
  }
  r[2 * ((int64_t)sg * nblk + b)] = z0;
  r[2 * ((int64_t)sg * nblk + b) + 1] = z1;
}
// ... and their LDS-tiled forms (sos_wave_blocks above: 3 TB/s instead of 0.7), for batches of >= PAR_FILT_TILED_MIN samples in all
__global__ __launch_bounds__(kWave) void k_sos_block_zero_bt(const double* __restrict__ u, int64_t L, int reverse,
                                                              const FiltParam* __restrict__ fp, int fp_stride, int64_t nblk,
                                                              double* __restrict__ r) {
  __shared__ double tile[kWave][kFiltTile + 1];
  const int sg = blockIdx.y;
  const FiltParam p = fp[(int64_t)sg * fp_stride];
  const Biquad q{p.b0, p.b1, p.b2, p.a1, p.a2};
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double z0 = 0.0, z1 = 0.0;
  sos_wave_blocks<false>(u + (int64_t)sg * L, L, reverse, q, nblk, z0, z1, nullptr, tile);
  if (b >= nblk) return;
  r[2 * ((int64_t)sg * nblk + b)] = z0;
  r[2 * ((int64_t)sg * nblk + b) + 1] = z1;
}
__global__ __launch_bounds__(kWave) void k_sos_block_run_bt(const double* __restrict__ u, int64_t L, int reverse,
                                                             const FiltParam* __restrict__ fp, int fp_stride, int64_t nblk,
                                                             const double* __restrict__ s, double* __restrict__ y) {
  __shared__ double tile[kWave][kFiltTile + 1];
  const int sg = blockIdx.y;
  const FiltParam p = fp[(int64_t)sg * fp_stride];
  const Biquad q{p.b0, p.b1, p.b2, p.a1, p.a2};
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double z0 = b < nblk ? s[2 * ((int64_t)sg * nblk + b)] : 0.0, z1 = b < nblk ? s[2 * ((int64_t)sg * nblk + b) + 1] : 0.0;
  sos_wave_blocks<true>(u + (int64_t)sg * L, L, reverse, q, nblk, z0, z1, y + (int64_t)sg * L, tile);
}
// super-blocks from a zero state (one thread per (signal, super-block))
__global__ void k_sos_super_zero_b(const double* __restrict__ r, int64_t nblk, const FiltParam* __restrict__ fp, int fp_stride,
                                   int64_t nsup, double* __restrict__ R) {
  const int64_t B = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (B >= nsup) return;
  const int sg = blockIdx.y;
  const FiltParam q = fp[(int64_t)sg * fp_stride];
  const double* rs = r + 2 * (int64_t)sg * nblk;
  const int64_t b0 = B * kFiltSuper, b1 = b0 + kFiltSuper < nblk ? b0 + kFiltSuper : nblk;
  double z0 = 0.0, z1 = 0.0;
  for (int64_t b = b0; b < b1; ++b) {
    const double n0 = q.p00 * z0 + q.p01 * z1 + rs[2 * b];
    const double n1 = q.p10 * z0 + q.p11 * z1 + rs[2 * b + 1];
    z0 = n0;
    z1 = n1;
  }
  R[2 * ((int64_t)sg * nsup + B)] = z0;
  R[2 * ((int64_t)sg * nsup + B) + 1] = z1;
}
// serial chain over the super-block boundaries (one thread per signal)
__global__ void k_sos_chain_b(const double* __restrict__ R, int64_t nsup, const FiltParam* __restrict__ fp, int fp_stride,
                              const double* __restrict__ latch, int nsig, double* __restrict__ S) {
  const int sg = blockIdx.x * blockDim.x + threadIdx.x;
  if (sg >= nsig) return;
  const FiltParam q = fp[(int64_t)sg * fp_stride];
  const double sc = latch[sg];
  double z0 = q.zi0 * sc, z1 = q.zi1 * sc;
  for (int64_t B = 0; B < nsup; ++B) {
    S[2 * ((int64_t)sg * nsup + B)] = z0;
    S[2 * ((int64_t)sg * nsup + B) + 1] = z1;
    const double n0 = q.q00 * z0 + q.q01 * z1 + R[2 * ((int64_t)sg * nsup + B)];
    const double n1 = q.q10 * z0 + q.q11 * z1 + R[2 * ((int64_t)sg * nsup + B) + 1];
    z0 = n0;
    z1 = n1;
  }
}
__global__ void k_sos_super_run_b(const double* __restrict__ r, int64_t nblk, const FiltParam* __restrict__ fp, int fp_stride,
                                  int64_t nsup, const double* __restrict__ S, double* __restrict__ s) {
  const int64_t B = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (B >= nsup) return;
  const int sg = blockIdx.y;
  const FiltParam q = fp[(int64_t)sg * fp_stride];
  const double* rs = r + 2 * (int64_t)sg * nblk;
  double* ss = s + 2 * (int64_t)sg * nblk;
  const int64_t b0 = B * kFiltSuper, b1 = b0 + kFiltSuper < nblk ? b0 + kFiltSuper : nblk;
  double z0 = S[2 * ((int64_t)sg * nsup + B)], z1 = S[2 * ((int64_t)sg * nsup + B) + 1];
  for (int64_t b = b0; b < b1; ++b) {
    ss[2 * b] = z0;
    ss[2 * b + 1] = z1;
    const double n0 = q.p00 * z0 + q.p01 * z1 + rs[2 * b];
    const double n1 = q.p10 * z0 + q.p11 * z1 + rs[2 * b + 1];
    z0 = n0;
    z1 = n1;
  }
}
__global__ void k_sos_block_run_b(const double* __restrict__ u, int64_t L, int reverse, const FiltParam* __restrict__ fp,
                                  int fp_stride, int64_t nblk, const double* __restrict__ s, double* __restrict__ y) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblk) return;
  const int sg = blockIdx.y;
  const FiltParam q = fp[(int64_t)sg * fp_stride];
  const double* us = u + (int64_t)sg * L;
  double* ys = y + (int64_t)sg * L;
  const int64_t k0 = b * kFiltBlock;
  const int64_t k1 = k0 + kFiltBlock < L ? k0 + kFiltBlock : L;
  double z0 = s[2 * ((int64_t)sg * nblk + b)], z1 = s[2 * ((int64_t)sg * nblk + b) + 1];
  for (int64_t k = k0; k < k1; ++k) {
    const int64_t idx = pass_index(k, L, reverse);
    const double xv = us[idx];
    const double yv = q.b0 * xv + z0;
    z0 = q.b1 * xv - q.a1 * yv + z1;
    z1 = q.b2 * xv - q.a2 * yv;
    ys[idx] = yv;
  }
}
__global__ void k_copy_mid_b(const double* __restrict__ src, int64_t L, int64_t pad, int64_t n, double* __restrict__ dst,
                             int64_t y_stride) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[(int64_t)blockIdx.y * y_stride + i] = src[(int64_t)blockIdx.y * L + i + pad];
}

static void mat_pow(double a00, double a01, double a10, double a11, int e, double* out) {
  double p00 = 1, p01 = 0, p10 = 0, p11 = 1;
  for (; e > 0; e >>= 1) {
    if (e & 1) {
      const double t00 = p00 * a00 + p01 * a10, t01 = p00 * a01 + p01 * a11;
      const double t10 = p10 * a00 + p11 * a10, t11 = p10 * a01 + p11 * a11;
      p00 = t00; p01 = t01; p10 = t10; p11 = t11;
    }
    const double s00 = a00 * a00 + a01 * a10, s01 = a00 * a01 + a01 * a11;
    const double s10 = a10 * a00 + a11 * a10, s11 = a10 * a01 + a11 * a11;
    a00 = s00; a01 = s01; a10 = s10; a11 = s11;
  }
  out[0] = p00; out[1] = p01; out[2] = p10; out[3] = p11;
}

}  // namespace par

extern "C" {

int64_t par_sosfiltfilt_work_len(int64_t n, int64_t padlen) {
  const int64_t L = n + 2 * padlen;
  const int64_t nblk = (L + par::kFiltBlock - 1) / par::kFiltBlock;
  const int64_t nsup = (nblk + par::kFiltSuper - 1) / par::kFiltSuper;
  return 2 * L + 4 * nblk + 4 * nsup + 8;
}

int par_sosfiltfilt_f64(int device, const double* sos, const double* zi, int n_sections, const double* x, int64_t n,
                        int64_t padlen, double* work, int64_t work_len, double* y, void* stream) {
  using namespace par;
  PAR_REQUIRE(sos && zi && x && work && y && n_sections >= 1 && n_sections <= 64, PAR_ERR_ARG,
              "par_sosfiltfilt_f64: bad args");
  PAR_REQUIRE(n > padlen && padlen >= 0, PAR_ERR_ARG,
              "par_sosfiltfilt_f64: the length of the input vector x must be greater than padlen, which is %lld",
              (long long)padlen);   // scipy's own ValueError text
  PAR_REQUIRE(work_len >= par_sosfiltfilt_work_len(n, padlen), PAR_ERR_WORKSPACE, "par_sosfiltfilt_f64: workspace too small");
  PAR_HIP_CHECK(hipSetDevice(device));
  hipStream_t st = as_stream(stream);
  const int64_t L = n + 2 * padlen;
  const int64_t nblk = ceil_div(L, kFiltBlock);
  double* bufA = work;
  double* bufB = work + L;
  const int64_t nsup = ceil_div(nblk, kFiltSuper);
  double* r = work + 2 * L;
  double* s = r + 2 * nblk;
  double* Rs = s + 2 * nblk + 2;          // [2 nsup] zero-state end states of the super-blocks (behind the two latch slots)
  double* Ss = Rs + 2 * nsup;             // [2 nsup] true start states of the super-blocks
  hipLaunchKernelGGL(k_odd_ext, dim3((unsigned)ceil_div(L, 256)), dim3(256), 0, st, x, n, padlen, bufA);
  double* cur = bufA;
  double* nxt = bufB;
  for (int dir = 0; dir < 2; ++dir) {
    // zi scales with the first sample this pass consumes: ext[0] forward, y[L-1] backward
    for (int sec = 0; sec < n_sections; ++sec) {
      const double* c = sos + 6 * sec;
      PAR_REQUIRE(c[3] == 1.0, PAR_ERR_ARG, "par_sosfiltfilt_f64: sos[%d,3] (a0) must be 1", sec);
      Biquad q{c[0], c[1], c[2], c[4], c[5]};
      // P = A^kFiltBlock, A = [[-a1, 1], [-a2, 0]]
      double p00 = 1, p01 = 0, p10 = 0, p11 = 1;
      double a00 = -q.a1, a01 = 1.0, a10 = -q.a2, a11 = 0.0;
      for (int e = kFiltBlock; e > 0; e >>= 1) {
        if (e & 1) {
          const double t00 = p00 * a00 + p01 * a10, t01 = p00 * a01 + p01 * a11;
          const double t10 = p10 * a00 + p11 * a10, t11 = p10 * a01 + p11 * a11;
          p00 = t00; p01 = t01; p10 = t10; p11 = t11;
        }
        const double s00 = a00 * a00 + a01 * a10, s01 = a00 * a01 + a01 * a11;
        const double s10 = a10 * a00 + a11 * a10, s11 = a10 * a01 + a11 * a11;
        a00 = s00; a01 = s01; a10 = s10; a11 = s11;
      }
      // the scale sample is the first element of the CASCADE input of this direction; after the first
      // section `cur` no longer holds it, so it is latched into work[-1] region (s tail) per direction.
      double* latch = s + 2 * nblk + (dir ? 1 : 0);
      if (sec == 0) {
        PAR_HIP_CHECK(hipMemcpyAsync(latch, cur + (dir ? L - 1 : 0), sizeof(double), hipMemcpyDeviceToDevice, st));
      }
      // LDS-tiled block kernels pay off from ~3 10^7 samples on (10^8: 27 -> 9 ms; 10^7: 1.6 -> 2.2 ms, measured)
      const bool tiled = L >= PAR_FILT_TILED_MIN;
      if (tiled) hipLaunchKernelGGL(k_sos_block_zero, dim3((unsigned)ceil_div(nblk, 64)), dim3(64), 0, st, cur, L, dir, q, nblk, r);
      else hipLaunchKernelGGL(k_sos_block_zero_direct, dim3((unsigned)ceil_div(nblk, 64)), dim3(64), 0, st, cur, L, dir, q, nblk, r);
      if (nsup <= 4) {
        hipLaunchKernelGGL(k_sos_chain, dim3(1), dim3(1), 0, st, r, nblk, p00, p01, p10, p11, zi[2 * sec], zi[2 * sec + 1],
                           latch, s);
      } else {
        // Q = P^kFiltSuper
        double q00 = 1, q01 = 0, q10 = 0, q11 = 1, b00 = p00, b01 = p01, b10 = p10, b11 = p11;
        for (int e = kFiltSuper; e > 0; e >>= 1) {
          if (e & 1) {
            const double t00 = q00 * b00 + q01 * b10, t01 = q00 * b01 + q01 * b11;
            const double t10 = q10 * b00 + q11 * b10, t11 = q10 * b01 + q11 * b11;
            q00 = t00; q01 = t01; q10 = t10; q11 = t11;
          }
          const double s00 = b00 * b00 + b01 * b10, s01 = b00 * b01 + b01 * b11;
          const double s10 = b10 * b00 + b11 * b10, s11 = b10 * b01 + b11 * b11;
          b00 = s00; b01 = s01; b10 = s10; b11 = s11;
        }
        hipLaunchKernelGGL(k_sos_super_zero, dim3((unsigned)ceil_div(nsup, 64)), dim3(64), 0, st, (const double*)r, nblk, p00, p01,
                           p10, p11, nsup, Rs);
        hipLaunchKernelGGL(k_sos_chain, dim3(1), dim3(1), 0, st, (const double*)Rs, nsup, q00, q01, q10, q11, zi[2 * sec],
                           zi[2 * sec + 1], latch, Ss);
        hipLaunchKernelGGL(k_sos_super_run, dim3((unsigned)ceil_div(nsup, 64)), dim3(64), 0, st, (const double*)r, nblk, p00, p01,
                           p10, p11, nsup, (const double*)Ss, s);
      }
      if (tiled) hipLaunchKernelGGL(k_sos_block_run, dim3((unsigned)ceil_div(nblk, 64)), dim3(64), 0, st, cur, L, dir, q, nblk, s, nxt);
      else hipLaunchKernelGGL(k_sos_block_run_direct, dim3((unsigned)ceil_div(nblk, 64)), dim3(64), 0, st, cur, L, dir, q, nblk, s, nxt);
      double* t = cur;
      cur = nxt;
      nxt = t;
    }
  }
  hipLaunchKernelGGL(k_copy_mid, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, cur, padlen, n, y);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

// per signal: two extended buffers, block states (zero-state + true), super-block states, latch; then the parameter table
int64_t par_sosfiltfilt_batch_work_len(int64_t n, int64_t padlen, int n_sig, int n_sections) {
  const int64_t L = n + 2 * padlen;
  const int64_t nblk = (L + par::kFiltBlock - 1) / par::kFiltBlock;
  const int64_t nsup = (nblk + par::kFiltSuper - 1) / par::kFiltSuper;
  return (int64_t)n_sig * (2 * L + 4 * nblk + 4 * nsup + 2) + (int64_t)n_sig * n_sections * 16 + 16;
}

int par_sosfiltfilt_batch_f64(int device, const double* sos, const double* zi, int n_filt, int n_sections, const double* x,
                              int64_t x_stride, int n_sig, int64_t n, int64_t padlen, double* work, int64_t work_len, double* y,
                              int64_t y_stride, void* stream) {
  using namespace par;
  PAR_REQUIRE(sos && zi && x && work && y && n_sections >= 1 && n_sections <= 64 && n_sig >= 1 && n_sig <= 65535, PAR_ERR_ARG,
              "par_sosfiltfilt_batch_f64: bad args");
  PAR_REQUIRE(n_filt == 1 || n_filt == n_sig, PAR_ERR_ARG,
              "par_sosfiltfilt_batch_f64: n_filt must be 1 (one cascade for all signals) or n_sig (one per signal), got %d for %d", n_filt,
              n_sig);
  PAR_REQUIRE(n > padlen && padlen >= 0, PAR_ERR_ARG,
              "par_sosfiltfilt_batch_f64: the length of the input vector x must be greater than padlen, which is %lld",
              (long long)padlen);
  PAR_REQUIRE(x_stride >= n && y_stride >= n, PAR_ERR_ARG, "par_sosfiltfilt_batch_f64: signal strides shorter than the signals");
  PAR_REQUIRE(work_len >= par_sosfiltfilt_batch_work_len(n, padlen, n_sig, n_sections), PAR_ERR_WORKSPACE,
              "par_sosfiltfilt_batch_f64: workspace too small");
  PAR_HIP_CHECK(hipSetDevice(device));
  hipStream_t st = as_stream(stream);
  const int64_t L = n + 2 * padlen;
  const int64_t nblk = ceil_div(L, kFiltBlock);
  const int64_t nsup = ceil_div(nblk, kFiltSuper);
  double* bufA = work;
  double* bufB = bufA + (int64_t)n_sig * L;
  double* r = bufB + (int64_t)n_sig * L;
  double* s = r + 2 * (int64_t)n_sig * nblk;
  double* Rs = s + 2 * (int64_t)n_sig * nblk;
  double* Ss = Rs + 2 * (int64_t)n_sig * nsup;
  double* latch = Ss + 2 * (int64_t)n_sig * nsup;            // [2][n_sig]
  FiltParam* fp = reinterpret_cast<FiltParam*>(latch + 2 * (int64_t)n_sig + ((2 * (int64_t)n_sig) & 1));   // 16-byte aligned
  // parameter table [n_sig][n_sections] (a shared cascade is replicated: the kernels index by signal)
  std::vector<FiltParam> host((size_t)n_sig * n_sections);
  for (int sg = 0; sg < n_sig; ++sg) {
    const int f = n_filt == 1 ? 0 : sg;
    for (int sec = 0; sec < n_sections; ++sec) {
      const double* c = sos + 6 * ((size_t)f * n_sections + sec);
      PAR_REQUIRE(c[3] == 1.0, PAR_ERR_ARG, "par_sosfiltfilt_batch_f64: sos[%d][%d,3] (a0) must be 1", f, sec);
      FiltParam& q = host[(size_t)sg * n_sections + sec];
      q.b0 = c[0]; q.b1 = c[1]; q.b2 = c[2]; q.a1 = c[4]; q.a2 = c[5];
      double P[4], Q[4];
      mat_pow(-q.a1, 1.0, -q.a2, 0.0, kFiltBlock, P);
      mat_pow(P[0], P[1], P[2], P[3], kFiltSuper, Q);
      q.p00 = P[0]; q.p01 = P[1]; q.p10 = P[2]; q.p11 = P[3];
      q.q00 = Q[0]; q.q01 = Q[1]; q.q10 = Q[2]; q.q11 = Q[3];
      q.zi0 = zi[2 * ((size_t)f * n_sections + sec)];
      q.zi1 = zi[2 * ((size_t)f * n_sections + sec) + 1];
      q.pad = 0.0;
    }
  }
  PAR_HIP_CHECK(hipMemcpyAsync(fp, host.data(), host.size() * sizeof(FiltParam), hipMemcpyHostToDevice, st));
  PAR_HIP_CHECK(hipStreamSynchronize(st));                   // `host` is pageable and dies with this call
  const dim3 gL((unsigned)ceil_div(L, 256), (unsigned)n_sig), gB((unsigned)ceil_div(nblk, 64), (unsigned)n_sig),
      gS((unsigned)ceil_div(nsup, 64), (unsigned)n_sig), gN((unsigned)ceil_div((int64_t)n_sig, 64));
  const bool tiled = (int64_t)n_sig * L >= PAR_FILT_TILED_MIN;     // the batch as a whole fills the GPU like one long signal does
  hipLaunchKernelGGL(k_odd_ext_b, gL, dim3(256), 0, st, x, x_stride, n, padlen, bufA);
  double* cur = bufA;
  double* nxt = bufB;
  for (int dir = 0; dir < 2; ++dir) {
    hipLaunchKernelGGL(k_latch_b, gN, dim3(64), 0, st, (const double*)cur, L, dir, n_sig, latch + (int64_t)dir * n_sig);
    for (int sec = 0; sec < n_sections; ++sec) {
      const FiltParam* fps = fp + sec;
      if (tiled) hipLaunchKernelGGL(k_sos_block_zero_bt, gB, dim3(64), 0, st, (const double*)cur, L, dir, fps, n_sections, nblk, r);
      else hipLaunchKernelGGL(k_sos_block_zero_b, gB, dim3(64), 0, st, (const double*)cur, L, dir, fps, n_sections, nblk, r);
      hipLaunchKernelGGL(k_sos_super_zero_b, gS, dim3(64), 0, st, (const double*)r, nblk, fps, n_sections, nsup, Rs);
      hipLaunchKernelGGL(k_sos_chain_b, gN, dim3(64), 0, st, (const double*)Rs, nsup, fps, n_sections,
                         (const double*)(latch + (int64_t)dir * n_sig), n_sig, Ss);
      hipLaunchKernelGGL(k_sos_super_run_b, gS, dim3(64), 0, st, (const double*)r, nblk, fps, n_sections, nsup, (const double*)Ss, s);
      if (tiled) hipLaunchKernelGGL(k_sos_block_run_bt, gB, dim3(64), 0, st, (const double*)cur, L, dir, fps, n_sections, nblk, (const double*)s, nxt);
      else hipLaunchKernelGGL(k_sos_block_run_b, gB, dim3(64), 0, st, (const double*)cur, L, dir, fps, n_sections, nblk, (const double*)s, nxt);
      double* t = cur;
      cur = nxt;
      nxt = t;
    }
  }
  hipLaunchKernelGGL(k_copy_mid_b, dim3((unsigned)ceil_div(n, 256), (unsigned)n_sig), dim3(256), 0, st, (const double*)cur, L, padlen, n, y,
                     y_stride);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

}  // extern "C"
