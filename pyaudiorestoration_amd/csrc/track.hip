// K_track -- wow/flutter pitch trackers on a device-resident magnitude spectrogram.
//
// Semantics (reference util/wow_detection.py): Track.freq_plus_tolerance :109-117, set_bin_limits
// :97-107, freq_2_bin :81-82, get_peak :119-134, is_peak :136-139, PeakTracker.trace :298-304,
// PeakTrackTracker.trace :310-327, CenterOfGravity :259-291; correlation.parabolic
// (util/correlation.py:42-46).  The spectrogram is FRAME-MAJOR float32 (what K_stft writes), so a
// frame's band is a contiguous run.  Band/bin arithmetic is float64 like the reference.
#include "par_common.h"
#include <math.h>
#include <mutex>
#include <set>

namespace par {

struct Band {
  int NL, NU;
  bool past_end;       // NU reached past the last bin before the clip (a numpy slice clips silently; a product with a
                       // full-length window does not)
};

__device__ __forceinline__ int freq_to_bin(double f, int fft_size, double sr, int bins) {
  long long b = llrint(f * (double)fft_size / sr);        // Python round(): half-to-even
  if (b > bins - 1) b = bins - 1;
  if (b < 1) b = 1;
  return (int)b;
}

__device__ __forceinline__ Band band_limits(double freq, double tol, int fft_size, double sr, int bins) {
  const double lf = log2(freq);
  double fL = exp2(lf - tol), fU = exp2(lf + tol);
  fL = fL > 1.0 ? fL : 1.0;                                // max(1.0, fL)
  fU = fU < sr / 2 ? fU : sr / 2;                          // min(sr/2, fU)
  Band b;
  b.NL = freq_to_bin(fL, fft_size, sr, bins);
  b.NU = freq_to_bin(fU, fft_size, sr, bins);
  while (b.NU - b.NL < 4) {                                // min_bins = 4
    b.NL -= 1;
    b.NU += 1;
  }
  // A band widened below bin 0 (both edges on bin 1: a frequency below the transform's resolution) is an EMPTY
  // numpy slice [-k:NU] in the reference, whose argmax raises; reported through `empty`, never silently clamped.
  b.past_end = b.NU > bins;
  if (b.NU > bins) b.NU = bins;
  return b;
}

// status bits: 1 empty band (ValueError), 2 peak on the last bin: is_peak() reads fft_frame[peak_i + 1] (IndexError),
// 4 band widened past the last bin: window(NU - NL) * spectrum[NL:NU] does not broadcast (ValueError); freq_2_bin caps
// both edges at bins - 1 and min_bins is 4, so such a slice always keeps >= 3 bins against a window of >= 4
__device__ __forceinline__ double peak_freq(const float* __restrict__ col, Band b, int bins, int fft_size, double sr,
                                            int* __restrict__ status) {
  int arg = b.NL;
  float best = col[b.NL];
  for (int k = b.NL + 1; k < b.NU; ++k) {
    const float v = col[k];
    if (v > best) {                                        // first occurrence of the maximum (np.argmax)
      best = v;
      arg = k;
    }
  }
  double x = (double)arg;
  if (b.past_end) {
    atomicOr(status, 4);
    return x / (double)fft_size * sr;
  }
  if (arg == bins - 1) {
    atomicOr(status, 2);
    return x / (double)fft_size * sr;
  }
  const double fm = (double)col[(arg - 1 + bins) % bins], f0 = (double)col[arg], fp = (double)col[arg + 1];
  if (fm < f0 && f0 > fp) {
    // parabolic(): xv = 1/2*(f[x-1]-f[x+1]) / (f[x-1]-2f[x]+f[x+1]) + x
    x = 0.5 * (fm - fp) / (fm - 2.0 * f0 + fp) + x;
  }
  return x / (double)fft_size * sr;
}

__device__ __forceinline__ bool empty_band(const Band& b, int* __restrict__ empty) {
  if (b.NL >= 0 && b.NL < b.NU) return false;
  atomicOr(empty, 1);
  return true;
}

__global__ void k_track_peak(const float* __restrict__ mag, int bins, int64_t pitch, int64_t frame_0, int64_t count,
                             double* __restrict__ freqs, int fft_size, double sr, double tol, int* __restrict__ empty) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const Band b = band_limits(freqs[i], tol, fft_size, sr, bins);     // PeakTracker: band follows the drawn trail
  if (empty_band(b, empty)) return;
  freqs[i] = peak_freq(mag + (frame_0 + i) * pitch, b, bins, fft_size, sr, empty);
}

// PeakTrackTracker: band fixed on the first trail frequency (read back by the host entry point).
__global__ void k_track_peak_fixed(const float* __restrict__ mag, int bins, int64_t pitch, int64_t frame_0, int64_t count,
                                   double centre, double* __restrict__ freqs, int fft_size, double sr, double tol,
                                   int* __restrict__ empty) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const Band b = band_limits(centre, i > 2 ? tol / 2 : tol, fft_size, sr, bins);
  if (empty_band(b, empty)) return;
  freqs[i] = peak_freq(mag + (frame_0 + i) * pitch, b, bins, fft_size, sr, empty);
}

// Peak / Peak Track on band magnitudes re-evaluated from the SIGNAL in float64 (r03).  The reference's numpy backend hands
// its trackers float64 CONTAINERS (util/fourier.py:136-157: float32 frames -> pocketfft -> / np.float64(sqrt(n_fft))) whose
// values carry pocketfft's single-precision rounding (numpy >= 2); this kernel evaluates the exact DFT of the same float32
// frame, which is not bit-parity with either backend but sits inside their spread (profiles/r03_p0_sensitivity.txt).
// K_stft's magnitudes are float32 and a few 1e-8 relative noisier than one float32 rounding, which the config-3 chain
// (running-sum positions, a window centre that jumps at half-integers) amplifies to 3.7e-5 of the output peak
// (profiles/r02_p0_sensitivity.txt).  Only the band [NL, NU) and the two neighbours of its peak matter to the tracker:
// <= a dozen bins x n_fft samples per frame, a windowed direct DFT in float64 -- ~0.03 GFLOP on config 3.  A thread's samples
// lie 256 apart: it takes its first twiddle from sincospi and turns it by exp(-2 pi i 256 k / N) from there (one sincospi
// pair per thread and bin instead of one per sample: ADVICE r03); wide bands on long transforms are refused by the caller
// (wow_detection._trace_refined: bins x n_fft x frames above 2e10 takes the spectrogram path).
// One 256-thread workgroup per frame.  The frame is the reference's: reflect-padded by n_fft/2, sample x window rounded to
// float32 (segment_array, util/fourier.py:160-166), zero-extended to n_fft * zeropad, spectrum / sqrt(n_fft), + 1e-7.
constexpr int kRefineMaxBins = 2048;
__device__ __forceinline__ double refined_mag(const float* __restrict__ x, int64_t n, int64_t xs, const float* __restrict__ win,
                                              int n_fft, int hop, int N, int64_t frame, int k, double* red) {
  // all 256 threads: sum_n xw[n] exp(-2 pi i k n / N)
  double re = 0.0, im = 0.0;
  const int64_t base = frame * hop - n_fft / 2;
  const int64_t m = 2 * (n - 1);
  double sn, cs, rs, rc;                                     // twiddle of this thread's current sample; its step per 256 samples
  sincospi(2.0 * (double)(((long long)k * threadIdx.x) % N) / (double)N, &sn, &cs);
  sincospi(2.0 * (double)(((long long)k * blockDim.x) % N) / (double)N, &rs, &rc);
  for (int q = threadIdx.x; q < n_fft; q += blockDim.x) {
    int64_t j = base + q;
    if (j < 0 || j >= n) {                                   // np.pad(..., mode='reflect')
      j = ((j % m) + m) % m;
      if (j >= n) j = m - j;
    }
    const float xw = x[j * xs] * win[q];                     // float32 product, like the reference's frame matrix
    re += (double)xw * cs;
    im -= (double)xw * sn;
    const double c2 = cs * rc - sn * rs;                     // (<= n_fft / 256 turns: 1e-15 of drift at 16384 points)
    sn = sn * rc + cs * rs;
    cs = c2;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    re += __shfl_xor(re, o, kWave);
    im += __shfl_xor(im, o, kWave);
  }
  __syncthreads();                                           // red[] free again
  if ((threadIdx.x & (kWave - 1)) == 0) {
    red[2 * (threadIdx.x / kWave)] = re;
    red[2 * (threadIdx.x / kWave) + 1] = im;
  }
  __syncthreads();
  re = im = 0.0;
  for (int w = 0; w < (int)(blockDim.x / kWave); ++w) {
    re += red[2 * w];
    im += red[2 * w + 1];
  }
  return sqrt(re * re + im * im) / sqrt((double)n_fft) + 1e-7;
}

__global__ __launch_bounds__(256) void k_track_peak_refined(const float* __restrict__ x, int64_t n, int64_t xs,
                                                            const float* __restrict__ win, int n_fft, int hop, int zeropad,
                                                            int bins, int64_t frame_0, int64_t count, int mode, double centre,
                                                            double* __restrict__ freqs, double sr, double tol,
                                                            int* __restrict__ status) {
  __shared__ double red[8];
  extern __shared__ double bandmag[];                        // [NU - NL]
  const int64_t i = blockIdx.x;
  const int fft_size = n_fft * zeropad;
  const Band b = mode == 0 ? band_limits(freqs[i], tol, fft_size, sr, bins)
                           : band_limits(centre, i > 2 ? tol / 2 : tol, fft_size, sr, bins);
  if (b.NL < 0 || b.NL >= b.NU) {
    if (threadIdx.x == 0) atomicOr(status, 1);
    return;
  }
  const int nb = b.NU - b.NL;
  if (nb > kRefineMaxBins) {
    if (threadIdx.x == 0) atomicOr(status, 8);
    return;
  }
  const int64_t frame = frame_0 + i;
  for (int k = 0; k < nb; ++k) {
    const double v = refined_mag(x, n, xs, win, n_fft, hop, fft_size, frame, b.NL + k, red);
    if (threadIdx.x == 0) bandmag[k] = v;
  }
  __syncthreads();
  int arg = b.NL;
  double best = bandmag[0];
  for (int k = 1; k < nb; ++k) {                             // every thread the same walk: first occurrence of the maximum
    if (bandmag[k] > best) {
      best = bandmag[k];
      arg = b.NL + k;
    }
  }
  double xq = (double)arg;
  if (b.past_end || arg == bins - 1) {                       // uniform across the block
    if (threadIdx.x == 0) {
      atomicOr(status, b.past_end ? 4 : 2);
      freqs[i] = xq / (double)fft_size * sr;
    }
    return;
  }
  const int km = (arg - 1 + bins) % bins, kp = arg + 1;
  const double fm = (km >= b.NL && km < b.NU) ? bandmag[km - b.NL] : refined_mag(x, n, xs, win, n_fft, hop, fft_size, frame, km, red);
  const double fp = (kp >= b.NL && kp < b.NU) ? bandmag[kp - b.NL] : refined_mag(x, n, xs, win, n_fft, hop, fft_size, frame, kp, red);
  if (fm < best && best > fp) xq = 0.5 * (fm - fp) / (fm - 2.0 * best + fp) + xq;     // parabolic()
  if (threadIdx.x == 0) freqs[i] = xq / (double)fft_size * sr;
}

// CenterOfGravity: the band of frame i+1 depends on the result of frame i -> one wave walks the frames,
// its 64 lanes share the bins of the current band.
__global__ __launch_bounds__(64) void k_track_cog(const float* __restrict__ mag, int bins, int64_t pitch, int64_t frame_0, int64_t count,
                                                   double* __restrict__ freqs, int fft_size, double sr, double tol,
                                                   int* __restrict__ empty) {
  // Frame i + 1's band follows from frame i's result: the frames are walked by one wave.  What the walk can hide it
  // hides: the band moves by whole bins and rarely, so the Hann weights and log2 of the bin frequencies are kept per
  // lane while (NL, NU) stay put, and frame i + 1's magnitudes are fetched with frame i's band before frame i is
  // reduced (a changed band refetches).  Bands wider than one wave's 64 lanes take the plain loop.
  const int lane = threadIdx.x;
  Band b = band_limits(freqs[0], tol, fft_size, sr, bins);
  int cNL = -1, cNU = -1;                  // the band the cached per-lane terms belong to
  double cw = 0.0, clog = 0.0;             // np.hanning(L)[lane], log2(frequency of bin NL + lane)
  float pre = 0.0f;                        // magnitude of bin pNL + lane of the frame about to be reduced
  int pNL = -1, pNU = -1;
  for (int64_t i = 0; i < count; ++i) {
    if (b.NL < 0 || b.NL >= b.NU) {                        // the reference's 0/0 centroid -> NaN -> int(round(nan)) raises
      if (lane == 0) atomicOr(empty, 1);
      return;
    }
    if (b.past_end) {                                      // hanning(NU - NL) against a shorter slice: ValueError
      if (lane == 0) atomicOr(empty, 4);
      return;
    }
    const float* col = mag + (frame_0 + i) * pitch;
    const int L = b.NU - b.NL;
    double num = 0.0, den = 0.0;
    if (L <= kWave) {
      if (b.NL != cNL || b.NU != cNU) {
        // np.hanning(L)[k] = 0.5 - 0.5*cos(2*pi*k/(L-1));  hanning(1) == [1.]
        cw = lane < L ? ((L == 1) ? 1.0 : 0.5 - 0.5 * cos(2.0 * M_PI * (double)lane / (double)(L - 1))) : 0.0;
        clog = lane < L ? log2((double)(b.NL + lane) / (double)fft_size * sr) : 0.0;
        cNL = b.NL;
        cNU = b.NU;
      }
      float m = (pNL == b.NL && pNU == b.NU) ? pre : (lane < L ? col[b.NL + lane] : 0.0f);
      if (i + 1 < count) {                                 // next frame, this band: in flight under the reduction below
        pre = lane < L ? col[pitch + b.NL + lane] : 0.0f;
        pNL = b.NL;
        pNU = b.NU;
      }
      const double wm = cw * (double)m;
      num = wm * clog;
      den = wm;
    } else {
      pNL = -1;
      for (int k = lane; k < L; k += kWave) {
        const double w = 0.5 - 0.5 * cos(2.0 * M_PI * (double)k / (double)(L - 1));
        const double wm = w * (double)col[b.NL + k];
        const double fr = (double)(b.NL + k) / (double)fft_size * sr;
        num += wm * log2(fr);
        den += wm;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      num += __shfl_xor(num, o, kWave);
      den += __shfl_xor(den, o, kWave);
    }
    const double f = exp2(num / den);
    if (lane == 0) freqs[i] = f;
    b = band_limits(f, tol, fft_size, sr, bins);
  }
}

// ---- CorrelationTracker (util/wow_detection.py:396-436) ----------------------------------------------------------
// Per frame i of the band [NL, NU): the band's magnitudes are resampled onto a uniform log2-frequency grid of
// n = 4 (NU - NL) points by a quadratic interpolating spline (interp1d(kind='quadratic')), Hann-windowed,
// cross-correlated with the next frame's (the frame behind the last one is all ones, as the reference's buffer is
// initialised), and the parabola-refined peak offset from the centre is that frame's log-frequency change.  The spline
// is linear in the data and its abscissae are the same for every frame, so the host hands over the n x nb matrix that
// maps band values to grid values (scipy's own spline applied to the identity) and the resampling is a small dense
// product.  Everything in float64 like the reference; all frames in one launch.
//   k_corr_resample: R[i][g] = hann_g * sum_b M[g][b] mag[i][NL + b]   (i = count: R = hann_g)
//   k_corr_peak:     same[j] = sum_m a[m + j - n/2] b[m] / (|a| |b|)  ('same' part of scipy.signal.correlate), first
//                    argmax, parabolic() with the reference's f[-1] wrap at peak 0; peak n-1 is its IndexError
//   k_corr_finish:   np.cumsum in order, scaled to octaves, freqs = 2^(log2 mean + drift)
__global__ __launch_bounds__(256) void k_corr_resample(const float* __restrict__ mag, int64_t bins, int NL, int nb, int64_t count,
                                                       const double* __restrict__ M, const double* __restrict__ wind, int n,
                                                       double* __restrict__ R) {
  extern __shared__ double yb[];                           // the frame's band
  const int64_t i = blockIdx.x;
  for (int b = threadIdx.x; b < nb; b += blockDim.x) yb[b] = i < count ? (double)mag[i * bins + NL + b] : 0.0;
  __syncthreads();
  for (int g = threadIdx.x; g < n; g += blockDim.x) {
    double acc = 1.0;
    if (i < count) {
      acc = 0.0;
      const double* row = M + (int64_t)g * nb;
      for (int b = 0; b < nb; ++b) acc += row[b] * yb[b];
    }
    R[i * n + g] = acc * wind[g];
  }
}

// One workgroup per frame pair.  No `same` array: a thread keeps the running (value, first index) maximum of the lags it
// evaluates, the workgroup reduces those pairs (ties -> the smaller lag, np.argmax's first occurrence), and the two
// neighbours of the winner are evaluated once more by the same expression (bit-identical to what the owner of that lag
// computed).  LDS_AB: the two frames sit in dynamic LDS (2 n doubles, raised above 64 KB by the host wrapper up to
// n = 8192); wider bands read them from HBM/L2.
template <bool LDS_AB>
__global__ __launch_bounds__(256) void k_corr_peak(const double* __restrict__ R, int n, double* __restrict__ changes,
                                                   int* __restrict__ status) {
  extern __shared__ double sh[];                           // a[n], b[n] when LDS_AB
  __shared__ double red[2][4];
  __shared__ double best_v[4];
  __shared__ int best_j[4];
  const int64_t i = blockIdx.x;
  const double* ga = R + i * n;
  const double* gb = R + (i + 1) * n;
  const double* a = LDS_AB ? sh : ga;
  const double* b = LDS_AB ? sh + n : gb;
  double na = 0.0, nbv = 0.0;
  for (int g = threadIdx.x; g < n; g += blockDim.x) {
    const double av = ga[g], bv = gb[g];
    if (LDS_AB) {
      sh[g] = av;
      sh[n + g] = bv;
    }
    na += av * av;
    nbv += bv * bv;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    na += __shfl_xor(na, o, kWave);
    nbv += __shfl_xor(nbv, o, kWave);
  }
  if ((threadIdx.x & (kWave - 1)) == 0) {
    red[0][threadIdx.x / kWave] = na;
    red[1][threadIdx.x / kWave] = nbv;
  }
  __syncthreads();
  const double inv = 1.0 / (sqrt(red[0][0] + red[0][1] + red[0][2] + red[0][3]) * sqrt(red[1][0] + red[1][1] + red[1][2] + red[1][3]));
  const int half = n / 2;
  auto same_at = [&](int j) {
    double acc = 0.0;
    const int lo = half - j > 0 ? half - j : 0;            // m + j - half >= 0
    const int hi = n + half - j < n ? n + half - j : n;    // m + j - half < n
    for (int m = lo; m < hi; ++m) acc += a[m + j - half] * b[m];
    return acc * inv;
  };
  double bv = -INFINITY;
  int bj = 0x7fffffff;
  bool seen_nan = false;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const double v = same_at(j);
    if (v != v && !seen_nan) {                             // np.argmax returns the first NaN
      seen_nan = true;
      bv = v;
      bj = j;
    }
    if (!seen_nan && v > bv) {                             // ascending j per thread: strict > keeps the first occurrence
      bv = v;
      bj = j;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(bv, o, kWave);
    const int oj = __shfl_xor(bj, o, kWave);
    const bool mine_nan = bv != bv, other_nan = ov != ov;
    const bool take = (other_nan && (!mine_nan || oj < bj)) || (!mine_nan && !other_nan && (ov > bv || (ov == bv && oj < bj)));
    if (take) {
      bv = ov;
      bj = oj;
    }
  }
  if ((threadIdx.x & (kWave - 1)) == 0) {
    best_v[threadIdx.x / kWave] = bv;
    best_j[threadIdx.x / kWave] = bj;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x / kWave); ++w) {
      const double ov = best_v[w];
      const int oj = best_j[w];
      const bool mine_nan = bv != bv, other_nan = ov != ov;
      if ((other_nan && (!mine_nan || oj < bj)) || (!mine_nan && !other_nan && (ov > bv || (ov == bv && oj < bj)))) {
        bv = ov;
        bj = oj;
      }
    }
    const int arg = bj < n ? bj : 0;
    if (arg == n - 1) atomicOr(status, 2);                 // parabolic() reads f[x + 1]: IndexError in the reference
    const double fm = same_at(arg == 0 ? n - 1 : arg - 1), f0 = same_at(arg), fp = same_at(arg == n - 1 ? arg : arg + 1);
    const double refined = 0.5 * (fm - fp) / (fm - 2.0 * f0 + fp) + (double)arg;
    changes[i] = (double)half - refined;
  }
}

__global__ void k_corr_finish(const double* __restrict__ changes, int64_t count, int n, double log_span, double log_mean,
                              double* __restrict__ freqs) {
  double c = 0.0;
  for (int64_t i = 0; i < count; ++i) {                    // np.cumsum: strictly in order
    c = c + changes[i];
    freqs[i] = exp2(log_mean + c / (double)n * log_span);
  }
}

// ---- zero crossings (ZeroCrossingTracker, util/wow_detection.py:340, 448-450) -----------------------------
// indices i with (x[i+1] > 0) != (x[i] > 0), ascending: count per 1024-sample tile, scan of the tile counts,
// ordered write (ballot + popcount inside a wave, LDS prefix across the 4 waves of a tile).
constexpr int kZcTile = 1024;

__device__ __forceinline__ bool crossing_at(const double* __restrict__ x, int64_t i, int64_t n) {
  return i + 1 < n && ((x[i + 1] > 0.0) != (x[i] > 0.0));
}

__global__ __launch_bounds__(256) void k_zc_count(const double* __restrict__ x, int64_t n, long long* __restrict__ counts) {
  __shared__ int part[4];
  const int64_t base = (int64_t)blockIdx.x * kZcTile;
  int c = 0;
#pragma unroll
  for (int q = 0; q < kZcTile / 256; ++q) c += crossing_at(x, base + q * 256 + threadIdx.x, n) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = (long long)part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of the tile counts by one workgroup (chunks of 1024 with a running carry); counts[n_tiles] = total
__global__ __launch_bounds__(1024) void k_zc_scan(long long* __restrict__ counts, int64_t n_tiles) {
  __shared__ long long buf[1024];
  __shared__ long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t c0 = 0; c0 < n_tiles; c0 += 1024) {
    const int64_t i = c0 + threadIdx.x;
    const long long v = i < n_tiles ? counts[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                 // Hillis-Steele inclusive scan
      const long long t = threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < n_tiles) counts[i] = carry + buf[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += buf[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[n_tiles] = carry;
}

__global__ __launch_bounds__(256) void k_zc_write(const double* __restrict__ x, int64_t n, const long long* __restrict__ offs,
                                                  long long* __restrict__ idx) {
  __shared__ int wave_cnt[4];
  const int64_t base = (int64_t)blockIdx.x * kZcTile;
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  long long out = offs[blockIdx.x];
  // a wave owns 64 consecutive samples per round, 4 rounds of 256: keeps the indices ascending
  for (int q = 0; q < kZcTile / 256; ++q) {
    const int64_t i = base + q * 256 + threadIdx.x;
    const bool hit = crossing_at(x, i, n);
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wave_cnt[w] = __popcll(m);
    __syncthreads();
    long long before = 0;
    for (int k = 0; k < w; ++k) before += wave_cnt[k];
    if (hit) idx[out + before + __popcll(m & ((1ull << lane) - 1ull))] = (long long)i;
    out += (long long)wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
}

// PartialsTracker (util/wow_detection.py:361-387) calls librosa.piptrack, a third-party routine that is not part of the
// reference checkout (librosa is unpinned in requirements.txt; restated from its published 0.10 source, parity unpinned):
// per frame, every local maximum of the thresholded magnitude column inside [fmin, fmax) becomes a pitch
//   (k + shift) sr / n_fft,  shift = -b/a,  a = S[k+1] + S[k-1] - 2 S[k],  b = (S[k+1] - S[k-1]) / 2   (0 unless |b| < |a|)
// with magnitude S[k] + b shift / 2; everything else is 0.  Threshold = `threshold` x the column maximum; local maximum:
// x[k] > x[k-1] and x[k] >= x[k+1] on the edge-padded, thresholded column.  One wave per frame; S = (mag - offset) * scale
// undoes get_mag's + 1e-7 and 1/sqrt(n_fft) so the magnitudes are librosa's |stft|.
__global__ __launch_bounds__(256) void k_piptrack(const float* __restrict__ mag, int bins, int64_t pitch, int64_t n_frames, float scale,
                                                  float offset, int fft_size, double sr, double fmin, double fmax,
                                                  float threshold, float* __restrict__ pitches, float* __restrict__ mags) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t fr = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  if (fr >= n_frames) return;
  const float* row = mag + fr * pitch;
  float m = -INFINITY;
  for (int k = lane; k < bins; k += kWave) {
    const float x = (row[k] - offset) * scale;
    m = x > m ? x : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float t = __shfl_xor(m, o, kWave);
    m = t > m ? t : m;
  }
  const float ref = threshold * m;
  for (int k = lane; k < bins; k += kWave) {
    const float x = (row[k] - offset) * scale;
    const float xm = k > 0 ? (row[k - 1] - offset) * scale : x;
    const float xp = k + 1 < bins ? (row[k + 1] - offset) * scale : x;
    const float t = x > ref ? x : 0.0f, tm = xm > ref ? xm : 0.0f, tp = xp > ref ? xp : 0.0f;
    const double f = (double)k * sr / (double)fft_size;
    float pitch = 0.0f, magv = 0.0f;
    if (t > tm && t >= tp && fmin <= f && f < fmax) {
      float shift = 0.0f, avg = 0.0f;
      if (k > 0 && k + 1 < bins) {
        const float a = xp + xm - 2.0f * x, b = (xp - xm) / 2.0f;
        shift = fabsf(b) < fabsf(a) ? -b / a : 0.0f;
        avg = b;
      }
      pitch = (float)(((double)k + (double)shift) * sr / (double)fft_size);
      magv = x + 0.5f * avg * shift;
    }
    pitches[fr * bins + k] = pitch;
    mags[fr * bins + k] = magv;
  }
}

}  // namespace par

extern "C" {

int64_t par_zero_crossings_work_len(int64_t n) { return n < 2 ? 2 : (n + par::kZcTile - 1) / par::kZcTile + 1; }

int par_zero_crossings_f64(int device, const double* x, int64_t n, int64_t* work, int64_t* idx, int64_t cap, int64_t* count,
                           void* stream) {
  using namespace par;
  PAR_REQUIRE(x && work && count && n >= 0, PAR_ERR_ARG, "par_zero_crossings_f64: bad args");
  *count = 0;
  if (n < 2) return PAR_OK;
  PAR_HIP_CHECK(hipSetDevice(device));
  hipStream_t s = as_stream(stream);
  const int64_t n_tiles = ceil_div(n, kZcTile);
  long long* offs = reinterpret_cast<long long*>(work);
  hipLaunchKernelGGL(k_zc_count, dim3((unsigned)n_tiles), dim3(256), 0, s, x, n, offs);
  hipLaunchKernelGGL(k_zc_scan, dim3(1), dim3(1024), 0, s, offs, n_tiles);
  PAR_HIP_CHECK(hipGetLastError());
  long long total = 0;
  PAR_HIP_CHECK(hipMemcpyAsync(&total, offs + n_tiles, sizeof(total), hipMemcpyDeviceToHost, s));
  PAR_HIP_CHECK(hipStreamSynchronize(s));
  *count = (int64_t)total;
  if (!idx) return PAR_OK;                                  // query form: count only
  PAR_REQUIRE(cap >= total, PAR_ERR_WORKSPACE, "par_zero_crossings_f64: %lld crossings do not fit idx[%lld]", total,
              (long long)cap);
  if (total == 0) return PAR_OK;
  hipLaunchKernelGGL(k_zc_write, dim3((unsigned)n_tiles), dim3(256), 0, s, x, n, offs, reinterpret_cast<long long*>(idx));
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

// Shared tail: read the "empty band" flag back (one 4-byte copy; the callers need the traced freqs on the host anyway).
static int check_empty(int* d_flag, hipStream_t s, const char* who) {
  int h = 0;
  PAR_HIP_CHECK(hipMemcpyAsync(&h, d_flag, sizeof(h), hipMemcpyDeviceToHost, s));
  PAR_HIP_CHECK(hipStreamSynchronize(s));
  PAR_REQUIRE(!(h & 1), PAR_ERR_EMPTY_BAND,
              "%s: a tracking band is empty (frequency below the transform's resolution: the reference's slice "
              "[NL:NU] with NL < 0 is empty and its argmax raises)", who);
  PAR_REQUIRE(!(h & 2), PAR_ERR_INDEX, "%s: the peak sits on the last bin (is_peak() reads fft_frame[peak_i + 1]: IndexError)", who);
  PAR_REQUIRE(!(h & 4), PAR_ERR_SHAPE, "%s: the band reaches past the last bin (operands could not be broadcast together: "
              "np.hanning(NU - NL) against the clipped slice)", who);
  return PAR_OK;
}

int par_track_peak_f64(int device, const float* mag, int64_t n_frames, int bins, int64_t mag_pitch, int64_t frame_0, int64_t count,
                       double* freqs, int fft_size, double sr, double tolerance_oct, int mode, int32_t* status,
                       void* stream) {
  using namespace par;
  const int64_t pitch = mag_pitch ? mag_pitch : bins;
  PAR_REQUIRE(pitch >= bins, PAR_ERR_ARG, "par_track_peak_f64: mag_pitch < bins");
  PAR_REQUIRE(status, PAR_ERR_ARG, "par_track_peak_f64: status word missing");
  PAR_REQUIRE(mag && freqs && bins >= 3 && count >= 0 && frame_0 >= 0 && frame_0 + count <= n_frames, PAR_ERR_ARG,
              "par_track_peak_f64: bad args (frame_0=%lld count=%lld n_frames=%lld)", (long long)frame_0,
              (long long)count, (long long)n_frames);
  PAR_REQUIRE(mode == 0 || mode == 1, PAR_ERR_ARG, "par_track_peak_f64: mode must be 0 or 1");
  if (count == 0) return PAR_OK;
  PAR_HIP_CHECK(hipSetDevice(device));
  hipStream_t s = as_stream(stream);
  PAR_HIP_CHECK(hipMemsetAsync(status, 0, sizeof(int32_t), s));
  if (mode == 0) {
    hipLaunchKernelGGL(k_track_peak, dim3((unsigned)ceil_div(count, 64)), dim3(64), 0, s, mag, bins, pitch, frame_0, count,
                       freqs, fft_size, sr, tolerance_oct, status);
  } else {
    double centre = 0.0;
    PAR_HIP_CHECK(hipMemcpyAsync(&centre, freqs, sizeof(double), hipMemcpyDeviceToHost, s));
    PAR_HIP_CHECK(hipStreamSynchronize(s));
    hipLaunchKernelGGL(k_track_peak_fixed, dim3((unsigned)ceil_div(count, 64)), dim3(64), 0, s, mag, bins, pitch, frame_0, count,
                       centre, freqs, fft_size, sr, tolerance_oct, status);
  }
  PAR_HIP_CHECK(hipGetLastError());
  return check_empty(status, s, "par_track_peak_f64");
}

// PeakTracker / PeakTrackTracker with the band magnitudes recomputed from the signal in float64 (see k_track_peak_refined).
// x: the channel the spectrogram was made from (float32, element stride x_stride), window: the STFT window (float32[n_fft]).
// Bands wider than 2048 bins: PAR_ERR_UNSUPPORTED (the caller falls back to the float32 spectrogram).
int par_track_peak_refined_f64(int device, const float* x, int64_t n, int64_t x_stride, int n_fft, int hop, int zeropad,
                               const float* window, int bins, int64_t n_frames, int64_t frame_0, int64_t count, double* freqs,
                               double sr, double tolerance_oct, int mode, int32_t* status, void* stream) {
  using namespace par;
  PAR_REQUIRE(x && window && freqs && status, PAR_ERR_ARG, "par_track_peak_refined_f64: null pointer");
  PAR_REQUIRE(n >= 2 && x_stride >= 1 && n_fft >= 2 && hop >= 1 && zeropad >= 1 && bins == n_fft * zeropad / 2 + 1 && count >= 0 &&
              frame_0 >= 0 && frame_0 + count <= n_frames && n_frames <= n / hop + 1, PAR_ERR_ARG,
              "par_track_peak_refined_f64: bad args (n=%lld n_fft=%d hop=%d zeropad=%d bins=%d frame_0=%lld count=%lld n_frames=%lld)",
              (long long)n, n_fft, hop, zeropad, bins, (long long)frame_0, (long long)count, (long long)n_frames);
  PAR_REQUIRE(mode == 0 || mode == 1, PAR_ERR_ARG, "par_track_peak_refined_f64: mode must be 0 or 1");
  if (count == 0) return PAR_OK;
  PAR_HIP_CHECK(hipSetDevice(device));
  hipStream_t s = as_stream(stream);
  PAR_HIP_CHECK(hipMemsetAsync(status, 0, sizeof(int32_t), s));
  double centre = 0.0;
  if (mode == 1) {
    PAR_HIP_CHECK(hipMemcpyAsync(&centre, freqs, sizeof(double), hipMemcpyDeviceToHost, s));
    PAR_HIP_CHECK(hipStreamSynchronize(s));
  }
  hipLaunchKernelGGL(k_track_peak_refined, dim3((unsigned)count), dim3(256), kRefineMaxBins * sizeof(double), s, x, n, x_stride,
                     window, n_fft, hop, zeropad, bins, frame_0, count, mode, centre, freqs, sr, tolerance_oct, status);
  PAR_HIP_CHECK(hipGetLastError());
  int h = 0;
  PAR_HIP_CHECK(hipMemcpyAsync(&h, status, sizeof(h), hipMemcpyDeviceToHost, s));
  PAR_HIP_CHECK(hipStreamSynchronize(s));
  PAR_REQUIRE(!(h & 8), PAR_ERR_UNSUPPORTED, "par_track_peak_refined_f64: a tracking band is wider than %d bins", kRefineMaxBins);
  return check_empty(status, s, "par_track_peak_refined_f64");
}

int par_track_cog_f64(int device, const float* mag, int64_t n_frames, int bins, int64_t mag_pitch, int64_t frame_0, int64_t count,
                      double* freqs, int fft_size, double sr, double tolerance_oct, int32_t* status, void* stream) {
  using namespace par;
  const int64_t pitch = mag_pitch ? mag_pitch : bins;
  PAR_REQUIRE(pitch >= bins, PAR_ERR_ARG, "par_track_cog_f64: mag_pitch < bins");
  PAR_REQUIRE(status, PAR_ERR_ARG, "par_track_cog_f64: status word missing");
  PAR_REQUIRE(mag && freqs && bins >= 3 && count >= 0 && frame_0 >= 0 && frame_0 + count <= n_frames, PAR_ERR_ARG,
              "par_track_cog_f64: bad args");
  if (count == 0) return PAR_OK;
  PAR_HIP_CHECK(hipSetDevice(device));
  PAR_HIP_CHECK(hipMemsetAsync(status, 0, sizeof(int32_t), as_stream(stream)));
  hipLaunchKernelGGL(k_track_cog, dim3(1), dim3(64), 0, as_stream(stream), mag, bins, pitch, frame_0, count, freqs, fft_size, sr,
                     tolerance_oct, status);
  PAR_HIP_CHECK(hipGetLastError());
  return check_empty(status, as_stream(stream), "par_track_cog_f64");
}

// CorrelationTracker on the band [NL, NU) of frames 0 .. count-1 (the reference ignores frame_0 here).  M: [n][nb]
// float64 row-major spline matrix, wind: np.hanning(n), both on the device; work: (count + 1) * n + count doubles.
// freqs[count] receives 2^(log_mean + cumulative drift).  status bit 1: a correlation peak sat on the last lag
// (the reference's parabolic() raises IndexError there).
int64_t par_track_corr_work_len(int64_t count, int n) { return (count + 1) * (int64_t)n + count; }

int par_track_corr_f64(int device, const float* mag, int64_t n_frames, int bins, int64_t mag_pitch, int NL, int NU, int64_t count,
                       const double* M, const double* wind, int n, double log_span, double log_mean, double* work,
                       double* freqs, int32_t* status, void* stream) {
  using namespace par;
  PAR_REQUIRE(mag && M && wind && work && freqs && status, PAR_ERR_ARG, "par_track_corr_f64: null pointer");
  // NU may reach past the last bin: the reference's slices [NL:NU] clip to the spectrum while its grid keeps
  // 4 (NU - NL) points (util/wow_detection.py:404-406), so the band has nb = min(NU, bins) - NL values and M is [n][nb]
  const int nb = (NU < bins ? NU : bins) - NL;
  PAR_REQUIRE(NL >= 0 && nb >= 3 && count >= 0 && count <= n_frames && n == 4 * (NU - NL), PAR_ERR_ARG,
              "par_track_corr_f64: bad band / grid (NL=%d NU=%d bins=%d n=%d count=%lld)", NL, NU, bins, n, (long long)count);
  PAR_HIP_CHECK(hipSetDevice(device));
  hipStream_t s = as_stream(stream);
  PAR_HIP_CHECK(hipMemsetAsync(status, 0, sizeof(int32_t), s));
  if (count == 0) return PAR_OK;
  double* R = work;
  double* changes = work + (count + 1) * (int64_t)n;
  const size_t band_lds = (size_t)nb * sizeof(double);
  PAR_REQUIRE(band_lds <= 131072, PAR_ERR_UNSUPPORTED, "par_track_corr_f64: band of %d bins (limit 16384)", nb);
  if (band_lds > 65536) {
    static std::mutex mu;
    static std::set<int> done;
    std::lock_guard<std::mutex> lk(mu);
    if (!done.count(device)) {
      PAR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_corr_resample), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
      done.insert(device);
    }
  }
  hipLaunchKernelGGL(k_corr_resample, dim3((unsigned)(count + 1)), dim3(256), band_lds, s, mag, mag_pitch ? mag_pitch : (int64_t)bins, NL, nb,
                     count, M, wind, n, R);
  const size_t ab_lds = 2 * (size_t)n * sizeof(double);
  if (ab_lds <= 131072) {                                  // n <= 8192: both frames in LDS (up to 128 of the CU's 160 KB)
    if (ab_lds > 65536) {
      static std::mutex mu;
      static std::set<int> done;
      std::lock_guard<std::mutex> lk(mu);
      if (!done.count(device)) {                           // the attribute belongs to the function object of the CURRENT device
        PAR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_corr_peak<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        done.insert(device);
      }
    }
    hipLaunchKernelGGL(k_corr_peak<true>, dim3((unsigned)count), dim3(256), ab_lds, s, (const double*)R, n, changes, status);
  } else {
    hipLaunchKernelGGL(k_corr_peak<false>, dim3((unsigned)count), dim3(256), 0, s, (const double*)R, n, changes, status);
  }
  hipLaunchKernelGGL(k_corr_finish, dim3(1), dim3(1), 0, s, (const double*)changes, count, n, log_span, log_mean, freqs);
  PAR_HIP_CHECK(hipGetLastError());
  int h = 0;
  PAR_HIP_CHECK(hipMemcpyAsync(&h, status, sizeof(h), hipMemcpyDeviceToHost, s));
  PAR_HIP_CHECK(hipStreamSynchronize(s));
  PAR_REQUIRE(!(h & 2), PAR_ERR_INDEX, "par_track_corr_f64: correlation peak on the last lag (index %d is out of bounds for "
              "axis 0 with size %d in the reference's parabolic())", n, n);
  return PAR_OK;
}

// librosa.piptrack on a frame-major magnitude spectrogram (see k_piptrack); pitches / mags: [n_frames][bins] float32.
int par_piptrack_f32(int device, const float* mag, int64_t n_frames, int bins, int64_t mag_pitch, float scale, float offset, int fft_size,
                     double sr, double fmin, double fmax, float threshold, float* pitches, float* mags, void* stream) {
  using namespace par;
  PAR_REQUIRE(mag && pitches && mags, PAR_ERR_ARG, "par_piptrack_f32: null pointer");
  PAR_REQUIRE(n_frames >= 0 && bins >= 2 && fft_size >= 2 && sr > 0, PAR_ERR_ARG, "par_piptrack_f32: bad sizes");
  if (n_frames == 0) return PAR_OK;
  PAR_HIP_CHECK(hipSetDevice(device));
  fmin = fmin > 0.0 ? fmin : 0.0;
  fmax = fmax < sr / 2 ? fmax : sr / 2;
  hipLaunchKernelGGL(k_piptrack, dim3((unsigned)ceil_div(n_frames, 4)), dim3(256), 0, as_stream(stream), mag, bins,
                     mag_pitch ? mag_pitch : (int64_t)bins, n_frames, scale,
                     offset, fft_size, sr, fmin, fmax, threshold, pitches, mags);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

}  // extern "C"
