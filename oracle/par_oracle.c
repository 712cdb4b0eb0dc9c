/*
 * par_oracle.c -- plain-C CPU restatement of the reference's varispeed hot path.
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP path at sizes numpy would be too slow for, and
 * the `cpu_baseline` ("kind": "port") of bench.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; nothing under pyaudiorestoration_amd/ does.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks every function here against golden
 * vectors captured from the real reference (oracle/gen_golden.py).
 *
 * Restates (reference file:line, HENDRIX-ZT2/pyaudiorestoration):
 *   oracle_speed_to_pos   util/resampling.py:93-137   speed_to_pos
 *   oracle_sinc           util/resampling.py:21-27, 51-90  sinc_wrapper -> sinc_core (float64 math like numba)
 *   oracle_sinc_mt        util/resampling.py:30-46    sinc_wrapper_mt: one contiguous chunk per thread
 *   oracle_stft(_mt)      util/fourier.py:23-29, 37-82, 136-166  stft / get_mag via the numpy framing (frames over threads)
 *   oracle_synth_*        SURVEY 8d closed-form workload (same as tests/inputs.py)
 * Build: gcc -O2 -fPIC -shared -pthread par_oracle.c -lm   (no -ffast-math: numpy order is kept)
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ speed_to_pos */
/* returns 0 ok; *len_out = written prefix (or trimmed length), *trimmed = 1 when the end trim fired.
 * out must hold at least `cap` doubles; positions beyond cap -> -2 (the reference raises there). */
int oracle_speed_to_pos(const double* st, const double* sp, int64_t m, int64_t n_in, double* out, int64_t cap,
                        int64_t* len_out, int* trimmed) {
  double err = 0.0, offset = st[0];
  int64_t w = 0;
  *trimmed = 0;
  for (int64_t i = 0; i + 1 < m; ++i) {
    const double period = st[i + 1] - st[i];
    const double mean = (sp[i] + sp[i + 1]) / 2.0;
    const double inerr = period * mean + err;
    const double r = nearbyint(inerr);                /* Python round(): half to even */
    if (r < 2.0) return -1;
    const int64_t n = (int64_t)r;
    err = inerr - r;
    if (w + n > cap) return -2;
    const double ds = sp[i + 1] - sp[i], nm1 = (double)(n - 1);
    double c = 0.0;
    for (int64_t k = 0; k < n; ++k) {
      const double bs = ((double)k / nm1) * ds + sp[i];
      c += 1.0 / bs;
      out[w + k] = c + offset;
    }
    offset = out[w + n - 1];
    if (out[w] <= (double)n_in && (double)n_in <= out[w + n - 1]) {
      int64_t arg = 0;
      double best = INFINITY;
      for (int64_t k = 0; k < n; ++k) {
        const double d = fabs(out[w + k] - (double)n_in);
        if (d < best) {
          best = d;
          arg = k;
        }
      }
      *len_out = w + arg;
      *trimmed = 1;
      return 0;
    }
    w += n;
  }
  *len_out = w;
  return 0;
}

/* numpy's float64 pairwise summation (np.add.reduce on a contiguous array; numpy/_core/src/umath/loops_utils.h.src,
 * pairwise_sum): < 8 elements sequential, <= 128 eight interleaved accumulators combined as a balanced tree plus a
 * sequential tail, otherwise split at floor(n/2) rounded down to a multiple of 8.  np.mean = this sum / n. */
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

/* int(mean(speeds) * (st[-1]-st[0]) * 1.01): the reference's buffer size (:108-109), with numpy's own summation
 * order so that the overflow test (`output[out_ind:out_ind+n] = sample_at` raising) fires on exactly the same curves */
int64_t oracle_end_guess(const double* st, const double* sp, int64_t m) {
  return (int64_t)((np_pairwise_sum(sp, m) / (double)m) * (st[m - 1] - st[0]) * 1.01);
}

/* ------------------------------------------------------------------ speed_to_pos, windows of it
 * The same positions as oracle_speed_to_pos (util/resampling.py:93-137) for the outputs [starts[w], starts[w] + width), w < count,
 * without the whole array (5.5 GB for the benchmark's file) and with the one parallel piece run on threads -- the SAME floating-point
 * operations in the SAME order per value:
 *   pass 1  the rounding chain of the segment lengths n_i (serial, O(m)): inerr = period * mean + err, n = round(inerr), err = inerr - n;
 *   pass 2  every segment's last cumsum value: c runs from 0.0 inside a segment, independent of every other segment (threads);
 *   pass 3  the offset chain  offset_{i+1} = c_last_i + offset_i  and the end trim (serial, O(m); the trim segment is walked);
 *   pass 4  the windows: a segment's cumsum from its first step, + offset.
 * out: [count][width] doubles; outputs at or beyond *len_out are NaN.  Test infrastructure like the rest of this file; pinned to
 * oracle_speed_to_pos bit for bit by tests/test_oracle_golden.py. */
typedef struct {
  const double* sp;
  const int64_t* n;
  double* clast;
  double* cfirst;
  int64_t a, b;
} SegSumJob;

static void* seg_sum_worker(void* v) {
  SegSumJob* j = (SegSumJob*)v;
  for (int64_t i = j->a; i < j->b; ++i) {
    const int64_t n = j->n[i];
    const double ds = j->sp[i + 1] - j->sp[i], nm1 = (double)(n - 1);
    double c = 0.0;
    for (int64_t k = 0; k < n; ++k) {
      const double bs = ((double)k / nm1) * ds + j->sp[i];
      c += 1.0 / bs;
      if (k == 0) j->cfirst[i] = c;
    }
    j->clast[i] = c;
  }
  return NULL;
}

int oracle_speed_to_pos_windows(const double* st, const double* sp, int64_t m, int64_t n_in, const int64_t* starts, int count,
                                int64_t width, double* out, int64_t* len_out, int* trimmed, int threads) {
  const int64_t nseg = m - 1, cap = oracle_end_guess(st, sp, m);
  int64_t* n = (int64_t*)malloc((size_t)(nseg + 1) * sizeof(int64_t));
  int64_t* w0 = (int64_t*)malloc((size_t)(nseg + 1) * sizeof(int64_t));
  double* clast = (double*)malloc((size_t)(nseg + 1) * sizeof(double));
  double* cfirst = (double*)malloc((size_t)(nseg + 1) * sizeof(double));
  double* off = (double*)malloc((size_t)(nseg + 1) * sizeof(double));
  if (!n || !w0 || !clast || !cfirst || !off) return -3;
  int rc = 0;
  double err = 0.0;
  int64_t used = nseg;                                   /* segments the reference visits (it stops where the cap is hit) */
  for (int64_t i = 0; i < nseg; ++i) {                   /* pass 1 */
    const double period = st[i + 1] - st[i];
    const double mean = (sp[i] + sp[i + 1]) / 2.0;
    const double inerr = period * mean + err;
    const double r = nearbyint(inerr);
    if (r < 2.0) { rc = -1; used = i; break; }
    n[i] = (int64_t)r;
    err = inerr - r;
  }
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  {                                                      /* pass 2 */
    pthread_t th[256];
    SegSumJob jobs[256];
    const int64_t per = (used + threads - 1) / threads;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
      const int64_t a = t * per, b = a + per < used ? a + per : used;
      if (a >= b) break;
      jobs[t] = (SegSumJob){sp, n, clast, cfirst, a, b};
      if (pthread_create(&th[t], NULL, seg_sum_worker, &jobs[t]) != 0) { seg_sum_worker(&jobs[t]); th[t] = 0; }
      ++started;
    }
    for (int t = 0; t < started; ++t) if (th[t]) pthread_join(th[t], NULL);
  }
  double offset = st[0];                                 /* pass 3 */
  int64_t w = 0;
  *trimmed = 0;
  *len_out = -1;
  int64_t visited = 0;
  for (int64_t i = 0; i < used; ++i) {
    if (w + n[i] > cap) { rc = -2; break; }
    w0[i] = w;
    off[i] = offset;
    visited = i + 1;
    const double first = cfirst[i] + offset, last = clast[i] + offset;
    if (first <= (double)n_in && (double)n_in <= last) {
      const double ds = sp[i + 1] - sp[i], nm1 = (double)(n[i] - 1);
      double c = 0.0, best = INFINITY;
      int64_t arg = 0;
      for (int64_t k = 0; k < n[i]; ++k) {
        const double bs = ((double)k / nm1) * ds + sp[i];
        c += 1.0 / bs;
        const double d = fabs((c + offset) - (double)n_in);
        if (d < best) { best = d; arg = k; }
      }
      *len_out = w + arg;
      *trimmed = 1;
      break;
    }
    offset = last;
    w += n[i];
  }
  if (*len_out < 0 && rc == 0) *len_out = w;
  if (rc == 0 || *len_out >= 0) {                        /* pass 4 */
    const int64_t total = *len_out;
    for (int c_ = 0; c_ < count; ++c_) {
      for (int64_t q = 0; q < width; ++q) out[(int64_t)c_ * width + q] = NAN;
      int64_t j = starts[c_];
      const int64_t j_end = j + width < total ? j + width : total;
      if (j < 0 || j >= j_end) continue;
      int64_t lo = 0, hi = visited - 1;                  /* segment of output j: the last one with w0 <= j */
      while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (w0[mid] <= j) lo = mid; else hi = mid - 1;
      }
      for (int64_t i = lo; i < visited && j < j_end; ++i) {
        const double ds = sp[i + 1] - sp[i], nm1 = (double)(n[i] - 1);
        double c = 0.0;
        for (int64_t k = 0; k < n[i] && w0[i] + k < j_end; ++k) {
          const double bs = ((double)k / nm1) * ds + sp[i];
          c += 1.0 / bs;
          if (w0[i] + k >= j) {
            out[(int64_t)c_ * width + (w0[i] + k - starts[c_])] = c + off[i];
            j = w0[i] + k + 1;
          }
        }
      }
    }
    if (*len_out >= 0) rc = 0;
  }
  free(n); free(w0); free(clast); free(cfirst); free(off);
  return rc;
}

/* ------------------------------------------------------------------ sinc_core */
static float hann32(int k, int NT) { return (float)(0.5 + 0.5 * cos(M_PI * (double)(k - NT) / (double)NT)); }

static void sinc_range(const double* pos, int64_t len_out, int64_t a, int64_t b, const float* sig, int64_t sig_stride,
                       int64_t len_in, int NT, const float* win, float* out, int64_t out_stride) {
  for (int64_t i = a; i < b; ++i) {
    const double p = pos[i];
    /* Python's int(round(p)) is an arbitrary-precision integer: a position beyond the int64 range selects an empty
     * slice of the signal (lower past the end, or upper far below 0) and the sum over it is 0.0.  Non-finite
     * positions make the reference raise; oracle_sinc refuses them before getting here. */
    if (!(fabs(p) < 9.0e18)) {
      out[i * out_stride] = 0.0f;
      continue;
    }
    const int64_t ind = (int64_t)nearbyint(p);
    const int64_t lower = ind - NT > 0 ? ind - NT : 0;
    const int64_t upper = ind + NT < len_in ? ind + NT : len_in;
    /* canonical period: true next position for all but the global last sample */
    double dp = (i + 1 < len_out) ? pos[i + 1] - p : p - pos[i - 1];
    if (!(dp > 1e-12)) dp = 1e-12;
    double fc = 1.0 / dp;
    if (fc > 1.0) fc = 1.0;
    const double shift = p - (double)ind;
    double acc = 0.0;
    for (int64_t k = 0; k < upper - lower; ++k) {
      double x = ((double)(k - NT) - shift) * fc;
      double y = M_PI * (x == 0.0 ? 1e-20 : x);
      acc += (double)sig[(lower + k) * sig_stride] * (sin(y) / y * fc) * (double)win[k];
    }
    out[i * out_stride] = (float)acc;
  }
}

int oracle_sinc(const double* pos, int64_t len_out, const float* sig, int64_t sig_stride, int64_t len_in, int NT,
                float* out, int64_t out_stride) {
  if (len_out < 2 || NT < 1) return -1;
  for (int64_t i = 0; i < len_out; ++i)
    if (!isfinite(pos[i])) return -3;               /* int(round(inf / nan)) raises in the reference */
  float* win = (float*)malloc(sizeof(float) * (2 * NT + 1));
  for (int k = 0; k <= 2 * NT; ++k) win[k] = hann32(k, NT);
  sinc_range(pos, len_out, 0, len_out, sig, sig_stride, len_in, NT, win, out, out_stride);
  free(win);
  return 0;
}

typedef struct {
  const double* pos;
  int64_t len_out, a, b;
  const float* sig;
  int64_t sig_stride, len_in;
  int NT;
  const float* win;
  float* out;
  int64_t out_stride;
} sinc_job;

static void* sinc_worker(void* v) {
  sinc_job* j = (sinc_job*)v;
  sinc_range(j->pos, j->len_out, j->a, j->b, j->sig, j->sig_stride, j->len_in, j->NT, j->win, j->out, j->out_stride);
  return NULL;
}

/* contiguous chunks of ceil(len/nthreads), exactly like sinc_wrapper_mt (:33-46) */
int oracle_sinc_mt(const double* pos, int64_t len_out, const float* sig, int64_t sig_stride, int64_t len_in, int NT,
                   float* out, int64_t out_stride, int nthreads) {
  if (len_out < 2 || NT < 1 || nthreads < 1) return -1;
  float* win = (float*)malloc(sizeof(float) * (2 * NT + 1));
  for (int k = 0; k <= 2 * NT; ++k) win[k] = hann32(k, NT);
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
  sinc_job* jobs = (sinc_job*)malloc(sizeof(sinc_job) * nthreads);
  const int64_t chunk = (len_out + nthreads - 1) / nthreads;
  int started = 0;
  for (int t = 0; t < nthreads; ++t) {
    int64_t a = t * chunk, b = a + chunk < len_out ? a + chunk : len_out;
    if (a >= b) break;
    sinc_job jb = {pos, len_out, a, b, sig, sig_stride, len_in, NT, win, out, out_stride};
    jobs[t] = jb;
    pthread_create(&th[t], NULL, sinc_worker, &jobs[t]);
    ++started;
  }
  for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
  free(th);
  free(jobs);
  free(win);
  return 0;
}

/* ------------------------------------------------------------------ STFT magnitude */
static int64_t reflect_idx(int64_t q, int64_t n) {
  if (n == 1) return 0;
  const int64_t P = 2 * (n - 1);
  q %= P;
  if (q < 0) q += P;
  return q < n ? q : P - q;
}

/* iterative radix-2 complex FFT, float64 */
static void fft_c(double* re, double* im, int n) {
  for (int i = 1, j = 0; i < n; ++i) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      double t = re[i]; re[i] = re[j]; re[j] = t;
      t = im[i]; im[i] = im[j]; im[j] = t;
    }
  }
  for (int len = 2; len <= n; len <<= 1) {
    const double ang = -2.0 * M_PI / len;
    for (int i = 0; i < n; i += len) {
      for (int k = 0; k < len / 2; ++k) {
        const double wr = cos(ang * k), wi = sin(ang * k);
        const int a = i + k, b = i + k + len / 2;
        const double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
        re[b] = re[a] - xr; im[b] = im[a] - xi;
        re[a] += xr; im[a] += xi;
      }
    }
  }
}

/* out: frame-major [frames][bins]; mode 0 complex interleaved (re,im) float, mode 1 |X|+1e-7 float */
static int stft_frames(const float* x, int64_t n, int64_t x_stride, int n_fft, int hop, int zeropad, const float* window,
                       float* out, int mode, int64_t f_lo, int64_t f_hi) {
  const int M = n_fft * zeropad;
  const int bins = M / 2 + 1;
  double* re = (double*)malloc(sizeof(double) * M);
  double* im = (double*)malloc(sizeof(double) * M);
  const double scale = 1.0 / sqrt((double)n_fft);
  for (int64_t f = f_lo; f < f_hi; ++f) {
    for (int t = 0; t < M; ++t) {
      im[t] = 0.0;
      re[t] = t < n_fft ? (double)(window[t] * x[reflect_idx(f * hop - n_fft / 2 + t, n) * x_stride]) : 0.0;
    }
    fft_c(re, im, M);
    for (int k = 0; k < bins; ++k) {
      const double a = re[k] * scale, b = im[k] * scale;
      if (mode == 0) {
        out[2 * (f * bins + k)] = (float)a;
        out[2 * (f * bins + k) + 1] = (float)b;
      } else {
        out[f * bins + k] = (float)(sqrt(a * a + b * b) + 1e-7);
      }
    }
  }
  free(re);
  free(im);
  return 0;
}

int oracle_stft(const float* x, int64_t n, int64_t x_stride, int n_fft, int hop, int zeropad, const float* window,
                float* out, int mode) {
  const int M = n_fft * zeropad;
  if (M < 2 || (M & (M - 1))) return -1;
  const int64_t frames = (n + 2 * (int64_t)(n_fft / 2) - n_fft) / hop + 1;
  return stft_frames(x, n, x_stride, n_fft, hop, zeropad, window, out, mode, 0, frames);
}

/* the same transform with the frames split over `threads` host threads (bench.py's CPU baseline of the STFT) */
typedef struct {
  const float* x;
  int64_t n, x_stride;
  int n_fft, hop, zeropad;
  const float* window;
  float* out;
  int mode;
  int64_t f_lo, f_hi;
} stft_job;
static void* stft_worker(void* p) {
  stft_job* j = (stft_job*)p;
  stft_frames(j->x, j->n, j->x_stride, j->n_fft, j->hop, j->zeropad, j->window, j->out, j->mode, j->f_lo, j->f_hi);
  return NULL;
}
int oracle_stft_mt(const float* x, int64_t n, int64_t x_stride, int n_fft, int hop, int zeropad, const float* window,
                   float* out, int mode, int threads) {
  const int M = n_fft * zeropad;
  if (M < 2 || (M & (M - 1))) return -1;
  const int64_t frames = (n + 2 * (int64_t)(n_fft / 2) - n_fft) / hop + 1;
  if (threads < 1) threads = 1;
  if (threads > 1024) threads = 1024;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  stft_job* jobs = (stft_job*)malloc(sizeof(stft_job) * threads);
  const int64_t per = (frames + threads - 1) / threads;
  int started = 0;
  for (int t = 0; t < threads; ++t) {
    const int64_t lo = t * per, hi = lo + per < frames ? lo + per : frames;
    if (lo >= hi) break;
    stft_job j = {x, n, x_stride, n_fft, hop, zeropad, window, out, mode, lo, hi};
    jobs[t] = j;
    pthread_create(&th[t], NULL, stft_worker, &jobs[t]);
    ++started;
  }
  for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
  free(th);
  free(jobs);
  return 0;
}

/* ------------------------------------------------------------------ synthetic workload */
static double splitmix_uniform(uint64_t idx, uint64_t seed) {
  uint64_t z = (idx ^ seed) + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

void oracle_synth_signal(float* out, int64_t start, int64_t count, double sr, uint64_t seed) {
  const double w1 = 2.0 * M_PI * 1000.0, w2 = 2.0 * M_PI * (0.45 * sr / 2.0);
  for (int64_t i = 0; i < count; ++i) {
    const double n = (double)(start + i);
    out[i] = (float)(0.25 * sin(w1 * n / sr) + 0.25 * sin(w2 * n / sr) + 0.1 * splitmix_uniform((uint64_t)(start + i), seed));
  }
}

void oracle_synth_curve(double* st, double* sp, int64_t m, double dur, double sr, double depth, double rate, double phase) {
  const double step = dur / (double)(m - 1);
  for (int64_t i = 0; i < m; ++i) {
    const double t = (i == m - 1) ? dur : (double)i * step;
    st[i] = t * sr;
    sp[i] = 1.0 + depth * sin(2.0 * M_PI * rate * t + phase);
  }
}
