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
// Everything is float64 with numpy's operation order and no FMA contraction, so the per-segment reciprocal sums are bit-identical to numpy's sequential cumsum.
//
// Round-1 structure: the two cross-segment serial chains (the error-diffused n_i recurrence and the
// float64 offset chain, both inherently order-dependent in floating point) run on the host between
// device stages; the O(len_out) work -- 1 division per output sample, done twice -- is on the GPU.
#include "par_common.h"
#include <math.h>
#include <vector>

// numpy evaluates every product and sum separately: no FMA contraction anywhere in this file, on the
// device and in the host-side chains alike.  build.py compiles THIS file with -ffp-contract=off (hip's
// __dmul_rn/__dadd_rn are plain inline operators parsed before any source-level pragma, so they still
// get fused under the default -ffp-contract=fast -- measured: positions off by 1 ulp).
#pragma clang fp contract(off)

namespace par {

struct PlanHeader {
  int64_t m;
  int64_t len_out;
  int64_t total_written;
  int32_t trimmed;
  int32_t pad;
};

// workspace layout: [PlanHeader | seg_start int64[m] | seg_off f64[m] | tmp f64[m]]
struct PlanView {
  PlanHeader* hdr;
  int64_t* seg_start;
  double* seg_off;
  double* tmp;
};
__host__ __device__ inline size_t plan_bytes(int64_t m) {
  return 64 + (size_t)m * (sizeof(int64_t) + 2 * sizeof(double));
}
inline PlanView plan_view(void* work, int64_t m) {
  char* b = static_cast<char*>(work);
  PlanView v;
  v.hdr = reinterpret_cast<PlanHeader*>(b);
  v.seg_start = reinterpret_cast<int64_t*>(b + 64);
  v.seg_off = reinterpret_cast<double*>(b + 64 + (size_t)m * 8);
  v.tmp = reinterpret_cast<double*>(b + 64 + (size_t)m * 16);
  return v;
}

// a_i (util/resampling.py:103,:111), exact numpy order: diff, (s0+s1)/2, product.
__global__ void k_seg_want(const double* __restrict__ st, const double* __restrict__ sp, int64_t nseg,
                           double* __restrict__ a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nseg) return;
  const double period = st[i + 1] - st[i];
  const double mean = (sp[i] + sp[i + 1]) / 2.0;
  a[i] = period * mean;
}

__device__ __forceinline__ double ramp_recip(long long k, double nm1, double ds, double s0) {
  // 1 / (k/(n-1) * ds + s0), each operation individually rounded (numpy, :120 and :125)
  const double q = (double)k / nm1;
  const double bs = q * ds + s0;      // not fused: this file is compiled with -ffp-contract=off
  return 1.0 / bs;
}

// S_i = last element of np.cumsum(1/block_speeds): strictly sequential float64 adds.
__global__ void k_seg_sum(const double* __restrict__ sp, const int64_t* __restrict__ seg_start, int64_t nseg,
                          double* __restrict__ S) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nseg) return;
  const long long n = seg_start[i + 1] - seg_start[i];
  const double s0 = sp[i], ds = sp[i + 1] - sp[i], nm1 = (double)(n - 1);
  double c = 0.0;
  for (long long k = 0; k < n; ++k) c = c + ramp_recip(k, nm1, ds, s0);
  S[i] = c;
}

// pos[start_i + k] = cumsum_k + offset_i  (:125).  One wave handles 64 consecutive segments and
// transposes 64x64 blocks through LDS so HBM writes are contiguous runs per segment.
constexpr int kFillSegs = 64;
__global__ __launch_bounds__(64) void k_pos_fill(const double* __restrict__ sp, const int64_t* __restrict__ seg_start,
                                                  const double* __restrict__ seg_off, int64_t nseg, int64_t len_out,
                                                  double* __restrict__ pos) {
  __shared__ double buf[kFillSegs][kFillSegs + 1];
  __shared__ long long s_start[kFillSegs];
  __shared__ long long s_n[kFillSegs];
  const int lane = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * kFillSegs + lane;
  long long n = 0, start = 0;
  double s0 = 1.0, ds = 0.0, nm1 = 1.0, off = 0.0;
  if (i < nseg) {
    start = seg_start[i];
    n = seg_start[i + 1] - start;
    if (start >= (long long)len_out) n = 0;
    s0 = sp[i];
    ds = sp[i + 1] - sp[i];
    nm1 = (double)(seg_start[i + 1] - start - 1);
    off = seg_off[i];
  }
  s_start[lane] = start;
  s_n[lane] = n;
  long long nmax = n;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    long long t = __shfl_xor(nmax, o, kWave);
    nmax = t > nmax ? t : nmax;
  }
  double c = 0.0;
  for (long long k0 = 0; k0 < nmax; k0 += kFillSegs) {
    // each lane advances its own segment by up to 64 samples (sequential adds), into LDS row `lane`
    for (int kk = 0; kk < kFillSegs; ++kk) {
      const long long k = k0 + kk;
      if (k < n) {
        c = c + ramp_recip(k, nm1, ds, s0);
        buf[lane][kk] = c + off;
      }
    }
    __syncthreads();
    // write out: for each segment row, lanes cover 64 consecutive output samples
    for (int seg = 0; seg < kFillSegs; ++seg) {
      const long long k = k0 + lane;
      const long long dst = s_start[seg] + k;
      if (k < s_n[seg] && dst < (long long)len_out) pos[dst] = buf[seg][lane];
    }
    __syncthreads();
  }
}

}  // namespace par

extern "C" {

size_t par_speed_plan_bytes(int64_t m) { return par::plan_bytes(m < 2 ? 2 : m); }

int par_speed_to_pos_plan(int device, const double* sampletimes, const double* speeds, int64_t m, int64_t n_in,
                          void* work, size_t work_bytes, int64_t* len_out, int* trimmed, void* stream) {
  using namespace par;
  PAR_REQUIRE(sampletimes && speeds && work && len_out && trimmed, PAR_ERR_ARG, "par_speed_to_pos_plan: null pointer");
  PAR_REQUIRE(m >= 2, PAR_ERR_ARG, "par_speed_to_pos_plan: need at least 2 speed samples (m=%lld)", (long long)m);
  PAR_REQUIRE(work_bytes >= plan_bytes(m), PAR_ERR_WORKSPACE, "par_speed_to_pos_plan: workspace %zu < %zu", work_bytes,
              plan_bytes(m));
  PAR_HIP_CHECK(hipSetDevice(device));
  hipStream_t s = as_stream(stream);
  const int64_t nseg = m - 1;
  PlanView pv = plan_view(work, m);

  // stage 1 (device): a_i
  hipLaunchKernelGGL(k_seg_want, dim3((unsigned)ceil_div(nseg, 256)), dim3(256), 0, s, sampletimes, speeds, nseg, pv.tmp);
  PAR_HIP_CHECK(hipGetLastError());
  std::vector<double> a(nseg), sp(m), st(m);
  PAR_HIP_CHECK(hipMemcpyAsync(a.data(), pv.tmp, nseg * sizeof(double), hipMemcpyDeviceToHost, s));
  PAR_HIP_CHECK(hipMemcpyAsync(sp.data(), speeds, m * sizeof(double), hipMemcpyDeviceToHost, s));
  PAR_HIP_CHECK(hipMemcpyAsync(st.data(), sampletimes, m * sizeof(double), hipMemcpyDeviceToHost, s));
  PAR_HIP_CHECK(hipStreamSynchronize(s));

  // stage 2 (host, serial by construction): error-diffused segment lengths (:113-118)
  std::vector<int64_t> start(m);
  double err = 0.0;
  int64_t acc = 0;
  for (int64_t i = 0; i < nseg; ++i) {
    const double inerr = a[i] + err;
    const double r = nearbyint(inerr);            // round-half-even, like Python round()
    PAR_REQUIRE(r >= 2.0 && r < 9.0e15, PAR_ERR_ARG,
                "par_speed_to_pos_plan: segment %lld has n=%g samples (reference needs n >= 2)", (long long)i, r);
    err = inerr - r;
    start[i] = acc;
    acc += (int64_t)r;
  }
  start[nseg] = acc;
  // end_guess buffer of the reference (:108-109): writing past it raises in numpy
  double mean_speed = 0.0;
  {
    // np.mean = pairwise sum / m; pairwise vs sequential only matters in the last ulp of an int() floor
    // of a value scaled by 1.01 -- restated with numpy's pairwise blocking (blocks of 128, unrolled by 8).
    struct PW {
      static double sum(const double* x, int64_t n) {
        if (n < 8) {
          double r = 0.0;
          for (int64_t i = 0; i < n; ++i) r += x[i];
          return r;
        }
        if (n <= 128) {
          double r[8];
          for (int j = 0; j < 8; ++j) r[j] = x[j];
          int64_t i;
          for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += x[i + j];
          double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
          for (; i < n; ++i) res += x[i];
          return res;
        }
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return sum(x, n2) + sum(x + n2, n - n2);
      }
    };
    mean_speed = PW::sum(sp.data(), m) / (double)m;
  }
  const double guess = mean_speed * (st[m - 1] - st[0]) * 1.01;
  const int64_t cap = (int64_t)guess;

  PAR_HIP_CHECK(hipMemcpyAsync(pv.seg_start, start.data(), m * sizeof(int64_t), hipMemcpyHostToDevice, s));
  // stage 3 (device): per-segment sequential reciprocal sums
  hipLaunchKernelGGL(k_seg_sum, dim3((unsigned)ceil_div(nseg, 64)), dim3(64), 0, s, speeds, pv.seg_start, nseg, pv.tmp);
  PAR_HIP_CHECK(hipGetLastError());
  std::vector<double> S(nseg);
  PAR_HIP_CHECK(hipMemcpyAsync(S.data(), pv.tmp, nseg * sizeof(double), hipMemcpyDeviceToHost, s));
  PAR_HIP_CHECK(hipStreamSynchronize(s));

  // stage 4 (host, serial by construction): offset chain (:125-126) + end trim (:129-135)
  std::vector<double> off(m);
  double offset = st[0];
  int64_t out_len = acc;
  int trim = 0;
  const double N = (double)n_in;
  for (int64_t i = 0; i < nseg; ++i) {
    off[i] = offset;
    PAR_REQUIRE(start[i + 1] <= cap, PAR_ERR_ARG,
                "par_speed_to_pos_plan: positions overflow the reference's end_guess buffer (%lld > %lld); it raises here",
                (long long)start[i + 1], (long long)cap);
    const double first = 1.0 / sp[i] + offset;    // k = 0: bs = 0/(n-1)*ds + s0 = s0
    const double last = S[i] + offset;
    if (first <= N && N <= last) {
      // argmin |pos - N| inside this one segment, first occurrence (np.argmin)
      const int64_t n = start[i + 1] - start[i];
      const double ds = sp[i + 1] - sp[i], nm1 = (double)(n - 1);
      double c = 0.0, best = INFINITY;
      int64_t arg = 0;
      for (int64_t k = 0; k < n; ++k) {
        const double bs = ((double)k / nm1) * ds + sp[i];
        c += 1.0 / bs;
        const double d = fabs((c + offset) - N);
        if (d < best) {
          best = d;
          arg = k;
        }
      }
      out_len = start[i] + arg;
      trim = 1;
      break;
    }
    offset = last;
  }
  off[nseg] = offset;
  PlanHeader h;
  h.m = m;
  h.len_out = out_len;
  h.total_written = acc;
  h.trimmed = trim;
  h.pad = 0;
  PAR_HIP_CHECK(hipMemcpyAsync(pv.seg_off, off.data(), m * sizeof(double), hipMemcpyHostToDevice, s));
  PAR_HIP_CHECK(hipMemcpyAsync(pv.hdr, &h, sizeof(h), hipMemcpyHostToDevice, s));
  PAR_HIP_CHECK(hipStreamSynchronize(s));
  *len_out = out_len;
  *trimmed = trim;
  return PAR_OK;
}

int par_speed_to_pos_fill(int device, const double* speeds, int64_t m, const void* work, double* pos, int64_t len_out,
                          void* stream) {
  using namespace par;
  PAR_REQUIRE(speeds && work && (pos || len_out == 0) && m >= 2, PAR_ERR_ARG, "par_speed_to_pos_fill: bad args");
  if (len_out == 0) return PAR_OK;
  PAR_HIP_CHECK(hipSetDevice(device));
  PlanView pv = plan_view(const_cast<void*>(work), m);
  const int64_t nseg = m - 1;
  hipLaunchKernelGGL(k_pos_fill, dim3((unsigned)ceil_div(nseg, kFillSegs)), dim3(64), 0, as_stream(stream), speeds,
                     pv.seg_start, pv.seg_off, nseg, len_out, pos);
  PAR_HIP_CHECK(hipGetLastError());
  return PAR_OK;
}

}  // extern "C"
