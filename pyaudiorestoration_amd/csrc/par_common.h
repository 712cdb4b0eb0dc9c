// Shared host/device helpers for libpar_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include "../../include/par_hip.h"

namespace par {

void set_error(const char* fmt, ...);

#define PAR_HIP_CHECK(call)                                                              \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      par::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return PAR_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

#define PAR_REQUIRE(cond, code, ...)   \
  do {                                 \
    if (!(cond)) {                     \
      par::set_error(__VA_ARGS__);     \
      return (code);                   \
    }                                  \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kMaxDevices = 16;

// wave-level reductions (64 lanes)
__device__ inline long long wave_min_ll(long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    long long t = __shfl_xor(v, o, kWave);
    v = t < v ? t : v;
  }
  return v;
}
__device__ inline long long wave_max_ll(long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    long long t = __shfl_xor(v, o, kWave);
    v = t > v ? t : v;
  }
  return v;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// cross-file launch helpers (pos.hip, sinc.hip) used by the pipelined resampler in varispeed.hip
int launch_pos_fill(const double* speeds, int64_t m, const void* work, double* pos, int64_t len_out, int64_t j_lo,
                    int64_t j_hi, hipStream_t s);
int launch_sinc(int device, const double* pos, int64_t len_out, int64_t j_begin, int64_t count, const float* sig,
                int64_t sig_stride, int64_t len_in, int NT, float* out, int64_t out_stride, hipStream_t s);
int launch_sinc_fused(int device, const double* speeds, int64_t m, const void* work, const void* aux, int64_t max_out,
                      int64_t len_out, const float* sig, const float* sig1, int64_t sig_stride, int64_t len_in, int NT,
                      float* out, float* out1, int64_t out_stride, hipStream_t s, int form = 0);
// several files in one call (include/par_hip.h par_fused_item: the same fields): merged launches where they all take the streaming
// kernel in one form, file by file otherwise
struct FusedBatchItem {
  const double* speeds;
  int64_t m;
  const void* work;
  const void* aux;
  int64_t max_out, len_out;
  const float* sig;
  const float* sig1;
  int64_t sig_stride, len_in;
  float* out;
  float* out1;
  int64_t out_stride;
};
int launch_sinc_fused_batch(int device, int n, const FusedBatchItem* items, int NT, hipStream_t s);
constexpr int64_t kSincTileOutputs = 1024;   // outputs per K_sinc workgroup (chunk boundaries align to it)

}  // namespace par
