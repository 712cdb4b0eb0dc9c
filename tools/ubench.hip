// Instruction-throughput microbenchmarks for gfx950 that price K_sinc's per-tap budget:
// plain f32 FMA, packed f32 FMA, v_rcp_f32, the fma+rcp+fma tap body, v_sin_f32, f64 FMA/div,
// and conflict-free ds_read_b32.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void k_fma(float* out, float a, float b) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], a, b);
  float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_pkfma(float* out, float a, float b) {
  f2 v[8]; f2 A = {a, a}, B = {b, b};
  for (int i = 0; i < 8; ++i) v[i] = (f2){threadIdx.x * 1e-3f + i, 1.0f * i};
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = __builtin_elementwise_fma(v[i], A, B);
  float s = 0; for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_rcp(float* out, float a) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i + a;
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_rcpf(v[i]);
  float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_sin(float* out, float a) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i + a;
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_sinf(v[i]);
  float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the unity-path tap body: x = fma(s,b,a); r = rcp(x); acc = fma(t, r, acc)   (t from a register)
__global__ void k_tap(float* out, float a, float b) {
  float s[4], acc[4], t[4];
  for (int i = 0; i < 4; ++i) { s[i] = threadIdx.x * 1e-3f + i; acc[i] = 0; t[i] = 1.0f + i; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = fmaf(s[i], b + u, a + it);
        acc[i] = fmaf(t[i], __builtin_amdgcn_rcpf(x), acc[i]);
      }
  }
  float r = 0; for (int i = 0; i < 4; ++i) r += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_dfma(double* out, double a, double b) {
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fma(v[i], a, b);
  double s = 0; for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_ddiv(double* out, double a) {
  double v[4];
  for (int i = 0; i < 4; ++i) v[i] = threadIdx.x * 1e-3 + i + a;
  for (int it = 0; it < ITERS / 4; ++it)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a / v[i];
  double s = 0; for (int i = 0; i < 4; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_lds(float* out, int off) {
  __shared__ float buf[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = i;
  __syncthreads();
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int base = threadIdx.x + off;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += buf[(base + it + i * 64) & 4095];
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * 8, threads = 256;
  const double lanes = (double)blocks * threads;
  printf("device %s  CUs %d  clock %.0f MHz  blocks %d x %d\n", p.name, p.multiProcessorCount, p.clockRate / 1e3, blocks, threads);
  void* out; CHECK(hipMalloc(&out, lanes * 8));
  const double simds = p.multiProcessorCount * 4.0, clk = p.clockRate * 1e3;
  auto report = [&](const char* name, double ms, double ops_per_lane) {
    const double lane_ops = lanes * ops_per_lane / (ms * 1e-3);
    const double wave_instr = lane_ops / 64.0;
    printf("%-28s %8.3f ms  %8.2f Tlane-op/s   %.2f cycles per wave-instr per SIMD (at %.0f MHz nominal)\n", name, ms,
           lane_ops / 1e12, simds * clk / wave_instr, p.clockRate / 1e3);
  };
  report("v_fma_f32", time_ms([&] { hipLaunchKernelGGL(k_fma, blocks, threads, 0, 0, (float*)out, 1.0001f, 0.5f); }), ITERS * 8.0);
  report("v_pk_fma_f32 (2 fma/lane)", time_ms([&] { hipLaunchKernelGGL(k_pkfma, blocks, threads, 0, 0, (float*)out, 1.0001f, 0.5f); }), ITERS * 8.0);
  report("v_rcp_f32", time_ms([&] { hipLaunchKernelGGL(k_rcp, blocks, threads, 0, 0, (float*)out, 1.5f); }), ITERS * 8.0);
  report("v_sin_f32", time_ms([&] { hipLaunchKernelGGL(k_sin, blocks, threads, 0, 0, (float*)out, 0.1f); }), ITERS * 8.0);
  report("tap body fma+rcp+fma (per tap)", time_ms([&] { hipLaunchKernelGGL(k_tap, blocks, threads, 0, 0, (float*)out, 3.0f, -1.0f); }), ITERS * 8.0);
  report("v_fma_f64", time_ms([&] { hipLaunchKernelGGL(k_dfma, blocks, threads, 0, 0, (double*)out, 1.0001, 0.5); }), ITERS * 8.0);
  report("f64 divide (IEEE)", time_ms([&] { hipLaunchKernelGGL(k_ddiv, blocks, threads, 0, 0, (double*)out, 1.5); }), ITERS * 1.0);
  report("ds_read_b32 (conflict-free)", time_ms([&] { hipLaunchKernelGGL(k_lds, blocks, threads, 0, 0, (float*)out, 0); }), ITERS * 8.0);
  return 0;
}
