// The Stockham exchange between two radix-8 stages of K_stft (a wave = one 512-point frame, 8 complex points per lane):
// destination lane (b, a), register q receives register a of lane (q, b) -- a rotation of the three 3-bit digits
// (register, lane-high, lane-low).  Two ways to do it:
//   LDS      8 ds_write_b64 to the autosort positions, wave fence, 8 ds_read_b64            (what K_stft does)
//   SHUFFLE  wavefront shuffles only: the source REGISTER depends on the destination lane, so every destination
//            register costs 8 ds_bpermute_b32 per component (one per candidate source register) + selects
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_exchange.hip -o tools/ubench_exchange
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int ITERS = 256;

__global__ __launch_bounds__(256) void k_lds(float2* out) {
  __shared__ float2 X[4][512 + 64 + 8];
  const int f = threadIdx.x >> 6, j = threadIdx.x & 63;
  float2 v[8];
  for (int r = 0; r < 8; ++r) v[r] = make_float2(j + r * 0.5f, r - j * 0.25f);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) { const int e = 8 * j + r; X[f][e + (e >> 3)] = v[r]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int e = j + 64 * q; v[q] = X[f][e + (e >> 3)]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r].x += 1.0f;          // keep the loop honest
  }
  float2 s = make_float2(0, 0);
  for (int r = 0; r < 8; ++r) { s.x += v[r].x; s.y += v[r].y; }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_shfl(float2* out) {
  const int j = threadIdx.x & 63;
  const int a = j & 7, b = j >> 3;
  float2 v[8];
  for (int r = 0; r < 8; ++r) v[r] = make_float2(j + r * 0.5f, r - j * 0.25f);
  for (int it = 0; it < ITERS; ++it) {
    float2 w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int src = (q << 3) | b;                       // lane (q, b)
      float2 got = make_float2(0, 0);
#pragma unroll
      for (int ar = 0; ar < 8; ++ar) {                    // the wanted register is a = j & 7: try all, keep one
        const float x = __shfl(v[ar].x, src, 64), y = __shfl(v[ar].y, src, 64);
        if (a == ar) got = make_float2(x, y);
      }
      w[q] = got;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) { v[r] = w[r]; v[r].x += 1.0f; }
  }
  float2 s = make_float2(0, 0);
  for (int r = 0; r < 8; ++r) { s.x += v[r].x; s.y += v[r].y; }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
  const int blocks = p.multiProcessorCount * 8;
  float2* out; (void)hipMalloc(&out, (size_t)blocks * 256 * sizeof(float2));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](const char* name, void (*k)(float2*)) {
    hipLaunchKernelGGL(k, blocks, 256, 0, 0, out); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      (void)hipEventRecord(e0); hipLaunchKernelGGL(k, blocks, 256, 0, 0, out); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double exch = (double)blocks * 4 * ITERS;       // frame exchanges
    printf("%-10s %8.3f ms  %7.1f ns per 512-point exchange per CU-wave slot, %6.2f G exchanges/s\n", name, best,
           best * 1e6 / (ITERS * 8.0), exch / (best * 1e-3) / 1e9);
  };
  printf("device %s, %d CUs, %d blocks x 256 (8 waves/SIMD)\n", p.name, p.multiProcessorCount, blocks);
  run("LDS", k_lds);
  run("SHUFFLE", k_shfl);
  return 0;
}
