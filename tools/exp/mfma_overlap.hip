// How much plain VALU work hides behind an MFMA on gfx950?  Per group: ONE MFMA (four independent accumulators in rotation)
// + N independent v_fma_f32; 16x16x32 f16 (16 cycles) against 32x32x16 f16 (32 cycles); nominal-clock cycles per group and SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_overlap.hip -o tools/exp/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int BIG, int N>
__global__ __launch_bounds__(64) void k(float* out, float a, float b, int iters) {
  float x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  h8 fa = {(_Float16)x0, 2, 3, 4, 5, 6, 7, 8};
  f4 c0 = {x0, x1, x2, x3}, c1 = c0, c2 = c0, c3 = c0;
  f16v d0, d1, d2, d3;
  for (int i = 0; i < 16; ++i) { d0[i] = x0 + i; d1[i] = x1 + i; d2[i] = x2 + i; d3[i] = x3 + i; }
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (BIG) {
        if ((r & 3) == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0\n" : "+v"(d0) : "v"(fa));
        if ((r & 3) == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0\n" : "+v"(d1) : "v"(fa));
        if ((r & 3) == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0\n" : "+v"(d2) : "v"(fa));
        if ((r & 3) == 3) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0\n" : "+v"(d3) : "v"(fa));
      } else {
        if ((r & 3) == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, %0\n" : "+v"(c0) : "v"(fa));
        if ((r & 3) == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, %0\n" : "+v"(c1) : "v"(fa));
        if ((r & 3) == 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, %0\n" : "+v"(c2) : "v"(fa));
        if ((r & 3) == 3) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, %0\n" : "+v"(c3) : "v"(fa));
      }
#pragma unroll
      for (int q = 0; q < N / 4; ++q)
        asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
      if (N % 4 >= 2) asm volatile("v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %2, %3\n" : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));
    }
  }
  float s = x0 + x1 + x2 + x3 + c0[0] + c1[1] + c2[2] + c3[3];
  for (int i = 0; i < 16; ++i) s += d0[i] + d1[i] + d2[i] + d3[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int BIG, int N>
static void run() {
  float* out;
  (void)hipMalloc(&out, 1 << 24);
  printf("%s + %2d v_fmac per MFMA:", BIG ? "32x32x16" : "16x16x32", N);
  for (int w : {1, 2, 3, 4, 8}) {
    const int blocks = 256 * 4 * w;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<BIG, N><<<blocks, 64>>>(out, 1.0001f, 0.5f, 16);
    (void)hipEventRecord(e0);
    k<BIG, N><<<blocks, 64>>>(out, 1.0001f, 0.5f, ITERS);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  w=%d %6.2f", w, ms * 2.4e6 / ((double)ITERS * 16 * w));
  }
  printf("\n");
  (void)hipFree(out);
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  run<0, 0>(); run<0, 2>(); run<0, 4>(); run<0, 8>(); run<0, 12>(); run<0, 16>();
  run<1, 0>(); run<1, 4>(); run<1, 8>(); run<1, 16>(); run<1, 24>(); run<1, 32>();
  return 0;
}
