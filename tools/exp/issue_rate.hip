// How fast can ONE wave issue, and what do a few instruction forms cost?  gfx950, 1 .. 8 waves per SIMD, long loops (the
// launch is a small part of the time).  Prints nominal-clock (2.4 GHz) SIMD cycles per instruction = time x 2.4 GHz / (instructions
// per wave x waves per SIMD): the throughput figure; x waves per SIMD = what one wave sees between two of its own instructions.
// Build: hipcc --offload-arch=gfx950 -O3 tools/exp/issue_rate.hip -o tools/exp/issue_rate ;  run: issue_rate <kind>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITERS 16384
#define R8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, float a, float b, int iters) {
  __shared__ float lds[512];
  lds[threadIdx.x] = a;
  lds[threadIdx.x + 64] = b;
  float x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  unsigned long long m = __ballot(threadIdx.x & 1);
  const unsigned la = threadIdx.x * 4;
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      asm volatile(R8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                      "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (KIND == 1) {
      asm volatile(R8("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                      "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n")
                   : "+v"(x0) : "v"(a), "v"(b));
    } else if (KIND == 2) {   // 4 fma + 4 salu
      asm volatile(R8("v_fma_f32 %0, %0, %4, %5\n s_add_u32 s20, s20, 1\n v_fma_f32 %1, %1, %4, %5\n s_add_u32 s21, s21, 1\n"
                      "v_fma_f32 %2, %2, %4, %5\n s_add_u32 s20, s20, 3\n v_fma_f32 %3, %3, %4, %5\n s_add_u32 s21, s21, 3\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "s20", "s21", "scc");
    } else if (KIND == 3) {   // 6 fma + 2 ds_read_b32 waited at the end of the group of 8
      asm volatile(R8("ds_read_b32 %6, %8\n v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %1, %1, %9, %10\n v_fma_f32 %2, %2, %9, %10\n"
                      "ds_read_b32 %7, %8 offset:256\n v_fma_f32 %3, %3, %9, %10\n v_fma_f32 %4, %4, %9, %10\n v_fma_f32 %5, %5, %9, %10\n s_waitcnt lgkmcnt(0)\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "=&v"(x6), "=&v"(x7) : "v"(la), "v"(a), "v"(b));
    } else if (KIND == 4) {   // v_cndmask e32 (VCC), VCC never written in the loop
      asm volatile(R8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                      "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    } else if (KIND == 5) {   // v_cndmask e64 with an SGPR pair
      asm volatile(R8("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                      "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "s"(m));
    } else if (KIND == 6) {   // v_cmp (writes VCC) + v_cndmask (reads it): the pair the compiler emits
      asm volatile(R8("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc\n"
                      "v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %4, vcc\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "vcc");
    } else if (KIND == 7) {   // v_cmp into an SGPR pair + v_cndmask from it
      asm volatile(R8("v_cmp_lt_f32 s[20:21], %0, %4\n v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cmp_lt_f32 s[22:23], %1, %4\n v_cndmask_b32 %1, %1, %4, s[22:23]\n"
                      "v_cmp_lt_f32 s[20:21], %2, %4\n v_cndmask_b32 %2, %2, %4, s[20:21]\n v_cmp_lt_f32 s[22:23], %3, %4\n v_cndmask_b32 %3, %3, %4, s[22:23]\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "s20", "s21", "s22", "s23");
    } else if (KIND == 8) {   // v_readfirstlane + s_and + v_add using it (VALU -> SALU -> VALU round trip), 2 chains
      asm volatile(R8("v_readfirstlane_b32 s20, %0\n s_and_b32 s20, s20, 0xff\n v_add_u32 %0, %0, s20\n v_fma_f32 %2, %2, %4, %5\n"
                      "v_readfirstlane_b32 s21, %1\n s_and_b32 s21, s21, 0xff\n v_add_u32 %1, %1, s21\n v_fma_f32 %3, %3, %4, %5\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "s20", "s21", "scc");
    } else if (KIND == 10) {  // one v_cmp -> VCC, three v_cndmask reading it
      asm volatile(R8("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n"
                      "v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "vcc");
    } else if (KIND == 11) {  // one v_cmp -> SGPR pair, three v_cndmask reading it
      asm volatile(R8("v_cmp_lt_f32 s[20:21], %0, %4\n v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %4, s[20:21]\n"
                      "v_cmp_lt_f32 s[22:23], %3, %4\n v_cndmask_b32 %3, %3, %4, s[22:23]\n v_cndmask_b32 %1, %1, %4, s[22:23]\n v_cndmask_b32 %2, %2, %4, s[22:23]\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "s20", "s21", "s22", "s23");
    } else if (KIND == 12) {  // v_cndmask VCC with fmas in between (does distance help?)
      asm volatile(R8("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_fma_f32 %3, %3, %4, %5\n v_cndmask_b32 %1, %1, %4, vcc\n"
                      "v_fma_f32 %3, %3, %4, %5\n v_cndmask_b32 %2, %2, %4, vcc\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc");
    } else if (KIND == 13) {  // s_mov vcc from an SGPR pair, then v_cndmask reading VCC three times
      asm volatile(R8("s_mov_b64 vcc, %5\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n"
                      "s_mov_b64 vcc, %5\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "s"(m) : "vcc");
    } else if (KIND == 14 || KIND == 15 || KIND == 16) {   // per group: 1 MFMA (4 independent accumulators in rotation) + 0 / 4 / 8 independent v_fma
      typedef _Float16 h8 __attribute__((ext_vector_type(8)));
      typedef float f4 __attribute__((ext_vector_type(4)));
      static_assert(true, "");
      h8 fa = {(_Float16)x4, 2, 3, 4, 5, 6, 7, 8};
      f4 c0 = {x0, x1, x2, x3}, c1 = {x1, x0, x2, x3}, c2 = {x2, x1, x0, x3}, c3 = {x3, x1, x2, x0};
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %4, %0\n" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(fa));
        if (KIND >= 15) asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n" : "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_16x16x32_f16 %1, %4, %4, %1\n" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(fa));
        if (KIND >= 16) asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n" : "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_16x16x32_f16 %2, %4, %4, %2\n" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(fa));
        if (KIND >= 15) asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n" : "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_16x16x32_f16 %3, %4, %4, %3\n" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(fa));
        if (KIND >= 16) asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n" : "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      }
      x0 += c0[0] + c1[1] + c2[2] + c3[3];
    } else if (KIND == 9) {   // 8 independent MFMAs 16x16x32 f16
      typedef _Float16 h8 __attribute__((ext_vector_type(8)));
      typedef float f4 __attribute__((ext_vector_type(4)));
      h8 fa = {1, 2, 3, 4, 5, 6, 7, 8}; f4 acc[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = f4{x0, x1, x2, x3};
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fa, acc[q], 0, 0, 0);
      x0 += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + acc[4][0] + acc[5][0] + acc[6][0] + acc[7][0];
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + lds[(threadIdx.x + 1) & 63];
}
template <int KIND>
static void run(const char* name) {
  float* out;
  hipMalloc(&out, 1 << 24);
  printf("%-52s", name);
  for (int w : {1, 2, 3, 4, 6, 8}) {
    const int blocks = 256 * 4 * w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 64>>>(out, 1.0001f, 0.5f, 64);
    hipEventRecord(e0);
    k<KIND><<<blocks, 64>>>(out, 1.0001f, 0.5f, ITERS);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  w=%d %5.2f", w, ms * 2.4e6 / ((double)ITERS * 64 * w));
  }
  printf("\n");
}
int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int kind = argc > 1 ? atoi(argv[1]) : -1;
  if (kind == 0) run<0>("8 independent v_fma_f32");
  if (kind == 1) run<1>("dependent v_fma_f32 chain");
  if (kind == 2) run<2>("v_fma / s_add alternating (per instruction)");
  if (kind == 3) run<3>("6 v_fma + 2 ds_read_b32, waited per group of 8 (+1 waitcnt)");
  if (kind == 4) run<4>("v_cndmask_b32 e32 (VCC, never written)");
  if (kind == 5) run<5>("v_cndmask_b32 e64 (SGPR pair)");
  if (kind == 6) run<6>("v_cmp -> VCC -> v_cndmask pairs (per instruction)");
  if (kind == 7) run<7>("v_cmp -> SGPR pair -> v_cndmask pairs (per instruction)");
  if (kind == 8) run<8>("readfirstlane / s_and / v_add(s) / v_fma (per instruction)");
  if (kind == 10) run<10>("1 v_cmp -> VCC, 3 v_cndmask (per instruction)");
  if (kind == 11) run<11>("1 v_cmp -> SGPR pair, 3 v_cndmask (per instruction)");
  if (kind == 12) run<12>("v_cmp -> VCC, 3 v_cndmask with v_fma between (per instr)");
  if (kind == 13) run<13>("s_mov vcc, 3 v_cndmask (per instruction)");
  if (kind == 14) run<14>("64 MFMA 16x16x32 f16 per iteration, nothing else (per 64)");
  if (kind == 15) run<15>("64 MFMA + 128 v_fma interleaved (per 64)");
  if (kind == 16) run<16>("64 MFMA + 256 v_fma interleaved (per 64)");
  if (kind == 9) run<9>("v_mfma_f32_16x16x32_f16, 8 accumulators (per MFMA)");
  return 0;
}
