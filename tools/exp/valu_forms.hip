// What do the VALU instruction FORMS of K_sinc's loops cost on gfx950?  8 independent destinations per group, 8 groups per
// iteration; prints nominal-clock (2.4 GHz) SIMD cycles per instruction at 1 .. 8 waves per SIMD (the throughput figure).
// Build: hipcc --offload-arch=gfx950 -O3 tools/exp/valu_forms.hip -o tools/exp/valu_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITERS 8192
#define R8(x) x x x x x x x x
#define G8(op) op(%0) op(%1) op(%2) op(%3) op(%4) op(%5) op(%6) op(%7)
#define OUTS "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
#define F0(d) "v_fma_f32 " #d ", " #d ", %8, %9\n"
#define F1(d) "v_fmac_f32 " #d ", %8, %9\n"
#define F2(d) "v_fmaak_f32 " #d ", " #d ", %8, 0x3fc90fdb\n"
#define F3(d) "v_fmamk_f32 " #d ", " #d ", 0x3fc90fdb, %8\n"
#define F4(d) "v_mul_f32 " #d ", 0x3fc90fdb, " #d "\n"
#define F5(d) "v_mul_f32 " #d ", %8, " #d "\n"
#define F6(d) "v_fma_f32 " #d ", " #d ", %10, %9\n"
#define F7(d) "v_fma_f32 " #d ", " #d ", 2.0, %9\n"
#define F8(d) "v_add_u32 " #d ", %8, " #d "\n"
#define F9(d) "v_lshl_or_b32 " #d ", " #d ", 4, %8\n"
#define F10(d) "v_cvt_f32_f16 " #d ", " #d "\n"
#define F11(d) "v_cvt_pk_f16_f32 " #d ", " #d ", %8\n"
#define F12(d) "v_rndne_f32 " #d ", " #d "\n"
#define F13(d) "v_med3_f32 " #d ", " #d ", 0, %8\n"
#define F14(d) "v_add3_u32 " #d ", " #d ", %8, %9\n"
#define F15(d) "v_rcp_f32 " #d ", " #d "\n"
#define F16(d) "v_cos_f32 " #d ", " #d "\n"
#define F17(d) "v_fma_f32 " #d ", -" #d ", |%8|, %9\n"
#define F18(d) "v_mul_f32 " #d ", %10, " #d "\n"
#define F19(d) "v_add_f32 " #d ", %8, " #d "\n"
#define F20(d) "v_sub_f32 " #d ", " #d ", %8\n"
#define F21(d) "v_bfe_u32 " #d ", " #d ", 10, 6\n"
#define F22(d) "v_and_b32 " #d ", 31, " #d "\n"
#define F23(d) "v_cvt_f32_i32 " #d ", " #d "\n"
#define F24(d) "v_cvt_i32_f32 " #d ", " #d "\n"
#define F25(d) "v_pk_fma_f32 " #d ", " #d ", %8, %9\n"
#define F26(d) "v_cmp_gt_f32 s[20:21], " #d ", %8\n"
#define F27(d) "v_fma_f32 " #d ", " #d ", %8, 0.5\n"
#define F28(d) "v_fmac_f32 " #d ", 0x3fc90fdb, %9\n"
#define F29(d) "v_mov_b32 " #d ", %8\n"
#define CASE(K, F) else if (KIND == K) asm volatile(R8(G8(F)) : OUTS : "v"(a), "v"(b), "s"(sa) : "s20", "s21")
template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, float a, float b, float sa, int iters) {
  float x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6}, pa = {a, b}, pb = {b, a};
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
    if (KIND == 25) asm volatile(R8(G8(F25)) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));
    CASE(0, F0); CASE(1, F1); CASE(2, F2); CASE(3, F3); CASE(4, F4); CASE(5, F5); CASE(6, F6); CASE(7, F7); CASE(8, F8); CASE(9, F9);
    CASE(10, F10); CASE(11, F11); CASE(12, F12); CASE(13, F13); CASE(14, F14); CASE(15, F15); CASE(16, F16); CASE(17, F17); CASE(18, F18);
    CASE(19, F19); CASE(20, F20); CASE(21, F21); CASE(22, F22); CASE(23, F23); CASE(24, F24); CASE(26, F26); CASE(27, F27); CASE(28, F28); CASE(29, F29);
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0[0] + p1[1] + p2[0] + p3[0] + p4[0] + p5[0] + p6[0] + p7[0];
}
template <int KIND>
static void run(const char* name) {
  float* out;
  hipMalloc(&out, 1 << 24);
  printf("%-44s", name);
  for (int w : {1, 2, 3, 4, 8}) {
    const int blocks = 256 * 4 * w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 64>>>(out, 1.0001f, 0.5f, 0.25f, 64);
    hipEventRecord(e0);
    k<KIND><<<blocks, 64>>>(out, 1.0001f, 0.5f, 0.25f, ITERS);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  w=%d %5.2f", w, ms * 2.4e6 / ((double)ITERS * 64 * w));
  }
  printf("\n");
  hipFree(out);
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  run<0>("v_fma_f32 v,v,v (VOP3)"); run<1>("v_fmac_f32 (VOP2)"); run<2>("v_fmaak_f32 literal"); run<3>("v_fmamk_f32 literal");
  run<4>("v_mul_f32 literal"); run<5>("v_mul_f32 v,v"); run<6>("v_fma_f32 with an SGPR"); run<7>("v_fma_f32 inline constant 2.0");
  run<27>("v_fma_f32 inline constant 0.5 as addend"); run<28>("v_fmac_f32 literal"); run<18>("v_mul_f32 with an SGPR"); run<19>("v_add_f32"); run<20>("v_sub_f32"); run<17>("v_fma_f32 neg/abs modifiers");
  run<8>("v_add_u32"); run<22>("v_and_b32 inline"); run<9>("v_lshl_or_b32"); run<21>("v_bfe_u32"); run<14>("v_add3_u32"); run<29>("v_mov_b32");
  run<10>("v_cvt_f32_f16"); run<11>("v_cvt_pk_f16_f32"); run<23>("v_cvt_f32_i32"); run<24>("v_cvt_i32_f32"); run<12>("v_rndne_f32"); run<13>("v_med3_f32");
  run<26>("v_cmp_gt_f32 -> SGPR pair"); run<15>("v_rcp_f32"); run<16>("v_cos_f32"); run<25>("v_pk_fma_f32");
  return 0;
}
