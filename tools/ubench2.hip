// Instruction-rate probe for gfx950 (K_sinc's placement prologue is priced with these): each kernel issues ITERS x 8
// independent copies of ONE instruction through inline asm (nothing for the compiler to fold) and the host converts
// the time into cycles per wave-instruction per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench2.hip -o tools/ubench2
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int ITERS = 1024;
#define X8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

// f32 -> f32 unary / binary with a register operand
#define K_F32(name, text)                                                        \
  __global__ void name(float* out, float a) {                                    \
    float v[8];                                                                  \
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i + a;              \
    for (int it = 0; it < ITERS; ++it) {                                         \
      asm volatile(text(0) text(1) text(2) text(3) text(4) text(5) text(6) text(7) \
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) \
                   : "v"(a));                                                    \
    }                                                                            \
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i];                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                              \
  }
#define K_F64(name, text)                                                        \
  __global__ void name(float* out, double a) {                                   \
    double v[8];                                                                 \
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3 + i + a;               \
    for (int it = 0; it < ITERS; ++it) {                                         \
      asm volatile(text(0) text(1) text(2) text(3) text(4) text(5) text(6) text(7) \
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) \
                   : "v"(a));                                                    \
    }                                                                            \
    double s = 0; for (int i = 0; i < 8; ++i) s += v[i];                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;                       \
  }
// f64 source, 32-bit destination (and the reverse): destination registers are scratch
#define K_CVT(name, text, DT, ST)                                                \
  __global__ void name(float* out, double a) {                                   \
    ST v[8]; DT d[8];                                                            \
    for (int i = 0; i < 8; ++i) { v[i] = (ST)(threadIdx.x * 1e-3 + i + a); d[i] = (DT)0; } \
    for (int it = 0; it < ITERS; ++it) {                                         \
      asm volatile(text(0) text(1) text(2) text(3) text(4) text(5) text(6) text(7) \
                   : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) \
                   : [s0] "v"(v[0]), [s1] "v"(v[1]), [s2] "v"(v[2]), [s3] "v"(v[3]), [s4] "v"(v[4]), [s5] "v"(v[5]), [s6] "v"(v[6]), [s7] "v"(v[7])); \
    }                                                                            \
    double s = 0; for (int i = 0; i < 8; ++i) s += (double)d[i];                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;                       \
  }

#define T_ADD32(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define T_MUL32(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define T_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define T_FMAC32(i) "v_fmac_f32 %" #i ", %8, %8\n"
#define T_RND32(i) "v_rndne_f32 %" #i ", %" #i "\n"
#define T_RCP32(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define T_MOV32(i) "v_mov_b32 %" #i ", %8\n"
#define T_CNDM(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define T_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define T_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define T_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define T_CMP32(i) "v_cmp_gt_f32 vcc, %" #i ", %8\n"
#define T_CVTI32F32(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define T_ADD64(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define T_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define T_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %8, %8\n"
#define T_RND64(i) "v_rndne_f64 %" #i ", %" #i "\n"
#define T_RCP64(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define T_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 3, %8\n"
#define T_CMP64(i) "v_cmp_gt_f64 vcc, %" #i ", %8\n"
#define T_CVT_F64_I32(i) "v_cvt_f64_i32 %" #i ", %[s" #i "]\n"
#define T_CVT_I32_F64(i) "v_cvt_i32_f64 %" #i ", %[s" #i "]\n"
#define T_CVT_F32_F64(i) "v_cvt_f32_f64 %" #i ", %[s" #i "]\n"
#define T_CVT_F64_F32(i) "v_cvt_f64_f32 %" #i ", %[s" #i "]\n"

K_F32(k_add32, T_ADD32)
K_F32(k_mul32, T_MUL32)
K_F32(k_fma32, T_FMA32)
K_F32(k_fmac32, T_FMAC32)
K_F32(k_rnd32, T_RND32)
K_F32(k_rcp32, T_RCP32)
K_F32(k_mov32, T_MOV32)
K_F32(k_cndm, T_CNDM)
K_F32(k_addu, T_ADDU)
K_F32(k_mullo, T_MULLO)
K_F32(k_mul24, T_MUL24)
K_F32(k_cmp32, T_CMP32)
K_F32(k_cvti32f32, T_CVTI32F32)
K_F64(k_add64, T_ADD64)
K_F64(k_mul64, T_MUL64)
K_F64(k_fma64, T_FMA64)
K_F64(k_rnd64, T_RND64)
K_F64(k_rcp64, T_RCP64)
K_F64(k_lshladd64, T_LSHLADD64)
K_F64(k_cmp64, T_CMP64)
K_CVT(k_cvt_f64_i32, T_CVT_F64_I32, double, int)
K_CVT(k_cvt_i32_f64, T_CVT_I32_F64, int, double)
K_CVT(k_cvt_f32_f64, T_CVT_F32_F64, float, double)
K_CVT(k_cvt_f64_f32, T_CVT_F64_F32, double, float)


// forms with a scalar (SGPR) operand and with three distinct VGPR sources
#define K_F32S(name, text)                                                       \
  __global__ void name(float* out, float a) {                                    \
    float v[8];                                                                  \
    const float b = a * 1.5f + threadIdx.x;                                      \
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i + a;              \
    for (int it = 0; it < ITERS; ++it) {                                         \
      asm volatile(text(0) text(1) text(2) text(3) text(4) text(5) text(6) text(7) \
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) \
                   : "s"(a), "v"(b), "v"(a));                                    \
    }                                                                            \
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i];                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                              \
  }
#define T_FMA_VSV(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define T_FMA_SVV(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define T_FMAC_SV(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define T_FMA_VVV(i) "v_fma_f32 %" #i ", %" #i ", %9, %10\n"
#define T_FMA_SSV(i) "v_fma_f32 %" #i ", %8, %8, %" #i "\n"
#define T_FMA_VSS(i) "v_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define T_FMA_NEG(i) "v_fma_f32 %" #i ", -%" #i ", %9, %" #i "\n"
#define T_FMAAK(i) "v_fmaak_f32 %" #i ", %" #i ", %9, 0x3f000000\n"
#define T_FMAMK(i) "v_fmamk_f32 %" #i ", %" #i ", 0x3f800100, %9\n"
#define T_MUL_SV(i) "v_mul_f32 %" #i ", %8, %" #i "\n"
#define T_ADD_VV2(i) "v_add_f32 %" #i ", %9, %10\n"
#define T_SUB_VV(i) "v_sub_f32 %" #i ", %" #i ", %9\n"
#define T_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %9, %10\n"
#define T_PKFMA_S(i) "v_pk_fma_f32 %" #i ", %9, s[20:21], %" #i "\n"
#define T_PKFMA_ACC(i) "v_pk_fma_f32 %" #i ", %9, %10, %" #i "\n"
#define T_PKADD_SEL(i) "v_pk_add_f32 %" #i ", %9, %9 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n"
#define T_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %9\n"
#define T_CNDM_S(i) "v_cndmask_b32 %" #i ", %" #i ", %9, s[20:21]\n"
#define T_MOV_DPP(i) "v_mov_b32_dpp %" #i ", %9 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define T_ADD_DPP(i) "v_add_f32_dpp %" #i ", %9, %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
K_F32S(k_fma_vsv, T_FMA_VSV)
K_F32S(k_fma_svv, T_FMA_SVV)
K_F32S(k_fmac_sv, T_FMAC_SV)
K_F32S(k_fma_vvv, T_FMA_VVV)
K_F32S(k_fma_ssv, T_FMA_SSV)
K_F32S(k_fma_vss, T_FMA_VSS)
K_F32S(k_fma_neg, T_FMA_NEG)
K_F32S(k_fmaak, T_FMAAK)
K_F32S(k_fmamk, T_FMAMK)
K_F32S(k_mul_sv, T_MUL_SV)
K_F32S(k_add_vv2, T_ADD_VV2)
K_F32S(k_sub_vv, T_SUB_VV)
K_F32S(k_cndm_s, T_CNDM_S)
K_F32S(k_mov_dpp, T_MOV_DPP)
K_F32S(k_add_dpp, T_ADD_DPP)
// packed f32: 64-bit register pairs
__global__ void k_pkfma2(float* out, float a) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 v[8];
  const f2 b = {a * 1.5f + threadIdx.x, a}, c = {a, a * 0.5f};
  for (int i = 0; i < 8; ++i) v[i] = (f2){threadIdx.x * 1e-3f + i + a, 1.0f * i};
  for (int it = 0; it < ITERS; ++it) {
    asm volatile(T_PKFMA(0) T_PKFMA(1) T_PKFMA(2) T_PKFMA(3) T_PKFMA(4) T_PKFMA(5) T_PKFMA(6) T_PKFMA(7)
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 : "s"(a), "v"(b), "v"(c));
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// packed f32 forms the K_sinc tap loops could use: accumulate (e, d) += (E, D) * (A, nA) with the constant pair in SGPRs,
// or in VGPRs; (E, D) = (sp + sm, sp - sm) from one register pair by op_sel / neg modifiers; packed multiply
#define K_PK(NAME, T)                                                                                                   \
  __global__ void NAME(float* out, float a) {                                                                            \
    typedef float f2 __attribute__((ext_vector_type(2)));                                                               \
    f2 v[8];                                                                                                             \
    const f2 b = {a * 1.5f + threadIdx.x, a}, c = {a, a * 0.5f};                                                         \
    for (int i = 0; i < 8; ++i) v[i] = (f2){threadIdx.x * 1e-3f + i + a, 1.0f * i};                                      \
    for (int it = 0; it < ITERS; ++it) {                                                                                 \
      asm volatile("s_mov_b32 s20, 0x3f800100\ns_mov_b32 s21, 0x3f7fff00\n" T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)      \
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])       \
                   : "s"(a), "v"(b), "v"(c)                                                                               \
                   : "s20", "s21");                                                                                       \
    }                                                                                                                    \
    float s = 0;                                                                                                         \
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;                                                                    \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                      \
  }
K_PK(k_pkfma_sgpr, T_PKFMA_S)
K_PK(k_pkfma_acc, T_PKFMA_ACC)
K_PK(k_pkadd_sel, T_PKADD_SEL)
K_PK(k_pkmul, T_PKMUL)

// pure LDS reads at lane-consecutive addresses (conflict-free), results discarded by the hardware wait only
template <int KIND>
__global__ void k_ldsread(float* out) {
  __shared__ __attribute__((aligned(16))) float buf[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) buf[i] = i;
  __syncthreads();
  typedef __attribute__((address_space(3))) float lf;
  const unsigned a32 = (unsigned)(size_t)(lf*)buf + threadIdx.x * (KIND == 2 ? 16 : 4);
  float r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  for (int it = 0; it < ITERS; ++it) {
    if (KIND == 0) {
      float x0, x1, x2, x3, x4, x5, x6, x7;
      asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:1024\n ds_read_b32 %2, %8 offset:2048\n ds_read_b32 %3, %8 offset:3072\n"
                   "ds_read_b32 %4, %8 offset:4096\n ds_read_b32 %5, %8 offset:5120\n ds_read_b32 %6, %8 offset:6144\n ds_read_b32 %7, %8 offset:7168\n s_waitcnt lgkmcnt(0)\n"
                   : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7) : "v"(a32));
      r0 = x0; r1 = x7;
    } else if (KIND == 1) {
      float2 x0, x1, x2, x3, x4, x5, x6, x7;
      asm volatile("ds_read2_b32 %0, %8 offset1:1\n ds_read2_b32 %1, %8 offset0:2 offset1:3\n ds_read2_b32 %2, %8 offset0:4 offset1:5\n ds_read2_b32 %3, %8 offset0:6 offset1:7\n"
                   "ds_read2_b32 %4, %8 offset0:8 offset1:9\n ds_read2_b32 %5, %8 offset0:10 offset1:11\n ds_read2_b32 %6, %8 offset0:12 offset1:13\n ds_read2_b32 %7, %8 offset0:14 offset1:15\n s_waitcnt lgkmcnt(0)\n"
                   : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7) : "v"(a32));
      r0 = x0.x; r1 = x7.y;
    } else {
      float4 x0, x1, x2, x3, x4, x5, x6, x7;
      asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:4096\n ds_read_b128 %2, %8 offset:8192\n ds_read_b128 %3, %8 offset:12288\n"
                   "ds_read_b128 %4, %8 offset:16384\n ds_read_b128 %5, %8 offset:20480\n ds_read_b128 %6, %8 offset:24576\n ds_read_b128 %7, %8 offset:28672\n s_waitcnt lgkmcnt(0)\n"
                   : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7) : "v"(a32));
      r0 = x0.x; r1 = x7.w;
    }
    r2 += r0; r3 += r1;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r2 + r3;
}

// f32 FMA stream and LDS read stream together (do they overlap?)
__global__ void k_fma_lds(float* out, float a) {
  __shared__ __attribute__((aligned(16))) float buf[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) buf[i] = i;
  __syncthreads();
  typedef __attribute__((address_space(3))) float lf;
  const unsigned a32 = (unsigned)(size_t)(lf*)buf + threadIdx.x * 4;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i + a;
  float acc = 0;
  for (int it = 0; it < ITERS; ++it) {
    float2 x0, x1;
    asm volatile("ds_read2_b32 %8, %10 offset1:1\n ds_read2_b32 %9, %10 offset0:2 offset1:3\n"
                 "v_fma_f32 %0, %0, %11, %11\n v_fma_f32 %1, %1, %11, %11\n v_fma_f32 %2, %2, %11, %11\n v_fma_f32 %3, %3, %11, %11\n"
                 "v_fma_f32 %4, %4, %11, %11\n v_fma_f32 %5, %5, %11, %11\n v_fma_f32 %6, %6, %11, %11\n v_fma_f32 %7, %7, %11, %11\n"
                 "v_fma_f32 %0, %0, %11, %11\n v_fma_f32 %1, %1, %11, %11\n v_fma_f32 %2, %2, %11, %11\n v_fma_f32 %3, %3, %11, %11\n"
                 "v_fma_f32 %4, %4, %11, %11\n v_fma_f32 %5, %5, %11, %11\n v_fma_f32 %6, %6, %11, %11\n v_fma_f32 %7, %7, %11, %11\n"
                 "s_waitcnt lgkmcnt(0)\n"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "=v"(x0), "=v"(x1)
                 : "v"(a32), "v"(a));
    acc += x0.x + x1.y;
  }
  float s = acc; for (int i = 0; i < 8; ++i) s += v[i];
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

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
  const int wps = argc > 1 ? atoi(argv[1]) : 8;            // waves per SIMD
  const int blocks = p.multiProcessorCount * wps, threads = 256;
  const double lanes = (double)blocks * threads;
  printf("device %s  CUs %d  clock %.0f MHz  blocks %d x %d (%d waves/SIMD)\n", p.name, p.multiProcessorCount, p.clockRate / 1e3, blocks, threads, wps);
  void* out; hipMalloc(&out, lanes * 8);
  const double simds = p.multiProcessorCount * 4.0, clk = p.clockRate * 1e3;
  auto report = [&](const char* name, double ms, double ops_per_lane) {
    const double wave_instr = lanes * ops_per_lane / (ms * 1e-3) / 64.0;
    printf("%-30s %8.3f ms  %7.2f cycles per wave-instr per SIMD (nominal clock)\n", name, ms, simds * clk / wave_instr);
  };
#define RUN32(k) report(#k, time_ms([&] { hipLaunchKernelGGL(k, blocks, threads, 0, 0, (float*)out, 1.0001f); }), ITERS * 8.0)
#define RUN64(k) report(#k, time_ms([&] { hipLaunchKernelGGL(k, blocks, threads, 0, 0, (float*)out, 1.0001); }), ITERS * 8.0)
  RUN32(k_fma32); RUN32(k_fmac32); RUN32(k_add32); RUN32(k_mul32); RUN32(k_mov32); RUN32(k_cndm); RUN32(k_addu); RUN32(k_mullo);
  RUN32(k_mul24); RUN32(k_cmp32); RUN32(k_rnd32); RUN32(k_cvti32f32); RUN32(k_rcp32);
  RUN32(k_fma_vsv); RUN32(k_fma_svv); RUN32(k_fmac_sv); RUN32(k_fma_vvv); RUN32(k_fma_ssv); RUN32(k_fma_vss); RUN32(k_fma_neg);
  RUN32(k_fmaak); RUN32(k_fmamk); RUN32(k_mul_sv); RUN32(k_add_vv2); RUN32(k_sub_vv); RUN32(k_cndm_s); RUN32(k_mov_dpp); RUN32(k_add_dpp);
  RUN32(k_pkfma2); RUN32(k_pkfma_sgpr); RUN32(k_pkfma_acc); RUN32(k_pkadd_sel); RUN32(k_pkmul);
  RUN64(k_fma64); RUN64(k_add64); RUN64(k_mul64); RUN64(k_rnd64); RUN64(k_rcp64); RUN64(k_lshladd64); RUN64(k_cmp64);
  RUN64(k_cvt_f64_i32); RUN64(k_cvt_i32_f64); RUN64(k_cvt_f32_f64); RUN64(k_cvt_f64_f32);
  report("ds_read_b32", time_ms([&] { hipLaunchKernelGGL(k_ldsread<0>, blocks, threads, 0, 0, (float*)out); }), ITERS * 8.0);
  report("ds_read2_b32", time_ms([&] { hipLaunchKernelGGL(k_ldsread<1>, blocks, threads, 0, 0, (float*)out); }), ITERS * 8.0);
  report("ds_read_b128", time_ms([&] { hipLaunchKernelGGL(k_ldsread<2>, blocks, threads, 0, 0, (float*)out); }), ITERS * 8.0);
  report("16 fma + 2 ds_read2 (per fma)", time_ms([&] { hipLaunchKernelGGL(k_fma_lds, blocks, threads, 0, 0, (float*)out, 1.0001f); }), ITERS * 16.0);
  return 0;
}
