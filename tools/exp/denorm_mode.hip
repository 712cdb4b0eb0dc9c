// Does MODE.FP_DENORM[3:2] = 0 make v_cvt_f16_f32 flush subnormal results, and v_cvt_f32_f16 flush subnormal inputs?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int flush) {
  if (flush) __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);
  const float x = in[threadIdx.x];
  const _Float16 h = (_Float16)x;
  unsigned short b;
  __builtin_memcpy(&b, &h, 2);
  const float back = (float)h;
  out[threadIdx.x * 2] = b;
  out[threadIdx.x * 2 + 1] = __float_as_uint(back);
}
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
// the packed conversion (v_cvt_pk_f16_f32, gfx950) and the mixed-precision fma that reads a half operand
__global__ void k2(const float* in, unsigned* out, float* lo) {
  __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);
  const float x0 = in[threadIdx.x], x1 = in[threadIdx.x ^ 1];
  const half2v h = {(_Float16)x0, (_Float16)x1};
  half2v c;
  asm volatile("v_pk_max_f16 %0, %1, %1" : "=v"(c) : "v"(h));
  out[threadIdx.x * 2] = __builtin_bit_cast(unsigned, h);
  out[threadIdx.x * 2 + 1] = __builtin_bit_cast(unsigned, c);
  lo[threadIdx.x] = (x0 - (float)h[0]) * 4096.0f;
}
int main() {
  float h_in[4] = {3.0e-5f, 6.2e-5f, 1.0e-7f, -5.0e-5f}, *d_in;
  unsigned h_out[8], *d_out;
  hipMalloc(&d_in, 16); hipMalloc(&d_out, 32);
  hipMemcpy(d_in, h_in, 16, hipMemcpyHostToDevice);
  for (int f = 0; f < 2; ++f) {
    k<<<1, 4>>>(d_in, d_out, f);
    hipMemcpy(h_out, d_out, 32, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; ++i) { float b; __builtin_memcpy(&b, &h_out[2 * i + 1], 4); printf("flush=%d x=%g half bits %04x back %g\n", f, h_in[i], h_out[2 * i], b); }
  }
  float h_lo[4], *d_lo;
  hipMalloc(&d_lo, 16);
  k2<<<1, 4>>>(d_in, d_out, d_lo);
  hipMemcpy(h_out, d_out, 32, hipMemcpyDeviceToHost);
  hipMemcpy(h_lo, d_lo, 16, hipMemcpyDeviceToHost);
  for (int i = 0; i < 4; ++i) printf("packed: x=%g pair bits %08x after v_pk_max %08x lo %g (x*4096 = %g)\n", h_in[i], h_out[2 * i], h_out[2 * i + 1], h_lo[i], h_in[i] * 4096.0f);
}
