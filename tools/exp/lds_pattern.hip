// Throughput of ds_read_b128 for lane address patterns (cycles per instruction per wave, 1 wave per SIMD and 4 waves/SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ void k(uint32_t* out, int reps, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x & 63;
  unsigned off;
  if (KIND == 0) off = l * 16;                                  // distinct aligned
  if (KIND == 1) off = (l & 15) * 8 + (l >> 4) * 16;            // ubench B operand: 8-byte stride, overlapping
  if (KIND == 2) off = (l & 15) * 2 + (l >> 4) * 16;            // first attempt: 2-byte stride
  if (KIND == 3) off = (l & 15) * 32 + (l >> 4) * 16;           // 32-byte stride (blocks of 16 halves), g interleaved
  if (KIND == 4) off = 0;                                       // broadcast
  if (KIND == 5) off = (l & 15) * 16 + (l >> 4) * 256;          // distinct, 16-lane groups far apart
  unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)lds + off + (threadIdx.x >> 6) * 2048;
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  u4 v[8], acc = {0, 0, 0, 0};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < 8; ++u) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[u]) : "v"(addr), "n"(u * 64));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y ^= v[u].y; acc.z += v[u].z; acc.w ^= v[u].w; }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND> void run(const char* name, uint32_t* d, unsigned long long* c) {
  for (int threads : {64, 256}) {
    hipLaunchKernelGGL(k<KIND>, dim3(256 * 4), dim3(threads), 0, 0, d, 2000, c);
    hipDeviceSynchronize();
    unsigned long long hc;
    hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    printf("%-44s %3d threads/WG: %.1f cycles per ds_read_b128 (wave view)\n", name, threads, hc / (2000.0 * 8));
  }
}
int main() {
  uint32_t* d; unsigned long long* c;
  hipMalloc(&d, 4096); hipMalloc(&c, 8);
  run<0>("distinct aligned 16 B", d, c);
  run<5>("distinct, groups 256 B apart", d, c);
  run<3>("32-byte stride + 16 g", d, c);
  run<1>("8-byte stride overlapping (B operand)", d, c);
  run<2>("2-byte stride overlapping", d, c);
  run<4>("broadcast", d, c);
  return 0;
}
