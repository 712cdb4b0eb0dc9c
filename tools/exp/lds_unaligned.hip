// Does ds_read_b128 / ds_read_b64 work at 2-byte-aligned LDS addresses on gfx950, and what does it cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out, int shift_bytes, int reps, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (unsigned short)i;
  __syncthreads();
  unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)lds + threadIdx.x * shift_bytes;
  uint4 v, acc = {0, 0, 0, 0};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr + (r & 7) * 16));
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x * 4 + 0] = acc.x; out[threadIdx.x * 4 + 1] = acc.y; out[threadIdx.x * 4 + 2] = acc.z; out[threadIdx.x * 4 + 3] = acc.w;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  uint32_t* d; unsigned long long* c;
  hipMalloc(&d, 64 * 16); hipMalloc(&c, 8);
  for (int sb : {16, 2, 4, 8, 34}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, sb, 1, c);
    uint32_t h[256]; unsigned long long hc;
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int w = 0; w < 4; ++w) {
      unsigned e0 = (l * sb) / 2 + 2 * w, want = e0 | ((e0 + 1) << 16);
      if (h[l * 4 + w] != want) ++bad;
    }
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, sb, 4096, c);
    hipDeviceSynchronize();
    hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    printf("lane stride %2d bytes: err=%s mismatches=%d  cycles/read=%.1f\n", sb, hipGetErrorString(e), bad, hc / 4096.0);
  }
  return 0;
}
