// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns K_sinc uses (the x2 correction of
// MI355X_MICROARCH.md is stated for 16 B/lane streaming reads).  Each kernel reads a buffer of known size exactly once:
//   k_read4   4 B per lane, consecutive lanes consecutive words (r01's staging loop)
//   k_read16  16 B per lane
//   k_dma4    global_load_lds_dword, 4 B per lane straight into LDS (r02's staging)
// Run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/ubench_fetch    and compare FETCH_SIZE (KiB) with the
// bytes printed here.       build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_fetch tools/ubench_fetch.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_read4(const float* __restrict__ p, size_t n, float* __restrict__ out) {
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void k_read16(const float4* __restrict__ p, size_t n4, float* __restrict__ out) {
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = p[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void k_dma4(const float* __restrict__ p, size_t n, float* __restrict__ out) {
  __shared__ float tile[4 * 1024];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  float acc = 0.0f;
  float* mine = tile + w * 1024;
  for (size_t base = ((size_t)blockIdx.x * 4 + w) * 1024; base + 1024 <= n; base += (size_t)gridDim.x * 4096) {
    const float* gp = p + base + l;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + q * 64),
                                       (__attribute__((address_space(3))) void*)(mine + q * 64), 4, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    acc += mine[(l * 17) & 1023];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const size_t n = (size_t)1 << 30;                 // 4 GiB of floats: far beyond L2 + MALL
  float *buf, *out;
  hipMalloc(&buf, n * sizeof(float));
  hipMalloc(&out, 4);
  hipMemset(buf, 0, n * sizeof(float));
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k_read4, dim3(8192), dim3(256), 0, 0, buf, n, out);
  hipLaunchKernelGGL(k_read16, dim3(8192), dim3(256), 0, 0, (const float4*)buf, n / 4, out);
  hipLaunchKernelGGL(k_dma4, dim3(8192), dim3(256), 0, 0, buf, n, out);
  hipDeviceSynchronize();
  printf("each kernel reads %zu bytes = %zu KiB exactly once\n", n * sizeof(float), n * sizeof(float) / 1024);
  return 0;
}
