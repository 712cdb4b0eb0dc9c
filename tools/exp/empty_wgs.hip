// Experiment: how much does a stream of tiny workgroups on ANOTHER queue slow K_sinc down?  (tools/exp/dispatch_contention.py)
#include <hip/hip_runtime.h>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void k_spin(int* p, int iters) {            // a little VALU work per workgroup, no memory
  float a = threadIdx.x;
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  if (a == 12345.678f && p) p[0] = 1;
}
extern "C" void launch_empty(int n_wg, int threads, void* stream) {
  hipLaunchKernelGGL(k_empty, dim3(n_wg), dim3(threads), 0, (hipStream_t)stream, (int*)nullptr);
}
extern "C" void launch_spin(int n_wg, int threads, int iters, void* stream) {
  hipLaunchKernelGGL(k_spin, dim3(n_wg), dim3(threads), 0, (hipStream_t)stream, (int*)nullptr, iters);
}
