// accuracy of the hardware v_sin_f32 / v_cos_f32 (input in revolutions) against float64, on a grid of [-1, 1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float* x, float* s, float* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { s[i] = __builtin_amdgcn_sinf(x[i]); c[i] = __builtin_amdgcn_cosf(x[i]); }
}
int main() {
  const int n = 1 << 22;
  std::vector<float> x(n), s(n), c(n);
  for (int i = 0; i < n; ++i) x[i] = (float)(-1.0 + 2.0 * (double)i / n);
  float *dx, *ds, *dc;
  hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, ds, dc, n);
  hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
  double es = 0, ec = 0, es_small = 0; int is = 0, ic = 0;
  for (int i = 0; i < n; ++i) {
    double a = 2.0 * M_PI * (double)x[i];
    double d1 = fabs((double)s[i] - sin(a)), d2 = fabs((double)c[i] - cos(a));
    if (d1 > es) { es = d1; is = i; }
    if (d2 > ec) { ec = d2; ic = i; }
    if (fabs(x[i]) < 0.01) es_small = fmax(es_small, d1 / fmax(fabs(sin(a)), 1e-30));
  }
  printf("v_sin_f32 max abs err %.3e at x=%g; v_cos_f32 max abs err %.3e at x=%g; sin relative err for |x|<0.01: %.3e\n", es, x[is], ec, x[ic], es_small);
  return 0;
}
