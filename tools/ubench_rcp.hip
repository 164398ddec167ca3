// Accuracy of v_rcp_f64 on gfx950 and of one / two Newton-Raphson refinements (used by K3's pivot reciprocal).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = x[i];
  double y = __builtin_amdgcn_rcp(d);
  r0[i] = y;
  y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
  r1[i] = y;
  y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
  r2[i] = y;
}
int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), r0(n), r1(n), r2(n);
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> u(-30.0, 30.0), m(1.0, 2.0);
  for (int i = 0; i < n; ++i) x[i] = std::ldexp(m(g), (int)u(g));
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / (long double)x[i];
    e0 = std::fmax(e0, (double)fabsl(((long double)r0[i] - t) / t));
    e1 = std::fmax(e1, (double)fabsl(((long double)r1[i] - t) / t));
    e2 = std::fmax(e2, (double)fabsl(((long double)r2[i] - t) / t));
  }
  printf("max rel err: v_rcp_f64 %.3e (2^%.1f)  +1 Newton %.3e (2^%.1f)  +2 Newton %.3e (2^%.1f)\n", e0, std::log2(e0), e1,
         std::log2(e1), e2, std::log2(e2));
  return 0;
}
