// Does the BLGP field of v_mfma_f64_4x4x4_4b_f64 act as NEG[2:0] (negate A, B, C) on gfx950?  Diagnostic.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out) {
  int l = threadIdx.x;
  double a = 1.0 + l, b = 2.0 + 0.5 * l, c = 100.0;
  out[l * 4 + 0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
  out[l * 4 + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 1);
  out[l * 4 + 2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 2);
  out[l * 4 + 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 4);
}
int main() {
  double* d; hipMalloc(&d, 64 * 4 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 21) {
    double p = h[l * 4] - 100.0;   // sum_k a b
    printf("lane %2d: blgp0 %.3f (ab=%.3f)  blgp1 %.3f  blgp2 %.3f  blgp4 %.3f\n", l, h[l * 4], p, h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
