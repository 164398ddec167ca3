// Does non-f64 VALU / SALU / LDS work issue in the shadow of v_mfma_f64_4x4x4_4b_f64 (one wave per SIMD)?
// Each iteration: 8 MFMAs with NV filler instructions after every MFMA.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
struct Stamp { unsigned long long c0, c1; };
// KIND: 0 none, 1 v_xor_b32 (int VALU), 2 v_fma_f32, 3 s_add (SALU), 4 v_readlane, 5 ds_read_b64 (no wait), 6 v_fma_f64, 7 v_mov_dpp
template <int KIND, int NV>
__global__ void __launch_bounds__(256) k(Stamp* st, double* out, int iters, double seed) {
  __shared__ double lds[512];
  lds[threadIdx.x] = seed; lds[threadIdx.x + 256] = seed;
  __syncthreads();
  double acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = 0.0;
  double a = seed + threadIdx.x * 1e-3, b = seed * 0.5 + threadIdx.x * 1e-4;
  unsigned iv[8]; float fv[8]; double dv[8]; int sv = iters;
  for (int i = 0; i < 8; ++i) { iv[i] = threadIdx.x + i; fv[i] = threadIdx.x * 0.5f + i; dv[i] = threadIdx.x + i; }
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int r = (i * NV + j) & 7;
        if (KIND == 1) asm volatile("v_xor_b32 %0, 0x1234, %0" : "+v"(iv[r]));
        if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(fv[r]));
        if (KIND == 3) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sv));
        if (KIND == 4) { int t; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(t) : "v"(iv[r])); sv += 0 * t; }
        if (KIND == 5) { double t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((threadIdx.x & 63) * 8)); }
        if (KIND == 6) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(dv[r]));
        if (KIND == 7) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(iv[r]));
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  double s = 0; for (int i = 0; i < 8; ++i) s += acc[i] + iv[i] + fv[i] + dv[i]; s += sv;
  asm volatile("" :: "v"(s));
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { Stamp x = {c0, c1}; st[blockIdx.x * 4 + (threadIdx.x >> 6)] = x; }
}
template <int KIND, int NV>
static void run(const char* name, int blocks) {
  Stamp* dst; double* dout; int nw = blocks * 4; const int IT = 2000;
  CHECK(hipMalloc(&dst, sizeof(Stamp) * nw)); CHECK(hipMalloc(&dout, sizeof(double) * blocks * 256));
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, NV>), dim3(blocks), dim3(256), 0, 0, dst, dout, IT, 1.0);
  CHECK(hipDeviceSynchronize());
  std::vector<Stamp> h(nw); CHECK(hipMemcpy(h.data(), dst, sizeof(Stamp) * nw, hipMemcpyDeviceToHost));
  std::vector<double> c(nw); for (int i = 0; i < nw; ++i) c[i] = double(h[i].c1 - h[i].c0) / IT / 8.0;
  std::sort(c.begin(), c.end());
  printf("%-46s cycles per (MFMA + %d fillers): %7.2f\n", name, NV, c[nw / 2]);
  CHECK(hipFree(dst)); CHECK(hipFree(dout));
}
int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0)); int n = prop.multiProcessorCount;
  run<0, 0>("mfma only", n);
  run<1, 1>("v_xor_b32", n); run<1, 2>("v_xor_b32", n); run<1, 3>("v_xor_b32", n); run<1, 4>("v_xor_b32", n); run<1, 6>("v_xor_b32", n);
  run<2, 2>("v_fma_f32", n); run<2, 4>("v_fma_f32", n);
  run<3, 2>("s_add_u32", n); run<3, 4>("s_add_u32", n); run<3, 8>("s_add_u32", n);
  run<4, 2>("v_readlane_b32", n); run<4, 4>("v_readlane_b32", n);
  run<5, 1>("ds_read_b64", n); run<5, 2>("ds_read_b64", n);
  run<6, 1>("v_fma_f64", n); run<6, 2>("v_fma_f64", n);
  run<7, 2>("v_mov_b32_dpp", n); run<7, 4>("v_mov_b32_dpp", n);
  return 0;
}
