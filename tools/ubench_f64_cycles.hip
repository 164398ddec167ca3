// In-kernel cycle stamps (s_memtime / s_memrealtime) for the f64 MFMA and f64 VALU FMA on gfx950.
// Reports shader cycles per instruction and the clock the chip holds, idle (1 wave on the chip)
// and loaded (every SIMD busy).  Diagnostic build only; nothing here ships in the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

struct Stamp { unsigned long long c0, c1, r0, r1; };

template <int NACC, int KIND>  // KIND 0: mfma 16x16x4, 1: valu fma, 2: mfma 4x4x4_4b
__global__ void __launch_bounds__(256) k_stamp(Stamp* st, double* out, int iters, double seed) {
  d4 acc[NACC];
  double v[NACC * 4];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  for (int i = 0; i < NACC * 4; ++i) v[i] = threadIdx.x * 1e-6 + i;
  double a = seed + threadIdx.x * 1e-3, b = seed * 0.5 + threadIdx.x * 1e-4;
  double fa = 1.0 + seed * 1e-9, fb = seed * 1e-7;
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      if (KIND == 2) acc[i][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i][0], 0, 0, 0);
      if (KIND == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i * 4 + j] = __builtin_fma(v[i * 4 + j], fa, fb);
      }
    }
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < NACC * 4; ++i) s += v[i];
  // make the result live before the closing stamp
  asm volatile("" :: "v"(s));
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    Stamp x = {c0, c1, r0, r1};
    st[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = x;
  }
}

template <int NACC, int KIND>
static void run(const char* name, int blocks, int threads, int iters, int instr_per_iter, double flop_per_instr) {
  Stamp* dst; double* dout;
  int nw = blocks * (threads / 64);
  CHECK(hipMalloc(&dst, sizeof(Stamp) * nw)); CHECK(hipMalloc(&dout, sizeof(double) * blocks * threads));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k_stamp<NACC, KIND>), dim3(blocks), dim3(threads), 0, 0, dst, dout, iters, 1.0);
  }
  CHECK(hipDeviceSynchronize());
  std::vector<Stamp> h(nw);
  CHECK(hipMemcpy(h.data(), dst, sizeof(Stamp) * nw, hipMemcpyDeviceToHost));
  std::vector<double> cyc(nw), clk(nw);
  for (int i = 0; i < nw; ++i) {
    cyc[i] = double(h[i].c1 - h[i].c0) / (double(iters) * instr_per_iter);
    clk[i] = double(h[i].c1 - h[i].c0) / double(h[i].r1 - h[i].r0) * 100.0;  // MHz
  }
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  double c = cyc[nw / 2], f = clk[nw / 2];
  printf("%-52s cyc/instr(median) %7.2f   clock %7.1f MHz   flop/clk/wave %6.2f\n", name, c, f, flop_per_instr / c);
  CHECK(hipFree(dst)); CHECK(hipFree(dout));
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int ncu = prop.multiProcessorCount;
  const int IT = 4000;
  run<4, 0>("mfma16x16x4 f64, 1 wave on chip, 4 acc", 1, 64, IT, 4, 2048);
  run<1, 0>("mfma16x16x4 f64, 1 wave on chip, 1 acc (dep chain)", 1, 64, IT, 1, 2048);
  run<8, 0>("mfma16x16x4 f64, 1 wave on chip, 8 acc", 1, 64, IT, 8, 2048);
  run<4, 0>("mfma16x16x4 f64, 1 wave/SIMD all CUs, 4 acc", ncu, 256, IT, 4, 2048);
  run<4, 0>("mfma16x16x4 f64, 2 waves/SIMD all CUs, 4 acc", ncu * 2, 256, IT, 4, 2048);
  run<4, 0>("mfma16x16x4 f64, 4 waves/SIMD all CUs, 4 acc", ncu * 4, 256, IT, 4, 2048);
  run<4, 2>("mfma4x4x4_4b f64, 1 wave on chip, 4 acc", 1, 64, IT, 4, 512);
  run<4, 2>("mfma4x4x4_4b f64, 1 wave/SIMD all CUs, 4 acc", ncu, 256, IT, 4, 512);
  run<4, 1>("v_fma_f64, 1 wave on chip, 16 indep", 1, 64, IT, 16, 128);
  run<4, 1>("v_fma_f64, 1 wave/SIMD all CUs, 16 indep", ncu, 256, IT, 16, 128);
  run<4, 1>("v_fma_f64, 2 waves/SIMD all CUs, 16 indep", ncu * 2, 256, IT, 16, 128);
  run<4, 1>("v_fma_f64, 4 waves/SIMD all CUs, 16 indep", ncu * 4, 256, IT, 16, 128);
  return 0;
}
