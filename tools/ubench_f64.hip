// Micro-benchmarks that pin the gfx950 facts the MVAR/ffDTF kernels rely on:
//   1. the lane<->element map of v_mfma_f64_16x16x4_f64 (A, B and C/D),
//   2. its issue rate (FLOP/clk/SIMD) with 1 and 2 waves per SIMD,
//   3. the v_fma_f64 VALU rate,
//   4. whether the MFMA pipe and the VALU pipe deliver f64 FMAs concurrently.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_f64.hip -o tools/ubench_f64
// Not part of the product; results are recorded in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void k_layout(const double* A /*16x4*/, const double* B /*4x16*/, double* Craw /*64 lanes x 4*/) {
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) Craw[l * 4 + r] = c[r];
}

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma_rate(double* out, int iters, double seed) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = seed + threadIdx.x * 1e-3, b = seed * 0.5 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) k_valu_rate(double* out, int iters, double seed) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-6 + i;
  double a = 1.0 + seed * 1e-9, b = seed * 1e-7;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Each wave interleaves NM MFMAs with NV independent VALU FMAs per iteration.
template <int NM, int NV>
__global__ void __launch_bounds__(256) k_mixed_rate(double* out, int iters, double seed) {
  d4 acc[NM];
  double v[NV];
  for (int i = 0; i < NM; ++i) acc[i] = (d4){0, 0, 0, 0};
  for (int i = 0; i < NV; ++i) v[i] = threadIdx.x * 1e-6 + i;
  double a = seed + threadIdx.x * 1e-3, b = seed * 0.5 + threadIdx.x * 1e-4;
  double fa = 1.0 + seed * 1e-9, fb = seed * 1e-7;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV / NM; ++j) v[i * (NV / NM) + j] = __builtin_fma(v[i * (NV / NM) + j], fa, fb);
    }
  }
  double s = 0;
  for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < NV; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Waves 0..3 of a 512-thread block run MFMAs, waves 4..7 run VALU FMAs (two waves per SIMD).
__global__ void __launch_bounds__(512) k_split_rate(double* out, int iters, double seed) {
  int wave = threadIdx.x >> 6;
  double s = 0;
  if (wave < 4) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = seed + threadIdx.x * 1e-3, b = seed * 0.5 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-6 + i;
    double fa = 1.0 + seed * 1e-9, fb = seed * 1e-7;
    // 4 MFMAs = 4*2048 flop; 16 FMAs*64 lanes*2 = 2048 flop -> run 4x the iterations worth per loop
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], fa, fb);
    }
    for (int i = 0; i < 16; ++i) s += v[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double time_ms(hipStream_t st, void (*launch)(hipStream_t)) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(st);  // warm
  CHECK(hipStreamSynchronize(st));
  CHECK(hipEventRecord(e0, st));
  for (int r = 0; r < 5; ++r) launch(st);
  CHECK(hipEventRecord(e1, st));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 5.0;
}

static double* g_out;
static const int ITERS = 20000;
static int g_blocks = 256;

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  int ncu = prop.multiProcessorCount;
  g_blocks = ncu;
  CHECK(hipMalloc(&g_out, sizeof(double) * 4096 * 512));
  hipStream_t st; CHECK(hipStreamCreate(&st));

  // ---- 1. layout ----
  {
    std::vector<double> A(64), B(64), C(256, 0.0), raw(256);
    for (int i = 0; i < 64; ++i) { A[i] = (double)(rand() % 17 - 8); B[i] = (double)(rand() % 13 - 6); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; C[i * 16 + j] = s; }
    double *dA, *dB, *dC;
    CHECK(hipMalloc(&dA, 64 * 8)); CHECK(hipMalloc(&dB, 64 * 8)); CHECK(hipMalloc(&dC, 256 * 8));
    CHECK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, st, dA, dB, dC);
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(raw.data(), dC, 256 * 8, hipMemcpyDeviceToHost));
    double e1 = 0, e2 = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
      int col = l & 15;
      int rowA = (l >> 4) + 4 * r;      // f64 map claimed by the guide
      int rowB = (l >> 4) * 4 + r;      // f32 16x16x4 style map
      e1 = fmax(e1, fabs(raw[l * 4 + r] - C[rowA * 16 + col]));
      e2 = fmax(e2, fabs(raw[l * 4 + r] - C[rowB * 16 + col]));
    }
    printf("layout: err(row=(lane>>4)+4*reg)=%g  err(row=4*(lane>>4)+reg)=%g\n", e1, e2);
  }

  // ---- 2. MFMA rate ----
  auto report = [&](const char* name, double ms, double flop) {
    double tf = flop / (ms * 1e-3) / 1e12;
    double per_clk_simd = flop / (ms * 1e-3) / (ncu * 4.0) / 2.4e9;
    printf("%-44s %9.3f ms  %8.2f TFLOP/s  %6.2f flop/clk/SIMD@2.4GHz\n", name, ms, tf, per_clk_simd);
  };
  {
    double ms = time_ms(st, [](hipStream_t s) { hipLaunchKernelGGL(k_mfma_rate<4>, dim3(g_blocks), dim3(256), 0, s, g_out, ITERS, 1.0); });
    report("mfma f64 16x16x4, 1 wave/SIMD, 4 acc", ms, (double)g_blocks * 4 * ITERS * 4 * 2048.0);
    ms = time_ms(st, [](hipStream_t s) { hipLaunchKernelGGL(k_mfma_rate<1>, dim3(g_blocks), dim3(256), 0, s, g_out, ITERS, 1.0); });
    report("mfma f64 16x16x4, 1 wave/SIMD, 1 acc (dep)", ms, (double)g_blocks * 4 * ITERS * 1 * 2048.0);
    ms = time_ms(st, [](hipStream_t s) { hipLaunchKernelGGL(k_mfma_rate<4>, dim3(g_blocks * 2), dim3(256), 0, s, g_out, ITERS, 1.0); });
    report("mfma f64 16x16x4, 2 waves/SIMD, 4 acc", ms, (double)g_blocks * 2 * 4 * ITERS * 4 * 2048.0);
    ms = time_ms(st, [](hipStream_t s) { hipLaunchKernelGGL(k_mfma_rate<16>, dim3(g_blocks), dim3(256), 0, s, g_out, ITERS / 4, 1.0); });
    report("mfma f64 16x16x4, 1 wave/SIMD, 16 acc", ms, (double)g_blocks * 4 * (ITERS / 4) * 16 * 2048.0);
  }
  // ---- 3. VALU rate ----
  {
    double ms = time_ms(st, [](hipStream_t s) { hipLaunchKernelGGL(k_valu_rate<16>, dim3(g_blocks), dim3(256), 0, s, g_out, ITERS, 1.0); });
    report("v_fma_f64, 1 wave/SIMD, 16 acc", ms, (double)g_blocks * 256 * ITERS * 16 * 2.0);
    ms = time_ms(st, [](hipStream_t s) { hipLaunchKernelGGL(k_valu_rate<16>, dim3(g_blocks * 2), dim3(256), 0, s, g_out, ITERS, 1.0); });
    report("v_fma_f64, 2 waves/SIMD, 16 acc", ms, (double)g_blocks * 2 * 256 * ITERS * 16 * 2.0);
    ms = time_ms(st, [](hipStream_t s) { hipLaunchKernelGGL(k_valu_rate<16>, dim3(g_blocks * 4), dim3(256), 0, s, g_out, ITERS, 1.0); });
    report("v_fma_f64, 4 waves/SIMD, 16 acc", ms, (double)g_blocks * 4 * 256 * ITERS * 16 * 2.0);
  }
  // ---- 4. concurrency ----
  {
    double ms = time_ms(st, [](hipStream_t s) { hipLaunchKernelGGL((k_mixed_rate<4, 16>), dim3(g_blocks), dim3(256), 0, s, g_out, ITERS, 1.0); });
    report("mixed in-wave: 4 mfma + 16 fma / iter", ms, (double)g_blocks * 4 * ITERS * (4 * 2048.0 + 16 * 128.0));
    ms = time_ms(st, [](hipStream_t s) { hipLaunchKernelGGL((k_mixed_rate<4, 32>), dim3(g_blocks), dim3(256), 0, s, g_out, ITERS, 1.0); });
    report("mixed in-wave: 4 mfma + 32 fma / iter", ms, (double)g_blocks * 4 * ITERS * (4 * 2048.0 + 32 * 128.0));
    ms = time_ms(st, [](hipStream_t s) { hipLaunchKernelGGL((k_mixed_rate<4, 64>), dim3(g_blocks), dim3(256), 0, s, g_out, ITERS, 1.0); });
    report("mixed in-wave: 4 mfma + 64 fma / iter", ms, (double)g_blocks * 4 * ITERS * (4 * 2048.0 + 64 * 128.0));
    ms = time_ms(st, [](hipStream_t s) { hipLaunchKernelGGL(k_split_rate, dim3(g_blocks), dim3(512), 0, s, g_out, ITERS, 1.0); });
    report("split waves: 4 mfma-waves + 4 valu-waves / CU", ms, (double)g_blocks * ITERS * (4 * 4 * 2048.0 + 4 * 64 * 128.0));
  }
  return 0;
}
