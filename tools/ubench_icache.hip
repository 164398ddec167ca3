// Instruction-fetch cost of straight-line code on gfx950: the same number of v_fma_f64 (8-byte encoding,
// 16 independent accumulators) executed (a) as one long unrolled stream of KB kilobytes run once and
// (b) as a small loop that stays in the instruction cache.  One wave on the chip, then one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
#define R1024(x) R4(R256(x))
#define FMA16 \
  asm volatile("v_fma_f64 %0, %0, %16, %17\n\tv_fma_f64 %1, %1, %16, %17\n\tv_fma_f64 %2, %2, %16, %17\n\tv_fma_f64 %3, %3, %16, %17\n\t" \
               "v_fma_f64 %4, %4, %16, %17\n\tv_fma_f64 %5, %5, %16, %17\n\tv_fma_f64 %6, %6, %16, %17\n\tv_fma_f64 %7, %7, %16, %17\n\t" \
               "v_fma_f64 %8, %8, %16, %17\n\tv_fma_f64 %9, %9, %16, %17\n\tv_fma_f64 %10, %10, %16, %17\n\tv_fma_f64 %11, %11, %16, %17\n\t" \
               "v_fma_f64 %12, %12, %16, %17\n\tv_fma_f64 %13, %13, %16, %17\n\tv_fma_f64 %14, %14, %16, %17\n\tv_fma_f64 %15, %15, %16, %17" \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), \
                 "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b), "v"(c));
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
// 1024 x 16 fma = 16384 instructions = 128 KB of code, run once
__global__ void straight(double* out, unsigned long long* cyc) {
  double a[16]; for (int i = 0; i < 16; ++i) a[i] = 1.0 + i + threadIdx.x * 1e-9;
  double b = 1.0000001, c = 1e-9;
  unsigned long long t0 = now();
  R1024(FMA16)
  unsigned long long t1 = now();
  double s = 0; for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// same work as a loop of 64 iterations x 16 x 16 fma (2 KB body)
__global__ void looped(double* out, unsigned long long* cyc, int iters) {
  double a[16]; for (int i = 0; i < 16; ++i) a[i] = 1.0 + i + threadIdx.x * 1e-9;
  double b = 1.0000001, c = 1e-9;
  unsigned long long t0 = now();
  for (int it = 0; it < iters; ++it) { R16(FMA16) }
  unsigned long long t1 = now();
  double s = 0; for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  const int maxb = 4096;
  hipMalloc(&out, (size_t)maxb * 256 * 8); hipMalloc(&cyc, maxb * 8);
  std::vector<unsigned long long> h(maxb);
  auto report = [&](const char* name, int blocks) {
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.begin() + blocks);
    printf("%-58s median %7.2f cycles/instr\n", name, (double)h[blocks / 2] / 16384.0);
  };
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(straight, dim3(1), dim3(64), 0, 0, out, cyc); hipDeviceSynchronize();
    if (rep) report("straight 128 KB, 1 wave on the chip", 1);
    hipLaunchKernelGGL(looped, dim3(1), dim3(64), 0, 0, out, cyc, 64); hipDeviceSynchronize();
    if (rep) report("loop 2 KB body, 1 wave on the chip", 1);
    hipLaunchKernelGGL(straight, dim3(1024), dim3(64), 0, 0, out, cyc); hipDeviceSynchronize();
    if (rep) report("straight 128 KB, 1 wave per SIMD", 1024);
    hipLaunchKernelGGL(looped, dim3(1024), dim3(64), 0, 0, out, cyc, 64); hipDeviceSynchronize();
    if (rep) report("loop 2 KB body, 1 wave per SIMD", 1024);
    hipLaunchKernelGGL(straight, dim3(1024), dim3(256), 0, 0, out, cyc); hipDeviceSynchronize();
    if (rep) report("straight 128 KB, 4 waves per SIMD (256-thread WGs)", 1024);
    hipLaunchKernelGGL(looped, dim3(1024), dim3(256), 0, 0, out, cyc, 64); hipDeviceSynchronize();
    if (rep) report("loop 2 KB body, 4 waves per SIMD", 1024);
  }
  return 0;
}
