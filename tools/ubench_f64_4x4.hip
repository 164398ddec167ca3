// Probes v_mfma_f64_4x4x4_4b_f64 on gfx950: (1) lane maps of A, B and D found by one-hot probing,
// (2) cycles per instruction for 1/2/4 waves per SIMD, (3) whether VALU f64 FMAs issue beside it
// (same wave and partner wave).  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// block (la, lb): a = 1 on lane la only, b = 1 on lane lb only; record D per lane.
__global__ void k_probe(double* out /* [64*64][64] */) {
  int la = blockIdx.x >> 6, lb = blockIdx.x & 63, l = threadIdx.x;
  double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  out[blockIdx.x * 64 + l] = d;
}

struct Stamp { unsigned long long c0, c1, r0, r1; };

// MODE 0: NM mfma4x4 per iter; MODE 1: NM mfma + NV fma interleaved in the same wave;
// MODE 2: even waves mfma, odd waves valu (NV fma per NM mfma-equivalents)
template <int NM, int NV, int MODE>
__global__ void __launch_bounds__(512) k_rate(Stamp* st, double* out, int iters, double seed) {
  double acc[NM];
  double v[NV > 0 ? NV : 1];
  for (int i = 0; i < NM; ++i) acc[i] = 0.0;
  for (int i = 0; i < NV; ++i) v[i] = threadIdx.x * 1e-6 + i;
  double a = seed + threadIdx.x * 1e-3, b = seed * 0.5 + threadIdx.x * 1e-4;
  double fa = 1.0 + seed * 1e-9, fb = seed * 1e-7;
  int wave = threadIdx.x >> 6;
  bool do_m = (MODE != 2) || ((wave & 4) == 0);
  bool do_v = (MODE == 1) || (MODE == 2 && (wave & 4) != 0);
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  if (MODE == 2) {
    if (do_m) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
      }
    } else {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = __builtin_fma(v[i], fa, fb);
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
        if (MODE == 1) {
#pragma unroll
          for (int j = 0; j < NV / NM; ++j) v[i * (NV / NM) + j] = __builtin_fma(v[i * (NV / NM) + j], fa, fb);
        }
      }
    }
  }
  double s = 0;
  for (int i = 0; i < NM; ++i) s += acc[i];
  for (int i = 0; i < NV; ++i) s += v[i];
  asm volatile("" :: "v"(s));
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    Stamp x = {c0, c1, r0, r1};
    st[blockIdx.x * (blockDim.x / 64) + wave] = x;
  }
  (void)do_v;
}

template <int NM, int NV, int MODE>
static void run(const char* name, int blocks, int threads, int iters) {
  Stamp* dst; double* dout;
  int wpb = threads / 64, nw = blocks * wpb;
  CHECK(hipMalloc(&dst, sizeof(Stamp) * nw)); CHECK(hipMalloc(&dout, sizeof(double) * blocks * threads));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_rate<NM, NV, MODE>), dim3(blocks), dim3(threads), 0, 0, dst, dout, iters, 1.0);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_rate<NM, NV, MODE>), dim3(blocks), dim3(threads), 0, 0, dst, dout, iters, 1.0);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<Stamp> h(nw);
  CHECK(hipMemcpy(h.data(), dst, sizeof(Stamp) * nw, hipMemcpyDeviceToHost));
  // per-wave cycles per loop iteration, split by role for MODE 2
  std::vector<double> cm, cv, clk;
  for (int i = 0; i < nw; ++i) {
    double c = double(h[i].c1 - h[i].c0) / iters;
    bool is_v = (MODE == 2) && (((i % wpb) & 4) != 0);
    (is_v ? cv : cm).push_back(c);
    clk.push_back(double(h[i].c1 - h[i].c0) / double(h[i].r1 - h[i].r0) * 100.0);
  }
  std::sort(cm.begin(), cm.end()); std::sort(cv.begin(), cv.end()); std::sort(clk.begin(), clk.end());
  double flop = 0;
  if (MODE == 2) flop = (double)blocks * iters * ((wpb / 2) * NM * 512.0 + (wpb / 2) * NV * 128.0);
  else flop = (double)nw * iters * (NM * 512.0 + (MODE == 1 ? NV * 128.0 : 0.0));
  printf("%-58s cyc/iter mfma-wave %8.1f", name, cm[cm.size() / 2]);
  if (!cv.empty()) printf("  valu-wave %8.1f", cv[cv.size() / 2]);
  printf("  clock %6.0f MHz  wall %7.3f ms  %7.2f TFLOP/s\n", clk[clk.size() / 2], ms, flop / (ms * 1e-3) / 1e12);
  CHECK(hipFree(dst)); CHECK(hipFree(dout));
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int ncu = prop.multiProcessorCount;
  // ---- lane maps ----
  {
    double* d; CHECK(hipMalloc(&d, sizeof(double) * 4096 * 64));
    hipLaunchKernelGGL(k_probe, dim3(4096), dim3(64), 0, 0, d);
    CHECK(hipDeviceSynchronize());
    std::vector<double> h(4096 * 64);
    CHECK(hipMemcpy(h.data(), d, sizeof(double) * 4096 * 64, hipMemcpyDeviceToHost));
    // For each (la, lb) list the D lanes that are non-zero.
    printf("probe: for A one-hot lane la and B one-hot lane lb, D lanes that become 1\n");
    for (int la = 0; la < 64; la += 1) {
      int cnt = 0;
      for (int lb = 0; lb < 64; ++lb) for (int l = 0; l < 64; ++l) if (h[(la * 64 + lb) * 64 + l] != 0.0) cnt++;
      if (la < 20 || la % 16 == 0) {
        printf("  la=%2d hits=%d :", la, cnt);
        for (int lb = 0; lb < 64; ++lb) for (int l = 0; l < 64; ++l) if (h[(la * 64 + lb) * 64 + l] != 0.0) printf(" (lb=%d->d=%d)", lb, l);
        printf("\n");
      }
    }
    // test hypothesis: block = lane>>4 ; A[i][k]: i = lane&3, k = (lane>>2)&3 ; B[k][j]: j = lane&3, k = (lane>>2)&3 ; D[i][j]: j = lane&3, i = (lane>>2)&3
    int bad1 = 0, bad2 = 0;
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) for (int l = 0; l < 64; ++l) {
      double got = h[(la * 64 + lb) * 64 + l];
      int ba = la >> 4, ia = la & 3, ka = (la >> 2) & 3;
      int bb = lb >> 4, jb = lb & 3, kb = (lb >> 2) & 3;
      int bd = l >> 4;
      // hyp 1: D lane: j = l&3, i = (l>>2)&3
      double e1 = (ba == bb && ba == bd && ka == kb && (l & 3) == jb && ((l >> 2) & 3) == ia) ? 1.0 : 0.0;
      // hyp 2: D lane: i = l&3, j = (l>>2)&3
      double e2 = (ba == bb && ba == bd && ka == kb && (l & 3) == ia && ((l >> 2) & 3) == jb) ? 1.0 : 0.0;
      if (got != e1) bad1++;
      if (got != e2) bad2++;
    }
    printf("hypothesis A[i=l&3][k=(l>>2)&3], B[k=(l>>2)&3][j=l&3], block=l>>4:  D[i=(l>>2)&3][j=l&3] mismatches=%d ; D[i=l&3][j=(l>>2)&3] mismatches=%d\n", bad1, bad2);
    CHECK(hipFree(d));
  }
  const int IT = 4000;
  run<8, 0, 0>("mfma4x4x4 x8/iter, 1 wave on chip", 1, 64, IT);
  run<8, 0, 0>("mfma4x4x4 x8/iter, 1 wave/SIMD all CUs", ncu, 256, IT);
  run<8, 0, 0>("mfma4x4x4 x8/iter, 2 waves/SIMD all CUs", ncu, 512, IT);
  run<8, 0, 0>("mfma4x4x4 x8/iter, 4 waves/SIMD all CUs", ncu * 2, 512, IT);
  run<8, 8, 1>("in-wave 8 mfma4x4 + 8 fma, 1 wave/SIMD", ncu, 256, IT);
  run<8, 16, 1>("in-wave 8 mfma4x4 + 16 fma, 1 wave/SIMD", ncu, 256, IT);
  run<8, 32, 1>("in-wave 8 mfma4x4 + 32 fma, 1 wave/SIMD", ncu, 256, IT);
  run<8, 16, 1>("in-wave 8 mfma4x4 + 16 fma, 2 waves/SIMD", ncu, 512, IT);
  run<8, 32, 1>("in-wave 8 mfma4x4 + 32 fma, 2 waves/SIMD", ncu, 512, IT);
  run<8, 16, 2>("split: waves0-3 8 mfma4x4/iter, waves4-7 16 fma/iter", ncu, 512, IT);
  run<8, 32, 2>("split: waves0-3 8 mfma4x4/iter, waves4-7 32 fma/iter", ncu, 512, IT);
  return 0;
}
