// Dependent-chain latencies on gfx950, one wave on the chip: cycles per step of a serial chain of each
// instruction kind that appears in K3's panel factorisation (s_memtime around N unrolled repetitions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define N 64
#define REP(x) x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__global__ void k(double* out, unsigned long long* cyc, double seed) {
  __shared__ double lds[256];
  const int l = threadIdx.x;
  double a = seed + l * 1e-9, b = 1.0000001, c = 1e-9;
  unsigned long long t0, t1;
  int idx = 0;
  // 0: dependent v_fma_f64
  t0 = now(); REP(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) t1 = now(); cyc[idx++] = t1 - t0;
  // 1: dependent v_mul_f64
  t0 = now(); REP(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));) t1 = now(); cyc[idx++] = t1 - t0;
  // 2: dependent v_rcp_f64
  t0 = now(); REP(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));) t1 = now(); cyc[idx++] = t1 - t0;
  // 3: dependent v_cndmask_b32 pair (64-bit select)
  { int lo = __double2loint(a), hi = __double2hiint(a);
    t0 = now(); REP(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %1, %1, %0, vcc" : "+v"(lo), "+v"(hi) :: );) t1 = now(); cyc[idx++] = t1 - t0;
    a += lo * 1e-300 + hi * 1e-300; }
  // 4: v_readlane_b32 -> v_add using the SGPR (dependent through SGPR)
  { int v = l;
    t0 = now(); REP(asm volatile("v_readlane_b32 s20, %0, 3\n\tv_add_u32 %0, s20, %0" : "+v"(v) :: "s20");) t1 = now(); cyc[idx++] = t1 - t0;
    a += v * 1e-300; }
  // 5: readlane of a double (2 readlanes) -> v_mul_f64 with the SGPR pair
  t0 = now(); REP(asm volatile("v_readlane_b32 s20, %0, 3\n\tv_readlane_b32 s21, %1, 3\n\tv_mul_f64 %2, s[20:21], %2" : : "v"(__double2loint(b)), "v"(__double2hiint(b)), "v"(a) : "s20", "s21");) t1 = now(); cyc[idx++] = t1 - t0;
  // 6: v_cmp -> s_cbranch_vccz (not taken) dependent on VALU compare
  { int v = l;
    t0 = now(); REP(asm volatile("v_cmp_lt_i32 vcc, -1, %0\n\ts_cbranch_vccz 1f\n\tv_add_u32 %0, 1, %0\n1:" : "+v"(v) :: "vcc");) t1 = now(); cyc[idx++] = t1 - t0;
    a += v * 1e-300; }
  // 7: exec-masked region: v_cmp + s_and_saveexec + ds_write_b128 + s_or exec, then ds_read_b128 + waitcnt + use
  { typedef int v4i __attribute__((ext_vector_type(4)));
    v4i v = {l, l + 1, l + 2, l + 3};
    t0 = now();
    REP(asm volatile("v_cmp_eq_u32 vcc, 5, %2\n\ts_and_saveexec_b64 s[20:21], vcc\n\tds_write_b128 %3, %0\n\ts_or_b64 exec, exec, s[20:21]\n\tds_read_b128 %0, %3\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f64 %1, %1, 1.0"
        : "+v"(v), "+v"(a) : "v"(l), "v"(0) : "vcc", "s20", "s21", "memory");)
    t1 = now(); cyc[idx++] = t1 - t0; a += v.x * 1e-300; }
  // 8: ds_write_b64 -> ds_read_b64 round trip (all lanes)
  t0 = now(); REP(asm volatile("ds_write_b64 %1, %0\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(l * 8) : "memory");) t1 = now(); cyc[idx++] = t1 - t0;
  // 9: dependent v_max_u32_dpp (with the two wait states)
  { unsigned v = l;
    t0 = now(); REP(asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v));) t1 = now(); cyc[idx++] = t1 - t0;
    a += v * 1e-300; }
  // 10: ds_bpermute_b32 round trip
  { int v = l;
    t0 = now(); REP(asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(((l + 1) & 63) * 4) : "memory");) t1 = now(); cyc[idx++] = t1 - t0;
    a += v * 1e-300; }
  // 11: dependent MFMA 4x4x4 f64 (same accumulator)
  t0 = now(); REP(asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));) t1 = now(); cyc[idx++] = t1 - t0;
  // 12: v_fma_f64 with 2 independent chains interleaved (ILP 2): cycles per PAIR
  { double a2 = a + 1.0;
    t0 = now(); REP(asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3" : "+v"(a), "+v"(a2) : "v"(b), "v"(c));) t1 = now(); cyc[idx++] = t1 - t0;
    a += a2 * 1e-300; }
  // 13: s_barrier with one wave (cost of the instruction itself)
  t0 = now(); REP(asm volatile("s_barrier" ::: "memory");) t1 = now(); cyc[idx++] = t1 - t0;
  // 14: empty (timer overhead)
  t0 = now(); t1 = now(); cyc[idx++] = t1 - t0;
  out[l] = a + lds[l & 1];
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 32 * 8);
  const char* names[] = {"v_fma_f64 dependent", "v_mul_f64 dependent", "v_rcp_f64 dependent", "v_cndmask_b32 x2 dependent",
                         "v_readlane -> VALU (via SGPR)", "2x v_readlane -> v_mul_f64 (SGPR pair)", "v_cmp -> s_cbranch_vccz -> VALU",
                         "masked ds_write_b128 -> ds_read_b128 -> use", "ds_write_b64 -> ds_read_b64 round trip",
                         "s_nop 1 + v_max_u32_dpp dependent", "ds_bpermute_b32 round trip", "mfma 4x4x4 f64 dependent",
                         "v_fma_f64 two chains (per pair)", "s_barrier (1 wave)", "timer overhead (total)"};
  std::vector<unsigned long long> best(15, ~0ull);
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    unsigned long long h[32];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 15; ++i) best[i] = std::min(best[i], h[i]);
  }
  for (int i = 0; i < 15; ++i)
    printf("%-48s %8.1f cycles/step\n", names[i], i == 14 ? (double)best[i] : (double)(best[i] - best[14]) / N);
  return 0;
}
