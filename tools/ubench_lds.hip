// LDS pipe throughput on gfx950: cycles the CU's LDS unit needs per instruction, by kind.  16 waves per CU
// (four 256-thread workgroups per CU would be the K3 configuration) issue N independent instructions each;
// cycles = (wall cycles of the slowest wave) / (instructions issued by all waves of that CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
template <int KIND>
__global__ void __launch_bounds__(1024) k(int* out, unsigned long long* cyc) {
  __shared__ v4i buf[1024 + 64];
  const int tid = threadIdx.x, l = tid & 63;
  buf[tid] = v4i{tid, 1, 2, 3};
  __syncthreads();
  v4i v = {l, l + 1, l + 2, l + 3};
  int a = 0;
  const int addr_lin = ((tid & 0x3c0) + l) * 16;          // conflict-free 16 B per lane
  const int addr_bc = (tid & 0x3c0) * 16;                 // every lane of the wave reads the same 16 B
  const int addr_b64 = ((tid & 0x3c0) + l) * 8;
  const int bp = ((l & 0x33) | 4) * 4;
  __syncthreads();
  const unsigned long long t0 = now();
  if (KIND == 0) { R64(asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr_lin) : "memory");) }
  if (KIND == 1) { R64(asm volatile("ds_write_b128 %1, %0" : : "v"(v), "v"(addr_lin) : "memory");) }
  if (KIND == 2) { R64(asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(a) : "v"(bp), "v"(l) : "memory");) }
  if (KIND == 3) { R64(asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr_bc) : "memory");) }
  if (KIND == 4) { R64(asm volatile("ds_read_b64 %0, %1" : "=v"(*(long long*)&v) : "v"(addr_b64) : "memory");) }
  if (KIND == 5) { R64(asm volatile("ds_read_b32 %0, %1" : "=v"(a) : "v"(addr_lin >> 2) : "memory");) }
  if (KIND == 7) { R64(asm volatile("ds_swizzle_b32 %0, %1 offset:0x93" : "=v"(a) : "v"(l) : "memory");) }   // and 0x13, or 0x4
  if (KIND == 8) { R64(asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(a) : "v"(((l + 1) & 63) * 4), "v"(l) : "memory");) }
  if (KIND == 6) {   // single-lane write (exec = 1 lane), as the pivot-row publish
    if (l == 5) { R64(asm volatile("ds_write_b128 %1, %0" : : "v"(v), "v"(addr_lin) : "memory");) }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = now();
  out[blockIdx.x * 1024 + tid] = v.x + a;
  if (l == 0) cyc[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
}
template <int KIND>
static void run(const char* name, int* out, unsigned long long* cyc) {
  const int nb = 256;
  std::vector<unsigned long long> h(nb * 16);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<KIND>, dim3(nb), dim3(1024), 0, 0, out, cyc);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), cyc, nb * 16 * 8, hipMemcpyDeviceToHost);
  std::vector<double> per;
  for (int b = 0; b < nb; ++b) per.push_back((double)*std::max_element(h.begin() + b * 16, h.begin() + b * 16 + 16) / (16.0 * 64.0));
  std::sort(per.begin(), per.end());
  printf("%-52s %6.2f cycles per instruction (CU LDS pipe)\n", name, per[nb / 2]);
}
int main() {
  int* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
  run<0>("ds_read_b128, 16 B/lane linear", out, cyc);
  run<1>("ds_write_b128, 16 B/lane linear", out, cyc);
  run<2>("ds_bpermute_b32", out, cyc);
  run<3>("ds_read_b128, whole wave reads one address", out, cyc);
  run<4>("ds_read_b64, 8 B/lane linear", out, cyc);
  run<5>("ds_read_b32, 4 B/lane linear", out, cyc);
  run<6>("ds_write_b128, one active lane", out, cyc);
  run<7>("ds_swizzle_b32 (and 0x13 | 0x4: quad broadcast in a row)", out, cyc);
  run<8>("ds_bpermute_b32, rotation by one lane (no sharing)", out, cyc);
  return 0;
}
