// K3's panel factorisation (64 rows x 4 complex columns, one wave, lane per row) in isolation: cycles per
// pivot column for several ways of writing it.  One wave per CU (256 workgroups) so the chip runs at its
// loaded clock but nothing competes for the SIMD.  Build: hipcc -O3 --offload-arch=gfx950 -std=c++17
//   -I hyperscanning_signal_analysis_amd/csrc tools/ubench_panel.hip -o tools/ubench_panel
#include "hmv_common.h"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace hmv;

#define FENCE()                                             \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                        \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
  } while (0)

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

__device__ __forceinline__ void reciprocal(double pr, double pi, double& dd, double& ivr, double& ivi) {
  dd = __builtin_fma(pr, pr, pi * pi);
  double y = __builtin_amdgcn_rcp(dd);
  y = __builtin_fma(__builtin_fma(-dd, y, 1.0), y, y);
  const double e = __builtin_fma(-dd, y, 1.0), tr = pr * y, ti = -pi * y;
  ivr = __builtin_fma(tr, e, tr);
  ivi = __builtin_fma(ti, e, ti);
}

template <int JJ>
__device__ __forceinline__ void eliminate(double2 (&x)[4], int l, int col, const double (&pvr)[4],
                                          const double (&pvi)[4], double ivr, double ivi) {
  const bool isp = (l == col);
  const double fr = x[JJ].x, fi = x[JJ].y;
  double mr = __builtin_fma(fi, ivi, -(fr * ivr));
  double mi = __builtin_fma(-fr, ivi, -(fi * ivr));
  mr = isp ? ivr : mr;
  mi = isp ? ivi : mi;
  const double keep = isp ? 0.0 : 1.0;
#pragma unroll
  for (int j2 = 0; j2 < 4; ++j2) {
    if (j2 == JJ) continue;
    const double br = x[j2].x * keep, bi = x[j2].y * keep;
    x[j2].x = __builtin_fma(-mi, pvi[j2], __builtin_fma(mr, pvr[j2], br));
    x[j2].y = __builtin_fma(mi, pvr[j2], __builtin_fma(mr, pvi[j2], bi));
  }
  x[JJ].x = mr;
  x[JJ].y = mi;
}

// VARIANT 0: the production code at commit "speculative diagonal reciprocal, single interchange branch"
// VARIANT 1: straight line, diagonal pivots, pivot row by 16 v_readlane, flag instead of branch
// VARIANT 2: straight line, diagonal pivots, pivot row through LDS (exec-masked write, broadcast read)
// VARIANT 3: as 2 but the pivot row by ds_bpermute (no exec change, no LDS memory)
// VARIANT 4: as 2, no search flag at all (lower bound of the LDS form)
// VARIANT 5: as 2 but the lane writes only the three other columns and reads them back as 3 x b128
template <int VARIANT>
__global__ void __launch_bounds__(64) panel_kernel(double* out, unsigned long long* cyc, double tau) {
  __shared__ double2 Srow[8];
  __shared__ double2 Nbuf[64 * 4];
  __shared__ int s_orig[64];
  __shared__ int s_swp[4];
  const int l = lane_id();
  s_orig[l] = l;
  unsigned long long total = 0;
  double acc = 0.0;
  static_for<16>([&](auto tc) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
    double2 x[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      x[c].x = 0.01 * ((l * 7 + c * 3 + t) % 11) + ((l == 4 * t + c) ? 4.0 : 0.0) + acc * 1e-30;
      x[c].y = 0.02 * ((l * 5 + c + 2 * t) % 7) - 0.05;
    }
    FENCE();
    const unsigned long long t0 = now();
    int rs[4], bad = 0;
    unsigned long long need = 0;
    static_for<4>([&](auto jjc) __attribute__((always_inline)) {
      constexpr int jj = decltype(jjc)::value;
      constexpr int col = 4 * t + jj;
      const bool valid = (l >= col);
      const double cand = __builtin_fabs(x[jj].x) + __builtin_fabs(x[jj].y);
      double dd, ivr, ivi;
      double pvr[4], pvi[4];
      if constexpr (VARIANT == 0) {
        double pr = readlane_f64(x[jj].x, col), pi = readlane_f64(x[jj].y, col);
        const double dc = __builtin_fabs(pr) + __builtin_fabs(pi);
        const bool need_search = __builtin_amdgcn_ballot_w64(valid && (tau * cand > dc)) != 0ull;
        reciprocal(pr, pi, dd, ivr, ivi);
        int rstar = col;
        if (__builtin_expect(need_search, 0)) {
          const unsigned key = valid ? __float_as_uint((float)cand) : 0u;
          const unsigned kmax = wave_max_u32(key);
          rstar = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(valid && key == kmax));
          if (rstar != col) {
            pr = readlane_f64(x[jj].x, rstar);
            pi = readlane_f64(x[jj].y, rstar);
            reciprocal(pr, pi, dd, ivr, ivi);
            if (l == col) {
#pragma unroll
              for (int j2 = 0; j2 < 4; ++j2) Srow[4 + j2] = x[j2];
            }
            if (l == 0) {
              const int oc = s_orig[col], orr = s_orig[rstar];
              s_orig[col] = orr;
              s_orig[rstar] = oc;
            }
          }
        }
        rs[jj] = rstar;
        if (l == rstar) {
#pragma unroll
          for (int j2 = 0; j2 < 4; ++j2) Srow[j2] = x[j2];
        }
        FENCE();
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
          const double2 v = Srow[j2];
          pvr[j2] = v.x;
          pvi[j2] = v.y;
        }
        if (__builtin_expect(rstar != col, 0)) {
#pragma unroll
          for (int j2 = 0; j2 < 4; ++j2) {
            const double2 cv = Srow[4 + j2];
            x[j2].x = (l == rstar) ? cv.x : x[j2].x;
            x[j2].y = (l == rstar) ? cv.y : x[j2].y;
          }
        }
        FENCE();
      } else if constexpr (VARIANT == 1) {
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
          pvr[j2] = readlane_f64(x[j2].x, col);
          pvi[j2] = readlane_f64(x[j2].y, col);
        }
        const double dc = __builtin_fabs(pvr[jj]) + __builtin_fabs(pvi[jj]);
        need |= __builtin_amdgcn_ballot_w64(valid && (tau * cand > dc));
        reciprocal(pvr[jj], pvi[jj], dd, ivr, ivi);
        rs[jj] = col;
      } else {
        const double pr = readlane_f64(x[jj].x, col), pi = readlane_f64(x[jj].y, col);
        if constexpr (VARIANT != 4) {
          const double dc = __builtin_fabs(pr) + __builtin_fabs(pi);
          need |= __builtin_amdgcn_ballot_w64(valid && (tau * cand > dc));
        }
        reciprocal(pr, pi, dd, ivr, ivi);
        rs[jj] = col;
        if constexpr (VARIANT == 3) {
#pragma unroll
          for (int j2 = 0; j2 < 4; ++j2) {
            pvr[j2] = (j2 == jj) ? pr : shfl_f64(x[j2].x, col);
            pvi[j2] = (j2 == jj) ? pi : shfl_f64(x[j2].y, col);
          }
        } else if constexpr (VARIANT == 5) {
          if (l == col) {
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2)
              if (j2 != jj) Srow[j2] = x[j2];
          }
          FENCE();
#pragma unroll
          for (int j2 = 0; j2 < 4; ++j2) {
            if (j2 == jj) { pvr[j2] = pr; pvi[j2] = pi; continue; }
            const double2 v = Srow[j2];
            pvr[j2] = v.x;
            pvi[j2] = v.y;
          }
          FENCE();
        } else {
          if (l == col) {
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) Srow[j2] = x[j2];
          }
          FENCE();
#pragma unroll
          for (int j2 = 0; j2 < 4; ++j2) {
            const double2 v = Srow[j2];
            pvr[j2] = v.x;
            pvi[j2] = v.y;
          }
          FENCE();
        }
      }
      if (!(dd > 0.0) && bad == 0) bad = col + 1;
      eliminate<jj>(x, l, col, pvr, pvi, ivr, ivi);
    });
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) Nbuf[l * 4 + jj] = x[jj];
    if (l == 0) {
      *reinterpret_cast<int4*>(&s_swp[0]) = make_int4(rs[0], rs[1], rs[2], rs[3]);
      if (bad != 0 || need != 0) s_orig[1] = bad + (int)need;
    }
    const unsigned long long t1 = now();
    total += t1 - t0;
    FENCE();
    acc += Nbuf[((l + 1) & 63) * 4].x;
  });
  out[blockIdx.x * 64 + l] = acc + s_orig[l] + s_swp[l & 3];
  if (l == 0) cyc[blockIdx.x] = total;
}

template <int V>
static void run(const char* name, double* out, unsigned long long* cyc) {
  const int nb = 256;
  std::vector<unsigned long long> h(nb);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(panel_kernel<V>, dim3(nb), dim3(64), 0, 0, out, cyc, 1.0);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-78s %7.1f cycles per pivot column\n", name, (double)h[nb / 2] / 64.0);
}

int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 64 * 8); hipMalloc(&cyc, 256 * 8);
  run<0>("0 production: diag test + branch, pivot row via exec-masked LDS", out, cyc);
  run<1>("1 straight line: 16 v_readlane per column, flag", out, cyc);
  run<2>("2 straight line: pivot row via exec-masked LDS write + broadcast read, flag", out, cyc);
  run<3>("3 straight line: pivot row via 12 ds_bpermute, flag", out, cyc);
  run<4>("4 as 2 without the search flag", out, cyc);
  run<5>("5 as 2, only the three other columns through LDS", out, cyc);
  return 0;
}
