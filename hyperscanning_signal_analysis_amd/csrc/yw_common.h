// Shared pieces of the two Yule-Walker solvers (yw_solve.hip: block LDL^T of the augmented matrix; yw_lwr.hip: block
// Levinson-Whittle recursion on the p + 1 lag blocks): tile geometry and the cooperative inverse of an SPD tile.
#pragma once
#include "hmv_common.h"
#include "hmv_kernels.h"

namespace hmv {

typedef double f64x2 __attribute__((ext_vector_type(2)));

#define HMV_WAVE_SYNC()                                     \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                        \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
  } while (0)

__host__ __device__ inline long long yw_tri(int a, int b) { return (long long)a * (a + 1) / 2 + b; }
// scratch tiles per window: the LDL^T form needs 2 T(p+1) + 2p, the Levinson-Whittle form 4p + 6; one more tile at the
// end belongs to neither (its last int is the window's "re-solve with the LDL^T" flag: yw_guard_ptr)
__host__ __device__ inline long long yw_ws_tiles_d(int p) {
  const long long a = 2 * yw_tri(p + 1, 0) + 2 * (long long)p, b = 4 * (long long)p + 6;
  return (a > b ? a : b) + 1;
}
__host__ __device__ inline int* yw_guard_ptr(double* ws, long long item, int p, int tile) {
  return reinterpret_cast<int*>(ws + (size_t)(item + 1) * yw_ws_tiles_d(p) * tile) - 1;
}

template <int NT>
struct YwCfg {
  static constexpr int MP = 16 * NT;
  static constexpr int S = (MP <= 38) ? 38 : 70;   // >= MP and = 6 (mod 32)
};

// Cooperative inverse of the symmetric positive definite MP x MP tile stored row-major (stride S) in LDS
// `Xs`, by the whole workgroup (same scheme as K3, real arithmetic, no pivoting): wave w < NT holds columns
// 16w..16w+15 in the D layout; the wave that owns the 4 panel columns of block step s factors them in a
// lane-per-row layout (pivots by v_readlane, Newton reciprocal), publishes N = M'[:, S] through LDS, and
// after one barrier every wave runs its 4*NT MFMAs of the rank-4 update.  Must be called by all 256 threads.
// dinv_out (global, may be null): the inverse;  logdet_out (global, may be null): log det of the tile;
// pm (LDS, 8 doubles, may be null): on return pm[0] = smallest and pm[1] = largest pivot met (the conditioning guard of
// the Levinson-Whittle solver), valid for every thread after the call (it ends with a barrier then).
template <int NT, int S>
__device__ __forceinline__ void spd_inverse_coop(const double* Xs, double* Pb, double* Nb2, int* s_info,
                                                 double* s_ld, double* dinv_out, double* logdet_out, int info_base,
                                                 double* pm = nullptr) {
  constexpr int MP = 16 * NT, NI = 4 * NT, NSTEP = MP / 4;
  const int l = lane_id();
  const int w = uni(threadIdx.x >> 6);
  const int i = l >> 4, cc = l & 15;
  const bool active = (w < NT);
  double m[NI];
  if (active) {
#pragma unroll
    for (int I = 0; I < NI; ++I) m[I] = Xs[(4 * I + i) * S + 16 * w + cc];
  }
  double mypiv = 1.0;
  double pmin = 1.7976931348623157e308, pmax = 0.0;      // wave-uniform, over the columns this wave factors
  static_for<NSTEP>([&](auto sc) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
    constexpr int ws = s >> 2, q = s & 3;
    double* Nb = Nb2 + (s & 1) * MP * 4;
    if (w == ws) {
      if ((cc >> 2) == q) {
#pragma unroll
        for (int I = 0; I < NI; ++I) Pb[(4 * I + i) * 4 + (cc & 3)] = m[I];
      }
      HMV_WAVE_SYNC();
      double x[4];
      {
        const int r = (l < MP) ? l : 0;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) x[jj] = Pb[r * 4 + jj];
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int col = 4 * s + jj;
        const double piv = readlane_f64(x[jj], col);
        if (!(piv > 0.0) && l == 0 && *s_info == 0) *s_info = info_base + col + 1;
        mypiv = (l == col) ? piv : mypiv;
        pmin = fmin(pmin, piv);
        pmax = fmax(pmax, piv);
        double inv = __builtin_amdgcn_rcp(piv);                  // v_rcp_f64 seed + 2 Newton steps
        inv = __builtin_fma(__builtin_fma(-piv, inv, 1.0), inv, inv);
        inv = __builtin_fma(__builtin_fma(-piv, inv, 1.0), inv, inv);
        double qv[4];
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) qv[j2] = (j2 == jj) ? inv : readlane_f64(x[j2], col) * inv;
        const double f = x[jj];
        const bool isp = (l == col);
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
          const double base = (j2 == jj) ? 0.0 : x[j2];
          const double nr = __builtin_fma(-f, qv[j2], base);
          x[j2] = isp ? qv[j2] : nr;
        }
      }
      if (l < MP) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) Nb[l * 4 + jj] = x[jj];
      }
    }
    __syncthreads();
    if (active) {
      const double u = m[s];
#pragma unroll
      for (int I = 0; I < NI; ++I) {
        double nv = Nb[(4 * I + (l & 3)) * 4 + (l >> 4)];
        if (I == s) nv -= ((l & 3) == (l >> 4)) ? 1.0 : 0.0;
        m[I] = mfma4(nv, u, m[I]);
      }
      if (w == ws && (cc >> 2) == q) {
#pragma unroll
        for (int I = 0; I < NI; ++I) m[I] = Nb[(4 * I + i) * 4 + (cc & 3)];
      }
    }
  });
  if (dinv_out && active) {
#pragma unroll
    for (int I = 0; I < NI; ++I) dinv_out[(size_t)(4 * I + i) * MP + 16 * w + cc] = m[I];
  }
  if (logdet_out) {          // pivots live on lane `col` of the wave that factored column `col`
    double v = row16_sum_dpp(log(mypiv));
    v = readlane_f64(v, 0) + readlane_f64(v, 16) + readlane_f64(v, 32) + readlane_f64(v, 48);
    if (l == 0) s_ld[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) *logdet_out = ((s_ld[0] + s_ld[1]) + s_ld[2]) + s_ld[3];
  }
  if (pm) {
    __syncthreads();                         // s_ld / pm may still be read from the previous call
    if (l == 0 && w < NT) {
      pm[2 + 2 * w] = pmin;
      pm[3 + 2 * w] = pmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double a = pm[2], b = pm[3];
      for (int k = 1; k < NT; ++k) {
        a = fmin(a, pm[2 + 2 * k]);
        b = fmax(b, pm[3 + 2 * k]);
      }
      pm[0] = a;
      pm[1] = b;
    }
    __syncthreads();
  }
}

}  // namespace hmv
