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

// The same inverse for TWO tiles at once (the Levinson-Whittle recursion needs Vb_q^-1 and Vf_q^-1 at the head of
// every order, and they are independent).  In spd_inverse_coop one wave factors a panel while the other three wait for
// it -- 2.4 k cycles per block step, 38 k per inverse, 29 % of a window's time (profiles/r03_k2_notes.md).  Here tile B's
// columns are dealt to the waves rotated by NT / 2, so the panels of step s of A and of B belong to DIFFERENT waves and
// are factored at the same time; one barrier per step serves both, and every wave then runs the rank-4 updates of
// its column blocks of both tiles (2 x 4 NT MFMAs).  ga / gb: the tiles in the D layout (row strips per wave); `img`: an
// LDS image buffer of MP x S doubles used for one tile after the other; Pb2 / Nb4: two panel and four N buffers.
// out_a / out_b (global): the inverses; logdet_b (may be null): log det of B; pm (LDS, >= 4 + 16 doubles): on return
// pm[0..1] = smallest / largest pivot of A, pm[2..3] of B.  Must be called by all 256 threads.
template <int NT, int S>
__device__ __forceinline__ void spd_inverse_coop2(const double (&ga)[NT][NT], const double (&gb)[NT][NT], double* img,
                                                  double* Pb2, double* Nb4, int* s_info, double* s_ld, double* out_a,
                                                  double* out_b, double* logdet_b, int info_base, double* pm) {
  constexpr int MP = 16 * NT, NI = 4 * NT, NSTEP = MP / 4, ROT = NT / 2;
  const int l = lane_id();
  const int w = uni(threadIdx.x >> 6);
  const int i = l >> 4, cc = l & 15;
  const bool active = (w < NT);
  const int wB = active ? (w + ROT) % NT : 0;          // the column block of B this wave holds
  double ma[NI], mb[NI];
  // ---- the two tiles through the one image buffer into the column-block layout
  __syncthreads();
#pragma unroll
  for (int ii = 0; ii < NT; ++ii)
#pragma unroll
    for (int J = 0; J < NT; ++J) img[(4 * (w * NT + ii) + i) * S + 16 * J + cc] = ga[ii][J];
  __syncthreads();
  if (active) {
#pragma unroll
    for (int I = 0; I < NI; ++I) ma[I] = img[(4 * I + i) * S + 16 * w + cc];
  }
  __syncthreads();
#pragma unroll
  for (int ii = 0; ii < NT; ++ii)
#pragma unroll
    for (int J = 0; J < NT; ++J) img[(4 * (w * NT + ii) + i) * S + 16 * J + cc] = gb[ii][J];
  __syncthreads();
  if (active) {
#pragma unroll
    for (int I = 0; I < NI; ++I) mb[I] = img[(4 * I + i) * S + 16 * wB + cc];
  }
  double mypiv_b = 1.0;
  double pmin_a = 1.7976931348623157e308, pmax_a = 0.0, pmin_b = 1.7976931348623157e308, pmax_b = 0.0;
  // panel of block step s, held by this wave in register group q of m[]: lane per row, four pivot columns, N -> LDS
  auto factor = [&](double (&m)[NI], auto sc, double* Pb, double* Nb, double& pmin, double& pmax, double* mypiv)
                    __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value, q = s & 3;
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
      if (mypiv) *mypiv = (l == col) ? piv : *mypiv;
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
  };
  auto update = [&](double (&m)[NI], auto sc, const double* Nb, bool owner) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value, q = s & 3;
    const double u = m[s];
#pragma unroll
    for (int I = 0; I < NI; ++I) {
      double nv = Nb[(4 * I + (l & 3)) * 4 + (l >> 4)];
      if (I == s) nv -= ((l & 3) == (l >> 4)) ? 1.0 : 0.0;
      m[I] = mfma4(nv, u, m[I]);
    }
    if (owner && (cc >> 2) == q) {
#pragma unroll
      for (int I = 0; I < NI; ++I) m[I] = Nb[(4 * I + i) * 4 + (cc & 3)];
    }
  };
  static_for<NSTEP>([&](auto sc) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
    constexpr int wa = s >> 2, wb = ((s >> 2) - ROT + NT) % NT;      // owners of panel s of A and of B
    double* NbA = Nb4 + (s & 1) * MP * 4;
    double* NbB = Nb4 + (2 + (s & 1)) * MP * 4;
    if (w == wa) factor(ma, sc, Pb2, NbA, pmin_a, pmax_a, nullptr);
    if (w == wb) factor(mb, sc, Pb2 + MP * 4, NbB, pmin_b, pmax_b, &mypiv_b);
    __syncthreads();
    if (active) {
      update(ma, sc, NbA, w == wa);
      update(mb, sc, NbB, w == wb);
    }
  });
  if (active) {
#pragma unroll
    for (int I = 0; I < NI; ++I) {
      out_a[(size_t)(4 * I + i) * MP + 16 * w + cc] = ma[I];
      out_b[(size_t)(4 * I + i) * MP + 16 * wB + cc] = mb[I];
    }
  }
  if (logdet_b) {            // B's pivots live on lane `col` of the wave that factored column `col`
    double v = row16_sum_dpp(log(mypiv_b));
    v = readlane_f64(v, 0) + readlane_f64(v, 16) + readlane_f64(v, 32) + readlane_f64(v, 48);
    // fixed order over B's column blocks 0 .. NT-1 (wave w factored block wB): the same sum as the one-tile routine's
    if (l == 0 && active) s_ld[wB] = v;
    if (l == 0 && !active) s_ld[w] = 0.0;
    __syncthreads();
    if (threadIdx.x == 0) *logdet_b = ((s_ld[0] + s_ld[1]) + s_ld[2]) + s_ld[3];
  }
  __syncthreads();                           // s_ld / pm may still be read from a previous call
  if (l == 0 && w < NT) {
    pm[4 + 4 * w] = pmin_a;
    pm[5 + 4 * w] = pmax_a;
    pm[6 + 4 * w] = pmin_b;
    pm[7 + 4 * w] = pmax_b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a0 = pm[4], a1 = pm[5], b0 = pm[6], b1 = pm[7];
    for (int k = 1; k < NT; ++k) {
      a0 = fmin(a0, pm[4 + 4 * k]);
      a1 = fmax(a1, pm[5 + 4 * k]);
      b0 = fmin(b0, pm[6 + 4 * k]);
      b1 = fmax(b1, pm[7 + 4 * k]);
    }
    pm[0] = a0;
    pm[1] = a1;
    pm[2] = b0;
    pm[3] = b1;
  }
  __syncthreads();
}

}  // namespace hmv
