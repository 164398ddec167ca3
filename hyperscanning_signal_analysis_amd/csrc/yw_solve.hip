// K2 -- Yule-Walker solve of every window: AR coefficients and residual covariance from R_0..R_p.
//
// Replaces `ar_coeff` (/root/reference/src/mtmvar.py:90-123): `np.linalg.solve(r_left, r_right)` on the
// (m p) x (m p) block-Toeplitz r_left, `variance = r_zero - x @ r_right`, reshape to (m, m, p).
//
// The normal-equation matrix is never materialised.  With G[a][b] = R_{a-b} (a >= b), R_{b-a}^T (a < b)
// and B[a] = R_{a+1}, the augmented symmetric matrix
//        Ghat = [ G    B  ]          (block row p = [ R_1^T ... R_p^T  R_0 ])
//               [ B^T  R_0]
// gets a block LDL^T factorisation with MP x MP tiles (left-looking, tile (a, b), b <= a):
//        Y[a][b]  = Ghat[a][b] - sum_{c<b} Lt[a][c] Y[b][c]^T
//        Lt[a][b] = Y[a][b] D_b^-1      (b < a),        D_b = Y[b][b]
// Then V = Y[p][p] is the residual covariance (= R_0 - B^T G^-1 B), the partial sums of Y[p][p] are the
// residual covariances of the lower orders (used by the model-order criterion), and the coefficients
// follow from a back substitution with the UNIT lower factor:  Z[b] = Lt[p][b] - sum_{c>b} Z[c] Lt[c][b],
// ar[:, :, b] = Z[b].  G is a Gram matrix of lagged data (biased estimator), i.e. symmetric positive
// (semi)definite, so no pivoting is needed; a non-positive pivot is reported through `info` and becomes
// numpy.linalg.LinAlgError("Singular matrix") in the Python layer, like the reference's dgesv failure.
//
// Mapping: one workgroup (4 waves) per window.  Every tile product is an MP x MP x MP real GEMM on
// v_mfma_f64_4x4x4_4b_f64 with both operands staged in LDS (row stride 6 mod 32 doubles: conflict-free
// A- and B-operand reads); wave w owns row blocks w*NT .. w*NT+NT-1 of the output tile.  The p tile
// inverses D_b^-1 run on wave 0 as an in-register blocked Gauss-Jordan (same scheme as K3, real, no pivot).
#include "hmv_common.h"
#include "hmv_kernels.h"

namespace hmv {

#define HMV_WAVE_SYNC()                                     \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                        \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
  } while (0)

__host__ __device__ inline long long yw_tri(int a, int b) { return (long long)a * (a + 1) / 2 + b; }
__host__ __device__ inline long long yw_ws_tiles_d(int p) { return 2 * yw_tri(p + 1, 0) + 2 * (long long)p; }
long long yw_ws_tiles(int p) { return yw_ws_tiles_d(p); }

template <int NT>
struct YwCfg {
  static constexpr int MP = 16 * NT;
  static constexpr int S = (MP <= 38) ? 38 : 70;   // >= MP and = 6 (mod 32)
};

// In-register inverse of a symmetric positive definite MP x MP tile held by ONE wave in the D layout.
template <int NT>
__device__ __forceinline__ void spd_inverse_wave(double (&t)[4 * NT][NT], double* Pb, double* Nb, int& info,
                                                 double& logdet, bool want_logdet) {
  constexpr int MP = 16 * NT, NI = 4 * NT, NJ = NT, NSTEP = MP / 4;
  const int l = lane_id();
  const int i = l >> 4, cc = l & 15;
  double mypiv = 1.0;   // lane c keeps pivot c; ONE log per lane after the sweep (log() inlined per column
                        // costs ~100 VGPRs and thousands of instructions)
  static_for<NSTEP>([&](auto sc) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
    constexpr int Js = s >> 2, q = s & 3;
    if ((cc >> 2) == q) {
#pragma unroll
      for (int I = 0; I < NI; ++I) Pb[(4 * I + i) * 4 + (cc & 3)] = t[I][Js];
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
      if (!(piv > 0.0) && info == 0) info = col + 1;
      mypiv = (l == col) ? piv : mypiv;
      const double inv = 1.0 / piv;
      double qv[4];
#pragma unroll
      for (int j2 = 0; j2 < 4; ++j2) qv[j2] = (j2 == jj) ? inv : readlane_f64(x[j2], col) * inv;
      const double f = x[jj];
      const bool isp = (l == col);
#pragma unroll
      for (int j2 = 0; j2 < 4; ++j2) {
        const double tr = f * qv[j2];
        const double nr = (j2 == jj) ? -tr : x[j2] - tr;
        x[j2] = isp ? qv[j2] : nr;
      }
    }
    if (l < MP) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) Nb[l * 4 + jj] = x[jj];
    }
    HMV_WAVE_SYNC();
    double u[NJ];
#pragma unroll
    for (int J = 0; J < NJ; ++J) u[J] = t[s][J];
#pragma unroll
    for (int I = 0; I < NI; ++I) {
      double nv = Nb[(4 * I + (l & 3)) * 4 + (l >> 4)];
      if (I == s) nv -= ((l & 3) == (l >> 4)) ? 1.0 : 0.0;
#pragma unroll
      for (int J = 0; J < NJ; ++J) t[I][J] = mfma4(nv, u[J], t[I][J]);
    }
    if ((cc >> 2) == q) {
#pragma unroll
      for (int I = 0; I < NI; ++I) t[I][Js] = Nb[(4 * I + i) * 4 + (cc & 3)];
    }
    HMV_WAVE_SYNC();
  });
  if (want_logdet) {
    double v = row16_sum_dpp(log(mypiv));          // padded lanes hold 1.0 -> log = 0
    logdet = readlane_f64(v, 0) + readlane_f64(v, 16) + readlane_f64(v, 32) + readlane_f64(v, 48);
  }
}

template <int NT>
__global__ void __launch_bounds__(256, 2) yw_kernel(YwArgs a) {
  using C = YwCfg<NT>;
  constexpr int MP = C::MP, S = C::S, NIW = NT, NJ = NT;
  constexpr int TILE = MP * MP;
  __shared__ double Xs[MP * S];
  __shared__ double Ys[MP * S];
  __shared__ double Pb[MP * 4];
  __shared__ double Nb[MP * 4];

  const int l = lane_id();
  const int wv = uni(threadIdx.x >> 6);
  const int i = l >> 4, cc = l & 15;
  const long long item = blockIdx.x;
  const int p = a.p;
  const double* R = a.R + (size_t)item * (p + 1) * TILE;
  double* ws = a.ws + (size_t)item * yw_ws_tiles_d(p) * TILE;
  const long long ntri = yw_tri(p + 1, 0);
  double* Yt = ws;                      // Y tiles  (lower triangle)
  double* Lt = ws + ntri * TILE;        // Lt tiles (lower triangle)
  double* Dinv = ws + 2 * ntri * TILE;  // p tiles
  double* Zt = Dinv + (size_t)p * TILE; // p tiles
  int info = 0;

  // ---- helpers -------------------------------------------------------------------------------
  // Both operand tiles of a product are fetched with ALL loads in flight before the first LDS store
  // (a load->store loop serialises on the ~1 us global latency 16 times per tile).
  constexpr int NPT = TILE / 256;   // elements per thread per tile (1, 4, 9, 16)
  auto stage2 = [&](const double* srcX, bool trX, const double* srcY, bool trY) {
    double vx[NPT], vy[NPT];
#pragma unroll
    for (int r = 0; r < NPT; ++r) vx[r] = srcX ? srcX[threadIdx.x + 256 * r] : 0.0;
#pragma unroll
    for (int r = 0; r < NPT; ++r) vy[r] = srcY[threadIdx.x + 256 * r];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < NPT; ++r) {
      const int idx = threadIdx.x + 256 * r;
      const int row = idx / MP, col = idx - row * MP;
      if (srcX) Xs[trX ? col * S + row : row * S + col] = vx[r];
      Ys[trY ? col * S + row : row * S + col] = vy[r];
    }
  };
  // acc[ii][J] += Xs(rows of this wave) * Ys^T
  auto gemm_nt = [&](double (&acc)[NIW][NJ]) {
    const double* xa = Xs + (4 * wv * NT + (l & 3)) * S + (l >> 4);
    const double* yb = Ys + cc * S + (l >> 4);
#pragma unroll 2
    for (int k0 = 0; k0 < MP; k0 += 4) {
      double av[NIW], bv[NJ];
#pragma unroll
      for (int ii = 0; ii < NIW; ++ii) av[ii] = xa[4 * ii * S + k0];
#pragma unroll
      for (int J = 0; J < NJ; ++J) bv[J] = yb[16 * J * S + k0];
#pragma unroll
      for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
        for (int J = 0; J < NJ; ++J) acc[ii][J] = mfma4(av[ii], bv[J], acc[ii][J]);
    }
  };
  auto store_tile = [&](double* dst, const double (&v)[NIW][NJ]) {
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) dst[(size_t)(4 * (wv * NT + ii) + i) * MP + 16 * J + cc] = v[ii][J];
  };
  auto strip_to_lds = [&](double* dst, const double (&v)[NIW][NJ]) {
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) dst[(4 * (wv * NT + ii) + i) * S + 16 * J + cc] = v[ii][J];
  };
  // wave 0: log det of the SPD tile currently in Xs (all strips written, barrier passed)
  auto tile_inverse_from_Xs = [&](double* dinv_out, double* logdet_out) {
    if (wv == 0) {
      double t[4 * NT][NT];
#pragma unroll
      for (int I = 0; I < 4 * NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) t[I][J] = Xs[(4 * I + i) * S + 16 * J + cc];
      double ld = 0.0;
      spd_inverse_wave<NT>(t, Pb, Nb, info, ld, logdet_out != nullptr);
      if (dinv_out) {
#pragma unroll
        for (int I = 0; I < 4 * NT; ++I)
#pragma unroll
          for (int J = 0; J < NT; ++J) dinv_out[(size_t)(4 * I + i) * MP + 16 * J + cc] = t[I][J];
      }
      if (logdet_out && l == 0) *logdet_out = ld;
    }
  };

  // ---- block LDL^T, left-looking ----------------------------------------------------------------
  for (int ta = 0; ta <= p; ++ta) {
    for (int tb = 0; tb <= ta; ++tb) {
      double g[NIW][NJ], acc[NIW][NJ];
#pragma unroll
      for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
        for (int J = 0; J < NJ; ++J) {
          const int row = 4 * (wv * NT + ii) + i, col = 16 * J + cc;
          double v;
          if (ta < p) v = R[(size_t)(ta - tb) * TILE + row * MP + col];
          else if (tb < p) v = R[(size_t)(tb + 1) * TILE + col * MP + row];
          else v = R[row * MP + col];
          g[ii][J] = v;
          acc[ii][J] = 0.0;
        }
      for (int c = 0; c < tb; ++c) {
        __syncthreads();
        stage2(Lt + yw_tri(ta, c) * TILE, false, Yt + yw_tri(tb, c) * TILE, false);
        __syncthreads();
        gemm_nt(acc);
        if (a.Vq_logdet && ta == p && tb == p) {   // V_{c+1} = R_0 - sum_{c' <= c} ...
          double vq[NIW][NJ];
#pragma unroll
          for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
            for (int J = 0; J < NJ; ++J) vq[ii][J] = g[ii][J] - acc[ii][J];
          __syncthreads();
          strip_to_lds(Xs, vq);
          __syncthreads();
          tile_inverse_from_Xs(nullptr, a.Vq_logdet + (size_t)item * p + c);
        }
      }
#pragma unroll
      for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
        for (int J = 0; J < NJ; ++J) g[ii][J] -= acc[ii][J];

      if (tb < ta) {
        store_tile(Yt + yw_tri(ta, tb) * TILE, g);
        __syncthreads();
        strip_to_lds(Xs, g);
        stage2(nullptr, false, Dinv + (size_t)tb * TILE, true);
        __syncthreads();
#pragma unroll
        for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
          for (int J = 0; J < NJ; ++J) acc[ii][J] = 0.0;
        gemm_nt(acc);
        store_tile(Lt + yw_tri(ta, tb) * TILE, acc);
      } else if (ta < p) {
        __syncthreads();
        strip_to_lds(Xs, g);
        __syncthreads();
        tile_inverse_from_Xs(Dinv + (size_t)ta * TILE, nullptr);
      } else {
        store_tile(a.V + (size_t)item * TILE, g);
      }
      __syncthreads();
    }
  }

  // ---- back substitution with the unit lower factor -----------------------------------------------
  for (int tb = p - 1; tb >= 0; --tb) {
    double g[NIW][NJ], acc[NIW][NJ];
    const double* src = Lt + yw_tri(p, tb) * TILE;
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) {
        g[ii][J] = src[(size_t)(4 * (wv * NT + ii) + i) * MP + 16 * J + cc];
        acc[ii][J] = 0.0;
      }
    for (int c = tb + 1; c < p; ++c) {
      __syncthreads();
      stage2(Zt + (size_t)c * TILE, false, Lt + yw_tri(c, tb) * TILE, true);
      __syncthreads();
      gemm_nt(acc);
    }
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) g[ii][J] -= acc[ii][J];
    store_tile(Zt + (size_t)tb * TILE, g);
    double* ar = a.ar + (size_t)item * TILE * p;
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J)
        ar[((size_t)(4 * (wv * NT + ii) + i) * MP + 16 * J + cc) * p + tb] = g[ii][J];
    __syncthreads();
  }
  if (wv == 0 && l == 0) a.info[item] = info;
}

int launch_yw(const YwArgs& a, int m_pad, hipStream_t st) {
  if (a.n_items == 0) return 0;
  const dim3 grid((unsigned)a.n_items), block(256);
  switch (m_pad) {
    case 16: hipLaunchKernelGGL(yw_kernel<1>, grid, block, 0, st, a); break;
    case 32: hipLaunchKernelGGL(yw_kernel<2>, grid, block, 0, st, a); break;
    case 48: hipLaunchKernelGGL(yw_kernel<3>, grid, block, 0, st, a); break;
    case 64: hipLaunchKernelGGL(yw_kernel<4>, grid, block, 0, st, a); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

}  // namespace hmv
