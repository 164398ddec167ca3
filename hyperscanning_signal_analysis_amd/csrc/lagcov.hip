// K1 -- lagged autocovariance of every window:  R_l = (1/n) X[:, :n-l] X[:, l:]^T,  l = 0..p.
//
// Replaces `count_corr` (/root/reference/src/mtmvar.py:35-87): p+1 `np.dot` calls per window, biased
// 1/n scaling for every lag (mtmvar.py:57,72), data NOT demeaned (quirk Q1).  The block-Toeplitz
// r_left / r_right of the reference are never materialised -- K2 consumes the p+1 blocks directly.
//
// Mapping: one workgroup = one window x one lag; wave w owns a strip of MP/4 rows of that lag's MP x MP
// accumulator (D layout, hmv_common.h).  The window is streamed through LDS in chunks of TC samples plus
// a halo of 32 lagged samples (all global loads of a chunk in flight before the LDS stores); samples past
// the window end and channels past m are staged as zeros, so every lag runs the same t-loop (the products
// with t + l >= n vanish).  Per 4-sample k-step a wave issues NT A-operand reads + NT B-operand reads
// (ds_read_b64, conflict-free because the row stride is 6 mod 32 doubles) for NT*NT v_mfma_f64_4x4x4_4b_f64.
#include "hmv_common.h"
#include "hmv_kernels.h"

namespace hmv {

constexpr int LC_TC = 64;      // samples per chunk
constexpr int LC_HALO = 32;    // max lag
constexpr int LC_S = LC_TC + LC_HALO + 6;   // 102 = 6 (mod 32)

// grid (items, p+1): one workgroup = one window x ONE lag; wave w owns the row strip 4*NT*w/4 .. of the
// MP x MP accumulator (NT row blocks x NT column groups = NT*NT accumulators per lane), so every lag gets
// the same four waves and no wave idles when p+1 is not a multiple of four.
template <int NT>
__global__ void __launch_bounds__(256, 4) lagcov_kernel(LagcovArgs a) {
  constexpr int MP = 16 * NT, NIW = NT, NJ = NT;
  __shared__ double xs[MP * LC_S];
  const int l = lane_id();
  const int wv = uni(threadIdx.x >> 6);
  const long long item = blockIdx.x;
  const int lag = blockIdx.y;
  const int i = l >> 4, cc = l & 15;
  const int n = a.n, m = a.m;
  const double* x = a.x + a.item_rec[item] * a.rec_stride + a.item_start[item];

  double acc[NIW][NJ];
#pragma unroll
  for (int I = 0; I < NIW; ++I)
#pragma unroll
    for (int J = 0; J < NJ; ++J) acc[I][J] = 0.0;

  constexpr int W = LC_TC + LC_HALO;   // 96 staged samples per channel
  constexpr int NLD = (MP * W + 255) / 256;
  for (int t0 = 0; t0 < n; t0 += LC_TC) {
    double stg[NLD];
#pragma unroll
    for (int r = 0; r < NLD; ++r) {                     // all loads of the chunk in flight, then the LDS stores
      const int idx = threadIdx.x + 256 * r;
      const int ch = idx / W, tt = idx - ch * W;
      const int t = t0 + tt;
      stg[r] = (idx < MP * W && ch < m && t < n) ? x[(size_t)ch * a.ld + t] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      const int idx = threadIdx.x + 256 * r;
      const int ch = idx / W, tt = idx - ch * W;
      if (idx < MP * W) xs[ch * LC_S + tt] = stg[r];
    }
    __syncthreads();
    const int steps = min(LC_TC, n - t0 + 3) >> 2;
    const double* xa = xs + (4 * NT * wv + (l & 3)) * LC_S + (l >> 4);
    const double* xb = xs + cc * LC_S + lag + (l >> 4);
#pragma unroll 2
    for (int ts = 0; ts < steps; ++ts) {
      double av[NIW], bv[NJ];
#pragma unroll
      for (int I = 0; I < NIW; ++I) av[I] = xa[4 * I * LC_S + 4 * ts];
#pragma unroll
      for (int J = 0; J < NJ; ++J) bv[J] = xb[16 * J * LC_S + 4 * ts];
#pragma unroll
      for (int I = 0; I < NIW; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) acc[I][J] = mfma4(av[I], bv[J], acc[I][J]);
    }
  }
  const double scale = 1.0 / (double)n;   // `corr_scale = 1 / n`, multiplied (mtmvar.py:57-59)
  double* R = a.R + ((size_t)item * (a.p + 1) + lag) * MP * MP;
#pragma unroll
  for (int I = 0; I < NIW; ++I)
#pragma unroll
    for (int J = 0; J < NJ; ++J) {
      const int row = 4 * (NT * wv + I) + i, col = 16 * J + cc;
      double v = acc[I][J] * scale;
      if (lag == 0 && row == col && row >= m) v = 1.0;   // padded channels: identity block keeps G SPD
      R[(size_t)row * MP + col] = v;
    }
}

int launch_lagcov(const LagcovArgs& a, int m_pad, hipStream_t st) {
  if (a.n_items == 0) return 0;
  if (a.p > LC_HALO) return -2;
  const dim3 grid((unsigned)a.n_items, a.p + 1), block(256);
  switch (m_pad) {
    case 16: hipLaunchKernelGGL(lagcov_kernel<1>, grid, block, 0, st, a); break;
    case 32: hipLaunchKernelGGL(lagcov_kernel<2>, grid, block, 0, st, a); break;
    case 48: hipLaunchKernelGGL(lagcov_kernel<3>, grid, block, 0, st, a); break;
    case 64: hipLaunchKernelGGL(lagcov_kernel<4>, grid, block, 0, st, a); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

}  // namespace hmv
