// K1 -- lagged autocovariance of every window:  R_l = (1/n) X[:, :n-l] X[:, l:]^T,  l = 0..p.
//
// Replaces `count_corr` (/root/reference/src/mtmvar.py:35-87): p+1 `np.dot` calls per window, biased
// 1/n scaling for every lag (mtmvar.py:57,72), data NOT demeaned (quirk Q1).  The block-Toeplitz
// r_left / r_right of the reference are never materialised -- K2 consumes the p+1 blocks directly.
//
// Mapping: one workgroup = one window x a group of LG = 3 consecutive lags; wave w owns a strip of MP/4 rows of
// each lag's MP x MP accumulator (D layout, hmv_common.h).  The window is streamed through LDS in chunks of TC samples plus
// a halo of 32 lagged samples (all global loads of a chunk in flight before the LDS stores); samples past
// the window end and channels past m are staged as zeros, so every lag runs the same t-loop (the products
// with t + l >= n vanish).  Per 4-sample k-step a wave issues NT A-operand reads + NT B-operand reads
// (ds_read_b64, conflict-free because the row stride is 6 mod 32 doubles) for NT*NT v_mfma_f64_4x4x4_4b_f64.
#include "hmv_common.h"
#include "hmv_kernels.h"
#include <cstdlib>

namespace hmv {

constexpr int LC_TC = 64;      // samples per chunk
constexpr int LC_HALO = 32;    // max lag
constexpr int LC_S = LC_TC + LC_HALO + 6;   // 102 = 6 (mod 32)

// grid (items, ceil((p+1) / LG)): one workgroup = one window x LG consecutive lags; wave w owns the row strip
// 4*NT*w .. of each lag's MP x MP accumulator (LG * NT * NT accumulators per lane).  The staged chunk (the costly
// part: global loads, two barriers, LDS stores) is shared by the LG lags -- per 4-sample k-step a wave reads NT
// A operands once and NT B operands per lag for LG*NT*NT MFMAs -- and the window is fetched ceil((p+1)/LG) times
// instead of p+1 times (round 1, LG = 1: 21x over-fetch, 0.53 of the f64 peak).
template <int NT, int LG>
__global__ void __launch_bounds__(256, 2) lagcov_kernel(LagcovArgs a) {
  constexpr int MP = 16 * NT, NIW = NT, NJ = NT;
  __shared__ double xs[MP * LC_S];
  const int l = lane_id();
  const int wv = uni(threadIdx.x >> 6);
  const long long item = blockIdx.x;
  const int lag0 = blockIdx.y * LG;
  const int nl = min(LG, a.p + 1 - lag0);          // lags of this workgroup (the last group may be short)
  const int i = l >> 4, cc = l & 15;
  const int n = a.n, m = a.m;            // n: samples summed over (window length, or hop in block mode)
  const bool blk = a.blocks != 0;
  const long long start = blk ? a.blk_first + item * (long long)n : a.item_start[item];
  const double* x = a.x + (blk ? 0 : a.item_rec[item] * a.rec_stride) + start;
  // samples with a value: the window itself, or in block mode everything up to the end of the recording (the lagged
  // partner x[t + l] of the last samples of a block lies in the next block)
  const long long vlen = blk ? a.blk_T - start : (long long)n;
  const bool mask_a = blk && (n & 3) != 0;  // block mode: the A operand must end at n even if real samples follow

  double acc[LG][NIW][NJ];
#pragma unroll
  for (int g = 0; g < LG; ++g)
#pragma unroll
    for (int I = 0; I < NIW; ++I)
#pragma unroll
      for (int J = 0; J < NJ; ++J) acc[g][I][J] = 0.0;

  constexpr int W = LC_TC + LC_HALO;   // 96 staged samples per channel
  constexpr int NLD = (MP * W + 255) / 256;
  for (int t0 = 0; t0 < n; t0 += LC_TC) {
    double stg[NLD];
#pragma unroll
    for (int r = 0; r < NLD; ++r) {                     // all loads of the chunk in flight, then the LDS stores
      const int idx = threadIdx.x + 256 * r;
      const int ch = idx / W, tt = idx - ch * W;
      const int t = t0 + tt;
      stg[r] = (idx < MP * W && ch < m && t < vlen) ? x[(size_t)ch * a.ld + t] : 0.0;
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
    const double* xb = xs + cc * LC_S + lag0 + (l >> 4);
    for (int ts = 0; ts < steps; ++ts) {
      double av[NIW];
#pragma unroll
      for (int I = 0; I < NIW; ++I) av[I] = xa[4 * I * LC_S + 4 * ts];
      if (mask_a && t0 + 4 * ts + 3 >= n) {             // uniform: only the last k-step of a block
        const bool in = t0 + 4 * ts + (l >> 4) < n;
#pragma unroll
        for (int I = 0; I < NIW; ++I) av[I] = in ? av[I] : 0.0;
      }
      static_for<LG>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        if (g < nl) {                                   // workgroup-uniform
          double bv[NJ];
#pragma unroll
          for (int J = 0; J < NJ; ++J) bv[J] = xb[16 * J * LC_S + 4 * ts + g];
#pragma unroll
          for (int I = 0; I < NIW; ++I)
#pragma unroll
            for (int J = 0; J < NJ; ++J) acc[g][I][J] = mfma4(av[I], bv[J], acc[g][I][J]);
        }
      });
    }
  }
  const double scale = blk ? 1.0 : 1.0 / (double)n;   // `corr_scale = 1 / n`, multiplied (mtmvar.py:57-59)
  static_for<LG>([&](auto gc) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value;
    if (g < nl) {
      const int lag = lag0 + g;
      double* R = a.R + ((size_t)item * (a.p + 1) + lag) * MP * MP;
#pragma unroll
      for (int I = 0; I < NIW; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) {
          const int row = 4 * (NT * wv + I) + i, col = 16 * J + cc;
          double v = acc[g][I][J] * scale;
          if (!blk && lag == 0 && row == col && row >= m) v = 1.0;   // padded channels: identity block keeps G SPD
          R[(size_t)row * MP + col] = v;
        }
    }
  });
}

// ---- windows from hop blocks ------------------------------------------------------------------------------
// With windows of n = k * hop samples every hop, each product x_i[t] x_j[t + l] belongs to up to k windows.  The
// block sums Q_l(b) = sum_{t in block b} x[t] x[t + l]^T (lagcov_kernel in block mode) are computed once, and
//     R_l(w) = (Q_l(w) + ... + Q_l(w + k - 1) - C_l(w)) / n,
//     C_l(w) = sum_{t = n - l}^{n - 1} x[s_w + t] x[s_w + t + l]^T     (the l products that reach past the window end)
// -- half the flops of the direct form at 50 % overlap.  Same biased 1/n estimator, no demeaning (mtmvar.py:57-59,
// 72-73); the sums are merely associated differently, so results agree with the direct form to rounding (not bitwise).
// grid (n_win, p + 1), block 256: 16 * NT * NT / 16 elements per thread.
template <int NT>
__global__ void __launch_bounds__(256) lagcomb_kernel(LagcombArgs a) {
  constexpr int MP = 16 * NT, TILE = MP * MP;
  __shared__ double xa[MP * 32], xb[MP * 32];        // x[:, s+n-l .. s+n-1] and x[:, s+n .. s+n+l-1]
  const long long w = blockIdx.x;
  const int lag = blockIdx.y;
  const int n = (int)(a.hop * a.k), m = a.m;
  const long long s = a.first + w * a.hop;
  for (int idx = threadIdx.x; idx < MP * lag; idx += 256) {
    const int ch = idx / lag, u = idx - ch * lag;
    const long long ta = s + n - lag + u, tb = s + n + u;
    xa[ch * 32 + u] = (ch < m && ta < a.T) ? a.x[(size_t)ch * a.ld + ta] : 0.0;
    xb[ch * 32 + u] = (ch < m && tb < a.T) ? a.x[(size_t)ch * a.ld + tb] : 0.0;
  }
  __syncthreads();
  const double inv_n = 1.0 / (double)n;
  const double* Q = a.Q + ((size_t)w * (a.p + 1) + lag) * TILE;
  double* R = a.R + ((size_t)w * (a.p + 1) + lag) * TILE;
  for (int e = threadIdx.x; e < TILE; e += 256) {
    const int row = e / MP, col = e - row * MP;
    double acc = 0.0;
    for (int j = 0; j < a.k; ++j) acc += Q[(size_t)j * (a.p + 1) * TILE + e];
    double c = 0.0;
    for (int u = 0; u < lag; ++u) c = __builtin_fma(xa[row * 32 + u], xb[col * 32 + u], c);
    double v = (acc - c) * inv_n;
    if (lag == 0 && row == col && row >= m) v = 1.0;    // padded channels: identity block keeps G SPD
    R[e] = v;
  }
}

int launch_lagcomb(const LagcombArgs& a, int m_pad, hipStream_t st) {
  if (a.n_win == 0) return 0;
  if (a.p > LC_HALO) return -2;
  const dim3 grid((unsigned)a.n_win, a.p + 1), block(256);
  switch (m_pad) {
    case 16: hipLaunchKernelGGL(lagcomb_kernel<1>, grid, block, 0, st, a); break;
    case 32: hipLaunchKernelGGL(lagcomb_kernel<2>, grid, block, 0, st, a); break;
    case 48: hipLaunchKernelGGL(lagcomb_kernel<3>, grid, block, 0, st, a); break;
    case 64: hipLaunchKernelGGL(lagcomb_kernel<4>, grid, block, 0, st, a); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

template <int LG>
static int launch_lagcov_lg(const LagcovArgs& a, int m_pad, hipStream_t st) {
  const dim3 grid((unsigned)a.n_items, (a.p + LG) / LG), block(256);
  switch (m_pad) {
    case 16: hipLaunchKernelGGL((lagcov_kernel<1, LG>), grid, block, 0, st, a); break;
    case 32: hipLaunchKernelGGL((lagcov_kernel<2, LG>), grid, block, 0, st, a); break;
    case 48: hipLaunchKernelGGL((lagcov_kernel<3, LG>), grid, block, 0, st, a); break;
    case 64: hipLaunchKernelGGL((lagcov_kernel<4, LG>), grid, block, 0, st, a); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

int launch_lagcov(const LagcovArgs& a, int m_pad, hipStream_t st) {
  if (a.n_items == 0) return 0;
  if (a.p > LC_HALO) return -2;
  // every lag's sum runs over the same samples in the same order whatever the grouping: same bits for any LG
  const long long lg = tuning(2 /* HMV_TUNE_LAG_GROUP */);
  switch (lg) {
    case 1: return launch_lagcov_lg<1>(a, m_pad, st);
    case 2: return launch_lagcov_lg<2>(a, m_pad, st);
    default: return launch_lagcov_lg<3>(a, m_pad, st);
  }
}

}  // namespace hmv
