// K2, second form -- block Levinson-Whittle recursion on the p + 1 lag blocks (Whittle 1963; Wiggins & Robinson 1965).
//
// Replaces `ar_coeff` (/root/reference/src/mtmvar.py:90-123) like yw_solve.hip, but uses what that solver ignores: the
// normal-equation matrix is block-TOEPLITZ.  With C(l) = R_l^T (l >= 0), C(-l) = R_l, the Yule-Walker equations read
//     sum_{k=1..q} A_k^(q) C(l - k) = C(l),  l = 1..q            (forward predictor of order q, error covariance Vf_q)
//     sum_{k=1..q} B_k^(q) C(k - l) = C(-l), l = 1..q            (backward predictor, error covariance Vb_q)
// and order q + 1 follows from order q with the partial correlation D_q = C(q+1) - sum_k A_k^(q) C(q+1-k):
//     A_{q+1}^(q+1) = D_q Vb_q^-1            B_{q+1}^(q+1) = D_q^T Vf_q^-1
//     A_k^(q+1) = A_k^(q) - A_{q+1}^(q+1) B_{q+1-k}^(q)        B_k^(q+1) = B_k^(q) - B_{q+1}^(q+1) A_{q+1-k}^(q)
//     Vf_{q+1} = Vf_q - A_{q+1}^(q+1) D_q^T                    Vb_{q+1} = Vb_q - B_{q+1}^(q+1) D_q
// ar[:, :, k] = A_{k+1}^(p), V = Vf_p, and the Vf_q of every lower order (the model-order criterion,
// mtmvar.py:551-601) come out on the way.  Per window: 115 tile products and 16 tile inverses at p = 8 against 184 + 8
// for the block LDL^T, and the state is 2p coefficient tiles instead of a (p+1)(p+2)/2-tile factor: ~170 tile moves
// instead of ~360.
//
// Numerics.  Levinson-type recursions are only weakly stable: the error grows with the condition of the lag-0 blocks'
// Schur complements (Vf_q, Vb_q), where the LDL^T of the whole Gram matrix loses cond * eps.  Measured against the
// reference's dgesv on the nearly collinear fixtures (tests/golden/g6_errors.npz): 3e-11 at cond 2e5 (LDL^T 3e-12),
// 4e-5 at cond 2e9 (2e-8).  Hence the guard: every tile inverse reports its smallest and largest pivot, and a window
// in which any inverse met min / max < HMV_LWR_GUARD is flagged in `guard[item]`; the launcher then re-solves exactly
// the flagged windows with the LDL^T kernel (one more launch whose other workgroups exit at once).  Well-conditioned
// windows (every window of the synthetic benchmark; cond ~ 1e4) never take that path.
//
// One workgroup of four waves per window walks the whole recursion in ONE launch; the tile products are the same
// MFMA kernel body as yw_solve.hip (operands staged through LDS in two k-halves, 39 KB, three workgroups per CU).
#include "yw_common.h"

namespace hmv {

#ifndef HMV_LWR_GUARD
#define HMV_LWR_GUARD 1e-7
#endif

template <int NT, bool VQ>
__global__ void __launch_bounds__(256, 3) yw_lwr_kernel(YwArgs a) {
  constexpr int MP = 16 * NT, KH = MP / 2, SH = KH + 6, NIW = NT, NJ = NT, TILE = MP * MP;
  constexpr int NV = (MP * KH / 2 + 255) / 256;
  constexpr int SI = YwCfg<NT>::S;
  constexpr int GEMM_D = 2 * MP * SH, INV_D = MP * SI, BUF_D = GEMM_D > INV_D ? GEMM_D : INV_D;
  __shared__ __attribute__((aligned(16))) double buf[BUF_D];
  __shared__ double Pb[2 * MP * 4];            // two panel and four N buffers: both inverses of an order at once
  __shared__ double Nb[4 * MP * 4];
  __shared__ int s_info;
  __shared__ double s_ld[4];
  __shared__ double s_pm[4 + 4 * 4];
  __shared__ int s_guard;
  double* Xh = buf;
  double* Yh = buf + MP * SH;
  const int wv = uni(threadIdx.x >> 6);
  const int p = a.p;
  const long long item = blockIdx.x;
  // scratch tiles of this window: A and B in two generations, Vf, Vb, their inverses, D, and one spare
  double* ws = a.ws + (size_t)item * yw_ws_tiles_d(p) * TILE;
  double* Agen[2] = {ws, ws + (size_t)p * TILE};
  double* Bgen[2] = {ws + (size_t)2 * p * TILE, ws + (size_t)3 * p * TILE};
  double* Vf = ws + (size_t)4 * p * TILE;
  double* Vb = Vf + TILE;
  double* VfI = Vb + TILE;
  double* VbI = VfI + TILE;
  double* Dq = VbI + TILE;
  const double* R = a.R + (size_t)item * (p + 1) * TILE;
  if (threadIdx.x == 0) {
    s_info = 0;
    s_guard = 0;
  }

  auto lane = [&]() __attribute__((always_inline)) {
    int lo;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo));
    return lo;
  };
  auto zero = [&](double (&v)[NIW][NJ]) __attribute__((always_inline)) {
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) v[ii][J] = 0.0;
  };
  // ---- operand staging, one k-half at a time (as in yw_solve.hip).  Plain image: dst[row][k] = src[row][kh*KH + k];
  // transposed image: dst[col][k] = src[kh*KH + k][col].
  auto fetch = [&](f64x2 (&v)[NV], const double* src, int kh, bool tr) __attribute__((always_inline)) {
    const int t0 = lane() + 64 * wv;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
      const int idx = t0 + 256 * r;
      if (NV * 256 == MP * KH / 2 || idx < MP * KH / 2) {
        if (!tr) {
          const int row = idx / (KH / 2), c2 = idx - row * (KH / 2);
          v[r] = *reinterpret_cast<const f64x2*>(src + (size_t)row * MP + kh * KH + 2 * c2);
        } else {
          const int k = idx / (MP / 2), c2 = idx - k * (MP / 2);
          v[r] = *reinterpret_cast<const f64x2*>(src + (size_t)(kh * KH + k) * MP + 2 * c2);
        }
      }
    }
  };
  auto park = [&](double* dst, const f64x2 (&v)[NV], bool tr) __attribute__((always_inline)) {
    const int t0 = lane() + 64 * wv;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
      const int idx = t0 + 256 * r;
      if (NV * 256 == MP * KH / 2 || idx < MP * KH / 2) {
        if (!tr) {
          const int row = idx / (KH / 2), c2 = idx - row * (KH / 2);
          double* d = dst + row * SH + 2 * c2;
          d[0] = v[r].x;
          d[1] = v[r].y;
        } else {
          const int k = idx / (MP / 2), c2 = idx - k * (MP / 2);
          dst[(2 * c2) * SH + k] = v[r].x;
          dst[(2 * c2 + 1) * SH + k] = v[r].y;
        }
      }
    }
  };
  auto park_strip = [&](const double (&v)[NIW][NJ], int kh) __attribute__((always_inline)) {
    const int l = lane(), i = l >> 4, cc = l & 15;
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) {
        const int col = 16 * J + cc - kh * KH;
        if (col >= 0 && col < KH) Xh[(4 * (wv * NT + ii) + i) * SH + col] = v[ii][J];
      }
  };
  // One k-half of a product: KH / 4 k-steps of NIW + NJ operand reads and NIW * NJ MFMAs.  The operands of step s + 1 are
  // requested BEFORE the MFMAs of step s are issued (two register sets, ping-pong): written as "reads, wait, MFMAs" per
  // step the LDS latency of every step was exposed -- ~200 of ~450 cycles per step with one wave per SIMD, the reason a
  // 64^3 product took 10 -- 13 k cycles for 4.1 k cycles of matrix pipe (profiles/r03_k2_notes.md).
#ifndef HMV_LWR_GEMM_PIPE
#define HMV_LWR_GEMM_PIPE 1
#endif
  auto gemm_half = [&](double (&acc)[NIW][NJ]) __attribute__((always_inline)) {
    const int l = lane();
    const double* xa = Xh + (4 * wv * NT + (l & 3)) * SH + (l >> 4);
    const double* yb = Yh + (l & 15) * SH + (l >> 4);
#if HMV_LWR_GEMM_PIPE
    constexpr int NS = KH / 4;
    double av[2][NIW], bv[2][NJ];
    auto rd = [&](int set, int k0) __attribute__((always_inline)) {
#pragma unroll
      for (int ii = 0; ii < NIW; ++ii) av[set][ii] = xa[4 * ii * SH + k0];
#pragma unroll
      for (int J = 0; J < NJ; ++J) bv[set][J] = yb[16 * J * SH + k0];
    };
    rd(0, 0);
    static_for<NS>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value, cur = s & 1;
      if constexpr (s + 1 < NS) rd(cur ^ 1, 4 * (s + 1));
      __builtin_amdgcn_sched_barrier(0);          // the requests above stay above the MFMAs below
#pragma unroll
      for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
        for (int J = 0; J < NJ; ++J) acc[ii][J] = mfma4(av[cur][ii], bv[cur][J], acc[ii][J]);
      __builtin_amdgcn_sched_barrier(0);
    });
#else
#pragma unroll 2
    for (int k0 = 0; k0 < KH; k0 += 4) {
      double av[NIW], bv[NJ];
#pragma unroll
      for (int ii = 0; ii < NIW; ++ii) av[ii] = xa[4 * ii * SH + k0];
#pragma unroll
      for (int J = 0; J < NJ; ++J) bv[J] = yb[16 * J * SH + k0];
#pragma unroll
      for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
        for (int J = 0; J < NJ; ++J) acc[ii][J] = mfma4(av[ii], bv[J], acc[ii][J]);
    }
#endif
  };
  // acc += X' * Y'^T;  X' = srcX (global tile; transposed if trX) or, if srcX == nullptr, the register tile xr;
  // Y' = srcY (transposed if trY).  I.e. trY = false: X' Y^T, trY = true: X' Y.
  auto product = [&](double (&acc)[NIW][NJ], const double* srcX, bool trX, const double (&xr)[NIW][NJ], const double* srcY,
                     bool trY) __attribute__((always_inline)) {
    f64x2 vx[NV], vy[NV];
    if (srcX) fetch(vx, srcX, 0, trX);
    fetch(vy, srcY, 0, trY);
    __syncthreads();
    if (srcX) park(Xh, vx, trX); else park_strip(xr, 0);
    park(Yh, vy, trY);
    if (srcX) fetch(vx, srcX, 1, trX);
    fetch(vy, srcY, 1, trY);
    __syncthreads();
    gemm_half(acc);
    __syncthreads();
    if (srcX) park(Xh, vx, trX); else park_strip(xr, 1);
    park(Yh, vy, trY);
    __syncthreads();
    gemm_half(acc);
  };
  auto store_tile = [&](double* dst, const double (&v)[NIW][NJ]) __attribute__((always_inline)) {
    const int l = lane(), i = l >> 4, cc = l & 15;
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) dst[(size_t)(4 * (wv * NT + ii) + i) * MP + 16 * J + cc] = v[ii][J];
  };
  // tile (or its transpose) -> this workgroup's register tile
  auto load_tile = [&](double (&v)[NIW][NJ], const double* src, bool tr) __attribute__((always_inline)) {
    const int l = lane(), i = l >> 4, cc = l & 15;
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) {
        const int row = 4 * (wv * NT + ii) + i, col = 16 * J + cc;
        v[ii][J] = tr ? src[(size_t)col * MP + row] : src[(size_t)row * MP + col];
      }
  };
  auto sub = [&](double (&g)[NIW][NJ], const double (&acc)[NIW][NJ]) __attribute__((always_inline)) {
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) g[ii][J] -= acc[ii][J];
  };
  // inverse of the SPD register tile g -> global tile `out`; conditioning guard; optional log det
  auto invert = [&](const double (&g)[NIW][NJ], double* out, double* logdet, int info_base) __attribute__((always_inline)) {
    const int l = lane(), i = l >> 4, cc = l & 15;
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) buf[(4 * (wv * NT + ii) + i) * SI + 16 * J + cc] = g[ii][J];
    __syncthreads();
    spd_inverse_coop<NT, SI>(buf, Pb, Nb, &s_info, s_ld, out, logdet, info_base, s_pm);
    if (threadIdx.x == 0 && !(s_pm[0] >= HMV_LWR_GUARD * s_pm[1])) s_guard = 1;      // also catches NaN
    __syncthreads();            // the inverse is in global memory for the whole workgroup
  };

  // both error covariances of an order inverted together (spd_inverse_coop2: their panels on different waves at the
  // same time, one barrier per block step for the two)
  auto invert2 = [&](const double (&ga)[NIW][NJ], const double (&gb)[NIW][NJ], double* out_a, double* out_b, double* logdet_b,
                     int info_base) __attribute__((always_inline)) {
    spd_inverse_coop2<NT, SI>(ga, gb, buf, Pb, Nb, &s_info, s_ld, out_a, out_b, logdet_b, info_base, s_pm);
    if (threadIdx.x == 0 && (!(s_pm[0] >= HMV_LWR_GUARD * s_pm[1]) || !(s_pm[2] >= HMV_LWR_GUARD * s_pm[3]))) s_guard = 1;
    __syncthreads();            // the inverses are in global memory for the whole workgroup
  };

  double g[NIW][NJ], acc[NIW][NJ], dacc[NIW][NJ];
  const double (&none)[NIW][NJ] = g;
  // ---- order 0: Vf = Vb = C(0) = R_0 (symmetric), D_0 = C(1) = R_1^T
  load_tile(g, R, false);
  store_tile(Vf, g);
  store_tile(Vb, g);
  load_tile(g, R + TILE, true);
  store_tile(Dq, g);
  __syncthreads();
  for (int q = 0; q < p; ++q) {
    const double* Ao = Agen[q & 1];
    const double* Bo = Bgen[q & 1];
    double* An = Agen[(q + 1) & 1];
    double* Bn = Bgen[(q + 1) & 1];
    const bool last = (q == p - 1);
    // ---- inverses of the two error covariances of order q (log det Vf_q is the criterion's term of order q)
    load_tile(g, Vb, false);
#ifndef HMV_LWR_SINGLE_INVERSES
    if (!last || (VQ && q >= 1)) {       // (the last order needs Vf^-1 only for its log det)
      load_tile(acc, Vf, false);
      invert2(g, acc, VbI, VfI, (VQ && q >= 1) ? a.Vq_logdet + (size_t)item * p + (q - 1) : nullptr, q * MP);
    } else {
      invert(g, VbI, nullptr, q * MP);
    }
#else
    invert(g, VbI, nullptr, q * MP);
    if (!last || (VQ && q >= 1)) {
      load_tile(g, Vf, false);
      invert(g, VfI, (VQ && q >= 1) ? a.Vq_logdet + (size_t)item * p + (q - 1) : nullptr, q * MP);
    }
#endif
    // ---- A_{q+1} = D Vb^-1;  Vf <- Vf - A_{q+1} D^T
    zero(acc);
    product(acc, Dq, false, none, VbI, false);           // Vb^-1 is symmetric: X Y^T = D Vb^-1
    store_tile(An + (size_t)q * TILE, acc);
    zero(dacc);
    product(dacc, nullptr, false, acc, Dq, false);       // A_{q+1} D^T
    load_tile(g, Vf, false);
    sub(g, dacc);
    store_tile(Vf, g);
    if (!last) {
      // ---- B_{q+1} = D^T Vf^-1;  Vb <- Vb - B_{q+1} D
      zero(acc);
      product(acc, Dq, true, none, VfI, false);
      store_tile(Bn + (size_t)q * TILE, acc);
      zero(dacc);
      product(dacc, nullptr, false, acc, Dq, true);      // B_{q+1} D
      load_tile(g, Vb, false);
      sub(g, dacc);
      store_tile(Vb, g);
    }
    __syncthreads();                                     // A_{q+1} / B_{q+1} are in global memory
    // ---- lower lags: A'_k = A_k - A_{q+1} B_{q-1-k},  B'_j = B_j - B_{q+1} A_{q-1-j}   (k, j = 0 .. q-1: lag k + 1)
    // and the next partial correlation D' = C(q+2) - sum_{k=0..q} A'_k C(q+1-k), C(l) = R_l^T
    zero(dacc);
    for (int k = 0; k < q; ++k) {
      zero(acc);
      product(acc, An + (size_t)q * TILE, false, none, Bo + (size_t)(q - 1 - k) * TILE, true);
      load_tile(g, Ao + (size_t)k * TILE, false);
      sub(g, acc);
      store_tile(An + (size_t)k * TILE, g);
      if (!last) {
        product(dacc, nullptr, false, g, R + (size_t)(q + 1 - k) * TILE, false);       // A'_k R_{q+1-k}^T
        zero(acc);
        product(acc, Bn + (size_t)q * TILE, false, none, Ao + (size_t)(q - 1 - k) * TILE, true);
        load_tile(g, Bo + (size_t)k * TILE, false);
        sub(g, acc);
        store_tile(Bn + (size_t)k * TILE, g);
      }
    }
    if (!last) {
      product(dacc, An + (size_t)q * TILE, false, none, R + TILE, false);              // A_{q+1} R_1^T
      load_tile(g, R + (size_t)(q + 2) * TILE, true);
      sub(g, dacc);
      store_tile(Dq, g);
    }
    __syncthreads();                                     // generation q + 1 complete
  }
  if (VQ) {                                              // log det Vf_p
    load_tile(g, Vf, false);
    invert(g, VfI, a.Vq_logdet + (size_t)item * p + (p - 1), p * MP);
  }
  // ---- outputs: V = Vf_p, ar[item][row][col][k] = A_{k+1}[row][col] (lag fastest: the reference's (m, m, p) layout)
  load_tile(g, Vf, false);
  store_tile(a.V + (size_t)item * TILE, g);
  if (threadIdx.x == 0) {
    a.info[item] = s_info;
    // (a singular window stays singular: nothing to re-solve)
    *yw_guard_ptr(a.ws, item, p, TILE) = (s_info == 0) ? s_guard : 0;
  }
  const double* Af = Agen[p & 1];
  double* ar = a.ar + (size_t)item * TILE * p;
  const int total = TILE * p;
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int e = idx / p, k = idx - e * p;
    ar[idx] = Af[(size_t)k * TILE + e];
  }
}

// HMV_TUNE_YW_FORM = 3 takes the software-pipelined form of the same recursion (yw_lwr2.hip): measured equal in time
// (1.73 vs 1.74 ms at 599 windows) and 4 % lower in HBM traffic -- see profiles/r03_k2_notes.md -- so this one stays
int launch_yw_lwr(const YwArgs& a, int m_pad, hipStream_t st) {
  if (a.n_items == 0) return 0;
  if (tuning(4 /* HMV_TUNE_YW_FORM */) == 3) return launch_yw_lwr2(a, m_pad, st);
  const dim3 grid((unsigned)a.n_items), block(256);
  const bool vq = (a.Vq_logdet != nullptr);
  switch (m_pad) {
    case 16: if (vq) hipLaunchKernelGGL((yw_lwr_kernel<1, true>), grid, block, 0, st, a);
             else hipLaunchKernelGGL((yw_lwr_kernel<1, false>), grid, block, 0, st, a); break;
    case 32: if (vq) hipLaunchKernelGGL((yw_lwr_kernel<2, true>), grid, block, 0, st, a);
             else hipLaunchKernelGGL((yw_lwr_kernel<2, false>), grid, block, 0, st, a); break;
    case 48: if (vq) hipLaunchKernelGGL((yw_lwr_kernel<3, true>), grid, block, 0, st, a);
             else hipLaunchKernelGGL((yw_lwr_kernel<3, false>), grid, block, 0, st, a); break;
    case 64: if (vq) hipLaunchKernelGGL((yw_lwr_kernel<4, true>), grid, block, 0, st, a);
             else hipLaunchKernelGGL((yw_lwr_kernel<4, false>), grid, block, 0, st, a); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

}  // namespace hmv
