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
// Two forms walk the same tile products in the same order (bit-identical results):
//   * tile form: one workgroup (4 waves) per (window, tile); the factorisation is launched tile column by tile
//     column (the diagonal tile and all tiles below it in ONE launch: the scaling Lt = Y D^-1 of a column is deferred
//     to the next column's launch, so nothing inside a launch waits for the diagonal tile's inverse), so a batch of
//     a few hundred windows still fills the chip; the back substitution and the emit are one more launch with one
//     workgroup per window;
//   * window form (yw_window_kernel): one workgroup per window does everything in one launch.
// Every tile product is an MP x MP x MP real GEMM on v_mfma_f64_4x4x4_4b_f64 with both operands staged in LDS (row
// stride 6 mod 32 doubles: conflict-free A- and B-operand reads); wave w owns row blocks w*NT .. w*NT+NT-1 of the
// output tile.  The p tile inverses D_b^-1 are a cooperative 4-wave blocked Gauss-Jordan (spd_inverse_coop: the wave
// that owns the panel columns factors them lane-per-row, every wave applies the rank-4 update on the matrix pipe;
// same scheme as K3, real, no pivoting).  The stage as a whole is bound by HBM traffic, not by the matrix pipe
// (~14 MB of tile operands and results per window).
#include "yw_common.h"
#include <cstdlib>

namespace hmv {

long long yw_ws_tiles(int p) { return yw_ws_tiles_d(p); }

struct YwPtrs {
  const double* R;
  double *Yt, *Lt, *Dinv, *Zt;
};
template <int MP>
__device__ __forceinline__ YwPtrs yw_ptrs(const YwArgs& a, long long item) {
  constexpr int TILE = MP * MP;
  const int p = a.p;
  double* ws = a.ws + (size_t)item * yw_ws_tiles_d(p) * TILE;
  const long long ntri = yw_tri(p + 1, 0);
  YwPtrs q;
  q.R = a.R + (size_t)item * (p + 1) * TILE;
  q.Yt = ws;
  q.Lt = ws + ntri * TILE;
  q.Dinv = ws + 2 * ntri * TILE;
  q.Zt = q.Dinv + (size_t)p * TILE;
  return q;
}

// ---- the kernel of both forms ---------------------------------------------------------------------------------
// Window form (MODE 0): one workgroup walks the whole left-looking schedule of its window in ONE launch, with the
// scratch tiles in global memory (private to the workgroup: only workgroup barriers order them).  Tile form: the same
// kernel body per tile -- MODE 2 one tile (diagonal or not) of one tile column per workgroup, launched tile column by
// tile column, then MODE 1 (back substitution + emit, one workgroup per window): p + 2 = 10 launches per batch (round 1:
// ~50; round 2a: 18), each filling the chip.  Same tile products in the same order either way: bit-identical results.
// Operands are staged through LDS in two k-halves (39 KB + the inverse's panels: four column-tile workgroups or three
// window workgroups per CU; round 1 staged whole tiles, 72-78 KB, two per CU), the second half's loads in flight
// behind the first half's MFMAs.
template <int NT>
struct YwWin {
  static constexpr int MP = 16 * NT, KH = MP / 2, SH = KH + 6, NIW = NT, NJ = NT, TILE = MP * MP;
  static constexpr int NV = (MP * KH / 2 + 255) / 256;          // 16-byte loads per thread and operand half
  static constexpr int SI = YwCfg<NT>::S;                        // row stride of the full tile image (inverse)
  static constexpr int GEMM_D = 2 * MP * SH, INV_D = MP * SI;
  static constexpr int BUF_D = GEMM_D > INV_D ? GEMM_D : INV_D;  // doubles shared by the two uses
};

// VQ: also the residual covariances of the lower orders (log det V_q, model-order criterion) -- its own
// instantiation because the extra tile and inverse do not fit the 168 registers that three workgroups per CU allow.
// MODE 0: the whole window.  MODE 1 (tile launch chain): only the back substitution and the emit, one launch instead
// of the p - 1 pivot-block launches + 2 emit launches of round 1.  MODE 2 (tile launch chain): ONE tile (ta, tb_arg),
// ta >= tb_arg, of one window per workgroup with this kernel's k-half staging (39 KB of LDS: four workgroups per CU
// instead of the two of round 1's whole-tile column kernel, and the second k-half's loads in flight behind the first
// half's MFMAs).
template <int NT, bool VQ, int MODE = 0>
__global__ void __launch_bounds__(256, VQ ? 2 : (MODE == 2 ? 4 : 3)) yw_window_kernel(YwArgs a, int tb_arg) {
  constexpr bool BACK_ONLY = (MODE == 1);
  using W = YwWin<NT>;
  constexpr int MP = W::MP, KH = W::KH, SH = W::SH, NIW = NT, NJ = NT, TILE = W::TILE, NV = W::NV;
  // The tile kernel (MODE 2) keeps the inverse's image unpadded (read once per inversion: its bank conflicts are
  // noise) with the panel buffers behind it, inside the operand block: 39 KB in all, four workgroups per CU.
  constexpr bool PACKED = (MODE == 2) && !VQ && (MP * MP + 12 * MP <= W::BUF_D);
  constexpr int SI = PACKED ? MP : W::SI;
  __shared__ __attribute__((aligned(16))) double buf[W::BUF_D];
  __shared__ double Pb_[PACKED ? 1 : MP * 4];
  __shared__ double Nb_[PACKED ? 1 : 2 * MP * 4];
  double* Pb = PACKED ? buf + MP * MP : Pb_;
  double* Nb = PACKED ? buf + MP * MP + 4 * MP : Nb_;
  __shared__ int s_info;
  __shared__ double s_ld[4];
  double* Xh = buf;                 // [MP][SH]  k-half of the A-operand tile
  double* Yh = buf + MP * SH;       // [MP][SH]  k-half of the B-operand tile (rows = output columns)
  const int wv = uni(threadIdx.x >> 6);
  const int p = a.p;
  long long item = blockIdx.x;
  int ta_col = 0;
  if (MODE == 2) {
    // XCD-aware block -> (window, tile) map: the p - tb + 1 tiles of one window read the same tb tiles Y[tb][c] and
    // the same D^-1; blocks b and b + 8 share an XCD (and its L2), so the tiles of a window sit 8 blocks apart.
    const unsigned ntile = (unsigned)(p - tb_arg + 1);
    const unsigned grp = blockIdx.x / (8u * ntile), r = blockIdx.x - grp * (8u * ntile);
    item = (long long)grp * 8 + (r & 7u);
    if (item >= a.n_items) return;                       // padding of the last group (whole workgroup)
    ta_col = tb_arg + (int)(r >> 3);                     // the diagonal tile first
  }
  if (MODE == 0 && a.only_guarded) {          // re-solve pass behind the Levinson-Whittle recursion: flagged windows only
    if (*yw_guard_ptr(a.ws, item, p, TILE) == 0) return;
  }
  const YwPtrs q = yw_ptrs<MP>(a, item);
  if (threadIdx.x == 0) s_info = 0;

  // Lane and thread indices are re-derived from an opaque lane id inside every helper: otherwise the compiler
  // hoists the (thread-constant) LDS and global offsets of every staging / tile-I/O variant out of the tile loops
  // and keeps them live across the whole kernel (237 spilled VGPRs at the 168 that three workgroups per CU allow).
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
  // ---- operand staging, one k-half at a time --------------------------------------------------------------
  // fetch: the 16-byte loads of a half into registers; park: registers -> LDS.  Plain image: dst[row][k] =
  // src[row][kh*KH + k]; transposed image: dst[col][k] = src[kh*KH + k][col].
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
  // this wave's row strip of a register tile -> the A-operand half (columns kh*KH .. of the tile)
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
  // acc += Xh(rows of this wave) * Yh^T over the staged k-half
  auto gemm_half = [&](double (&acc)[NIW][NJ]) __attribute__((always_inline)) {
    const int l = lane();
    const double* xa = Xh + (4 * wv * NT + (l & 3)) * SH + (l >> 4);
    const double* yb = Yh + (l & 15) * SH + (l >> 4);
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
  };
  // acc += X * Y'^T with X = srcX (global tile) or, if srcX == nullptr, the register tile xr of this workgroup;
  // Y' = srcY, transposed if trY.  k runs 0 .. MP-1 in ascending order, as in the tile kernels (same bits).
  // The loads of the second half are in flight behind the MFMAs of the first.
  auto product = [&](double (&acc)[NIW][NJ], const double* srcX, const double (&xr)[NIW][NJ], const double* srcY,
                     bool trY) __attribute__((always_inline)) {
    f64x2 vx[NV], vy[NV];
    if (srcX) fetch(vx, srcX, 0, false);
    fetch(vy, srcY, 0, trY);
    __syncthreads();                          // the previous product has finished reading Xh / Yh
    if (srcX) park(Xh, vx, false); else park_strip(xr, 0);
    park(Yh, vy, trY);
    if (srcX) fetch(vx, srcX, 1, false);
    fetch(vy, srcY, 1, trY);
    __syncthreads();
    gemm_half(acc);
    __syncthreads();
    if (srcX) park(Xh, vx, false); else park_strip(xr, 1);
    park(Yh, vy, trY);
    __syncthreads();
    gemm_half(acc);
  };
  auto load_G = [&](double (&g)[NIW][NJ], int ta, int tb) __attribute__((always_inline)) {
    const int l = lane(), i = l >> 4, cc = l & 15;
    // tile (ta, tb) of the augmented matrix straight from the lag covariances: R_{ta-tb}, R_{tb+1}^T or R_0
    const bool tr = !(ta < p) && (tb < p);
    const double* src = q.R + (size_t)((ta < p) ? (ta - tb) : (tb < p ? tb + 1 : 0)) * TILE;
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) {
        const int row = 4 * (wv * NT + ii) + i, col = 16 * J + cc;
        g[ii][J] = tr ? src[col * MP + row] : src[row * MP + col];
      }
  };
  auto store_tile = [&](double* dst, const double (&v)[NIW][NJ]) __attribute__((always_inline)) {
    const int l = lane(), i = l >> 4, cc = l & 15;
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) dst[(size_t)(4 * (wv * NT + ii) + i) * MP + 16 * J + cc] = v[ii][J];
  };
  auto load_tile = [&](double (&v)[NIW][NJ], const double* src) __attribute__((always_inline)) {
    const int l = lane(), i = l >> 4, cc = l & 15;
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) v[ii][J] = src[(size_t)(4 * (wv * NT + ii) + i) * MP + 16 * J + cc];
  };
  // full tile image for the cooperative inverse (aliases the operand halves)
  auto tile_to_lds = [&](const double (&v)[NIW][NJ]) __attribute__((always_inline)) {
    const int l = lane(), i = l >> 4, cc = l & 15;
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) buf[(4 * (wv * NT + ii) + i) * SI + 16 * J + cc] = v[ii][J];
    __syncthreads();
  };

  double g[NIW][NJ], acc[NIW][NJ];
  const double (&none)[NIW][NJ] = g;
  if (MODE == 2) {
    // One tile (ta, tb), ta >= tb, of tile column tb.  Nothing in a launch depends on anything else in it: the scaling
    // by the pivot block's inverse, Lt[ta][tb-1] = Y[ta][tb-1] D_{tb-1}^-1, is deferred to the workgroup that first
    // needs it -- this one, one launch later -- so the diagonal tile (its product sum and its inversion) runs beside
    // the tiles below it instead of in a launch of its own between two column launches (round 2a: 18 launches, the
    // nine diagonal ones with one workgroup per window, a third of the chain's time).
    const int tb = tb_arg, ta = ta_col;
    if (tb > 0) {
      load_tile(g, q.Yt + yw_tri(ta, tb - 1) * TILE);
      zero(acc);
      product(acc, nullptr, g, q.Dinv + (size_t)(tb - 1) * TILE, true);
      store_tile(q.Lt + yw_tri(ta, tb - 1) * TILE, acc);
      if (ta == p) store_tile(q.Zt + (size_t)(tb - 1) * TILE, acc);      // start value of the back substitution
      __syncthreads();                 // ... and operand of this workgroup's last product below
    }
    load_G(g, ta, tb);
    zero(acc);
    for (int c = 0; c < tb; ++c) {
      product(acc, q.Lt + yw_tri(ta, c) * TILE, none, q.Yt + yw_tri(tb, c) * TILE, false);
      if (VQ && ta == p && tb == p) {   // V_{c+1}: residual covariance of order c+1 (model-order criterion)
        double vq[NIW][NJ];
#pragma unroll
        for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
          for (int J = 0; J < NJ; ++J) vq[ii][J] = g[ii][J] - acc[ii][J];
        tile_to_lds(vq);
        spd_inverse_coop<NT, SI>(buf, Pb, Nb, &s_info, s_ld, nullptr, a.Vq_logdet + (size_t)item * p + c, tb * MP);
      }
    }
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) g[ii][J] -= acc[ii][J];
    if (ta != tb) {
      store_tile(q.Yt + yw_tri(ta, tb) * TILE, g);
    } else if (tb == p) {
      store_tile(a.V + (size_t)item * TILE, g);
    } else {
      tile_to_lds(g);
      spd_inverse_coop<NT, SI>(buf, Pb, Nb, &s_info, s_ld, q.Dinv + (size_t)tb * TILE, nullptr, tb * MP);
      __syncthreads();
      if (threadIdx.x == 0 && s_info != 0) atomicCAS(&a.info[item], 0, s_info);
    }
    return;
  }
  for (int tb = 0; tb <= (BACK_ONLY ? -1 : p); ++tb) {
    // ---- diagonal tile: D = G[tb][tb] - sum_c Lt[tb][c] Y[tb][c]^T ; D^-1 (tb < p) or V (tb == p)
    load_G(g, tb, tb);
    zero(acc);
    for (int c = 0; c < tb; ++c) {
      product(acc, q.Lt + yw_tri(tb, c) * TILE, none, q.Yt + yw_tri(tb, c) * TILE, false);
      if (VQ && tb == p) {   // V_{c+1}: residual covariance of order c+1 (model-order criterion)
        double vq[NIW][NJ];
#pragma unroll
        for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
          for (int J = 0; J < NJ; ++J) vq[ii][J] = g[ii][J] - acc[ii][J];
        tile_to_lds(vq);
        spd_inverse_coop<NT, SI>(buf, Pb, Nb, &s_info, s_ld, nullptr, a.Vq_logdet + (size_t)item * p + c, tb * MP);
      }
    }
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) g[ii][J] -= acc[ii][J];
    if (tb == p) {
      store_tile(a.V + (size_t)item * TILE, g);
      break;
    }
    tile_to_lds(g);
    spd_inverse_coop<NT, SI>(buf, Pb, Nb, &s_info, s_ld, q.Dinv + (size_t)tb * TILE, nullptr, tb * MP);
    __syncthreads();                   // D^-1 is in global memory for the whole workgroup
    // ---- tiles below it: Y = G - sum_c Lt[ta][c] Y[tb][c]^T ; Lt = Y D^-1
    for (int ta = tb + 1; ta <= p; ++ta) {
      load_G(g, ta, tb);
      zero(acc);
      for (int c = 0; c < tb; ++c)
        product(acc, q.Lt + yw_tri(ta, c) * TILE, none, q.Yt + yw_tri(tb, c) * TILE, false);
#pragma unroll
      for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
        for (int J = 0; J < NJ; ++J) g[ii][J] -= acc[ii][J];
      store_tile(q.Yt + yw_tri(ta, tb) * TILE, g);
      zero(acc);
      product(acc, nullptr, g, q.Dinv + (size_t)tb * TILE, true);
      store_tile(q.Lt + yw_tri(ta, tb) * TILE, acc);
      if (ta == p) store_tile(q.Zt + (size_t)tb * TILE, acc);      // start value of the back substitution
    }
    __syncthreads();                   // column tb of Y / Lt is complete before column tb + 1 reads it
  }
  if (!BACK_ONLY && threadIdx.x == 0 && s_info != 0) a.info[item] = s_info;
  // ---- back substitution with the unit lower factor: Z[b] = Lt[p][b] - sum_{c > b} Z[c] Lt[c][b].  Row by row
  // (b = p-2 .. 0) with Z[b] kept in registers while its terms c = p-1 .. b+1 are subtracted one product at a time:
  // every Z[b] sees the same subtractions in the same order as in the pivot-block-by-pivot-block form of the tile
  // kernels (same bits), but is read and written once instead of once per term.
  __syncthreads();
  for (int b = p - 2; b >= 0; --b) {
    load_tile(g, q.Zt + (size_t)b * TILE);
    for (int c = p - 1; c > b; --c) {
      zero(acc);
      product(acc, q.Zt + (size_t)c * TILE, none, q.Lt + yw_tri(c, b) * TILE, true);
#pragma unroll
      for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
        for (int J = 0; J < NJ; ++J) g[ii][J] -= acc[ii][J];
    }
    store_tile(q.Zt + (size_t)b * TILE, g);
    __syncthreads();                   // Z[b] is in global memory before row b - 1 reads it
  }
  // ---- ar[item][row][col][k] = Z[k][row][col]  (lag fastest: the reference's (m, m, p) layout)
  double* ar = a.ar + (size_t)item * TILE * p;
  const int total = TILE * p;
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int e = idx / p, k = idx - e * p;
    ar[idx] = q.Zt[(size_t)k * TILE + e];
  }
}

template <int NT>
static int launch_yw_nt(const YwArgs& a, hipStream_t st) {
  const int p = a.p;
  const unsigned n = (unsigned)a.n_items;
  if (const hipError_t e = hipMemsetAsync(a.info, 0, sizeof(int) * a.n_items, st)) return (int)e;
  for (int tb = 0; tb <= p; ++tb) {          // tile column tb: p - tb + 1 tiles per window, the diagonal one included
    const dim3 grid(((n + 7) / 8) * 8 * (unsigned)(p - tb + 1));
    if (a.Vq_logdet && tb == p) hipLaunchKernelGGL((yw_window_kernel<NT, true, 2>), grid, dim3(256), 0, st, a, tb);
    else hipLaunchKernelGGL((yw_window_kernel<NT, false, 2>), grid, dim3(256), 0, st, a, tb);
  }
  // back substitution + emit: one workgroup per window, one launch (round 1: p - 1 pivot-block launches, each
  // re-reading and re-writing the Z tiles it updates, then two emit launches)
  hipLaunchKernelGGL((yw_window_kernel<NT, false, 1>), dim3(n), dim3(256), 0, st, a, 0);
  return (int)hipGetLastError();
}

int launch_yw(const YwArgs& a_in, int m_pad, hipStream_t st) {
  YwArgs a = a_in;
  a.only_guarded = 0;
  if (a.n_items == 0) return 0;
  // No LDL^T form asked for: the block Levinson-Whittle recursion (half the tile products, a quarter of the state), then
  // the one-launch LDL^T over the windows whose tile inverses tripped the conditioning guard (normally none: every other
  // workgroup of that launch reads one int and exits).
  const long long form = tuning(4 /* HMV_TUNE_YW_FORM */);
  if (a.tiled < 0 && form != 1) {
    int rc = launch_yw_lwr(a, m_pad, st);
    if (rc) return rc;
    a.only_guarded = 1;
    a.tiled = 0;
  }
  // One launch per batch is the low-latency form (small batches, small tiles).  At 64 channels and hundreds of
  // windows both forms are bound by the same HBM traffic (~14 MB of tile operands per window, far more than any
  // cache holds for a resident batch) and the launch chain, whose every launch streams with the whole chip, is the
  // faster one: 2.0 ms against 2.26 ms for 599 windows (profiles/r02_ab_notes.md).
  const bool one_launch = a.tiled == 0 || (a.tiled < 0 && !(m_pad == 64 && a.n_items >= 128));
  if (one_launch) {
    if (!a.only_guarded)
      if (const hipError_t e = hipMemsetAsync(a.info, 0, sizeof(int) * a.n_items, st)) return (int)e;
    const dim3 grid((unsigned)a.n_items), block(256);
    const bool vq = (a.Vq_logdet != nullptr);
    switch (m_pad) {
      case 16: if (vq) hipLaunchKernelGGL((yw_window_kernel<1, true>), grid, block, 0, st, a, 0);
               else hipLaunchKernelGGL((yw_window_kernel<1, false>), grid, block, 0, st, a, 0); break;
      case 32: if (vq) hipLaunchKernelGGL((yw_window_kernel<2, true>), grid, block, 0, st, a, 0);
               else hipLaunchKernelGGL((yw_window_kernel<2, false>), grid, block, 0, st, a, 0); break;
      case 48: if (vq) hipLaunchKernelGGL((yw_window_kernel<3, true>), grid, block, 0, st, a, 0);
               else hipLaunchKernelGGL((yw_window_kernel<3, false>), grid, block, 0, st, a, 0); break;
      case 64: if (vq) hipLaunchKernelGGL((yw_window_kernel<4, true>), grid, block, 0, st, a, 0);
               else hipLaunchKernelGGL((yw_window_kernel<4, false>), grid, block, 0, st, a, 0); break;
      default: return -1;
    }
    return (int)hipGetLastError();
  }
  switch (m_pad) {
    case 16: return launch_yw_nt<1>(a, st);
    case 32: return launch_yw_nt<2>(a, st);
    case 48: return launch_yw_nt<3>(a, st);
    case 64: return launch_yw_nt<4>(a, st);
    default: return -1;
  }
}

}  // namespace hmv
