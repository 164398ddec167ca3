// K5 -- multivariate spectra  S(f) = H(f) V H(f)^T  (plain transpose, NOT conjugate: quirk Q3).
//
// Replaces the per-frequency loop of `multivariate_spectra` (/root/reference/src/mtmvar.py:165-201,
// the product at :199).  One workgroup (4 waves) per (window, frequency): H (complex, from K3) and V
// (real, from K2) are staged in LDS a quarter of the k range at a time, T = H V costs two real MP^3 GEMMs,
// S = T H^T four more, all on v_mfma_f64_4x4x4_4b_f64 with wave w owning row blocks w*NT.. of the output.
#include "hmv_common.h"
#include "hmv_kernels.h"

namespace hmv {

typedef double f64x2 __attribute__((ext_vector_type(2)));

// Operands go through LDS a quarter of the k range at a time (KQ = MP/4 columns): two A-operand images and two
// B-operand images of MP x KQ doubles (45 KB at 64 channels, so two workgroups share a CU; the round-1 kernel kept
// four whole MP x MP tiles = 143 KB, one workgroup per CU, nothing in flight behind the MFMAs: 16.3 ms per
// 599-window dyad against 7.4 ms of pure MFMA time).  The next quarter's global loads are in flight behind the MFMAs
// of the current one.
//   stage 1, T = H V:     per quarter  A0/A1 = Re/Im H[:, kq], B0 = V[kq, :]^T        Tr += A0 B0^T, Ti += A1 B0^T
//   stage 2, S = T H^T:   per quarter  A0/A1 = Tr/Ti[:, kq] (own row strip, from registers), B0/B1 = Re/Im H[:, kq]
//                                      Sr += A0 B0^T - A1 B1^T,  Si += A0 B1^T + A1 B0^T
// k ascends 0 .. MP-1 in every sum, as in the round-1 kernel: same bits.
// Symmetric V (the residual covariance of this library's own fit): S = H V H^T is complex SYMMETRIC, so stage 2 only
// computes the 4-row x 16-column units (r4, J) that touch the upper triangle, r4 <= 4 J + 3 -- 2 NT (NT + 1) of the
// 4 NT^2 units, 40 of 64 at 64 channels -- and every element above the diagonal is written to both places.  The units
// are dealt to the FOUR waves of the workgroup in contiguous runs of NT (NT + 1) / 2 (J descending, r4 ascending): every
// wave the same number of MFMAs, its A operands from any row strip of the T images (all in LDS), one or two distinct B
// column blocks.
template <int NT>
struct SymDeal {
  static constexpr int U = NT * (NT + 1) / 2;              // units per wave (4 waves whatever NT is)
  static constexpr int J_of(int g) { int J = NT - 1; while (g >= 4 * (J + 1)) { g -= 4 * (J + 1); --J; } return J; }
  static constexpr int r4_of(int g) { int J = NT - 1; while (g >= 4 * (J + 1)) { g -= 4 * (J + 1); --J; } return g; }
};

// Workgroups per CU of the symmetric form: its 40 (instead of 64) accumulators would let three fit, but only with 34
// spilled registers (168 VGPRs): measured 26.9 k windows/s (ffDTF + spectra, C2) against 28.4 k with two (same box)
#ifndef HMV_K5_SYM_WGS
#define HMV_K5_SYM_WGS 2
#endif
template <int NT, bool SYM>
__global__ void __launch_bounds__(256, SYM ? HMV_K5_SYM_WGS : 2) spectra_kernel(SpecArgs a) {
  constexpr int MP = 16 * NT, NIW = NT, NJ = NT, KQ = MP / 4, SQ = KQ + 6, TILE = MP * MP;
  constexpr int NH = (MP * KQ + 255) / 256;          // complex elements of an H quarter per thread
  constexpr int NVV = (MP * KQ / 2 + 255) / 256;     // 16-byte loads of a V quarter per thread
  __shared__ __attribute__((aligned(16))) double A0[MP * SQ], A1[MP * SQ], B0[MP * SQ], B1[MP * SQ];
  const int wv = uni(threadIdx.x >> 6);
  long long gw = blockIdx.x;                       // item * F + f
  const long long item = gw / a.F;
  if (a.S_mmf && a.F % 64 == 0) {
    // Output straight into the reference's (m, m, F) layout: one workgroup owns ONE frequency, so its 4096 elements are
    // 16 bytes each, F * 16 bytes apart -- eight consecutive frequencies make one 128-byte line.  Blocks b and b + 8 share
    // an XCD (and its L2) and are dispatched back to back: the map below gives them consecutive frequencies, so the
    // eight partial writes of a line meet in ONE L2 instead of in eight.
    const int r = (int)(gw - item * a.F), c = r & 7, b8 = (r >> 3) & 7, a64 = r >> 6;
    gw = item * a.F + (64 * a64 + 8 * c + b8);
  }
  const f64x2* H = reinterpret_cast<const f64x2*>(a.H) + (size_t)gw * TILE;
  const double* V = a.V + (size_t)item * TILE;
  // thread coordinates re-derived from an opaque lane id per helper (keeps the offsets out of long live ranges)
  auto lane = [&]() __attribute__((always_inline)) {
    int lo;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo));
    return lo;
  };
  auto fetch_h = [&](f64x2 (&v)[NH], int kq) __attribute__((always_inline)) {
    const int t0 = lane() + 64 * wv;
#pragma unroll
    for (int r = 0; r < NH; ++r) {
      const int idx = t0 + 256 * r;
      if (NH * 256 == MP * KQ || idx < MP * KQ) {
        const int row = idx / KQ, k = idx - row * KQ;
        v[r] = H[(size_t)row * MP + kq * KQ + k];
      }
    }
  };
  auto park_h = [&](double* re, double* im, const f64x2 (&v)[NH]) __attribute__((always_inline)) {
    const int t0 = lane() + 64 * wv;
#pragma unroll
    for (int r = 0; r < NH; ++r) {
      const int idx = t0 + 256 * r;
      if (NH * 256 == MP * KQ || idx < MP * KQ) {
        const int row = idx / KQ, k = idx - row * KQ;
        re[row * SQ + k] = v[r].x;
        im[row * SQ + k] = v[r].y;
      }
    }
  };
  // B0[col][k] = V[kq*KQ + k][col]
  auto fetch_v = [&](f64x2 (&v)[NVV], int kq) __attribute__((always_inline)) {
    const int t0 = lane() + 64 * wv;
#pragma unroll
    for (int r = 0; r < NVV; ++r) {
      const int idx = t0 + 256 * r;
      if (NVV * 256 == MP * KQ / 2 || idx < MP * KQ / 2) {
        const int k = idx / (MP / 2), c2 = idx - k * (MP / 2);
        v[r] = *reinterpret_cast<const f64x2*>(V + (size_t)(kq * KQ + k) * MP + 2 * c2);
      }
    }
  };
  auto park_v = [&](const f64x2 (&v)[NVV]) __attribute__((always_inline)) {
    const int t0 = lane() + 64 * wv;
#pragma unroll
    for (int r = 0; r < NVV; ++r) {
      const int idx = t0 + 256 * r;
      if (NVV * 256 == MP * KQ / 2 || idx < MP * KQ / 2) {
        const int k = idx / (MP / 2), c2 = idx - k * (MP / 2);
        B0[(2 * c2) * SQ + k] = v[r].x;
        B0[(2 * c2 + 1) * SQ + k] = v[r].y;
      }
    }
  };
  // this wave's row strip of a register tile, columns kq*KQ .. -> A image
  auto park_strip = [&](double* dst, const double (&v)[NIW][NJ], int kq) __attribute__((always_inline)) {
    const int l = lane(), i = l >> 4, cc = l & 15;
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) {
        const int col = 16 * J + cc - kq * KQ;
        if (col >= 0 && col < KQ) dst[(4 * (wv * NT + ii) + i) * SQ + col] = v[ii][J];
      }
  };
  auto gemm_q = [&](double (&acc)[NIW][NJ], const double* Xq, const double* Yq, bool negate) __attribute__((always_inline)) {
    const int l = lane();
    const double* xa = Xq + (4 * wv * NT + (l & 3)) * SQ + (l >> 4);
    const double* yb = Yq + (l & 15) * SQ + (l >> 4);
#pragma unroll
    for (int k0 = 0; k0 < KQ; k0 += 4) {
      double av[NIW], bv[NJ];
#pragma unroll
      for (int ii = 0; ii < NIW; ++ii) av[ii] = xa[4 * ii * SQ + k0];
#pragma unroll
      for (int J = 0; J < NJ; ++J) bv[J] = yb[16 * J * SQ + k0];
#pragma unroll
      for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
        for (int J = 0; J < NJ; ++J)
          acc[ii][J] = negate ? mfma4_nega(av[ii], bv[J], acc[ii][J]) : mfma4(av[ii], bv[J], acc[ii][J]);
    }
  };

  double tr[NIW][NJ], ti[NIW][NJ];
#pragma unroll
  for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
    for (int J = 0; J < NJ; ++J) tr[ii][J] = ti[ii][J] = 0.0;
  f64x2 hv[NH], vv[NVV];
  fetch_h(hv, 0);
  fetch_v(vv, 0);
  for (int kq = 0; kq < 4; ++kq) {                 // ---- stage 1
    __syncthreads();                               // the previous quarter has been read
    park_h(A0, A1, hv);
    park_v(vv);
    if (kq < 3) {
      fetch_h(hv, kq + 1);
      fetch_v(vv, kq + 1);
    } else {
      fetch_h(hv, 0);                              // first quarter of stage 2
    }
    __syncthreads();
    gemm_q(tr, A0, B0, false);                     // Tr += Hr V
    gemm_q(ti, A1, B0, false);                     // Ti += Hi V
  }
  if constexpr (SYM) {                             // ---- stage 2, upper triangle only
    using D = SymDeal<NT>;
    auto stage2 = [&](auto wc) __attribute__((always_inline)) {
      constexpr int W = decltype(wc)::value;
      double ur[D::U], ui[D::U];
#pragma unroll
      for (int u = 0; u < D::U; ++u) ur[u] = ui[u] = 0.0;
      for (int kq = 0; kq < 4; ++kq) {
        __syncthreads();
        park_strip(A0, tr, kq);
        park_strip(A1, ti, kq);
        park_h(B0, B1, hv);
        if (kq < 3) fetch_h(hv, kq + 1);
        __syncthreads();
        const int l = lane();
        const int ao = (l & 3) * SQ + (l >> 4), bo = (l & 15) * SQ + (l >> 4);
#pragma unroll
        for (int k0 = 0; k0 < KQ; k0 += 4) {
          static_for<D::U>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value, g = W * D::U + u, r4 = D::r4_of(g), J = D::J_of(g);
            const double a0 = A0[ao + 4 * r4 * SQ + k0], a1 = A1[ao + 4 * r4 * SQ + k0];
            const double b0 = B0[bo + 16 * J * SQ + k0], b1 = B1[bo + 16 * J * SQ + k0];
            ur[u] = mfma4(a0, b0, ur[u]);          // Tr Hr^T
            ur[u] = mfma4_nega(a1, b1, ur[u]);     // - Ti Hi^T
            ui[u] = mfma4(a0, b1, ui[u]);          // Tr Hi^T
            ui[u] = mfma4(a1, b0, ui[u]);          // + Ti Hr^T
          });
        }
      }
      const int l = lane(), i = l >> 4, cc = l & 15;
      const int m = a.m, f = (int)(gw - item * a.F);
      double2* So = reinterpret_cast<double2*>(a.S_mmf) + (size_t)item * m * m * a.F + f;
      static_for<D::U>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, g = W * D::U + u, r4 = D::r4_of(g), J = D::J_of(g);
        const int row = 4 * r4 + i, col = 16 * J + cc;
        if (row <= col && col < m) {
          const double2 v = make_double2(ur[u], ui[u]);
          So[((size_t)row * m + col) * a.F] = v;
          if (row < col) So[((size_t)col * m + row) * a.F] = v;
        }
      });
    };
    switch (wv) {
      case 0: stage2(std::integral_constant<int, 0>{}); break;
      case 1: stage2(std::integral_constant<int, 1>{}); break;
      case 2: stage2(std::integral_constant<int, 2>{}); break;
      default: stage2(std::integral_constant<int, 3>{}); break;
    }
    return;
  } else {
  double sr[NIW][NJ], si[NIW][NJ];
#pragma unroll
  for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
    for (int J = 0; J < NJ; ++J) sr[ii][J] = si[ii][J] = 0.0;
  for (int kq = 0; kq < 4; ++kq) {                 // ---- stage 2
    __syncthreads();
    park_strip(A0, tr, kq);
    park_strip(A1, ti, kq);
    park_h(B0, B1, hv);
    if (kq < 3) fetch_h(hv, kq + 1);
    __syncthreads();
    gemm_q(sr, A0, B0, false);                     // Tr Hr^T
    gemm_q(sr, A1, B1, true);                      // - Ti Hi^T
    gemm_q(si, A0, B1, false);                     // Tr Hi^T
    gemm_q(si, A1, B0, false);                     // + Ti Hr^T
  }
  const int l = lane(), i = l >> 4, cc = l & 15;
  if (a.S_mmf) {
    const int m = a.m, f = (int)(gw - item * a.F);
    double2* So = reinterpret_cast<double2*>(a.S_mmf) + (size_t)item * m * m * a.F + f;
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) {
        const int row = 4 * (wv * NT + ii) + i, col = 16 * J + cc;
        if (row < m && col < m) So[((size_t)row * m + col) * a.F] = make_double2(sr[ii][J], si[ii][J]);
      }
    return;
  }
  double2* So = reinterpret_cast<double2*>(a.S) + (size_t)gw * TILE;
#pragma unroll
  for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
    for (int J = 0; J < NJ; ++J)
      So[(size_t)(4 * (wv * NT + ii) + i) * MP + 16 * J + cc] = make_double2(sr[ii][J], si[ii][J]);
  }
}

int launch_spectra(const SpecArgs& a, int m_pad, hipStream_t st) {
  const long long n = a.n_items * (long long)a.F;
  if (n == 0) return 0;
  const dim3 grid((unsigned)n), block(256);
  if (a.sym && a.S_mmf) {
    switch (m_pad) {
      case 16: hipLaunchKernelGGL((spectra_kernel<1, true>), grid, block, 0, st, a); break;
      case 32: hipLaunchKernelGGL((spectra_kernel<2, true>), grid, block, 0, st, a); break;
      case 48: hipLaunchKernelGGL((spectra_kernel<3, true>), grid, block, 0, st, a); break;
      case 64: hipLaunchKernelGGL((spectra_kernel<4, true>), grid, block, 0, st, a); break;
      default: return -1;
    }
    return (int)hipGetLastError();
  }
  switch (m_pad) {
    case 16: hipLaunchKernelGGL((spectra_kernel<1, false>), grid, block, 0, st, a); break;
    case 32: hipLaunchKernelGGL((spectra_kernel<2, false>), grid, block, 0, st, a); break;
    case 48: hipLaunchKernelGGL((spectra_kernel<3, false>), grid, block, 0, st, a); break;
    case 64: hipLaunchKernelGGL((spectra_kernel<4, false>), grid, block, 0, st, a); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

}  // namespace hmv
