// K5 -- multivariate spectra  S(f) = H(f) V H(f)^T  (plain transpose, NOT conjugate: quirk Q3).
//
// Replaces the per-frequency loop of `multivariate_spectra` (/root/reference/src/mtmvar.py:165-201,
// the product at :199).  One workgroup (4 waves) per (window, frequency): H (complex, from K3) and V
// (real, from K2) are staged in LDS, T = H V costs two real MP^3 GEMMs, S = T H^T four more, all on
// v_mfma_f64_4x4x4_4b_f64 with wave w owning row blocks w*NT.. of the output (same tile GEMM as K2).
#include "hmv_common.h"
#include "hmv_kernels.h"

namespace hmv {

template <int NT>
__global__ void __launch_bounds__(256) spectra_kernel(SpecArgs a) {
  constexpr int MP = 16 * NT, NIW = NT, NJ = NT;
  constexpr int S = (MP <= 38) ? 38 : 70;
  constexpr int TILE = MP * MP;
  __shared__ double L0[MP * S];   // Hr
  __shared__ double L1[MP * S];   // Hi
  __shared__ double L2[MP * S];   // V^T, then Tr
  __shared__ double L3[MP * S];   // Ti
  const int l = lane_id();
  const int wv = uni(threadIdx.x >> 6);
  const int i = l >> 4, cc = l & 15;
  const long long gw = blockIdx.x;                 // item * F + f
  const long long item = gw / a.F;
  const double2* H = reinterpret_cast<const double2*>(a.H) + (size_t)gw * TILE;
  const double* V = a.V + (size_t)item * TILE;

  for (int idx = threadIdx.x; idx < TILE; idx += 256) {
    const int row = idx / MP, col = idx - row * MP;
    const double2 h = H[idx];
    L0[row * S + col] = h.x;
    L1[row * S + col] = h.y;
    L2[col * S + row] = V[idx];
  }
  __syncthreads();

  auto gemm_nt = [&](double (&acc)[NIW][NJ], const double* Xs, const double* Ys, bool negate) {
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
        for (int J = 0; J < NJ; ++J)
          acc[ii][J] = negate ? mfma4_nega(av[ii], bv[J], acc[ii][J]) : mfma4(av[ii], bv[J], acc[ii][J]);
    }
  };

  double tr[NIW][NJ], ti[NIW][NJ];
#pragma unroll
  for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
    for (int J = 0; J < NJ; ++J) tr[ii][J] = ti[ii][J] = 0.0;
  gemm_nt(tr, L0, L2, false);   // Tr = Hr V
  gemm_nt(ti, L1, L2, false);   // Ti = Hi V
  __syncthreads();
#pragma unroll
  for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
    for (int J = 0; J < NJ; ++J) {
      const int o = (4 * (wv * NT + ii) + i) * S + 16 * J + cc;
      L2[o] = tr[ii][J];
      L3[o] = ti[ii][J];
    }
  __syncthreads();
  double sr[NIW][NJ], si[NIW][NJ];
#pragma unroll
  for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
    for (int J = 0; J < NJ; ++J) sr[ii][J] = si[ii][J] = 0.0;
  gemm_nt(sr, L2, L0, false);   // Tr Hr^T
  gemm_nt(sr, L3, L1, true);    // - Ti Hi^T
  gemm_nt(si, L2, L1, false);   // Tr Hi^T
  gemm_nt(si, L3, L0, false);   // + Ti Hr^T
  double2* So = reinterpret_cast<double2*>(a.S) + (size_t)gw * TILE;
#pragma unroll
  for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
    for (int J = 0; J < NJ; ++J)
      So[(size_t)(4 * (wv * NT + ii) + i) * MP + 16 * J + cc] = make_double2(sr[ii][J], si[ii][J]);
}

int launch_spectra(const SpecArgs& a, int m_pad, hipStream_t st) {
  const long long n = a.n_items * (long long)a.F;
  if (n == 0) return 0;
  const dim3 grid((unsigned)n), block(256);
  switch (m_pad) {
    case 16: hipLaunchKernelGGL(spectra_kernel<1>, grid, block, 0, st, a); break;
    case 32: hipLaunchKernelGGL(spectra_kernel<2>, grid, block, 0, st, a); break;
    case 48: hipLaunchKernelGGL(spectra_kernel<3>, grid, block, 0, st, a); break;
    case 64: hipLaunchKernelGGL(spectra_kernel<4>, grid, block, 0, st, a); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

}  // namespace hmv
