// Measures built on the per-frequency matrices of K3 / K5:
//   pack_c128_kernel   (items, m, m, F) complex -> kernel layout [item][f][MP][MP], identity on the padding
//   pcoh_kernel        partial coherence from the INVERSE spectral matrix (mtmvar.py:287-338)
//   gpdc_kernel        generalised partial directed coherence from A(f) and diag(V) (mtmvar.py:388-468)
// All HBM-bound elementwise / small-reduction kernels; the heavy part (the inversion) is K3 in its general mode.
#include "hmv_common.h"
#include "hmv_kernels.h"

namespace hmv {

// grid: (ceil(F/32) * n_items, m); block 256.  Inverse of transpose_c128_kernel.
__global__ void __launch_bounds__(256) pack_c128_kernel(const double2* in, double2* out, int F, int m, int MP) {
  __shared__ double2 tile[32][65];
  const int nft = (F + 31) / 32;
  const long long item = blockIdx.x / nft;
  const int f0 = (blockIdx.x % nft) * 32;
  const int i = blockIdx.y;                                        // row, 0 .. MP-1 (padding rows included)
  const int fx = threadIdx.x & 31, jy = threadIdx.x >> 5;
  if (i < m) {
    const double2* src = in + (((size_t)item * m + i) * m) * F;    // + j*F + f
    for (int j = jy; j < 64; j += 8) {
      const int f = f0 + fx;
      tile[fx][j] = (j < m && f < F) ? src[(size_t)j * F + f] : make_double2(0.0, 0.0);
    }
  }
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  double2* dst = out + ((size_t)item * F * MP + (size_t)i) * MP;   // + f*MP*MP + j
  for (int r = ty; r < 32; r += 4) {
    const int f = f0 + r;
    if (f < F && tx < MP) {
      double2 v = (i < m && tx < m) ? tile[r][tx] : make_double2((tx == i && i >= m) ? 1.0 : 0.0, 0.0);
      dst[(size_t)f * MP * MP + tx] = v;
    }
  }
}

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// principal square root of a complex number (numpy's branch: real part >= 0, sign of the imaginary part kept)
__device__ __forceinline__ double2 csqrt_principal(double2 z) {
  const double r = hypot(z.x, z.y);
  if (r == 0.0) return make_double2(0.0, z.y);
  double sr, si;
  if (z.x >= 0.0) {
    sr = sqrt(0.5 * (r + z.x));
    si = z.y / (2.0 * sr);
  } else {
    si = copysign(sqrt(0.5 * (r - z.x)), z.y);
    sr = z.y / (2.0 * si);
  }
  return make_double2(sr, si);
}
__device__ __forceinline__ double2 cdiv(double2 a, double2 b) {     // Smith's algorithm
  if (fabs(b.x) >= fabs(b.y)) {
    const double t = b.y / b.x, d = b.x + b.y * t;
    return make_double2((a.x + a.y * t) / d, (a.y - a.x * t) / d);
  }
  const double t = b.x / b.y, d = b.x * t + b.y;
  return make_double2((a.x * t + a.y) / d, (a.y * t - a.x) / d);
}

// Partial coherence.  The reference takes minors M_ij = det(S without row i, column j) and returns
// kappa_ij = M_ij / sqrt(M_ii M_jj) (1 on the diagonal, 0 where the denominator vanishes).  With
// M_ij = (-1)^(i+j) det(S) (S^-1)_ji and det(S) = r u (r > 0, |u| = 1) the positive factor r cancels:
//   kappa_ij = (-1)^(i+j) u (S^-1)_ji / sqrt(u^2 (S^-1)_ii (S^-1)_jj)        (same branch of the root).
// grid: n_items * F; block 256.  Sinv, out: complex [item][f][MP][MP]; detph: [item*F][2] = u.
__global__ void __launch_bounds__(256) pcoh_kernel(const double2* Sinv, const double* detph, double2* out, int m, int MP) {
  __shared__ double2 dg[64];
  const long long gw = blockIdx.x;
  const double2* Z = Sinv + (size_t)gw * MP * MP;
  double2* O = out + (size_t)gw * MP * MP;
  if (threadIdx.x < MP) dg[threadIdx.x] = Z[(size_t)threadIdx.x * MP + threadIdx.x];
  __syncthreads();
  const double2 u = make_double2(detph[2 * gw], detph[2 * gw + 1]);
  const double2 u2 = cmul(u, u);
  for (int e = threadIdx.x; e < MP * MP; e += 256) {
    const int i = e / MP, j = e - i * MP;
    double2 k = make_double2(0.0, 0.0);
    if (i < m && j < m) {
      if (i == j) {
        k = make_double2(1.0, 0.0);
      } else {
        const double2 den = csqrt_principal(cmul(u2, cmul(dg[i], dg[j])));
        if (den.x != 0.0 || den.y != 0.0) {
          double2 num = cmul(u, Z[(size_t)j * MP + i]);
          if ((i + j) & 1) num = make_double2(-num.x, -num.y);
          k = cdiv(num, den);
        }
      }
    }
    O[e] = k;
  }
}

// GPDC_ij(f) = (|A_ij| / sigma_i) / sqrt(sum_k |A_kj|^2 / sigma_k^2), 0 where the denominator vanishes.
// grid: n_items * F; block 256.  A complex [item][f][MP][MP], V [item][MP][MP], out real [item][f][MP][MP].
__global__ void __launch_bounds__(256) gpdc_kernel(const double2* A, const double* V, double* out, int F, int m, int MP) {
  __shared__ double w[64][65];
  __shared__ double colsum[64];
  __shared__ double var[64];
  const long long gw = blockIdx.x;
  const long long item = gw / F;
  const double2* Z = A + (size_t)gw * MP * MP;
  if (threadIdx.x < MP) var[threadIdx.x] = V[(size_t)item * MP * MP + (size_t)threadIdx.x * MP + threadIdx.x];
  __syncthreads();
  for (int e = threadIdx.x; e < MP * MP; e += 256) {
    const int i = e / MP, j = e - i * MP;
    const double2 v = Z[e];
    const double ab = hypot(v.x, v.y);
    w[i][j] = (i < m && j < m) ? ab : 0.0;
  }
  __syncthreads();
  if (threadIdx.x < MP) {
    const int j = threadIdx.x;
    double acc = 0.0;
    for (int k = 0; k < m; ++k) acc += (w[k][j] * w[k][j]) / var[k];     // fixed order
    colsum[j] = sqrt(acc);
  }
  __syncthreads();
  double* O = out + (size_t)gw * MP * MP;
  for (int e = threadIdx.x; e < MP * MP; e += 256) {
    const int i = e / MP, j = e - i * MP;
    double g = 0.0;
    if (i < m && j < m && colsum[j] != 0.0) g = (w[i][j] / sqrt(var[i])) / colsum[j];
    O[e] = g;
  }
}

int launch_pack_c128(const double* in, double* out, long long n_items, int F, int m, int m_pad, hipStream_t st) {
  if (n_items == 0 || F == 0) return 0;
  const dim3 grid((unsigned)(((F + 31) / 32) * n_items), m_pad);
  hipLaunchKernelGGL(pack_c128_kernel, grid, dim3(256), 0, st, reinterpret_cast<const double2*>(in),
                     reinterpret_cast<double2*>(out), F, m, m_pad);
  return (int)hipGetLastError();
}
int launch_pcoh(const double* Sinv, const double* detph, double* out, long long n_items, int F, int m, int m_pad, hipStream_t st) {
  if (n_items * F == 0) return 0;
  hipLaunchKernelGGL(pcoh_kernel, dim3((unsigned)(n_items * F)), dim3(256), 0, st, reinterpret_cast<const double2*>(Sinv),
                     detph, reinterpret_cast<double2*>(out), m, m_pad);
  return (int)hipGetLastError();
}
int launch_gpdc(const double* A, const double* V, double* out, long long n_items, int F, int m, int m_pad, hipStream_t st) {
  if (n_items * F == 0) return 0;
  hipLaunchKernelGGL(gpdc_kernel, dim3((unsigned)(n_items * F)), dim3(256), 0, st, reinterpret_cast<const double2*>(A), V,
                     out, F, m, m_pad);
  return (int)hipGetLastError();
}

}  // namespace hmv
