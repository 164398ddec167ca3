// K4 -- full-frequency normalisation + layout change to the reference's (m, m, F) arrays.
//
// Replaces the m^2-iteration Python loop of `full_freq_dtf` (/root/reference/src/mtmvar.py:281-283):
//     ff[i, j, f] = dtf[i, j, f] / sum_{j', f'} dtf[i, j', f'].
// K3 leaves P[item][f][i][j] = |H_ij(f)|^2 (kernel-natural layout, j contiguous) and the per-(f, i)
// partial sums rowsum[item][f][i].  Here:
//   den_kernel      den[item][i] = sum_f rowsum[item][f][i]   (fixed order: four ascending quarter sums, then
//                   ((s0 + s1) + s2) + s3 -- bit-reproducible and identical to the fused normaliser of K3)
//   norm_kernel     out[item][i][j][f] = P[item][f][i][j] / den[item][i]  via a 64(f) x 64(j) LDS tile so
//                   that both the global reads (j contiguous) and the writes (f contiguous) are full
//                   512-byte runs.  HBM-bound: 2 x 8 B per output element.
// `normalise == 0` gives the plain |H|^2 of `dtf_multivariate` (mtmvar.py:232).
#include "hmv_common.h"
#include "hmv_kernels.h"

namespace hmv {

// block = 4 * m_pad threads: thread row q sums the q-th quarter of the frequency grid in ascending order, the
// four partial sums are added in the fixed order ((s0 + s1) + s2) + s3 -- the same scheme as the normaliser
// inside K3 (tf_inv.hip, window_denominators), so both give bit-identical denominators.
__global__ void __launch_bounds__(256) den_kernel(const double* rowsum, double* den, int F, int m_pad) {
  __shared__ double dpart[4 * 64];
  const long long item = blockIdx.x;
  const int ty = threadIdx.x / m_pad, i = threadIdx.x - ty * m_pad;
  const double* rs = rowsum + (size_t)item * F * m_pad + i;
  const int fq = (F + 3) >> 2;
  const int f1 = min(F, (ty + 1) * fq);
  double acc = 0.0;
  int f = ty * fq;
  for (; f + 16 <= f1; f += 16) {          // 16 loads in flight, summed in ascending f
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = rs[(size_t)(f + k) * m_pad];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += v[k];
  }
  for (; f < f1; ++f) acc += rs[(size_t)f * m_pad];
  dpart[ty * m_pad + i] = acc;
  __syncthreads();
  if (threadIdx.x < m_pad) {
    const int t = threadIdx.x;
    den[(size_t)item * m_pad + t] = ((dpart[t] + dpart[m_pad + t]) + dpart[2 * m_pad + t]) + dpart[3 * m_pad + t];
  }
}

// grid: ceil(F/64) * n_items * m; block 256
__global__ void __launch_bounds__(256) norm_kernel(NormArgs a) {
  __shared__ double tile[64][65];
  const int MP = a.m_pad, m = a.m, F = a.F;
  const int nft = (F + 63) / 64;
  // row index i fastest: 64 consecutive workgroups read the 64 adjacent 512-byte rows of one P[item][f]
  // matrix (one contiguous 32 KB region per f) instead of touching it 64 times far apart in time
  const int i = blockIdx.x % m;
  const long long tile_id = blockIdx.x / m;
  // newest first: K3 wrote the items in ascending order, the last ones may still sit in the memory-side cache
  const long long item = a.n_items - 1 - tile_id / nft;
  const int f0 = (int)(tile_id % nft) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const double* P = a.P + ((size_t)item * F * MP + (size_t)i) * MP;   // + f*MP*MP + j
  // read: rows f, columns j (contiguous); all 16 loads of a thread in flight before the first LDS write
  // (the kernel is bound by bytes in flight: 4 workgroups x 32 KB per CU cover the HBM latency)
  double stage[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int f = f0 + ty + 4 * k;
    stage[k] = (f < F && tx < m) ? __builtin_nontemporal_load(P + (size_t)f * MP * MP + tx) : 0.0;
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) tile[ty + 4 * k][tx] = stage[k];
  __syncthreads();
  // one reciprocal per output row, one multiply per element (the normaliser fused into K3 does exactly the same,
  // so both paths give the same bits; against a true division this is <= 1.5 ulp)
  double scale = 1.0;
  if (a.normalise) scale = 1.0 / a.den[(size_t)item * MP + i];
  double* out = a.out + (((size_t)item * m + i) * m) * F;             // + j*F + f
#pragma unroll
  for (int r = ty; r < 64; r += 4) {
    const int j = r;
    const int f = f0 + tx;
    if (j < m && f < F) {
      const double v = tile[tx][j];
      __builtin_nontemporal_store(v * scale, out + (size_t)j * F + f);
    }
  }
}

// complex [n_items][F][MP][MP] -> [n_items][m][m][F]; grid (ceil(F/32) * n_items, m), block 256
__global__ void __launch_bounds__(256) transpose_c128_kernel(const double2* in, double2* out, int F, int m, int MP) {
  __shared__ double2 tile[32][65];
  const int nft = (F + 31) / 32;
  const long long item = blockIdx.x / nft;
  const int f0 = (blockIdx.x % nft) * 32;
  const int i = blockIdx.y;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const double2* src = in + ((size_t)item * F * MP + (size_t)i) * MP;
  for (int r = ty; r < 32; r += 4) {
    const int f = f0 + r;
    double2 v = make_double2(0.0, 0.0);
    if (f < F && tx < m) v = src[(size_t)f * MP * MP + tx];
    tile[r][tx] = v;
  }
  __syncthreads();
  double2* dst = out + (((size_t)item * m + i) * m) * F;
  const int fx = threadIdx.x & 31, jy = threadIdx.x >> 5;   // 32 f x 8 j per pass
  for (int j = jy; j < 64; j += 8) {
    const int f = f0 + fx;
    if (j < m && f < F) dst[(size_t)j * F + f] = tile[fx][j];
  }
}

// ---- small elementwise companions of the path (nothing of this arithmetic is left to the host framework) -----
// out[e] = (in[0][e] + in[1][e] + ... + in[T-1][e]) / T : the trial average of count_corr
// (/root/reference/src/mtmvar.py:78-85: the totals are accumulated trial by trial, then divided by `trials`).
__global__ void __launch_bounds__(256) trial_mean_kernel(const double* in, double* out, long long n, int trials) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  double acc = 0.0;
  for (int t = 0; t < trials; ++t) acc = acc + in[(size_t)t * n + e];
  out[e] = (trials > 1) ? acc / (double)trials : acc;
}
// dDTF = ffDTF * |partial coherence| (mtmvar.py:341-385, `ff_dtf * np.abs(kappa)`), both (items, m, m, F).
__global__ void __launch_bounds__(256) ddtf_kernel(const double* ff, const double2* kappa, double* out, long long n) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const double2 k = kappa[e];
  out[e] = ff[e] * hypot(k.x, k.y);
}
// out[r][b] = sum of in[r][f] over the bins lo[b] <= f < hi[b], r = one (item, i, j) row of an (items, m, m, F)
// array, ascending f (the band-integrated product that is gathered across GPUs; the reference's graph plots sum
// ffDTF over a frequency range the same way, mtmvar.py:984-987).  One wave per row, 4 rows per workgroup.
__global__ void __launch_bounds__(256) band_sums_kernel(const double* in, const int* lo, const int* hi, double* out,
                                                        long long rows, int F, int nb) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int l = threadIdx.x & 63;
  const double* src = in + (size_t)r * F;
  for (int b = 0; b < nb; ++b) {
    double acc = 0.0;
    for (int f = lo[b] + l; f < hi[b]; f += 64) acc += src[f];      // lane partial sums, ascending f
    acc = row16_sum_dpp(acc);                                       // fixed-order tree over the 64 lanes
    const double t = ((readlane_f64(acc, 0) + readlane_f64(acc, 16)) + readlane_f64(acc, 32)) + readlane_f64(acc, 48);
    if (l == 0) out[(size_t)r * nb + b] = t;
  }
}

// The same sums for grids that are a multiple of 32 points, as a streaming kernel: 16 lanes per row, four rows per wave
// and pass, every row read once with 16-byte loads (the band membership comes from a 0 / 1 weight table in LDS, so a
// value costs two FMAs per band and no compare), one 16-lane DPP tree per band and row.  The first kernel reads the row
// once per band with 8-byte loads and reduces over 64 lanes: 1.87 ms for a 5 GB dyad against 0.9 ms of HBM time.  Lane
// partial sums run over ascending f; the tree order is fixed: deterministic, equal to the first kernel to rounding.
template <int NB>
__global__ void __launch_bounds__(256) band_sums_stream_kernel(const double* in, const int* lo, const int* hi, double* out,
                                                               long long rows, int F, int nb, int b0) {
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) double wgt[];          // [NB][F]
  for (int e = threadIdx.x; e < NB * F; e += 256) {
    const int b = e / F, f = e - b * F;
    wgt[e] = (b0 + b < nb && f >= lo[b0 + b] && f < hi[b0 + b]) ? 1.0 : 0.0;
  }
  __syncthreads();
  const int l = threadIdx.x & 63, q = l >> 4, c = l & 15;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (long long)gridDim.x * 4;
  const int nk = F / 32;
  for (long long r0 = wave * 4; r0 < rows; r0 += nwave * 4) {
    const long long r = r0 + q;
    const bool live = r < rows;
    const f64x2* src = reinterpret_cast<const f64x2*>(in + (size_t)(live ? r : 0) * F) + c;
    double acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.0;
    for (int k = 0; k < nk; ++k) {
      const f64x2 v = __builtin_nontemporal_load(src + 16 * k);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const f64x2 wv = *reinterpret_cast<const f64x2*>(wgt + b * F + 32 * k + 2 * c);
        acc[b] = __builtin_fma(v.y, wv.y, __builtin_fma(v.x, wv.x, acc[b]));
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const double t = row16_sum_dpp(acc[b]);
      if (c == 0 && live && b0 + b < nb) out[(size_t)r * nb + b0 + b] = t;
    }
  }
}

int launch_trial_mean(const double* in, double* out, long long n, int trials, hipStream_t st) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(trial_mean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n, trials);
  return (int)hipGetLastError();
}
int launch_ddtf(const double* ff, const double* kappa, double* out, long long n, hipStream_t st) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(ddtf_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ff,
                     reinterpret_cast<const double2*>(kappa), out, n);
  return (int)hipGetLastError();
}
int launch_band_sums(const double* in, const int* lo, const int* hi, double* out, long long rows, int F, int nb,
                     hipStream_t st) {
  if (rows == 0 || nb == 0) return 0;
  if (F % 32 == 0 && F <= 1024 && reinterpret_cast<uintptr_t>(in) % 16 == 0) {
    constexpr int NB = 8;
    const long long want = (rows + 15) / 16;
    const unsigned grid = (unsigned)(want < 4096 ? want : 4096);
    for (int b0 = 0; b0 < nb; b0 += NB)
      hipLaunchKernelGGL(band_sums_stream_kernel<NB>, dim3(grid), dim3(256), (size_t)NB * F * sizeof(double), st, in, lo, hi,
                         out, rows, F, nb, b0);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(band_sums_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, in, lo, hi, out, rows, F, nb);
  return (int)hipGetLastError();
}

int launch_ffdtf_norm(const NormArgs& a, hipStream_t st) {
  if (a.n_items == 0) return 0;
  if (a.normalise)
    hipLaunchKernelGGL(den_kernel, dim3((unsigned)a.n_items), dim3(4 * a.m_pad), 0, st, a.rowsum, a.den, a.F, a.m_pad);
  const dim3 grid((unsigned)(((a.F + 63) / 64) * a.n_items * a.m));
  hipLaunchKernelGGL(norm_kernel, grid, dim3(256), 0, st, a);
  return (int)hipGetLastError();
}

int launch_transpose_c128(const double* in, double* out, long long n_items, int F, int m, int m_pad, hipStream_t st) {
  if (n_items == 0) return 0;
  const dim3 grid((unsigned)(((F + 31) / 32) * n_items), m);
  hipLaunchKernelGGL(transpose_c128_kernel, grid, dim3(256), 0, st, reinterpret_cast<const double2*>(in),
                     reinterpret_cast<double2*>(out), F, m, m_pad);
  return (int)hipGetLastError();
}

}  // namespace hmv
