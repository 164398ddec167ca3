// DPSS (Slepian) tapers on the GPU: the taper generator of the multitaper PSD.
//
// The reference's PSD (src/psd.py:7-33 -> mne.time_frequency.psd_array_multitaper) gets its tapers from
// scipy.signal.windows.dpss(M, NW, Kmax, sym=False, norm=2, return_ratios=True) (sym=False: the symmetric window of
// M + 1 points without its last sample -- 2-norm, signs and ratios all belong to the M + 1 points), LAPACK dstebz + dstein
// on the symmetric tridiagonal matrix that commutes with the concentration operator (Percival & Walden 1993):
//     d_i = ((M - 1 - 2 i) / 2)^2 cos(2 pi W),  i = 0..M-1;    e_i = (i + 1)(M - 1 - i) / 2,  i = 0..M-2;   W = NW / M.
// On the host that costs 55 s for a 220-s segment at 500 Hz (M = 110 000, 438 tapers) against 76 ms for the whole
// PSD on the GPU.  The same algorithm here, restated for the device (PARITY UNPINNED like the rest of the PSD leg:
// checked against SciPy's own dpss in tests/test_psd.py, ~1e-10):
//   dpss_bisect_kernel   the Kmax largest eigenvalues by Sturm counts, one wave per eigenvalue, 64-way multisection
//                        (11 rounds of 64 probes instead of ~55 bisection steps); d_i and e_i come from their closed
//                        forms, nothing is read from memory;
//   dpss_invit_kernel    eigenvectors by inverse iteration, one THREAD per eigenvector (a tridiagonal LU with partial
//                        pivoting and its solves are first-order recurrences: nothing to parallelise inside one vector,
//                        everything across the ~440 vectors); factors and iterates live in [i][k] arrays, so the 64
//                        lanes of a wave touch 512 contiguous bytes per step; three iterations from a fixed
//                        pseudo-random start, as dstein does after its convergence test;
//   dpss_finish_kernel   2-norm, SciPy's sign convention (even tapers: positive mean; odd tapers: positive first lobe),
//                        transposition to [k][M];
//   dpss_ratio_kernel    concentration ratios lambda_k = v_k^T B v_k, B_mn = sin(2 pi W (m-n)) / (pi (m-n)), through the
//                        spectra of the tapers and of the kernel row (hipFFT): (1/L) sum_f |V_k(f)|^2 Bhat(f)
//                        -- SciPy's "autocorrelation technique" without the inverse transforms.
#include "hmv_common.h"
#include "hmv_kernels.h"
#include <hipfft/hipfft.h>
#include <cmath>

namespace hmv {

namespace {

struct DpssMat {
  long long M;
  double c2w;               // cos(2 pi W)
  __device__ __forceinline__ double d(long long i) const {
    const double h = 0.5 * (double)(M - 1 - 2 * i);
    return h * h * c2w;
  }
  __device__ __forceinline__ double e(long long i) const { return 0.5 * (double)(i + 1) * (double)(M - 1 - i); }
};

// number of eigenvalues < x  (LAPACK dstebz / dlaebz recurrence)
__device__ __forceinline__ long long sturm_count(const DpssMat& T, double x, double pivmin) {
  double q = T.d(0) - x;
  if (fabs(q) < pivmin) q = -pivmin;
  long long c = q < 0.0 ? 1 : 0;
  for (long long i = 1; i < T.M; ++i) {
    const double ee = T.e(i - 1);
    q = T.d(i) - x - ee * ee / q;
    if (fabs(q) < pivmin) q = -pivmin;
    c += q < 0.0 ? 1 : 0;
  }
  return c;
}

// one wave per eigenvalue: k-th largest, k = blockIdx.x
__global__ void __launch_bounds__(64) dpss_bisect_kernel(DpssMat T, int K, double glo, double ghi, double pivmin, double* lam) {
  const int k = blockIdx.x, t = threadIdx.x;
  if (k >= K) return;
  const long long idx = T.M - 1 - k;               // ascending index of the eigenvalue
  double lo = glo, hi = ghi;                       // count(lo) <= idx < count(hi)
  for (int round = 0; round < 11; ++round) {
    const double step = (hi - lo) / 65.0;
    const double x = lo + step * (double)(t + 1);
    const long long c = sturm_count(T, x, pivmin);
    const unsigned long long above = __builtin_amdgcn_ballot_w64(c >= idx + 1);
    if (above == 0ull) {
      lo = lo + step * 64.0;
    } else {
      const int ts = (int)__builtin_ctzll(above);
      hi = lo + step * (double)(ts + 1);
      lo = lo + step * (double)ts;
    }
  }
  if (t == 0) lam[k] = 0.5 * (lo + hi);
}

// One thread per eigenvector.  Arrays are [i][KP] (eigenvector index fastest).  Follows LAPACK dlagtf / dlagts(job=-1)
// as used by dstein: LU of T - lambda I with row interchanges between neighbours, then x <- (T - lambda I)^-1 x with
// tiny pivots perturbed, three times, the iterate rescaled before every solve.  Only a handful of waves run, so
// nothing hides memory latency: the recurrences are walked in blocks of DB steps whose loads are all issued first.
constexpr int DB = 16;
__global__ void __launch_bounds__(64) dpss_invit_kernel(DpssMat T, int K, int KP, const double* lam, double onenrm, double eps,
                                                        double* a, double* b, double* c, double* dd, unsigned char* in,
                                                        double* x) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= K) return;
  const long long n = T.M;
  const double lambda = lam[k];
  auto at = [&](long long i) { return (size_t)i * KP + k; };
  // ---- dlagtf (d_i, e_i from their closed forms: this pass only stores)
  double tol = 0.0, alast = 0.0;
  {
    double ak = T.d(0) - lambda;                                   // a[i] of the running elimination
    double bk = n > 1 ? T.e(0) : 0.0;                              // b[i] (superdiagonal), may be rewritten by an interchange
    double scale1 = fabs(ak) + fabs(bk);
    for (long long i = 0; i + 1 < n; ++i) {
      const double ci = T.e(i);                                    // subdiagonal c[i]
      double a1 = T.d(i + 1) - lambda;                             // a[i+1]
      double b1 = (i + 2 < n) ? T.e(i + 1) : 0.0;                  // b[i+1]
      const double scale2 = fabs(ci) + fabs(a1) + ((i + 2 < n) ? fabs(b1) : 0.0);
      const double piv1 = (ak == 0.0) ? 0.0 : fabs(ak) / scale1;
      double astore = ak, bstore = bk, cstore = 0.0, dstore = 0.0;
      unsigned char flag = 0;
      if (ci == 0.0) {
        scale1 = scale2;
      } else {
        const double piv2 = fabs(ci) / scale2;
        if (piv2 <= piv1) {                                        // no interchange
          scale1 = scale2;
          cstore = ci / ak;
          a1 -= cstore * bk;
        } else {                                                   // rows i and i+1 change places
          flag = 1;
          const double mult = ak / ci;
          astore = ci;
          const double temp = a1;
          a1 = bk - mult * temp;
          if (i + 2 < n) {
            dstore = b1;
            b1 = -mult * dstore;
          }
          bstore = temp;
          cstore = mult;
        }
      }
      a[at(i)] = astore; b[at(i)] = bstore; c[at(i)] = cstore; dd[at(i)] = dstore; in[at(i)] = flag;
      tol = fmax(tol, fmax(fabs(astore), fmax(fabs(bstore), fabs(dstore))));      // dlagts' tolerance: largest factor entry
      ak = a1;
      bk = b1;
    }
    a[at(n - 1)] = ak;
    alast = fabs(ak);
    tol = fmax(tol, alast) * eps;
    if (tol == 0.0) tol = eps;
  }
  // ---- start vector: fixed pseudo-random numbers in (-1, 1) (dstein: dlarnv uniform(-1, 1))
  double xmax = 0.0;
  unsigned long long s = 0x9E3779B97F4A7C15ull * (unsigned long long)(k + 1);
  for (long long i = 0; i < n; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const double v = (double)(long long)(s >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    x[at(i)] = v;
    xmax = fmax(xmax, fabs(v));
  }
  for (int it = 0; it < 3; ++it) {
    // scale the iterate as dstein does: |x|_max -> n * onenrm * max(eps, |a[n-1]|)
    const double scl = (double)n * onenrm * fmax(eps, alast) / xmax;
    // ---- dlagts, job = -1: forward substitution with the interchanges (the rescaling folded in)
    double yprev = x[at(0)] * scl;
    for (long long i0 = 1; i0 < n; i0 += DB) {
      double xv[DB], cv[DB];
      unsigned char fv[DB];
#pragma unroll
      for (int u = 0; u < DB; ++u) {
        const long long i = i0 + u;
        const bool ok = i < n;
        xv[u] = ok ? x[at(i)] : 0.0;
        cv[u] = ok ? c[at(i - 1)] : 0.0;
        fv[u] = ok ? in[at(i - 1)] : 0;
      }
#pragma unroll
      for (int u = 0; u < DB; ++u) {
        const long long i = i0 + u;
        if (i < n) {
          double yi = xv[u] * scl;
          if (fv[u] == 0) {
            yi -= cv[u] * yprev;
            x[at(i - 1)] = yprev;
          } else {
            const double temp = yprev;
            x[at(i - 1)] = yi;
            yi = temp - cv[u] * yi;
          }
          yprev = yi;
        }
      }
    }
    x[at(n - 1)] = yprev;
    // back substitution, tiny pivots replaced by +-tol (doubled until the quotient is representable)
    double y1 = 0.0, y2 = 0.0;                                      // x[i+1], x[i+2]
    xmax = 0.0;
    for (long long i0 = n - 1; i0 >= 0; i0 -= DB) {
      double xv[DB], av[DB], bv[DB], dv[DB];
#pragma unroll
      for (int u = 0; u < DB; ++u) {
        const long long i = i0 - u;
        const bool ok = i >= 0;
        xv[u] = ok ? x[at(i)] : 0.0;
        av[u] = ok ? a[at(i)] : 1.0;
        bv[u] = (ok && i + 1 < n) ? b[at(i)] : 0.0;
        dv[u] = (ok && i + 2 < n) ? dd[at(i)] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < DB; ++u) {
        const long long i = i0 - u;
        if (i >= 0) {
          const double temp = xv[u] - bv[u] * y1 - dv[u] * y2;
          double ak = av[u];
          double pert = copysign(tol, ak);
          while (fabs(ak) < 1.0 && (fabs(ak) < 1e-300 ? true : fabs(temp) > fabs(ak) * 1e300)) {
            ak += pert;
            pert *= 2.0;
          }
          const double xi = temp / ak;
          x[at(i)] = xi;
          xmax = fmax(xmax, fabs(xi));
          y2 = y1;
          y1 = xi;
        }
      }
    }
  }
}

// one workgroup per taper: 2-norm, sign convention, [i][KP] -> [k][M] in `full` (all M points, for the ratios) and the
// first Mout points of each in `tapers` (sym=False: the periodic window is the symmetric one of M + 1 points cut short)
__global__ void __launch_bounds__(256) dpss_finish_kernel(const double* x, long long M, int KP, long long Mout, double* full,
                                                          double* tapers) {
  __shared__ double part[256];
  __shared__ long long firsts[256];
  const int k = blockIdx.x;
  double ss = 0.0;
  for (long long i = threadIdx.x; i < M; i += 256) {
    const double v = x[(size_t)i * KP + k];
    ss += v * v;
  }
  part[threadIdx.x] = ss;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  const double inv = 1.0 / sqrt(part[0]);
  __syncthreads();
  double sign = 1.0;
  if ((k & 1) == 0) {                      // symmetric tapers: positive average
    double sm = 0.0;
    for (long long i = threadIdx.x; i < M; i += 256) sm += x[(size_t)i * KP + k];
    part[threadIdx.x] = sm;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
      __syncthreads();
    }
    if (part[0] < 0.0) sign = -1.0;
  } else {                                 // antisymmetric tapers: the first point above the noise is positive
    const double thresh = fmax(1e-7, 1.0 / (double)M);
    long long first = M;
    for (long long i = threadIdx.x; i < M; i += 256) {
      const double v = x[(size_t)i * KP + k] * inv;
      if (v * v > thresh) {
        first = i;
        break;
      }
    }
    firsts[threadIdx.x] = first;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) firsts[threadIdx.x] = min(firsts[threadIdx.x], firsts[threadIdx.x + s]);
      __syncthreads();
    }
    const long long f0 = firsts[0];
    if (f0 < M && x[(size_t)f0 * KP + k] < 0.0) sign = -1.0;
  }
  const double scale = sign * inv;
  for (long long i = threadIdx.x; i < M; i += 256) {
    const double v = x[(size_t)i * KP + k] * scale;
    full[(size_t)k * M + i] = v;
    if (i < Mout) tapers[(size_t)k * Mout + i] = v;
  }
}

// padded FFT inputs: rows 0..K-1 the tapers, row K the concentration kernel b[n] = sin(2 pi W n) / (pi n) (even, wrapped)
__global__ void __launch_bounds__(256) dpss_pad_kernel(const double* tapers, long long M, int K, double W, long long L, double* y) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= L) return;
  const int r = blockIdx.y;
  double v = 0.0;
  if (r < K) {
    if (t < M) v = tapers[(size_t)r * M + t];
  } else {
    const long long nn = (t < M) ? t : ((L - t < M) ? L - t : -1);
    if (nn == 0) v = 2.0 * W;
    else if (nn > 0) v = sin(2.0 * 3.14159265358979323846 * W * (double)nn) / (3.14159265358979323846 * (double)nn);
  }
  y[(size_t)r * L + t] = v;
}

// ratio_k = (1/L) sum_f w_f |V_k(f)|^2 Re Bhat(f), one-sided spectra of an even length L (w = 1, 2, ..., 2, 1)
__global__ void __launch_bounds__(256) dpss_ratio_kernel(const double2* X, long long nf, long long L, int K, double* ratios) {
  __shared__ double part[256];
  const int k = blockIdx.x;
  const double2* V = X + (size_t)k * nf;
  const double2* B = X + (size_t)K * nf;
  double acc = 0.0;
  for (long long f = threadIdx.x; f < nf; f += 256) {
    const double2 v = V[f];
    const double w = (f == 0 || f == nf - 1) ? 1.0 : 2.0;
    acc += w * (v.x * v.x + v.y * v.y) * B[f].x;
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) ratios[k] = part[0] / (double)L;
}

long long fft_len(long long M) {
  long long L = 2;
  while (L < 2 * M) L <<= 1;
  return L;
}
size_t al256(size_t b) { return (b + 255) & ~size_t(255); }

}  // namespace

long long dpss_workspace_bytes(long long M_out, int K, int sym) {
  const long long M = sym ? M_out : M_out + 1;
  const long long KP = ((K + 63) / 64) * 64, L = fft_len(M), nf = L / 2 + 1;
  size_t o = 0;
  o += al256(sizeof(double) * K);                               // eigenvalues of the tridiagonal matrix
  o += 5 * al256(sizeof(double) * (size_t)M * KP);              // a, b, c, d, x
  o += al256((size_t)M * KP);                                   // interchange flags
  const size_t fft = al256(sizeof(double) * (size_t)(K + 1) * L) + al256(sizeof(double) * 2 * (size_t)(K + 1) * nf);
  // the FFT buffers reuse three of the factor arrays (dead by then; the fourth holds the full-length windows) when they fit
  const size_t fac = 3 * al256(sizeof(double) * (size_t)M * KP);
  if (fft > fac) o += fft - fac;
  return (long long)o;
}

int launch_dpss(long long M_out, double NW, int K, int sym, double* tapers, double* ratios, void* workspace, hipStream_t st) {
  if (M_out < 2 || K < 1 || K > M_out || !(NW > 0.0) || NW >= 0.5 * (double)M_out) return -2;
  const long long M = sym ? M_out : M_out + 1;      // scipy's _extend(): solve M + 1 points, drop the last
  const long long KP = ((K + 63) / 64) * 64, L = fft_len(M), nf = L / 2 + 1;
  const double W = NW / (double)M;
  DpssMat T{M, cos(2.0 * 3.14159265358979323846 * W)};
  // Gershgorin interval and 1-norm on the host (closed forms; O(M) doubles of arithmetic)
  double glo = 1e300, ghi = -1e300, onenrm = 0.0, emax = 0.0;
  for (long long i = 0; i < M; ++i) {
    const double h = 0.5 * (double)(M - 1 - 2 * i), d = h * h * T.c2w;
    const double el = i > 0 ? 0.5 * (double)i * (double)(M - i) : 0.0;
    const double er = i + 1 < M ? 0.5 * (double)(i + 1) * (double)(M - 1 - i) : 0.0;
    glo = fmin(glo, d - el - er);
    ghi = fmax(ghi, d + el + er);
    onenrm = fmax(onenrm, fabs(d) + el + er);
    emax = fmax(emax, er);
  }
  const double eps = 2.220446049250313e-16, safemin = 2.2250738585072014e-308;
  const double tnorm = fmax(fabs(glo), fabs(ghi));
  glo -= 2.0 * tnorm * eps * (double)M + 2.0 * safemin;
  ghi += 2.0 * tnorm * eps * (double)M + 2.0 * safemin;
  const double pivmin = safemin * fmax(1.0, emax * emax);
  char* base = static_cast<char*>(workspace);
  size_t o = 0;
  double* lam = reinterpret_cast<double*>(base + o);  o += al256(sizeof(double) * K);
  const size_t arr = al256(sizeof(double) * (size_t)M * KP);
  double* x = reinterpret_cast<double*>(base + o);    o += arr;
  double* a = reinterpret_cast<double*>(base + o);    o += arr;
  double* b = reinterpret_cast<double*>(base + o);    o += arr;
  double* c = reinterpret_cast<double*>(base + o);    o += arr;
  double* dd = reinterpret_cast<double*>(base + o);   o += arr;
  unsigned char* in = reinterpret_cast<unsigned char*>(base + o);
  hipLaunchKernelGGL(dpss_bisect_kernel, dim3(K), dim3(64), 0, st, T, K, glo, ghi, pivmin, lam);
  hipLaunchKernelGGL(dpss_invit_kernel, dim3((unsigned)(KP / 64)), dim3(64), 0, st, T, K, (int)KP, lam, onenrm, eps, a, b, c, dd,
                     in, x);
  double* full = a;                          // the factors are dead once x is final
  hipLaunchKernelGGL(dpss_finish_kernel, dim3(K), dim3(256), 0, st, x, M, (int)KP, M_out, full, tapers);
  if (ratios) {
    // FFT buffers over the (now dead) factor arrays b .. d (+ whatever dpss_workspace_bytes added behind them)
    double* y = b;
    double2* X = reinterpret_cast<double2*>(reinterpret_cast<char*>(b) + al256(sizeof(double) * (size_t)(K + 1) * L));
    hipLaunchKernelGGL(dpss_pad_kernel, dim3((unsigned)((L + 255) / 256), K + 1), dim3(256), 0, st, full, M, K, W, L, y);
    hipfftHandle plan;
    int len[1] = {(int)L};
    if (hipfftPlanMany(&plan, 1, len, nullptr, 1, (int)L, nullptr, 1, (int)nf, HIPFFT_D2Z, K + 1) != HIPFFT_SUCCESS) return -20;
    int rc = 0;
    if (hipfftSetStream(plan, st) != HIPFFT_SUCCESS) rc = -21;
    if (!rc && hipfftExecD2Z(plan, y, reinterpret_cast<hipfftDoubleComplex*>(X)) != HIPFFT_SUCCESS) rc = -22;
    if (!rc) hipLaunchKernelGGL(dpss_ratio_kernel, dim3(K), dim3(256), 0, st, X, nf, L, K, ratios);
    (void)hipStreamSynchronize(st);          // the plan's work area goes with the plan
    (void)hipfftDestroy(plan);
    if (rc) return rc;
  }
  return (int)hipGetLastError();
}

}  // namespace hmv
