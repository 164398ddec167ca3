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
//   dpss_bisect_kernel   the Kmax largest eigenvalues by Sturm counts, 256 probes per eigenvalue and round (7 rounds of
//                        257-way multisection instead of ~55 bisection steps); d_i and e_i come from their closed
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

// number of eigenvalues < x  (LAPACK dstebz / dlaebz recurrence).  110 000 dependent steps per probe, eleven rounds of
// probes: what a step costs is what the kernel costs.  d_i and e_i come from running values (h -= 1, i + 1 += 1,
// M - 1 - i -= 1: small integers and half-integers, exact in double, so the SAME numbers as the closed forms), and the
// quotient e^2 / q is e^2 * (1 / q) with the hardware reciprocal and one Newton step (~1e-16 relative: it moves a Sturm
// count only where q is within rounding of zero, i.e. it perturbs the probe by an ulp) instead of the ~15-instruction
// IEEE division: 120 -> ~50 ms at 110 000 points.
__device__ __forceinline__ long long sturm_count(const DpssMat& T, double x, double pivmin) {
  double h = 0.5 * (double)(T.M - 1);               // (M - 1 - 2 i) / 2
  double fi = 1.0, gi = (double)(T.M - 1);          // i + 1, M - 1 - i
  double q = h * h * T.c2w - x;
  if (fabs(q) < pivmin) q = -pivmin;
  long long c = q < 0.0 ? 1 : 0;
  for (long long i = 1; i < T.M; ++i) {
    const double ee = 0.5 * fi * gi;                // e(i - 1)
    h -= 1.0;
    fi += 1.0;
    gi -= 1.0;
    double r = __builtin_amdgcn_rcp(q);
    r = __builtin_fma(__builtin_fma(-q, r, 1.0), r, r);
    q = h * h * T.c2w - x - ee * ee * r;
    if (fabs(q) < pivmin) q = -pivmin;
    c += q < 0.0 ? 1 : 0;
  }
  return c;
}

// one workgroup of 256 probes per eigenvalue: k-th largest, k = blockIdx.x.  257-way multisection until the bracket is
// below the spacing of doubles at the eigenvalue (seven rounds at 110 000 points: the Gershgorin interval is ~6e9 wide and
// the wanted eigenvalues ~3e9, so anything past a 2e16-fold reduction is below an ulp -- the 64-probe version ran eleven
// rounds, the last two inside one ulp).  Four waves per eigenvalue are free: 440 eigenvalues leave most SIMDs idle.
__global__ void __launch_bounds__(256) dpss_bisect_kernel(DpssMat T, int K, double glo, double ghi, double pivmin, double* lam) {
  __shared__ int s_first[4];
  const int k = blockIdx.x, t = threadIdx.x, w = t >> 6;
  if (k >= K) return;
  const long long idx = T.M - 1 - k;               // ascending index of the eigenvalue
  double lo = glo, hi = ghi;                       // count(lo) <= idx < count(hi)
  for (int round = 0; round < 12; ++round) {
    if (hi - lo <= 2.220446049250313e-16 * fmax(fabs(lo), fabs(hi))) break;      // workgroup-uniform
    const double step = (hi - lo) / 257.0;
    const double x = lo + step * (double)(t + 1);
    const long long c = sturm_count(T, x, pivmin);
    const unsigned long long above = __builtin_amdgcn_ballot_w64(c >= idx + 1);
    if ((t & 63) == 0) s_first[w] = above ? 64 * w + (int)__builtin_ctzll(above) : 256;
    __syncthreads();
    const int ts = min(min(s_first[0], s_first[1]), min(s_first[2], s_first[3]));   // first probe with count > idx
    __syncthreads();
    if (ts == 256) {
      lo = lo + step * 256.0;
    } else {
      hi = lo + step * (double)(ts + 1);
      lo = lo + step * (double)ts;
    }
  }
  if (t == 0) lam[k] = 0.5 * (lo + hi);
}

// One WAVE per eigenvector.  The arithmetic is LAPACK's dlagtf / dlagts(job = -1) as dstein uses them -- LU of T - lambda I
// with row interchanges between neighbours, then x <- (T - lambda I)^-1 x with tiny pivots perturbed, three times, the
// iterate rescaled before every solve -- i.e. first-order recurrences: nothing to parallelise inside one vector.  The
// first version ran one THREAD per vector over [i][k] arrays: ~440 threads = 7 waves on the whole chip, every block of
// 16 steps waiting ~2 us for its loads with nothing to hide them: 198 ms for 110 000 points.  Here every vector has a
// wave of its own (440 chains on 440 SIMDs): the 64 lanes fetch the next block of BL points of every array with coalesced
// loads ([k][i] arrays) while the recurrence walks the current block out of LDS, and store the finished block the same
// way; all lanes run the (uniform) recurrence, lane 0 writes.  Same operations in the same order per vector: same bits.
constexpr int BL = 256, DB = 16;
__global__ void __launch_bounds__(64) dpss_invit_kernel(DpssMat T, int K, const double* lam, double onenrm, double eps, double* a,
                                                        double* b, double* c, double* dd, unsigned char* in, double* x) {
  __shared__ double s_in[2][4][BL];          // double-buffered inputs of a block (x / a / b / d, or x / c / flag)
  __shared__ double s_out[5][BL];            // the block's results
  const int k = blockIdx.x, l = threadIdx.x;
  if (k >= K) return;
  const long long n = T.M;
  const double lambda = lam[k];
  double* ak_ = a + (size_t)k * n;
  double* bk_ = b + (size_t)k * n;
  double* ck_ = c + (size_t)k * n;
  double* dk_ = dd + (size_t)k * n;
  unsigned char* ik_ = in + (size_t)k * n;
  double* xk_ = x + (size_t)k * n;
  auto wsync = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  // ---- dlagtf (d_i, e_i from their closed forms: this pass only stores)
  double tol = 0.0, alast = 0.0;
  {
    double ak = T.d(0) - lambda;                                   // a[i] of the running elimination
    double bk = n > 1 ? T.e(0) : 0.0;                              // b[i] (superdiagonal), may be rewritten by an interchange
    double scale1 = fabs(ak) + fabs(bk);
    // d_i and e_i from running values instead of 64-bit integer conversions per step (h -= 1, i + 1 += 1, M - 1 - i -= 1:
    // half-integers and integers below 2^53, exact, so the same numbers as T.d / T.e)
    double hn = 0.5 * (double)(n - 1) - 1.0;                       // (M - 1 - 2 (i + 1)) / 2 at i = 0
    double f1 = 1.0, g1 = (double)(n - 1);                         // i + 1, M - 1 - i
    for (long long i0 = 0; i0 + 1 < n; i0 += BL) {
      const int cnt = (int)((n - 1 - i0 < BL) ? (n - 1 - i0) : BL);
      for (int u = 0; u < cnt; ++u) {
        const long long i = i0 + u;
        const double ci = 0.5 * f1 * g1;                             // subdiagonal c[i] = e(i)
        double a1 = hn * hn * T.c2w - lambda;                        // a[i+1] = d(i+1) - lambda
        f1 += 1.0;
        g1 -= 1.0;
        hn -= 1.0;
        double b1 = (i + 2 < n) ? 0.5 * f1 * g1 : 0.0;               // b[i+1] = e(i+1)
        const double scale2 = fabs(ci) + fabs(a1) + ((i + 2 < n) ? fabs(b1) : 0.0);
        double astore = ak, bstore = bk, cstore = 0.0, dstore = 0.0;
        double flag = 0.0;
        if (ci == 0.0) {
          scale1 = scale2;
        } else {
          // dlagtf compares piv2 = |c_i| / scale2 with piv1 = |a| / scale1 (0 if a == 0): the same decision from the
          // cross products (both scales are positive), two divisions fewer on a 110 000-step dependent chain; the two forms
          // can differ only when the candidates tie to rounding, where either pivot is as good
          if (fabs(ci) * scale1 <= fabs(ak) * scale2) {              // no interchange
            scale1 = scale2;
            cstore = ci / ak;
            a1 -= cstore * bk;
          } else {                                                   // rows i and i+1 change places
            flag = 1.0;
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
        if (l == 0) {
          s_out[0][u] = astore; s_out[1][u] = bstore; s_out[2][u] = cstore; s_out[3][u] = dstore; s_out[4][u] = flag;
        }
        tol = fmax(tol, fmax(fabs(astore), fmax(fabs(bstore), fabs(dstore))));      // dlagts' tolerance: largest factor entry
        ak = a1;
        bk = b1;
      }
      wsync();
      for (int u = l; u < cnt; u += 64) {
        ak_[i0 + u] = s_out[0][u]; bk_[i0 + u] = s_out[1][u]; ck_[i0 + u] = s_out[2][u]; dk_[i0 + u] = s_out[3][u];
        ik_[i0 + u] = (unsigned char)(s_out[4][u] != 0.0);
      }
      wsync();
    }
    if (l == 0) ak_[n - 1] = ak;
    alast = fabs(ak);
    tol = fmax(tol, alast) * eps;
    if (tol == 0.0) tol = eps;
  }
  // ---- start vector: fixed pseudo-random numbers in (-1, 1) (dstein: dlarnv uniform(-1, 1))
  double xmax = 0.0;
  {
    unsigned long long s = 0x9E3779B97F4A7C15ull * (unsigned long long)(k + 1);
    for (long long i0 = 0; i0 < n; i0 += BL) {
      const int cnt = (int)((n - i0 < BL) ? (n - i0) : BL);
      for (int u = 0; u < cnt; ++u) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const double v = (double)(long long)(s >> 11) * (2.0 / 9007199254740992.0) - 1.0;
        if (l == 0) s_out[0][u] = v;
        xmax = fmax(xmax, fabs(v));
      }
      wsync();
      for (int u = l; u < cnt; u += 64) xk_[i0 + u] = s_out[0][u];
      wsync();
    }
  }
  __threadfence_block();
  for (int it = 0; it < 3; ++it) {
    // scale the iterate as dstein does: |x|_max -> n * onenrm * max(eps, |a[n-1]|)
    const double scl = (double)n * onenrm * fmax(eps, alast) / xmax;
    // ---- dlagts, job = -1: forward substitution with the interchanges (the rescaling folded in).  Step i (1 .. n-1)
    // reads x[i], c[i-1], in[i-1] and fixes x[i-1]; blocks of BL steps, the next block's inputs fetched ahead.
    double yprev = xk_[0] * scl;
    {
      const long long nsteps = n - 1;                                // steps i = 1 .. n-1
      auto fetch = [&](double (&r)[3][BL / 64], long long i0) __attribute__((always_inline)) {       // step index i0 + u -> i = i0 + u + 1
#pragma unroll
        for (int q = 0; q < BL / 64; ++q) {
          const long long i = i0 + l + 64 * q + 1;
          const bool ok = i < n;
          r[0][q] = ok ? xk_[i] : 0.0;
          r[1][q] = ok ? ck_[i - 1] : 0.0;
          r[2][q] = ok ? (double)ik_[i - 1] : 0.0;
        }
      };
      auto stash = [&](int bufi, const double (&r)[3][BL / 64]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < BL / 64; ++q) {
          s_in[bufi][0][l + 64 * q] = r[0][q];
          s_in[bufi][1][l + 64 * q] = r[1][q];
          s_in[bufi][2][l + 64 * q] = r[2][q];
        }
      };
      double r[3][BL / 64];
      fetch(r, 0);
      stash(0, r);
      wsync();
      int bufi = 0;
      for (long long i0 = 0; i0 < nsteps; i0 += BL, bufi ^= 1) {
        const int cnt = (int)((nsteps - i0 < BL) ? (nsteps - i0) : BL);
        const bool more = i0 + BL < nsteps;
        if (more) fetch(r, i0 + BL);                                 // in flight behind this block's recurrence
        for (int u0 = 0; u0 < cnt; u0 += DB) {
          double xv[DB], cv[DB], fv[DB];
#pragma unroll
          for (int u = 0; u < DB; ++u) {
            const int uu = (u0 + u < BL) ? u0 + u : BL - 1;
            xv[u] = s_in[bufi][0][uu]; cv[u] = s_in[bufi][1][uu]; fv[u] = s_in[bufi][2][uu];
          }
#pragma unroll
          for (int u = 0; u < DB; ++u) {
            if (u0 + u < cnt) {
              double yi = xv[u] * scl;
              double outv;
              if (fv[u] == 0.0) {
                yi -= cv[u] * yprev;
                outv = yprev;
              } else {
                const double temp = yprev;
                outv = yi;
                yi = temp - cv[u] * yi;
              }
              if (l == 0) s_out[0][u0 + u] = outv;                   // x[i - 1]
              yprev = yi;
            }
          }
        }
        if (more) stash(bufi ^ 1, r);
        wsync();
        for (int u = l; u < cnt; u += 64) xk_[i0 + u] = s_out[0][u];  // x[i-1], i = i0 + u + 1
        wsync();
      }
      if (l == 0) xk_[n - 1] = yprev;
    }
    __threadfence_block();
    wsync();
    // ---- back substitution, tiny pivots replaced by +-tol (doubled until the quotient is representable); i = n-1 .. 0
    double y1 = 0.0, y2 = 0.0;                                      // x[i+1], x[i+2]
    xmax = 0.0;
    {
      auto fetch = [&](double (&r)[4][BL / 64], long long ib) __attribute__((always_inline)) {       // block = i in (ib - BL, ib], u = ib - i
#pragma unroll
        for (int q = 0; q < BL / 64; ++q) {
          const long long i = ib - (l + 64 * q);
          const bool ok = i >= 0;
          r[0][q] = ok ? xk_[i] : 0.0;
          r[1][q] = ok ? ak_[i] : 1.0;
          r[2][q] = (ok && i + 1 < n) ? bk_[i] : 0.0;
          r[3][q] = (ok && i + 2 < n) ? dk_[i] : 0.0;
        }
      };
      auto stash = [&](int bufi, const double (&r)[4][BL / 64]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < BL / 64; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) s_in[bufi][j][l + 64 * q] = r[j][q];
      };
      double r[4][BL / 64];
      fetch(r, n - 1);
      stash(0, r);
      wsync();
      int bufi = 0;
      for (long long ib = n - 1; ib >= 0; ib -= BL, bufi ^= 1) {
        const int cnt = (int)((ib + 1 < BL) ? (ib + 1) : BL);
        const bool more = ib - BL >= 0;
        if (more) fetch(r, ib - BL);
        for (int u0 = 0; u0 < cnt; u0 += DB) {
          double xv[DB], av[DB], bv[DB], dv[DB];
#pragma unroll
          for (int u = 0; u < DB; ++u) {
            const int uu = (u0 + u < BL) ? u0 + u : BL - 1;
            xv[u] = s_in[bufi][0][uu]; av[u] = s_in[bufi][1][uu]; bv[u] = s_in[bufi][2][uu]; dv[u] = s_in[bufi][3][uu];
          }
#pragma unroll
          for (int u = 0; u < DB; ++u) {
            if (u0 + u < cnt) {
              const double temp = xv[u] - bv[u] * y1 - dv[u] * y2;
              double ak = av[u];
              double pert = copysign(tol, ak);
              while (fabs(ak) < 1.0 && (fabs(ak) < 1e-300 ? true : fabs(temp) > fabs(ak) * 1e300)) {
                ak += pert;
                pert *= 2.0;
              }
              const double xi = temp / ak;
              if (l == 0) s_out[0][u0 + u] = xi;
              xmax = fmax(xmax, fabs(xi));
              y2 = y1;
              y1 = xi;
            }
          }
        }
        if (more) stash(bufi ^ 1, r);
        wsync();
        for (int u = l; u < cnt; u += 64) xk_[ib - u] = s_out[0][u];
        wsync();
      }
    }
    __threadfence_block();
    wsync();
  }
}

// one workgroup per taper: 2-norm, sign convention, [i][KP] -> [k][M] in `full` (all M points, for the ratios) and the
// first Mout points of each in `tapers` (sym=False: the periodic window is the symmetric one of M + 1 points cut short)
__global__ void __launch_bounds__(256) dpss_finish_kernel(const double* x, long long M, int KP, long long Mout, double* full,
                                                          double* tapers) {
  (void)KP;                                // (x is [k][M] since the wave-per-vector inverse iteration)
  __shared__ double part[256];
  __shared__ long long firsts[256];
  const int k = blockIdx.x;
  double ss = 0.0;
  for (long long i = threadIdx.x; i < M; i += 256) {
    const double v = x[(size_t)k * M + i];
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
    for (long long i = threadIdx.x; i < M; i += 256) sm += x[(size_t)k * M + i];
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
      const double v = x[(size_t)k * M + i] * inv;
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
    if (f0 < M && x[(size_t)k * M + f0] < 0.0) sign = -1.0;
  }
  const double scale = sign * inv;
  for (long long i = threadIdx.x; i < M; i += 256) {
    const double v = x[(size_t)k * M + i] * scale;
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
  hipLaunchKernelGGL(dpss_bisect_kernel, dim3(K), dim3(256), 0, st, T, K, glo, ghi, pivmin, lam);
  hipLaunchKernelGGL(dpss_invit_kernel, dim3((unsigned)K), dim3(64), 0, st, T, K, lam, onenrm, eps, a, b, c, dd, in, x);
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
