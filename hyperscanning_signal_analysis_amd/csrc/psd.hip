// Multitaper power spectral density (reference: src/psd.py:7-33 ->
// mne.time_frequency.psd_array_multitaper, mne==1.11.0, NOT available offline: parity unpinned, the algorithm
// is restated from its published description -- DPSS tapers with eigenvalue weights, one-sided periodogram):
//   y[ch][k][t] = (x[ch][t] - mean_t x[ch]) * taper[k][t]
//   X[ch][k][f] = rfft(y[ch][k])                                   (hipFFT / rocFFT, batched D2Z)
//   psd[ch][f]  = 2 / sum_k w_k^2 * sum_k w_k^2 |X[ch][k][f]|^2    for the bins lo..hi (w_k = sqrt(eigenvalue_k))
// The DC and Nyquist bins carry the usual 1/sqrt(2) amplitude factor of a one-sided spectrum.
// The tapers come from the host (scipy.signal.windows.dpss, as the reference's dependency does).
#include "hmv_common.h"
#include "hmv_kernels.h"
#include <hipfft/hipfft.h>
#include <map>
#include <mutex>
#include <utility>

namespace hmv {

__global__ void __launch_bounds__(256) psd_mean_kernel(const double* x, long long ld, long long n, double* mean) {
  __shared__ double part[256];
  const double* row = x + (size_t)blockIdx.x * ld;
  double acc = 0.0;
  for (long long t = threadIdx.x; t < n; t += 256) acc += row[t];
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {           // fixed tree: deterministic
    if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) mean[blockIdx.x] = part[0] / (double)n;
}

// grid (ceil(n/256), K, channels of the chunk)
__global__ void __launch_bounds__(256) psd_taper_kernel(const double* x, long long ld, const double* mean, const double* tapers,
                                                        long long n, int K, double* y) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int k = blockIdx.y, ch = blockIdx.z;
  y[((size_t)ch * K + k) * n + t] = (x[(size_t)ch * ld + t] - mean[ch]) * tapers[(size_t)k * n + t];
}

// grid (ceil(nb/256), channels of the chunk)
__global__ void __launch_bounds__(256) psd_power_kernel(const double2* X, const double* w, long long nfreq, long long n, int K,
                                                        long long lo, long long nb, double* psd, long long psd_ld) {
  const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
  if (b >= nb) return;
  const int ch = blockIdx.y;
  const long long f = lo + b;
  double acc = 0.0, wsum = 0.0;
  for (int k = 0; k < K; ++k) {                 // fixed order: deterministic
    const double2 v = X[((size_t)ch * K + k) * nfreq + f];
    const double w2 = w[k] * w[k];
    acc += w2 * (v.x * v.x + v.y * v.y);
    wsum += w2;
  }
  double scale = 2.0 / wsum;
  if (f == 0 || ((n & 1) == 0 && f == nfreq - 1)) scale *= 0.5;      // (1/sqrt 2)^2 on DC / Nyquist
  psd[(size_t)ch * psd_ld + b] = acc * scale;
}

// ---------------------------------------------------------------- awkward lengths: a pruned chirp-z transform
// mne transforms at n_fft = n_times, and a segment cut by event times has any length: 110 001 = 3 * 37 * 991 for a 220 s
// segment at 500 Hz (the cut is inclusive at both ends).  rocFFT then runs Bluestein's algorithm at 262 144 complex points,
// forward and back, for every (channel, taper) although the caller keeps 6 381 of the 55 001 bins (1 .. 30 Hz): 0.75 s
// per config-5 dyad, 96 % of its GPU time.  What runs instead when the length has a prime factor > 13: the bins
// j = -hi .. hi of the DFT of length n as a chirp-z transform (Bluestein restricted to the outputs that are wanted),
//     X[j0 + m] = W^(m^2/2) * sum_t [ z[t] W^(j0 t + t^2/2) ] W^(-(m - t)^2 / 2),     W = exp(-2 pi i / n),  j0 = -hi,
// i.e. one circular convolution of length L = 2^ceil(log2(n + 2 hi)) (131 072 here) -- and z carries TWO real tapered
// series, z = y_a + i y_b, which come apart afterwards as X_a[j] = (Z[j] + conj Z[-j]) / 2, X_b[j] = (Z[j] - conj Z[-j]) / 2i
// (that is what the negative bins are for).  A quarter of Bluestein's points per pair of tapers; the chirp phases are
// reduced modulo 2 n in integers, so they are exact.  Same definition, same bins, rounding-level differences.
struct CztPlan {
  double2* chirp = nullptr;     // [n]   W^(j0 t + t^2/2)
  double2* bf = nullptr;        // [L]   FFT_L of the circular image of W^(-n^2/2), scaled by 1/L
  double2* eout = nullptr;      // [2 hi + 1]  W^(m^2/2)
  long long L = 0;
};

__device__ __forceinline__ double2 unit_phase(long long num, long long den2) {      // exp(-i pi num / (den2 / 2)), num mod den2
  double s, c;
  sincospi(-(double)num / (double)(den2 / 2), &s, &c);
  return make_double2(c, s);
}
__global__ void __launch_bounds__(256) czt_tables_kernel(long long n, long long hi, long long L, double2* chirp, double2* b,
                                                         double2* eout) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n2 = 2 * n, M = 2 * hi + 1;
  if (idx < n) {                                   // chirp[t] = exp(-i pi (t^2 + 2 j0 t) / n), j0 = -hi
    long long u = (idx * idx) % n2 - (2 * hi * idx) % n2;
    u = ((u % n2) + n2) % n2;
    chirp[idx] = unit_phase(u, n2);
  }
  if (idx < L) {                                   // b_circ[q]: lag q for 0 <= q < M, lag q - L for q > L - n, else 0
    long long lag;
    bool live = true;
    if (idx < M) lag = idx;
    else if (idx > L - n) lag = L - idx;           // (negative lag: its square is the same)
    else { lag = 0; live = false; }
    double2 v = make_double2(0.0, 0.0);
    if (live) {
      const double2 e = unit_phase((lag * lag) % n2, n2);     // exp(-i pi lag^2 / n); b = its conjugate
      v = make_double2(e.x, -e.y);
    }
    b[idx] = v;
  }
  if (idx < M) eout[idx] = unit_phase((idx * idx) % n2, n2);
}
__global__ void __launch_bounds__(256) czt_scale_kernel(double2* b, long long L) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx < L) {
    const double s = 1.0 / (double)L;
    b[idx] = make_double2(b[idx].x * s, b[idx].y * s);
  }
}
// a[(ch * P + pr)][t] = ((x - mean) * taper_a + i (x - mean) * taper_b)[t] * chirp[t] for t < n, 0 up to L.
// grid (ceil(L / 256), P, channels of the chunk)
__global__ void __launch_bounds__(256) czt_pack_kernel(const double* x, long long ld, const double* mean, const double* tapers,
                                                       const double2* chirp, long long n, long long L, int K, int P, double2* a) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= L) return;
  const int pr = blockIdx.y, ch = blockIdx.z;
  double2 v = make_double2(0.0, 0.0);
  if (t < n) {
    const double xm = x[(size_t)ch * ld + t] - mean[ch];
    const double ya = xm * tapers[(size_t)(2 * pr) * n + t];
    const double yb = (2 * pr + 1 < K) ? xm * tapers[(size_t)(2 * pr + 1) * n + t] : 0.0;
    const double2 c = chirp[t];
    v = make_double2(ya * c.x - yb * c.y, ya * c.y + yb * c.x);
  }
  a[((size_t)ch * P + pr) * L + t] = v;
}
// A[row][l] *= bf[l]; grid (ceil(L / 256), min(rows, 32768)): rows may exceed the 65 535 of a grid's second dimension
__global__ void __launch_bounds__(256) czt_mul_kernel(double2* A, const double2* bf, long long L, long long rows) {
  const long long l = (long long)blockIdx.x * 256 + threadIdx.x;
  if (l >= L) return;
  const double2 v = bf[l];
  for (long long r = blockIdx.y; r < rows; r += gridDim.y) {
    double2* p = A + (size_t)r * L + l;
    const double2 u = *p;
    *p = make_double2(u.x * v.x - u.y * v.y, u.x * v.y + u.y * v.x);
  }
}
// psd[ch][b] from the convolution outputs C[(ch * P + pr)][m], m = j + hi;  grid (ceil(nb / 256), channels of the chunk)
__global__ void __launch_bounds__(256) czt_power_kernel(const double2* C, const double2* eout, const double* w, long long L,
                                                        long long n, long long hi, int K, int P, long long lo, long long nb,
                                                        double* psd, long long psd_ld) {
  const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
  if (b >= nb) return;
  const int ch = blockIdx.y;
  const long long j = lo + b, mp = hi + j, mm = hi - j;
  const double2 ep = eout[mp], em = eout[mm];
  double acc = 0.0, wsum = 0.0;
  for (int pr = 0; pr < P; ++pr) {                 // fixed order: deterministic
    const double2* row = C + ((size_t)ch * P + pr) * L;
    const double2 cp = row[mp], cm = row[mm];
    const double2 zp = make_double2(cp.x * ep.x - cp.y * ep.y, cp.x * ep.y + cp.y * ep.x);       // Z[j]
    const double2 zm = make_double2(cm.x * em.x - cm.y * em.y, cm.x * em.y + cm.y * em.x);       // Z[-j]
    // X_a = (Z[j] + conj Z[-j]) / 2,  X_b = (Z[j] - conj Z[-j]) / (2 i)
    const double ar = 0.5 * (zp.x + zm.x), ai = 0.5 * (zp.y - zm.y);
    const double br = 0.5 * (zp.y + zm.y), bi = -0.5 * (zp.x - zm.x);
    const double wa = w[2 * pr] * w[2 * pr];
    acc += wa * (ar * ar + ai * ai);
    wsum += wa;
    if (2 * pr + 1 < K) {
      const double wb = w[2 * pr + 1] * w[2 * pr + 1];
      acc += wb * (br * br + bi * bi);
      wsum += wb;
    }
  }
  double scale = 2.0 / wsum;
  if (j == 0 || ((n & 1) == 0 && j == n / 2)) scale *= 0.5;
  psd[(size_t)ch * psd_ld + b] = acc * scale;
}

namespace {
std::mutex g_plan_mutex;
std::map<std::pair<long long, int>, hipfftHandle> g_plans;   // (n, batch) -> plan, created once
std::map<std::pair<long long, int>, hipfftHandle> g_zplans;  // (L, batch) -> complex plan of the chirp-z path
std::map<std::pair<long long, long long>, CztPlan> g_czt;    // (n, hi) -> tables

bool smooth_length(long long n) {                  // every prime factor <= 13: rocFFT has kernels, no Bluestein
  for (int f : {2, 3, 5, 7, 11, 13})
    while (n % f == 0) n /= f;
  return n == 1;
}
long long czt_length(long long n, long long hi) {  // 0: the plain transform is the better choice
  if (smooth_length(n) || hi < 0) return 0;
  long long L = 1;
  while (L < n + 2 * hi + 1) L <<= 1;
  long long LB = 1;                                // what Bluestein would take
  while (LB < 2 * n - 1) LB <<= 1;
  return (L <= LB) ? L : 0;        // (two tapers per transform: ahead even at Bluestein's own length)
}
// A batch over many recordings meets a new segment length with almost every file: the caches are bounded (plans carry
// device work buffers), oldest-first is not worth the bookkeeping -- past the bound everything goes and is rebuilt.
constexpr size_t kMaxCached = 24;
void trim_caches(hipStream_t st) {                 // g_plan_mutex held
  if (g_plans.size() + g_zplans.size() <= kMaxCached && g_czt.size() <= kMaxCached) return;
  (void)st;
  (void)hipDeviceSynchronize();                    // nothing in flight, on any stream, uses them (rare: once per 24 lengths)
  for (auto& kv : g_plans) hipfftDestroy(kv.second);
  for (auto& kv : g_zplans) hipfftDestroy(kv.second);
  for (auto& kv : g_czt) {
    (void)hipFree(kv.second.chirp);
    (void)hipFree(kv.second.bf);
    (void)hipFree(kv.second.eout);
  }
  g_plans.clear();
  g_zplans.clear();
  g_czt.clear();
}
hipfftHandle zplan(long long L, int batch) {
  auto it = g_zplans.find({L, batch});
  if (it != g_zplans.end()) return it->second;
  hipfftHandle plan;
  int len[1] = {(int)L};
  if (hipfftPlanMany(&plan, 1, len, nullptr, 1, (int)L, nullptr, 1, (int)L, HIPFFT_Z2Z, batch) != HIPFFT_SUCCESS) return nullptr;
  g_zplans[{L, batch}] = plan;
  return plan;
}
}

long long psd_workspace_bytes(long long ch_chunk, long long n, int K) {
  const long long nfreq = n / 2 + 1;
  long long body = ch_chunk * K * n + 2 * ch_chunk * K * nfreq;          // y and X of the plain path (doubles)
  if (!smooth_length(n)) {      // the chirp-z path: ceil(K / 2) complex rows of up to Bluestein's own length per channel
    long long LB = 1;
    while (LB < 2 * n - 1) LB <<= 1;
    const long long czt = 2 * ch_chunk * ((K + 1) / 2) * LB;
    if (czt > body) body = czt;
  }
  return (long long)sizeof(double) * (body + ch_chunk + 64);
}

int launch_psd(const double* x, long long n_ch, long long n, long long ld, const double* tapers, const double* w, int K,
               long long lo, long long hi, double* psd, void* workspace, long long ch_chunk, hipStream_t st) {
  if (n_ch == 0) return 0;
  const long long nfreq = n / 2 + 1, nb = hi - lo + 1;
  char* base = static_cast<char*>(workspace);
  double* mean = reinterpret_cast<double*>(base);
  {
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    trim_caches(st);
  }
  if (const long long L = czt_length(n, hi)) {
    const int P = (K + 1) / 2;
    // (psd_workspace_bytes covers ch_chunk * P rows of L <= Bluestein's length)
    CztPlan pl;
    {
      std::lock_guard<std::mutex> lock(g_plan_mutex);
      auto it = g_czt.find({n, hi});
      if (it == g_czt.end()) {
        const long long M = 2 * hi + 1;
        if (hipMalloc(&pl.chirp, sizeof(double2) * n) != hipSuccess || hipMalloc(&pl.bf, sizeof(double2) * L) != hipSuccess ||
            hipMalloc(&pl.eout, sizeof(double2) * M) != hipSuccess)
          return -23;
        pl.L = L;
        hipLaunchKernelGGL(czt_tables_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, st, n, hi, L, pl.chirp, pl.bf, pl.eout);
        hipfftHandle p1 = zplan(L, 1);
        if (!p1) return -20;
        if (hipfftSetStream(p1, st) != HIPFFT_SUCCESS) return -21;
        if (hipfftExecZ2Z(p1, reinterpret_cast<hipfftDoubleComplex*>(pl.bf), reinterpret_cast<hipfftDoubleComplex*>(pl.bf),
                          HIPFFT_FORWARD) != HIPFFT_SUCCESS)
          return -22;
        hipLaunchKernelGGL(czt_scale_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, st, pl.bf, L);
        g_czt[{n, hi}] = pl;
      } else {
        pl = it->second;
      }
    }
    double2* A = reinterpret_cast<double2*>(mean + ((ch_chunk + 63) / 64) * 64);
    for (long long c0 = 0; c0 < n_ch; c0 += ch_chunk) {
      const long long c = (n_ch - c0 < ch_chunk) ? (n_ch - c0) : ch_chunk;
      const int batch = (int)(c * P);
      hipLaunchKernelGGL(psd_mean_kernel, dim3((unsigned)c), dim3(256), 0, st, x + (size_t)c0 * ld, ld, n, mean);
      hipLaunchKernelGGL(czt_pack_kernel, dim3((unsigned)((L + 255) / 256), P, (unsigned)c), dim3(256), 0, st, x + (size_t)c0 * ld,
                         ld, mean, tapers, pl.chirp, n, L, K, P, A);
      {
        std::lock_guard<std::mutex> lock(g_plan_mutex);
        hipfftHandle plan = zplan(L, batch);
        if (!plan) return -20;
        if (hipfftSetStream(plan, st) != HIPFFT_SUCCESS) return -21;
        hipfftDoubleComplex* Az = reinterpret_cast<hipfftDoubleComplex*>(A);
        if (hipfftExecZ2Z(plan, Az, Az, HIPFFT_FORWARD) != HIPFFT_SUCCESS) return -22;
        hipLaunchKernelGGL(czt_mul_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)(batch < 32768 ? batch : 32768)), dim3(256), 0, st,
                           A, pl.bf, L, (long long)batch);
        if (hipfftExecZ2Z(plan, Az, Az, HIPFFT_BACKWARD) != HIPFFT_SUCCESS) return -22;
      }
      hipLaunchKernelGGL(czt_power_kernel, dim3((unsigned)((nb + 255) / 256), (unsigned)c), dim3(256), 0, st, A, pl.eout, w, L, n,
                         hi, K, P, lo, nb, psd + (size_t)c0 * nb, nb);
    }
    return (int)hipGetLastError();
  }
  double* y = mean + ((ch_chunk + 63) / 64) * 64;
  double2* X = reinterpret_cast<double2*>(y + (size_t)ch_chunk * K * n);
  for (long long c0 = 0; c0 < n_ch; c0 += ch_chunk) {
    const long long c = (n_ch - c0 < ch_chunk) ? (n_ch - c0) : ch_chunk;
    const int batch = (int)(c * K);
    hipfftHandle plan;
    {
      std::lock_guard<std::mutex> lock(g_plan_mutex);
      auto it = g_plans.find({n, batch});
      if (it == g_plans.end()) {
        int len[1] = {(int)n};
        if (hipfftPlanMany(&plan, 1, len, nullptr, 1, (int)n, nullptr, 1, (int)nfreq, HIPFFT_D2Z, batch) != HIPFFT_SUCCESS)
          return -20;
        g_plans[{n, batch}] = plan;
      } else {
        plan = it->second;
      }
    }
    hipLaunchKernelGGL(psd_mean_kernel, dim3((unsigned)c), dim3(256), 0, st, x + (size_t)c0 * ld, ld, n, mean);
    hipLaunchKernelGGL(psd_taper_kernel, dim3((unsigned)((n + 255) / 256), K, (unsigned)c), dim3(256), 0, st,
                       x + (size_t)c0 * ld, ld, mean, tapers, n, K, y);
    {   // a cached plan carries ONE stream: binding it and enqueueing on it are one critical section, so two host
        // threads that share the plan on different streams cannot re-bind it under each other
      std::lock_guard<std::mutex> lock(g_plan_mutex);
      if (hipfftSetStream(plan, st) != HIPFFT_SUCCESS) return -21;
      if (hipfftExecD2Z(plan, y, reinterpret_cast<hipfftDoubleComplex*>(X)) != HIPFFT_SUCCESS) return -22;
    }
    hipLaunchKernelGGL(psd_power_kernel, dim3((unsigned)((nb + 255) / 256), (unsigned)c), dim3(256), 0, st, X, w, nfreq, n, K,
                       lo, nb, psd + (size_t)c0 * nb, nb);
  }
  return (int)hipGetLastError();
}

}  // namespace hmv
