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

namespace {
std::mutex g_plan_mutex;
std::map<std::pair<long long, int>, hipfftHandle> g_plans;   // (n, batch) -> plan, created once
}

long long psd_workspace_bytes(long long ch_chunk, long long n, int K) {
  const long long nfreq = n / 2 + 1;
  return (long long)sizeof(double) * (ch_chunk * K * n + 2 * ch_chunk * K * nfreq + ch_chunk + 64);
}

int launch_psd(const double* x, long long n_ch, long long n, long long ld, const double* tapers, const double* w, int K,
               long long lo, long long hi, double* psd, void* workspace, long long ch_chunk, hipStream_t st) {
  if (n_ch == 0) return 0;
  const long long nfreq = n / 2 + 1, nb = hi - lo + 1;
  char* base = static_cast<char*>(workspace);
  double* mean = reinterpret_cast<double*>(base);
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
