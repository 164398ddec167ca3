// K3 -- per-frequency transfer matrix:  A(f) = I - sum_k A_k z_k(f),  H(f) = A(f)^-1,  |H|^2 + row sums.
//
// Replaces the Python loop of the reference's `mvar_transfer_function`
// (/root/reference/src/mtmvar.py:126-162: one np.linalg.inv per frequency) and the `|H|^2` of
// `dtf_multivariate` (mtmvar.py:232).  This is 74 % of the path's flops (8 m^3 per frequency).
//
// Design (MI355X-first, see DESIGN.md section "K3"):
//   * ONE wavefront owns ONE (window, frequency) matrix for its whole life.  The MP x MP complex matrix
//     (MP = 16*NT <= 64) sits in the wave's registers in the D layout of hmv_common.h (64 complex values
//     per lane at MP = 64 = 256 VGPRs of the 512 a single wave per SIMD may use).  No workgroup barrier,
//     no global round trip: four independent waves per CU, one per SIMD.
//   * Inversion = in-place blocked Gauss-Jordan, 4 pivot columns per block step:
//       1. the 4-column panel goes through LDS into a lane-per-row layout (64 rows = 64 lanes),
//       2. four sequential pivot steps on the panel: wave-wide arg-max of |re|+|im| (LAPACK izamax
//          metric), optional threshold (tau) to keep the diagonal, explicit row interchange, complex
//          reciprocal, elimination inside the panel.  This yields N = M'[:, S] (the new panel block),
//       3. the pending row interchanges are applied to the register-resident matrix through LDS,
//       4. the rank-4 update  M <- M + (N - E_S) * M[S, :]  runs on v_mfma_f64_4x4x4_4b_f64: the A operand
//          (N - E_S) is read from LDS, the B operand M[S, :] is already in the lane's own registers
//          because rows 4s..4s+3 of the D layout are exactly the B-operand layout,
//       5. the panel columns are overwritten with N.
//     Row interchanges leave the inverse with permuted columns; lane c tracks which original row sits
//     in row c (`orig`), and the output is written to column orig[c].
//   * A(f) is assembled from the AR coefficients ([row][col][lag], lag fastest = the reference's own
//     (m, m, p) layout) with 16-byte loads that hit L2 (all F waves of a window read the same 256 KB).
//
// Outputs (all optional):  P[item][f][row][col] = |H|^2 (scratch layout, transposed to (m, m, F) by K4),
// rowsum[item][f][row] = sum_col |H|^2, H / A as interleaved complex128 [item][f][row][col].
#include "hmv_common.h"
#include "hmv_kernels.h"

namespace hmv {

#define HMV_WAVE_SYNC()                                     \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                        \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
  } while (0)

// Phase stamps for the diagnostic build (-DHMV_STAMP): shares of a matrix's life, never its length.
#ifdef HMV_STAMP
#define HMV_T(idx)                                                                     \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    unsigned long long t_;                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    tsum[idx] += t_ - tlast;                                                           \
    tlast = t_;                                                                        \
  } while (0)
#else
#define HMV_T(idx) do { } while (0)
#endif

template <int NT>
struct TfLds {
  static constexpr int MP = 16 * NT;
  static constexpr int PBUF = MP * 5;   // double2 units, row stride 80 B (bank-conflict-free lane=row reads)
  static constexpr int NBUF = MP * 4;   // double2 units, row stride 64 B
  static constexpr int ROWB = MP;       // one matrix row, double2 units
  static constexpr int TOTAL = PBUF + NBUF + 2 * ROWB;
};

template <int NT>
__global__ void __launch_bounds__(256, 1) tf_inv_kernel(TfArgs a) {
  constexpr int MP = 16 * NT, NI = 4 * NT, NJ = NT, NSTEP = MP / 4;
  using L = TfLds<NT>;
  __shared__ double2 smem[4][L::TOTAL];

  const int l = lane_id();
  const int wv = uni(threadIdx.x >> 6);
  const long long gw = (long long)blockIdx.x * 4 + wv;
  if (gw >= a.n_items * (long long)a.F) return;
  const long long item = gw / a.F;
  const int f = uni((int)(gw - item * a.F));
  const int p = a.p;
  const int i = l >> 4, cc = l & 15;

  double2* Pbuf = smem[wv];
  double2* Nbuf = Pbuf + L::PBUF;
  double2* bufA = Nbuf + L::NBUF;
  double2* bufB = bufA + L::ROWB;

  double re[NI][NJ], im[NI][NJ];
#ifdef HMV_STAMP
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#endif

  // ---------------------------------------------------------------- A(f)
  {
    const double* ar = a.ar + (size_t)item * MP * MP * p;
    const double* tw = a.tw + (size_t)f * p * 2;
    // Lag chunks of KC coefficients: a batch of 2 row blocks x NJ column groups issues all of its
    // 16-byte loads back to back (explicit staging array -- left to itself the register allocator
    // serialises every load behind an s_waitcnt), then runs the FMAs.  KC = 8 reads each element's
    // 64 contiguous bytes exactly once (full 128-B lines, L2 -> L1 traffic = the 256 KB once).
#pragma unroll
    for (int I = 0; I < NI; ++I)
#pragma unroll
      for (int J = 0; J < NJ; ++J) {
        re[I][J] = (4 * I + i == 16 * J + cc) ? 1.0 : 0.0;
        im[I][J] = 0.0;
      }
    const double* e0 = ar + ((size_t)i * MP + cc) * p;
    auto chunk = [&](auto kc_tag, int k0) __attribute__((always_inline)) {
      constexpr int KC = decltype(kc_tag)::value;            // 8, 4 or 2 lags, 16-byte loads
      double zr[KC], zi[KC];
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        zr[k] = tw[2 * (k0 + k)];
        zi[k] = tw[2 * (k0 + k) + 1];
      }
      static_for<NI / 2>([&](auto ic) __attribute__((always_inline)) {
        constexpr int I0 = 2 * decltype(ic)::value;
        double2 v[2][NJ][KC / 2];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int J = 0; J < NJ; ++J) {
            const double2* e = reinterpret_cast<const double2*>(e0 + ((size_t)(4 * (I0 + d)) * MP + 16 * J) * p + k0);
#pragma unroll
            for (int h = 0; h < KC / 2; ++h) v[d][J][h] = e[h];
          }
        __builtin_amdgcn_sched_barrier(0);   // all loads of the batch are in flight before the first FMA
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int J = 0; J < NJ; ++J) {
            double sr = re[I0 + d][J], si = im[I0 + d][J];
#pragma unroll
            for (int h = 0; h < KC / 2; ++h) {
              sr = __builtin_fma(-v[d][J][h].x, zr[2 * h], sr);
              si = __builtin_fma(-v[d][J][h].x, zi[2 * h], si);
              sr = __builtin_fma(-v[d][J][h].y, zr[2 * h + 1], sr);
              si = __builtin_fma(-v[d][J][h].y, zi[2 * h + 1], si);
            }
            re[I0 + d][J] = sr;
            im[I0 + d][J] = si;
          }
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    int k0 = 0;
    if ((p & 1) == 0) {
      for (; k0 + 8 <= p; k0 += 8) chunk(std::integral_constant<int, 8>{}, k0);
      for (; k0 + 2 <= p; k0 += 2) chunk(std::integral_constant<int, 2>{}, k0);
    }
    for (; k0 < p; ++k0) {
      const double zr = tw[2 * k0], zi = tw[2 * k0 + 1];
      static_for<NI / 2>([&](auto ic) __attribute__((always_inline)) {
        constexpr int I0 = 2 * decltype(ic)::value;
        double v[2][NJ];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int J = 0; J < NJ; ++J) v[d][J] = e0[((size_t)(4 * (I0 + d)) * MP + 16 * J) * p + k0];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int J = 0; J < NJ; ++J) {
            re[I0 + d][J] = __builtin_fma(-v[d][J], zr, re[I0 + d][J]);
            im[I0 + d][J] = __builtin_fma(-v[d][J], zi, im[I0 + d][J]);
          }
      });
    }
    if (a.A) {
      double2* Ao = reinterpret_cast<double2*>(a.A) + (size_t)gw * MP * MP;
#pragma unroll
      for (int I = 0; I < NI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) Ao[(size_t)(4 * I + i) * MP + 16 * J + cc] = make_double2(re[I][J], im[I][J]);
    }
  }

  HMV_T(0);
  int orig = l;    // lane c: original row index now sitting in row c
  int info = 0;
  const double tau = a.tau;

  // ---------------------------------------------------------------- blocked Gauss-Jordan
  static_for<NSTEP>([&](auto sc) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
    constexpr int Js = s >> 2, q = s & 3;
    // 1. panel -> LDS -> lane-per-row
    if ((cc >> 2) == q) {
#pragma unroll
      for (int I = 0; I < NI; ++I) Pbuf[(4 * I + i) * 5 + (cc & 3)] = make_double2(re[I][Js], im[I][Js]);
    }
    HMV_WAVE_SYNC();
    double xr[4], xi[4];
    {
      const int r = (l < MP) ? l : 0;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const double2 v = Pbuf[r * 5 + jj];
        xr[jj] = v.x;
        xi[jj] = v.y;
      }
    }
    int swp[4];
    HMV_T(1);
    // 2. four pivot steps inside the panel
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int col = 4 * s + jj;
      const double cand = (l >= col && l < MP) ? (__builtin_fabs(xr[jj]) + __builtin_fabs(xi[jj])) : -1.0;
      // arg-max of |re|+|im| over the not-yet-pivoted rows: float-rounded magnitude as a 32-bit key
      // (monotonic for non-negative floats), DPP max, then the lowest lane holding the maximum.
      const unsigned key = (cand < 0.0) ? 0u : __float_as_uint((float)cand);
      const unsigned kmax = wave_max_u32(key);
      const unsigned long long hit = __ballot(key == kmax);
      int rstar = uni(kmax == 0u ? col : (int)__builtin_ctzll(hit));
      const double vmax = readlane_f64(cand, rstar);
      if (tau < 1.0) {
        const double dc = readlane_f64(cand, col);
        if (dc >= tau * vmax) rstar = col;
      }
      if (!(vmax > 0.0) && info == 0) info = col + 1;
      swp[jj] = rstar;
      if (rstar != col) {
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
          const double ar_ = readlane_f64(xr[j2], col), br_ = readlane_f64(xr[j2], rstar);
          const double ai_ = readlane_f64(xi[j2], col), bi_ = readlane_f64(xi[j2], rstar);
          xr[j2] = (l == col) ? br_ : ((l == rstar) ? ar_ : xr[j2]);
          xi[j2] = (l == col) ? bi_ : ((l == rstar) ? ai_ : xi[j2]);
        }
        const int oc = __builtin_amdgcn_readlane(orig, col), orr = __builtin_amdgcn_readlane(orig, rstar);
        orig = (l == col) ? orr : ((l == rstar) ? oc : orig);
      }
      const double pr = readlane_f64(xr[jj], col), pi = readlane_f64(xi[jj], col);
      const double dd = pr * pr + pi * pi;
      double invd = __builtin_amdgcn_rcp(dd);                    // v_rcp_f64 seed + 2 Newton steps
      invd = __builtin_fma(__builtin_fma(-dd, invd, 1.0), invd, invd);
      invd = __builtin_fma(__builtin_fma(-dd, invd, 1.0), invd, invd);
      const double ivr = pr * invd, ivi = -pi * invd;
      double qr[4], qi[4];
#pragma unroll
      for (int j2 = 0; j2 < 4; ++j2) {
        if (j2 == jj) {
          qr[j2] = ivr;
          qi[j2] = ivi;
        } else {
          const double ar_ = readlane_f64(xr[j2], col), ai_ = readlane_f64(xi[j2], col);
          qr[j2] = ar_ * ivr - ai_ * ivi;
          qi[j2] = ar_ * ivi + ai_ * ivr;
        }
      }
      const double fr = xr[jj], fi = xi[jj];
      const bool isp = (l == col);
#pragma unroll
      for (int j2 = 0; j2 < 4; ++j2) {
        const double tr = fr * qr[j2] - fi * qi[j2], ti = fr * qi[j2] + fi * qr[j2];
        const double nr = (j2 == jj) ? -tr : xr[j2] - tr;
        const double ni = (j2 == jj) ? -ti : xi[j2] - ti;
        xr[j2] = isp ? qr[j2] : nr;
        xi[j2] = isp ? qi[j2] : ni;
      }
    }
    HMV_T(2);
    // N = M'[:, S] in lane-per-row layout -> LDS (A-operand source and panel write-back source)
    if (l < MP) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) Nbuf[l * 4 + jj] = make_double2(xr[jj], xi[jj]);
    }
    // 3. pending row interchanges on the register-resident matrix
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int col = 4 * s + jj;
      const int rstar = swp[jj];
      if (rstar != col) {
        const int Ist = rstar >> 2, ist = rstar & 3;
        if (i == jj) {
#pragma unroll
          for (int J = 0; J < NJ; ++J) bufA[J * 16 + cc] = make_double2(re[s][J], im[s][J]);
        }
        HMV_WAVE_SYNC();
        // expanded at compile time (a run-time loop here is folded into re[Ist][..], i.e. scratch)
        static_for<NI - s>([&](auto ic) __attribute__((always_inline)) {
          constexpr int I = s + decltype(ic)::value;
          if (I == Ist) {
            if (i == ist) {
#pragma unroll
              for (int J = 0; J < NJ; ++J) bufB[J * 16 + cc] = make_double2(re[I][J], im[I][J]);
#pragma unroll
              for (int J = 0; J < NJ; ++J) {
                const double2 v = bufA[J * 16 + cc];
                re[I][J] = v.x;
                im[I][J] = v.y;
              }
            }
          }
        });
        HMV_WAVE_SYNC();
        if (i == jj) {
#pragma unroll
          for (int J = 0; J < NJ; ++J) {
            const double2 v = bufB[J * 16 + cc];
            re[s][J] = v.x;
            im[s][J] = v.y;
          }
        }
        HMV_WAVE_SYNC();
      }
    }
    HMV_WAVE_SYNC();
    HMV_T(3);
    // 4. rank-4 update on the matrix cores
    double ur[NJ], ui[NJ];
#pragma unroll
    for (int J = 0; J < NJ; ++J) {
      ur[J] = re[s][J];
      ui[J] = im[s][J];
    }
    // A operand of row block I+1 is fetched from LDS before the 16 MFMAs of row block I are issued,
    // so its ~100-cycle latency hides behind them (in-order issue: the read must precede them).
    double2 nvn = Nbuf[(l & 3) * 4 + (l >> 4)];
    static_for<NI>([&](auto ic) __attribute__((always_inline)) {
      constexpr int I = decltype(ic)::value;
      const double2 nv = nvn;
      if (I + 1 < NI) nvn = Nbuf[(4 * (I + 1) + (l & 3)) * 4 + (l >> 4)];
      __builtin_amdgcn_sched_barrier(0);
      double nr = nv.x;
      const double ni = nv.y;
      if (I == s) nr -= ((l & 3) == (l >> 4)) ? 1.0 : 0.0;
      const double nni = -ni;
#pragma unroll
      for (int J = 0; J < NJ; ++J) re[I][J] = mfma4(nr, ur[J], re[I][J]);
#pragma unroll
      for (int J = 0; J < NJ; ++J) im[I][J] = mfma4(nr, ui[J], im[I][J]);
#pragma unroll
      for (int J = 0; J < NJ; ++J) re[I][J] = mfma4(nni, ui[J], re[I][J]);
#pragma unroll
      for (int J = 0; J < NJ; ++J) im[I][J] = mfma4(ni, ur[J], im[I][J]);
    });
    HMV_T(4);
    // 5. panel columns <- N
    if ((cc >> 2) == q) {
#pragma unroll
      for (int I = 0; I < NI; ++I) {
        const double2 v = Nbuf[(4 * I + i) * 4 + (cc & 3)];
        re[I][Js] = v.x;
        im[I][Js] = v.y;
      }
    }
    HMV_WAVE_SYNC();
    HMV_T(5);
  });

  // ---------------------------------------------------------------- outputs
  int oc[NJ];
#pragma unroll
  for (int J = 0; J < NJ; ++J) oc[J] = __builtin_amdgcn_ds_bpermute((16 * J + cc) << 2, orig);

  if (l == 0) a.info[gw] = info;

  if (a.H) {
    double2* Ho = reinterpret_cast<double2*>(a.H) + (size_t)gw * MP * MP;
#pragma unroll
    for (int I = 0; I < NI; ++I)
#pragma unroll
      for (int J = 0; J < NJ; ++J) Ho[(size_t)(4 * I + i) * MP + oc[J]] = make_double2(re[I][J], im[I][J]);
  }
  if (a.P) {
    double* Po = a.P + (size_t)gw * MP * MP;
    double* rs = a.rowsum + (size_t)gw * MP;
#pragma unroll
    for (int I = 0; I < NI; ++I) {
      double acc = 0.0;
#pragma unroll
      for (int J = 0; J < NJ; ++J) {
        const double v = re[I][J] * re[I][J] + im[I][J] * im[I][J];
        Po[(size_t)(4 * I + i) * MP + oc[J]] = v;
        acc += v;
      }
      acc = row16_sum_dpp(acc);
      if (cc == 0) rs[4 * I + i] = acc;
    }
  }
#ifdef HMV_STAMP
  HMV_T(6);
  if (a.stamps && l == 0) {
    for (int k = 0; k < 8; ++k) a.stamps[gw * 8 + k] = tsum[k];
  }
#endif
}

// ---------------------------------------------------------------- twiddles
// tw[f][k] = exp(-(k+1) * 2*pi*1j * freqs[f] / fs), same operation order as mtmvar.py:153.
__global__ void twiddle_kernel(const double* freqs, int F, double fs, int p, double* tw) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= F * p) return;
  const int f = idx / p, k = idx - f * p;
  const double c = (double)(-(k + 1) * 2) * 3.141592653589793;   // ((-m*2)*pi)
  const double th = c * freqs[f] / fs;
  double s, co;
  sincos(th, &s, &co);
  tw[2 * idx] = co;
  tw[2 * idx + 1] = s;
}

int launch_twiddles(const double* freqs, int F, double fs, int p, double* tw, hipStream_t st) {
  const int n = F * p;
  hipLaunchKernelGGL(twiddle_kernel, dim3((n + 255) / 256), dim3(256), 0, st, freqs, F, fs, p, tw);
  return (int)hipGetLastError();
}

int launch_tf_inv(const TfArgs& a, int m_pad, hipStream_t st) {
  const long long waves = a.n_items * (long long)a.F;
  if (waves == 0) return 0;
  const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  switch (m_pad) {
    case 16: hipLaunchKernelGGL(tf_inv_kernel<1>, grid, block, 0, st, a); break;
    case 32: hipLaunchKernelGGL(tf_inv_kernel<2>, grid, block, 0, st, a); break;
    case 48: hipLaunchKernelGGL(tf_inv_kernel<3>, grid, block, 0, st, a); break;
    case 64: hipLaunchKernelGGL(tf_inv_kernel<4>, grid, block, 0, st, a); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

}  // namespace hmv
