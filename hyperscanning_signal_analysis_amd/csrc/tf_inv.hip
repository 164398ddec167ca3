// K3 -- per-frequency transfer matrix:  A(f) = I - sum_k A_k z_k(f),  H(f) = A(f)^-1,  |H|^2 + row sums.
//
// Replaces the Python loop of the reference's `mvar_transfer_function`
// (/root/reference/src/mtmvar.py:126-162: one np.linalg.inv per frequency) and the `|H|^2` of
// `dtf_multivariate` (mtmvar.py:232).  This is 74 % of the path's flops (8 m^3 per frequency).
//
// Measured facts that shape the design (tools/ubench_f64*.hip, tools/ubench_overlap.hip):
//   * v_mfma_f64_4x4x4_4b_f64 issues every ~17 cycles and HOLDS the SIMD's vector issue while it runs: an
//     f64 MFMA and VALU work never overlap on one SIMD (a single v_xor after an MFMA costs +13 cycles),
//     LDS instructions do.  So the budget of a SIMD is the SUM of MFMA and VALU issue cycles, the only
//     thing that hides latency is more waves per SIMD, and serial VALU work must be done ONCE.
//
// Design:
//   * One workgroup of NT waves (MP = 16*NT <= 64 channels) owns one (window, frequency) matrix.  Wave w
//     holds the four 4-column blocks w, w+NT, w+2NT, w+3NT (block-cyclic) of the MP x MP complex matrix in
//     registers with the four blocks of the 4x4x4 MFMA on four ROW blocks (lane (i, b, j) holds rows
//     16*Ig+4*b+i, columns 4*(Jl*NT+w)+j in register [Ig][Jl]), so ONE column block (= one pivot panel) can
//     be updated by 4*NT MFMAs on its own: 64 accumulator VGPRs at MP = 64, four workgroups resident per CU.
//   * Inversion = in-place blocked Gauss-Jordan, 4 pivot columns per block step s, with a one-step
//     look-ahead.  Panel s is column block s: owner wave s % NT, register block s / NT, so the ownership
//     rotates every step.  During step s (N_s = M'[:, S_s] is in LDS):
//       - the NEXT owner applies update s to its panel block only (4*NT MFMAs), moves the block through LDS
//         into a lane-per-row layout (64 rows = 64 lanes) and runs the four pivot steps of panel s+1:
//         wave-wide arg-max of |re|+|im| (LAPACK izamax metric; optional threshold tau), row interchange,
//         complex reciprocal, elimination inside the panel; the pivot row is broadcast through LDS, not
//         v_readlane.  It publishes N_{s+1} and the interchange list and DEFERS update s of its other three
//         blocks to the start of step s+1 (N_s stays valid in a ring of three buffers),
//       - the other waves apply the interchanges of step s (rare; through LDS) and the rank-4 update
//         M <- M + (N_s - E_S) * M[S, :] of their four blocks on the matrix pipe (A operand from LDS, B operand
//         = the pivot rows by one ds_swizzle inside the 16-lane row); the owner of panel s replaces its
//         panel block by N_s instead of updating it,
//       - one workgroup barrier per step.
//     The serial factorisation (the critical path) thus overlaps the other waves' MFMA work, and every wave
//     runs one factorisation in NT steps.
//     Row interchanges leave the inverse with permuted columns; `orig[c]` (LDS) tracks which original row
//     sits in row c and the output column of stored column c is orig[c].
//   * A(f) is assembled from the AR coefficients re-packed once per item into this register order
//     (ar_pack_kernel), so every per-frequency read is a fully coalesced 16-byte load.
//
// Outputs (all optional):  P[item][f][row][col] = |H|^2 (scratch layout, transposed to (m, m, F) by K4),
// rowsum[item][f][row] = sum_col |H|^2, H / A as interleaved complex128 [item][f][row][col].
#include "hmv_common.h"
#include "hmv_kernels.h"

namespace hmv {

// Workgroups per CU of the 64-channel instance: 4 = 128 VGPRs (8 spilled).  5 would need 96: the compiler then spills
// 1 764 registers (the 64 accumulators + the factorisation's 52 or the update's operands do not fit), measured again
// in round 2 (profiles/r02_ab_notes.md); kept as a build knob for the next attempt.
#ifndef HMV_K3_WGS
#define HMV_K3_WGS 4
#endif

// Row worker payload loads: 1 = device-scope (sc1) buffer loads -- the pairing the memory model asks for with the
// write-through (sc1) publish stores -- 0 = non-temporal loads (L2-served, bypassing this CU's L1 like sc1 loads; every
// line is written write-through before `ready` is raised and read exactly once per launch).  Same-box A/B, three
// interleaved rounds (profiles/r03_k3_ab_notes.md): sc1 7.78 / 7.72 / 7.71 ms, nt 7.64 / 7.52 / 7.53 ms -- the sc1 form
// costs 2.3 % of K3, above the 1 % it was allowed, so the nt form stays the default; bench.py re-checks the whole
// 5 GB output against the separate K4 pass after every run, and `make SC1=1` builds the other form.
#ifndef HMV_ROW_LOADS_SC1
#define HMV_ROW_LOADS_SC1 0
#endif

#define HMV_LDS_FENCE()                                     \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                        \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
  } while (0)

// Phase stamps for the diagnostic build (-DHMV_STAMP, `make stamp`): shares of a wave's life, not its length.
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
  static constexpr int NRING = 3;            // N_s is read during steps s and s+1, N_{s+1} written during s
  // double2 units
  static constexpr int PBUF = MP * 5;        // panel, row stride 80 B (conflict-free lane-per-row reads)
  static constexpr int NBUF = MP * 4;        // N = M'[:, S], row stride 64 B (a bank-conflict-free swizzled image was
                                             // measured 2 % SLOWER: profiles/r02_ab_notes.md)
  static constexpr int SROW = 8;             // pivot row + displaced row of the current pivot column
  static constexpr int SWAPB = NT * 32;      // per wave: two matrix-row segments of 16 columns
  static constexpr int RSUM = NT * MP / 2;   // per wave row-sum partials (doubles)
  static constexpr int TOTAL = PBUF + NRING * NBUF + SROW + SWAPB + RSUM;
};

typedef double f64x2 __attribute__((ext_vector_type(2)));
// Device-scope write-through stores (global_store ... sc1): the bytes go to memory past the XCD's L2.
// The s_nop 1 is the hazard slot hipcc pads by itself after its own stores and cannot see inside an asm
// statement: a VMEM store of more than 64 bits reads its data VGPRs late, and a VALU write to them within the
// next two wait states (gfx940 family) corrupts the stored value (found the hard way: m = 48 only, half of
// every 16-byte store wrong in some windows).
__device__ __forceinline__ void store_sc1_b128(double* p, f64x2 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_sc1_b64(double* p, double v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// ---------------------------------------------------------------- ffDTF normalisation inside K3
// out[i][j][f] = |H_ij(f)|^2 * (1 / den[i]),  den[i] = sum_{j,f} |H_ij(f)|^2   (mtmvar.py:281-283), from K3's own
// outputs while the rest of the chip keeps inverting.  Same arithmetic and summation order as den_kernel /
// norm_kernel (ffdtf_norm.hip): a window normalised here and one normalised by K4 are bit-identical.
//
// Who does it (all measured on the chip, profiles/r02_norm_trace_last_arriver.txt):
//   * the workgroup whose arrival completes window w only adds up the denominators (128 KB of row sums) and
//     raises ready[w].  Letting it normalise the whole window does NOT work: the completing workgroup always
//     sits on the slowest XCD, normalising slows that XCD further, so ~97 % of all windows were normalised on ONE
//     XCD, up to 37 at a time behind its ~0.55 TB/s path to memory -- K3 took 26 ms instead of 8.4;
//   * output row i of window w is written by a FIXED workgroup of window w + lag (frequency slot f = i, i + F, ..)
//     when it has finished its own matrix: blocks are dealt round-robin over XCDs and CUs, so the rows of every
//     window spread over the whole chip, 256 KB of traffic per workgroup.  With in-order dispatch window w has
//     long been complete by then; a workgroup that finds ready[w] still clear does not wait (no spin, no
//     residency assumption): it appends the row to a list that norm_missed_kernel works off after K3.
// What a row costs is instruction issue on SIMDs shared with three inverting workgroups, so the loop moves 16
// bytes per lane and instruction (loads, LDS, stores), multiplies by one reciprocal, and has no workgroup barrier:
// every wave transposes its own 16(f) x 32(j) tiles in a private LDS region (ds_write2_b64 / ds_read2_b64,
// conflict-free with the odd row stride), four tiles in flight.  Published windows keep |H|^2 ROW-major over
// frequency, Pp[item][i][f][j] (publish step of the kernel): the slab of one output row is F*MP contiguous
// doubles.  Needs F % 16 == 0 and a 16-byte aligned output (the launcher checks; otherwise K4 does the job).
template <int NT>
struct NormLds {
  static constexpr int MP = 16 * NT;
  static constexpr int FC = 16, JC = (MP % 32 == 0) ? 32 : 16, TS = JC + 1;     // wave tile: FC frequencies x JC columns
  static constexpr int DOUBLES = NT * FC * TS + 5 * MP;
};

// den[item][:] from the row sums of all frequencies; every thread of the workgroup; `lds`: NormLds doubles.
template <int NT>
__device__ __forceinline__ void window_denominators(const TfArgs& a, int item, double* lds, const int tid) {
  constexpr int MP = 16 * NT;
  using N = NormLds<NT>;
  double* dpart = lds + NT * N::FC * N::TS;      // [4][MP] partial sums over the four quarters of the grid
  const int F = a.F, t = tid;
  const int ty = t / MP, tx = t - ty * MP;         // 64 * NT / MP = 4 thread rows
  const double* rs = a.rowsum + (size_t)item * F * MP + tx;
  const int fq = (F + 3) >> 2;
  const int f1 = min(F, (ty + 1) * fq);
  double acc = 0.0;
  int f = ty * fq;
  for (; f + 16 <= f1; f += 16) {     // 16 loads in flight, summed in ascending f
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = rs[(size_t)(f + k) * MP];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += v[k];
  }
  for (; f < f1; ++f) acc += rs[(size_t)f * MP];
  dpart[ty * MP + tx] = acc;
  __syncthreads();
  if (t < MP) {
    const double d = ((dpart[t] + dpart[MP + t]) + dpart[2 * MP + t]) + dpart[3 * MP + t];
    store_sc1_b64(a.den + (size_t)item * MP + t, d);
  }
}

// Output row i of window `item`; every thread of the workgroup, no workgroup barrier inside.
template <int NT>
__device__ __forceinline__ void normalise_row(const TfArgs& a, int item, int i, double* lds, const int tid) {
  constexpr int MP = 16 * NT;
  using N = NormLds<NT>;
  constexpr int FC = N::FC, JC = N::JC, TS = N::TS;
  constexpr int J2 = JC / 2, FR = 64 / J2, NL = FC / FR, NQ = JC / 8, NJC = MP / JC, D = 4;
  const int F = a.F, m = a.m;
  const int l = tid & 63, wv = uni(tid >> 6);
  double* tile = lds + wv * (FC * TS);     // this wave's private tile
  const double r = 1.0 / __hip_atomic_load(a.den + (size_t)item * MP + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // wave-trips of this row: (16-frequency chunk, JC-column chunk), dealt round-robin to the waves
  const int total = (F / FC) * NJC;
  const int fr = l / J2, j2 = l - fr * J2;           // load side: FR frequency rows x J2 column pairs per instruction
  const int f2 = l & 7, jr = l >> 3;                 // store side: 8 frequency pairs x 8 columns per instruction
#if !HMV_ROW_LOADS_SC1
  const double* Pw = a.P + ((size_t)item * MP + i) * F * MP + (size_t)fr * MP + 2 * j2;     // Pp[item][i][f][j]
#endif
  double* ow = a.ff + ((size_t)item * m + i) * m * F + (size_t)jr * F + 2 * f2;
#if HMV_ROW_LOADS_SC1
  // The slab was written by OTHER workgroups of this launch with write-through (sc1) stores and is read here with
  // device-scope (sc1) loads -- the pairing the memory model asks for (MI355X_MICROARCH.md, inter-workgroup visibility:
  // every store and every load of the handed-off bytes sc1, stores drained before the flag) -- as buffer loads, which
  // hipcc counts in its own s_waitcnt bookkeeping (an asm global_load ... sc1 would not be).  One descriptor per row slab
  // (128 KB: 32-bit offsets suffice).  Measured against the non-temporal loads of round 2: profiles/r03_k3_ab_notes.md.
  const double* slab = a.P + ((size_t)item * MP + i) * F * MP;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(slab), 0, F * MP * 8, 0x00020000);
  const int off0 = (fr * MP + 2 * j2) * 8;
#endif
  auto issue = [&](f64x2 (&v)[NL], int u) __attribute__((always_inline)) {
    if (u < total) {
      const int jc = u % NJC, f0 = (u / NJC) * FC;
#if HMV_ROW_LOADS_SC1
      const int off = off0 + (f0 * MP + jc * JC) * 8;
#pragma unroll
      for (int k = 0; k < NL; ++k)
        v[k] = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(rs, off + k * FR * MP * 8, 0, 16 /* sc1 */));
#else
      const double* src = Pw + (size_t)f0 * MP + jc * JC;
#pragma unroll
      for (int k = 0; k < NL; ++k)
        v[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(src + (size_t)(k * FR) * MP));
#endif
    }
  };
  auto drain = [&](const f64x2 (&v)[NL], int u) __attribute__((always_inline)) {
    if (u < total) {
      const int jc = u % NJC, f0 = (u / NJC) * FC, j0 = jc * JC;
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        double* w = tile + (fr + k * FR) * TS + 2 * j2;
        w[0] = v[k].x;
        w[1] = v[k].y;
      }
      HMV_LDS_FENCE();               // one wave: LDS operations execute in order, this only pins the compiler
      double* dst = ow + (size_t)j0 * F + f0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const double x0 = tile[(2 * f2) * TS + jr + 8 * q], x1 = tile[(2 * f2 + 1) * TS + jr + 8 * q];
        f64x2 o;
        o.x = x0 * r;
        o.y = x1 * r;
        if (j0 + jr + 8 * q < m) __builtin_nontemporal_store(o, reinterpret_cast<f64x2*>(dst + (size_t)(8 * q) * F));
      }
      HMV_LDS_FENCE();
    }
  };
  f64x2 buf[D][NL];
  static_for<D>([&](auto dc) __attribute__((always_inline)) { issue(buf[decltype(dc)::value], wv + NT * decltype(dc)::value); });
  for (int u = wv; u < total; u += NT * D) {
    static_for<D>([&](auto dc) __attribute__((always_inline)) {
      constexpr int d = decltype(dc)::value;
      drain(buf[d], u + NT * d);
      issue(buf[d], u + NT * (d + D));
    });
  }
}

// Reduced product: the band sums of output row i of window `item` instead of the row itself,
//     bands[item][i][j][b] = sum_{lo[b] <= f < hi[b]} |H_ij(f)|^2 * (1 / den[i]),
// for callers that integrate ffDTF over frequency bands anyway (the reference's graph plots, mtmvar.py:984-987; the
// product that crosses PCIe / xGMI, distributed.py) -- the 8.4 MB per window of the full array are then never written.
// Same arithmetic as band_sums_stream_kernel (ffdtf_norm.hip) on the array normalise_row would have written: lane c of
// a 16-lane row owns the frequency pairs (32 k + 2 c, 32 k + 2 c + 1), k ascending, every value first rounded to the
// ffDTF element (x * r), added through 0 / 1 weights by FMA, then one 16-lane DPP tree -- the same bits, which the
// tests compare.  A wave covers 8 columns per trip (two per lane: the kernel lives in 96 VGPRs), the workgroup MP / 2,
// so the slab Pp[item][i][f][j] is read in two column halves, once per pass of NBP bands.  The weight table lives in
// `lds` (`lds_doubles` of it, >= F): built here from the bin ranges.  Every thread of the workgroup; F % 32 == 0.
template <int NT>
__device__ __forceinline__ void band_row(const TfArgs& a, int item, int i, double* lds, const int lds_doubles, const int tid) {
  constexpr int MP = 16 * NT, NBP = 5, D = 4;
  const int F = a.F, m = a.m, nb = a.nb;
  const int l = tid & 63, wv = uni(tid >> 6);
  const int c = l & 15, jl = 8 * wv + 2 * (l >> 4);        // this lane's columns: jl, jl + 1 (+ MP / 2 in the second half)
  const double r = 1.0 / __hip_atomic_load(a.den + (size_t)item * MP + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int fit = lds_doubles / F;
  const int nbp = fit < NBP ? fit : NBP;                   // bands per pass (the launcher checked fit >= 1)
  const int nk = F >> 5;
  for (int b0 = 0; b0 < nb; b0 += nbp) {
    __syncthreads();                                       // the previous pass's table has been read
    for (int e = tid; e < nbp * F; e += 64 * NT) {
      const int b = e / F, f = e - b * F, bb = b0 + b;
      lds[e] = (bb < nb && f >= a.band_lo[bb] && f < a.band_hi[bb]) ? 1.0 : 0.0;
    }
    __syncthreads();
    const double* wl = lds + 2 * c;
#pragma unroll 1
    for (int jh = 0; jh < 2; ++jh) {
      const int j0 = jl + jh * (MP / 2);
      const double* Pw = a.P + ((size_t)item * MP + i) * F * MP + (size_t)(2 * c) * MP + j0;     // Pp[item][i][2c][j0]
      double acc[NBP][2];
#pragma unroll
      for (int b = 0; b < NBP; ++b) acc[b][0] = acc[b][1] = 0.0;
      auto issue = [&](f64x2 (&v)[2], int k) __attribute__((always_inline)) {
        if (k < nk) {
          const double* src = Pw + (size_t)(32 * k) * MP;
          v[0] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(src));          // frequency 32 k + 2 c
          v[1] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(src + MP));     // frequency 32 k + 2 c + 1
        }
      };
      auto drain = [&](const f64x2 (&v)[2], int k) __attribute__((always_inline)) {
        if (k < nk) {
          const double x0[2] = {v[0].x * r, v[0].y * r}, x1[2] = {v[1].x * r, v[1].y * r};
#pragma unroll
          for (int b = 0; b < NBP; ++b) {
            if (b < nbp) {
              const f64x2 w = *reinterpret_cast<const f64x2*>(wl + b * F + 32 * k);
              acc[b][0] = __builtin_fma(x1[0], w.y, __builtin_fma(x0[0], w.x, acc[b][0]));
              acc[b][1] = __builtin_fma(x1[1], w.y, __builtin_fma(x0[1], w.x, acc[b][1]));
            }
          }
        }
      };
      f64x2 buf[D][2];
      static_for<D>([&](auto dc) __attribute__((always_inline)) { issue(buf[decltype(dc)::value], decltype(dc)::value); });
      for (int k = 0; k < nk; k += D) {
        static_for<D>([&](auto dc) __attribute__((always_inline)) {
          constexpr int d = decltype(dc)::value;
          drain(buf[d], k + d);
          issue(buf[d], k + d + D);
        });
      }
      double* ow = a.bands + (((size_t)item * m + i) * m + j0) * nb + b0;
#pragma unroll
      for (int b = 0; b < NBP; ++b) {
        if (b < nbp && b0 + b < nb) {
          const double t0 = row16_sum_dpp(acc[b][0]), t1 = row16_sum_dpp(acc[b][1]);
          if (c == 0 && j0 < m) ow[b] = t0;
          if (c == 0 && j0 + 1 < m) ow[nb + b] = t1;
        }
      }
    }
  }
}

// Rows that found their window unfinished inside K3 (a.missed[0] = how many, a.missed[1..] = item * MP + i).  Runs
// after K3 on the same stream, so everything is visible; normally the list is empty.
template <int NT>
__global__ void __launch_bounds__(64 * NT) norm_missed_kernel(TfArgs a) {
  __shared__ double lds[2 * TfLds<NT>::TOTAL];           // as much as the row workers inside K3 have
  static_assert(NormLds<NT>::DOUBLES <= 2 * TfLds<NT>::TOTAL, "normaliser tiles do not fit");
  // the rows that came up too early inside K3, then every row of the batch's last windows (items >= tail0: there is no
  // later workgroup in the launch to take them; K3 has ended, so their windows are complete)
  const int n = a.missed[0];
  const long long tail0 = a.fuse_items > a.lag ? a.fuse_items - a.lag : 0;
  const long long total = n + (a.fuse_items - tail0) * a.m;
  for (long long k = blockIdx.x; k < total; k += gridDim.x) {
    int item, i;
    if (k < n) {
      const int e = a.missed[1 + k];
      item = e / (16 * NT);
      i = e % (16 * NT);
    } else {
      const long long r = k - n;
      item = (int)(tail0 + r / a.m);
      i = (int)(r % a.m);
    }
    if (a.bands) band_row<NT>(a, item, i, lds, 2 * TfLds<NT>::TOTAL, (int)threadIdx.x);
    else normalise_row<NT>(a, item, i, lds, (int)threadIdx.x);
    __syncthreads();
  }
}

// ---------------------------------------------------------------- outputs of one inverted matrix
// Shared by the compiler-scheduled body (tf_inv_kernel) and the hand-scheduled 64-channel body (tf_inv64_asm_kernel):
// H / |H|^2 / row sums, the write-through publish and the in-kernel ffDTF normalisation.  `smem` is the inversion's
// LDS block (free after the last barrier of the inversion), `rsum` NT * MP doubles inside it that the publish tile does
// not cover, `s_orig` the column permutation left by the row interchanges, `s_flag` one int of LDS.
// BANDS: the reduced-product row worker (band_row) is compiled in.  Off for the compiler-scheduled 64-channel kernel only:
// the launcher sends band-sum launches of that shape to the hand-scheduled body, and the extra code costs the
// compiler-scheduled one ~100 spilled registers.
template <int NT, bool GEN, bool BANDS = true>
__device__ __forceinline__ void tf_outputs(const TfArgs& a, const int item, const int f, const long long gw, const int w,
                                           const int wv, double (&re)[NT][4], double (&im)[NT][4], double2* smem,
                                           double* rsum, const int* s_orig, int* s_flag, const double* s_det
#ifdef HMV_STAMP
                                           , unsigned long long (&tsum)[8], unsigned long long& tlast
#endif
) {
  constexpr int MP = 16 * NT, NG = NT;
  using L = TfLds<NT>;
  constexpr int NR = L::NRING;
  int& s_info = *s_flag;
  // ---------------------------------------------------------------- outputs
  // Lane coordinates are re-derived from an opaque lane id: reusing the prologue's row indices would keep
  // them alive across the whole sweep (they were spilled to scratch once: 2.5 GB per launch).
  int lo;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo));
  const int rowo = 4 * ((lo >> 2) & 3) + (lo >> 4), jo = lo & 3;
  // thread index rebuilt from the (scalar) wave index and the opaque lane id: nothing here depends on the register the
  // hardware delivered threadIdx.x in, which a hand-written body in front of this function has overwritten
  const int tid = 64 * wv + lo;
  int oc[4];      // output column of stored column c is orig[c] (last written before the final barrier)
#pragma unroll
  for (int Jl = 0; Jl < 4; ++Jl) oc[Jl] = s_orig[4 * (Jl * NT + w) + jo];
  if (w == 0 && lo == 0) {
    a.info[gw] = s_info;
    if constexpr (GEN) {
      if (a.detph) {
        a.detph[2 * gw] = s_det[0];
        a.detph[2 * gw + 1] = s_det[1];
      }
    }
  }

  if (a.H) {
    double2* Ho = reinterpret_cast<double2*>(a.H) + (size_t)gw * MP * MP + (size_t)rowo * MP;
#pragma unroll
    for (int Ig = 0; Ig < NG; ++Ig)
#pragma unroll
      for (int Jl = 0; Jl < 4; ++Jl) Ho[(size_t)(16 * Ig) * MP + oc[Jl]] = make_double2(re[Ig][Jl], im[Ig][Jl]);
  }
  if (a.P) {
    double* rs = rsum + w * MP + rowo;
    const bool publish = !GEN && (a.ff != nullptr || a.bands != nullptr) && item < a.fuse_items;     // workgroup-uniform
    if (!publish) {
      double* Po = a.P + (size_t)gw * MP * MP + (size_t)rowo * MP;
#pragma unroll
      for (int Ig = 0; Ig < NG; ++Ig) {
        double acc = 0.0;
#pragma unroll
        for (int Jl = 0; Jl < 4; ++Jl) {
          const double v = re[Ig][Jl] * re[Ig][Jl] + im[Ig][Jl] * im[Ig][Jl];
          Po[(size_t)(16 * Ig) * MP + oc[Jl]] = v;
          acc += v;
        }
        acc += dpp_f64<0xB1>(acc);      // sum over the four lanes j of the quad (fixed order)
        acc += dpp_f64<0x4E>(acc);
        if (jo == 0) rs[16 * Ig] = acc;
      }
      __syncthreads();
      if (w == 0 && lo < MP) {
        double t = 0.0;
#pragma unroll
        for (int ww = 0; ww < NT; ++ww) t += rsum[ww * MP + lo];   // fixed order: bit-reproducible
        a.rowsum[(size_t)gw * MP + lo] = t;
      }
    } else {
      // This matrix will be read by ANOTHER workgroup inside this launch (the one that completes the window),
      // so it is published write-through: |H|^2 goes through LDS (16*G rows at a time) and leaves as whole rows,
      // 16 bytes per lane, with device-scope (sc1) stores that bypass the non-coherent L2 -- no L2 write-back
      // (a per-workgroup release fence walks the whole 4 MB L2 and serialises: 36 ms instead of 8.4, measured).
      constexpr int G = (NT == 3) ? 1 : (NT == 4 ? 2 : NT), TS = MP + 2, C2 = MP / 2, CNT = 16 * G * C2;
      static_assert(16 * G * TS <= 2 * (L::PBUF + NR * L::NBUF + L::SROW + L::SWAPB), "publish tile overlaps the row sums");
      double* tile = reinterpret_cast<double*>(smem);
      // published layout Pp[item][row][f][col]: row-major over frequency (see normalise_row)
      double* Pg = a.P + (size_t)item * MP * a.F * MP + (size_t)f * MP;
      static_for<NG / G>([&](auto pc) __attribute__((always_inline)) {
        constexpr int pass = decltype(pc)::value;
        if (pass > 0) __syncthreads();            // the previous pass has been read out
        static_for<G>([&](auto gc) __attribute__((always_inline)) {
          constexpr int g = decltype(gc)::value, Ig = pass * G + g;
          double acc = 0.0;
#pragma unroll
          for (int Jl = 0; Jl < 4; ++Jl) {
            const double v = re[Ig][Jl] * re[Ig][Jl] + im[Ig][Jl] * im[Ig][Jl];
            tile[(16 * g + rowo) * TS + oc[Jl]] = v;
            acc += v;
          }
          acc += dpp_f64<0xB1>(acc);
          acc += dpp_f64<0x4E>(acc);
          if (jo == 0) rs[16 * Ig] = acc;
        });
        __syncthreads();
#pragma unroll
        for (int idx = tid; idx < CNT; idx += 64 * NT) {
          const int row = idx / C2, c2 = idx - row * C2;
          const f64x2 v = *reinterpret_cast<const f64x2*>(tile + row * TS + 2 * c2);
          store_sc1_b128(Pg + (size_t)(16 * G * pass + row) * a.F * MP + 2 * c2, v);
        }
      });
      if (w == 0 && lo < MP) {
        double t = 0.0;
#pragma unroll
        for (int ww = 0; ww < NT; ++ww) t += rsum[ww * MP + lo];   // same order as above
        store_sc1_b64(a.rowsum + (size_t)gw * MP + lo, t);
      }
    }
  }
  HMV_T(4);          // |H|^2, LDS transposition, stores issued

  // ---------------------------------------------------------------- fused normalisation (ffDTF)
  // This workgroup's |H|^2 and row sums were stored write-through (sc1).  Every storing wave drains its stores,
  // workgroup barrier, then ONE lane counts the matrix on its window (device-scope atomic).  The workgroup whose
  // count completes the window (agent-scope acquire: its CU's L1 is invalidated) adds up the denominators and
  // raises ready[window].  Nobody waits for anybody.
  if constexpr (!GEN) {
    if (a.ff != nullptr || a.bands != nullptr) {             // kernel-uniform
      static_assert(NormLds<NT>::DOUBLES <= 2 * L::TOTAL, "normaliser tiles do not fit the inversion's LDS block");
      double* nlds = reinterpret_cast<double*>(smem);
      if (item < a.fuse_items) {                             // workgroup-uniform
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          const int old = __hip_atomic_fetch_add(a.wcount + item, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int last = (old == a.F - 1) ? 1 : 0;
          if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          s_info = last;
        }
        __syncthreads();
        HMV_T(5);    // stores drained, matrix counted
        if (s_info != 0) {
          window_denominators<NT>(a, item, nlds, tid);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (tid == 0) __hip_atomic_store(a.ready + item, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // Rows f, f + F, ... of window item - lag are this workgroup's to normalise.
      const int wl = item - a.lag;
      if (wl >= 0 && wl < a.fuse_items - a.lag && f < a.m) { // workgroup-uniform (rows of the last `lag` fused items: norm_missed_kernel)
        __syncthreads();
        if (tid == 0) {
          // No acquire fence here (it costs ~7 us with four workgroups on the CU, 38 000 times per launch): the
          // flag and the denominator are read with device-scope (sc1) loads, and the |H|^2 rows with non-temporal
          // loads that bypass this CU's L1 -- every one of those lines was written write-through before ready[wl]
          // was raised and is read exactly once in the whole launch, so no cache can hold an older copy of it.
          const int rdy = __hip_atomic_load(a.ready + wl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (!rdy) {
            for (int i = f; i < a.m; i += a.F) {
              const int pos = __hip_atomic_fetch_add(a.missed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              a.missed[1 + pos] = wl * MP + i;
            }
          }
          s_info = rdy;
        }
        __syncthreads();
        HMV_T(6);    // denominators (rarely), flag of the window whose row is this workgroup's
        if (s_info != 0) {
          if (BANDS && a.bands) {
            if constexpr (BANDS)
              for (int i = f; i < a.m; i += a.F) band_row<NT>(a, wl, i, nlds, 2 * L::TOTAL, tid);
          } else {
            for (int i = f; i < a.m; i += a.F) normalise_row<NT>(a, wl, i, nlds, tid);
          }
        }
      }
    }
  }
#ifdef HMV_STAMP
  HMV_T(7);          // the row
  if (a.stamps && lo == 0) {
    for (int k = 0; k < 8; ++k) a.stamps[(gw * NT + wv) * 8 + k] = tsum[k];
  }
#endif
}

// GEN = false: A(f) from the AR coefficients (the hot path).  GEN = true: the same inversion of arbitrary
// complex matrices Zin[item][f][MP][MP] (partial coherence of a spectral matrix, mtmvar.py:287-338), which
// also returns the unit-modulus phase of the determinant (product of the pivots, sign of the interchanges).
template <int NT, bool GEN>
__global__ void __launch_bounds__(64 * NT, (NT == 4) ? HMV_K3_WGS : 2) tf_inv_kernel(TfArgs a) {
  constexpr int MP = 16 * NT, NG = NT, NSTEP = MP / 4;
  using L = TfLds<NT>;
  constexpr int NR = L::NRING;
  __shared__ double2 smem[L::TOTAL];
  __shared__ int s_orig[MP];
  __shared__ __attribute__((aligned(16))) int s_swp[NR][4];
  __shared__ int s_info;
  __shared__ double s_det[2];                   // GEN: running unit-modulus product of the pivots

  const int l = lane_id();
  const int wv = uni(threadIdx.x >> 6);         // hardware wave index inside the workgroup
  // Column blocks owned by this wave, rotated per workgroup by a hash of the block id: co-resident
  // workgroups run in near lockstep and wave k of every workgroup tends to sit on the same SIMD, so without
  // the rotation the resident panel factorisations (one wave each) pile up on one SIMD.
  const int w = (wv + (int)((blockIdx.x * 2654435761u) >> 20)) % NT;
  // One workgroup = one (window, frequency) matrix.  (A runtime loop over several frequencies per workgroup
  // was tried to amortise the publish step below: the loop-carried state costs 525 spilled VGPRs.)
  const int item = uni((int)(blockIdx.x / (unsigned)a.F));   // wave-uniform: keep it in SGPRs
  const int f = uni((int)(blockIdx.x - (unsigned)item * (unsigned)a.F));
  const long long gw = (long long)item * a.F + f;
  const int p = a.p;
  // "X layout" (hmv_common.h lane maps with the four MFMA blocks on four ROW blocks): lane (i, b, j) holds
  // rows 16*Ig + 4*b + i and columns 4*(Jl*NT + w) + j in register [Ig][Jl].
  const int i = l >> 4, b = (l >> 2) & 3, j = l & 3;
  const int rowl = 4 * b + i;                   // row inside a 16-row group
  const int colw = 4 * w + j;                   // column inside a group of 4*NT columns

  double2* Pbuf = smem;
  double2* Nbuf = Pbuf + L::PBUF;
  double2* Srow = Nbuf + NR * L::NBUF;
  double2* swapb = Srow + L::SROW + wv * 32;
  double* rsum = reinterpret_cast<double*>(Srow + L::SROW + L::SWAPB);

  double re[NG][4], im[NG][4];
#ifdef HMV_STAMP
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#endif

  // ---------------------------------------------------------------- A(f), this wave's 16 columns
  if constexpr (GEN) {
    const double2* Zi = reinterpret_cast<const double2*>(a.Zin) + (size_t)gw * MP * MP + (size_t)rowl * MP + colw;
#pragma unroll
    for (int Ig = 0; Ig < NG; ++Ig)
#pragma unroll
      for (int Jl = 0; Jl < 4; ++Jl) {
        const double2 v = Zi[(size_t)(16 * Ig) * MP + 4 * NT * Jl];
        re[Ig][Jl] = v.x;
        im[Ig][Jl] = v.y;
      }
  } else {
    // Coefficients in the packed layout written by ar_pack_kernel: for wave w, register (Ig, Jl) and lag pair
    // h the 64 lanes' (a[2h], a[2h+1]) are 1 KB contiguous, so every load instruction is one fully coalesced
    // 16-B/lane read (the reference (m, m, p) layout costs 64 partial cache lines per instruction).
    const int P2 = (p + 1) >> 1;
    const double2* ax = reinterpret_cast<const double2*>(a.arx) +
                        ((size_t)item * NT + w) * (size_t)(NG * 4 * 64) * P2 + l;
    const double* tw = a.tw + (size_t)f * p * 2;
#pragma unroll
    for (int Ig = 0; Ig < NG; ++Ig)
#pragma unroll
      for (int Jl = 0; Jl < 4; ++Jl) {
        re[Ig][Jl] = (16 * Ig + rowl == 4 * NT * Jl + colw) ? 1.0 : 0.0;
        im[Ig][Jl] = 0.0;
      }
    // Chunks of HC lag pairs.  The loads roll: DEPTH register blocks' worth (DEPTH * HC 16-byte loads per lane) are in
    // flight while one block is consumed -- the coefficients come out of the XCD's L2 at ~2000 cycles per round trip
    // with every CU pulling, and a round trip per batch of 8 loads was 12.7 % of a wave's life (tools/k3_stamps.py).
    // Explicit staging + sched_barrier: left alone the compiler serialises every load behind an s_waitcnt.
    auto chunk = [&](auto hc_tag, int h0) __attribute__((always_inline)) {
      constexpr int HC = decltype(hc_tag)::value;
      constexpr int NB = NG * 4, DEPTH = (HC == 4) ? 3 : 8;       // 12 (HC = 4) or 8 (HC = 1) loads in flight
      double zr[2 * HC], zi[2 * HC];
#pragma unroll
      for (int k = 0; k < 2 * HC; ++k) {
        const int lag = 2 * h0 + k;          // the padding lag of an odd order has a zero coefficient
        zr[k] = (lag < p) ? tw[2 * lag] : 0.0;
        zi[k] = (lag < p) ? tw[2 * lag + 1] : 0.0;
      }
      double2 v[DEPTH][HC];
      auto issue = [&](auto bc) __attribute__((always_inline)) {
        constexpr int B = decltype(bc)::value;
        const double2* e = ax + ((size_t)(B * P2 + h0)) * 64;
#pragma unroll
        for (int h = 0; h < HC; ++h) v[B % DEPTH][h] = e[h * 64];
      };
      static_for<(DEPTH < NB ? DEPTH : NB)>([&](auto bc) __attribute__((always_inline)) { issue(bc); });
      static_for<NB>([&](auto bc) __attribute__((always_inline)) {
        constexpr int B = decltype(bc)::value, Ig = B >> 2, Jl = B & 3;
        __builtin_amdgcn_sched_barrier(0);
        double sr = re[Ig][Jl], si = im[Ig][Jl];
#pragma unroll
        for (int h = 0; h < HC; ++h) {
          sr = __builtin_fma(-v[B % DEPTH][h].x, zr[2 * h], sr);
          si = __builtin_fma(-v[B % DEPTH][h].x, zi[2 * h], si);
          sr = __builtin_fma(-v[B % DEPTH][h].y, zr[2 * h + 1], sr);
          si = __builtin_fma(-v[B % DEPTH][h].y, zi[2 * h + 1], si);
        }
        re[Ig][Jl] = sr;
        im[Ig][Jl] = si;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (B + DEPTH < NB) issue(std::integral_constant<int, B + DEPTH>{});
      });
    };
    int h0 = 0;
    for (; h0 + 4 <= P2; h0 += 4) chunk(std::integral_constant<int, 4>{}, h0);
    for (; h0 < P2; ++h0) chunk(std::integral_constant<int, 1>{}, h0);
    if (a.A) {
      double2* Ao = reinterpret_cast<double2*>(a.A) + (size_t)gw * MP * MP + (size_t)rowl * MP + colw;
#pragma unroll
      for (int Ig = 0; Ig < NG; ++Ig)
#pragma unroll
        for (int Jl = 0; Jl < 4; ++Jl)
          Ao[(size_t)(16 * Ig) * MP + 4 * NT * Jl] = make_double2(re[Ig][Jl], im[Ig][Jl]);
    }
  }
  if (w == 0) {
    if (l < MP) s_orig[l] = l;
    if (l == 0) {
      s_info = 0;
      s_det[0] = 1.0;
      s_det[1] = 0.0;
    }
  }
  const double tau = a.tau;
  HMV_T(0);

  // ---------------------------------------------------------------- building blocks
  // Panel t (columns 4t..4t+3) = register block J of its owner: -> LDS -> lane-per-row, four pivot steps,
  // N_t and the interchange list -> LDS.  Runs on ONE wave; it is the workgroup's critical path, so it
  // outranks the other workgroups' MFMA work on this SIMD.
  auto factor_panel = [&](auto tc, auto jc) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value, J = decltype(jc)::value;
    double2* Nout = Nbuf + (t % NR) * L::NBUF;
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int Ig = 0; Ig < NG; ++Ig) Pbuf[(16 * Ig + rowl) * 5 + j] = make_double2(re[Ig][J], im[Ig][J]);
    HMV_LDS_FENCE();
    double2 x[4];     // (re, im) adjacent: LDS transfers need no register shuffling
    {
      const int r = (l < MP) ? l : 0;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) x[jj] = Pbuf[r * 5 + jj];
    }
    int rs[4];               // pivot rows of the four columns (wave-uniform)
    double dds[4];           // |pivot|^2 of the four columns (wave-uniform): zero / NaN pivots are looked for once, below
    double dpr = 1.0, dpi = 0.0;   // GEN: unit-modulus product of this panel's pivots (wave-uniform)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int col = 4 * t + jj;
      const bool valid = (l >= col) && (l < MP);
      // Every lane keeps its current row in its own Pbuf slot (where the panel arrived: column 0 finds it there), so
      // the pivot row -- whichever it turns out to be -- is one broadcast read away and nothing on the common path
      // is exec-masked or branched around (a masked region costs a v_cmp / s_and_saveexec / s_cbranch_execz
      // sequence, ~40 cycles of the workgroup's critical path per pivot column).
      if (jj > 0) {
        if (NT == 4 || l < MP) {
#pragma unroll
          for (int j2 = 0; j2 < 4; ++j2) Pbuf[l * 5 + j2] = x[j2];
        }
        HMV_LDS_FENCE();
      }
      // Pivot = arg-max of |re|+|im| (LAPACK izamax metric) over the not-yet-pivoted rows, or the diagonal
      // when it is within tau of the maximum.  Fast path: no row exceeds the diagonal (one compare + ballot;
      // the common case for the A(f) of a stable model).  Otherwise the float-rounded magnitude is a 32-bit
      // key (monotonic for non-negative floats): DPP max, lowest lane holding the maximum.
      const double cand = __builtin_fabs(x[jj].x) + __builtin_fabs(x[jj].y);
      // The diagonal is read (v_readlane) and its reciprocal started at once, before the test resolves; an
      // interchange (slow path) repeats both for the row it picks.  1/(pr + i pi) = (pr - i pi) / dd with
      // 1/dd from the v_rcp_f64 seed (2^-24) and two Newton steps, the second folded into the products.
      double pr = readlane_f64(x[jj].x, col), pi = readlane_f64(x[jj].y, col);
      const double dc = __builtin_fabs(pr) + __builtin_fabs(pi);
      const bool need_search = __builtin_amdgcn_ballot_w64(valid && (tau * cand > dc)) != 0ull;
      double dd, ivr, ivi;
      auto reciprocal = [&]() __attribute__((always_inline)) {
        dd = __builtin_fma(pr, pr, pi * pi);
        double y = __builtin_amdgcn_rcp(dd);
        y = __builtin_fma(__builtin_fma(-dd, y, 1.0), y, y);
        const double e = __builtin_fma(-dd, y, 1.0), tr = pr * y, ti = -pi * y;
        ivr = __builtin_fma(tr, e, tr);
        ivi = __builtin_fma(ti, e, ti);
      };
      reciprocal();
      // Everything only an interchange needs (search, second reciprocal, displaced row, orig[]) sits behind
      // one wave-uniform branch; the pivot row goes through LDS and is broadcast to every lane.
      int rstar = col;
      if (__builtin_expect(need_search, 0)) {
        const unsigned key = valid ? __float_as_uint((float)cand) : 0u;
        const unsigned kmax = wave_max_u32(key);
        rstar = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(valid && key == kmax));
        if (rstar != col) {
          pr = readlane_f64(x[jj].x, rstar);
          pi = readlane_f64(x[jj].y, rstar);
          reciprocal();
#pragma unroll
          for (int j2 = 0; j2 < 4; ++j2) {       // the lane that holds the pivot row takes the displaced row (slot col)
            const double2 cv = Pbuf[col * 5 + j2];
            x[j2].x = (l == rstar) ? cv.x : x[j2].x;
            x[j2].y = (l == rstar) ? cv.y : x[j2].y;
          }
          if (l == 0) {
            const int oc = s_orig[col], orr = s_orig[rstar];
            s_orig[col] = orr;
            s_orig[rstar] = oc;
          }
        }
      }
      rs[jj] = rstar;
      if constexpr (GEN) {       // det *= pivot / |pivot|, negated by an interchange
        double rsq = __builtin_amdgcn_rsq(dd);
        rsq = rsq * __builtin_fma(-0.5 * dd * rsq, rsq, 1.5);
        rsq = rsq * __builtin_fma(-0.5 * dd * rsq, rsq, 1.5);
        const double sg = (rstar != col) ? -rsq : rsq;
        const double ur = pr * sg, ui = pi * sg;
        const double nr_ = __builtin_fma(dpr, ur, -(dpi * ui)), ni_ = __builtin_fma(dpr, ui, dpi * ur);
        dpr = nr_;
        dpi = ni_;
      }
      double2 pv[4];                 // the pivot row as it was parked (before any interchange touched the registers)
      {
        const double2* prow = Pbuf + uni(rstar) * 5;
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) pv[j2] = prow[j2];
      }
      HMV_LDS_FENCE();
      dds[jj] = dd;
      // Elimination with the per-row multiplier mu = -x_jj / pivot:  x <- x + mu * (pivot row), and column
      // jj becomes mu itself (in-place inverse).  The pivot row's lane takes mu = 1/pivot on a zeroed row,
      // which yields the scaled pivot row and 1/pivot in column jj from the same FMAs: no per-element
      // selects and no separately scaled pivot row (27 VALU ops per column instead of 44).
      const bool isp = (l == col);
      const double fr = x[jj].x, fi = x[jj].y;
      double mr = __builtin_fma(fi, ivi, -(fr * ivr));
      double mi = __builtin_fma(-fr, ivi, -(fi * ivr));
      mr = isp ? ivr : mr;
      mi = isp ? ivi : mi;
      // (Scaling the pivot row as row + (1/pivot - 1) * row would save the zeroed copy -- 5 instructions per column --
      // and was measured 0.1 ms SLOWER: the six multiplications are independent work that holds the issue port between
      // the dependent steps of the chain, profiles/r02_ab_notes.md.)
      const double keep = isp ? 0.0 : 1.0;
#pragma unroll
      for (int j2 = 0; j2 < 4; ++j2) {
        if (j2 == jj) continue;
        const double br = x[j2].x * keep, bi = x[j2].y * keep;
        x[j2].x = __builtin_fma(-mi, pv[j2].y, __builtin_fma(mr, pv[j2].x, br));
        x[j2].y = __builtin_fma(mi, pv[j2].x, __builtin_fma(mr, pv[j2].y, bi));
      }
      x[jj].x = mr;
      x[jj].y = mi;
    }
    if (NT == 4 || l < MP) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) Nout[l * 4 + jj] = x[jj];
    }
    // zero (or NaN) pivot: one test for the panel on the common path, the column is worked out only if it fails
    const bool all_ok = (dds[0] > 0.0) && (dds[1] > 0.0) && (dds[2] > 0.0) && (dds[3] > 0.0);
    if (l == 0) {
      // interchange list as row distances (0 = none): the other waves test the four of them with one OR
      *reinterpret_cast<int4*>(&s_swp[t % NR][0]) =
          make_int4(rs[0] - 4 * t, rs[1] - (4 * t + 1), rs[2] - (4 * t + 2), rs[3] - (4 * t + 3));
      if (__builtin_expect(!all_ok, 0)) {
        const int bad = !(dds[0] > 0.0) ? 4 * t + 1 : (!(dds[1] > 0.0) ? 4 * t + 2 : (!(dds[2] > 0.0) ? 4 * t + 3 : 4 * t + 4));
        if (s_info == 0) s_info = bad;
      }
      if constexpr (GEN) {
        const double qr = s_det[0], qi = s_det[1];
        s_det[0] = __builtin_fma(qr, dpr, -(qi * dpi));
        s_det[1] = __builtin_fma(qr, dpi, qi * dpr);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // A operands of update t: (N_t - E_S) rows 16*Ig + (l & 15), k = l >> 4 (one LDS round trip)
  auto load_n = [&](auto tc, double (&nr)[NG], double (&ni)[NG]) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
    const double2* Nt = Nbuf + (t % NR) * L::NBUF;
    const double dlt = (b == (t & 3) && j == i) ? 1.0 : 0.0;
#pragma unroll
    for (int Ig = 0; Ig < NG; ++Ig) {
      const double2 v = Nt[(16 * Ig + (l & 15)) * 4 + (l >> 4)];
      nr[Ig] = (Ig == (t >> 2)) ? v.x - dlt : v.x;
      ni[Ig] = v.y;
    }
  };
  // Rank-4 update t of register block J on the matrix pipe.  B operand: pivot row 4t+k at this lane's
  // column, held by lane (i = k, b = t & 3, j) in register [t >> 2][J] -> one swizzle inside the 16-lane row.
  auto update_block = [&](auto tc, auto jc, const double (&nr)[NG], const double (&ni)[NG])
                          __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value, J = decltype(jc)::value;
    // ds_swizzle (bit mode: lane' = (lane & 0x13) | (b << 2) inside each half wave) costs 2.2 cycles of the
    // LDS pipe against 6.1 for ds_bpermute and needs no address register (tools/ubench_lds.hip)
    constexpr int pat = 0x13 | (((t & 3) << 2) << 5);
    const double ur = swizzle_f64<pat>(re[t >> 2][J]), ui = swizzle_f64<pat>(im[t >> 2][J]);
#pragma unroll
    for (int Ig = 0; Ig < NG; ++Ig) {
      re[Ig][J] = mfma4(nr[Ig], ur, re[Ig][J]);
      im[Ig][J] = mfma4(nr[Ig], ui, im[Ig][J]);
    }
#pragma unroll
    for (int Ig = 0; Ig < NG; ++Ig) {
      re[Ig][J] = mfma4_nega(ni[Ig], ui, re[Ig][J]);     // re -= ni * ui (NEG modifier)
      im[Ig][J] = mfma4(ni[Ig], ur, im[Ig][J]);
    }
  };
  // Row interchanges of step t on this wave's 16 columns (rare; through LDS).
  auto interchange = [&](auto tc) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value, Igp = t >> 2, bp = t & 3;
    const int4 swv = *reinterpret_cast<const int4*>(&s_swp[t % NR][0]);      // row distances, 0 = no interchange
    if (__builtin_expect(uni(swv.x | swv.y | swv.z | swv.w) == 0, 1)) return;   // one branch on the common path
    const int swr[4] = {uni(swv.x) + 4 * t, uni(swv.y) + 4 * t + 1, uni(swv.z) + 4 * t + 2, uni(swv.w) + 4 * t + 3};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int col = 4 * t + jj;
      const int rstar = swr[jj];
      if (rstar != col) {
        // row `col` = (Ig Igp, b bp, i jj); row rstar = (Ig rstar>>4, b (rstar>>2)&3, i rstar&3)
        const int Igs = rstar >> 4, bs = (rstar >> 2) & 3, is = rstar & 3;
        if (i == jj && b == bp) {
#pragma unroll
          for (int Jl = 0; Jl < 4; ++Jl) swapb[Jl * 4 + j] = make_double2(re[Igp][Jl], im[Igp][Jl]);
        }
        HMV_LDS_FENCE();
        // expanded at compile time (a run-time loop here is folded into re[Igs], i.e. scratch)
        static_for<NG - Igp>([&](auto ic) __attribute__((always_inline)) {
          constexpr int Ig = Igp + decltype(ic)::value;
          if (Ig == Igs) {
            if (i == is && b == bs) {
#pragma unroll
              for (int Jl = 0; Jl < 4; ++Jl) swapb[16 + Jl * 4 + j] = make_double2(re[Ig][Jl], im[Ig][Jl]);
#pragma unroll
              for (int Jl = 0; Jl < 4; ++Jl) {
                const double2 v = swapb[Jl * 4 + j];
                re[Ig][Jl] = v.x;
                im[Ig][Jl] = v.y;
              }
            }
          }
        });
        HMV_LDS_FENCE();
        if (i == jj && b == bp) {
#pragma unroll
          for (int Jl = 0; Jl < 4; ++Jl) {
            const double2 v = swapb[16 + Jl * 4 + j];
            re[Igp][Jl] = v.x;
            im[Igp][Jl] = v.y;
          }
        }
        HMV_LDS_FENCE();
      }
    }
  };
  // panel block <- N_t (owner of panel t)
  auto take_panel = [&](auto tc, auto jc) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value, J = decltype(jc)::value;
    const double2* Nt = Nbuf + (t % NR) * L::NBUF;
#pragma unroll
    for (int Ig = 0; Ig < NG; ++Ig) {
      const double2 v = Nt[(16 * Ig + rowl) * 4 + j];
      re[Ig][J] = v.x;
      im[Ig][J] = v.y;
    }
  };

  // ---------------------------------------------------------------- blocked Gauss-Jordan
  if (w == 0) factor_panel(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  HMV_T(1);
  __syncthreads();
  HMV_T(2);
  static_for<NSTEP>([&](auto sc) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
    constexpr int own = s % NT, Js = s / NT;                      // owner wave / register block of panel s
    constexpr bool has_next = (s + 1 < NSTEP);
    constexpr int nxt = (s + 1) % NT, Jn = has_next ? (s + 1) / NT : 0;
    // The next owner carries the workgroup's critical path from this barrier to the end of its factorisation.
    if (has_next && w == nxt) __builtin_amdgcn_s_setprio(3);
    // Every update is a guarded slot of one register block in straight-line code (wave-uniform guards):
    // large if/else bodies that each rewrite many accumulators make the register allocator copy and spill.
    // ---- A. the owner of panel s factored it during step s-1 and deferred update s-1 of its other blocks
    if constexpr (s >= 1) {
      constexpr int t = s - 1;
      double ar_[NG], ai_[NG];
#pragma unroll
      for (int Ig = 0; Ig < NG; ++Ig) ar_[Ig] = ai_[Ig] = 0.0;
      if (w == own) load_n(std::integral_constant<int, t>{}, ar_, ai_);
      static_for<4>([&](auto jc) __attribute__((always_inline)) {
        constexpr int Jl = decltype(jc)::value;
        // not its panel block (updated before the factorisation), nor panel block t itself (NT == 1)
        if constexpr (Jl != Js && !(NT == 1 && Jl == t / NT)) {
          if (w == own) update_block(std::integral_constant<int, t>{}, jc, ar_, ai_);
        }
      });
    }
    HMV_T(3);
    __builtin_amdgcn_sched_barrier(0);     // keep the next operand loads below the deferred MFMAs (VGPR budget)
    // ---- B. interchanges of step s, A operands of update s
    double nr[NG], ni[NG];
    load_n(sc, nr, ni);
    interchange(sc);
    HMV_T(3);
    // ---- C. update s.  The next owner does its panel block only, then factors panel s+1; the owner of
    // panel s replaces its panel block by N_s.
    const bool is_nxt = has_next && (w == nxt);
    if (w == own) take_panel(sc, std::integral_constant<int, Js>{});
    if (is_nxt || !(w == own && Jn == Js)) update_block(sc, std::integral_constant<int, Jn>{}, nr, ni);
    static_for<4>([&](auto jc) __attribute__((always_inline)) {
      constexpr int Jl = decltype(jc)::value;
      if constexpr (Jl != Jn) {
        if (!is_nxt && !(w == own && Jl == Js)) update_block(sc, jc, nr, ni);
      }
    });
    HMV_T(3);
    __builtin_amdgcn_sched_barrier(0);
    // last in program order (the next owner skipped the slots above): the A operands are dead here, the
    // factorisation runs with only the accumulators live
    if (is_nxt) factor_panel(std::integral_constant<int, s + 1>{}, std::integral_constant<int, Jn>{});
    HMV_T(1);
    __syncthreads();
    HMV_T(2);
  });

  HMV_T(3);
  tf_outputs<NT, GEN, (NT < 4)>(a, item, f, gw, w, wv, re, im, smem, rsum, s_orig, &s_info, s_det
#ifdef HMV_STAMP
                      , tsum, tlast
#endif
  );
}

// ---------------------------------------------------------------- hand-scheduled 64-channel body
// Same arithmetic as tf_inv_kernel<4, false> (A(f) build and blocked Gauss-Jordan inversion, operation for operation,
// hence the same bits: tests/test_gpu_parity.py), as ONE asm statement generated by csrc/gen/k3gen.py with every VGPR
// placed by hand: 96 registers instead of 128, i.e. five workgroups per CU instead of four.  K3 is bound by the latency
// of the per-step chain (panel-block update -> LDS hand-over -> panel factorisation -> barrier), which only more
// resident matrices cover; the compiler needs 128 registers for the body and spills ~1 700 at 96.
// The stream is executed on a CPU emulator against NumPy, and its wait states / wait counts are checked there
// (tests/test_k3_asm_cpu.py), because hipcc does neither inside an asm statement.
#ifndef HMV_K3A_INC
#define HMV_K3A_INC "tf_inv64_body.inc"      // tools/dbg/k3a_variant.sh builds A/B variants from other option sets
#endif
#include HMV_K3A_INC
#ifndef HMV_K3A_WGS
#ifdef HMV_STAMP
#define HMV_K3A_WGS 4      // the stamped body pins twelve more SGPRs: the compiler then needs spare VGPRs to park its own
#else
#define HMV_K3A_WGS 5
#endif
#endif
static_assert(K3A_PBUF == 0 && K3A_NBUF == 16 * TfLds<4>::PBUF && K3A_RSUM == 16 * (TfLds<4>::TOTAL - TfLds<4>::RSUM),
              "LDS map of the generated body and of the epilogue disagree");

__global__ void __launch_bounds__(256, HMV_K3A_WGS) tf_inv64_asm_kernel(TfArgs a) {
  constexpr int NT = 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[K3A_LDS_TOTAL];
  int* s_orig = reinterpret_cast<int*>(lds + K3A_SORIG);
  int* s_flag = reinterpret_cast<int*>(lds + K3A_SINFO);
  const int wv = uni(threadIdx.x >> 6);
  const int w = (wv + (int)((blockIdx.x * 2654435761u) >> 20)) % NT;      // same rotation as tf_inv_kernel
  const int item = uni((int)(blockIdx.x / (unsigned)a.F));
  const int f = uni((int)(blockIdx.x - (unsigned)item * (unsigned)a.F));
  const long long gw = (long long)item * a.F + f;
  if (w == 0) {
    const int l = lane_id();
    s_orig[l] = l;
    if (l == 0) *s_flag = 0;
  }
  const int p = a.p, P2 = (p + 1) >> 1;
  const double* arx = a.arx + ((size_t)item * NT + w) * (size_t)(16 * 64 * 2) * P2;
  const double* tw = a.tw + (size_t)f * p * 2;
  const double tau = a.tau;
  const unsigned ldsbase = (unsigned)(uintptr_t)lds;
  double acc[32];
#ifdef HMV_STAMP
  // the stamped stream keeps eight 32-bit phase sums (csrc/gen/k3gen.py, stamp()): folded into the common four here,
  // written out in full behind the per-wave records (a.stamps + 8 * waves) for tools/k3_stamps.py
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0, q0, q1, q2, q3;
  unsigned tl32;
  K3A_BODY_STAMPED(acc, arx, tw, p, w, tau, ldsbase, q0, q1, q2, q3, tl32);
  {
    const unsigned long long d[8] = {q0 & 0xffffffffull, q0 >> 32, q1 & 0xffffffffull, q1 >> 32,
                                     q2 & 0xffffffffull, q2 >> 32, q3 & 0xffffffffull, q3 >> 32};
    tsum[0] = d[0]; tsum[2] = d[1]; tsum[3] = d[2] + d[3] + d[4]; tsum[1] = d[5] + d[6] + d[7];
    unsigned long long now;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
    tlast = (now & ~0xffffffffull) | tl32;
    if (tlast > now) tlast -= 0x100000000ull;
    if (a.stamps && lane_id() == 0) {
      unsigned long long* det = a.stamps + ((size_t)a.n_items * a.F * NT + (size_t)(gw * NT + wv)) * 8;
      for (int k = 0; k < 8; ++k) det[k] = d[k];
    }
  }
#else
  K3A_BODY(acc, arx, tw, p, w, tau, ldsbase);
#endif
  double re[NT][4], im[NT][4];
#pragma unroll
  for (int Ig = 0; Ig < 4; ++Ig)
#pragma unroll
    for (int Jl = 0; Jl < 4; ++Jl) {
      re[Ig][Jl] = acc[2 * (4 * Ig + Jl)];
      im[Ig][Jl] = acc[2 * (4 * Ig + Jl) + 1];
    }
  double2* smem = reinterpret_cast<double2*>(lds);
  double* rsum = reinterpret_cast<double*>(lds + K3A_RSUM);
  tf_outputs<NT, false>(a, item, f, gw, w, wv, re, im, smem, rsum, s_orig, s_flag, nullptr
#ifdef HMV_STAMP
                        , tsum, tlast
#endif
  );
}

// ---------------------------------------------------------------- coefficient packing
// ar (reference layout [item][row][col][lag], rows/cols padded to MP) -> arx[item][w][Ig][Jl][h][lane][2]:
// lane (i, b, j) of wave w finds lags (2h, 2h+1) of element (16*Ig + 4*b + i, 4*(Jl*NT + w) + j) at its own
// 16-byte slot (zero for the padding lag of an odd order).  262 KB per item, once per item instead of once
// per (item, frequency).
template <int NT>
__global__ void __launch_bounds__(256) ar_pack_kernel(const double* ar, double* arx, long long n_items, int p) {
  constexpr int MP = 16 * NT;
  const int P2 = (p + 1) >> 1;
  const long long per_item = (long long)MP * MP * P2;       // double2 slots
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_items * per_item) return;
  const long long item = idx / per_item;
  long long r = idx - item * per_item;
  const int lane = (int)(r & 63); r >>= 6;
  const int h = (int)(r % P2); r /= P2;
  const int Jl = (int)(r & 3); r >>= 2;
  const int Ig = (int)(r % NT);
  const int w = (int)(r / NT);
  const int row = 16 * Ig + 4 * ((lane >> 2) & 3) + (lane >> 4), col = 4 * (Jl * NT + w) + (lane & 3);
  const double* e = ar + ((size_t)item * MP * MP + (size_t)row * MP + col) * p;
  double2 v;
  v.x = e[2 * h];
  v.y = (2 * h + 1 < p) ? e[2 * h + 1] : 0.0;
  reinterpret_cast<double2*>(arx)[idx] = v;
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

// largest frequency grid the in-kernel band sums take: one band's 0 / 1 weights must fit the row worker's LDS block
int tf_band_max_F(int m_pad) {
  switch (m_pad) {
    case 16: return (2 * TfLds<1>::TOTAL) & ~31;
    case 32: return (2 * TfLds<2>::TOTAL) & ~31;
    case 48: return (2 * TfLds<3>::TOTAL) & ~31;
    case 64: return (2 * TfLds<4>::TOTAL) & ~31;
  }
  return 0;
}

long long tf_workspace_doubles(long long n_items, int m_pad, int p) {
  return n_items * (long long)m_pad * m_pad * 2 * ((p + 1) / 2);
}

int launch_tf_inv(const TfArgs& a_in, int m_pad, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
  TfArgs a = a_in;
  if (a.n_items == 0 || a.F == 0) return 0;
  const bool fused = ((a.ff != nullptr || a.bands != nullptr) && a.fuse_items > 0);
  if (fused) {
    if (!a.P || !a.den || !a.wcount || !a.ready || !a.missed || a.lag < 1) return -3;
    if (a.bands && (!a.band_lo || !a.band_hi || a.nb < 1 || a.F % 32 != 0 || a.F > tf_band_max_F(m_pad) || a.A)) return -3;
    // wcount, ready and missed[0] are one zeroed block (capi.hip lays them out back to back)
    if (const hipError_t e = hipMemsetAsync(a.wcount, 0, sizeof(int) * (2 * (size_t)a.n_items + 1), st)) return (int)e;
  } else {
    a.ff = nullptr;
    a.bands = nullptr;
  }
  const long long n = a.n_items * (long long)a.F;
  const dim3 grid((unsigned)n);
  const long long slots = a.n_items * (long long)m_pad * m_pad * ((a.p + 1) / 2);
  const dim3 pgrid((unsigned)((slots + 255) / 256));
  double* arx = const_cast<double*>(a.arx);
  switch (m_pad) {
    case 16: hipLaunchKernelGGL(ar_pack_kernel<1>, pgrid, dim3(256), 0, st, a.ar, arx, a.n_items, a.p); break;
    case 32: hipLaunchKernelGGL(ar_pack_kernel<2>, pgrid, dim3(256), 0, st, a.ar, arx, a.n_items, a.p); break;
    case 48: hipLaunchKernelGGL(ar_pack_kernel<3>, pgrid, dim3(256), 0, st, a.ar, arx, a.n_items, a.p); break;
    case 64: hipLaunchKernelGGL(ar_pack_kernel<4>, pgrid, dim3(256), 0, st, a.ar, arx, a.n_items, a.p); break;
    default: return -1;
  }
  // the caller's events bracket the K3 kernel alone: not the packing kernel in front of it, not norm_missed_kernel behind it
  // (what rocprofv3 reports as the kernel's duration must be what bench.py prices against the roofline)
  if (ev_start)
    if (const hipError_t e = hipEventRecord(ev_start, st)) return (int)e;
  switch (m_pad) {
    case 16: hipLaunchKernelGGL((tf_inv_kernel<1, false>), grid, dim3(64), 0, st, a); break;
    case 32: hipLaunchKernelGGL((tf_inv_kernel<2, false>), grid, dim3(128), 0, st, a); break;
    case 48: hipLaunchKernelGGL((tf_inv_kernel<3, false>), grid, dim3(192), 0, st, a); break;
    case 64:
      if (tuning(5) > 0) {        // measurement knob: dynamic LDS nobody uses, to hold fewer workgroups per CU
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tf_inv64_asm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)tuning(5));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tf_inv_kernel<4, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)tuning(5));
      }
      // the hand-scheduled body has no A(f) output (asked for by the staged API only)
      if ((tuning(3 /* HMV_TUNE_K3_FORM */) != 1 || a.bands != nullptr) && a.A == nullptr)
        hipLaunchKernelGGL(tf_inv64_asm_kernel, grid, dim3(256), (unsigned)tuning(5 /* HMV_TUNE_K3_LDS_PAD */), st, a);
      else
        hipLaunchKernelGGL((tf_inv_kernel<4, false>), grid, dim3(256), (unsigned)tuning(5), st, a);
      break;
  }
  if (ev_stop)
    if (const hipError_t e = hipEventRecord(ev_stop, st)) return (int)e;
  if (fused) {          // rows whose window was not complete in time (normally none) and the rows of the last `lag` windows
    const long long tail_rows = (a.fuse_items > a.lag ? a.lag : a.fuse_items) * a.m;
    const dim3 mgrid((unsigned)(tail_rows > 2048 ? 2048 : (tail_rows < 256 ? 256 : tail_rows)));
    switch (m_pad) {
      case 16: hipLaunchKernelGGL(norm_missed_kernel<1>, mgrid, dim3(64), 0, st, a); break;
      case 32: hipLaunchKernelGGL(norm_missed_kernel<2>, mgrid, dim3(128), 0, st, a); break;
      case 48: hipLaunchKernelGGL(norm_missed_kernel<3>, mgrid, dim3(192), 0, st, a); break;
      case 64: hipLaunchKernelGGL(norm_missed_kernel<4>, mgrid, dim3(256), 0, st, a); break;
    }
  }
  return (int)hipGetLastError();
}

// general complex inverse of a.Zin (kernel layout), a.H = inverse, a.detph = phase of the determinants
int launch_cinv(const TfArgs& a_in, int m_pad, hipStream_t st) {
  TfArgs a = a_in;
  a.ff = nullptr;
  a.bands = nullptr;
  a.fuse_items = 0;
  const long long n = a.n_items * (long long)a.F;
  if (n == 0) return 0;
  const dim3 grid((unsigned)n);
  switch (m_pad) {
    case 16: hipLaunchKernelGGL((tf_inv_kernel<1, true>), grid, dim3(64), 0, st, a); break;
    case 32: hipLaunchKernelGGL((tf_inv_kernel<2, true>), grid, dim3(128), 0, st, a); break;
    case 48: hipLaunchKernelGGL((tf_inv_kernel<3, true>), grid, dim3(192), 0, st, a); break;
    case 64: hipLaunchKernelGGL((tf_inv_kernel<4, true>), grid, dim3(256), 0, st, a); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

}  // namespace hmv
