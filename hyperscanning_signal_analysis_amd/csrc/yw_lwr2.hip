// K2, the block Levinson-Whittle recursion of yw_lwr.hip (same equations, same order of the tile products, same guard)
// as a SOFTWARE PIPELINE.  Replaces `ar_coeff` (/root/reference/src/mtmvar.py:90-123) and, through the V_q of every
// order, `mvar_criterion` (:551-601).
//
// What was wrong with the first form (profiles/r03_kernel_stats.csv, r03_pmc_*.csv: 1.72 ms and 7.3 GB per 599-window
// step): one workgroup walks ~130 dependent tile operations, and every product exposed two global-load latencies
// and four workgroup barriers around 2 us of MFMA time -- ~10 us per operation, so the kernel took as long for 300
// windows as for 600 -- and it fetched BOTH operands of every product and every subtrahend from global memory: ~380 tile
// moves of 32 KB per window.  Here:
//   * the X operand of every product is a register tile (A_{q+1}, B_{q+1}, the tile just updated, D and D^T): each wave
//     parks its own 16-row strip in LDS, quarter of the k range by quarter, with no barrier (nobody else reads it);
//   * the Y operands stream through two quarter images in LDS; the loads of quarter t + 2 -- of the NEXT product when
//     this one runs out -- are in flight behind the MFMAs of quarters t and t + 1, so a phase is park, one barrier,
//     64 MFMAs per wave, and the pipeline never drains inside an order (both tile inverses of an order come first);
//   * a subtrahend is the accumulator's initial value and the product is subtracted by the
//     MFMA itself (negated A operand): no separate tile subtraction, no temporary.
// Per lower lag k of order q that is 6 tile reads and 2 writes instead of 7 + 2 + temporaries; per window ~300 tile
// moves.  LDS 42 KB and <= 168 VGPRs: three workgroups per CU, i.e. all 599 windows of the north-star batch resident.
// Results equal the first form to rounding (the accumulation starts from the subtrahend instead of from zero).
#include "yw_common.h"

namespace hmv {

#ifndef HMV_LWR_GUARD
#define HMV_LWR_GUARD 1e-7
#endif
#ifndef HMV_LWR2_STORES          // 0: compiler stores, 1: uncounted asm stores, 2: asm stores drained tile by tile (debug)
#define HMV_LWR2_STORES 0
#endif

// Diagnostic build (-DHMV_LWR2_STAMP, tools/dbg/k2_ab.py): cycles of wave 0 per section, summed over the orders, left in
// the window's last scratch tile (which only holds the guard word at its end).
#ifdef HMV_LWR2_STAMP
#define LWR2_T(idx)                                                                    \
  do {                                                                                 \
    unsigned long long t_;                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
    tsum[idx] += t_ - tlast;                                                           \
    tlast = t_;                                                                        \
  } while (0)
#else
#define LWR2_T(idx) do { } while (0)
#endif

template <int NT, bool VQ>
__global__ void __launch_bounds__(256, 3) yw_lwr2_kernel(YwArgs a) {
  constexpr int MP = 16 * NT, KQ = MP / 4, SQ = KQ + 6, NIW = NT, NJ = NT, TILE = MP * MP;
  constexpr int NVQ = (MP * KQ / 2 + 255) / 256;                   // 16-byte loads of a quarter tile per thread
  constexpr bool FULLQ = (NVQ * 256 == MP * KQ / 2);
  constexpr int SI = YwCfg<NT>::S;
  constexpr int GEMM_D = 3 * MP * SQ, INV_D = MP * SI, BUF_D = GEMM_D > INV_D ? GEMM_D : INV_D;
  __shared__ __attribute__((aligned(16))) double buf[BUF_D];      // X quarter + two Y quarters | the inverse's image
  __shared__ double Pb[MP * 4];
  __shared__ double Nb[2 * MP * 4];
  __shared__ int s_info;
  __shared__ double s_ld[4];
  __shared__ double s_pm[2 + 2 * 4];
  __shared__ int s_guard;
  double* Xq = buf;
  double* Yq0 = buf + MP * SQ;
  double* Yq1 = buf + 2 * MP * SQ;
  const int wv = uni(threadIdx.x >> 6);
  const int p = a.p;
  const long long item = blockIdx.x;
  // scratch tiles of this window: A and B in two generations, Vf, Vb, their inverses, D in two generations
  double* ws = a.ws + (size_t)item * yw_ws_tiles_d(p) * TILE;
  double* Agen[2] = {ws, ws + (size_t)p * TILE};
  double* Bgen[2] = {ws + (size_t)2 * p * TILE, ws + (size_t)3 * p * TILE};
  double* Vf = ws + (size_t)4 * p * TILE;
  double* Vb = Vf + TILE;
  double* VfI = Vb + TILE;
  double* VbI = VfI + TILE;
  double* Dgen[2] = {VbI + TILE, VbI + 2 * TILE};
  const double* R = a.R + (size_t)item * (p + 1) * TILE;
  if (threadIdx.x == 0) {
    s_info = 0;
    s_guard = 0;
  }

  // lane / thread indices re-derived from an opaque lane id in every helper (yw_solve.hip: otherwise every staging offset
  // stays live across the kernel)
  auto lane = [&]() __attribute__((always_inline)) {
    int lo;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo));
    return lo;
  };
  // Thread-constant offsets of the staging helpers, computed ONCE (each phase used to re-derive them from the lane id:
  // ~150 integer instructions per phase, which with one wave per SIMD is as long as the phase's 64 MFMAs -- the stamped
  // build showed 1.4 k cycles of k-step loop and 1.4 k cycles of everything else per phase).
  const int tid0 = lane() + 64 * wv;
  int fo_plain[NVQ], fo_tr[NVQ], po_plain[NVQ], po_tr[NVQ];
#pragma unroll
  for (int r = 0; r < NVQ; ++r) {
    const int idx = tid0 + 256 * r;
    const int row = idx / (KQ / 2), c2 = idx - row * (KQ / 2);
    const int k = idx / (MP / 2), d2 = idx - k * (MP / 2);
    fo_plain[r] = row * MP + 2 * c2;            // + t * KQ
    fo_tr[r] = k * MP + 2 * d2;                 // + t * KQ * MP
    po_plain[r] = row * SQ + 2 * c2;
    po_tr[r] = (2 * d2) * SQ + k;
  }
  const int ln = lane();
  const int xrow0 = (4 * wv * NT + (ln >> 4)) * SQ + (ln & 15);      // parkx: + 4 * ii * SQ + 16 * J - t * KQ
  const int xa0 = (4 * wv * NT + (ln & 3)) * SQ + (ln >> 4);         // gemmq A operand
  const int yb0 = (ln & 15) * SQ + (ln >> 4);                        // gemmq B operand
  // ---- Y operand, quarter t of the k range.  Plain image: Y'[row][k] = src[row][t*KQ + k];  transposed image:
  // Y'[col][k] = src[t*KQ + k][col].  (product() computes acc +- X' Y'^T.)
  auto fetchq = [&](f64x2 (&v)[NVQ], const double* src, int t, bool tr) __attribute__((always_inline)) {
    const double* s0 = src + (tr ? t * KQ * MP : t * KQ);
#pragma unroll
    for (int r = 0; r < NVQ; ++r) {
      if (FULLQ || tid0 + 256 * r < MP * KQ / 2) v[r] = *reinterpret_cast<const f64x2*>(s0 + (tr ? fo_tr[r] : fo_plain[r]));
    }
  };
  auto parkq = [&](double* dst, const f64x2 (&v)[NVQ], bool tr) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < NVQ; ++r) {
      if (FULLQ || tid0 + 256 * r < MP * KQ / 2) {
        if (!tr) {
          double* d = dst + po_plain[r];
          d[0] = v[r].x;
          d[1] = v[r].y;
        } else {
          dst[po_tr[r]] = v[r].x;
          dst[po_tr[r] + SQ] = v[r].y;
        }
      }
    }
  };
  // ---- X operand: this wave's row strip of a register tile, columns t*KQ .. (read back by this wave only)
  auto parkx = [&](const double (&x)[NIW][NJ], int t) __attribute__((always_inline)) {
    const int cc = ln & 15;
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) {
        const int col = 16 * J + cc - t * KQ;
        if (col >= 0 && col < KQ) Xq[xrow0 + 4 * ii * SQ + 16 * J - t * KQ] = x[ii][J];
      }
  };
  auto gemmq = [&](double (&acc)[NIW][NJ], const double* Yq, auto negc) __attribute__((always_inline)) {
    constexpr bool NEG = decltype(negc)::value;
    const double* xa = Xq + xa0;
    const double* yb = Yq + yb0;
    constexpr int NS = KQ / 4;
    double av[2][NIW], bv[2][NJ];
    auto rd = [&](int set, int k0) __attribute__((always_inline)) {
#pragma unroll
      for (int ii = 0; ii < NIW; ++ii) av[set][ii] = xa[4 * ii * SQ + k0];
#pragma unroll
      for (int J = 0; J < NJ; ++J) bv[set][J] = yb[16 * J * SQ + k0];
    };
    rd(0, 0);
    static_for<NS>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s2 = decltype(sc)::value, cur = s2 & 1;
      if constexpr (s2 + 1 < NS) rd(cur ^ 1, 4 * (s2 + 1));          // the next step's operands, requested first
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
        for (int J = 0; J < NJ; ++J)
          acc[ii][J] = NEG ? mfma4_nega(av[cur][ii], bv[cur][J], acc[ii][J]) : mfma4(av[cur][ii], bv[cur][J], acc[ii][J]);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // ---- the pipeline.  Invariant on entry to product(): quarters 0 and 1 of its Y operand are in flight in vy0 / vy1
  // (prime(), or the previous product's last two phases).  Phase t: park Y quarter t in image t & 1 (last read two
  // phases ago, i.e. before the previous phase's barrier), issue the loads of quarter t + 2 (of the next product from
  // t = 2 on), park the X quarter, ONE barrier, MFMAs.  acc +- X Y'^T.
#ifdef HMV_LWR2_STAMP
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#endif
  f64x2 vy0[NVQ], vy1[NVQ];
  auto prime = [&](const double* src, bool tr) __attribute__((always_inline)) {
    fetchq(vy0, src, 0, tr);
    fetchq(vy1, src, 1, tr);
  };
#ifdef HMV_LWR2_STAMP
  auto gemmq_timed = [&](double (&acc)[NIW][NJ], const double* Yq, auto negc) __attribute__((always_inline)) {
    unsigned long long ta, tb;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ta)::"memory");
    gemmq(acc, Yq, negc);
    asm volatile("s_nop 0" ::"v"(acc[NIW - 1][NJ - 1]));     // (the last MFMA has been issued)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb)::"memory");
    tsum[7] += tb - ta;
  };
#define LWR2_GEMM gemmq_timed
#else
#define LWR2_GEMM gemmq
#endif
  auto product = [&](double (&acc)[NIW][NJ], const double (&x)[NIW][NJ], const double* ysrc, bool ytr, const double* nsrc,
                     bool ntr, auto negc) __attribute__((always_inline)) {
    parkq(Yq0, vy0, ytr);
    fetchq(vy0, ysrc, 2, ytr);
    parkx(x, 0);
    __syncthreads();
    LWR2_GEMM(acc, Yq0, negc);
    parkq(Yq1, vy1, ytr);
    fetchq(vy1, ysrc, 3, ytr);
    parkx(x, 1);
    __syncthreads();
    LWR2_GEMM(acc, Yq1, negc);
    parkq(Yq0, vy0, ytr);
    if (nsrc) fetchq(vy0, nsrc, 0, ntr);
    parkx(x, 2);
    __syncthreads();
    LWR2_GEMM(acc, Yq0, negc);
    parkq(Yq1, vy1, ytr);
    if (nsrc) fetchq(vy1, nsrc, 1, ntr);
    parkx(x, 3);
    __syncthreads();
    LWR2_GEMM(acc, Yq1, negc);
  };
  constexpr std::integral_constant<bool, false> POS{};
  constexpr std::integral_constant<bool, true> NEGA{};

  auto zero = [&](double (&v)[NIW][NJ]) __attribute__((always_inline)) {
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) v[ii][J] = 0.0;
  };
  // Tile stores are asm statements the compiler does not count.  On gfx9-family targets loads and stores share ONE
  // counter (vmcnt) and may retire out of order with respect to each other, so with a store in flight hipcc turns every
  // wait for a load into s_waitcnt vmcnt(0) -- i.e. each product would wait for the prefetches issued just before it
  // (measured: the pipeline 30 % SLOWER than the unpipelined form).  Invisible stores only make the compiler's counted
  // waits longer (the hardware count includes them), never shorter than a load needs: loads retire in order among
  // themselves.  What the compiler can no longer do is drain them before a barrier that publishes the tiles to the
  // other waves: publish() does (end of every order).  8-byte stores: no late-read hazard on the data registers.  But
  // hipcc's hazard recognizer does not look inside asm statements either: a VMEM instruction that reads the result of
  // an MFMA needs up to 18 wait states behind it (gfx940 family), which the s_nops supply -- without them a store
  // wrote what the register held BEFORE the product's last MFMAs (singular tiles two orders later).
  auto store_tile = [&](double* dst, const double (&v)[NIW][NJ]) __attribute__((always_inline)) {
    const int l = lane(), i = l >> 4, cc = l & 15;
#if HMV_LWR2_STORES == 0
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) dst[(size_t)(4 * (wv * NT + ii) + i) * MP + 16 * J + cc] = v[ii][J];
#else
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) {
        double* d = dst + (size_t)(4 * (wv * NT + ii) + i) * MP + 16 * J + cc;
        // (the wait states sit INSIDE every statement: the MFMA that produces v[ii][J] may be scheduled right in front of it)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3\n\tglobal_store_dwordx2 %0, %1, off" ::"v"(d), "v"(v[ii][J]) : "memory");
      }
#if HMV_LWR2_STORES == 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#endif
  };
  auto publish = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  // tile (or its transpose) -> this workgroup's register tile
  auto load_tile = [&](double (&v)[NIW][NJ], const double* src, bool tr) __attribute__((always_inline)) {
    const int l = lane(), i = l >> 4, cc = l & 15;
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) {
        const int row = 4 * (wv * NT + ii) + i, col = 16 * J + cc;
        v[ii][J] = tr ? src[(size_t)col * MP + row] : src[(size_t)row * MP + col];
      }
  };
  // inverse of the SPD register tile g -> global tile `out`; conditioning guard; optional log det
  auto invert = [&](const double (&g)[NIW][NJ], double* out, double* logdet, int info_base) __attribute__((always_inline)) {
    const int l = lane(), i = l >> 4, cc = l & 15;
    __syncthreads();            // the pipeline's images are no longer read
#pragma unroll
    for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
      for (int J = 0; J < NJ; ++J) buf[(4 * (wv * NT + ii) + i) * SI + 16 * J + cc] = g[ii][J];
    __syncthreads();
    spd_inverse_coop<NT, SI>(buf, Pb, Nb, &s_info, s_ld, out, logdet, info_base, s_pm);
    if (threadIdx.x == 0 && !(s_pm[0] >= HMV_LWR_GUARD * s_pm[1])) s_guard = 1;      // also catches NaN
    __syncthreads();            // the inverse is in global memory for the whole workgroup
  };

  double g[NIW][NJ];
  // ---- order 0: Vf = Vb = C(0) = R_0 (symmetric), D_0 = C(1) = R_1^T
  load_tile(g, R, false);
  store_tile(Vf, g);
  store_tile(Vb, g);
  load_tile(g, R + TILE, true);
  store_tile(Dgen[0], g);
  publish();
  for (int q = 0; q < p; ++q) {
    const double* Ao = Agen[q & 1];
    const double* Bo = Bgen[q & 1];
    double* An = Agen[(q + 1) & 1];
    double* Bn = Bgen[(q + 1) & 1];
    const double* Dc = Dgen[q & 1];
    double* Dn = Dgen[(q + 1) & 1];
    const bool last = (q == p - 1);
    LWR2_T(0);       // setup / end-of-order barrier
    // ---- inverses of the two error covariances of order q (log det Vf_q is the criterion's term of order q)
    load_tile(g, Vb, false);
    invert(g, VbI, nullptr, q * MP);
    if (!last || (VQ && q >= 1)) {       // (the last order needs Vf^-1 only for its log det)
      load_tile(g, Vf, false);
      invert(g, VfI, (VQ && q >= 1) ? a.Vq_logdet + (size_t)item * p + (q - 1) : nullptr, q * MP);
    }
    LWR2_T(1);       // inverses
    // ---- A_{q+1} = D Vb^-1 (Vb^-1 is symmetric: X Y'^T with Y' = Vb^-1);  Vf <- Vf - A_{q+1} D^T
    double anew[NIW][NJ];
    {
      double xd[NIW][NJ], vacc[NIW][NJ];
      prime(VbI, false);
      load_tile(xd, Dc, false);
      load_tile(vacc, Vf, false);
      zero(anew);
      product(anew, xd, VbI, false, Dc, false, POS);
      store_tile(An + (size_t)q * TILE, anew);
      const double* nx = (q > 0) ? Bo + (size_t)(q - 1) * TILE : (last ? nullptr : R + TILE);
      product(vacc, anew, Dc, false, nx, q > 0, NEGA);
      store_tile(Vf, vacc);
    }
    LWR2_T(2);       // A_{q+1}, Vf
    // ---- lower lags, forward side: A'_k = A_k - A_{q+1} B_{q-1-k}   (k = 0 .. q-1: lag k + 1)
    {
      double acc[NIW][NJ];
      for (int k = 0; k < q; ++k) {
        const bool more = (k + 1 < q);
        // The accumulator's initial value is loaded here, not one product ahead: a fourth live tile does not fit the
        // 168 registers (the compiler spilled the prefetched tile load by load, each behind an s_waitcnt vmcnt(0)).
        load_tile(acc, Ao + (size_t)k * TILE, false);
        // Y of the next product: the next lag's B tile, or the first tile of the D pass
        const double* nx = more ? Bo + (size_t)(q - 2 - k) * TILE : (last ? nullptr : R + (size_t)(q + 1) * TILE);
        product(acc, anew, Bo + (size_t)(q - 1 - k) * TILE, true, nx, more, NEGA);
        store_tile(An + (size_t)k * TILE, acc);
      }
    }
    LWR2_T(3);       // forward side
    // ---- the next partial correlation D' = C(q+2) - sum_{k=0..q} A'_k C(q+1-k), C(l) = R_l^T (A'_q = A_{q+1}), as a
    // pass of its own over the tiles just stored: with its accumulator live next to A_{q+1}, the tile being updated and
    // the next one's initial value the kernel does not fit 168 registers (411 spilled).  Every lane reads back exactly
    // the elements it stored itself.
    if (!last) {
      double dacc[NIW][NJ], x[NIW][NJ];
      zero(dacc);
      for (int k = 0; k <= q; ++k) {
        const bool more = (k < q);
        load_tile(x, An + (size_t)k * TILE, false);
        product(dacc, x, R + (size_t)(q + 1 - k) * TILE, false, more ? R + (size_t)(q - k) * TILE : VfI, false, POS);
      }
      load_tile(g, R + (size_t)(q + 2) * TILE, true);
#pragma unroll
      for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
        for (int J = 0; J < NJ; ++J) g[ii][J] -= dacc[ii][J];
      store_tile(Dn, g);
    }
    LWR2_T(4);       // D pass
    if (!last) {
      // ---- B_{q+1} = D^T Vf^-1;  Vb <- Vb - B_{q+1} D;  backward side: B'_k = B_k - B_{q+1} A_{q-1-k}
      double bnew[NIW][NJ], acc[NIW][NJ];
      {
        double xd[NIW][NJ];
        load_tile(xd, Dc, true);
        load_tile(acc, Vb, false);
        zero(bnew);
        product(bnew, xd, VfI, false, Dc, true, POS);
        store_tile(Bn + (size_t)q * TILE, bnew);
        product(acc, bnew, Dc, true, (q > 0) ? Ao + (size_t)(q - 1) * TILE : nullptr, true, NEGA);
        store_tile(Vb, acc);
      }
      for (int k = 0; k < q; ++k) {
        const bool more = (k + 1 < q);
        load_tile(acc, Bo + (size_t)k * TILE, false);
        product(acc, bnew, Ao + (size_t)(q - 1 - k) * TILE, true, more ? Ao + (size_t)(q - 2 - k) * TILE : nullptr, true, NEGA);
        store_tile(Bn + (size_t)k * TILE, acc);
      }
    }
    LWR2_T(5);       // B_{q+1}, Vb, backward side
    publish();                                           // generation q + 1 is in global memory for the whole workgroup
  }
  LWR2_T(0);
  if (VQ) {                                              // log det Vf_p
    load_tile(g, Vf, false);
    invert(g, VfI, a.Vq_logdet + (size_t)item * p + (p - 1), p * MP);
  }
  // ---- outputs: V = Vf_p, ar[item][row][col][k] = A_{k+1}[row][col] (lag fastest: the reference's (m, m, p) layout)
  load_tile(g, Vf, false);
  store_tile(a.V + (size_t)item * TILE, g);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) {
    a.info[item] = s_info;
    // (a singular window stays singular: nothing to re-solve)
    *yw_guard_ptr(a.ws, item, p, TILE) = (s_info == 0) ? s_guard : 0;
  }
  const double* Af = Agen[p & 1];
  double* ar = a.ar + (size_t)item * TILE * p;
  const int total = TILE * p;
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int e = idx / p, k = idx - e * p;
    ar[idx] = Af[(size_t)k * TILE + e];
  }
#ifdef HMV_LWR2_STAMP
  LWR2_T(6);         // log det, V, emit
  if (threadIdx.x == 0) {
    unsigned long long* out = reinterpret_cast<unsigned long long*>(ws + (size_t)(yw_ws_tiles_d(p) - 1) * TILE);
    for (int k = 0; k < 8; ++k) out[k] = tsum[k];
  }
#endif
}

int launch_yw_lwr2(const YwArgs& a, int m_pad, hipStream_t st) {
  if (a.n_items == 0) return 0;
  const dim3 grid((unsigned)a.n_items), block(256);
  const bool vq = (a.Vq_logdet != nullptr);
  switch (m_pad) {
    case 16: if (vq) hipLaunchKernelGGL((yw_lwr2_kernel<1, true>), grid, block, 0, st, a);
             else hipLaunchKernelGGL((yw_lwr2_kernel<1, false>), grid, block, 0, st, a); break;
    case 32: if (vq) hipLaunchKernelGGL((yw_lwr2_kernel<2, true>), grid, block, 0, st, a);
             else hipLaunchKernelGGL((yw_lwr2_kernel<2, false>), grid, block, 0, st, a); break;
    case 48: if (vq) hipLaunchKernelGGL((yw_lwr2_kernel<3, true>), grid, block, 0, st, a);
             else hipLaunchKernelGGL((yw_lwr2_kernel<3, false>), grid, block, 0, st, a); break;
    case 64: if (vq) hipLaunchKernelGGL((yw_lwr2_kernel<4, true>), grid, block, 0, st, a);
             else hipLaunchKernelGGL((yw_lwr2_kernel<4, false>), grid, block, 0, st, a); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

}  // namespace hmv
