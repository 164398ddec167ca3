// Shared device helpers for the hypermvar HIP kernels (gfx950 / MI355X only).
//
// Register tile layout used by every kernel in this directory ("D layout"):
//   a wavefront (64 lanes) owns an MP x MP matrix, MP = 16*NT, as NI = MP/4 row blocks (I)
//   times NJ = MP/16 column groups (J); lane l = 16*i + cc holds element
//       (row 4*I + i, col 16*J + cc)            i = l >> 4, cc = l & 15
//   in register [I][J].  This is the C/D layout of v_mfma_f64_4x4x4_4b_f64 when its four
//   4x4 blocks are four neighbouring column blocks of one 4-row strip, and at the same time the
//   B-operand layout of a k-step made of rows 4I..4I+3 (measured lane maps: tools/ubench_f64_4x4.hip):
//       A[b][i][k] on lane 16k + 4b + i,  B[b][k][j] on lane 16k + 4b + j,  D[b][i][j] on lane 16i + 4b + j.
//   v_mfma_f64_4x4x4_4b_f64 issues every ~17 cycles from ONE wave per SIMD (30 flop/clk/SIMD, the
//   measured f64 ceiling of the chip); v_mfma_f64_16x16x4_f64 needs 144 cycles for 4x the flops, so
//   the 4x4x4 form is the one used everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

#define HMV_WAVE 64

namespace hmv {

__device__ __forceinline__ double mfma4(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}
// c - a*b: for the f64 MFMAs the BLGP field is NEG[2:0] (bit 0 negates A; checked on gfx950 with
// tools/ubench_mfma_neg.hip), so a complex product needs no VALU negation of an operand.
__device__ __forceinline__ double mfma4_nega(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 1);
}

// compile-time loop: body(std::integral_constant<int, S>) for S = 0..N-1, always fully expanded so that
// register-array indices stay static (a plain `#pragma unroll` is refused for bodies this large and
// the arrays then land in scratch).
template <typename F, int... S>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, S...>) {
  (f(std::integral_constant<int, S>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// wave-uniform value -> SGPR
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}

__device__ __forceinline__ double shfl_f64(double v, int src_lane) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_ds_bpermute(src_lane << 2, u.i[0]);
  u.i[1] = __builtin_amdgcn_ds_bpermute(src_lane << 2, u.i[1]);
  return u.d;
}

template <int PATTERN>
__device__ __forceinline__ double swizzle_f64(double v) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_ds_swizzle(u.i[0], PATTERN);
  u.i[1] = __builtin_amdgcn_ds_swizzle(u.i[1], PATTERN);
  return u.d;
}

__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src_lane) {
  union { unsigned long long q; int i[2]; } u;
  u.q = v;
  u.i[0] = __builtin_amdgcn_ds_bpermute(src_lane << 2, u.i[0]);
  u.i[1] = __builtin_amdgcn_ds_bpermute(src_lane << 2, u.i[1]);
  return u.q;
}

// max over the 64 lanes of a 64-bit key (all lanes receive the result)
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
  const int l = lane_id();
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    unsigned long long o = shfl_u64(k, l ^ m);
    k = o > k ? o : k;
  }
  return k;
}

// ---- DPP helpers (gfx9-family controls: quad_perm, row_half_mirror, row_mirror) ------------------------
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
// max over each row of 16 lanes, result in all 16 lanes: four v_max_u32 with a DPP source operand (the
// builtin form costs v_mov + s_nop + v_mov_dpp + v_max per step).  The s_nop 1 pairs are the two wait
// states a DPP read needs after a VALU write of its source; hipcc pads nothing inside an asm statement.
__device__ __forceinline__ unsigned row16_max_u32(unsigned v) {
  asm("s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}
// wave-uniform max of a 32-bit key over the 64 lanes (returned in an SGPR): row maxima, then the two
// wave-level DPP broadcasts (lane 15 of row r into row r+1 for rows 1 and 3; lane 31 into rows 2 and 3)
// leave the maximum in lane 63 -- 6 DPP ops + 1 v_readlane instead of 4 v_readlane + scalar/vector maxes.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = row16_max_u32(v);
  asm("v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_update_dpp(u.i[0], u.i[0], CTRL, 0xf, 0xf, false);
  u.i[1] = __builtin_amdgcn_update_dpp(u.i[1], u.i[1], CTRL, 0xf, 0xf, false);
  return u.d;
}
// sum over each row of 16 lanes via DPP (fixed order => deterministic), result in all 16 lanes
__device__ __forceinline__ double row16_sum_dpp(double v) {
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  v += dpp_f64<0x141>(v);
  v += dpp_f64<0x140>(v);
  return v;
}

// sum over the 16 lanes that share l >> 4 (all 16 receive the result)
__device__ __forceinline__ double row16_sum(double v) {
  const int l = lane_id();
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) v += shfl_f64(v, l ^ m);
  return v;
}

}  // namespace hmv
