// ofdis_dev.h -- shared host/device definitions for the gfx950 kernels.
//
// Arithmetic contract (see include/ofdis.h): fp32, no contraction (the library is compiled with
// -ffp-contract=off), IEEE divide/sqrt (hipcc default), reference operation order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ofdis.h"

namespace ofdis {

constexpr int kWave = 64;  // CDNA wavefront

// Per-level geometry, derived exactly as the reference does (oflow.cpp:138-157, patchgrid.cpp:42-48).
struct LevelGeom {
  int level;
  int w, h;            // unpadded level size
  int pad;             // imgpadding
  int tmp_w, tmp_h;    // padded plane size
  int noc;
  float lb, ubw, ubh;  // valid patch-centre range (tmp_lb, tmp_ubw, tmp_ubh)
  // patch grid
  int P, steps, nopw, noph, nop, offw, offh, novals;
  unsigned steps_magic;  // ceil(2^32 / steps) for steps > 1 (0 for steps == 1): n / steps == umulhi(n, magic), 0 <= n < 65536
  size_t plane_elems;  // tmp_w*tmp_h*noc
};

// ----------------------------------------------------------------------------- wave primitives
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x, float old = 0.0f) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
// same, lanes without a source read 0 (bound_ctrl): no "old" register has to be prepared
template <int CTRL>
__device__ __forceinline__ float dpp_mov0(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// lane l <- lane l-1 (lane 0 keeps `old`); lane l <- lane l+1 (lane 63 keeps `old`)
__device__ __forceinline__ float wave_from_prev(float x) { return dpp_mov0<0x138>(x); }  // wave_shr:1, lane 0 <- 0
__device__ __forceinline__ float wave_from_next(float x) { return dpp_mov0<0x130>(x); }  // wave_shl:1, lane 63 <- 0

// The same lane shifts through the LDS crossbar (ds_bpermute_b32; no LDS memory involved).  On gfx950 every DPP (and SDWA,
// v_permlane*, v_readlane, transcendental, packed-fp32, mad24) instruction drops the issuing wavefront out of the 2-clock
// VALU issue rate for the next ~50-100 instructions (profiles/README.md "VALU issue model"); ds_bpermute does not.
// byte address = 4 * source lane; the shift wraps around (lane 0 <- lane 63) instead of filling with zero.
__device__ __forceinline__ float lane_read(float x, int byte_addr) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, x)));
}

// 64-lane butterfly all-reduce.  Order of the additions (this IS the documented reduction order,
// mirrored by oracle/eigen_shim -DOFDIS_SHIM_WAVE64 and oracle_set_reduce_order(1)):
//   pairs at lane distance 1, then 2, 4, 8, 16, 32; every lane ends with the same bits.
// Distances 1,2 are quad permutes, 4 and 8 the half-row / row mirrors (equivalent to xor once the
// smaller groups are uniform), 16 and 32 the gfx950 v_permlane{16,32}_swap.
// v_permlane{16,32}_swap exchange halves between TWO registers (vdst, src):
//   permlane16_swap: odd 16-lane rows of vdst <-> even rows of src
//   permlane32_swap: lanes 32-63 of vdst    <-> lanes 0-31 of src
// With both registers holding x, vdst ends as {r0,r0,r2,r2} / {lo,lo} and src as {r1,r1,r3,r3} /
// {hi,hi}; their sum is the butterfly step.  Written as inline asm because hipcc (ROCm 7.2) lowers
// __builtin_amdgcn_permlane*_swap's second result to the first register (observed on gfx950:
// "v_permlane16_swap v2, v3; v_add_f32 v2, v2, v2").  The s_nop 1 is the two wait states the
// "VALU write -> v_permlane read" hazard needs (cdna_hip_programming.md T21); it must sit inside
// the string because the compiler does not pad around asm statements.
__device__ __forceinline__ float swap16_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float swap32_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
// the first five steps: every 32-lane half reduced on its own (all its lanes end with the half's sum)
__device__ __forceinline__ float half_wave_sum(float x) {
  x = x + dpp_mov<0xB1>(x);   // quad_perm [1,0,3,2]
  x = x + dpp_mov<0x4E>(x);   // quad_perm [2,3,0,1]
  x = x + dpp_mov<0x141>(x);  // row_half_mirror
  x = x + dpp_mov<0x140>(x);  // row_mirror
  return swap16_sum(x);
}
__device__ __forceinline__ float wave_sum(float x) {
  x = x + dpp_mov<0xB1>(x);   // quad_perm [1,0,3,2]
  x = x + dpp_mov<0x4E>(x);   // quad_perm [2,3,0,1]
  x = x + dpp_mov<0x141>(x);  // row_half_mirror
  x = x + dpp_mov<0x140>(x);  // row_mirror
  x = swap16_sum(x);
  x = swap32_sum(x);
  return x;
}

// ----------------------------------------------------------------------------- IEEE divide / sqrt, trimmed
// hipcc expands a correctly rounded fp32 `a / b` to  v_div_scale x2, v_rcp, 6 fma/mul, v_div_fmas, v_div_fixup
// and `sqrtf` to a range scale, v_sqrt, a +-1 ulp residual test, an unscale and a class test.  The scale steps
// only act on operands outside the ranges below, so the kernels whose operands provably stay inside them use
// the same sequences without them -- same bits, 8 instead of 11 and 9 instead of 15 instructions, and the refined
// reciprocal is shared by every quotient with the same denominator:
//   div_by(a, b, rcp_refined(b)) == a / b   for b normal with |b| < 2^126, and a == 0 or 2^-102 <= |a|, and
//                                           |a / b| normal (v_div_scale_f32 is the identity there); zero, inf
//                                           and NaN operands give the IEEE result through v_div_fixup.
//   sqrt_rn(x) == sqrtf(x)                  for x >= 2^-96 (incl. +inf, NaN)
// tests/test_gpu_kernels.py::test_trimmed_div_sqrt checks both against the compiler's expansion on the device.
__device__ __forceinline__ float rcp_refined(float b) {
  const float r0 = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, r0, 1.0f);
  return __builtin_fmaf(e, r0, r0);
}
// Reciprocal and square root without a transcendental instruction (v_rcp_f32 / v_sqrt_f32 are among the instructions that
// cost a wavefront its 2-clock issue rate, see lane_read; in the fused TV kernel's stream a v_sqrt_f32 costs ~120 clocks
// where the 20 plain instructions below cost ~45).
//   rcp_newton(b):  integer seed (relative error <= 5.1 %) and four Newton steps, the last two at rounding level: the
//                   correctly rounded 1/b except when 1/b lies within ~2^-47 of a rounding boundary (measured: 2e-7 of
//                   random operands, then 1 ulp off) -- the quality of v_rcp_f32 + one Newton step (rcp_refined), which
//                   is what the quotient sequences below are specified for.  The one systematic exception of the Newton
//                   step, a denominator whose significand is all ones (the step ties to even, downwards), is patched by
//                   the integer increment at the end (Markstein: with the correctly rounded reciprocal the final quotient
//                   step is exact for every numerator; 1.0f / 1.9999999f is the counter-example without the patch).
//                   b normal, 2^-120 <= |b| <= 2^120, either sign.  nb = -b.
//   rcp_from(b, nb, y): the same from a seed y with relative error <= 1e-4 (two steps).
//   sqrt_newton(x, y): correctly rounded sqrt(x) for normal x >= 2^-96: integer seed of 1/sqrt(x) (3.4 %), two Newton steps
//                   (4.7e-6), s = x*y corrected once (error 2e-11 before rounding), then the same +-1 ulp residual
//                   selection as sqrt_rn.  y returns the 1/sqrt(x) estimate: the seed of the reciprocal that always
//                   follows a square root in the TV system (quotients by a norm).
__device__ __forceinline__ float rcp_allones_patch(float r, float b) {
  const int bi = __builtin_bit_cast(int, b);
  return __builtin_bit_cast(float, __builtin_bit_cast(int, r) + ((bi & 0x7fffff) == 0x7fffff ? 1 : 0));
}
__device__ __forceinline__ float rcp_from(float b, float nb, float y) {
  float r = y;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float e = __builtin_fmaf(nb, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
  }
  return rcp_allones_patch(r, b);
}
__device__ __forceinline__ float rcp_newton(float b, float nb) {
  float r = __builtin_bit_cast(float, 0x7EF31000 - __builtin_bit_cast(int, b));
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float e = __builtin_fmaf(nb, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
  }
  return rcp_from(b, nb, r);
}
__device__ __forceinline__ float sqrt_newton(float x, float& y_out) {
  float y = __builtin_bit_cast(float, 0x5f376400 - (__builtin_bit_cast(int, x) >> 1));
  const float h = 0.5f * x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float t = y * y;
    const float u = __builtin_fmaf(-h, t, 1.5f);
    y = y * u;
  }
  float s = x * y;
  const float e = __builtin_fmaf(-s, s, x);
  s = __builtin_fmaf(e, 0.5f * y, s);
  const float sd = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
  const float su = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
  const float ed = __builtin_fmaf(-sd, s, x);
  const float eu = __builtin_fmaf(-su, s, x);
  float r = (ed <= 0.0f) ? sd : s;
  r = (eu > 0.0f) ? su : r;
  y_out = y;
  return r;
}
__device__ __forceinline__ float div_by(float a, float b, float r) {
  float q = a * r;
  float e = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(e, r, q);
  e = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(e, r, q);
  return __builtin_amdgcn_div_fixupf(q, b, a);
}
__device__ __forceinline__ float div_rn(float a, float b) { return div_by(a, b, rcp_refined(b)); }
// div_by without v_div_fixup_f32 (which only acts on zero / infinite / NaN operands): for a FINITE numerator and a normal,
// finite denominator the refined quotient already is the result -- a == 0 gives 0 (possibly with the other sign of
// zero, which no consumer in the fused TV kernel can observe: every quotient there is multiplied into sums that are
// compared by value).  Saves one 8-byte-encoded VALU instruction per quotient.
// nb = -b is passed in (computed once per denominator): with the negation a plain operand the residuals can be
// v_fmac_f32 (4-byte encoding, destination = addend) where the numerator dies, instead of v_fma_f32 with a source
// modifier (8-byte encoding = two issue slots on gfx950, profiles/README.md).
__device__ __forceinline__ float div_by_finite(float a, float nb, float r) {
  float q = a * r;
  float e = __builtin_fmaf(nb, q, a);
  q = __builtin_fmaf(e, r, q);
  e = __builtin_fmaf(nb, q, a);
  return __builtin_fmaf(e, r, q);
}
__device__ __forceinline__ float sqrt_rn(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sd = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
  const float su = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
  const float ed = __builtin_fmaf(-sd, s, x);
  const float eu = __builtin_fmaf(-su, s, x);
  float r = (ed <= 0.0f) ? sd : s;
  r = (eu > 0.0f) ? su : r;
  return r;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// "diag" plane layout used for the SOR solver's operands (7 system planes, du, dv): pixel (x,y) of a
// w x h plane lives at ((x+y) mod w)*h + y, i.e. wrapped anti-diagonal d = (x+y) mod w is ONE
// contiguous row of h floats (a bijection onto w*h, no padding).  The wavefront SOR reads/writes
// one such row per step (ofdis_sor.hip).
// (0 <= x < w, 0 <= y < h: x + y wraps at most once unless h > w, so the general modulo -- ~35 instructions for a
// run-time divisor -- is kept off the common path.)
__host__ __device__ __forceinline__ int diag_index(int x, int y, int w, int h) {
  int d = x + y;
  if (d >= w) d -= w;
  if (d >= w) d %= w;
  return d * h + y;
}

// Blocks of one frame stay on one XCD: the dispatcher places block n on XCD n % 8 (observed, used for L2
// affinity only -- correctness does not depend on it), so block n works on frame (n/8/bpf)*8 + n%8.
// Launch ((nframes+7)/8)*8*blocks_per_frame blocks and skip frame >= nframes.
// With fewer than 8 frames that mapping would leave whole XCDs without work (one 1080p frame on 32 of the 256 CUs:
// measured 8x slower), so there the blocks of a frame are simply consecutive and spread over all XCDs.
__device__ __forceinline__ void xcd_frame_map(int n, int blocks_per_frame, int nframes, int& frame, int& blk) {
  if (nframes < 8) {
    frame = n / blocks_per_frame;
    blk = n - frame * blocks_per_frame;
    return;
  }
  const int xcd = n & 7;
  const int m = n >> 3;
  frame = (m / blocks_per_frame) * 8 + xcd;
  blk = m % blocks_per_frame;
}

}  // namespace ofdis
