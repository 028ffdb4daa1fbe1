// ofdis_dev.h -- shared host/device definitions for the gfx950 kernels.
//
// Arithmetic contracts (see include/ofdis.h).  Every kernel file is compiled TWICE from the same source
// (of_dis_amd/build.py), once per contract, into its own object set and namespace:
//   ofdis::exact  (-DOFDIS_CONTRACT=0 -ffp-contract=off): fp32, every operation separately rounded, correctly rounded
//                 divide / sqrt, the reference's operation order -- bit-identical to the reference build;
//   ofdis::fused  (-DOFDIS_CONTRACT=1 -ffp-contract=fast): the tolerance contract of BASELINE.json's north star (EPE <
//                 1e-3 px against the reference): multiply-adds contract into v_fma_f32, quotients and roots are the
//                 hardware's 1-ulp v_rcp_f32 / v_rsq_f32 / v_sqrt_f32, sums may be re-associated where `kFusedContract`
//                 selects a shorter form.  Same algorithm, same reduction shapes, same control flow.
// ofdis_capi.hip picks the launcher set per context (ofdis_tuning::contract).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/ofdis.h"

#ifndef OFDIS_CONTRACT
#define OFDIS_CONTRACT 0
#endif
#if OFDIS_CONTRACT
#define OFDIS_KNS fused
#else
#define OFDIS_KNS exact
#endif

namespace ofdis {

constexpr bool kFusedContract = OFDIS_CONTRACT != 0;  // the contract of THIS translation unit
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

namespace OFDIS_KNS {  // device helpers: one copy per contract (same source, different rounding of a * b + c)

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

// ----------------------------------------------------------------------------- divide / sqrt
// EXACT contract.  hipcc expands a correctly rounded fp32 `a / b` to  v_div_scale x2, v_rcp, 6 fma/mul, v_div_fmas,
// v_div_fixup and `sqrtf` to a range scale, v_sqrt, a +-1 ulp residual test, an unscale and a class test.  The scale steps
// only act on operands outside the ranges below, so the kernels whose operands provably stay inside them use
// the same sequences without them -- same bits, 8 instead of 11 and 9 instead of 15 instructions, and the refined
// reciprocal is shared by every quotient with the same denominator:
//   div_by(a, b, rcp_refined(b)) == a / b   for b normal with |b| < 2^126, and a == 0 or 2^-102 <= |a|, and
//                                           |a / b| normal (v_div_scale_f32 is the identity there); zero, inf
//                                           and NaN operands give the IEEE result through v_div_fixup.
//   sqrt_rn(x) == sqrtf(x)                  for x >= 2^-96 (incl. +inf, NaN)
// tests/test_gpu_kernels.py::test_trimmed_div_sqrt checks both against the compiler's expansion on the device.
// FUSED contract.  The same names are the hardware approximations (v_rcp_f32 / v_sqrt_f32 / v_rsq_f32: 1 ulp): a quotient
// is ONE multiply by the shared reciprocal, a root one instruction.  The operand ranges above are the same, so no
// denormal / overflow handling is needed either (a zero denominator gives inf / NaN like the division would).
__device__ __forceinline__ float rcp_refined(float b) {
  const float r0 = __builtin_amdgcn_rcpf(b);
  if constexpr (kFusedContract) return r0;
  const float e = __builtin_fmaf(-b, r0, 1.0f);
  return __builtin_fmaf(e, r0, r0);
}
__device__ __forceinline__ float div_by(float a, float b, float r) {
  float q = a * r;
  if constexpr (kFusedContract) return q;
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
  if constexpr (kFusedContract) return q;
  float e = __builtin_fmaf(nb, q, a);
  q = __builtin_fmaf(e, r, q);
  e = __builtin_fmaf(nb, q, a);
  return __builtin_fmaf(e, r, q);
}
__device__ __forceinline__ float sqrt_rn(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  if constexpr (kFusedContract) return s;
  const float sd = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
  const float su = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
  const float ed = __builtin_fmaf(-sd, s, x);
  const float eu = __builtin_fmaf(-su, s, x);
  float r = (ed <= 0.0f) ? sd : s;
  r = (eu > 0.0f) ? su : r;
  return r;
}
// a / sqrt(x): exact contract = the two correctly rounded operations of the reference; fused = a * v_rsq_f32(x)
__device__ __forceinline__ float div_by_sqrt(float a, float x) {
  if constexpr (kFusedContract) return a * __builtin_amdgcn_rsqf(x);
  return div_rn(a, sqrt_rn(x));
}

}  // namespace OFDIS_KNS

// norm > outlierthresh (patch.cpp:199) is tested on the SQUARED norm: sqrt is monotonic and correctly rounded, so
// sqrtf(x) > t  <=>  x > X with X = the largest float whose square root rounds to <= t (host sqrtf is correctly rounded)
inline float outlier_sq_threshold(float t) {
  if (!(t >= 0.0f) || !isfinite(t)) return t;  // NaN / negative / inf: keep the comparison's outcome (never / always / never)
  float x = t * t;
  while (x > 0.0f && sqrtf(x) > t) x = nextafterf(x, 0.0f);
  while (sqrtf(nextafterf(x, INFINITY)) <= t) x = nextafterf(x, INFINITY);
  return x;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Internal layout of the patch results (the patch kernels write them, the densify kernels read them; per frame nop * 2 and
// nop * novals floats as in the reference, but ordered for the accesses of a wavefront, which works along a GRID ROW --
// sixteen neighbouring patches in the patch kernels, consecutive pixels of an image row in the densify kernels):
//   p       [gy][gx][2]                       the reference's index is ip = gx * noph + gy (patchgrid.cpp:62-69)
//   pweight [gy][patch row r][gx][P * noc]    entry (r, col, c) of patch (gx, gy) at column col * noc + c of its row
// so that one patch row of the neighbouring patches of a grid row is ONE contiguous run: the patch kernel's weight stores
// and the densification's weight loads touch 2-4 cache lines per instruction instead of one line per patch (round 4).
// ofdis_patchgrid_level() returns p in the reference's order through launch_patch_p_reference_order().
__host__ __device__ __forceinline__ int patch_slot(const LevelGeom& g, int gx, int gy) { return gy * g.nopw + gx; }
__host__ __device__ __forceinline__ size_t pweight_row(const LevelGeom& g, int gx, int gy, int r) {
  return ((size_t)(gy * g.P + r) * g.nopw + gx) * (size_t)(g.P * g.noc);
}
// entry with the reference's linear index k = (r * P + col) * noc + c of patch (gx, gy)
__host__ __device__ __forceinline__ size_t pweight_entry(const LevelGeom& g, int gx, int gy, int k) {
  const int rowlen = g.P * g.noc, r = k / rowlen;
  return pweight_row(g, gx, gy, r) + (size_t)(k - r * rowlen);
}

// RGB patches (noc = 3) that lie inside the image on the left, right and top also have a COMPACT weight array (round 5):
//   pixw    [gy][patch row r][gx][P]          one float per patch PIXEL: max(2,|r_0|) + max(2,|r_1|) + max(2,|r_2|), the
// denominator of the pixel's densification weight (patchgrid.cpp:256-259) -- all the densification ever uses of the three
// channel errors of such a patch; a third of the bytes.  Patches that overlap the left / right / top border keep the full
// pweight vector: the reference's running pointer shifts their entries (patchgrid.cpp:242,256-257; ofdis_densify.h).
__host__ __device__ __forceinline__ size_t pixw_row(const LevelGeom& g, int gx, int gy, int r) {
  return ((size_t)(gy * g.P + r) * g.nopw + gx) * (size_t)g.P;
}
// does the reference's running pweight pointer reach every pixel of patch (gx, gy) unshifted?  (no patch pixel outside the
// image to the left, to the right or above; rows below the image come after everything that is read)
__host__ __device__ __forceinline__ bool patch_weights_unshifted(const LevelGeom& g, int gx, int gy) {
  const int lb = -g.P / 2, ub = g.P / 2 - 1;
  const int rxi = gx * g.steps + g.offw, ryi = gy * g.steps + g.offh;
  return (rxi + lb >= 0) & (rxi + ub <= g.w - 1) & (ryi + lb >= 0);
}

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

// "sdiag" record layout of the fused TV path (ofdis_prep.hip writes it, ofdis_fused.hip walks it): S consecutive frames form
// a STRIP, laid side by side as one image of S*w columns; pixel (x, y) of the strip's frame fs lives in diag row
// d = (fs*w + x + y) mod (S*w), slot y -- every wrapped anti-diagonal of the strip is one contiguous row of h records, and a
// wavefront whose lane j handles column t - j at step t reads / writes exactly one such row per step.  S = 1 is the plain
// per-frame diag layout above.  The index is in records; a record is 8 floats (derivatives), 3 (wx, wy, mask) or 2 (du, dv).
__host__ __device__ __forceinline__ size_t sdiag_index(int frame, int x, int y, int w, int h, int S) {
  const int sg = frame / S, fs = frame - sg * S;
  const int rw = S * w;
  int d = fs * w + x + y;
  if (d >= rw) d -= rw;
  if (d >= rw) d %= rw;  // h > S*w only
  return ((size_t)sg * rw + d) * h + y;
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
