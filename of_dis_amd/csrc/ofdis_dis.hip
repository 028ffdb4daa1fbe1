// ofdis_dis.hip -- Dense Inverse Search kernels for gfx950 (CDNA4).
//
//   patch_optimize_*      : the whole inverse-compositional Gauss-Newton loop of a patch in registers, several
//                           patches per wavefront
//                           (patchgrid.cpp:98-141,195-211; patch.cpp:57-402).
//   densify_kernel        : AggregateFlowDense (patchgrid.cpp:213-275,377-397) as a gather that
//                           visits a pixel's covering patches in the reference's scatter order.
//
// Mapping.  The reference's patch vector has novals = noc*P*P entries, ordered (row, col, channel)
// (patch.cpp:306-324).  A patch is owned by LPP lanes of a wavefront:
//   * gray P = 8 (operating points 1, 2; patch_optimize_gray8_kernel): 4 lanes per patch, 16 patches per wavefront,
//     a lane holds two adjacent patch columns (16 entries) -- see the comment at that kernel;
//   * RGB P = 12 (BASELINE configs[3]), gray P = 12 (operating points 3, 4) and RGB P = 8 (run_OF_RGB's default):
//     16 lanes per patch, four patches per wavefront, a lane holds a 3x3 / 2x2 pixel block
//     (patch_optimize_rgb12_kernel under the fused contract, patch_optimize_rgb12x_kernel under the exact one);
//   * other patches of at most 64 entries: 8 lanes per patch (generic kernel, M = 1);
//   * everything else (other patch sizes, cost function 2, ofdis_tuning::rgb12 = 0): one patch per wavefront, lane l
//     owns entries l, l+64, ... (generic kernel, M per lane) -- the mapping whose summation order the exact contract documents.
// The stereo-depth mode's 1-D search is a template parameter of the first two.
// The template T, its gradients Tx, Ty, the residual and the weights never leave VGPRs (the exact contract's 16-lane
// kernel hands the interpolated values from the block layout to the entry-chain layout through LDS); the reductions of an
// iteration (mean, Tx.r, Ty.r, |r|) are in-lane sums plus DPP / permlane steps in the documented order
// (ofdis_dev.h, DESIGN.md "Reduction order").  The bilinear taps are loads from the padded level
// image, which is 41 KB at op-point 2 and therefore L1/L2 resident; blocks are mapped so that all patches of a
// frame run on one XCD (its L2 then holds that frame's four planes once).
#include <math.h>
#include <stdlib.h>

#include "ofdis_kernels.h"
#include "ofdis_densify.h"

namespace ofdis {
namespace OFDIS_KNS {  // the arithmetic contract this file is being compiled for (ofdis_dev.h)

// Sum over one patch vector in the documented reduction order (DESIGN.md "reduction order"; mirrored by
// oracle/eigen_shim -DOFDIS_SHIM_WAVE64 and oracle_set_reduce_order(1)).
//
//  novals > 64 (LPP = 64, one patch per wavefront): lane l accumulates entries l, l+64, ... sequentially,
//     then the 64-lane butterfly of ofdis_dev.h (distances 1,2,4,8,16,32).
//  novals <= 64 (LPP = 8, eight patches per wavefront): lane pl of the patch's 8 lanes owns entries
//     pl, pl+8, ..., pl+56 (for an 8x8 gray patch: one patch column) and accumulates them sequentially IN
//     THE LANE -- 7 plain adds, no cross-lane traffic -- then the 8 partials are combined at lane distance
//     4, 1, 2.  This is the shape of Eigen's own SSE reduction (two 4-wide accumulators = 8 stride-8 partial
//     sums, their sum, two horizontal adds) and costs 12 instructions instead of 31 for the butterfly over
//     64 virtual lanes.
template <int M, int LPP>
__device__ __forceinline__ float patch_sum(const float (&x)[M * (64 / LPP)], const bool (&valid)[M * (64 / LPP)]) {
  if constexpr (LPP == 64) {
    float c = valid[0] ? x[0] : 0.0f;  // starts FROM the first element (never "0 + x")
#pragma unroll
    for (int m = 1; m < M; ++m)
      if (valid[m]) c = c + x[m];
    return wave_sum(c);
  } else if constexpr (LPP == 32) {
    // novals > 64 with TWO patches per wavefront: lane pl of a patch's 32 lanes stands for the lanes pl and pl + 32 of the
    // one-patch mapping and keeps their two accumulation chains apart (x[2m] = entry pl + 64 m, x[2m+1] = entry
    // pl + 32 + 64 m); the butterfly steps at distance 1 ... 16 never leave a 32-lane half, so each chain is reduced over
    // the patch's 32 lanes exactly as its half was, and the last step (distance 32: lower half + upper half, in that
    // order in every lane) is one in-lane add.  Same additions, same order: same bits.
    float ca = valid[0] ? x[0] : 0.0f, cb = valid[1] ? x[1] : 0.0f;
#pragma unroll
    for (int m = 1; m < M; ++m) {
      if (valid[2 * m]) ca = ca + x[2 * m];
      if (valid[2 * m + 1]) cb = cb + x[2 * m + 1];
    }
    return half_wave_sum(ca) + half_wave_sum(cb);
  } else {
    static_assert(LPP == 8 && M == 1, "novals <= 64 uses 8 lanes per patch");
    float c = valid[0] ? x[0] : 0.0f;
#pragma unroll
    for (int q = 1; q < 8; ++q)
      if (valid[q]) c = c + x[q];
    // distance 4 inside each group of 8 lanes: lanes 0-3 take lane+4 (row_shl:4 written to DPP banks 0 and 2),
    // lanes 4-7 take lane-4 (row_shr:4 written to banks 1 and 3)
    const int ci = __builtin_bit_cast(int, c);
    int o = __builtin_amdgcn_update_dpp(0, ci, 0x104, 0xf, 0x5, false);
    o = __builtin_amdgcn_update_dpp(o, ci, 0x114, 0xf, 0xA, false);
    c = c + __builtin_bit_cast(float, o);
    c = c + dpp_mov<0xB1>(c);  // distance 1
    c = c + dpp_mov<0x4E>(c);  // distance 2
    return c;
  }
}

// NV: novals as a compile-time constant (0 = read it from the arguments): entry slots that are real patch entries in every
//     lane then lose their validity selects (all of them when NV == 64*M), and the tuned instantiations (NV > 0) use the
//     trimmed correctly-rounded divide / square root of ofdis_dev.h (same bits inside the operand ranges DESIGN.md
//     "Divide and square root" states for the patch kernels: quotients whose numerators are sums of products of
//     image-derived values, square roots of differences of such values);
// COST: the cost function as a compile-time constant, or -1 to read it from the arguments.
template <int M, int LPP, int NV, int COST>
__global__ __launch_bounds__(256) void patch_optimize_kernel(const DisArgs a) {
  constexpr int Q = 64 / LPP;  // patches per wavefront == accumulation chains per lane
  constexpr int E = M * Q;     // patch entries per lane
  constexpr bool FULL = NV == 64 * M;
  constexpr bool TRIM = NV > 0;
  const int costfct = COST >= 0 ? COST : a.costfct;
  const LevelGeom& g = a.g;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int blocks_per_frame = (g.nop + 4 * Q - 1) / (4 * Q);
  int frame, blk;
  xcd_frame_map(blockIdx.x, blocks_per_frame, a.nframes, frame, blk);
  if (frame >= a.nframes) return;  // block-uniform
  const int sub = lane / LPP;
  const int pl = lane % LPP;
  int ip = (blk * 4 + wave) * Q + sub;
  const bool live = ip < g.nop;  // uniform per patch
  if (Q == 1 && !live) return;
  if (!live) ip = g.nop - 1;     // idle lane group: shadows the last patch, never stores

  const int noc = g.noc, P = g.P, tw = g.tmp_w, nv = NV > 0 ? NV : g.novals;
  const int lb = -P / 2;
  const size_t plane = g.plane_elems;
  const float* __restrict__ imA = a.im_a + (size_t)frame * plane;
  const float* __restrict__ imAx = a.im_a_dx + (size_t)frame * plane;
  const float* __restrict__ imAy = a.im_a_dy + (size_t)frame * plane;
  const float* __restrict__ imB = a.im_b + (size_t)frame * plane;

  // grid position (patchgrid.cpp:62-69): ip = gx*noph + gy
  const int gx = ip / g.noph, gy = ip - gx * g.noph;
  const float rx = (float)(gx * g.steps + g.offw), ry = (float)(gy * g.steps + g.offh);

  // per-lane entry offsets relative to the patch centre in the padded, interleaved plane
  int off[E], kidx[E];
  bool valid[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int k = (e / Q) * 64 + (e % Q) * LPP + pl;
    kidx[e] = k;
    valid[e] = (NV > 0 && (e / Q) * 64 + 64 <= NV) ? true : (k < nv);  // compile-time true for whole 64-entry groups
    const int kk = valid[e] ? k : 0;
    const int c = kk % noc, q = kk / noc;
    const int col = q % P, row = q / P;
    off[e] = ((row + lb) * tw + (col + lb)) * noc + c;
  }
  int offb[E];  // ... in bytes, for the buffer loads of the second image
#pragma unroll
  for (int e = 0; e < E; ++e) offb[e] = off[e] * 4;
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc((void*)imB, 0, (int)(plane * sizeof(float)), 0x00020000);
  const int nocb = noc * 4, upb = tw * noc * 4;
  auto ldB = [&](int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, voff, soff, 0));
  };
  // x / novals: when novals is a power of two the product with its reciprocal is the same correctly
  // rounded value as the reference's division (both round the same real number), at 1/10 the cost
  const float fnv = (float)nv;
  const bool nv_pow2 = FULL ? ((M & (M - 1)) == 0) : ((nv & (nv - 1)) == 0);
  const float inv_nv = 1.0f / fnv;
  const float rcp_nv = rcp_refined(fnv);
  auto div_nv = [&](float x) { return nv_pow2 ? x * inv_nv : (TRIM ? div_by(x, fnv, rcp_nv) : x / fnv); };

  // ---- InitializePatch: template + gradients at the integer reference position (patch.cpp:287-332)
  float T[E], Tx[E], Ty[E];
  {
    const int px = (int)roundf(rx) + g.pad, py = (int)roundf(ry) + g.pad;
    const int base = (py * tw + px) * noc;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const unsigned o = (unsigned)(base + off[e]);
      T[e] = valid[e] ? imA[o] : 0.0f;
      Tx[e] = valid[e] ? imAx[o] : 0.0f;
      Ty[e] = valid[e] ? imAy[o] : 0.0f;
    }
    if (a.patnorm > 0) {
      const float mean = div_nv(patch_sum<M, LPP>(T, valid));
#pragma unroll
      for (int e = 0; e < E; ++e) T[e] -= mean;
    }
  }
  // ---- ComputeHessian (patch.cpp:71-88) and its Cholesky factor (Eigen LLT, call site patch.cpp:184;
  //      semantics as written out in oracle/eigen_shim/Eigen/Core).  H is constant per patch, so the
  //      factor is computed once instead of once per iteration.
  float l00, l10, l11;
  {
    float pxx[E], pxy[E], pyy[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      pxx[e] = Tx[e] * Tx[e];
      pxy[e] = Tx[e] * Ty[e];
      pyy[e] = Ty[e] * Ty[e];
    }
    float H00 = patch_sum<M, LPP>(pxx, valid);
    const float H01 = patch_sum<M, LPP>(pxy, valid);
    float H11 = patch_sum<M, LPP>(pyy, valid);
    if (a.stereo) {  // 1x1 Hessian of the horizontal displacement (patch.cpp:83-87)
      if (H00 == 0.0f) H00 = (float)((double)H00 + 1e-10);
      l00 = H00; l10 = 0.0f; l11 = 1.0f;
      if (!(l00 <= 0.0f)) l00 = sqrtf(l00);
    } else {
      if (H00 * H11 - H01 * H01 == 0.0f) {  // float += double literal in the reference
        H00 = (float)((double)H00 + 1e-10);
        H11 = (float)((double)H11 + 1e-10);
      }
      l00 = H00; l10 = H01; l11 = H11;
      if (!(l00 <= 0.0f)) {
        l00 = sqrtf(l00);
        l10 = l10 / l00;
        const float x = l11 - l10 * l10;
        if (!(x <= 0.0f)) l11 = sqrtf(x);
      }
    }
  }

  // ---- InitializeFromCoarserOF (patchgrid.cpp:195-211)
  float pin0 = 0.0f, pin1 = 0.0f;
  if (a.flow_prev) {
    const int x = (int)floorf(rx / 2), y = (int)floorf(ry / 2);
    const int i = y * (g.w / 2) + x;
    if (a.stereo) {  // one channel
      pin0 = (a.flow_prev + (size_t)frame * (size_t)(g.w / 2) * (g.h / 2))[i] * 2;
    } else {
      const float* fp = a.flow_prev + (size_t)frame * (size_t)(g.w / 2) * (g.h / 2) * 2;
      pin0 = fp[2 * i] * 2;
      pin1 = fp[2 * i + 1] * 2;
    }
  }

  // ---- OptimizeIter (patch.cpp:159-212).  All state below is uniform per patch; the lane groups of a
  //      wave diverge like ordinary SIMT branches (every cross-lane operation used here stays inside
  //      the LPP lanes of one patch).
  const float r00 = rcp_refined(l00), r11 = rcp_refined(l11);  // TRIM: shared by the four quotients of every solve
  float p0 = pin0, p1 = pin1;
  float ptx = rx + p0, pty = ry + p1;
  const float stx = ptx, sty = pty;
  float dp0 = 0.0f, dp1 = 0.0f;
  float dpsq = 1e-10f, dpsq_init = 1e-10f, mares = 1e20f, mares_old = 1e20f;
  int cnt = 0;
  bool converged = false;
  float pdiff[E], pw[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { pdiff[e] = 0.0f; pw[e] = 0.0f; }  // pw = 0: the reference's never-written pweight

  // OptimizeComputeErrImg (patch.cpp:264-284) = getPatchStaticBil (335-402) + LossComputeErrorImage (223-262)
  auto compute_err = [&]() {
    int pos0 = (int)ceilf(ptx + .00001f), pos1 = (int)ceilf(pty + .00001f);
    const int pos2 = (int)floorf(ptx), pos3 = (int)floorf(pty);
    const float r0 = ptx - (float)pos2, r1 = pty - (float)pos3;
    const float we0 = r0 * r1, we1 = (1 - r0) * r1, we2 = r0 * (1 - r1), we3 = (1 - r0) * (1 - r1);
    pos0 += g.pad;
    pos1 += g.pad;
    // the four taps of an entry through ONE buffer resource: a 32-bit per-lane byte offset (the upper-left tap d; the
    // position is >= one row + one pixel inside the plane for every in-bounds patch) and wave-uniform offsets to the
    // other three -- one integer add per entry instead of a 64-bit address per tap
    const int based = ((pos1 - 1) * tw + (pos0 - 1)) * noc * 4;
    float v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int vo = based + offb[e];
      const float td = ldB(vo, 0), tc = ldB(vo, nocb), tb = ldB(vo, upb), ta = ldB(vo, upb + nocb);
      v[e] = valid[e] ? (we0 * ta + we1 * tb + we2 * tc + we3 * td) : 0.0f;
    }
    if (a.patnorm > 0) {
      const float mean = div_nv(patch_sum<M, LPP>(v, valid));
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] -= mean;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float d = v[e] - T[e];
      if (costfct == 1) {
        d = copysignf(TRIM ? sqrt_rn(fabsf(d)) : sqrtf(fabsf(d)), d);
      } else if (costfct == 2) {
        const float bsq = 5.0f * 5.0f, bsq2 = bsq * 2.0f;
        d = copysignf(sqrtf((sqrtf(1.0f + (d * d) / bsq) - 1.0f) * bsq2), d);
      }
      pdiff[e] = valid[e] ? d : 0.0f;
      pw[e] = fabsf(pdiff[e]);
    }
    dpsq = dp0 * dp0 + dp1 * dp1;
    if (cnt == 1) dpsq_init = dpsq;
    mares_old = mares;
    mares = div_nv(patch_sum<M, LPP>(pw, valid));
    // patch.cpp:279-282.  The reference evaluates all terms with bitwise &,|; the two ratios have no
    // side effects, so they are only computed where they can decide (min_iter <= cnt < max_iter).
    bool go = (cnt < a.max_iter) && (mares > a.res_thresh);
    // div_rn == IEEE division except (by < 2 ulp) for numerators below 2^-102, where neither ratio can be within
    // 2 ulp of its threshold
    if (go && cnt >= a.min_iter)
      go = (div_rn(dpsq, dpsq_init) >= a.dp_thresh_sq) && (div_rn(mares, mares_old) <= a.dr_thresh);
    if (!go) converged = true;
  };
  auto oob = [&](float x, float y) { return (x < g.lb) | (y < g.lb) | (x > g.ubw) | (y > g.ubh); };

  // OptimizeStart (patch.cpp:120-156)
  if (oob(ptx, pty) || !(isfinite(ptx) && isfinite(pty))) {
    converged = true;
  } else {
    cnt = 0; dpsq = 1e-10f; dpsq_init = 1e-10f; mares = 1e5f; mares_old = 1e20f;
    compute_err();
  }
  while (!converged) {
    cnt++;
    float gxr[E], gyr[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      gxr[e] = Tx[e] * pdiff[e];
      gyr[e] = Ty[e] * pdiff[e];
    }
    const float b0 = patch_sum<M, LPP>(gxr, valid);
    const float b1 = patch_sum<M, LPP>(gyr, valid);
    // delta_p = LLT(H).solve(b)
    if (a.stereo) {  // patch.cpp:180-193: 1x1 system, then the disparity sign constraint of the camera side
      dp0 = (b0 / l00) / l00;
      dp1 = 0.0f;
      p0 -= dp0;
      p0 = a.camlr == 0 ? ((0.0f < p0) ? 0.0f : p0) : ((p0 < 0.0f) ? 0.0f : p0);  // std::min / std::max (p, 0)
    } else if (TRIM) {  // shared refined reciprocals, as in the gray 8x8 kernel
      const float y0 = div_by(b0, l00, r00);
      const float y1 = div_by(b1 - l10 * y0, l11, r11);
      dp1 = div_by(y1, l11, r11);
      dp0 = div_by(y0 - l10 * dp1, l00, r00);
      p0 -= dp0;
      p1 -= dp1;
    } else {
      const float y0 = b0 / l00;
      const float y1 = (b1 - l10 * y0) / l11;
      dp1 = y1 / l11;
      dp0 = (y0 - l10 * dp1) / l00;
      p0 -= dp0;
      p1 -= dp1;
    }
    ptx = rx + p0;
    pty = ry + p1;
    const float ex = stx - ptx, ey = sty - pty;
    // a NaN position passes every comparison of the reference and then indexes out of bounds
    // (SURVEY.md 7-4b); here it is treated as an outlier.
    // branch-free (selects): the three tests are cheap next to the divergence bookkeeping of a nested branch.
    // norm > outlierthresh (patch.cpp:199) without the square root: sqrt is monotonic and correctly rounded, so
    // sqrtf(x) > t  <=>  x > X, X = the largest float whose square root rounds to <= t (found on the host).
    const bool reset = (ex * ex + ey * ey > a.outlier_sq_max) | oob(ptx, pty) | !(isfinite(ptx) & isfinite(pty));
    p0 = reset ? pin0 : p0;
    p1 = reset ? pin1 : p1;
    ptx = rx + p0;
    pty = ry + p1;
    converged = converged | reset;
    compute_err();
  }

  if (live) {  // results in the internal grid-row-major layout (ofdis_dev.h: patch_slot, pweight_entry)
    float* pout = a.p_out + ((size_t)frame * g.nop + patch_slot(g, gx, gy)) * 2;
    if (pl == 0) {
      pout[0] = p0;
      pout[1] = p1;
    }
    float* pwout = a.pweight + (size_t)frame * g.nop * nv;
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (valid[e]) pwout[pweight_entry(g, gx, gy, kidx[e])] = pw[e];
  }
}

// ------------------------------------------------------------------------------------ gray 8x8 fast path
// Specialisation of patch_optimize_kernel<1, 8, true, COST> for the gray operating points 1 and 2
// (noc = 1, P = 8, novals = 64): same arithmetic, same reduction order, bit-identical results.
//
// Four lanes per patch, sixteen patches per wavefront.  Lane pl owns the adjacent patch COLUMNS 2pl and 2pl+1
// (entries r*8 + 2pl and r*8 + 2pl + 1, r = row), held as pairs.  The documented reduction order for <= 64
// entries (eight column sums p_0..p_7; q_i = p_i + p_(i+4); (q0 + q1) + (q2 + q3)) becomes two in-lane chains of
// 7 adds, the pair exchanged with the lane two over (q_2pl', q_2pl'+1), one in-lane add and one exchange with
// the neighbouring lane -- the same additions, operands commuted.  The per-patch scalar work (2x2 solve,
// position update, termination tests: ~45 % of an iteration's instructions) is paid once per 16 patches.
//
// The four bilinear taps of row r are
//     a = I[row r][col]   b = I[row r][col-1]   c = I[row r-1][col]   d = I[row r-1][col-1]
// so c,d of a row are a,b of the row above, and a lane's two columns need the three image columns 2pl-1, 2pl,
// 2pl+1: ONE 12-byte buffer load per image row, 9 per iteration.  The four lanes of a patch then touch one
// 36-byte span per instruction, i.e. one cache line (two when it straddles): the kernel is bound by L1 tag
// lookups per distinct line as much as by VALU issue, and this is the minimum (9 rows) per patch evaluation.
// The row stride is a wave-uniform SGPR offset, the per-lane byte offset is computed once per iteration.
// A pair is two plain floats: gfx950's SIMDs are 32 lanes wide, a packed v_pk_*_f32 occupies the issue port twice as
// long as a scalar op, so packing buys nothing and costs the moves that build register pairs.
struct f2 {
  float x, y;
};
__device__ __forceinline__ f2 operator+(f2 a, f2 b) { return f2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f2 operator-(f2 a, f2 b) { return f2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ f2 operator*(f2 a, f2 b) { return f2{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ f2 operator-(f2 a, float b) { return f2{a.x - b, a.y - b}; }
__device__ __forceinline__ f2 operator*(float a, f2 b) { return f2{a * b.x, a * b.y}; }

__device__ __forceinline__ float gray8_combine(float c0, float c1) {  // column sums p_2pl, p_2pl+1 of this lane
  // (the same exchanges through the LDS crossbar, ds_swizzle_b32, measured the same kernel time: profiles/README.md r03_a)
  c0 = c0 + dpp_mov<0x4E>(c0);  // lane distance 2 = column distance 4: q_(2pl mod 4)
  c1 = c1 + dpp_mov<0x4E>(c1);  //                                       q_(2pl mod 4 + 1)
  float q = c0 + c1;            // lanes 0,2: q0 + q1; lanes 1,3: q2 + q3
  q = q + dpp_mov<0xB1>(q);     // lane distance 1
  return q;
}
__device__ __forceinline__ float gray8_sum(const f2 (&x)[8]) {
  f2 c = x[0];
#pragma unroll
  for (int r = 1; r < 8; ++r) c = c + x[r];
  return gray8_combine(c.x, c.y);
}
// same order for sum |x|
__device__ __forceinline__ float gray8_abs_sum(const f2 (&x)[8]) {
  float cx = fabsf(x[0].x), cy = fabsf(x[0].y);
#pragma unroll
  for (int r = 1; r < 8; ++r) {
    cx = cx + fabsf(x[r].x);
    cy = cy + fabsf(x[r].y);
  }
  return gray8_combine(cx, cy);
}

template <int COST, bool STEREO>
__global__ __launch_bounds__(256) void patch_optimize_gray8_kernel(const DisArgs a) {
  constexpr int LPP = 4, Q = 16, R = 8;
  const int costfct = COST >= 0 ? COST : a.costfct;
  const LevelGeom& g = a.g;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;  // wavefronts per block
  const int blocks_per_frame = (g.nop + wpb * Q - 1) / (wpb * Q);
  int frame, blk;
  xcd_frame_map(blockIdx.x, blocks_per_frame, a.nframes, frame, blk);
  if (frame >= a.nframes) return;                // block-uniform
  if ((blk * wpb + wave) * Q >= g.nop) return;   // wave-uniform: no patch for this wavefront
  const int sub = lane / LPP;
  const int pl = lane % LPP;
  // Which 16 patches a wavefront takes is a free choice (a patch's results go to its own index ip = gx*noph + gy): it takes
  // 16 patches that are NEIGHBOURS ALONG A GRID ROW (row-major counter jp), not 16 consecutive indices (= a grid column).
  // Patches of one grid row read the same image rows, `steps` pixels apart, so one load instruction of the wavefront -- 16
  // patches x one 36-byte span -- touches 2-4 cache lines instead of 16 different ones: the kernel's memory time is the
  // texture-address path's time per distinct line, not bytes (profiles/README.md round 4).
  int jp = (blk * wpb + wave) * Q + sub;
  const bool live = jp < g.nop;
  if (!live) jp = g.nop - 1;  // idle lane group: shadows the last patch, never stores
  const int gy = jp / g.nopw, gx = jp - gy * g.nopw;  // (results are stored by (gx, gy): ofdis_dev.h patch_slot / pweight_row)

  const int tw = g.tmp_w;
  const size_t plane = g.plane_elems;
  auto plane_rsrc = [&](const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)frame * plane), 0, (int)(plane * sizeof(float)), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsB = plane_rsrc(a.im_b);
  const int row_bytes = tw * 4;

  const float rx = (float)(gx * g.steps + g.offw), ry = (float)(gy * g.steps + g.offh);
  const float inv_nv = 1.0f / 64.0f;  // x/64 == x*(1/64): both round the same real number

  // ---- InitializePatch (patch.cpp:287-332): pair r = row r of this lane's two columns
  f2 T[R], Tx[R], Ty[R];
  float meanT = 0.0f;  // fused contract: mean of the stored, mean-normalised template (compute_err)
  {
    const int px = (int)roundf(rx) + g.pad, py = (int)roundf(ry) + g.pad;
    // one 8-byte buffer load per plane and patch row: per-lane byte offset of the lane's first column in the patch's first
    // row, the row as a scalar offset
    const __amdgpu_buffer_rsrc_t rsA = plane_rsrc(a.im_a), rsAx = plane_rsrc(a.im_a_dx), rsAy = plane_rsrc(a.im_a_dy);
    const int vbase = ((py - 4) * tw + px - 4 + 2 * pl) * 4;
    auto ld2 = [&](const __amdgpu_buffer_rsrc_t& rs, int r) {
      const auto t = __builtin_amdgcn_raw_buffer_load_b64(rs, vbase, r * row_bytes, 0);
      const unsigned u0 = t[0], u1 = t[1];  // (through scalars, see compute_err)
      return f2{__builtin_bit_cast(float, u0), __builtin_bit_cast(float, u1)};
    };
#pragma unroll
    for (int r = 0; r < R; ++r) {
      T[r] = ld2(rsA, r);
      Tx[r] = ld2(rsAx, r);
      Ty[r] = ld2(rsAy, r);
    }
    if (a.patnorm > 0) {
      const float mean = gray8_sum(T) * inv_nv;
#pragma unroll
      for (int r = 0; r < R; ++r) T[r] = T[r] - mean;
      if constexpr (kFusedContract) meanT = gray8_sum(T) * inv_nv;  // (what rounding left of the template's mean)
    }
  }
  // ---- ComputeHessian + Cholesky factor (patch.cpp:71-88, Eigen LLT as in oracle/eigen_shim)
  float l00, l10, l11;
  {
    f2 pxx[R], pxy[R], pyy[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      pxx[r] = Tx[r] * Tx[r];
      pxy[r] = Tx[r] * Ty[r];
      pyy[r] = Ty[r] * Ty[r];
    }
    float H00 = gray8_sum(pxx);
    if constexpr (STEREO) {  // 1x1 Hessian of the horizontal displacement (patch.cpp:83-87)
      if (H00 == 0.0f) H00 = (float)((double)H00 + 1e-10);
      l00 = H00; l10 = 0.0f; l11 = 1.0f;
      if (!(l00 <= 0.0f)) l00 = sqrtf(l00);
    } else {
      const float H01 = gray8_sum(pxy);
      float H11 = gray8_sum(pyy);
      if (H00 * H11 - H01 * H01 == 0.0f) {
        H00 = (float)((double)H00 + 1e-10);
        H11 = (float)((double)H11 + 1e-10);
      }
      l00 = H00; l10 = H01; l11 = H11;
      if (!(l00 <= 0.0f)) {
        l00 = sqrtf(l00);
        l10 = l10 / l00;
        const float x = l11 - l10 * l10;
        if (!(x <= 0.0f)) l11 = sqrtf(x);
      }
    }
  }
  // ---- InitializeFromCoarserOF (patchgrid.cpp:195-211)
  float pin0 = 0.0f, pin1 = 0.0f;
  if (a.flow_prev) {
    const int x = (int)floorf(rx / 2), y = (int)floorf(ry / 2);
    const int i = y * (g.w / 2) + x;
    if constexpr (STEREO) {  // one channel
      pin0 = (a.flow_prev + (size_t)frame * (size_t)(g.w / 2) * (g.h / 2))[i] * 2;
    } else {
      const float* fp = a.flow_prev + (size_t)frame * (size_t)(g.w / 2) * (g.h / 2) * 2;
      pin0 = fp[2 * i] * 2;
      pin1 = fp[2 * i + 1] * 2;
    }
  }
  // ---- OptimizeIter (patch.cpp:159-212)
  // Register budget: the residual vector lives only inside one evaluation.  What the next Gauss-Newton step
  // needs of it -- the two gradient sums -- is reduced while it is fresh, and the weights |r| go to memory at the
  // evaluation after which the patch stops (every patch has exactly one), so only T, Tx, Ty stay resident.
  const float r00 = rcp_refined(l00), r11 = rcp_refined(l11);  // shared by the four quotients of every solve
  float p0 = pin0, p1 = pin1;
  float ptx = rx + p0, pty = ry + p1;
  const float stx = ptx, sty = pty;
  float dp0 = 0.0f, dp1 = 0.0f;
  float dpsq = 1e-10f, dpsq_init = 1e-10f, mares = 1e20f, mares_old = 1e20f;
  float b0 = 0.0f, b1 = 0.0f;  // sum Tx.r, sum Ty.r of the latest evaluation (patch.cpp:178-181)
  int cnt = 0;
  bool converged = false;
  // row r of this patch's weights: 8 floats inside the run of its grid row's patches (ofdis_dev.h: pweight_row); rows are
  // nopw * 8 floats apart.  The wavefront's 16 patches are neighbours along the grid row: one store instruction = 512 B
  float2* const pwout = reinterpret_cast<float2*>(a.pweight + (size_t)frame * g.nop * 64 + pweight_row(g, gx, gy, 0)) + pl;
  const int pwstride = g.nopw * 4;  // float2 per patch row

  auto cost = [&](float d) {
    if (costfct == 1) return copysignf(sqrtf(fabsf(d)), d);
    const float bsq = 5.0f * 5.0f, bsq2 = bsq * 2.0f;
    return copysignf(sqrtf((sqrtf(1.0f + (d * d) / bsq) - 1.0f) * bsq2), d);
  };
  auto compute_err = [&](bool stop) {  // patch.cpp:264-284, 335-402, 223-262
    int pos0 = (int)ceilf(ptx + .00001f), pos1 = (int)ceilf(pty + .00001f);
    const int pos2 = (int)floorf(ptx), pos3 = (int)floorf(pty);
    const float r0 = ptx - (float)pos2, r1 = pty - (float)pos3;
    const float we0 = r0 * r1, we1 = (1 - r0) * r1, we2 = r0 * (1 - r1), we3 = (1 - r0) * (1 - r1);
    pos0 += g.pad;
    pos1 += g.pad;
    // byte offset of (row pos1-5, column pos0-5+2pl) = the left tap of the lane's first column; rows rr = 0..8
    // follow at rr*row_bytes
    const int voff = ((pos1 - 5) * tw + pos0 - 5 + 2 * pl) * 4;
    f2 A[9], Bn[9];
#pragma unroll
    for (int rr = 0; rr < 9; ++rr) {
      const auto t = __builtin_amdgcn_raw_buffer_load_b96(rsB, voff, rr * row_bytes, 0);
      // (through a scalar: __builtin_bit_cast applied directly to a vector element reads element 0, ROCm 7.2)
      const unsigned u0 = t[0], u1 = t[1], u2 = t[2];
      const float c0 = __builtin_bit_cast(float, u0), c1 = __builtin_bit_cast(float, u1),
                  c2 = __builtin_bit_cast(float, u2);
      A[rr] = f2{c1, c2};
      Bn[rr] = f2{c0, c1};
    }
    f2 v[R];
    if constexpr (kFusedContract) {
      // fused contract: the template is subtracted inside the interpolation's multiply-add chain (the first product becomes
      // an fma with -T) and the patch mean is taken of that difference: mean(v) = mean(v - T) + mean(T), mean(T) being the
      // rounding residue of the normalised template (meanT) -- 16 subtractions and 16 multiplies fewer per evaluation
#pragma unroll
      for (int r = 0; r < R; ++r) v[r] = we3 * Bn[r] + (we2 * A[r] + (we1 * Bn[r + 1] + (we0 * A[r + 1] - T[r])));
      if (a.patnorm > 0) {
        const float mean = gray8_sum(v) * inv_nv + meanT;
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = v[r] - mean;
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) v[r] = we0 * A[r + 1] + we1 * Bn[r + 1] + we2 * A[r] + we3 * Bn[r];
      if (a.patnorm > 0) {
        const float mean = gray8_sum(v) * inv_nv;
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = v[r] - mean;
      }
    }
    f2 gxr[R], gyr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      f2 d = kFusedContract ? v[r] : v[r] - T[r];
      if (costfct != 0) d = f2{cost(d.x), cost(d.y)};
      v[r] = d;  // the residual (patch.cpp:230-261); the weights are |residual|
      gxr[r] = Tx[r] * d;
      gyr[r] = Ty[r] * d;
    }
    b0 = gray8_sum(gxr);
    b1 = gray8_sum(gyr);
    dpsq = dp0 * dp0 + dp1 * dp1;
    if (cnt == 1) dpsq_init = dpsq;
    mares_old = mares;
    mares = gray8_abs_sum(v) * inv_nv;
    bool go = (cnt < a.max_iter) && (mares > a.res_thresh);
    // div_rn == IEEE division except (by < 2 ulp) for numerators below 2^-102, where neither ratio can be within
    // 2 ulp of its threshold
    if (go && cnt >= a.min_iter)
      go = (div_rn(dpsq, dpsq_init) >= a.dp_thresh_sq) && (div_rn(mares, mares_old) <= a.dr_thresh);
    if (!go | stop) {
      converged = true;
      if (live) {
#pragma unroll
        for (int r = 0; r < R; ++r) pwout[r * pwstride] = make_float2(fabsf(v[r].x), fabsf(v[r].y));
      }
    }
  };
  auto oob = [&](float x, float y) { return (x < g.lb) | (y < g.lb) | (x > g.ubw) | (y > g.ubh); };

  if (oob(ptx, pty) || !(isfinite(ptx) && isfinite(pty))) {
    converged = true;  // OptimizeStart (patch.cpp:120-156): no evaluation, pweight keeps its initial zeros
    if (live) {
#pragma unroll
      for (int r = 0; r < R; ++r) pwout[r * pwstride] = make_float2(0.0f, 0.0f);
    }
  } else {
    dpsq = 1e-10f; dpsq_init = 1e-10f; mares = 1e5f; mares_old = 1e20f;
    compute_err(false);
  }
  while (!converged) {
    cnt++;
    // delta_p = LLT(H).solve(b) (patch.cpp:184).  div_by(a, b, rcp_refined(b)) == a / b unless 0 < |a| < 2^-102
    // (ofdis_dev.h), which a sum of products of image values cannot be (DESIGN.md "Arithmetic contract")
    if constexpr (STEREO) {  // patch.cpp:180-193: 1x1 system, then the disparity sign constraint of the camera side
      dp0 = (b0 / l00) / l00;
      dp1 = 0.0f;
      p0 -= dp0;
      p0 = a.camlr == 0 ? ((0.0f < p0) ? 0.0f : p0) : ((p0 < 0.0f) ? 0.0f : p0);  // std::min / std::max (p, 0)
    } else {
      const float y0 = div_by(b0, l00, r00);
      const float y1 = div_by(b1 - l10 * y0, l11, r11);
      dp1 = div_by(y1, l11, r11);
      dp0 = div_by(y0 - l10 * dp1, l00, r00);
      p0 -= dp0;
      p1 -= dp1;
    }
    ptx = rx + p0;
    pty = ry + p1;
    const float ex = stx - ptx, ey = sty - pty;
    // branch-free (selects): the three tests are cheap next to the divergence bookkeeping of a nested branch.
    // norm > outlierthresh (patch.cpp:199) without the square root: sqrt is monotonic and correctly rounded, so
    // sqrtf(x) > t  <=>  x > X, X = the largest float whose square root rounds to <= t (found on the host).
    const bool reset = (ex * ex + ey * ey > a.outlier_sq_max) | oob(ptx, pty) | !(isfinite(ptx) & isfinite(pty));
    p0 = reset ? pin0 : p0;
    p1 = reset ? pin1 : p1;
    ptx = rx + p0;
    pty = ry + p1;
    compute_err(reset);
  }

  if (live && pl == 0) {
    float* pout = a.p_out + ((size_t)frame * g.nop + patch_slot(g, gx, gy)) * 2;
    pout[0] = p0;
    pout[1] = p1;
  }
}

// ------------------------------------------------------------------------------------ RGB 12x12, fused contract
// Operating points 3 / 4 and BASELINE configs[3] (noc = 3, P = 12: 432 entries per patch).  The generic kernel above gives a
// patch a whole wavefront, lane l owning entries l, l + 64, ...: four 4-byte tap loads per entry, four 64-lane butterflies per
// iteration and the scalar solve repeated in 64 lanes -- 353 instructions per patch and iteration, and its summation order
// (64 strided partials + butterfly) IS the documented order of the exact contract.  The fused contract leaves the order of a
// sum free (ofdis_dev.h), which opens the mapping of the gray 8x8 kernel to this geometry:
//   * a patch is 16 lanes (one DPP row), FOUR patches per wavefront, neighbours along a grid row (same image rows);
//   * lane (rg, cg) = (pl / 4, pl % 4) owns the 3 x 3 pixel block rows 3 rg .. 3 rg + 2, columns 3 cg .. 3 cg + 2: 27 entries,
//     9 contiguous floats per row; T, Tx, Ty of the block live in VGPRs;
//   * an evaluation needs the 4 x 4 pixel window around the block (one row above, one pixel to the left): per window row 12
//     contiguous floats = three 16-byte buffer loads, 12 loads per evaluation for 27 entries (the generic kernel: 108), the
//     row a scalar offset, one per-lane byte offset per evaluation;
//   * sums: 27 in-lane terms, then four DPP steps inside the 16-lane row; the 2x2 solve, position update and termination
//     tests are paid once per four patches.
// About 105 instead of 353 instructions per patch and iteration.  Same algorithm, same taps, same control flow: results
// within the fused contract's tolerance of the exact kernel (tests/test_gpu_contract.py).
// N contiguous floats at byte offset voff (per lane) + soff (scalar): 16-byte buffer loads, then the widest load for the rest
template <int N>
__device__ __forceinline__ void buffer_load_floats(const __amdgpu_buffer_rsrc_t& rs, int voff, int soff, float* dst) {
  auto asf = [](unsigned u) { return __builtin_bit_cast(float, u); };  // (vector elements go through scalars, see compute_err)
#pragma unroll
  for (int q = 0; q + 4 <= N; q += 4) {
    const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 4 * q, soff, 0);
    const unsigned u0 = t[0], u1 = t[1], u2 = t[2], u3 = t[3];
    dst[q] = asf(u0); dst[q + 1] = asf(u1); dst[q + 2] = asf(u2); dst[q + 3] = asf(u3);
  }
  constexpr int F = N / 4 * 4;
  if constexpr (N % 4 == 3) {
    const auto t = __builtin_amdgcn_raw_buffer_load_b96(rs, voff + 4 * F, soff, 0);
    const unsigned u0 = t[0], u1 = t[1], u2 = t[2];
    dst[F] = asf(u0); dst[F + 1] = asf(u1); dst[F + 2] = asf(u2);
  } else if constexpr (N % 4 == 2) {
    const auto t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + 4 * F, soff, 0);
    const unsigned u0 = t[0], u1 = t[1];
    dst[F] = asf(u0); dst[F + 1] = asf(u1);
  } else if constexpr (N % 4 == 1) {
    dst[F] = asf(__builtin_amdgcn_raw_buffer_load_b32(rs, voff + 4 * F, soff, 0));
  }
}

__device__ __forceinline__ float row16_sum(float x) {  // all-reduce inside a DPP row of 16 lanes
  x = x + dpp_mov<0xB1>(x);   // quad_perm [1,0,3,2]
  x = x + dpp_mov<0x4E>(x);   // quad_perm [2,3,0,1]
  x = x + dpp_mov<0x141>(x);  // row_half_mirror
  x = x + dpp_mov<0x140>(x);  // row_mirror
  return x;
}

// NOC = 1: the same mapping for gray 12 x 12 patches (operating points 3 and 4 of run_OF_INT / run_DE_INT): 9 entries per
// lane, one 16-byte load per window row.  BS = 2: RGB 8 x 8 patches (operating points 1 and 2 of run_OF_RGB / run_DE_RGB):
// a 2 x 2 pixel block per lane, 12 entries, a 3 x 3 pixel window.  STEREO: the 1-D search of the depth mode
// (patch.cpp:83-87, 180-193).
template <int COST, int NOC, bool STEREO, int BS>
__global__ __launch_bounds__(256, 3) void patch_optimize_rgb12_kernel(const DisArgs a) {
  constexpr int Q = 4, NE = BS * BS * NOC, NV = 16 * NE;  // patches per wavefront, entries per lane, entries per patch
  constexpr int RL = BS * NOC;                            // floats per row of the lane's BS x BS pixel block
  constexpr int P = 4 * BS, HP = P / 2;                   // patch size
  constexpr bool TYL = NOC == 3 && BS == 3;               // the template's y gradient lives in LDS
  constexpr bool PIXW = NOC == 3 && BS == 3;              // compact per-pixel weights (ofdis_dev.h: pixw_row)
  // The template's y gradient lives in LDS, [entry / 4][thread] as 16-byte groups (a lane reads its own seven groups once
  // per evaluation, conflict-free): 27 registers less = 168 without scratch = three wavefronts per SIMD instead of two
  typedef float f4l __attribute__((ext_vector_type(4)));
  __shared__ f4l tyl[TYL ? 7 * 256 : 1];  // (9 or 12 entries per lane: registers)
  const LevelGeom& g = a.g;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int blocks_per_frame = (g.nop + 4 * Q - 1) / (4 * Q);
  int frame, blk;
  xcd_frame_map(blockIdx.x, blocks_per_frame, a.nframes, frame, blk);
  if (frame >= a.nframes) return;               // block-uniform
  if ((blk * 4 + wave) * Q >= g.nop) return;    // wave-uniform: no patch for this wavefront
  const int sub = lane >> 4, pl = lane & 15;
  const int rg = pl >> 2, cg = pl & 3;
  int jp = (blk * 4 + wave) * Q + sub;          // row-major patch counter (see the gray 8x8 kernel)
  const bool live = jp < g.nop;
  if (!live) jp = g.nop - 1;                    // idle lane group: shadows the last patch, never stores
  const int gy = jp / g.nopw, gx = jp - gy * g.nopw;  // (results are stored by (gx, gy): ofdis_dev.h patch_slot / pweight_row)

  const int tw = g.tmp_w;
  const size_t plane = g.plane_elems;
  auto plane_rsrc = [&](const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)frame * plane), 0, (int)(plane * sizeof(float)), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsB = plane_rsrc(a.im_b);
  const int row_bytes = tw * 4 * NOC;
  const float rx = (float)(gx * g.steps + g.offw), ry = (float)(gy * g.steps + g.offh);
  const float inv_nv = 1.0f / (float)NV;

  // ---- InitializePatch (patch.cpp:287-332): the lane's 3 x 3 pixel block, entry e = (row rr, pixel xx, channel c) at
  //      e = rr * 9 + xx * 3 + c
  float T[NE], Tx[NE], Ty[NE];
  float meanT = 0.0f;  // mean of the stored (mean-normalised) template: a rounding residue, kept so that the residual's mean is exact
  {
    const int px = (int)roundf(rx) + g.pad, py = (int)roundf(ry) + g.pad;
    const __amdgpu_buffer_rsrc_t rsA = plane_rsrc(a.im_a), rsAx = plane_rsrc(a.im_a_dx), rsAy = plane_rsrc(a.im_a_dy);
    const int vbase = ((py - HP + BS * rg) * tw + px - HP + BS * cg) * (4 * NOC);
#pragma unroll
    for (int rr = 0; rr < BS; ++rr) {
      buffer_load_floats<RL>(rsA, vbase, rr * row_bytes, T + RL * rr);
      buffer_load_floats<RL>(rsAx, vbase, rr * row_bytes, Tx + RL * rr);
      buffer_load_floats<RL>(rsAy, vbase, rr * row_bytes, Ty + RL * rr);
    }
    if (a.patnorm > 0) {
      float c = T[0];
#pragma unroll
      for (int e = 1; e < NE; ++e) c += T[e];
      const float mean = row16_sum(c) * inv_nv;
      float c2 = 0.0f;
#pragma unroll
      for (int e = 0; e < NE; ++e) { T[e] -= mean; c2 += T[e]; }
      meanT = row16_sum(c2) * inv_nv;
    }
  }
  // ---- ComputeHessian + Cholesky factor (patch.cpp:71-88, Eigen LLT as in oracle/eigen_shim)
  float l00, l10, l11;
  {
    float hxx = 0.0f, hxy = 0.0f, hyy = 0.0f;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      hxx += Tx[e] * Tx[e];
      hxy += Tx[e] * Ty[e];
      hyy += Ty[e] * Ty[e];
    }
    float H00 = row16_sum(hxx);
    if constexpr (STEREO) {  // 1x1 Hessian of the horizontal displacement (patch.cpp:83-87)
      if (H00 == 0.0f) H00 = (float)((double)H00 + 1e-10);
      l00 = H00; l10 = 0.0f; l11 = 1.0f;
      if (!(l00 <= 0.0f)) l00 = sqrtf(l00);
    } else {
      const float H01 = row16_sum(hxy);
      float H11 = row16_sum(hyy);
      if (H00 * H11 - H01 * H01 == 0.0f) {
        H00 = (float)((double)H00 + 1e-10);
        H11 = (float)((double)H11 + 1e-10);
      }
      l00 = H00; l10 = H01; l11 = H11;
      if (!(l00 <= 0.0f)) {
        l00 = sqrtf(l00);
        l10 = l10 / l00;
        const float x = l11 - l10 * l10;
        if (!(x <= 0.0f)) l11 = sqrtf(x);
      }
    }
  }
  if constexpr (TYL) {
#pragma unroll
    for (int q = 0; q < 7; ++q)
      tyl[q * 256 + threadIdx.x] = f4l{Ty[4 * q], Ty[4 * q + 1], Ty[4 * q + 2], q < 6 ? Ty[4 * q + 3] : 0.0f};
  }
  // ---- InitializeFromCoarserOF (patchgrid.cpp:195-211)
  float pin0 = 0.0f, pin1 = 0.0f;
  if (a.flow_prev) {
    const int x = (int)floorf(rx / 2), y = (int)floorf(ry / 2);
    const int i = y * (g.w / 2) + x;
    if constexpr (STEREO) {  // one channel
      pin0 = (a.flow_prev + (size_t)frame * (size_t)(g.w / 2) * (g.h / 2))[i] * 2;
    } else {
      const float* fp = a.flow_prev + (size_t)frame * (size_t)(g.w / 2) * (g.h / 2) * 2;
      pin0 = fp[2 * i] * 2;
      pin1 = fp[2 * i + 1] * 2;
    }
  }
  // ---- OptimizeIter (patch.cpp:159-212); state uniform per patch (16 lanes)
  const float r00 = rcp_refined(l00), r11 = rcp_refined(l11);
  float p0 = pin0, p1 = pin1;
  float ptx = rx + p0, pty = ry + p1;
  const float stx = ptx, sty = pty;
  float dp0 = 0.0f, dp1 = 0.0f;
  float dpsq = 1e-10f, dpsq_init = 1e-10f, mares = 1e20f, mares_old = 1e20f;
  float b0 = 0.0f, b1 = 0.0f;
  int cnt = 0;
  bool converged = false;
  // this lane's 3 rows of 9 weights within the patch's 432 (entry (row, col, c) at (row * 12 + col) * 3 + c)
  // (internal layout, ofdis_dev.h: pweight_row -- patch rows are nopw * 36 floats apart)
  float* const pwout = a.pweight + (size_t)frame * g.nop * NV + pweight_row(g, gx, gy, BS * rg) + RL * cg;
  const int pwstride = g.nopw * P * NOC;
  // A patch whose weights the densification reads unshifted (ofdis_dev.h: patch_weights_unshifted -- all but the patches on
  // the left / right / top border) stores ONE float per pixel, the denominator max(2,|r_0|) + max(2,|r_1|) + max(2,|r_2|) of
  // the pixel's weight (patchgrid.cpp:256-259, the same three operations in the same order), instead of its 432 |r|: the lane
  // holds the three channels of each of its nine pixels.  108 instead of 324 bytes per lane.
  const bool compact = PIXW && a.pixw != nullptr && patch_weights_unshifted(g, gx, gy);
  float* const pxout = (PIXW && a.pixw) ? a.pixw + (size_t)frame * g.nop * 144 + pixw_row(g, gx, gy, 3 * rg) + 3 * cg : nullptr;
  const int pxstride = g.nopw * 12;
  auto store_pw = [&](const float (&v)[NE], bool zero) {
    if constexpr (PIXW) {
      if (compact) {
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
          for (int xx = 0; xx < 3; ++xx) {
            const int e = rr * 9 + xx * 3;
            float sden = fmaxf(2.0f, fabsf(v[e]));
            sden += fmaxf(2.0f, fabsf(v[e + 1]));
            sden += fmaxf(2.0f, fabsf(v[e + 2]));
            pxout[rr * pxstride + xx] = zero ? 6.0f : sden;  // (never evaluated: |r| = 0 three times)
          }
        return;
      }
    }
#pragma unroll
    for (int rr = 0; rr < BS; ++rr)
#pragma unroll
      for (int q = 0; q < RL; ++q) pwout[rr * pwstride + q] = zero ? 0.0f : fabsf(v[rr * RL + q]);
  };

  auto compute_err = [&](bool stop) {  // patch.cpp:264-284, 335-402, 223-262
    int pos0 = (int)ceilf(ptx + .00001f), pos1 = (int)ceilf(pty + .00001f);
    const int pos2 = (int)floorf(ptx), pos3 = (int)floorf(pty);
    const float r0 = ptx - (float)pos2, r1 = pty - (float)pos3;
    const float we0 = r0 * r1, we1 = (1 - r0) * r1, we2 = r0 * (1 - r1), we3 = (1 - r0) * (1 - r1);
    pos0 += g.pad;
    pos1 += g.pad;
    // window: rows pos1 - 7 + 3 rg + j, j = 0..3; pixels pos0 - 7 + 3 cg + i, i = 0..3 (12 floats per row).  Entry (rr, xx, c)
    // takes a = W[rr+1][xx+1], b = W[rr+1][xx], c = W[rr][xx+1], d = W[rr][xx] (patch.cpp:335-402)
    const int voff = ((pos1 - (HP + 1) + BS * rg) * tw + pos0 - (HP + 1) + BS * cg) * (4 * NOC);
    float W[BS + 1][(BS + 1) * NOC];
#pragma unroll
    for (int j = 0; j <= BS; ++j) buffer_load_floats<(BS + 1) * NOC>(rsB, voff, j * row_bytes, W[j]);
    float d[NE];
    float se = 0.0f;
#pragma unroll
    for (int rr = 0; rr < BS; ++rr)
#pragma unroll
      for (int q = 0; q < RL; ++q) {  // q = xx * NOC + c: the left neighbour pixel is NOC floats back
        const int e = rr * RL + q;
        float v = we0 * W[rr + 1][q + NOC] - T[e];
        v = we1 * W[rr + 1][q] + v;
        v = we2 * W[rr][q + NOC] + v;
        v = we3 * W[rr][q] + v;
        d[e] = v;  // interpolated value minus the (mean-normalised) template
        se += v;
      }
    if (a.patnorm > 0) {  // the patch's own mean (patch.cpp:401): mean(v) = mean(v - T) + mean(T)
      const float mean = row16_sum(se) * inv_nv + meanT;
#pragma unroll
      for (int e = 0; e < NE; ++e) d[e] -= mean;
    } else {              // T was not normalised: d = v - T already
    }
    float g0 = 0.0f, g1 = 0.0f, sa = 0.0f;
    if constexpr (TYL) {
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const f4l ty = tyl[q * 256 + threadIdx.x];
        const float tyq[4] = {ty.x, ty.y, ty.z, ty.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = 4 * q + i;
          if (e < NE) {
            float r = d[e];
            if (COST == 1) r = copysignf(__builtin_amdgcn_sqrtf(fabsf(r)), r);  // L1 (patch.cpp:238-246)
            d[e] = r;
            g0 += Tx[e] * r;
            g1 += tyq[i] * r;
            sa += fabsf(r);
          }
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        float r = d[e];
        if (COST == 1) r = copysignf(__builtin_amdgcn_sqrtf(fabsf(r)), r);
        d[e] = r;
        g0 += Tx[e] * r;
        g1 += Ty[e] * r;
        sa += fabsf(r);
      }
    }
    b0 = row16_sum(g0);
    b1 = row16_sum(g1);
    dpsq = dp0 * dp0 + dp1 * dp1;
    if (cnt == 1) dpsq_init = dpsq;
    mares_old = mares;
    mares = row16_sum(sa) * inv_nv;
    bool go = (cnt < a.max_iter) && (mares > a.res_thresh);
    if (go && cnt >= a.min_iter)
      go = (div_rn(dpsq, dpsq_init) >= a.dp_thresh_sq) && (div_rn(mares, mares_old) <= a.dr_thresh);
    if (!go | stop) {
      converged = true;
      if (live) store_pw(d, false);
    }
  };
  auto oob = [&](float x, float y) { return (x < g.lb) | (y < g.lb) | (x > g.ubw) | (y > g.ubh); };

  if (oob(ptx, pty) || !(isfinite(ptx) && isfinite(pty))) {
    converged = true;  // OptimizeStart (patch.cpp:120-156): no evaluation, pweight keeps its initial zeros
    if (live) store_pw(T, true);
  } else {
    dpsq = 1e-10f; dpsq_init = 1e-10f; mares = 1e5f; mares_old = 1e20f;
    compute_err(false);
  }
  while (!converged) {
    cnt++;
    if constexpr (STEREO) {  // patch.cpp:180-193: 1x1 system, then the disparity sign constraint of the camera side
      dp0 = (b0 / l00) / l00;
      dp1 = 0.0f;
      p0 -= dp0;
      p0 = a.camlr == 0 ? ((0.0f < p0) ? 0.0f : p0) : ((p0 < 0.0f) ? 0.0f : p0);  // std::min / std::max (p, 0)
    } else {
      const float y0 = div_by(b0, l00, r00);
      const float y1 = div_by(b1 - l10 * y0, l11, r11);
      dp1 = div_by(y1, l11, r11);
      dp0 = div_by(y0 - l10 * dp1, l00, r00);
      p0 -= dp0;
      p1 -= dp1;
    }
    ptx = rx + p0;
    pty = ry + p1;
    const float ex = stx - ptx, ey = sty - pty;
    const bool reset = (ex * ex + ey * ey > a.outlier_sq_max) | oob(ptx, pty) | !(isfinite(ptx) & isfinite(pty));
    p0 = reset ? pin0 : p0;
    p1 = reset ? pin1 : p1;
    ptx = rx + p0;
    pty = ry + p1;
    compute_err(reset);
  }
  if (live && pl == 0) {
    float* pout = a.p_out + ((size_t)frame * g.nop + patch_slot(g, gx, gy)) * 2;
    pout[0] = p0;
    pout[1] = p1;
  }
}

// ------------------------------------------------------------------------------------ RGB 12x12, exact contract
// The same 16-lanes-per-patch mapping for the EXACT contract.  Its sums must run in the documented order for more than 64
// entries -- 64 strided partials x[l], x[l + 64], ... accumulated in turn, then the butterfly at distance 1, 2, 4, 8, 16, 32
// (patch_sum<M, 64>, oracle/eigen_shim -DOFDIS_SHIM_WAVE64) -- which is tied to entry indices, not to the 3x3 pixel blocks the
// taps want.  So an evaluation works in TWO layouts:
//   A  "blocks": lane (rg, cg) owns the 3x3 pixel block as above: the 12 sixteen-byte tap loads, the interpolation
//      ((we0 a + we1 b) + we2 c) + we3 d in the reference's order;
//   B  "chains": lane pl owns the four virtual lanes L = 16 c + pl (c = 0..3) of the one-patch-per-wavefront mapping, i.e.
//      the entries k = L + 64 m: the template, its gradients, the residual and every sum.  A sum = the four chains
//      accumulated in turn (m ascending), each reduced over the patch's 16 lanes by the four DPP steps (distance 1, 2, 4, 8
//      of the documented butterfly), then (s0 + s1) + (s2 + s3): distance 16 and 32, in the operand order of
//      swap16_sum / swap32_sum.  Same additions in the same order as the generic kernel: same bits.
// The interpolated values go from A to B through LDS once per evaluation (27 four-byte writes and reads per lane, inside
// the wavefront: LDS operations of a wavefront execute in order, no barrier).  About 150 instead of 515 instructions per
// patch and iteration.
// NOC = 1: gray 12 x 12 (144 entries: chain 0 has three entries k = pl, pl + 64, pl + 128, chains 1-3 two).  STEREO: the
// 1-D search of the depth mode.
// BS = 2: RGB 8 x 8 (192 entries: three per chain).
template <int COST, int NOC, bool STEREO, int BS>
__global__ __launch_bounds__(256, 3) void patch_optimize_rgb12x_kernel(const DisArgs a) {
  constexpr int Q = 4, NV = 16 * BS * BS * NOC;
  constexpr int P = 4 * BS, HP = P / 2;
  constexpr bool PIXW = NOC == 3 && BS == 3;
  constexpr int RL = P * NOC;          // floats per patch row
  constexpr int MC = (NV + 63) / 64;   // slots per chain (7 / 3); NB = 4 chains
  constexpr int NB = 4 * MC;
  __shared__ float xl[4 * Q * NV];  // [wavefront][patch][entry]: the A -> B hand-over
  const LevelGeom& g = a.g;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int blocks_per_frame = (g.nop + 4 * Q - 1) / (4 * Q);
  int frame, blk;
  xcd_frame_map(blockIdx.x, blocks_per_frame, a.nframes, frame, blk);
  if (frame >= a.nframes) return;               // block-uniform
  if ((blk * 4 + wave) * Q >= g.nop) return;    // wave-uniform: no patch for this wavefront
  const int sub = lane >> 4, pl = lane & 15;
  const int rg = pl >> 2, cg = pl & 3;
  int jp = (blk * 4 + wave) * Q + sub;
  const bool live = jp < g.nop;
  if (!live) jp = g.nop - 1;
  const int gy = jp / g.nopw, gx = jp - gy * g.nopw;
  float* const xp = xl + (wave * Q + sub) * NV;  // this patch's hand-over vector

  const int tw = g.tmp_w;
  const size_t plane = g.plane_elems;
  auto plane_rsrc = [&](const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)frame * plane), 0, (int)(plane * sizeof(float)), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsB = plane_rsrc(a.im_b);
  const int row_bytes = tw * 4 * NOC;
  const float rx = (float)(gx * g.steps + g.offw), ry = (float)(gy * g.steps + g.offh);
  const float fnv = (float)NV, rcp_nv = rcp_refined(fnv);
  auto div_nv = [&](float x) { return div_by(x, fnv, rcp_nv); };  // == x / 432, correctly rounded (ofdis_dev.h)
  // layout B: entry i = c * MC + m of this lane is k = 16 c + pl + 64 m; RGB: chain 3 has six entries (k < 432)
  auto kB = [&](int c, int m) { return 16 * c + pl + 64 * m; };
  auto vB = [](int c, int m) { return 16 * c + 64 * m + 15 < NV; };  // (NV is a multiple of 16: the same for every lane)
  // sum of one value per entry in the documented order (see above): the four chain sums (accumulated by the caller, m
  // ascending, starting FROM the first entry), each reduced over the patch's 16 lanes, then distance 16 and 32
  auto finish4 = [&](const float (&sc)[4]) {
    const float r0 = row16_sum(sc[0]), r1 = row16_sum(sc[1]), r2 = row16_sum(sc[2]), r3 = row16_sum(sc[3]);
    return (r0 + r1) + (r2 + r3);
  };
  auto sumB = [&](const float (&x)[NB]) {
    float sc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float t = x[c * MC];
#pragma unroll
      for (int m = 1; m < MC; ++m)
        if (vB(c, m)) t = t + x[c * MC + m];
      sc[c] = t;
    }
    return finish4(sc);
  };

  // ---- InitializePatch (patch.cpp:287-332) in layout B
  float T[NB], Tx[NB], Ty[NB];
  {
    const int px = (int)roundf(rx) + g.pad, py = (int)roundf(ry) + g.pad;
    const float* __restrict__ imA = a.im_a + (size_t)frame * plane;
    const float* __restrict__ imAx = a.im_a_dx + (size_t)frame * plane;
    const float* __restrict__ imAy = a.im_a_dy + (size_t)frame * plane;
    const int base = ((py - HP) * tw + px - HP) * NOC;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < MC; ++m) {
        const int i = c * MC + m;
        if (!vB(c, m)) { T[i] = Tx[i] = Ty[i] = 0.0f; continue; }
        const int k = kB(c, m), row = k / RL, col = k - row * RL;
        const unsigned o = (unsigned)(base + row * tw * NOC + col);
        T[i] = imA[o]; Tx[i] = imAx[o]; Ty[i] = imAy[o];
      }
    if (a.patnorm > 0) {
      const float mean = div_nv(sumB(T));
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < MC; ++m)
          if (vB(c, m)) T[c * MC + m] -= mean;
    }
  }
  // ---- ComputeHessian + Cholesky factor (patch.cpp:71-88, Eigen LLT as in oracle/eigen_shim)
  float l00, l10, l11;
  {
    float sxx[4], sxy[4], syy[4];  // chain sums on the fly (the products are not kept)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < MC; ++m) {
        if (!vB(c, m)) continue;
        const int i = c * MC + m;
        const float xx = Tx[i] * Tx[i], xy = Tx[i] * Ty[i], yy = Ty[i] * Ty[i];
        sxx[c] = m ? sxx[c] + xx : xx;
        sxy[c] = m ? sxy[c] + xy : xy;
        syy[c] = m ? syy[c] + yy : yy;
      }
    float H00 = finish4(sxx);
    if constexpr (STEREO) {  // 1x1 Hessian of the horizontal displacement (patch.cpp:83-87)
      if (H00 == 0.0f) H00 = (float)((double)H00 + 1e-10);
      l00 = H00; l10 = 0.0f; l11 = 1.0f;
      if (!(l00 <= 0.0f)) l00 = sqrtf(l00);
    } else {
      const float H01 = finish4(sxy);
      float H11 = finish4(syy);
      if (H00 * H11 - H01 * H01 == 0.0f) {
        H00 = (float)((double)H00 + 1e-10);
        H11 = (float)((double)H11 + 1e-10);
      }
      l00 = H00; l10 = H01; l11 = H11;
      if (!(l00 <= 0.0f)) {
        l00 = sqrtf(l00);
        l10 = l10 / l00;
        const float x = l11 - l10 * l10;
        if (!(x <= 0.0f)) l11 = sqrtf(x);
      }
    }
  }
  // ---- InitializeFromCoarserOF (patchgrid.cpp:195-211)
  float pin0 = 0.0f, pin1 = 0.0f;
  if (a.flow_prev) {
    const int x = (int)floorf(rx / 2), y = (int)floorf(ry / 2);
    const int i = y * (g.w / 2) + x;
    if constexpr (STEREO) {  // one channel
      pin0 = (a.flow_prev + (size_t)frame * (size_t)(g.w / 2) * (g.h / 2))[i] * 2;
    } else {
      const float* fp = a.flow_prev + (size_t)frame * (size_t)(g.w / 2) * (g.h / 2) * 2;
      pin0 = fp[2 * i] * 2;
      pin1 = fp[2 * i + 1] * 2;
    }
  }
  // ---- OptimizeIter (patch.cpp:159-212)
  const float r00 = rcp_refined(l00), r11 = rcp_refined(l11);
  float p0 = pin0, p1 = pin1;
  float ptx = rx + p0, pty = ry + p1;
  const float stx = ptx, sty = pty;
  float dp0 = 0.0f, dp1 = 0.0f;
  float dpsq = 1e-10f, dpsq_init = 1e-10f, mares = 1e20f, mares_old = 1e20f;
  float b0 = 0.0f, b1 = 0.0f;
  int cnt = 0;
  bool converged = false;
  float* const pwf = a.pweight + (size_t)frame * g.nop * NV;
  // (see patch_optimize_rgb12_kernel: one float per pixel for the patches the densification reads unshifted.  The residual
  // lives in layout B here, a pixel's three channels in three lanes: |r| goes through the hand-over vector once more and is
  // read back by pixel blocks, layout A)
  const bool compact = PIXW && a.pixw != nullptr && patch_weights_unshifted(g, gx, gy);
  float* const pxout = (PIXW && a.pixw) ? a.pixw + (size_t)frame * g.nop * 144 + pixw_row(g, gx, gy, 3 * rg) + 3 * cg : nullptr;
  const int pxstride = g.nopw * 12;
  auto store_pw = [&](const float (&v)[NB], bool zero) {  // |residual| of this lane's entries (layout B)
    if constexpr (PIXW) {
      if (compact) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int m = 0; m < MC; ++m)
            if (vB(c, m)) xp[kB(c, m)] = zero ? 0.0f : fabsf(v[c * MC + m]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
          for (int xx = 0; xx < 3; ++xx) {
            const float* e = xp + (3 * rg + rr) * 36 + 9 * cg + 3 * xx;
            float sden = fmaxf(2.0f, e[0]);
            sden += fmaxf(2.0f, e[1]);
            sden += fmaxf(2.0f, e[2]);
            pxout[rr * pxstride + xx] = sden;
          }
        __builtin_amdgcn_wave_barrier();  // (a later evaluation of the wavefront's other patches does not touch this vector)
        return;
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < MC; ++m) {
        if (!vB(c, m)) continue;
        const int k = kB(c, m), row = k / RL;  // (RL floats per patch row: a compile-time divisor)
        pwf[pweight_row(g, gx, gy, row) + (k - row * RL)] = zero ? 0.0f : fabsf(v[c * MC + m]);
      }
  };

  auto compute_err = [&](bool stop) {  // patch.cpp:264-284, 335-402, 223-262
    int pos0 = (int)ceilf(ptx + .00001f), pos1 = (int)ceilf(pty + .00001f);
    const int pos2 = (int)floorf(ptx), pos3 = (int)floorf(pty);
    const float r0 = ptx - (float)pos2, r1 = pty - (float)pos3;
    const float we0 = r0 * r1, we1 = (1 - r0) * r1, we2 = r0 * (1 - r1), we3 = (1 - r0) * (1 - r1);
    pos0 += g.pad;
    pos1 += g.pad;
    {  // layout A: the lane's 3x3 pixel block from its 4x4 pixel window, handed over entry by entry
      const int voff = ((pos1 - (HP + 1) + BS * rg) * tw + pos0 - (HP + 1) + BS * cg) * (4 * NOC);
      float W[BS + 1][(BS + 1) * NOC];
#pragma unroll
      for (int j = 0; j <= BS; ++j) buffer_load_floats<(BS + 1) * NOC>(rsB, voff, j * row_bytes, W[j]);
      __builtin_amdgcn_wave_barrier();  // (the previous evaluation's reads of the hand-over vector come first)
#pragma unroll
      for (int rr = 0; rr < BS; ++rr)
#pragma unroll
        for (int q = 0; q < BS * NOC; ++q)
          xp[(BS * rg + rr) * RL + BS * NOC * cg + q] =
              we0 * W[rr + 1][q + NOC] + we1 * W[rr + 1][q] + we2 * W[rr][q + NOC] + we3 * W[rr][q];  // patch.cpp:391
      __builtin_amdgcn_wave_barrier();
    }
    float v[NB];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < MC; ++m) v[c * MC + m] = vB(c, m) ? xp[kB(c, m)] : 0.0f;
    if (a.patnorm > 0) {
      const float mean = div_nv(sumB(v));
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < MC; ++m)
          if (vB(c, m)) v[c * MC + m] -= mean;
    }
    float sgx[4], sgy[4], spw[4];  // chain sums on the fly: Tx . r, Ty . r, |r|
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < MC; ++m) {
        if (!vB(c, m)) continue;
        const int i = c * MC + m;
        float d = v[i] - T[i];
        if (COST == 1) d = copysignf(sqrt_rn(fabsf(d)), d);  // L1 (patch.cpp:238-246)
        v[i] = d;
        const float gx_ = Tx[i] * d, gy_ = Ty[i] * d, pw_ = fabsf(d);
        sgx[c] = m ? sgx[c] + gx_ : gx_;
        sgy[c] = m ? sgy[c] + gy_ : gy_;
        spw[c] = m ? spw[c] + pw_ : pw_;
      }
    b0 = finish4(sgx);
    b1 = finish4(sgy);
    dpsq = dp0 * dp0 + dp1 * dp1;
    if (cnt == 1) dpsq_init = dpsq;
    mares_old = mares;
    mares = div_nv(finish4(spw));
    bool go = (cnt < a.max_iter) && (mares > a.res_thresh);
    if (go && cnt >= a.min_iter)
      go = (div_rn(dpsq, dpsq_init) >= a.dp_thresh_sq) && (div_rn(mares, mares_old) <= a.dr_thresh);
    if (!go | stop) {
      converged = true;
      if (live) store_pw(v, false);
    }
  };
  auto oob = [&](float x, float y) { return (x < g.lb) | (y < g.lb) | (x > g.ubw) | (y > g.ubh); };

  if (oob(ptx, pty) || !(isfinite(ptx) && isfinite(pty))) {
    converged = true;  // OptimizeStart (patch.cpp:120-156): no evaluation, pweight keeps its initial zeros
    if (live) store_pw(T, true);
  } else {
    dpsq = 1e-10f; dpsq_init = 1e-10f; mares = 1e5f; mares_old = 1e20f;
    compute_err(false);
  }
  while (!converged) {
    cnt++;
    if constexpr (STEREO) {  // patch.cpp:180-193: 1x1 system, then the disparity sign constraint of the camera side
      dp0 = (b0 / l00) / l00;
      dp1 = 0.0f;
      p0 -= dp0;
      p0 = a.camlr == 0 ? ((0.0f < p0) ? 0.0f : p0) : ((p0 < 0.0f) ? 0.0f : p0);  // std::min / std::max (p, 0)
    } else {
      const float y0 = div_by(b0, l00, r00);
      const float y1 = div_by(b1 - l10 * y0, l11, r11);
      dp1 = div_by(y1, l11, r11);
      dp0 = div_by(y0 - l10 * dp1, l00, r00);
      p0 -= dp0;
      p1 -= dp1;
    }
    ptx = rx + p0;
    pty = ry + p1;
    const float ex = stx - ptx, ey = sty - pty;
    const bool reset = (ex * ex + ey * ey > a.outlier_sq_max) | oob(ptx, pty) | !(isfinite(ptx) & isfinite(pty));
    p0 = reset ? pin0 : p0;
    p1 = reset ? pin1 : p1;
    ptx = rx + p0;
    pty = ry + p1;
    compute_err(reset);
  }
  if (live && pl == 0) {
    float* pout = a.p_out + ((size_t)frame * g.nop + patch_slot(g, gx, gy)) * 2;
    pout[0] = p0;
    pout[1] = p1;
  }
}

// (a template so that the exact contract's object never instantiates the kernel above)
template <bool FUSED>
hipError_t launch_patch_optimize_rgb12(const DisArgs& a, hipStream_t s) {
  const int wpf = (a.g.nop + 3) / 4;  // four patches per wavefront
  const int blocks_per_frame = (wpf + 3) / 4;
  const dim3 gd(((a.nframes + 7) / 8) * 8 * blocks_per_frame), bd(256);
  const bool l1 = a.costfct != 0;
#define OFDIS_P16L(KERNEL, NOC, BS)                                                                          \
  do {                                                                                                       \
    if (a.stereo) {                                                                                          \
      if (l1) hipLaunchKernelGGL((KERNEL<1, NOC, true, BS>), gd, bd, 0, s, a);                               \
      else hipLaunchKernelGGL((KERNEL<0, NOC, true, BS>), gd, bd, 0, s, a);                                  \
    } else {                                                                                                 \
      if (l1) hipLaunchKernelGGL((KERNEL<1, NOC, false, BS>), gd, bd, 0, s, a);                              \
      else hipLaunchKernelGGL((KERNEL<0, NOC, false, BS>), gd, bd, 0, s, a);                                 \
    }                                                                                                        \
  } while (0)
#define OFDIS_P16L_GEOM(KERNEL)                                                                              \
  do {                                                                                                       \
    if (a.g.P == 8) OFDIS_P16L(KERNEL, 3, 2);                                                                \
    else if (a.g.noc == 1) OFDIS_P16L(KERNEL, 1, 3);                                                         \
    else OFDIS_P16L(KERNEL, 3, 3);                                                                           \
  } while (0)
  if constexpr (FUSED) OFDIS_P16L_GEOM(patch_optimize_rgb12_kernel);
  else OFDIS_P16L_GEOM(patch_optimize_rgb12x_kernel);
#undef OFDIS_P16L_GEOM
#undef OFDIS_P16L
  return hipGetLastError();
}

// Does launch_patch_optimize run a kernel that honours DisArgs::pixw for these arguments?  (The 16-lanes-per-patch RGB 12x12
// kernels of either contract; the caller passes pixw to the patch kernel and to the densification only then.)
bool patch_pixel_weights_supported(const DisArgs& a, const ofdis_tuning* tnp) {
  const ofdis_tuning tn = tnp ? *tnp : tuning();
  const int rgb12_lpp = tn.rgb12_lpp == 0 ? 16 : tn.rgb12_lpp;
  return a.g.novals == 432 && a.g.P == 12 && !a.stereo && (a.costfct == 0 || a.costfct == 1) && tn.rgb12 && rgb12_lpp == 16;
}

hipError_t launch_patch_optimize(const DisArgs& a, hipStream_t s, const ofdis_tuning* tnp) {
  const int M = (a.g.novals + 63) / 64;
  const bool full = a.g.novals == 64 * M;
  const ofdis_tuning tn = tnp ? *tnp : tuning();
  const bool gray8 = a.g.noc == 1 && a.g.P == 8 && tn.gray8;
  // RGB 12x12 (operating points 3 and 4, BASELINE configs[3]): 432 entries, 6 full groups of 64 + 48.  ofdis_tuning::rgb12_lpp = 32:
  // two patches per wavefront (the scalar solve, predicates and every reduction instruction shared by two patches:
  // 929 -> 705 instructions per patch and iteration, 82 -> 127 VGPRs).  Measured on configs[3] the same 19.6 +- 0.4 ms per
  // 16-pair level-1 launch either way -- patches that reset early leave their wavefront's other half running alone
  // (PMC: 18 % fewer VALU instructions, same time) -- so one patch per wavefront stayed the default until the
  // 16-lanes-per-patch kernels (round 4) took it over.
  const int rgb12_lpp = tn.rgb12_lpp == 0 ? 16 : tn.rgb12_lpp;
  const bool rgb12 = a.g.novals == 432 && !a.stereo && (a.costfct == 0 || a.costfct == 1) && tn.rgb12;
  // 12x12 patches, RGB or gray, and RGB 8x8 patches, flow or stereo, have their own mapping (16 lanes per patch, four
  // patches per wavefront: ofdis_tuning::rgb12 with rgb12_lpp = 16, the default)
  const bool p12 = ((a.g.P == 12 && (a.g.noc == 1 || a.g.noc == 3)) || (a.g.P == 8 && a.g.noc == 3)) &&
                   (a.costfct == 0 || a.costfct == 1) && tn.rgb12;
  if (p12 && rgb12_lpp == 16) return launch_patch_optimize_rgb12<kFusedContract>(a, s);
  if (a.pixw) return hipErrorInvalidValue;  // (only the kernels above write the compact weights)
  const int lpp = (M <= 1) ? (gray8 ? 4 : 8) : ((rgb12 && rgb12_lpp == 32) ? 32 : 64);  // lanes per patch
  const int ppw = 64 / lpp;                           // patches per wavefront
  const int wpf = (a.g.nop + ppw - 1) / ppw;          // wavefronts per frame
  const bool fast = gray8 && M <= 1;
  const int wpb = (fast && wpf < 4) ? wpf : 4;        // wavefronts per block (the generic kernels assume 4)
  const int blocks_per_frame = (wpf + wpb - 1) / wpb;
  const int grid = ((a.nframes + 7) / 8) * 8 * blocks_per_frame;
  const dim3 gd(grid), bd(wpb * 64);
  if (M <= 1 && gray8 && a.stereo)
    hipLaunchKernelGGL((patch_optimize_gray8_kernel<-1, true>), gd, bd, 0, s, a);
  else if (M <= 1 && gray8 && a.costfct == 0)
    hipLaunchKernelGGL((patch_optimize_gray8_kernel<0, false>), gd, bd, 0, s, a);
  else if (M <= 1 && gray8)
    hipLaunchKernelGGL((patch_optimize_gray8_kernel<-1, false>), gd, bd, 0, s, a);
  else if (M <= 1 && full && a.costfct == 0)
    hipLaunchKernelGGL((patch_optimize_kernel<1, 8, 64, 0>), gd, bd, 0, s, a);
  else if (M <= 1 && full)
    hipLaunchKernelGGL((patch_optimize_kernel<1, 8, 64, -1>), gd, bd, 0, s, a);
  else if (M <= 1)
    hipLaunchKernelGGL((patch_optimize_kernel<1, 8, 0, -1>), gd, bd, 0, s, a);
  else if (M <= 3)
    hipLaunchKernelGGL((patch_optimize_kernel<3, 64, 0, -1>), gd, bd, 0, s, a);
  else if (rgb12 && lpp == 32 && a.costfct == 0)
    hipLaunchKernelGGL((patch_optimize_kernel<7, 32, 432, 0>), gd, bd, 0, s, a);
  else if (rgb12 && lpp == 32)
    hipLaunchKernelGGL((patch_optimize_kernel<7, 32, 432, 1>), gd, bd, 0, s, a);
  else if (rgb12 && a.costfct == 0)
    hipLaunchKernelGGL((patch_optimize_kernel<7, 64, 432, 0>), gd, bd, 0, s, a);
  else if (rgb12)
    hipLaunchKernelGGL((patch_optimize_kernel<7, 64, 432, 1>), gd, bd, 0, s, a);
  else if (M <= 7)
    hipLaunchKernelGGL((patch_optimize_kernel<7, 64, 0, -1>), gd, bd, 0, s, a);
  else if (M <= 12)
    hipLaunchKernelGGL((patch_optimize_kernel<12, 64, 0, -1>), gd, bd, 0, s, a);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------ densify
// Reference: for ip ascending, for every pixel of the patch inside the image:
//   absw = 1/max(2,|r|)  (RGB: 1/sum_c max(2,|r_c|));  we += absw;  flow += p*absw;
// then flow /= we where we > 0 (patchgrid.cpp:221-271, 377-397).  A pixel is covered by at most
// ceil(P/steps)^2 patches; visiting them with gx ascending then gy ascending reproduces the
// reference's ip order, so the sums are bit-identical without atomics.
// Forward-backward merging (usefbcon = 1, patchgrid.cpp:277-375): after the grid's own patches, every patch of the
// COMPLEMENTARY grid (ascending index) splats its negated displacement around its final position with bilinear
// weights: patch pixel (xt,yt) adds w0 to pixel (xt,yt), w1 to (xt-1,yt), w2 to (xt,yt-1), w3 to (xt-1,yt-1).
// As a gather, pixel t therefore receives from one patch, in the reference's order, w0 from source t, w1 from t+(1,0),
// w2 from t+(0,1), w3 from t+(1,1).  The block first compacts -- preserving the index order -- the patches whose
// footprint reaches its rows into LDS (256 candidates per round), then every pixel walks that list.
struct FbCand {
  int gx, gy, pos0, pos1;
  float w0, w1, w2, w3, p0, p1;
  int wbase;         // offset of the patch's first weight row within the frame's weights (ofdis_dev.h: pweight_row)
  int L, Rr, T, Bt;  // valid source rectangle of the patch in patch coordinates (condition patchgrid.cpp:321)
};

// FB (forward-backward merging): a block is a 16 x 16 pixel TILE and a wavefront a 16 x 4 strip of it, so that the list a
// pixel walks holds the complementary patches that reach its tile (~40) instead of every patch of its rows (~100 at level 3
// of operating point 2, 3 000 at operating point 3) and a wavefront skips -- uniformly -- those that miss its strip: the
// walk was 634 us per 512-pair level-3 launch against 25 us for the grid's own patches.
// FBN: 0 = no forward-backward merging, else the channel count of the complementary grid's weights (1 / 3: compile-time, so
// that the gray walk does not carry the RGB running-pointer arithmetic)
template <bool PLANAR, int FBN>
__global__ __launch_bounds__(256) void densify_kernel(const DensifyArgs a) {
  constexpr bool FB = FBN != 0;
  const LevelGeom& g = a.g;
  const int npx = g.w * g.h;
  const int tiles_x = (g.w + 15) / 16;
  const int blocks_per_frame = FB ? tiles_x * ((g.h + 15) / 16) : (npx + 255) / 256;
  int frame, blk;
  xcd_frame_map(blockIdx.x, blocks_per_frame, a.nframes, frame, blk);  // a frame's blocks share one XCD's L2
  if (frame >= a.nframes) return;
  constexpr bool fb = FB;  // (= a.cg_p != nullptr: the launcher picks the instantiation)
  int y, x, i;
  bool active;
  if constexpr (FB) {
    const int ty = blk / tiles_x, tx = blk - ty * tiles_x;
    x = tx * 16 + (threadIdx.x & 15);
    y = ty * 16 + (threadIdx.x >> 4);
    active = x < g.w && y < g.h;
    i = y * g.w + x;
  } else {
    i = blk * 256 + threadIdx.x;
    active = i < npx;
    if (!active) return;
    // i / w by multiply-high with ceil(2^32 / w) (exact for i * w < 2^32, checked by the launcher; 0 = divide)
    auto split = [&](int n, int d) { return a.idx_magic ? (int)__umulhi((unsigned)n, a.idx_magic) : n / d; };
    y = split(i, g.w);
    x = i - y * g.w;
  }
  const long long idx = (long long)frame * npx + i;
  const int P = g.P, lb = -P / 2, ub = P / 2 - 1, st = g.steps, noc = g.noc;
  float we = 0.0f, fu = 0.0f, fv = 0.0f;
  if (active) {
    const float* pf = a.p + (size_t)frame * g.nop * 2;
    const float* pwf = a.pweight + (size_t)frame * g.nop * g.novals;
    const float* pxf = a.pixw ? a.pixw + (size_t)frame * g.nop * g.P * g.P : nullptr;
    if (g.noc == 1 && g.P <= 2 * g.steps)  // at most 2 x 2 covering patches (uniform)
      densify_accumulate_gray<2>(g, pf, pwf, x, y, we, fu, fv);
    else
      densify_accumulate(g, pf, pwf, x, y, we, fu, fv, pxf);
  }
  if constexpr (FB) {
    __shared__ FbCand cand[256];
    __shared__ int wave_cnt[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tyb = blk / tiles_x, txb = blk - tyb * tiles_x;
    const int yb0 = tyb * 16, yb1 = min(g.h, yb0 + 16) - 1;  // rows / columns of this block's pixels
    const int xb0 = txb * 16, xb1 = min(g.w, xb0 + 16) - 1;
    const int yw0 = yb0 + 4 * wave, yw1 = yw0 + 3;           // ... and rows of this wavefront's strip
    const float* cpf = a.cg_p + (size_t)frame * g.nop * 2;
    const float* cpwf = a.cg_pweight + (size_t)frame * g.nop * g.novals;
    const int rowlen = P * noc, wstride = g.nopw * rowlen;  // floats per patch row / between two rows of one patch
    for (int base = 0; base < g.nop; base += 256) {
      const int ipc = base + threadIdx.x;
      FbCand c;
      bool hit = false;
      if (ipc < g.nop) {
        const int gx = ipc / g.noph, gy = ipc - gx * g.noph;  // candidates in the reference's index order
        c.gx = gx;
        c.gy = gy;
        c.p0 = cpf[2 * patch_slot(g, gx, gy)];
        c.p1 = cpf[2 * patch_slot(g, gx, gy) + 1];
        // GetPointPos(): pt_iter = pt_ref + p_iter (patch.cpp:214-221)
        const float rp0 = (float)(gx * st + g.offw) + c.p0, rp1 = (float)(gy * st + g.offh) + c.p1;
        c.pos0 = (int)ceil((double)rp0 + .00001);  // the reference adds a DOUBLE here (patchgrid.cpp:304-305)
        c.pos1 = (int)ceil((double)rp1 + .00001);
        const float r0 = rp0 - (float)(int)floorf(rp0), r1 = rp1 - (float)(int)floorf(rp1);
        c.w0 = r0 * r1;
        c.w1 = (1 - r0) * r1;
        c.w2 = r0 * (1 - r1);
        c.w3 = (1 - r0) * (1 - r1);
        // (once per candidate instead of once per pixel and candidate)
        c.wbase = (int)pweight_row(g, gx, gy, 0);
        c.L = max(0, 1 - (c.pos0 + lb));
        c.Rr = min(P - 1, g.w - 2 - (c.pos0 + lb));
        c.T = max(0, 1 - (c.pos1 + lb));
        c.Bt = min(P - 1, g.h - 2 - (c.pos1 + lb));
        // source rows ys = pos1 + lb .. pos1 + ub feed target rows ys - 1 and ys (columns likewise)
        hit = (c.pos1 + ub >= yb0) && (c.pos1 + lb - 1 <= yb1) && (c.pos0 + ub >= xb0) && (c.pos0 + lb - 1 <= xb1);
      }
      const unsigned long long bal = __ballot(hit);
      const int before = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) wave_cnt[wave] = __popcll(bal);
      __syncthreads();
      int off = 0, n = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < wave) off += wave_cnt[q];
        n += wave_cnt[q];
      }
      if (hit) cand[off + before] = c;
      __syncthreads();
      {
        for (int k = 0; k < n; ++k) {
          const FbCand& cc = cand[k];
          if ((cc.pos1 + ub < yw0) | (cc.pos1 + lb - 1 > yw1)) continue;  // (wave-uniform) misses this strip's rows
          if (!active) continue;
          const int L = cc.L, Rr = cc.Rr, T = cc.T, Bt = cc.Bt;
          const int in_row = Rr - L + 1;
          const int dx = x - cc.pos0 - lb, dy = y - cc.pos1 - lb;
          const float* wrow = cpwf + cc.wbase;  // weight row r of this patch: wrow + r * wstride
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int kx = dx + (q & 1), ky = dy + (q >> 1);
            if (kx < L || kx > Rr || ky < T || ky > Bt) continue;
            const float wq = q == 0 ? cc.w0 : (q == 1 ? cc.w1 : (q == 2 ? cc.w2 : cc.w3));
            float absw;
            if constexpr (FBN == 1) {
              absw = 1.0f / fmaxf(2.0f, wrow[ky * wstride + kx]);
            } else {  // running pointer: +1 per visited pixel, +2 more per pixel that passed the condition
              const int pidx = ky * P + kx + 2 * ((ky - T) * in_row + (kx - L));
              auto wat = [&](int k) { const int r = k / rowlen; return wrow[r * wstride + (k - r * rowlen)]; };
              absw = fmaxf(2.0f, wat(pidx));
              absw += fmaxf(2.0f, wat(pidx + 1));
              absw += fmaxf(2.0f, wat(pidx + 2));
              absw = 1.0f / absw;
            }
            we += wq * absw;
            fu -= wq * (cc.p0 * absw);  // reversed flow
            fv -= wq * (cc.p1 * absw);
          }
        }
      }
      __syncthreads();
    }
    if (!active) return;
  }
  if (we > 0) {
    fu /= we;
    fv /= we;
  }
  if (PLANAR) {
    a.wx[idx] = fu;
    a.wy[idx] = fv;
  } else if (a.stereo) {
    a.flow_aos[idx] = fu;  // one channel (patchgrid.cpp:267,390)
  } else {
    reinterpret_cast<float2*>(a.flow_aos)[idx] = make_float2(fu, fv);
  }
}

// Gray patches with P = 8 on a grid of step 4 (operating point 2), AoS output: the 4 pixels x0 .. x0+3 of a row with
// x0 = 4 bx + offw - 4 are covered by the same (at most) 2 x 2 patches -- columns bx-1 and bx of the grid, where they are the
// patch entries 4..7 and 0..3 of one patch row -- so a thread takes such a quad: one aligned 16-byte load of weights and
// one 8-byte load of the displacement per patch instead of four scattered 4-byte loads per PIXEL, the index arithmetic
// once per quad, and 32 contiguous bytes out.  Same candidates in the same order (grid column ascending, then grid row),
// the same additions: same bits as densify_kernel.
__global__ __launch_bounds__(256) void densify_quad_kernel(const DensifyArgs a, const int nbx) {
  const LevelGeom& g = a.g;
  const int w = g.w, h = g.h;
  const int quads = nbx * h;
  const int blocks_per_frame = (quads + 255) / 256;
  int frame, blk;
  xcd_frame_map(blockIdx.x, blocks_per_frame, a.nframes, frame, blk);  // a frame's blocks share one XCD's L2
  if (frame >= a.nframes) return;
  const int q = blk * 256 + threadIdx.x;
  if (q >= quads) return;
  const int y = q / nbx, bx = q - y * nbx;
  const int x0 = 4 * bx + g.offw - 4;
  const int yy = y - g.offh + 4;  // >= 0: offh < steps
  const int by = yy >> 2, ty = yy & 3;
  const float* __restrict__ pf = a.p + (size_t)frame * g.nop * 2;
  const float* __restrict__ pwf = a.pweight + (size_t)frame * g.nop * 64;
  float4 pw[4];
  float2 pp[4];
  bool ok[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {  // c = 2 * (grid column: 0 = bx-1, 1 = bx) + (grid row: 0 = by-1, 1 = by)
    const int gx = bx - 1 + (c >> 1), gy = by - 1 + (c & 1);
    ok[c] = (gx >= 0) & (gx < g.nopw) & (gy >= 0) & (gy < g.noph);
    const int gxc = clampi(gx, 0, g.nopw - 1), gyc = clampi(gy, 0, g.noph - 1);
    const int kx0 = (c >> 1) ? 0 : 4, ky = (c & 1) ? ty : ty + 4;
    // (internal layout: consecutive threads = consecutive grid columns read consecutive 32-byte patch rows)
    pw[c] = *reinterpret_cast<const float4*>(pwf + pweight_row(g, gxc, gyc, ky) + kx0);
    pp[c] = *reinterpret_cast<const float2*>(pf + 2 * patch_slot(g, gxc, gyc));
  }
  float2* out = reinterpret_cast<float2*>(a.flow_aos) + ((size_t)frame * h + y) * w;
  float2 res[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float we = 0.0f, fu = 0.0f, fv = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float r = t == 0 ? pw[c].x : (t == 1 ? pw[c].y : (t == 2 ? pw[c].z : pw[c].w));
      const float absw = div_rn(1.0f, fmaxf(2.0f, r));  // == 1.0f / x: numerator 1, denominator >= 2 (ofdis_dev.h)
      const float nwe = we + absw, nfu = fu + pp[c].x * absw, nfv = fv + pp[c].y * absw;
      we = ok[c] ? nwe : we;
      fu = ok[c] ? nfu : fu;
      fv = ok[c] ? nfv : fv;
    }
    if (we > 0) {
      fu /= we;
      fv /= we;
    }
    res[t] = make_float2(fu, fv);
  }
  // a quad inside the image goes out as 32 contiguous bytes in two stores (consecutive threads: consecutive 32-byte pieces);
  // the quads that straddle the left / right border pixel by pixel
  if (x0 >= 0 && x0 + 3 < w) {
    typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
    f4a8* o4 = reinterpret_cast<f4a8*>(out + x0);
    o4[0] = f4a8{res[0].x, res[0].y, res[1].x, res[1].y};
    o4[1] = f4a8{res[2].x, res[2].y, res[3].x, res[3].y};
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int x = x0 + t;
      if (x >= 0 && x < w) out[x] = res[t];
    }
  }
}

// p in the reference's patch order (ip = gx * noph + gy, patchgrid.cpp:62-69) from the internal grid-row-major array: the
// public p_out of ofdis_patchgrid_level
__global__ __launch_bounds__(256) void patch_p_reference_order_kernel(const LevelGeom g, int nframes, const float2* __restrict__ src,
                                                                      float2* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)nframes * g.nop) return;
  const int frame = (int)(i / g.nop), ip = (int)(i - (long long)frame * g.nop);
  const int gx = ip / g.noph, gy = ip - gx * g.noph;
  dst[i] = src[(size_t)frame * g.nop + patch_slot(g, gx, gy)];
}
hipError_t launch_patch_p_reference_order(const LevelGeom& g, int nframes, const float* p_internal, float* p_out, hipStream_t s) {
  const long long n = (long long)nframes * g.nop;
  hipLaunchKernelGGL(patch_p_reference_order_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g, nframes,
                     reinterpret_cast<const float2*>(p_internal), reinterpret_cast<float2*>(p_out));
  return hipGetLastError();
}

hipError_t launch_densify(const DensifyArgs& a_in, hipStream_t s) {
  DensifyArgs a = a_in;
  if (a.flow_aos && !a.stereo && !a.cg_p && a.g.noc == 1 && a.g.P == 8 && a.g.steps == 4 && a.g.offw < 4 && a.g.offh < 4) {
    const int nbx = (a.g.w + 3 - a.g.offw) / 4 + 1;  // quads per row: the last one holds column w - 1
    const int blocks_per_frame = (nbx * a.g.h + 255) / 256;
    hipLaunchKernelGGL(densify_quad_kernel, dim3(((a.nframes + 7) / 8) * 8 * blocks_per_frame), dim3(256), 0, s, a, nbx);
    return hipGetLastError();
  }
  {
    const unsigned long long d = a.g.w, npx = (unsigned long long)a.g.w * a.g.h;
    // floor(n * ceil(2^32/d) / 2^32) == floor(n / d) for n * (d - 1) < 2^32 (n < npx here)
    a.idx_magic = (d > 1 && npx * d < (1ull << 32)) ? (unsigned)(((1ull << 32) + d - 1) / d) : 0u;
  }
  // (A patch-major variant -- one wavefront per 32x32 tile accumulating in LDS, one contiguous 256-byte weight read per
  // patch -- was measured at twice the time of this gather: ~90 dependent LDS read-modify-write rounds per tile.)
  const int blocks_per_frame = (a.g.w * a.g.h + 255) / 256;
  const long long blocks = (long long)((a.nframes + 7) / 8) * 8 * blocks_per_frame;
  if (a.cg_p) {  // forward-backward merging: 16 x 16 pixel tiles
    const long long fblocks = (long long)((a.nframes + 7) / 8) * 8 * ((a.g.w + 15) / 16) * ((a.g.h + 15) / 16);
    if (a.g.noc == 1) {
      if (a.flow_aos) hipLaunchKernelGGL((densify_kernel<false, 1>), dim3((unsigned)fblocks), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((densify_kernel<true, 1>), dim3((unsigned)fblocks), dim3(256), 0, s, a);
    } else {
      if (a.flow_aos) hipLaunchKernelGGL((densify_kernel<false, 3>), dim3((unsigned)fblocks), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((densify_kernel<true, 3>), dim3((unsigned)fblocks), dim3(256), 0, s, a);
    }
  } else if (a.flow_aos) hipLaunchKernelGGL((densify_kernel<false, 0>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((densify_kernel<true, 0>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace OFDIS_KNS
}  // namespace ofdis
