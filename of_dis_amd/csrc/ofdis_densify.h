// ofdis_densify.h -- the per-pixel gather of PatGridClass::AggregateFlowDense (patchgrid.cpp:213-275), shared by
// densify_kernel (ofdis_dis.hip) and the fused densify + warp kernel (ofdis_tv.hip).
#pragma once
#include "ofdis_dev.h"

namespace ofdis {
namespace OFDIS_KNS {  // the arithmetic contract this file is being compiled for (ofdis_dev.h)

// Reference: for ip ascending, for every pixel of the patch inside the image:
//   absw = 1/max(2,|r|)  (RGB: 1/sum_c max(2,|r_c|));  we += absw;  flow += p*absw;
// A pixel is covered by at most ceil(P/steps)^2 patches; visiting them with gx ascending then gy ascending
// reproduces the reference's ip order, so the sums are bit-identical without atomics.
// pf = this frame's displacements, pwf = its weights, both in the internal grid-row-major layout (ofdis_dev.h: patch_slot,
// pweight_row); accumulates into we, fu, fv.
__device__ __forceinline__ void densify_accumulate(const LevelGeom& g, const float* __restrict__ pf,
                                                   const float* __restrict__ pwf, int x, int y, float& we, float& fu,
                                                   float& fv, const float* __restrict__ pxf = nullptr) {
  // pxf: this frame's compact per-pixel weight denominators of the RGB patches that are read unshifted (ofdis_dev.h: pixw_row;
  // written by the patch kernel INSTEAD of their pweight), or null
  const int P = g.P, lb = -P / 2, ub = P / 2 - 1, st = g.steps, noc = g.noc;
  // rx + lb <= x <= rx + ub, rx = gx*st + offw
  // floor(n / steps) for 0 <= n < 65536 as a multiply-high with ceil(2^32 / steps): exact because the rounding
  // excess n * (magic * steps - 2^32) stays below 2^32 / steps for these n (steps <= P <= 27)
  const unsigned magic = g.steps_magic;
  auto div_st = [&](int n) { return magic ? (int)__umulhi((unsigned)n, magic) : n; };
  int gx_lo = (x - ub - g.offw + st - 1);
  gx_lo = gx_lo < 0 ? 0 : div_st(gx_lo);
  int gx_hi = x - lb - g.offw;
  gx_hi = gx_hi < 0 ? -1 : div_st(gx_hi);
  if (gx_hi > g.nopw - 1) gx_hi = g.nopw - 1;
  int gy_lo = (y - ub - g.offh + st - 1);
  gy_lo = gy_lo < 0 ? 0 : div_st(gy_lo);
  int gy_hi = y - lb - g.offh;
  gy_hi = gy_hi < 0 ? -1 : div_st(gy_hi);
  if (gy_hi > g.noph - 1) gy_hi = g.noph - 1;
  for (int gx = gx_lo; gx <= gx_hi; ++gx)
    for (int gy = gy_lo; gy <= gy_hi; ++gy) {
      const int ip = patch_slot(g, gx, gy);
      const int rxi = gx * st + g.offw, ryi = gy * st + g.offh;
      const int kx = x - rxi - lb, ky = y - ryi - lb;
      // The reference walks pweight with a RUNNING pointer: +1 per visited patch pixel and, for RGB,
      // +2 more only for pixels inside the image (patchgrid.cpp:242,256-257), so for RGB patches that
      // overlap the border the entries are shifted.  Closed form of that pointer for pixel (kx,ky):
      float absw;
      if (noc == 1) {
        const float pw0 = pwf[pweight_row(g, gx, gy, ky) + kx];
        absw = div_rn(1.0f, fmaxf(2.0f, pw0));  // == 1.0f / x: numerator 1, denominator >= 2 (ofdis_dev.h)
      } else {
        const int left_out = max(0, -(rxi + lb)), right_out = max(0, rxi + ub - (g.w - 1));
        const int top_out = max(0, -(ryi + lb));
        float pw0, pw1, pw2;
        if ((left_out | right_out | top_out) == 0 && pxf) {
          // (the patch kernel already added max(2,|r_0|) + max(2,|r_1|) + max(2,|r_2|) of this pixel, same order)
          absw = div_rn(1.0f, pxf[pixw_row(g, gx, gy, ky) + kx]);
          we += absw;
          fu += pf[2 * ip] * absw;
          fv += pf[2 * ip + 1] * absw;
          continue;
        }
        if ((left_out | right_out | top_out) == 0) {  // the patch lies inside the image: the pointer is (ky * P + kx) * 3
          const float* pw = pwf + pweight_row(g, gx, gy, ky) + kx * 3;
          pw0 = pw[0]; pw1 = pw[1]; pw2 = pw[2];
        } else {  // shifted entries of a border patch: three consecutive LINEAR indices, possibly across a row end
          const int in_row = P - left_out - right_out;
          const int pidx = top_out * P + (ky - top_out) * (3 * in_row + (P - in_row)) + left_out + (kx - left_out) * 3;
          pw0 = pwf[pweight_entry(g, gx, gy, pidx)];
          pw1 = pwf[pweight_entry(g, gx, gy, pidx + 1)];
          pw2 = pwf[pweight_entry(g, gx, gy, pidx + 2)];
        }
        absw = fmaxf(2.0f, pw0);
        absw += fmaxf(2.0f, pw1);
        absw += fmaxf(2.0f, pw2);
        absw = div_rn(1.0f, absw);
      }
      we += absw;
      fu += pf[2 * ip] * absw;
      fv += pf[2 * ip + 1] * absw;
    }
}

// The same sums for gray patches when at most C x C patches cover a pixel (C = ceil(P / steps); 2 at operating points
// 1 and 2): the candidate patches are enumerated with clamped indices so that ALL weight and displacement loads are
// requested before the first is used (the loop above waits for each patch in turn), then accumulated in the same
// order, skipping the candidates that do not exist.  Same additions in the same order => same bits.
template <int C>
__device__ __forceinline__ void densify_accumulate_gray(const LevelGeom& g, const float* __restrict__ pf,
                                                        const float* __restrict__ pwf, int x, int y, float& we, float& fu,
                                                        float& fv) {
  const int P = g.P, lb = -P / 2, ub = P / 2 - 1, st = g.steps;
  const unsigned magic = g.steps_magic;
  auto div_st = [&](int n) { return magic ? (int)__umulhi((unsigned)n, magic) : n; };
  int gx_lo = (x - ub - g.offw + st - 1);
  gx_lo = gx_lo < 0 ? 0 : div_st(gx_lo);
  int gx_hi = x - lb - g.offw;
  gx_hi = gx_hi < 0 ? -1 : div_st(gx_hi);
  if (gx_hi > g.nopw - 1) gx_hi = g.nopw - 1;
  int gy_lo = (y - ub - g.offh + st - 1);
  gy_lo = gy_lo < 0 ? 0 : div_st(gy_lo);
  int gy_hi = y - lb - g.offh;
  gy_hi = gy_hi < 0 ? -1 : div_st(gy_hi);
  if (gy_hi > g.noph - 1) gy_hi = g.noph - 1;
  float pw[C * C], p0[C * C], p1[C * C];
  bool ok[C * C];
#pragma unroll
  for (int a = 0; a < C; ++a)
#pragma unroll
    for (int b = 0; b < C; ++b) {
      const int gx = gx_lo + a, gy = gy_lo + b;
      ok[a * C + b] = (gx <= gx_hi) & (gy <= gy_hi);
      const int gxc = min(gx, g.nopw - 1), gyc = min(gy, g.noph - 1);  // gx_lo, gy_lo >= 0
      const int ip = patch_slot(g, gxc, gyc);
      const int kx = x - (gxc * st + g.offw) - lb, ky = y - (gyc * st + g.offh) - lb;
      // (clamped: in range for every candidate, the right entry for every EXISTING one)
      pw[a * C + b] = pwf[pweight_row(g, gxc, gyc, clampi(ky, 0, P - 1)) + clampi(kx, 0, P - 1)];
      p0[a * C + b] = pf[2 * ip];
      p1[a * C + b] = pf[2 * ip + 1];
    }
#pragma unroll
  for (int k = 0; k < C * C; ++k) {
    const float absw = div_rn(1.0f, fmaxf(2.0f, pw[k]));
    const float nwe = we + absw, nfu = fu + p0[k] * absw, nfv = fv + p1[k] * absw;
    we = ok[k] ? nwe : we;
    fu = ok[k] ? nfu : fu;
    fv = ok[k] ? nfv : fv;
  }
}

}  // namespace OFDIS_KNS
}  // namespace ofdis
