// ofdis_fused.h -- what the variants of the fused TV kernel share (ofdis_fused.hip: throughput, multi-wave, split;
// ofdis_fused_xcu.hip: cross-CU): the trimmed quotients, the per-pixel records and the gray data term.
#pragma once
#include "ofdis_kernels.h"
#include "ofdis_tvmath.h"

namespace ofdis {
namespace OFDIS_KNS {  // the arithmetic contract this file is being compiled for (ofdis_dev.h)

// Quotients of this kernel: denominators are normal and positive by construction (n >= 0.01, sqrt(.. + 1e-6) >= 1e-3,
// det >= (sum of edge weights)^2 > 0) and numerators are finite for finite images, so v_div_fixup_f32 has nothing to
// fix (ofdis_dev.h: div_by_finite).
struct FDen {  // a denominator prepared once for all its quotients
  float nb, r;
};
__device__ __forceinline__ FDen fden(float b) {
  // 0 - b, not -b: a subtraction from +0 is not a negation for the compiler (signed zeros), so it stays one plain
  // instruction and is not folded back into a source modifier (VOP3) of every fma that uses it
  return FDen{0.0f - b, rcp_refined(b)};
}
__device__ __forceinline__ float fdiv_by(float a, const FDen& d) { return div_by_finite(a, d.nb, d.r); }
__device__ __forceinline__ float fdiv_rn(float a, float b) { return fdiv_by(a, fden(b)); }
__device__ __forceinline__ float fdiv_by_sqrt(float num, float x) { return fdiv_rn(num, sqrt_rn(x)); }  // num / sqrt(x)

struct FSlot {
  float a11, a12, a22, b1, b2, sh, sv;  // system of pixel (j, tau - j); a** become the block inverse at step tau
  float dur, dvr;                       // old du,dv of the right neighbour (row tau+1)
  float hl, vt;                         // left / top edge weights (= sh of the left, sv of the upper pixel)
};
struct FRow {
  float wx, wy, du, dv;  // the pixel's (wx, wy) record and its du, dv of before this iteration
};
struct FDer {  // the pixel's derivative record, in the order ofdis_prep.hip stores it
  float ix, iz, ixx, ixz, iy, ixy, iyz, iyy;
};

// Data term of one gray pixel: ofdis_tvmath.h data_term() with the divisions and square roots written out
// (ofdis_dev.h: div_by / sqrt_rn; same bits for the operand ranges the launcher guarantees) and the refined
// reciprocal of each normaliser shared by the two quotients that use it.
// The warp's mask (opticalflow_aux.c:352,381: it multiplies both weights) is not an operand: ofdis_prep.hip stores an
// all-zero record for a masked pixel, and zero derivatives give the same coefficients as zero weights -- every product
// below is then +-0 * finite and every accumulator ends as +0 either way (sums of signed zeros starting from +0).
template <bool BRIGHT>
__device__ __forceinline__ void data_term_gray(const FDer& D, float u, float v, float hd3, float hg3, float& a11,
                                               float& a12, float& a22, float& b1, float& b2) {
  const float ix = D.ix, iy = D.iy, iz = D.iz, ixx = D.ixx, ixy = D.ixy, iyy = D.iyy, ixz = D.ixz, iyz = D.iyz;
  a11 = 0.0f; a12 = 0.0f; a22 = 0.0f; b1 = 0.0f; b2 = 0.0f;
  float tmp, tmp2, n1, n2;
  if (BRIGHT) {  // hd3 != 0 (opticalflow_aux.c:352)
    tmp = iz + ix * u + iy * v;
    n1 = ix * ix + iy * iy + DATANORM;
    const FDen d1 = fden(n1);
    tmp = fdiv_by_sqrt(hd3, fdiv_by(3 * tmp * tmp, d1) + EPS_COLOR);
    tmp = fdiv_by(tmp, d1);
    a11 += tmp * ix * ix;
    a12 += tmp * ix * iy;
    a22 += tmp * iy * iy;
    b1 -= tmp * iz * ix;
    b2 -= tmp * iz * iy;
  }
  n1 = ixx * ixx + ixy * ixy + DATANORM;
  n2 = iyy * iyy + ixy * ixy + DATANORM;
  const FDen d1 = fden(n1), d2 = fden(n2);
  tmp = ixz + ixx * u + ixy * v;
  tmp2 = iyz + ixy * u + iyy * v;
  tmp = fdiv_by_sqrt(hg3, fdiv_by(3 * tmp * tmp, d1) + fdiv_by(3 * tmp2 * tmp2, d2) + EPS_GRAD);
  tmp2 = fdiv_by(tmp, d2);
  tmp = fdiv_by(tmp, d1);
  a11 += tmp * ixx * ixx + tmp2 * ixy * ixy;
  a12 += tmp * ixx * ixy + tmp2 * ixy * iyy;
  a22 += tmp2 * iyy * iyy + tmp * ixy * ixy;
  b1 -= tmp * ixx * ixz + tmp2 * ixy * iyz;
  b2 -= tmp2 * iyy * iyz + tmp * ixy * ixz;
  a11 *= 3; a12 *= 3; a22 *= 3; b1 *= 3; b2 *= 3;
}

// Data term of one RGB pixel (opticalflow_aux.c:383-427 = ofdis_tvmath.h data_term(), noc == 3, mask 1), the same operations
// in the same order with the quotients written out as above; a masked pixel is an all-zero record in every channel (the
// argument of data_term_gray holds channel by channel).
template <bool BRIGHT>
__device__ __forceinline__ void data_term_rgb(const FDer (&D)[3], float u, float v, float hd3, float hg3, float& a11,
                                              float& a12, float& a22, float& b1, float& b2) {
  a11 = 0.0f; a12 = 0.0f; a22 = 0.0f; b1 = 0.0f; b2 = 0.0f;
  if (BRIGHT) {
    float t[3];
    FDen dn[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      t[c] = D[c].iz + D[c].ix * u + D[c].iy * v;
      dn[c] = fden(D[c].ix * D[c].ix + D[c].iy * D[c].iy + DATANORM);
    }
    const float tmp = fdiv_by_sqrt(hd3, fdiv_by(t[0] * t[0], dn[0]) + fdiv_by(t[1] * t[1], dn[1]) + fdiv_by(t[2] * t[2], dn[2]) + EPS_COLOR);
    const float tt[3] = {fdiv_by(tmp, dn[0]), fdiv_by(tmp, dn[1]), fdiv_by(tmp, dn[2])};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a11 += tt[c] * D[c].ix * D[c].ix;
      a12 += tt[c] * D[c].ix * D[c].iy;
      a22 += tt[c] * D[c].iy * D[c].iy;
      b1 -= tt[c] * D[c].iz * D[c].ix;
      b2 -= tt[c] * D[c].iz * D[c].iy;
    }
  }
  FDen d1[3], d2[3];
  float t1[3], t2[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    d1[c] = fden(D[c].ixx * D[c].ixx + D[c].ixy * D[c].ixy + DATANORM);
    d2[c] = fden(D[c].iyy * D[c].iyy + D[c].ixy * D[c].ixy + DATANORM);
    t1[c] = D[c].ixz + D[c].ixx * u + D[c].ixy * v;
    t2[c] = D[c].iyz + D[c].ixy * u + D[c].iyy * v;
  }
  const float tmp = fdiv_by_sqrt(hg3, fdiv_by(t1[0] * t1[0], d1[0]) + fdiv_by(t2[0] * t2[0], d2[0]) + fdiv_by(t1[1] * t1[1], d1[1]) +
                                          fdiv_by(t2[1] * t2[1], d2[1]) + fdiv_by(t1[2] * t1[2], d1[2]) + fdiv_by(t2[2] * t2[2], d2[2]) + EPS_GRAD);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float w1 = fdiv_by(tmp, d1[c]), w2 = fdiv_by(tmp, d2[c]);
    a11 += w1 * D[c].ixx * D[c].ixx + w2 * D[c].ixy * D[c].ixy;
    a12 += w1 * D[c].ixx * D[c].ixy + w2 * D[c].ixy * D[c].iyy;
    a22 += w2 * D[c].iyy * D[c].iyy + w1 * D[c].ixy * D[c].ixy;
    b1 -= w1 * D[c].ixx * D[c].ixz + w2 * D[c].ixy * D[c].iyz;
    b2 -= w2 * D[c].iyy * D[c].iyz + w1 * D[c].ixy * D[c].ixz;
  }
}

// Workgroup barrier of the multi-wave variant's step loop.  Only LDS traffic crosses wavefronts there (the du/dv ring), so
// only the LDS counter is drained: __syncthreads() also waits for vmcnt(0), i.e. for the global row loads that are
// deliberately kept 3-5 steps in flight, and would expose one memory latency per step.
__device__ __forceinline__ void mw_step_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int SLOT_FLOATS = 11;  // FSlot
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Launch of the cross-CU variant (ofdis_fused_xcu.hip); waves = frame groups, R = lanes per frame of a group
hipError_t launch_tv_fused_xcu(const FusedArgs& a, const FusedXcu& x, int waves, int R, hipStream_t s);
// Levels of 65 ... 128 rows: two wavefronts per strip (ofdis_fused_tall.hip)
bool tv_fused_tall_supported(const TvGeom& t, int iterations);
hipError_t launch_tv_fused_tall(const FusedArgs& a, hipStream_t s, int group);

}  // namespace OFDIS_KNS
}  // namespace ofdis
