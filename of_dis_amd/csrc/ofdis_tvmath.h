// ofdis_tvmath.h -- per-pixel arithmetic of the TV-L1 system shared by the tiled kernel (ofdis_tv.hip) and
// the fused diagonal kernel (ofdis_fused.hip).  Operation order = the reference's (opticalflow_aux.c).
#pragma once
#include "ofdis_dev.h"

namespace ofdis {
namespace OFDIS_KNS {  // the arithmetic contract this file is being compiled for (ofdis_dev.h)

// 3-tap flow derivative of refine_variational.cpp:47-48: coeffs = { -0.5, -0, 0.5 }
#define D3_C0 (-0.5f)
#define D3_C1 (-0.0f)
#define D3_C2 (0.5f)
// 5-tap derivative filter of refine_variational.cpp:45-46 through convolve_extract_coeffs(even=0)
// (image.c:338-349): coeffs = { 1/12, -8/12, -0, 8/12, -1/12 }
#define D5_C0 (1.0f / 12.0f)
#define D5_C1 (-8.0f / 12.0f)
#define D5_C2 (-0.0f)
#define D5_C3 (-(-8.0f / 12.0f))
#define D5_C4 (-(1.0f / 12.0f))
#define EPS_SMOOTH (0.001f * 0.001f)
#define EPS_COLOR (0.001f * 0.001f)
#define EPS_GRAD (0.001f * 0.001f)
#define DATANORM (0.1f * 0.1f)

// data term of one pixel (opticalflow_aux.c:342-427).  D(k,c): derivative plane k, channel c.
template <typename DF>
__device__ __forceinline__ void data_term(DF D, int noc, float m, float u, float v, float hd3, float hg3, float& a11,
                                          float& a12, float& a22, float& b1, float& b2) {
  a11 = 0.0f; a12 = 0.0f; a22 = 0.0f; b1 = 0.0f; b2 = 0.0f;
  if (noc == 1) {
    const float ix = D(0, 0), iy = D(1, 0), iz = D(2, 0), ixx = D(3, 0), ixy = D(4, 0), iyy = D(5, 0), ixz = D(6, 0),
                iyz = D(7, 0);
    float tmp, tmp2, n1, n2;
    if (hd3 != 0.0f) {
      tmp = iz + ix * u + iy * v;
      n1 = ix * ix + iy * iy + DATANORM;
      tmp = m * hd3 / sqrtf(3 * tmp * tmp / n1 + EPS_COLOR);
      tmp /= n1;
      a11 += tmp * ix * ix;
      a12 += tmp * ix * iy;
      a22 += tmp * iy * iy;
      b1 -= tmp * iz * ix;
      b2 -= tmp * iz * iy;
    }
    n1 = ixx * ixx + ixy * ixy + DATANORM;
    n2 = iyy * iyy + ixy * ixy + DATANORM;
    tmp = ixz + ixx * u + ixy * v;
    tmp2 = iyz + ixy * u + iyy * v;
    tmp = m * hg3 / sqrtf(3 * tmp * tmp / n1 + 3 * tmp2 * tmp2 / n2 + EPS_GRAD);
    tmp2 = tmp / n2;
    tmp /= n1;
    a11 += tmp * ixx * ixx + tmp2 * ixy * ixy;
    a12 += tmp * ixx * ixy + tmp2 * ixy * iyy;
    a22 += tmp2 * iyy * iyy + tmp * ixy * ixy;
    b1 -= tmp * ixx * ixz + tmp2 * ixy * iyz;
    b2 -= tmp2 * iyy * iyz + tmp * ixy * ixz;
    a11 *= 3; a12 *= 3; a22 *= 3; b1 *= 3; b2 *= 3;
  } else {
    float ix[3], iy[3], iz[3], ixx[3], ixy[3], iyy[3], ixz[3], iyz[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ix[c] = D(0, c); iy[c] = D(1, c); iz[c] = D(2, c); ixx[c] = D(3, c);
      ixy[c] = D(4, c); iyy[c] = D(5, c); ixz[c] = D(6, c); iyz[c] = D(7, c);
    }
    if (hd3 != 0.0f) {
      float t[3], nn[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        t[c] = iz[c] + ix[c] * u + iy[c] * v;
        nn[c] = ix[c] * ix[c] + iy[c] * iy[c] + DATANORM;
      }
      float tmp = m * hd3 / sqrtf(t[0] * t[0] / nn[0] + t[1] * t[1] / nn[1] + t[2] * t[2] / nn[2] + EPS_COLOR);
      const float tt[3] = {tmp / nn[0], tmp / nn[1], tmp / nn[2]};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        a11 += tt[c] * ix[c] * ix[c];
        a12 += tt[c] * ix[c] * iy[c];
        a22 += tt[c] * iy[c] * iy[c];
        b1 -= tt[c] * iz[c] * ix[c];
        b2 -= tt[c] * iz[c] * iy[c];
      }
    }
    float n1[3], n2[3], t1[3], t2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      n1[c] = ixx[c] * ixx[c] + ixy[c] * ixy[c] + DATANORM;
      n2[c] = iyy[c] * iyy[c] + ixy[c] * ixy[c] + DATANORM;
      t1[c] = ixz[c] + ixx[c] * u + ixy[c] * v;
      t2[c] = iyz[c] + ixy[c] * u + iyy[c] * v;
    }
    const float tmp = m * hg3 /
                      sqrtf(t1[0] * t1[0] / n1[0] + t2[0] * t2[0] / n2[0] + t1[1] * t1[1] / n1[1] +
                            t2[1] * t2[1] / n2[1] + t1[2] * t1[2] / n1[2] + t2[2] * t2[2] / n2[2] + EPS_GRAD);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float w1 = tmp / n1[c], w2 = tmp / n2[c];
      a11 += w1 * ixx[c] * ixx[c] + w2 * ixy[c] * ixy[c];
      a12 += w1 * ixx[c] * ixy[c] + w2 * ixy[c] * iyy[c];
      a22 += w2 * iyy[c] * iyy[c] + w1 * ixy[c] * ixy[c];
      b1 -= w1 * ixx[c] * ixz[c] + w2 * ixy[c] * iyz[c];
      b2 -= w2 * iyy[c] * iyz[c] + w1 * ixy[c] * ixz[c];
    }
  }
}

}  // namespace OFDIS_KNS
}  // namespace ofdis
