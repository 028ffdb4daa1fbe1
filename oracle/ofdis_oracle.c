/* ofdis_oracle.c -- plain-C restatement of the OF_DIS hot path.  TEST INFRASTRUCTURE ONLY
 * (see ofdis_oracle.h: never linked into or called by the product).
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * The arithmetic is written so that, compiled with -ffp-contract=off on SSE (no FMA), each fp32
 * operation happens in the same order as in the reference's SSE path; tests/test_oracle_vs_ref.py
 * checks bit-equality against the reference sources compiled in place (oracle/_ref).
 *
 * Not restated (out of scope): stereo mode (SELECTMODE==2), forward-backward
 * merging (usefbcon), the dead DeepFlow code in FDF1.0.1.
 */
#include "ofdis_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int g_wave64 = 0;
void oracle_set_reduce_order(int wave64) { g_wave64 = wave64 ? 1 : 0; }
int oracle_get_reduce_order(void) { return g_wave64; }

/* Eigen's .sum() (patch.cpp:74-76,178-179,278,331,401): order unspecified by Eigen; see
 * oracle/eigen_shim/Eigen/Core for the two orders. */
static float reduce_sum(const float* x, int n) {
  if (g_wave64 && n <= 64) { /* 8 stride-8 partials, then distance 4, 1, 2 (see eigen_shim/Eigen/Core) */
    float p[8];
    for (int l = 0; l < 8; ++l) {
      float s = 0.0f;
      int any = 0;
      for (int k = l; k < n; k += 8) {
        s = any ? s + x[k] : x[k];
        any = 1;
      }
      p[l] = any ? s : 0.0f;
    }
    const float q0 = p[0] + p[4], q1 = p[1] + p[5], q2 = p[2] + p[6], q3 = p[3] + p[7];
    return (q0 + q1) + (q2 + q3);
  }
  if (g_wave64) {
    float part[64];
    for (int l = 0; l < 64; ++l) {
      float s = 0.0f;
      int any = 0;
      for (int k = l; k < n; k += 64) {
        s = any ? s + x[k] : x[k];
        any = 1;
      }
      part[l] = any ? s : 0.0f;
    }
    for (int o = 1; o < 64; o <<= 1)
      for (int i = 0; i < 64; i += 2 * o) part[i] = part[i] + part[i + o];
    return part[0];
  }
  float s = 0.0f;
  for (int i = 0; i < n; ++i) s += x[i];
  return s;
}

static float* falloc(size_t n) {
  float* p = (float*)calloc(n ? n : 1, sizeof(float));
  if (!p) {
    fprintf(stderr, "ofdis_oracle: out of memory\n");
    exit(1);
  }
  return p;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ======================================================================= FDF1.0.1 kernels */

/* opticalflow_aux.c:18-60 image_warp; MINMAX_TA image.h:6-8 */
void oracle_image_warp(float* dst, float* mask, const float* src, const float* wx, const float* wy, int w, int h,
                       int noc) {
  for (int j = 0; j < h; ++j)
    for (int i = 0; i < w; ++i) {
      const int o = j * w + i;
      const float xx = i + wx[o];
      const float yy = j + wy[o];
      const int x = (int)floor(xx);
      const int y = (int)floor(yy);
      const float dx = xx - x;
      const float dy = yy - y;
      mask[o] = (xx >= 0 && xx <= w - 1 && yy >= 0 && yy <= h - 1);
      const int x1 = clampi(x, 0, w - 1), x2 = clampi(x + 1, 0, w - 1);
      const int y1 = clampi(y, 0, h - 1), y2 = clampi(y + 1, 0, h - 1);
      for (int c = 0; c < noc; ++c) {
        const float* s = src + (size_t)c * w * h;
        dst[(size_t)c * w * h + o] = s[y1 * w + x1] * (1.0f - dx) * (1.0f - dy) + s[y1 * w + x2] * dx * (1.0f - dy) +
                                     s[y2 * w + x1] * (1.0f - dx) * dy + s[y2 * w + x2] * dx * dy;
      }
    }
}

/* image.c:327-350 convolve_extract_coeffs(even=0) for order 2 ({0,-8/12,1/12}, refine_variational.cpp:45)
 * and order 1 ({0,-0.5}, refine_variational.cpp:47) */
static void deriv5_coeffs(float c[5]) {
  const float half[3] = {0.0f, -8.0f / 12.0f, 1.0f / 12.0f};
  for (int i = 0; i <= 2; ++i) {
    c[2 - i] = +half[i];
    c[2 + i] = -half[i];
  }
}
static void deriv3_coeffs(float c[3]) {
  const float half[2] = {0.0f, -0.5f};
  for (int i = 0; i <= 1; ++i) {
    c[1 - i] = +half[i];
    c[1 + i] = -half[i];
  }
}

/* image.c:466-502 convolve_horiz_fast_5 (replicated borders via shifted row copies) */
static void conv_h5(float* dst, const float* src, int w, int h, const float c[5]) {
  for (int j = 0; j < h; ++j) {
    const float* s = src + (size_t)j * w;
    float* d = dst + (size_t)j * w;
    for (int i = 0; i < w; ++i)
      d[i] = c[0] * s[clampi(i - 2, 0, w - 1)] + c[1] * s[clampi(i - 1, 0, w - 1)] + c[2] * s[i] +
             c[3] * s[clampi(i + 1, 0, w - 1)] + c[4] * s[clampi(i + 2, 0, w - 1)];
  }
}
/* image.c:401-434 convolve_vert_fast_5 (folded coefficients on the first/last two rows) */
static void conv_v5(float* dst, const float* src, int w, int h, const float c[5]) {
  const float c012 = c[0] + c[1] + c[2], c01 = c[0] + c[1], c34 = c[3] + c[4], c234 = c[2] + c[3] + c[4];
  for (int j = 0; j < h; ++j) {
    float* d = dst + (size_t)j * w;
    const float* s0 = src + (size_t)j * w;
    for (int i = 0; i < w; ++i) {
      if (j == 0)
        d[i] = c012 * s0[i] + c[3] * s0[i + w] + c[4] * s0[i + 2 * w];
      else if (j == 1)
        d[i] = c01 * s0[i - w] + c[2] * s0[i] + c[3] * s0[i + w] + c[4] * s0[i + 2 * w];
      else if (j == h - 2)
        d[i] = c[0] * s0[i - 2 * w] + c[1] * s0[i - w] + c[2] * s0[i] + c34 * s0[i + w];
      else if (j == h - 1)
        d[i] = c[0] * s0[i - 2 * w] + c[1] * s0[i - w] + c234 * s0[i];
      else
        d[i] = c[0] * s0[i - 2 * w] + c[1] * s0[i - w] + c[2] * s0[i] + c[3] * s0[i + w] + c[4] * s0[i + 2 * w];
    }
  }
}
/* image.c:436-464 convolve_horiz_fast_3 */
static void conv_h3(float* dst, const float* src, int w, int h, const float c[3]) {
  for (int j = 0; j < h; ++j) {
    const float* s = src + (size_t)j * w;
    float* d = dst + (size_t)j * w;
    for (int i = 0; i < w; ++i) d[i] = c[0] * s[clampi(i - 1, 0, w - 1)] + c[1] * s[i] + c[2] * s[clampi(i + 1, 0, w - 1)];
  }
}
/* image.c:376-399 convolve_vert_fast_3 */
static void conv_v3(float* dst, const float* src, int w, int h, const float c[3]) {
  const float c01 = c[0] + c[1], c12 = c[1] + c[2];
  for (int j = 0; j < h; ++j) {
    float* d = dst + (size_t)j * w;
    const float* s0 = src + (size_t)j * w;
    for (int i = 0; i < w; ++i) {
      if (j == 0)
        d[i] = c01 * s0[i] + c[2] * s0[i + w];
      else if (j == h - 1)
        d[i] = c[0] * s0[i - w] + c12 * s0[i];
      else
        d[i] = c[0] * s0[i - w] + c[1] * s0[i] + c[2] * s0[i + w];
    }
  }
}

/* opticalflow_aux.c:65-116 get_derivatives.  out: 8 groups of noc planes Ix,Iy,Iz,Ixx,Ixy,Iyy,Ixz,Iyz */
void oracle_get_derivatives(const float* im1, const float* im2, float* out, int w, int h, int noc) {
  float c[5];
  deriv5_coeffs(c);
  const size_t n = (size_t)w * h;
  float* avg = falloc(n);
  for (int ch = 0; ch < noc; ++ch) {
    const float* a = im1 + ch * n;
    const float* b = im2 + ch * n;
    float* Ix = out + (0 * noc + ch) * n;
    float* Iy = out + (1 * noc + ch) * n;
    float* Iz = out + (2 * noc + ch) * n;
    float* Ixx = out + (3 * noc + ch) * n;
    float* Ixy = out + (4 * noc + ch) * n;
    float* Iyy = out + (5 * noc + ch) * n;
    float* Ixz = out + (6 * noc + ch) * n;
    float* Iyz = out + (7 * noc + ch) * n;
    for (size_t i = 0; i < n; ++i) {
      avg[i] = 0.5f * (b[i] + a[i]); /* opticalflow_aux.c:81 */
      Iz[i] = b[i] - a[i];           /* :82 */
    }
    conv_h5(Ix, avg, w, h, c);  /* :86 */
    conv_v5(Iy, avg, w, h, c);  /* :87 */
    conv_h5(Ixx, Ix, w, h, c);  /* :88 */
    conv_v5(Ixy, Ix, w, h, c);  /* :89 */
    conv_v5(Iyy, Iy, w, h, c);  /* :90 */
    conv_h5(Ixz, Iz, w, h, c);  /* :91 */
    conv_v5(Iyz, Iz, w, h, c);  /* :92 */
  }
  free(avg);
}

#define EPS_SMOOTH (0.001f * 0.001f) /* opticalflow_aux.c:14 */
#define EPS_COLOR (0.001f * 0.001f)  /* :11 */
#define EPS_GRAD (0.001f * 0.001f)   /* :12 */

/* opticalflow_aux.c:123-165 compute_smoothness */
void oracle_compute_smoothness(float* sh, float* sv, const float* uu, const float* vv, float quarter_alpha, int w,
                               int h) {
  float c[3];
  deriv3_coeffs(c);
  const size_t n = (size_t)w * h;
  float *ux = falloc(n), *vx = falloc(n), *uy = falloc(n), *vy = falloc(n), *s = falloc(n);
  conv_h3(ux, uu, w, h, c);
  conv_h3(vx, vv, w, h, c);
  conv_v3(uy, uu, w, h, c);
  conv_v3(vy, vv, w, h, c);
  for (size_t i = 0; i < n; ++i) /* :138 */
    s[i] = quarter_alpha / sqrtf(ux[i] * ux[i] + uy[i] * uy[i] + vx[i] * vx[i] + vy[i] * vy[i] + EPS_SMOOTH);
  for (int j = 0; j < h; ++j) {
    for (int i = 0; i < w - 1; ++i) sh[j * w + i] = s[j * w + i] + s[j * w + i + 1]; /* :150 */
    sh[j * w + w - 1] = 0.0f;                                                         /* :154 */
  }
  for (int j = 0; j < h - 1; ++j)
    for (int i = 0; i < w; ++i) sv[j * w + i] = s[j * w + i] + s[(j + 1) * w + i]; /* :160 */
  for (int i = 0; i < w; ++i) sv[(h - 1) * w + i] = 0.0f;                          /* :163 */
  free(ux); free(vx); free(uy); free(vy); free(s);
}

/* opticalflow_aux.c:310-438 compute_data.  derivs: 8 groups of noc planes; out5: a11,a12,a22,b1,b2 */
void oracle_compute_data(float* out5, const float* mask, const float* du, const float* dv, const float* derivs,
                         float half_delta_over3, float half_gamma_over3, int w, int h, int noc) {
  const size_t n = (size_t)w * h;
  const float dnorm = 0.1f * 0.1f; /* :10, used unparenthesised inside a brace initialiser :315 */
  float *A11 = out5, *A12 = out5 + n, *A22 = out5 + 2 * n, *B1 = out5 + 3 * n, *B2 = out5 + 4 * n;
#define D(k, c) (derivs[((size_t)(k) * noc + (c)) * n + i])
  for (size_t i = 0; i < n; ++i) {
    float a11 = 0.0f, a12 = 0.0f, a22 = 0.0f, b1 = 0.0f, b2 = 0.0f; /* memset :335-339 */
    const float u = du[i], v = dv[i], m = mask[i];
    if (noc == 1) {
      const float ix = D(0, 0), iy = D(1, 0), iz = D(2, 0), ixx = D(3, 0), ixy = D(4, 0), iyy = D(5, 0),
                  ixz = D(6, 0), iyz = D(7, 0);
      float tmp, tmp2, n1, n2;
      if (half_delta_over3) { /* :347-366 */
        tmp = iz + ix * u + iy * v;
        n1 = ix * ix + iy * iy + dnorm;
        tmp = m * half_delta_over3 / sqrtf(3 * tmp * tmp / n1 + EPS_COLOR);
        tmp /= n1;
        a11 += tmp * ix * ix;
        a12 += tmp * ix * iy;
        a22 += tmp * iy * iy;
        b1 -= tmp * iz * ix;
        b2 -= tmp * iz * iy;
      }
      /* :381-405 */
      n1 = ixx * ixx + ixy * ixy + dnorm;
      n2 = iyy * iyy + ixy * ixy + dnorm;
      tmp = ixz + ixx * u + ixy * v;
      tmp2 = iyz + ixy * u + iyy * v;
      tmp = m * half_gamma_over3 / sqrtf(3 * tmp * tmp / n1 + 3 * tmp2 * tmp2 / n2 + EPS_GRAD);
      tmp2 = tmp / n2;
      tmp /= n1;
      a11 += tmp * ixx * ixx + tmp2 * ixy * ixy;
      a12 += tmp * ixx * ixy + tmp2 * ixy * iyy;
      a22 += tmp2 * iyy * iyy + tmp * ixy * ixy;
      b1 -= tmp * ixx * ixz + tmp2 * ixy * iyz;
      b2 -= tmp2 * iyy * iyz + tmp * ixy * ixz;
      /* :420-427 */
      a11 *= 3; a12 *= 3; a22 *= 3; b1 *= 3; b2 *= 3;
    } else {
      float ix[3], iy[3], iz[3], ixx[3], ixy[3], iyy[3], ixz[3], iyz[3];
      for (int c = 0; c < 3; ++c) {
        ix[c] = D(0, c); iy[c] = D(1, c); iz[c] = D(2, c); ixx[c] = D(3, c);
        ixy[c] = D(4, c); iyy[c] = D(5, c); ixz[c] = D(6, c); iyz[c] = D(7, c);
      }
      if (half_delta_over3) { /* :347-377 */
        float t[3], nn[3];
        for (int c = 0; c < 3; ++c) {
          t[c] = iz[c] + ix[c] * u + iy[c] * v;
          nn[c] = ix[c] * ix[c] + iy[c] * iy[c] + dnorm;
        }
        float tmp = m * half_delta_over3 /
                    sqrtf(t[0] * t[0] / nn[0] + t[1] * t[1] / nn[1] + t[2] * t[2] / nn[2] + EPS_COLOR);
        float tmp3 = tmp / nn[2], tmp2 = tmp / nn[1];
        tmp /= nn[0];
        const float tt[3] = {tmp, tmp2, tmp3};
        for (int c = 0; c < 3; ++c) {
          a11 += tt[c] * ix[c] * ix[c];
          a12 += tt[c] * ix[c] * iy[c];
          a22 += tt[c] * iy[c] * iy[c];
          b1 -= tt[c] * iz[c] * ix[c];
          b2 -= tt[c] * iz[c] * iy[c];
        }
      }
      /* :381-418 */
      float n1[3], n2[3], t1[3], t2[3];
      for (int c = 0; c < 3; ++c) {
        n1[c] = ixx[c] * ixx[c] + ixy[c] * ixy[c] + dnorm;
        n2[c] = iyy[c] * iyy[c] + ixy[c] * ixy[c] + dnorm;
        t1[c] = ixz[c] + ixx[c] * u + ixy[c] * v;
        t2[c] = iyz[c] + ixy[c] * u + iyy[c] * v;
      }
      float tmp = m * half_gamma_over3 /
                  sqrtf(t1[0] * t1[0] / n1[0] + t2[0] * t2[0] / n2[0] + t1[1] * t1[1] / n1[1] + t2[1] * t2[1] / n2[1] +
                        t1[2] * t1[2] / n1[2] + t2[2] * t2[2] / n2[2] + EPS_GRAD);
      float w1[3], w2[3];
      w2[2] = tmp / n2[2]; w1[2] = tmp / n1[2]; w2[1] = tmp / n2[1]; w1[1] = tmp / n1[1]; w2[0] = tmp / n2[0];
      w1[0] = tmp / n1[0];
      for (int c = 0; c < 3; ++c) {
        a11 += w1[c] * ixx[c] * ixx[c] + w2[c] * ixy[c] * ixy[c];
        a12 += w1[c] * ixx[c] * ixy[c] + w2[c] * ixy[c] * iyy[c];
        a22 += w2[c] * iyy[c] * iyy[c] + w1[c] * ixy[c] * ixy[c];
        b1 -= w1[c] * ixx[c] * ixz[c] + w2[c] * ixy[c] * iyz[c];
        b2 -= w2[c] * iyy[c] * iyz[c] + w1[c] * ixy[c] * ixz[c];
      }
    }
    A11[i] = a11; A12[i] = a12; A22[i] = a22; B1[i] = b1; B2[i] = b2;
  }
#undef D
}

/* opticalflow_aux.c:172-199 sub_laplacian (scatter form; per pixel the four updates arrive in the
 * order -left, +right, -top, +bottom) */
void oracle_sub_laplacian(float* dst, const float* src, const float* wh, const float* wv, int w, int h) {
  for (int j = 0; j < h; ++j)
    for (int i = 0; i < w - 1; ++i) {
      const int o = j * w + i;
      const float tmp = wh[o] * (src[o + 1] - src[o]);
      dst[o] += tmp;
      dst[o + 1] -= tmp;
    }
  for (int j = 0; j < h - 1; ++j)
    for (int i = 0; i < w; ++i) {
      const int o = j * w + i;
      const float tmp = wv[o] * (src[o + w] - src[o]);
      dst[o] += tmp;
      dst[o + w] -= tmp;
    }
}

/* solver.c:19-72 sor_coupled_slow_but_readable (single thread) */
void oracle_sor_coupled_slow(float* du, float* dv, const float* a11, const float* a12, const float* a22,
                             const float* b1, const float* b2, const float* sh, const float* sv, int iterations,
                             float omega, int w, int h) {
  for (int iter = 0; iter < iterations; ++iter)
    for (int j = 0; j < h; ++j)
      for (int i = 0; i < w; ++i) {
        const int o = j * w + i;
        float sigma_u = 0.0f, sigma_v = 0.0f, sum_dpsis = 0.0f;
        if (j > 0) {
          sigma_u -= sv[o - w] * du[o - w];
          sigma_v -= sv[o - w] * dv[o - w];
          sum_dpsis += sv[o - w];
        }
        if (i > 0) {
          sigma_u -= sh[o - 1] * du[o - 1];
          sigma_v -= sh[o - 1] * dv[o - 1];
          sum_dpsis += sh[o - 1];
        }
        if (j < h - 1) {
          sigma_u -= sv[o] * du[o + w];
          sigma_v -= sv[o] * dv[o + w];
          sum_dpsis += sv[o];
        }
        if (i < w - 1) {
          sigma_u -= sh[o] * du[o + 1];
          sigma_v -= sh[o] * dv[o + 1];
          sum_dpsis += sh[o];
        }
        const float A11 = a11[o] + sum_dpsis, A12 = a12[o], A22 = a22[o] + sum_dpsis;
        const float B1 = b1[o] - sigma_u, B2 = b2[o] - sigma_v;
        du[o] = (1.0f - omega) * du[o] + omega / A11 * (B1 - A12 * dv[o]);
        dv[o] = (1.0f - omega) * dv[o] + omega / A22 * (B2 - A12 * du[o]);
      }
}

/* solver.c:77-421 sor_coupled: lexicographic block Gauss-Seidel/SOR; the first sweep replaces
 * a11,a12,a22 by the inverse of the per-pixel 2x2 block (note the swap A11=a22+D, A22=a11+D).
 * The three row classes of the reference only differ by which vertical terms are present; they are
 * written out because "x + 0*y" is not bit-identical to "x" for the absent terms in general. */
void oracle_sor_coupled(float* du, float* dv, float* a11, float* a12, float* a22, const float* b1,
                        const float* b2, const float* sh, const float* sv, int iterations, float omega, int w,
                        int h) {
  if (w < 2 || h < 2 || iterations < 1) { /* solver.c:80-83 */
    oracle_sor_coupled_slow(du, dv, a11, a12, a22, b1, b2, sh, sv, iterations, omega, w, h);
    return;
  }
  for (int iter = 0; iter < iterations; ++iter)
    for (int j = 0; j < h; ++j)
      for (int i = 0; i < w; ++i) {
        const int o = j * w + i;
        const float hl = (i > 0) ? sh[o - 1] : 0.0f; /* f1: shifted copy with f1[0]=0, solver.c:95,104 */
        const float hr = sh[o];
        const float dur = (i < w - 1) ? du[o + 1] : 0.0f; /* f2/f3 zero beyond width-1, solver.c:98-99 */
        const float dvr = (i < w - 1) ? dv[o + 1] : 0.0f;
        if (iter == 0) { /* solver.c:112-120 etc. */
          float dpsis;
          if (j == 0)
            dpsis = hl + hr + sv[o];
          else if (j == h - 1)
            dpsis = hl + hr + sv[o - w];
          else
            dpsis = hl + hr + sv[o - w] + sv[o];
          const float A11 = a22[o] + dpsis, A22 = a11[o] + dpsis;
          const float det = A11 * A22 - a12[o] * a12[o];
          a11[o] = A11 / det;
          a22[o] = A22 / det;
          a12[o] /= -det;
        }
        float s1, s2;
        if (j == 0) { /* solver.c:122-123 */
          s1 = hr * dur + sv[o] * du[o + w] + b1[o];
          s2 = hr * dvr + sv[o] * dv[o + w] + b2[o];
        } else if (j == h - 1) { /* solver.c:226-227 */
          s1 = hr * dur + sv[o - w] * du[o - w] + b1[o];
          s2 = hr * dvr + sv[o - w] * dv[o - w] + b2[o];
        } else { /* solver.c:174-175 */
          s1 = hr * dur + sv[o - w] * du[o - w] + sv[o] * du[o + w] + b1[o];
          s2 = hr * dvr + sv[o - w] * dv[o - w] + sv[o] * dv[o + w] + b2[o];
        }
        float B1, B2;
        if (i == 0) { /* left block, k==0: no left neighbour term at all, solver.c:124-125 */
          B1 = s1;
          B2 = s2;
        } else { /* solver.c:127-128 */
          B1 = hl * du[o - 1] + s1;
          B2 = hl * dv[o - 1] + s2;
        }
        const float u0 = du[o], v0 = dv[o];
        du[o] = u0 + omega * (a11[o] * B1 + a12[o] * B2 - u0); /* solver.c:129 */
        dv[o] = v0 + omega * (a12[o] * B1 + a22[o] * B2 - v0); /* solver.c:130 */
      }
}

/* ======================================================================= geometry helpers */

typedef struct {
  int w, h, pad, tmp_w, tmp_h, level;
  float lb, ubw, ubh;
} lvgeom;

/* oflow.cpp:138-157 */
static lvgeom level_geom(const ofdis_params* p, int sl) {
  lvgeom g;
  const float sc_fct = (float)pow(2, -sl);
  g.h = (int)(p->height * sc_fct);
  g.w = (int)(p->width * sc_fct);
  g.pad = p->imgpadding;
  g.lb = -(float)p->p_samp_s / 2;
  g.ubw = (float)(g.w + p->p_samp_s / 2 - 2);
  g.ubh = (float)(g.h + p->p_samp_s / 2 - 2);
  g.tmp_w = g.w + 2 * g.pad;
  g.tmp_h = g.h + 2 * g.pad;
  g.level = sl;
  return g;
}

size_t oracle_plane_elems(const ofdis_params* p, int level) {
  lvgeom g = level_geom(p, level);
  return (size_t)g.tmp_w * g.tmp_h * p->noc;
}

/* ======================================================================= VarRefClass */

/* refine_variational.cpp:25-116 (ctor), 120-149 (copyimage), 153-241 (RefLevelOF) */
int oracle_varref_level(const ofdis_params* p, int level, const float* im_a, const float* im_b, float* flow) {
  const lvgeom g = level_geom(p, level);
  const int w = g.w, h = g.h, noc = p->noc;
  const size_t n = (size_t)w * h;
  if (h < 4) return OFDIS_ERR_INVALID; /* convolve_vert_fast_5 reads rows 0..3 unconditionally */
  const int n_inner = p->tv_innerit * (level + 1);          /* :36 */
  const float quarter_alpha = 0.25f * p->tv_alpha;          /* :40 */
  const float half_gamma_over3 = p->tv_gamma * 0.5f / 3.0f; /* :41 */
  const float half_delta_over3 = p->tv_delta * 0.5f / 3.0f; /* :42 */

  float *wx = falloc(n), *wy = falloc(n);
  for (size_t i = 0; i < n; ++i) { /* :56-68 */
    wx[i] = flow[2 * i];
    wy[i] = flow[2 * i + 1];
  }
  float *I1 = falloc(n * noc), *I2 = falloc(n * noc);
  for (int y = 0; y < h; ++y) /* copyimage :120-149: strip the padding, planarise */
    for (int x = 0; x < w; ++x)
      for (int c = 0; c < noc; ++c) {
        const size_t src = ((size_t)(y + g.pad) * g.tmp_w + (x + g.pad)) * noc + c;
        I1[(size_t)c * n + y * w + x] = im_a[src];
        I2[(size_t)c * n + y * w + x] = im_b[src];
      }
  float *du = falloc(n), *dv = falloc(n), *mask = falloc(n), *sh = falloc(n), *sv = falloc(n), *uu = falloc(n),
        *vv = falloc(n), *sys = falloc(5 * n), *w_im2 = falloc(n * noc), *derivs = falloc(8 * n * noc);
  oracle_image_warp(w_im2, mask, I2, wx, wy, w, h, noc);    /* :182 */
  oracle_get_derivatives(I1, w_im2, derivs, w, h, noc);     /* :184 */
  memcpy(uu, wx, n * sizeof(float));                        /* :189 */
  memcpy(vv, wy, n * sizeof(float));
  for (int it = 0; it < n_inner; ++it) {                    /* :192-218 */
    oracle_compute_smoothness(sh, sv, uu, vv, quarter_alpha, w, h);
    oracle_compute_data(sys, mask, du, dv, derivs, half_delta_over3, half_gamma_over3, w, h, noc);
    oracle_sub_laplacian(sys + 3 * n, wx, sh, sv, w, h);
    oracle_sub_laplacian(sys + 4 * n, wy, sh, sv, w, h);
    oracle_sor_coupled(du, dv, sys, sys + n, sys + 2 * n, sys + 3 * n, sys + 4 * n, sh, sv, p->tv_solverit,
                       p->tv_sor, w, h);
    for (size_t i = 0; i < n; ++i) {
      uu[i] = wx[i] + du[i];
      vv[i] = wy[i] + dv[i];
    }
  }
  for (size_t i = 0; i < n; ++i) { /* :220-221, 92-99 */
    flow[2 * i] = uu[i];
    flow[2 * i + 1] = vv[i];
  }
  free(wx); free(wy); free(I1); free(I2); free(du); free(dv); free(mask); free(sh); free(sv);
  free(uu); free(vv); free(sys); free(w_im2); free(derivs);
  return OFDIS_OK;
}

/* ======================================================================= PatClass / PatGridClass */

typedef struct {
  const ofdis_params* p;
  lvgeom g;
  int novals;
  float outlierthresh, dp_thresh_sq;
} patctx;

/* patch.cpp:287-332 getPatchStaticNNGrad */
static void patch_static_nn_grad(const patctx* c, const float* img, const float* img_dx, const float* img_dy,
                                 float mx, float my, float* T, float* Tx, float* Ty) {
  const int P = c->p->p_samp_s, noc = c->p->noc;
  const int px = (int)(round(mx) + c->g.pad), py = (int)(round(my) + c->g.pad);
  const int lb = -P / 2, ub = P / 2 - 1;
  int k = 0;
  for (int j = lb; j <= ub; ++j)
    for (int i = lb; i <= ub; ++i) {
      const int idx = ((px + i) + (py + j) * c->g.tmp_w) * noc;
      for (int ch = 0; ch < noc; ++ch, ++k) {
        T[k] = img[idx + ch];
        Tx[k] = img_dx[idx + ch];
        Ty[k] = img_dy[idx + ch];
      }
    }
  if (c->p->patnorm > 0) { /* :331 */
    const float m = reduce_sum(T, c->novals) / c->novals;
    for (k = 0; k < c->novals; ++k) T[k] -= m;
  }
}

/* patch.cpp:335-402 getPatchStaticBil */
static void patch_static_bil(const patctx* c, const float* img, float mx, float my, float* out) {
  const int P = c->p->p_samp_s, noc = c->p->noc;
  int pos0 = (int)ceil(mx + .00001f), pos1 = (int)ceil(my + .00001f);
  const int pos2 = (int)floor(mx), pos3 = (int)floor(my);
  const float r0 = mx - (float)pos2, r1 = my - (float)pos3;
  const float we0 = r0 * r1, we1 = (1 - r0) * r1, we2 = r0 * (1 - r1), we3 = (1 - r0) * (1 - r1);
  pos0 += c->g.pad;
  pos1 += c->g.pad;
  const int lb = -P / 2, ub = P / 2 - 1;
  const int tw = c->g.tmp_w;
  int k = 0;
  for (int y = pos1 + lb; y <= pos1 + ub; ++y)
    for (int x = pos0 + lb; x <= pos0 + ub; ++x)
      for (int ch = 0; ch < noc; ++ch, ++k) {
        const float a = img[(x + y * tw) * noc + ch], b = img[(x - 1 + y * tw) * noc + ch];
        const float cc = img[(x + (y - 1) * tw) * noc + ch], d = img[(x - 1 + (y - 1) * tw) * noc + ch];
        out[k] = we0 * a + we1 * b + we2 * cc + we3 * d; /* :391 */
      }
  if (c->p->patnorm > 0) { /* :401 */
    const float m = reduce_sum(out, c->novals) / c->novals;
    for (k = 0; k < c->novals; ++k) out[k] -= m;
  }
}

/* patch.cpp:223-262 LossComputeErrorImage */
static void loss_error_image(const patctx* c, float* pdiff, float* pweight, const float* T) {
  const int n = c->novals;
  const float bsq = 5.0f * 5.0f, bsq2 = bsq * 2.0f; /* normoutlier oflow.h:63, oflow.cpp:106-107 */
  for (int k = 0; k < n; ++k) {
    float d = pdiff[k] - T[k];
    if (c->p->costfct == 1)
      d = copysignf(sqrtf(fabsf(d)), d); /* :243 */
    else if (c->p->costfct == 2)
      d = copysignf(sqrtf((sqrtf(1.0f + (d * d) / bsq) - 1.0f) * bsq2), d); /* :252-258 */
    pdiff[k] = d;
    pweight[k] = fabsf(d);
  }
}

typedef struct {
  float p_in[2], p_iter[2], delta_p[2], pt_iter[2], pt_st[2];
  float dpsq, dpsq_init, mares, mares_old;
  int cnt, converged;
} pstate;

/* patch.cpp:264-284 OptimizeComputeErrImg */
static void compute_err_img(const patctx* c, const float* im_b, pstate* s, const float* T, float* pdiff,
                            float* pweight, float* absbuf) {
  patch_static_bil(c, im_b, s->pt_iter[0], s->pt_iter[1], pdiff);
  loss_error_image(c, pdiff, pweight, T);
  s->dpsq = s->delta_p[0] * s->delta_p[0] + s->delta_p[1] * s->delta_p[1];
  if (s->cnt == 1) s->dpsq_init = s->dpsq;
  s->mares_old = s->mares;
  for (int k = 0; k < c->novals; ++k) absbuf[k] = fabsf(pweight[k]);
  s->mares = reduce_sum(absbuf, c->novals) / c->novals;
  const ofdis_params* p = c->p;
  if (!((s->cnt < p->max_iter) & (s->mares > p->res_thresh) &
        ((s->cnt < p->min_iter) | (s->dpsq / s->dpsq_init >= c->dp_thresh_sq)) &
        ((s->cnt < p->min_iter) | (s->mares / s->mares_old <= p->dr_thresh))))
    s->converged = 1;
}

static int out_of_bounds(const patctx* c, const float* pt) { /* patch.cpp:135-136, 200-201 */
  return pt[0] < c->g.lb || pt[1] < c->g.lb || pt[0] > c->g.ubw || pt[1] > c->g.ubh;
}

/* 2x2 LLT solve as defined in oracle/eigen_shim/Eigen/Core (Eigen call site patch.cpp:184) */
static void llt_solve(float h00, float h10, float h11, float* b) {
  float l00 = h00, l10 = h10, l11 = h11;
  if (!(l00 <= 0.0f)) {
    l00 = sqrtf(l00);
    l10 = l10 / l00;
    const float x = l11 - l10 * l10;
    if (!(x <= 0.0f)) l11 = sqrtf(x);
  }
  const float y0 = b[0] / l00;
  const float y1 = (b[1] - l10 * y0) / l11;
  const float x1 = y1 / l11;
  const float x0 = (y0 - l10 * x1) / l00;
  b[0] = x0;
  b[1] = x1;
}

int oracle_patchgrid_level(const ofdis_params* p, int level, const float* im_a, const float* im_a_dx,
                           const float* im_a_dy, const float* im_b, const float* flow_prev, float* p_out,
                           float* pweight_out, float* flow_out, int* nopatches_out) {
  if (p->usefbcon) return OFDIS_ERR_UNSUPPORTED;
  patctx c;
  c.p = p;
  c.g = level_geom(p, level);
  c.novals = p->noc * p->p_samp_s * p->p_samp_s;
  c.outlierthresh = (float)p->p_samp_s / 2;      /* oflow.cpp:82 */
  c.dp_thresh_sq = p->dp_thresh * p->dp_thresh;  /* oflow.cpp:88 */
  const int P = p->p_samp_s, noc = p->noc, nv = c.novals;
  const int w = c.g.w, h = c.g.h;
  /* patchgrid.cpp:42-48; steps oflow.cpp:91 */
  int steps = (int)floor(P * (1 - p->patove));
  if (steps < 1) steps = 1;
  const int nopw = (int)ceil((float)w / (float)steps), noph = (int)ceil((float)h / (float)steps);
  const int offw = (int)floor((w - (nopw - 1) * steps) / 2), offh = (int)floor((h - (noph - 1) * steps) / 2);
  const int nop = nopw * noph;
  if (nopatches_out) *nopatches_out = nop;

  float* T = falloc(nv), *Tx = falloc(nv), *Ty = falloc(nv), *pdiff = falloc(nv), *prod = falloc(nv),
       *absbuf = falloc(nv);
  float* pw_all = falloc((size_t)nop * nv);
  float* p_all = falloc((size_t)nop * 2);

  for (int gx = 0; gx < nopw; ++gx)
    for (int gy = 0; gy < noph; ++gy) {
      const int ip = gx * noph + gy; /* patchgrid.cpp:66 */
      const float rx = (float)(gx * steps + offw), ry = (float)(gy * steps + offh);
      float* pweight = pw_all + (size_t)ip * nv; /* zero = defined value of the never-written case */
      /* InitializePatch patch.cpp:57-69 */
      patch_static_nn_grad(&c, im_a, im_a_dx, im_a_dy, rx, ry, T, Tx, Ty);
      /* ComputeHessian patch.cpp:71-88 */
      for (int k = 0; k < nv; ++k) prod[k] = Tx[k] * Tx[k];
      float H00 = reduce_sum(prod, nv);
      for (int k = 0; k < nv; ++k) prod[k] = Tx[k] * Ty[k];
      const float H01 = reduce_sum(prod, nv);
      for (int k = 0; k < nv; ++k) prod[k] = Ty[k] * Ty[k];
      float H11 = reduce_sum(prod, nv);
      if (H00 * H11 - H01 * H01 == 0) {
        H00 += 1e-10; /* float += double, patch.cpp:80-81 */
        H11 += 1e-10;
      }
      /* InitializeFromCoarserOF patchgrid.cpp:195-211 */
      float pin[2] = {0.0f, 0.0f};
      if (flow_prev) {
        const int x = (int)floor(rx / 2), y = (int)floor(ry / 2);
        const int i = y * (w / 2) + x;
        pin[0] = flow_prev[2 * i] * 2;
        pin[1] = flow_prev[2 * i + 1] * 2;
      }
      /* OptimizeIter patch.cpp:159-212 with OptimizeStart :120-156 */
      pstate s;
      memset(&s, 0, sizeof(s));
      s.dpsq = 1e-10f; s.dpsq_init = 1e-10f; s.mares = 1e20f; s.mares_old = 1e20f; /* ResetPatch :99-117 */
      s.p_in[0] = s.p_iter[0] = pin[0];
      s.p_in[1] = s.p_iter[1] = pin[1];
      s.pt_iter[0] = rx + s.p_iter[0];
      s.pt_iter[1] = ry + s.p_iter[1];
      s.pt_st[0] = s.pt_iter[0];
      s.pt_st[1] = s.pt_iter[1];
      if (out_of_bounds(&c, s.pt_iter)) {
        s.converged = 1; /* pdiff = tmp; pweight untouched */
      } else {
        s.cnt = 0; s.dpsq = 1e-10f; s.dpsq_init = 1e-10f; s.mares = 1e5f; s.mares_old = 1e20f; s.converged = 0;
        compute_err_img(&c, im_b, &s, T, pdiff, pweight, absbuf);
      }
      while (!s.converged) {
        s.cnt++;
        for (int k = 0; k < nv; ++k) prod[k] = Tx[k] * pdiff[k];
        s.delta_p[0] = reduce_sum(prod, nv);
        for (int k = 0; k < nv; ++k) prod[k] = Ty[k] * pdiff[k];
        s.delta_p[1] = reduce_sum(prod, nv);
        llt_solve(H00, H01, H11, s.delta_p);
        s.p_iter[0] -= s.delta_p[0];
        s.p_iter[1] -= s.delta_p[1];
        s.pt_iter[0] = rx + s.p_iter[0];
        s.pt_iter[1] = ry + s.p_iter[1];
        const float ex = s.pt_st[0] - s.pt_iter[0], ey = s.pt_st[1] - s.pt_iter[1];
        if (sqrtf(ex * ex + ey * ey) > c.outlierthresh || out_of_bounds(&c, s.pt_iter)) {
          s.p_iter[0] = s.p_in[0];
          s.p_iter[1] = s.p_in[1];
          s.pt_iter[0] = rx + s.p_iter[0];
          s.pt_iter[1] = ry + s.p_iter[1];
          s.converged = 1;
        }
        compute_err_img(&c, im_b, &s, T, pdiff, pweight, absbuf);
      }
      p_all[2 * ip] = s.p_iter[0];
      p_all[2 * ip + 1] = s.p_iter[1];
    }

  if (p_out) memcpy(p_out, p_all, sizeof(float) * 2 * nop);
  if (pweight_out) memcpy(pweight_out, pw_all, sizeof(float) * (size_t)nop * nv);

  if (flow_out) { /* AggregateFlowDense patchgrid.cpp:213-275, 377-397 */
    float* we = falloc((size_t)w * h);
    memset(flow_out, 0, sizeof(float) * 2 * (size_t)w * h);
    const int lb = -P / 2, ub = P / 2 - 1;
    const float minerrval = 2.0f; /* oflow.h:62 */
    for (int gx = 0; gx < nopw; ++gx)
      for (int gy = 0; gy < noph; ++gy) {
        const int ip = gx * noph + gy;
        const float* pweight = pw_all + (size_t)ip * nv;
        const float rx = (float)(gx * steps + offw), ry = (float)(gy * steps + offh);
        /* pweight is a RUNNING pointer in the reference: +1 per visited pixel (loop header,
         * patchgrid.cpp:242) and, for RGB, +2 more only when the pixel lies inside the image
         * (patchgrid.cpp:256-257).  For RGB patches that overlap the image border the weights are
         * therefore read from shifted entries; reproduced literally. */
        for (int y = lb; y <= ub; ++y)
          for (int x = lb; x <= ub; ++x, ++pweight) {
            const int yt = (int)(y + ry), xt = (int)(x + rx);
            if (xt >= 0 && yt >= 0 && xt < w && yt < h) {
              const int i = yt * w + xt;
              float absw;
              if (noc == 1) {
                absw = 1.0f / (float)(fmaxf(minerrval, *pweight));
              } else {
                absw = (float)(fmaxf(minerrval, *pweight)); ++pweight;
                absw += (float)(fmaxf(minerrval, *pweight)); ++pweight;
                absw += (float)(fmaxf(minerrval, *pweight));
                absw = 1.0f / absw;
              }
              we[i] += absw;
              flow_out[2 * i] += p_all[2 * ip] * absw;
              flow_out[2 * i + 1] += p_all[2 * ip + 1] * absw;
            }
          }
      }
    for (int i = 0; i < w * h; ++i)
      if (we[i] > 0) {
        flow_out[2 * i] /= we[i];
        flow_out[2 * i + 1] /= we[i];
      }
    free(we);
  }
  free(T); free(Tx); free(Ty); free(pdiff); free(prod); free(absbuf); free(pw_all); free(p_all);
  return OFDIS_OK;
}

/* ======================================================================= OFClass */

/* oflow.cpp:184-337 coarse-to-fine loop */
int oracle_flow(const ofdis_params* p, const float* const* im_a, const float* const* im_a_dx,
                const float* const* im_a_dy, const float* const* im_b, float* outflow, const float* initflow,
                float* level_flows) {
  if (p->usefbcon) return OFDIS_ERR_UNSUPPORTED;
  if (p->sc_l > p->sc_f || p->sc_l < 0) return OFDIS_ERR_INVALID;
  float* prev = NULL;
  size_t lf_off = 0;
  for (int sl = p->sc_f; sl >= p->sc_l; --sl) {
    const lvgeom g = level_geom(p, sl);
    float* cur = (sl == p->sc_l) ? outflow : falloc((size_t)g.w * g.h * 2);
    const float* init = (sl < p->sc_f) ? prev : initflow; /* oflow.cpp:209-220 */
    int rc = oracle_patchgrid_level(p, sl, im_a[sl], im_a_dx[sl], im_a_dy[sl], im_b[sl], init, NULL, NULL, cur, NULL);
    if (rc == OFDIS_OK && p->usetvref) rc = oracle_varref_level(p, sl, im_a[sl], im_b[sl], cur);
    if (rc != OFDIS_OK) {
      if (cur != outflow) free(cur);
      free(prev);
      return rc;
    }
    if (level_flows) {
      memcpy(level_flows + lf_off, cur, sizeof(float) * 2 * (size_t)g.w * g.h);
      lf_off += 2 * (size_t)g.w * g.h;
    }
    free(prev);
    prev = (cur == outflow) ? NULL : cur;
  }
  free(prev);
  return OFDIS_OK;
}

/* ======================================================================= run_dense.cpp host steps */

/* run_dense.cpp:298-305 */
void oracle_padded_size(int width_org, int height_org, int sc_f, int* width, int* height) {
  const int scfct = (int)pow(2, sc_f);
  int padw = 0, padh = 0;
  int div = width_org % scfct;
  if (div > 0) padw = scfct - div;
  div = height_org % scfct;
  if (div > 0) padh = scfct - div;
  *width = width_org + padw;
  *height = height_org + padh;
}

static int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
  }
  return i;
}

/* run_dense.cpp:130-178 ConstructImgPyramide, :298-311 padding, :326-327 convertTo.
 * OpenCV calls restated: copyMakeBorder(REPLICATE), resize(.5,.5,INTER_LINEAR) == 2x2 box mean,
 * Sobel(ksize 3, scale 1/8, BORDER_DEFAULT = reflect-101), copyMakeBorder(REPLICATE / CONSTANT 0).
 * For 8-bit input every intermediate is a dyadic rational representable in fp32, so any summation
 * order gives identical bits. */
void oracle_build_pyramid(const ofdis_params* p, const uint8_t* img_u8, int width_org, int height_org,
                          float* const* img, float* const* dx, float* const* dy) {
  const int noc = p->noc, W = p->width, H = p->height, pad = p->imgpadding;
  const int padw = W - width_org, padh = H - height_org;
  const int left = (int)floor((float)padw / 2.0f), top = (int)floor((float)padh / 2.0f);
  float* prev = falloc((size_t)W * H * noc);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const int sy = clampi(y - top, 0, height_org - 1), sx = clampi(x - left, 0, width_org - 1);
      for (int c = 0; c < noc; ++c) prev[((size_t)y * W + x) * noc + c] = (float)img_u8[((size_t)sy * width_org + sx) * noc + c];
    }
  int w = W, h = H;
  for (int l = 0; l <= p->sc_f; ++l) {
    if (l > 0) { /* cv::resize x0.5 */
      const int w2 = w / 2, h2 = h / 2;
      float* cur = falloc((size_t)w2 * h2 * noc);
      for (int y = 0; y < h2; ++y)
        for (int x = 0; x < w2; ++x)
          for (int c = 0; c < noc; ++c) {
            const float a = prev[((size_t)(2 * y) * w + 2 * x) * noc + c], b = prev[((size_t)(2 * y) * w + 2 * x + 1) * noc + c];
            const float cc = prev[((size_t)(2 * y + 1) * w + 2 * x) * noc + c], d = prev[((size_t)(2 * y + 1) * w + 2 * x + 1) * noc + c];
            cur[((size_t)y * w2 + x) * noc + c] = ((a + b) + (cc + d)) * 0.25f;
          }
      free(prev);
      prev = cur;
      w = w2;
      h = h2;
    }
    const int tw = w + 2 * pad, th = h + 2 * pad;
    if (img[l]) {
      for (int y = 0; y < th; ++y)
        for (int x = 0; x < tw; ++x) {
          const int sy = clampi(y - pad, 0, h - 1), sx = clampi(x - pad, 0, w - 1);
          for (int c = 0; c < noc; ++c) img[l][((size_t)y * tw + x) * noc + c] = prev[((size_t)sy * w + sx) * noc + c];
        }
    }
    if (dx[l] && dy[l]) {
      memset(dx[l], 0, sizeof(float) * (size_t)tw * th * noc);
      memset(dy[l], 0, sizeof(float) * (size_t)tw * th * noc);
      for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
          for (int c = 0; c < noc; ++c) {
            float v[3][3];
            for (int j = -1; j <= 1; ++j)
              for (int i = -1; i <= 1; ++i)
                v[j + 1][i + 1] = prev[((size_t)reflect101(y + j, h) * w + reflect101(x + i, w)) * noc + c];
            const float gx = (v[0][2] - v[0][0]) + 2.0f * (v[1][2] - v[1][0]) + (v[2][2] - v[2][0]);
            const float gy = (v[2][0] - v[0][0]) + 2.0f * (v[2][1] - v[0][1]) + (v[2][2] - v[0][2]);
            dx[l][((size_t)(y + pad) * tw + x + pad) * noc + c] = gx * 0.125f;
            dy[l][((size_t)(y + pad) * tw + x + pad) * noc + c] = gy * 0.125f;
          }
    }
  }
  free(prev);
}

/* run_dense.cpp:406-414: flowout *= 2^sc_l; cv::resize(x 2^sc_l, INTER_LINEAR); crop.
 * cv::resize bilinear for CV_32FC2 restated from OpenCV's documented rule (resize.cpp, INTER_LINEAR: source
 * coordinate (d + 0.5) * scale - 0.5 in double, cast to float, floor, fraction; index clamped with the fraction
 * forced to 0 at both borders; horizontal pass S[sx]*(1-fx) + S[sx+1]*fx, then the vertical pass on two such rows).
 * OpenCV itself is not available here; the restatement is pinned by tests/test_upsample_pin.py instead: exact
 * hand vectors derived from that rule with rational arithmetic (bit-for-bit, all four borders and odd crops) and
 * torch's OpenCV-compatible align_corners=False bilinear to a few ulp. */
void oracle_upsample_crop(const ofdis_params* p, const float* flow, int width_org, int height_org, float* out) {
  const int s = 1 << p->sc_l;
  const int sw = p->width >> p->sc_l, sh = p->height >> p->sc_l;
  const int W = p->width, H = p->height;
  const int padw = W - width_org, padh = H - height_org;
  const int left = (int)floor((float)padw / 2.0f), top = (int)floor((float)padh / 2.0f);
  const float scf = (float)s;
  const double inv = 1.0 / (double)s;
  for (int y = 0; y < height_org; ++y) {
    const int Y = y + top;
    float fy = (float)((Y + 0.5) * inv - 0.5);
    int sy = (int)floor(fy);
    fy -= sy;
    if (sy < 0) { sy = 0; fy = 0; }
    if (sy >= sh - 1) { sy = sh - 1; fy = 0; }
    const int sy1 = (sy + 1 < sh) ? sy + 1 : sy;
    for (int x = 0; x < width_org; ++x) {
      const int X = x + left;
      float fx = (float)((X + 0.5) * inv - 0.5);
      int sx = (int)floor(fx);
      fx -= sx;
      if (sx < 0) { sx = 0; fx = 0; }
      if (sx >= sw - 1) { sx = sw - 1; fx = 0; }
      const int sx1 = (sx + 1 < sw) ? sx + 1 : sx;
      for (int c = 0; c < 2; ++c) {
        const float v00 = (s > 1 ? flow[2 * (sy * sw + sx) + c] * scf : flow[2 * (sy * sw + sx) + c]);
        const float v01 = (s > 1 ? flow[2 * (sy * sw + sx1) + c] * scf : flow[2 * (sy * sw + sx1) + c]);
        const float v10 = (s > 1 ? flow[2 * (sy1 * sw + sx) + c] * scf : flow[2 * (sy1 * sw + sx) + c]);
        const float v11 = (s > 1 ? flow[2 * (sy1 * sw + sx1) + c] * scf : flow[2 * (sy1 * sw + sx1) + c]);
        const float r0 = v00 * (1.0f - fx) + v01 * fx;
        const float r1 = v10 * (1.0f - fx) + v11 * fx;
        out[2 * ((size_t)y * width_org + x) + c] = r0 * (1.0f - fy) + r1 * fy;
      }
    }
  }
}
