// ofdis_de.hip -- the variational refinement of the stereo-depth mode (the reference's run_DE_* binaries,
// compile-time SELECTMODE=2): VarRefClass::RefLevelDE (refine_variational.cpp:245-336) = image_warp with a zero
// vertical flow, get_derivatives, then per fixed-point iteration
//     compute_smoothness(uu, 0)  +  compute_data_DE  +  sub_laplacian(b1, wx)     (opticalflow_aux.c:123-199,446-548)
//     sor_coupled_slow_but_readable_DE                                            (solver.c:428-466)
//     uu = min|max(wx + du, 0)   by camera side                                   (refine_variational.cpp:299-316)
// One unknown per pixel.  This mode is outside the benchmarked path: the kernels favour plainness over speed
// (per-pixel gathers, one launch per solver sweep); the solver keeps the reference's lexicographic Gauss-Seidel
// order with the same anti-diagonal wavefront as ofdis_sor.hip.
#include "ofdis_kernels.h"
#include "ofdis_tvmath.h"

namespace ofdis {
namespace OFDIS_KNS {  // the arithmetic contract this file is being compiled for (ofdis_dev.h)

// compute_smoothness at one pixel with vv == 0 (its derivative terms are exact zeros): replicate borders
// horizontally (image.c:436-464), folded coefficients on the first / last row (image.c:376-399)
__device__ __forceinline__ float de_smooth_at(const float* __restrict__ uu, int x, int y, int w, int h, float qa) {
  const float* r = uu + (size_t)y * w;
  const float uc = r[x], ul = r[x > 0 ? x - 1 : 0], ur = r[x < w - 1 ? x + 1 : w - 1];
  const float ux = D3_C0 * ul + D3_C1 * uc + D3_C2 * ur;
  float uy;
  if (y == 0) uy = (D3_C0 + D3_C1) * uc + D3_C2 * r[w + x];
  else if (y == h - 1) uy = D3_C0 * r[x - w] + (D3_C1 + D3_C2) * uc;
  else uy = D3_C0 * r[x - w] + D3_C1 * uc + D3_C2 * r[w + x];
  return qa / sqrtf(ux * ux + uy * uy + EPS_SMOOTH);
}

// compute_data_DE for one pixel (opticalflow_aux.c:446-548).  D(k,c): derivative plane k, channel c.
template <typename DF>
__device__ __forceinline__ void data_term_de(DF D, int noc, float m, float u, float hd3, float hg3, float& a11,
                                             float& b1) {
  a11 = 0.0f; b1 = 0.0f;
  if (noc == 1) {
    const float ix = D(0, 0), iy = D(1, 0), iz = D(2, 0), ixx = D(3, 0), ixy = D(4, 0), iyy = D(5, 0), ixz = D(6, 0),
                iyz = D(7, 0);
    float tmp, tmp2, n1, n2;
    if (hd3 != 0.0f) {
      tmp = iz + ix * u;
      n1 = ix * ix + iy * iy + DATANORM;
      tmp = m * hd3 / sqrtf(3 * tmp * tmp / n1 + EPS_COLOR);
      tmp /= n1;
      a11 += tmp * ix * ix;
      b1 -= tmp * iz * ix;
    }
    n1 = ixx * ixx + ixy * ixy + DATANORM;
    n2 = iyy * iyy + ixy * ixy + DATANORM;
    tmp = ixz + ixx * u;
    tmp2 = iyz + ixy * u;
    tmp = m * hg3 / sqrtf(3 * tmp * tmp / n1 + 3 * tmp2 * tmp2 / n2 + EPS_GRAD);
    tmp2 = tmp / n2;
    tmp /= n1;
    a11 += tmp * ixx * ixx + tmp2 * ixy * ixy;
    b1 -= tmp * ixx * ixz + tmp2 * ixy * iyz;
    a11 *= 3;
    b1 *= 3;
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
        t[c] = iz[c] + ix[c] * u;
        nn[c] = ix[c] * ix[c] + iy[c] * iy[c] + DATANORM;
      }
      const float tmp = m * hd3 / sqrtf(t[0] * t[0] / nn[0] + t[1] * t[1] / nn[1] + t[2] * t[2] / nn[2] + EPS_COLOR);
      const float t3 = tmp / nn[2], t2 = tmp / nn[1], t1 = tmp / nn[0];  // :479
      a11 += t1 * ix[0] * ix[0];
      b1 -= t1 * iz[0] * ix[0];
      a11 += t2 * ix[1] * ix[1];
      b1 -= t2 * iz[1] * ix[1];
      a11 += t3 * ix[2] * ix[2];
      b1 -= t3 * iz[2] * ix[2];
    }
    float n1[3], n2[3], t1[3], t2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      n1[c] = ixx[c] * ixx[c] + ixy[c] * ixy[c] + DATANORM;
      n2[c] = iyy[c] * iyy[c] + ixy[c] * ixy[c] + DATANORM;
      t1[c] = ixz[c] + ixx[c] * u;
      t2[c] = iyz[c] + ixy[c] * u;
    }
    const float tmp = m * hg3 /
                      sqrtf(t1[0] * t1[0] / n1[0] + t2[0] * t2[0] / n2[0] + t1[1] * t1[1] / n1[1] +
                            t2[1] * t2[1] / n2[1] + t1[2] * t1[2] / n1[2] + t2[2] * t2[2] / n2[2] + EPS_GRAD);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float w1 = tmp / n1[c], w2 = tmp / n2[c];
      a11 += w1 * ixx[c] * ixx[c] + w2 * ixy[c] * ixy[c];
      b1 -= w1 * ixx[c] * ixz[c] + w2 * ixy[c] * iyz[c];
    }
  }
}

// a11, b1, smooth_horiz, smooth_vert of every pixel -> planes 0..3 of `sys` in the solver's diag layout.
// Inputs row-major: mask, wx, uu [B][h][w], derivs [B][8*noc][h][w]; du in diag layout.
__global__ __launch_bounds__(256) void de_system_kernel(const DeSystemArgs a) {
  const int w = a.t.w, h = a.t.h, noc = a.t.noc;
  const int npx = w * h;
  const long long gi = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gi >= (long long)npx * a.t.nframes) return;
  const int frame = (int)(gi / npx);
  const int i = (int)(gi - (long long)frame * npx);
  const int y = i / w, x = i - y * w;
  const size_t fo = (size_t)frame * npx;
  const float* uu = a.uu + fo;
  const float* wx = a.wx + fo;
  const float sc = de_smooth_at(uu, x, y, w, h, a.quarter_alpha);
  const float sh_c = (x < w - 1) ? sc + de_smooth_at(uu, x + 1, y, w, h, a.quarter_alpha) : 0.0f;  // :150-154
  const float sv_c = (y < h - 1) ? sc + de_smooth_at(uu, x, y + 1, w, h, a.quarter_alpha) : 0.0f;  // :158-163
  const size_t dg = fo + diag_index(x, y, w, h);
  float a11, b1;
  const float* dbase = a.derivs + (size_t)frame * 8 * noc * npx + i;
  auto D = [&](int kk, int c) { return dbase[((size_t)kk * noc + c) * npx]; };
  data_term_de(D, noc, a.mask[fo + i], a.du[dg], a.half_delta_over3, a.half_gamma_over3, a11, b1);
  // sub_laplacian(b1, wx, smooth_horiz, smooth_vert) in its scatter order: -left, +right, -top, +bottom
  const float wxc = wx[i];
  if (x > 0) b1 -= (de_smooth_at(uu, x - 1, y, w, h, a.quarter_alpha) + sc) * (wxc - wx[i - 1]);
  if (x < w - 1) b1 += sh_c * (wx[i + 1] - wxc);
  if (y > 0) b1 -= (de_smooth_at(uu, x, y - 1, w, h, a.quarter_alpha) + sc) * (wxc - wx[i - w]);
  if (y < h - 1) b1 += sv_c * (wx[i + w] - wxc);
  float* out = a.sys + (size_t)frame * 4 * npx + diag_index(x, y, w, h);
  out[0] = a11;
  out[(size_t)npx] = b1;
  out[(size_t)2 * npx] = sh_c;
  out[(size_t)3 * npx] = sv_c;
}

hipError_t launch_de_system(const DeSystemArgs& a, hipStream_t s) {
  const long long total = (long long)a.t.w * a.t.h * a.t.nframes;
  hipLaunchKernelGGL(de_system_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

// One sweep of sor_coupled_slow_but_readable_DE over frames of at most 64 rows: lane = row j, step t handles
// column t - j; the updated left value is the lane's own previous result, the updated top value the previous
// lane's previous result, right and bottom come from the not-yet-updated diag row t+1.
__global__ __launch_bounds__(256) void de_sor_sweep_kernel(const DeSorArgs a, const int R) {
  const int w = a.t.w, h = a.t.h;
  const int npx = w * h;
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int G = 64 / R;
  if (wid * G >= a.t.nframes) return;
  int f = wid * G + lane / R;
  const int jr = lane % R;
  const bool row_ok = (f < a.t.nframes) && (jr < h);
  if (f >= a.t.nframes) f = a.t.nframes - 1;
  const int j = jr < h ? jr : h - 1;
  const bool has_top = j > 0, has_bot = j < h - 1;
  const float omega = a.omega;
  const float* __restrict__ sysf = a.sys + (size_t)f * 4 * npx + j;
  float* __restrict__ duf = a.du + (size_t)f * npx + j;
  auto next_row = [&](int r) { return (r + 1 == w) ? 0 : r + 1; };
  float own = duf[0];        // du of pixel (j, t - j) before the sweep: "right" of the previous step
  float res_prev = 0.0f;     // this lane's result of the previous step (its left neighbour, updated)
  float sh_prev = 0.0f, sv_prev = 0.0f;
  int row = 0;
  for (int t = 0; t <= (w - 1) + (h - 1); ++t) {
    const int i = t - j;
    const int o = row * h, o1 = next_row(row) * h;
    const float a11 = sysf[o], b1 = sysf[(size_t)npx + o], sh = sysf[(size_t)2 * npx + o], sv = sysf[(size_t)3 * npx + o];
    const float right = duf[o1];                  // (j, i+1), old
    const float bottom = wave_from_next(right);   // lane j+1 at column i-1: its right is (j+1, i), old
    const float top = wave_from_prev(res_prev);   // (j-1, i), updated one step ago
    const float vt = wave_from_prev(sv_prev);     // smooth_vert(j-1, i)
    float sigma = 0.0f, sum = 0.0f;
    if (has_top) { sigma -= vt * top; sum += vt; }
    if (i > 0) { sigma -= sh_prev * res_prev; sum += sh_prev; }
    if (has_bot) { sigma -= sv * bottom; sum += sv; }
    if (i < w - 1) { sigma -= sh * right; sum += sh; }
    const float A11 = a11 + sum;
    const float B1 = b1 - sigma;
    const float res = (1.0f - omega) * own + omega * (B1 / A11);
    if (row_ok && i >= 0 && i < w) duf[o] = res;
    res_prev = res;
    sh_prev = sh;
    sv_prev = sv;
    own = right;
    row = next_row(row);
  }
}

// any height: one thread per frame walks the pixels in raster order (correct, slow)
__global__ void de_sor_serial_kernel(const DeSorArgs a) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= a.t.nframes) return;
  const int w = a.t.w, h = a.t.h;
  const size_t npx = (size_t)w * h;
  const float* sys = a.sys + (size_t)f * 4 * npx;
  float* du = a.du + (size_t)f * npx;
  for (int it = 0; it < a.iterations; ++it)
    for (int j = 0; j < h; ++j)
      for (int i = 0; i < w; ++i) {
        const int c = diag_index(i, j, w, h);
        float sigma = 0.0f, sum = 0.0f;
        if (j > 0) { const int n = diag_index(i, j - 1, w, h); sigma -= sys[3 * npx + n] * du[n]; sum += sys[3 * npx + n]; }
        if (i > 0) { const int n = diag_index(i - 1, j, w, h); sigma -= sys[2 * npx + n] * du[n]; sum += sys[2 * npx + n]; }
        if (j < h - 1) { sigma -= sys[3 * npx + c] * du[diag_index(i, j + 1, w, h)]; sum += sys[3 * npx + c]; }
        if (i < w - 1) { sigma -= sys[2 * npx + c] * du[diag_index(i + 1, j, w, h)]; sum += sys[2 * npx + c]; }
        const float A11 = sys[c] + sum;
        const float B1 = sys[npx + c] - sigma;
        du[c] = (1.0f - a.omega) * du[c] + a.omega * (B1 / A11);
      }
}

hipError_t launch_de_sor(const DeSorArgs& a, hipStream_t s) {
  const int h = a.t.h;
  if (h <= 64 && a.t.w >= 2) {
    const int R = h <= 16 ? 16 : (h <= 32 ? 32 : 64);
    const int G = 64 / R;
    const int waves = (a.t.nframes + G - 1) / G;
    for (int it = 0; it < a.iterations; ++it)
      hipLaunchKernelGGL(de_sor_sweep_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, a, R);
  } else {
    hipLaunchKernelGGL(de_sor_serial_kernel, dim3((a.t.nframes + 63) / 64), dim3(64), 0, s, a);
  }
  return hipGetLastError();
}

// uu = min(wx + du, 0) for the left camera (camlr == 0), max(., 0) for the right one (SSE minps / maxps of
// refine_variational.cpp:303-315: the second operand, zero, is returned when the first is NaN); wx, uu row-major,
// du diag.  With `out` set also writes the plane as the level's result (flow has one channel).
__global__ __launch_bounds__(256) void de_update_kernel(TvGeom t, const float* wx, const float* du, float* uu, float* out,
                                                        int camlr) {
  const int npx = t.w * t.h;
  const long long gi = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gi >= (long long)npx * t.nframes) return;
  const int frame = (int)(gi / npx);
  const int i = (int)(gi - (long long)frame * npx);
  const int y = i / t.w, x = i - y * t.w;
  const float v = wx[gi] + du[(size_t)frame * npx + diag_index(x, y, t.w, t.h)];
  const float r = camlr == 0 ? (v < 0.0f ? v : 0.0f) : (v > 0.0f ? v : 0.0f);
  uu[gi] = r;
  if (out) out[gi] = r;
}

hipError_t launch_de_update(const TvGeom& t, const float* wx, const float* du, float* uu, float* out, int camlr,
                            hipStream_t s) {
  const long long total = (long long)t.w * t.h * t.nframes;
  hipLaunchKernelGGL(de_update_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, t, wx, du, uu, out, camlr);
  return hipGetLastError();
}

}  // namespace OFDIS_KNS
}  // namespace ofdis
