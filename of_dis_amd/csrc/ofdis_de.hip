// ofdis_de.hip -- the variational refinement of the stereo-depth mode (the reference's run_DE_* binaries,
// compile-time SELECTMODE=2): VarRefClass::RefLevelDE (refine_variational.cpp:245-336) = image_warp with a zero
// vertical flow, get_derivatives, then per fixed-point iteration
//     compute_smoothness(uu, 0)  +  compute_data_DE  +  sub_laplacian(b1, wx)     (opticalflow_aux.c:123-199,446-548)
//     sor_coupled_slow_but_readable_DE                                            (solver.c:428-466)
//     uu = min|max(wx + du, 0)   by camera side                                   (refine_variational.cpp:299-316)
// One unknown per pixel.  The solver keeps the reference's lexicographic Gauss-Seidel order with the same
// anti-diagonal wavefront and sweep pipelining as ofdis_sor.hip; the system kernel is tiled like tv_system_kernel.
#include "ofdis_kernels.h"
#include "ofdis_tvmath.h"
#include "ofdis_fused.h"

namespace ofdis {
namespace OFDIS_KNS {  // the arithmetic contract this file is being compiled for (ofdis_dev.h)

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
// Inputs row-major: mask, wx [B][h][w], derivs [B][8*noc][h][w]; du in diag layout.  The flow the smoothness weights
// are taken of, uu = min|max(wx + du, 0) by camera side (refine_variational.cpp:299-316; plain wx before the first
// solve, :283), is formed here from wx and du instead of being written by one kernel and read back by the next.
// One workgroup per 32 x 16 tile (16 rows: 0.71 against 0.78 ms per 1024 KITTI-sized pairs with 32 -- levels of 48 / 24 / 12
// rows fill 16-row tiles; 8 rows: 0.78): wx and du (halo 2) are staged once, the smoothness weight of every pixel of the
// tile + halo 1 is computed once (the per-pixel version evaluated five of them per pixel), du comes in and the four
// planes go out through LDS with the rotated enumeration of tv_system_kernel, so that both sides of the diag layout
// move as contiguous runs instead of one cache line per lane.
#ifndef OFDIS_DE_TILE_H
#define OFDIS_DE_TILE_H 16
#endif
constexpr int DT = 32, DTH = OFDIS_DE_TILE_H;   // output tile: DT columns x DTH rows (DTH <= DT)
constexpr int DU_W = DT + 4, DU_H = DTH + 4;    // uu / wx / du tiles (halo 2)
constexpr int DS_W = DT + 2, DS_H = DTH + 2;    // smoothness tile (halo 1)
constexpr int DT_PIX = DT * DTH / 256;          // pixels per thread

__global__ __launch_bounds__(256) void de_system_kernel(const DeSystemArgs a) {
  constexpr int IN_FLOATS = 3 * DU_W * DU_H + DS_W * DS_H;
  constexpr int OUT_FLOATS = 4 * DT * DTH;
  __shared__ float lds[IN_FLOATS > OUT_FLOATS ? IN_FLOATS : OUT_FLOATS];
  float* uu_t = lds;                    // halo 2
  float* wx_t = uu_t + DU_W * DU_H;     // halo 2
  float* du_t = wx_t + DU_W * DU_H;     // halo 2
  float* s_t = du_t + DU_W * DU_H;      // halo 1
  const int w = a.t.w, h = a.t.h, noc = a.t.noc;
  const int npx = w * h;
  const int tiles_x = (w + DT - 1) / DT;
  int frame, tile;
  xcd_frame_map(blockIdx.x, tiles_x * ((h + DTH - 1) / DTH), a.t.nframes, frame, tile);
  if (frame >= a.t.nframes) return;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int x0 = tx * DT, y0 = ty * DTH;
  const int tid = threadIdx.x;
  const size_t fo = (size_t)frame * npx;

  // stage 0: wx (row-major) and du (diag layout, rotated enumeration) on tile + halo 2 at border-clamped coordinates
  for (int n = tid; n < DU_W * DU_H; n += 256) {
    const int qy = n / DU_W, qx = n - qy * DU_W;
    const int y = clampi(y0 + qy - 2, 0, h - 1), x = clampi(x0 + qx - 2, 0, w - 1);
    wx_t[n] = a.wx[fo + y * w + x];
  }
  for (int n = tid; n < DU_W * DU_H; n += 256) {
    const int qy = n % DU_H, r = n / DU_H;
    int qx = r - qy;
    if (qx < 0) qx += DU_W;
    const int y = clampi(y0 + qy - 2, 0, h - 1), x = clampi(x0 + qx - 2, 0, w - 1);
    du_t[qy * DU_W + qx] = a.du[fo + diag_index(x, y, w, h)];
  }
  __syncthreads();
  // uu: SSE minps / maxps of refine_variational.cpp:303-315 (the second operand, zero, is returned for a NaN)
  for (int n = tid; n < DU_W * DU_H; n += 256) {
    const float v = wx_t[n] + du_t[n];
    uu_t[n] = a.clamp < 0 ? wx_t[n] : (a.clamp == 0 ? (v < 0.0f ? v : 0.0f) : (v > 0.0f ? v : 0.0f));
  }
  __syncthreads();
  // stage 1: compute_smoothness with vv == 0 (its derivative terms are exact zeros) on tile + halo 1: replicate
  // borders horizontally (image.c:436-464), folded coefficients on the first / last row (image.c:376-399).
  // Only in-image entries are ever read back.
  for (int n = tid; n < DS_W * DS_H; n += 256) {
    const int qy = n / DS_W, qx = n - qy * DS_W;
    const int y = y0 + qy - 1, x = x0 + qx - 1;
    float sval = 0.0f;
    if (y >= 0 && y < h && x >= 0 && x < w) {
      const int c = (qy + 1) * DU_W + qx + 1;
      const float uc = uu_t[c];
      const float ux = D3_C0 * uu_t[c - 1] + D3_C1 * uc + D3_C2 * uu_t[c + 1];
      float uy;
      if (y == 0) uy = (D3_C0 + D3_C1) * uc + D3_C2 * uu_t[c + DU_W];
      else if (y == h - 1) uy = D3_C0 * uu_t[c - DU_W] + (D3_C1 + D3_C2) * uc;
      else uy = D3_C0 * uu_t[c - DU_W] + D3_C1 * uc + D3_C2 * uu_t[c + DU_W];
      sval = a.quarter_alpha / sqrtf(ux * ux + uy * uy + EPS_SMOOTH);
    }
    s_t[n] = sval;
  }
  __syncthreads();
  // stage 2: per pixel -- compute_data_DE, then sub_laplacian(b1, wx, smooth_horiz, smooth_vert) in its scatter
  // order: -left, +right, -top, +bottom (opticalflow_aux.c:172-199)
  float res[DT_PIX][4];
  const int qx = tid % DT;
#pragma unroll
  for (int k = 0; k < DT_PIX; ++k) {
    const int ry = tid / DT + k * (256 / DT);
    const int y = y0 + ry, x = x0 + qx;
#pragma unroll
    for (int q = 0; q < 4; ++q) res[k][q] = 0.0f;
    if (y >= h || x >= w) continue;
    const int i = y * w + x;
    const int sc_i = (ry + 1) * DS_W + qx + 1;  // in the halo-1 tile
    const int uc = (ry + 2) * DU_W + qx + 2;    // in the halo-2 tiles
    const float sc = s_t[sc_i];
    const float sh_c = (x < w - 1) ? sc + s_t[sc_i + 1] : 0.0f;      // opticalflow_aux.c:150-154
    const float sv_c = (y < h - 1) ? sc + s_t[sc_i + DS_W] : 0.0f;   // :158-163
    float a11, b1;
    const float* dbase = a.derivs + (size_t)frame * 8 * noc * npx + i;
    auto D = [&](int kk, int c) { return dbase[((size_t)kk * noc + c) * npx]; };
    data_term_de(D, noc, a.mask[fo + i], du_t[uc], a.half_delta_over3, a.half_gamma_over3, a11, b1);
    const float wxc = wx_t[uc];
    if (x > 0) b1 -= (s_t[sc_i - 1] + sc) * (wxc - wx_t[uc - 1]);
    if (x < w - 1) b1 += sh_c * (wx_t[uc + 1] - wxc);
    if (y > 0) b1 -= (s_t[sc_i - DS_W] + sc) * (wxc - wx_t[uc - DU_W]);
    if (y < h - 1) b1 += sv_c * (wx_t[uc + DU_W] - wxc);
    res[k][0] = a11; res[k][1] = b1; res[k][2] = sh_c; res[k][3] = sv_c;
  }
  __syncthreads();  // input tiles dead: reuse the LDS as the output staging area [plane][ry][qx]
#pragma unroll
  for (int k = 0; k < DT_PIX; ++k) {
    const int ry = tid / DT + k * (256 / DT);
#pragma unroll
    for (int q = 0; q < 4; ++q) lds[(q * DTH + ry) * DT + qx] = res[k][q];
  }
  __syncthreads();
  // stage 3: the four planes in diag layout, rotated enumeration (lane -> x-1, y+1)
  for (int n = tid; n < DT * DTH; n += 256) {
    const int ry = n % DTH, r = n / DTH;
    const int rx = (r - ry) & (DT - 1);
    const int y = y0 + ry, x = x0 + rx;
    if (y < h && x < w) {
      float* out = a.sys + (size_t)frame * 4 * npx + diag_index(x, y, w, h);
#pragma unroll
      for (int q = 0; q < 4; ++q) out[(size_t)q * npx] = lds[(q * DTH + ry) * DT + rx];
    }
  }
}

hipError_t launch_de_system(const DeSystemArgs& a, hipStream_t s) {
  const int tiles = ((a.t.w + DT - 1) / DT) * ((a.t.h + DTH - 1) / DTH);
  hipLaunchKernelGGL(de_system_kernel, dim3(((a.t.nframes + 7) / 8) * 8 * tiles), dim3(256), 0, s, a);
  return hipGetLastError();
}

// sor_coupled_slow_but_readable_DE, all NS sweeps of the call in one pass
// (the scheme of sor_wave_kernel, ofdis_sor.hip): lane = row j, at step t sweep s is at column t - j - 2s.  The
// updated left value is the lane's own previous result of the same sweep, the updated top value the previous
// lane's; own, right and bottom are the previous sweep's results (the stored du for sweep 0: right and bottom
// come from diag row t+1).  Loads run PD steps ahead through a register ring that also hands each pixel's
// coefficients from sweep 0 to the trailing sweeps.
struct DeSlot {
  float a11, b1, sh, sv;  // loaded (diag row tau)
  float dur;              // stored du of the right neighbour (diag row tau+1)
  float hl, vt;           // left / top edge weights, filled in at step tau
};

// MAXT == 0: frames of at most 64 rows, 64 / R frames per wavefront, four wavefronts per workgroup.
// MAXT > 0 (the scheme of sor_block_kernel): one workgroup per frame, wavefront k owns rows 64k .. 64k+63, all
// wavefronts advance in lock step; the values that cross a wavefront boundary travel through a double-buffered LDS
// mailbox written at the end of a step and read at the start of the next.  Up to 16 wavefronts (h <= 1024).
template <int NS, int PD, int MAXT>
__global__ __launch_bounds__(MAXT > 0 ? MAXT : 256) void de_sor_kernel(const DeSorArgs a, const int R) {
  constexpr bool BLOCK = MAXT > 0;
  constexpr int LIFE = (2 * (NS - 1) > 1) ? 2 * (NS - 1) : 1;
  constexpr int RS = PD + LIFE + 1;
  __shared__ float mail_top[BLOCK ? 2 : 1][BLOCK ? 16 : 1][NS + 1];  // lane 63 of wave k -> lane 0 of wave k+1
  __shared__ float mail_bot[BLOCK ? 2 : 1][BLOCK ? 16 : 1][NS];      // lane 0 of wave k -> lane 63 of wave k-1
  const int w = a.t.w, h = a.t.h;
  const int npx = w * h;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  int f, jr;
  bool row_ok, first_lane = false, last_lane = false;
  if constexpr (BLOCK) {
    f = blockIdx.x;
    jr = threadIdx.x;
    row_ok = jr < h;
    first_lane = lane == 0 && wave > 0;
    last_lane = lane == 63 && wave + 1 < (int)(blockDim.x >> 6);
  } else {
    const int wid = blockIdx.x * 4 + wave;
    const int G = 64 / R;
    if (wid * G >= a.t.nframes) return;
    f = wid * G + lane / R;
    jr = lane % R;
    row_ok = (f < a.t.nframes) && (jr < h);
    if (f >= a.t.nframes) f = a.t.nframes - 1;
  }
  const int j = jr < h ? jr : h - 1;
  const bool has_top = j > 0, has_bot = j < h - 1;
  const float omega = a.omega;
  const float* __restrict__ sysf = a.sys + (size_t)f * 4 * npx + j;
  float* __restrict__ duf = a.du + (size_t)f * npx + j;

  DeSlot ring[RS];
#pragma unroll
  for (int r = 0; r < RS; ++r) ring[r] = DeSlot{0, 0, 0, 0, 0, 0, 0};
  float ru[NS], ru2[NS];  // result of sweep s one / two steps ago
#pragma unroll
  for (int s = 0; s < NS; ++s) ru[s] = ru2[s] = 0.0f;
  auto load_slot = [&](DeSlot& sl, int drow, int drow1) {
    const int o = drow * h;
    sl.a11 = sysf[o];
    sl.b1 = sysf[(size_t)npx + o];
    sl.sh = sysf[(size_t)2 * npx + o];
    sl.sv = sysf[(size_t)3 * npx + o];
    sl.dur = duf[drow1 * h];
  };
  auto next_row = [&](int r) { return (r + 1 == w) ? 0 : r + 1; };
  int lrow = 0;
#pragma unroll
  for (int q = 0; q < PD; ++q) {
    load_slot(ring[q], lrow, next_row(lrow));
    lrow = next_row(lrow);
  }
  ring[RS - 1].dur = duf[0];  // "right" of step -1 = own of step 0
  int srow = (w - ((2 * (NS - 1)) % w)) % w;  // diag row of the pixel the last sweep finishes at step 0
  if constexpr (BLOCK) {
    // mailbox for step 0: nothing has been computed yet, but sweep 0's "bottom" of lane 63 is the next wavefront's
    // lane-0 right value of step 0, which is loaded
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < NS; ++q) mail_bot[1][wave][q] = 0.0f;
      mail_bot[1][wave][0] = ring[0].dur;
    }
    if (lane == 63) {
#pragma unroll
      for (int q = 0; q < NS + 1; ++q) mail_top[1][wave][q] = 0.0f;
    }
    __syncthreads();
  }
  const int tend = (w - 1) + (h - 1) + 2 * (NS - 1);
  for (int t0 = 0; t0 <= tend; t0 += RS) {
#pragma unroll
    for (int u = 0; u < RS; ++u) {
      const int t = t0 + u;
      load_slot(ring[(u + PD) % RS], lrow, next_row(lrow));
      lrow = next_row(lrow);
      float top_sv = 0.0f, top_u[NS], bot_u[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) top_u[s] = bot_u[s] = 0.0f;
      if constexpr (BLOCK) {
        const int rd = (t + 1) & 1;  // written at the end of step t-1
        if (first_lane) {
          top_sv = mail_top[rd][wave - 1][NS];
#pragma unroll
          for (int s = 0; s < NS; ++s) top_u[s] = mail_top[rd][wave - 1][s];
        }
        if (last_lane) {
#pragma unroll
          for (int s = 0; s < NS; ++s) bot_u[s] = mail_bot[rd][wave + 1][s];
        }
      }
      {
        DeSlot& c = ring[u];
        const DeSlot& p = ring[(u + RS - 1) % RS];
        c.hl = p.sh;                   // smooth_horiz(j, i0 - 1): used for i0 > 0 only
        c.vt = wave_from_prev(p.sv);   // smooth_vert(j - 1, i0): lane j-1 was at column i0 one step ago
        if (BLOCK && first_lane) c.vt = top_sv;
      }
      float nu[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int i = t - j - 2 * s;
        const DeSlot& c = ring[(u - 2 * s + 2 * RS) % RS];
        float own, right, bottom;
        if (s == 0) {
          const DeSlot& p = ring[(u + RS - 1) % RS];
          own = p.dur;
          right = c.dur;
          bottom = wave_from_next(c.dur);  // lane j+1 is at column i-1: its right is (j+1, i)
        } else {
          own = ru2[s - 1];
          right = ru[s - 1];
          bottom = wave_from_next(ru[s - 1]);
        }
        if (BLOCK && last_lane) bottom = bot_u[s];
        float top = wave_from_prev(ru[s]);
        if (BLOCK && first_lane) top = top_u[s];
        const float left = ru[s];
        float sigma = 0.0f, sum = 0.0f;                                  // solver.c:436-456
        if (has_top) { sigma -= c.vt * top; sum += c.vt; }
        if (i > 0) { sigma -= c.hl * left; sum += c.hl; }
        if (has_bot) { sigma -= c.sv * bottom; sum += c.sv; }
        if (i < w - 1) { sigma -= c.sh * right; sum += c.sh; }
        const float A11 = c.a11 + sum;
        const float B1 = c.b1 - sigma;
        nu[s] = (1.0f - omega) * own + omega * (B1 / A11);
      }
      {
        const int i = t - j - 2 * (NS - 1);
        if (row_ok && i >= 0 && i < w) duf[srow * h] = nu[NS - 1];
        srow = next_row(srow);
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        ru2[s] = ru[s];
        ru[s] = nu[s];
      }
      if constexpr (BLOCK) {  // publish for step t+1
        const int wr = t & 1;
        if (lane == 63) {
#pragma unroll
          for (int s = 0; s < NS; ++s) mail_top[wr][wave][s] = ru[s];
          mail_top[wr][wave][NS] = ring[u].sv;  // slot of step t: the next step's top weight
        }
        if (lane == 0) {
          // bottom of sweep s at step t+1: sweep 0 -> this lane's right value of step t+1; sweep s > 0 -> its
          // sweep s-1 result of step t
          mail_bot[wr][wave][0] = ring[(u + 1) % RS].dur;
#pragma unroll
          for (int s = 1; s < NS; ++s) mail_bot[wr][wave][s] = ru[s - 1];
        }
        // only the LDS mailboxes cross wavefronts: drain the LDS counter, not the loads requested PD steps ahead
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
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
  constexpr int PD = 3;
  if (a.t.w < 2 || h > 1024) {
    hipLaunchKernelGGL(de_sor_serial_kernel, dim3((a.t.nframes + 63) / 64), dim3(64), 0, s, a);
    return hipGetLastError();
  }
  const int R = h <= 16 ? 16 : (h <= 32 ? 32 : 64);
  const int waves = (a.t.nframes + 64 / R - 1) / (64 / R);
  const int threads = ((h + 63) / 64) * 64;
  for (int left = a.iterations; left > 0;) {   // at most four sweeps share a pass (three above 64 rows)
    const int ns = h <= 64 ? (left < 4 ? left : 4) : (left < 3 ? left : 3);
    if (h <= 64) {
      const dim3 g((waves + 3) / 4), b(256);
      switch (ns) {
        case 1: hipLaunchKernelGGL((de_sor_kernel<1, PD, 0>), g, b, 0, s, a, R); break;
        case 2: hipLaunchKernelGGL((de_sor_kernel<2, PD, 0>), g, b, 0, s, a, R); break;
        case 3: hipLaunchKernelGGL((de_sor_kernel<3, PD, 0>), g, b, 0, s, a, R); break;
        default: hipLaunchKernelGGL((de_sor_kernel<4, PD, 0>), g, b, 0, s, a, R); break;
      }
    } else {
      const dim3 g(a.t.nframes), b(threads);
      switch (ns) {
        case 1: hipLaunchKernelGGL((de_sor_kernel<1, PD, 1024>), g, b, 0, s, a, 64); break;
        case 2: hipLaunchKernelGGL((de_sor_kernel<2, PD, 1024>), g, b, 0, s, a, 64); break;
        default: hipLaunchKernelGGL((de_sor_kernel<3, PD, 1024>), g, b, 0, s, a, 64); break;
      }
    }
    left -= ns;
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------ fused stereo refinement
// Every fixed-point iteration of a stereo level of at most 64 rows in ONE launch: the scheme of tv_fused_kernel's throughput
// mapping (ofdis_fused.hip: lane = row, one anti-diagonal per step, the system of a pixel assembled in registers one step
// before the pipelined solver sweeps consume it, the iterations of a frame back to back as n_inner * w columns) with one
// unknown per pixel:
//   * records in: the derivative records of derivatives_kernel's record form (NOC arrays) and the (wx, 0) records;
//   * uu = wx in the first iteration, min | max (wx + du, 0) by camera side after it (refine_variational.cpp:283, 299-316);
//     the smoothness weight is compute_smoothness with vv == 0 (its terms are exact zeros);
//   * the data term is the flow mode's with v == 0 (compute_data_DE, opticalflow_aux.c:446-548, is compute_data without the
//     second unknown: data_term_gray / data_term_rgb, a11 and b1 of them), the Laplacian of wx in its scatter order;
//   * the solver is sor_coupled_slow_but_readable_DE's update (solver.c:436-456) with zero edge weights standing for the
//     absent neighbours (the sums start at +0 and never become -0: adding or subtracting +-0 x finite changes no bit);
//   * du goes back to its diag plane (the next iteration of the same wavefront reads it w columns later; de_update forms the
//     clamped result).
// n_inner x (de_system + de_sor) become one kernel: the derivative records are read once per iteration, nothing else moves.
struct DeFRow {
  float wx, du;
  bool first;  // the pixel is in its first fixed-point iteration (or past the last): du == 0 and uu = wx, unclamped
};
struct DeFSlot {
  float a11, b1, sh, sv, dur, hl, vt;
};

template <int NS, bool BRIGHT, int NOC>
__global__ __launch_bounds__(256) void de_fused_kernel(const DeFusedArgs a, const int R) {
  constexpr int U = 6, PDW = 5, PDD = NOC == 3 ? 2 : 3;
  static_assert(2 * (NS - 1) + 1 < U, "slot ring too small for this many pipelined sweeps");
  const int w = a.t.w, h = a.t.h;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int G = 64 / R;  // frames per wavefront
  const int s0 = wid * G;
  if (s0 >= a.t.nframes) return;
  int fl = lane / R;
  const int jr = lane % R;
  const bool row_ok = (s0 + fl < a.t.nframes) && (jr < h);
  if (s0 + fl >= a.t.nframes) fl = a.t.nframes - 1 - s0;
  const int j = jr < h ? jr : h - 1;
  const bool has_top = j > 0, has_bot = j < h - 1;
  const float omega = a.omega, qa = a.quarter_alpha, hd3 = a.half_delta_over3, hg3 = a.half_gamma_over3;
  const int camlr = a.camlr;
  const int nst = min(G, a.t.nframes - s0);
  const size_t recs = (size_t)w * h;  // records of a frame
  auto rsrc = [&](const float* base, int rec_floats) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)s0 * recs * rec_floats), 0, (int)(nst * recs * rec_floats * 4),
                                             0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsW = rsrc(a.wrec, 2), rsU = rsrc(a.du, 1), rsD = rsrc(a.d8, 8);
  const __amdgpu_buffer_rsrc_t rsD1 = rsrc(a.d8 + (NOC == 3 ? (size_t)a.t.nframes * recs * 8 : 0), 8);
  const __amdgpu_buffer_rsrc_t rsD2 = rsrc(a.d8 + (NOC == 3 ? (size_t)a.t.nframes * recs * 16 : 0), 8);
  const int vrec = fl * (int)recs + j;
  const int vo8 = vrec * 32, vo2 = vrec * 8, vo1 = vrec * 4;
  auto asf = [](unsigned u) { return __builtin_bit_cast(float, u); };

  DeFRow W[6];
  FDer D[PDD][NOC];
  float uu[3], sm[3];
  DeFSlot slot[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) { W[r] = DeFRow{0, 0, true}; slot[r] = DeFSlot{1, 0, 1, 1, 0, 0, 0}; }  // (finite fill-phase systems)
#pragma unroll
  for (int r = 0; r < 3; ++r) { uu[r] = 0.0f; sm[r] = 1.0f; }
#pragma unroll
  for (int r = 0; r < PDD; ++r)
#pragma unroll
    for (int c = 0; c < NOC; ++c) D[r][c] = FDer{0, 0, 0, 0, 0, 0, 0, 0};
  float ru[NS], ru2[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) ru[s] = ru2[s] = 0.0f;
  float ldx = 0.0f;

  auto wrap_row = [&](int r) { r %= w; return r < 0 ? r + w : r; };
  auto next_row = [&](int r) { return (r + 1 == w) ? 0 : r + 1; };
  // zero: no memory request for du (an offset beyond the resource reads +0): first iteration, or past the last one
  auto load_w = [&](DeFRow& r, int drow, bool zero) {
    r.wx = asf(__builtin_amdgcn_raw_buffer_load_b32(rsW, vo2, drow * h * 8, 0));
    r.du = asf(__builtin_amdgcn_raw_buffer_load_b32(rsU, zero ? 0x7ffffff0 : vo1, drow * h * 4, 0));
    r.first = zero;
  };
  auto load_d1 = [&](FDer& r, const __amdgpu_buffer_rsrc_t& rs, int o) {
    const auto lo = __builtin_amdgcn_raw_buffer_load_b128(rs, vo8, o, 0);
    const auto hi = __builtin_amdgcn_raw_buffer_load_b128(rs, vo8 + 16, o, 0);
    const unsigned l0 = lo[0], l1 = lo[1], l2 = lo[2], l3 = lo[3], h0 = hi[0], h1 = hi[1], h2 = hi[2], h3 = hi[3];
    r.ix = asf(l0); r.iz = asf(l1); r.ixx = asf(l2); r.ixz = asf(l3);
    r.iy = asf(h0); r.ixy = asf(h1); r.iyz = asf(h2); r.iyy = asf(h3);
  };
  auto load_d = [&](FDer (&r)[NOC], int drow) {
    const int o = drow * h * 32;
    load_d1(r[0], rsD, o);
    if constexpr (NOC == 3) {
      load_d1(r[1], rsD1, o);
      load_d1(r[2], rsD2, o);
    }
  };
  // ring index of diag row rho is (rho + 3) mod ring size; loop variable k = t + 3, u = k % 6: row t + c at index (u + c) % size
  load_w(W[2], wrap_row(-1), true);
  load_w(W[3], wrap_row(0), true);
  load_w(W[4], wrap_row(1), true);
  int rowW = wrap_row(PDW - 3), rowD = wrap_row(PDD - 3);
  int srow = wrap_row(-3 - 2 * (NS - 1));  // row the last sweep finishes at step t = -3
  auto wrap_col = [&](int c) { c %= w; return c < 0 ? c + w : c; };
  int x2 = wrap_col(-1 - j);                    // this lane's x on diag row t + 2
  bool x1_last = (wrap_col(-2 - j) == w - 1);   // row t + 1 is the last column
  const int wtot = a.n_inner * w;
  const int tend = (wtot - 1) + (h - 1) + 2 * (NS - 1);
  int ig = -3 - j - 2 * (NS - 1);  // column (over all iterations) the last sweep finishes at step t
  bool first_w = true;
  for (int k0 = 0; k0 <= tend + 3; k0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // ---- (1) loads: wx / du row t + 5, derivative row t + PDD
      load_w(W[(u + PDW) % 6], rowW, first_w);
      rowW = next_row(rowW);
      load_d(D[(u + PDD) % PDD], rowD);
      rowD = next_row(rowD);
      // ---- (2) uu of row t + 3
      {
        const DeFRow& r = W[(u + 3) % 6];
        const float v = r.wx + r.du;
        const float cl = camlr == 0 ? (v < 0.0f ? v : 0.0f) : (v > 0.0f ? v : 0.0f);  // minps / maxps (x, 0): 0 for a NaN
        uu[u % 3] = r.first ? v : cl;
      }
      // ---- (3) smoothness of row t + 2 (opticalflow_aux.c:128-140 with vv == 0; see tv_fused_kernel for the scaling)
      const bool x2_last = (x2 == w - 1);
      {
        const float uc = uu[(u + 2) % 3];
        float ul = uu[(u + 1) % 3], ur = uu[u % 3];
        float ut = wave_from_prev(uu[(u + 1) % 3]), ub = wave_from_next(uu[u % 3]);
        if (x2 == 0) ul = uc;
        if (x2_last) ur = uc;
        if (!has_top) ut = uc;
        if (!has_bot) ub = uc;
        const float ex = ur - ul, ey = ub - ut;
        sm[(u + 2) % 3] = fdiv_by_sqrt(qa, 0.25f * (ex * ex + ey * ey) + EPS_SMOOTH);
      }
      // ---- (4) system of the pixel on row t + 1
      {
        const float sc = sm[(u + 1) % 3], s_r = sm[(u + 2) % 3], s_d = wave_from_next(sm[(u + 2) % 3]);
        const float sh_c = x1_last ? 0.0f : sc + s_r;
        const float sv_c = has_bot ? sc + s_d : 0.0f;
        const DeFRow& rc = W[(u + 1) % 6];
        const DeFRow& rm = W[u % 6];
        const DeFRow& rp = W[(u + 2) % 6];
        float a11, a12, a22, b1, b2;
        if constexpr (NOC == 1) data_term_gray<BRIGHT>(D[(u + 1) % PDD][0], rc.du, 0.0f, hd3, hg3, a11, a12, a22, b1, b2);
        else data_term_rgb<BRIGHT>(D[(u + 1) % PDD], rc.du, 0.0f, hd3, hg3, a11, a12, a22, b1, b2);
        const float wx_u = wave_from_prev(rm.wx), wx_d = wave_from_next(rp.wx);
        const float sh_l = slot[u % 6].sh;
        const float sv_t = wave_from_prev(slot[u % 6].sv);
        const float rdx = rp.wx - rc.wx;
        b1 -= sh_l * ldx;
        b1 += sh_c * rdx;
        ldx = rdx;
        b1 -= sv_t * (rc.wx - wx_u);
        b1 += sv_c * (wx_d - rc.wx);
        DeFSlot& o = slot[(u + 1) % 6];
        o.a11 = a11; o.b1 = b1; o.sh = sh_c; o.sv = sv_c; o.dur = rp.du; o.hl = sh_l; o.vt = sv_t;
      }
      x1_last = x2_last;
      x2 = x2_last ? 0 : x2 + 1;
      // ---- (5) solver step t: sweep s is at column t - j - 2 s (solver.c:436-456)
      float nu[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const DeFSlot& c = slot[(u - 2 * s + 12) % 6];
        float own, right, bottom;
        if (s == 0) {
          const DeFSlot& p = slot[(u + 5) % 6];
          own = p.dur;
          right = c.dur;
          bottom = wave_from_next(c.dur);
        } else {
          own = ru2[s - 1];
          right = ru[s - 1];
          bottom = wave_from_next(ru[s - 1]);
        }
        const float top = wave_from_prev(ru[s]);
        const float left = ru[s];
        float sigma = 0.0f, sum = 0.0f;
        sigma -= c.vt * top;    sum += c.vt;
        sigma -= c.hl * left;   sum += c.hl;
        sigma -= c.sv * bottom; sum += c.sv;
        sigma -= c.sh * right;  sum += c.sh;
        const float A11 = c.a11 + sum;
        const float B1 = c.b1 - sigma;
        nu[s] = (1.0f - omega) * own + omega * fdiv_rn(B1, A11);
      }
      {
        const int cw = ig + (PDW + 1 + 2 * (NS - 1));
        first_w = (cw < w) | (cw >= wtot);  // for the next step's row
        const bool on = row_ok & (ig >= 0) & (ig < wtot);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, nu[NS - 1]), rsU, on ? vo1 : 0x7ffffff0, srow * h * 4, 0);
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        ru2[s] = ru[s];
        ru[s] = nu[s];
      }
      srow = next_row(srow);
      ++ig;
    }
  }
}

bool de_fused_supported(const TvGeom& t, int iterations) {
  return (t.noc == 1 || t.noc == 3) && t.h >= 4 && t.h <= 64 && t.w >= 16 && iterations >= 1 && iterations <= 3;
}

hipError_t launch_de_fused(const DeFusedArgs& a, hipStream_t s) {
  if (!de_fused_supported(a.t, a.iterations) || a.n_inner < 1 || !tv_fused_params_ok(a.quarter_alpha, a.half_delta_over3, a.half_gamma_over3))
    return hipErrorInvalidValue;
  const int h = a.t.h;
  const int R = h <= 16 ? 16 : (h <= 32 ? 32 : 64);
  const int waves = (a.t.nframes + 64 / R - 1) / (64 / R);
  const dim3 g((waves + 3) / 4), b(256);
  const bool bright = a.half_delta_over3 != 0.0f;
#define OFDIS_DE_FUSED(NS, NOC)                                                                  \
  if (bright) hipLaunchKernelGGL((de_fused_kernel<NS, true, NOC>), g, b, 0, s, a, R);            \
  else hipLaunchKernelGGL((de_fused_kernel<NS, false, NOC>), g, b, 0, s, a, R)
#define OFDIS_DE_FUSED_NS(NOC)                                                                   \
  switch (a.iterations) {                                                                        \
    case 1: OFDIS_DE_FUSED(1, NOC); break;                                                       \
    case 2: OFDIS_DE_FUSED(2, NOC); break;                                                       \
    default: OFDIS_DE_FUSED(3, NOC); break;                                                      \
  }
  if (a.t.noc == 3) { OFDIS_DE_FUSED_NS(3) } else { OFDIS_DE_FUSED_NS(1) }
#undef OFDIS_DE_FUSED_NS
#undef OFDIS_DE_FUSED
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
  if (uu) uu[gi] = r;
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
