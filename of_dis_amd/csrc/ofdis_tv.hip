// ofdis_tv.hip -- FDF1.0.1 TV-L1 refinement kernels for gfx950 (everything except the SOR sweep).
//
//   warp_kernel        image_warp            opticalflow_aux.c:18-60
//   derivatives_kernel get_derivatives       opticalflow_aux.c:65-116 (+ image.c:401-434,466-502)
//   tv_system_kernel   compute_smoothness + compute_data + 2x sub_laplacian fused
//                                            opticalflow_aux.c:123-199, 310-438 (+ image.c:376-399,436-464)
//   tv_finish_kernel   uu=wx+du, vv=wy+dv -> AoS      refine_variational.cpp:209-221, 92-99
//   flow_split_kernel  AoS -> planar                  refine_variational.cpp:56-68
//
// All kernels are batched over frames (blockIdx / flat index carries the frame), planes are packed
// row-major [frame][...][h][w].  The stencil kernels stage a tile plus halo in LDS; the border
// rules of the reference (replicated columns, folded coefficients on the first/last rows) are
// applied per stage exactly as the reference applies them per convolution call.
#include <stdlib.h>

#include "ofdis_kernels.h"
#include "ofdis_tvmath.h"

namespace ofdis {
namespace OFDIS_KNS {  // the arithmetic contract this file is being compiled for (ofdis_dev.h)

// ------------------------------------------------------------------------------------------ warp
// image_warp (opticalflow_aux.c:18-60).  HBM-streaming kernel: per pixel it reads wx, wy, writes the
// warped value(s) and the mask; the four bilinear taps come from the frame's own padded plane, which is
// L2-resident (41 KB at op-point 2).  VEC = pixels per thread: 4 consecutive x (16-byte loads/stores of
// wx, wy, dst, mask) when w % 4 == 0, else 1.
// NOC is a compile-time constant so that the channel loops unroll and the taps of the (four) pixels a thread handles are
// all requested before the first one is used; with a run-time channel count every pixel's loads were waited for in turn.
template <bool PADDED, int NOC>
__device__ __forceinline__ void warp_pixel(const WarpArgs& a, int frame, int i, int j, float fx, float fy, float& m,
                                           float* out /*[noc]*/) {
  const int w = a.t.w, h = a.t.h;
  constexpr int noc = NOC;
  const float xx = i + fx;
  const float yy = j + fy;
  const int x = (int)floorf(xx), y = (int)floorf(yy);
  const float dx = xx - (float)x, dy = yy - (float)y;
  m = (xx >= 0 && xx <= (float)(w - 1) && yy >= 0 && yy <= (float)(h - 1)) ? 1.0f : 0.0f;
  const int x1 = clampi(x, 0, w - 1), x2 = clampi(x + 1, 0, w - 1);
  const int y1 = clampi(y, 0, h - 1), y2 = clampi(y + 1, 0, h - 1);
#pragma unroll
  for (int c = 0; c < noc; ++c) {
    float s11, s12, s21, s22;
    if (PADDED) {
      const float* s = a.src + (size_t)frame * a.tmp_w * a.tmp_h * noc;
      s11 = s[((y1 + a.pad) * a.tmp_w + x1 + a.pad) * noc + c];
      s12 = s[((y1 + a.pad) * a.tmp_w + x2 + a.pad) * noc + c];
      s21 = s[((y2 + a.pad) * a.tmp_w + x1 + a.pad) * noc + c];
      s22 = s[((y2 + a.pad) * a.tmp_w + x2 + a.pad) * noc + c];
    } else {
      const float* s = a.src + ((size_t)frame * noc + c) * w * h;
      s11 = s[y1 * w + x1];
      s12 = s[y1 * w + x2];
      s21 = s[y2 * w + x1];
      s22 = s[y2 * w + x2];
    }
    out[c] = s11 * (1.0f - dx) * (1.0f - dy) + s12 * dx * (1.0f - dy) + s21 * (1.0f - dx) * dy + s22 * dx * dy;
  }
}

typedef float f4 __attribute__((ext_vector_type(4)));
// The row-major kernel streams: flow planes are read once, the warped image and the mask written once -- non-temporal
// hints measured 4.49 -> 4.73 TB/s on 587 MB launches.  (Inside the fused-TV pipeline the flow was written by the
// previous kernel and still sits in L2 / Infinity Cache: there the hints cost 10 %, so warp_diag_kernel does not use them.)
__device__ __forceinline__ f4 nt_load(const f4* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void nt_store(f4 v, f4* p) { __builtin_nontemporal_store(v, p); }

// 1-D grid of (column chunks x row groups) blocks per frame, a frame's blocks on ONE XCD (xcd_frame_map: its source plane is
// fetched into one L2 instead of up to eight); block = TX x (256/TX) threads with TX a power of two >= w/VEC (<= 256): pixel
// coordinates come from shifts and masks -- no 64-bit index arithmetic.
template <bool PADDED, int VEC, int NOC>
__global__ __launch_bounds__(256) void warp_kernel(const WarpArgs a, const int tx_shift, const int gx, const int gy) {
  const int w = a.t.w, h = a.t.h;
  constexpr int noc = NOC;
  const int npx = w * h;
  int frame, blk;
  xcd_frame_map(blockIdx.x, gx * gy, a.t.nframes, frame, blk);
  if (frame >= a.t.nframes) return;
  const int by = blk / gx, bx = blk - by * gx;
  const size_t fo = (size_t)frame * npx;
  const int tx = threadIdx.x & ((1 << tx_shift) - 1), ty = threadIdx.x >> tx_shift;
  const int i = ((bx << tx_shift) + tx) * VEC;
  const int j = by * (256 >> tx_shift) + ty;
  if (i >= w || j >= h) return;
  const int o = j * w + i;
  if constexpr (VEC == 4) {
    const f4 fxv = nt_load(reinterpret_cast<const f4*>(a.wx + fo + o));
    const f4 fyv = nt_load(reinterpret_cast<const f4*>(a.wy + fo + o));
    float4 m;
    float r0[3], r1[3], r2[3], r3[3];
    // Gray fast path: where the flow is smooth the four pixels of a thread have CONSECUTIVE tap columns in the same two
    // source rows, away from every border -- their 16 taps are 2 x 5 consecutive floats: four loads instead of sixteen
    // 4-byte gathers (the kernel is bound by its memory instructions, not its bytes).  Same values into the same
    // expression as warp_pixel: same bits.  Threads for which it does not hold take the general path below.
    bool fast = false;
    if constexpr (NOC == 1) {
      const float fxs[4] = {fxv.x, fxv.y, fxv.z, fxv.w}, fys[4] = {fyv.x, fyv.y, fyv.z, fyv.w};
      float xx[4], yy[4];
      int xb[4], yb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xx[k] = (i + k) + fxs[k];
        yy[k] = j + fys[k];
        xb[k] = (int)floorf(xx[k]);
        yb[k] = (int)floorf(yy[k]);
      }
      fast = (xb[1] == xb[0] + 1) & (xb[2] == xb[0] + 2) & (xb[3] == xb[0] + 3) & (yb[1] == yb[0]) & (yb[2] == yb[0]) &
             (yb[3] == yb[0]) & (xb[0] >= 0) & (xb[0] + 4 <= w - 1) & (yb[0] >= 0) & (yb[0] + 1 <= h - 1);
      if (fast) {  // (inside the image: every mask is 1, no tap is clamped)
        const float* s = PADDED ? a.src + (size_t)frame * a.tmp_w * a.tmp_h + (size_t)(yb[0] + a.pad) * a.tmp_w + xb[0] + a.pad
                                : a.src + (size_t)frame * npx + (size_t)yb[0] * w + xb[0];
        const int pitch = PADDED ? a.tmp_w : w;
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        const f4u t1 = *reinterpret_cast<const f4u*>(s), t2 = *reinterpret_cast<const f4u*>(s + pitch);
        const float R1[5] = {t1.x, t1.y, t1.z, t1.w, s[4]}, R2[5] = {t2.x, t2.y, t2.z, t2.w, s[pitch + 4]};
        float out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float dx = xx[k] - (float)xb[k], dy = yy[k] - (float)yb[k];
          out[k] = R1[k] * (1.0f - dx) * (1.0f - dy) + R1[k + 1] * dx * (1.0f - dy) + R2[k] * (1.0f - dx) * dy + R2[k + 1] * dx * dy;
        }
        m = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        r0[0] = out[0]; r1[0] = out[1]; r2[0] = out[2]; r3[0] = out[3];
      }
    }
    if (!fast) {
    warp_pixel<PADDED, NOC>(a, frame, i + 0, j, fxv.x, fyv.x, m.x, r0);
    warp_pixel<PADDED, NOC>(a, frame, i + 1, j, fxv.y, fyv.y, m.y, r1);
    warp_pixel<PADDED, NOC>(a, frame, i + 2, j, fxv.z, fyv.z, m.z, r2);
    warp_pixel<PADDED, NOC>(a, frame, i + 3, j, fxv.w, fyv.w, m.w, r3);
    }
    nt_store((f4){m.x, m.y, m.z, m.w}, reinterpret_cast<f4*>(a.mask + fo + o));
#pragma unroll
    for (int c = 0; c < noc; ++c)
      nt_store((f4){r0[c], r1[c], r2[c], r3[c]}, reinterpret_cast<f4*>(a.dst + ((size_t)frame * noc + c) * npx + o));
  } else {
    float m, r[3];
    warp_pixel<PADDED, NOC>(a, frame, i, j, a.wx[fo + o], a.wy[fo + o], m, r);
    a.mask[fo + o] = m;
#pragma unroll
    for (int c = 0; c < noc; ++c) a.dst[((size_t)frame * noc + c) * npx + o] = r[c];
  }
}

hipError_t launch_warp(const WarpArgs& a, hipStream_t s) {
  const bool v4 = (a.t.w % 4) == 0;
  if (a.t.noc != 1 && a.t.noc != 3) return hipErrorInvalidValue;
  const int cols = a.t.w / (v4 ? 4 : 1);
  int tx_shift = 0;
  while ((1 << tx_shift) < cols && tx_shift < 8) ++tx_shift;
  const int rows_per_block = 256 >> tx_shift;
  const int gx = (cols + (1 << tx_shift) - 1) >> tx_shift, gy = (a.t.h + rows_per_block - 1) / rows_per_block;
  const long long blocks = (long long)((a.t.nframes + 7) / 8) * 8 * gx * gy;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  const dim3 g((unsigned)blocks), b(256);
#define OFDIS_WARP_LAUNCH(P, V)                                                                          \
  do {                                                                                                   \
    if (a.t.noc == 1) hipLaunchKernelGGL((warp_kernel<P, V, 1>), g, b, 0, s, a, tx_shift, gx, gy);     \
    else hipLaunchKernelGGL((warp_kernel<P, V, 3>), g, b, 0, s, a, tx_shift, gx, gy);                  \
  } while (0)
  if (a.src_padded) {
    if (v4) OFDIS_WARP_LAUNCH(true, 4);
    else OFDIS_WARP_LAUNCH(true, 1);
  } else {
    if (v4) OFDIS_WARP_LAUNCH(false, 4);
    else OFDIS_WARP_LAUNCH(false, 1);
  }
#undef OFDIS_WARP_LAUNCH
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------ derivatives
// (the 5-tap derivative filter coefficients D5_C* live in ofdis_tvmath.h)

constexpr int DT_W = 32, DT_H = 16;         // output tile
constexpr int DA_W = DT_W + 8, DA_H = DT_H + 8;  // avg / Iz tile (halo 4)
constexpr int DX_W = DT_W + 4, DX_H = DT_H + 4;  // Ix / Iy tile (halo 2)

// horizontal 5-tap on an LDS tile whose entries are already border-replicated (image.c:466-502)
__device__ __forceinline__ float h5(const float* t, int pitch, int qy, int qx) {
  const float* r = t + qy * pitch + qx;
  return D5_C0 * r[-2] + D5_C1 * r[-1] + D5_C2 * r[0] + D5_C3 * r[1] + D5_C4 * r[2];
}
// vertical 5-tap with the folded coefficients of the first/last two image rows (image.c:401-434);
// j = image row of the output sample
__device__ __forceinline__ float v5(const float* t, int pitch, int qy, int qx, int j, int h) {
  const float* r = t + qy * pitch + qx;
  const float m2 = r[-2 * pitch], m1 = r[-pitch], s0 = r[0], p1 = r[pitch], p2 = r[2 * pitch];
  if (j == 0) return (D5_C0 + D5_C1 + D5_C2) * s0 + D5_C3 * p1 + D5_C4 * p2;
  if (j == 1) return (D5_C0 + D5_C1) * m1 + D5_C2 * s0 + D5_C3 * p1 + D5_C4 * p2;
  if (j == h - 2) return D5_C0 * m2 + D5_C1 * m1 + D5_C2 * s0 + (D5_C3 + D5_C4) * p1;
  if (j == h - 1) return D5_C0 * m2 + D5_C1 * m1 + (D5_C2 + D5_C3 + D5_C4) * s0;
  return D5_C0 * m2 + D5_C1 * m1 + D5_C2 * s0 + D5_C3 * p1 + D5_C4 * p2;
}

// (The fused TV path has its own warp + derivatives kernel, ofdis_prep.hip; this tiled one serves RGB / tall levels and
// the per-function entry point.)
// RECORDS (round 6, the RGB fused TV path): instead of the 8 * noc row-major planes the kernel writes what tv_fused_kernel
// walks -- per channel an array of 8-float records {Ix, Iz, Ixx, Ixz, Iy, Ixy, Iyz, Iyy} in the diag layout (ofdis_dev.h:
// diag_index; all zero where the warp's mask is zero, see ofdis_fused.h), channel c's array after channel c-1's for all frames
// of the launch, and the (wx, wy) records of the same layout -- staged per tile in LDS and written with the rotated
// enumeration of tv_system_kernel: a wavefront's lanes walk anti-diagonals of the tile, i.e. runs of consecutive records.
template <bool PADDED, bool RECORDS>
__global__ __launch_bounds__(256) void derivatives_kernel(const DerivArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * DA_H * DA_W + 2 * DX_H * DX_W + (RECORDS ? DT_H * DT_W * 10 : 0)];
  float* rec_t = lds + 2 * DA_H * DA_W + 2 * DX_H * DX_W;  // [pixel of the tile][8]
  float* wrec_t = rec_t + DT_H * DT_W * 8;                 // [pixel of the tile][2]
  float* avg_t = lds;
  float* iz_t = avg_t + DA_H * DA_W;
  float* ix_t = iz_t + DA_H * DA_W;
  float* iy_t = ix_t + DX_H * DX_W;
  const int w = a.t.w, h = a.t.h, noc = a.t.noc;
  const int npx = w * h;
  const int tiles_x = (w + DT_W - 1) / DT_W;
  int frame, tile;
  xcd_frame_map(blockIdx.x, tiles_x * ((h + DT_H - 1) / DT_H), a.t.nframes, frame, tile);
  if (frame >= a.t.nframes) return;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int x0 = tx * DT_W, y0 = ty * DT_H;
  const int tid = threadIdx.x;

  for (int c = 0; c < noc; ++c) {
    // stage 0: avg = 0.5*(im2w + im1), Iz = im2w - im1 on tile + halo 4, at border-clamped coordinates
    for (int n = tid; n < DA_H * DA_W; n += 256) {
      const int qy = n / DA_W, qx = n - qy * DA_W;
      const int y = clampi(y0 + qy - 4, 0, h - 1), x = clampi(x0 + qx - 4, 0, w - 1);
      float i1;
      if (PADDED)
        i1 = a.im1[(size_t)frame * a.tmp_w * a.tmp_h * noc + ((size_t)(y + a.pad) * a.tmp_w + x + a.pad) * noc + c];
      else
        i1 = a.im1[((size_t)frame * noc + c) * npx + y * w + x];
      const float i2 = a.im2w[((size_t)frame * noc + c) * npx + y * w + x];
      avg_t[n] = 0.5f * (i2 + i1);
      iz_t[n] = i2 - i1;
    }
    __syncthreads();
    // stage 1: Ix = d/dx avg, Iy = d/dy avg on tile + halo 2.  An entry outside the image holds the
    // value of the nearest image pixel (that is what the second convolution's replication reads).
    for (int n = tid; n < DX_H * DX_W; n += 256) {
      const int qy = n / DX_W, qx = n - qy * DX_W;
      const int y = clampi(y0 + qy - 2, 0, h - 1), x = clampi(x0 + qx - 2, 0, w - 1);
      const int ay = y - y0 + 4, ax = x - x0 + 4;  // position inside the avg tile
      ix_t[n] = h5(avg_t, DA_W, ay, ax);
      iy_t[n] = v5(avg_t, DA_W, ay, ax, y, h);
    }
    __syncthreads();
    // stage 2: the eight derivative values of this thread's (two) pixels
    constexpr int NPIX = DT_W * DT_H / 256;
    const int qx = tid % DT_W;
#pragma unroll
    for (int k = 0; k < NPIX; ++k) {
      const int ry = tid / DT_W + k * (256 / DT_W);
      const int yc = clampi(y0 + ry, 0, h - 1);  // rows/columns beyond the image are computed on clamped
      const int ey = ry + 2, ex = qx + 2;        // coordinates and never stored
      const int ay = ry + 4, ax = qx + 4;
      float res[8];
      res[0] = ix_t[ey * DX_W + ex];
      res[1] = iy_t[ey * DX_W + ex];
      res[2] = iz_t[ay * DA_W + ax];
      res[3] = h5(ix_t, DX_W, ey, ex);
      res[4] = v5(ix_t, DX_W, ey, ex, yc, h);
      res[5] = v5(iy_t, DX_W, ey, ex, yc, h);
      res[6] = h5(iz_t, DA_W, ay, ax);
      res[7] = v5(iz_t, DA_W, ay, ax, yc, h);
      const int y = y0 + ry, x = x0 + qx;
      if constexpr (RECORDS) {
        const bool in = y < h && x < w;
        const float m = in ? a.mask[(size_t)frame * npx + y * w + x] : 0.0f;
        f4* st = reinterpret_cast<f4*>(rec_t + (ry * DT_W + qx) * 8);
        const f4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
        st[0] = m != 0.0f ? (f4){res[0], res[2], res[3], res[6]} : zero;  // Ix, Iz, Ixx, Ixz
        st[1] = m != 0.0f ? (f4){res[1], res[4], res[7], res[5]} : zero;  // Iy, Ixy, Iyz, Iyy
        if (c == 0) {
          const size_t o = (size_t)frame * npx + (in ? y * w + x : 0);
          wrec_t[(ry * DT_W + qx) * 2] = a.wx[o];
          wrec_t[(ry * DT_W + qx) * 2 + 1] = a.wy[o];
        }
      } else if (y < h && x < w) {
        float* out = a.out + ((size_t)frame * 8 * noc + c) * npx + y * w + x;
        const size_t ks = (size_t)noc * npx;
#pragma unroll
        for (int q = 0; q < 8; ++q) out[q * ks] = res[q];
      }
    }
    __syncthreads();
    if constexpr (RECORDS) {
      float* recs = a.rec_d8 + ((size_t)c * a.t.nframes + frame) * npx * 8;
      for (int n = tid; n < DT_H * DT_W; n += 256) {
        const int ry = n % DT_H, r = n / DT_H;
        const int rx = (r - ry) & (DT_W - 1);
        const int y = y0 + ry, x = x0 + rx;
        if (y < h && x < w) {
          const size_t idx = diag_index(x, y, w, h);
          const f4* st = reinterpret_cast<const f4*>(rec_t + (ry * DT_W + rx) * 8);
          f4* out = reinterpret_cast<f4*>(recs + idx * 8);
          out[0] = st[0];
          out[1] = st[1];
          if (c == 0) {
            typedef float f2v __attribute__((ext_vector_type(2)));
            *reinterpret_cast<f2v*>(a.rec_w + ((size_t)frame * npx + idx) * 2) =
                *reinterpret_cast<const f2v*>(wrec_t + (ry * DT_W + rx) * 2);
          }
        }
      }
      __syncthreads();
    }
  }
}

hipError_t launch_derivatives(const DerivArgs& a, hipStream_t s) {
  if (a.t.h < 4) return hipErrorInvalidValue;  // the reference's vertical filter reads rows 0..3
  const int tiles = ((a.t.w + DT_W - 1) / DT_W) * ((a.t.h + DT_H - 1) / DT_H);
  const dim3 g(((a.t.nframes + 7) / 8) * 8 * tiles), b(256);
  if (a.rec_d8) {
    if (!a.rec_w || !a.mask || !a.wx || !a.wy) return hipErrorInvalidValue;
    if (a.im1_padded) hipLaunchKernelGGL((derivatives_kernel<true, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((derivatives_kernel<false, true>), g, b, 0, s, a);
  } else if (a.im1_padded) hipLaunchKernelGGL((derivatives_kernel<true, false>), g, b, 0, s, a);
  else hipLaunchKernelGGL((derivatives_kernel<false, false>), g, b, 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------ TV system
#ifndef OFDIS_ST_H
#define OFDIS_ST_H 32
#endif
constexpr int ST_W = 32, ST_H = OFDIS_ST_H;       // output tile
constexpr int SU_W = ST_W + 4, SU_H = ST_H + 4;  // wx/wy/du/dv tile (halo 2)
constexpr int SS_W = ST_W + 2, SS_H = ST_H + 2;  // smoothness tile (halo 1)
constexpr int ST_PIX = ST_W * ST_H / 256;        // pixels per thread

// One 32x32 image tile per block.  du/dv arrive in the solver's diag layout and the seven outputs
// leave in it; both transpositions go through LDS with a "rotated" enumeration (consecutive lanes
// walk an anti-diagonal of the tile: x-1, y+1), which makes the global side contiguous runs of up
// to 32 floats and the LDS side conflict-free (pitch even => (pitch-1) odd).
__global__ __launch_bounds__(256) void tv_system_kernel(const SystemArgs a) {
  constexpr int IN_FLOATS = 4 * SU_H * SU_W + SS_H * SS_W;
  constexpr int OUT_FLOATS = 7 * ST_H * ST_W;
  __shared__ float lds[IN_FLOATS > OUT_FLOATS ? IN_FLOATS : OUT_FLOATS];
  float* du_t = lds;
  float* dv_t = du_t + SU_H * SU_W;
  float* wx_t = dv_t + SU_H * SU_W;
  float* wy_t = wx_t + SU_H * SU_W;
  float* s_t = wy_t + SU_H * SU_W;
  const int w = a.t.w, h = a.t.h, noc = a.t.noc;
  const int npx = w * h;
  const int tiles_x = (w + ST_W - 1) / ST_W;
  int frame, tile;
  xcd_frame_map(blockIdx.x, tiles_x * ((h + ST_H - 1) / ST_H), a.t.nframes, frame, tile);
  if (frame >= a.t.nframes) return;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int x0 = tx * ST_W, y0 = ty * ST_H;
  const int tid = threadIdx.x;
  const size_t fo = (size_t)frame * npx;

  // stage 0a: wx, wy (row-major) on tile + halo 2 at border-clamped coordinates
  for (int n = tid; n < SU_H * SU_W; n += 256) {
    const int qy = n / SU_W, qx = n - qy * SU_W;
    const int y = clampi(y0 + qy - 2, 0, h - 1), x = clampi(x0 + qx - 2, 0, w - 1);
    const size_t o = fo + y * w + x;
    wx_t[n] = a.wx[o];
    wy_t[n] = a.wy[o];
  }
  // stage 0b: du, dv (diag layout) on the same region, rotated enumeration
  for (int n = tid; n < SU_H * SU_W; n += 256) {
    const int qy = n % SU_H, r = n / SU_H;
    int qx = r - qy;
    if (qx < 0) qx += SU_W;
    const int y = clampi(y0 + qy - 2, 0, h - 1), x = clampi(x0 + qx - 2, 0, w - 1);
    const size_t o = fo + diag_index(x, y, w, h);
    du_t[qy * SU_W + qx] = a.du[o];
    dv_t[qy * SU_W + qx] = a.dv[o];
  }
  __syncthreads();
  // stage 1: smoothness = quarter_alpha / sqrt(|grad uu|^2 + |grad vv|^2 + eps) on tile + halo 1 with
  // uu = wx + du, vv = wy + dv (refine_variational.cpp:210-216; opticalflow_aux.c:128-140).
  // Only in-image entries are ever read back.
  for (int n = tid; n < SS_H * SS_W; n += 256) {
    const int qy = n / SS_W, qx = n - qy * SS_W;
    const int y = y0 + qy - 1, x = x0 + qx - 1;
    float sval = 0.0f;
    if (y >= 0 && y < h && x >= 0 && x < w) {
      const int c = (qy + 1) * SU_W + qx + 1;  // position in the halo-2 tiles
      auto uu = [&](int o) { return wx_t[c + o] + du_t[c + o]; };
      auto vv = [&](int o) { return wy_t[c + o] + dv_t[c + o]; };
      const float ux = D3_C0 * uu(-1) + D3_C1 * uu(0) + D3_C2 * uu(1);   // image.c:436-464
      const float vx = D3_C0 * vv(-1) + D3_C1 * vv(0) + D3_C2 * vv(1);
      float uy, vy;                                                    // image.c:376-399
      if (y == 0) {
        uy = (D3_C0 + D3_C1) * uu(0) + D3_C2 * uu(SU_W);
        vy = (D3_C0 + D3_C1) * vv(0) + D3_C2 * vv(SU_W);
      } else if (y == h - 1) {
        uy = D3_C0 * uu(-SU_W) + (D3_C1 + D3_C2) * uu(0);
        vy = D3_C0 * vv(-SU_W) + (D3_C1 + D3_C2) * vv(0);
      } else {
        uy = D3_C0 * uu(-SU_W) + D3_C1 * uu(0) + D3_C2 * uu(SU_W);
        vy = D3_C0 * vv(-SU_W) + D3_C1 * vv(0) + D3_C2 * vv(SU_W);
      }
      sval = a.quarter_alpha / sqrtf(ux * ux + uy * uy + vx * vx + vy * vy + EPS_SMOOTH);
    }
    s_t[n] = sval;
  }
  __syncthreads();
  // stage 2: per pixel -- data term, then the Laplacian of the CURRENT flow (wx, wy) subtracted from
  // the right-hand side in the reference's scatter order: -left, +right, -top, +bottom
  // (opticalflow_aux.c:172-199).  Results stay in registers until the input tiles are dead.
  float res[ST_PIX][7];
  const int qx = tid % ST_W;
#pragma unroll
  for (int k = 0; k < ST_PIX; ++k) {
    const int ry = tid / ST_W + k * (256 / ST_W);
    const int y = y0 + ry, x = x0 + qx;
#pragma unroll
    for (int q = 0; q < 7; ++q) res[k][q] = 0.0f;
    if (y >= h || x >= w) continue;
    const size_t o = fo + y * w + x;
    const int sy = ry + 1, sx = qx + 1;  // in s tile
    const int uc = (ry + 2) * SU_W + qx + 2;  // in halo-2 tiles
    const float sc = s_t[sy * SS_W + sx];
    const float sh_c = (x < w - 1) ? sc + s_t[sy * SS_W + sx + 1] : 0.0f;      // opticalflow_aux.c:150-154
    const float sv_c = (y < h - 1) ? sc + s_t[(sy + 1) * SS_W + sx] : 0.0f;    // :158-163
    float a11, a12, a22, b1, b2;
    const float* dbase = a.derivs + (size_t)frame * 8 * noc * npx + (size_t)y * w + x;
    auto D = [&](int kk, int c) { return dbase[((size_t)kk * noc + c) * npx]; };
    data_term(D, noc, a.mask[o], du_t[uc], dv_t[uc], a.half_delta_over3, a.half_gamma_over3, a11, a12, a22, b1, b2);
    const float wxc = wx_t[uc], wyc = wy_t[uc];
    if (x > 0) {
      const float sh_l = s_t[sy * SS_W + sx - 1] + sc;
      b1 -= sh_l * (wxc - wx_t[uc - 1]);
      b2 -= sh_l * (wyc - wy_t[uc - 1]);
    }
    if (x < w - 1) {
      b1 += sh_c * (wx_t[uc + 1] - wxc);
      b2 += sh_c * (wy_t[uc + 1] - wyc);
    }
    if (y > 0) {
      const float sv_t = s_t[(sy - 1) * SS_W + sx] + sc;
      b1 -= sv_t * (wxc - wx_t[uc - SU_W]);
      b2 -= sv_t * (wyc - wy_t[uc - SU_W]);
    }
    if (y < h - 1) {
      b1 += sv_c * (wx_t[uc + SU_W] - wxc);
      b2 += sv_c * (wy_t[uc + SU_W] - wyc);
    }
    res[k][0] = a11; res[k][1] = a12; res[k][2] = a22; res[k][3] = b1; res[k][4] = b2; res[k][5] = sh_c; res[k][6] = sv_c;
  }
  __syncthreads();  // input tiles dead: reuse the LDS as the output staging area [plane][ry][qx]
#pragma unroll
  for (int k = 0; k < ST_PIX; ++k) {
    const int ry = tid / ST_W + k * (256 / ST_W);
#pragma unroll
    for (int q = 0; q < 7; ++q) lds[(q * ST_H + ry) * ST_W + qx] = res[k][q];
  }
  __syncthreads();
  // stage 3: write the seven planes in diag layout, rotated enumeration (lane -> x-1, y+1)
  for (int n = tid; n < ST_H * ST_W; n += 256) {
    const int ry = n % ST_H, r = n / ST_H;
    const int rx = (r - ry) & (ST_W - 1);
    const int y = y0 + ry, x = x0 + rx;
    if (y < h && x < w) {
      float* out = a.sys + (size_t)frame * 7 * npx + diag_index(x, y, w, h);
#pragma unroll
      for (int q = 0; q < 7; ++q) out[(size_t)q * npx] = lds[(q * ST_H + ry) * ST_W + rx];
    }
  }
}

hipError_t launch_tv_system(const SystemArgs& a, hipStream_t s) {
  const int tiles = ((a.t.w + ST_W - 1) / ST_W) * ((a.t.h + ST_H - 1) / ST_H);
  hipLaunchKernelGGL(tv_system_kernel, dim3(((a.t.nframes + 7) / 8) * 8 * tiles), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------ finish / split
// uu = wx + du, vv = wy + dv -> AoS flow.  du/dv live in the solver's diag layout: a 32x32 tile is
// gathered through LDS with the rotated enumeration (contiguous runs on the global side), then
// written row-major as float2.
__global__ __launch_bounds__(256) void tv_finish_kernel(int w, int h, int nframes, const float* wx, const float* wy,
                                                        const float* du, const float* dv, float2* flow) {
  constexpr int TW = 32, TH = 32;
  __shared__ float du_t[TH * TW];
  __shared__ float dv_t[TH * TW];
  const int npx = w * h;
  const int tiles_x = (w + TW - 1) / TW;
  int frame, tile;
  xcd_frame_map(blockIdx.x, tiles_x * ((h + TH - 1) / TH), nframes, frame, tile);
  if (frame >= nframes) return;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int x0 = tx * TW, y0 = ty * TH;
  const size_t fo = (size_t)frame * npx;
  for (int n = threadIdx.x; n < TH * TW; n += 256) {
    const int ry = n % TH, r = n / TH;
    const int rx = (r - ry) & (TW - 1);
    const int y = y0 + ry, x = x0 + rx;
    if (y < h && x < w) {
      const size_t o = fo + diag_index(x, y, w, h);
      du_t[ry * TW + rx] = du[o];
      dv_t[ry * TW + rx] = dv[o];
    }
  }
  __syncthreads();
  const int qx = threadIdx.x % TW;
  for (int ry = threadIdx.x / TW; ry < TH; ry += 256 / TW) {
    const int y = y0 + ry, x = x0 + qx;
    if (y < h && x < w) {
      const size_t o = fo + (size_t)y * w + x;
      flow[o] = make_float2(wx[o] + du_t[ry * TW + qx], wy[o] + dv_t[ry * TW + qx]);
    }
  }
}
hipError_t launch_tv_finish(const TvGeom& t, const float* wx, const float* wy, const float* du, const float* dv,
                            float* flow_aos, hipStream_t s) {
  const int tiles = ((t.w + 31) / 32) * ((t.h + 31) / 32);
  hipLaunchKernelGGL(tv_finish_kernel, dim3(((t.nframes + 7) / 8) * 8 * tiles), dim3(256), 0, s, t.w, t.h, t.nframes, wx, wy,
                     du, dv, reinterpret_cast<float2*>(flow_aos));
  return hipGetLastError();
}

// The fused path's variant: the densified flow (wx, wy) already sits in the output array, du / dv are the fused kernel's
// sdiag records (ofdis_dev.h): uu = wx + du, vv = wy + dv in place.
__global__ __launch_bounds__(256) void tv_finish_records_kernel(int w, int h, int nframes, int S, float2* flow,
                                                                const float2* __restrict__ uv) {
  constexpr int TW = 32, TH = 32;
  __shared__ float2 t[TH * TW];
  const int npx = w * h;
  const int tiles_x = (w + TW - 1) / TW;
  int frame, tile;
  xcd_frame_map(blockIdx.x, tiles_x * ((h + TH - 1) / TH), nframes, frame, tile);
  if (frame >= nframes) return;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int x0 = tx * TW, y0 = ty * TH;
  for (int n = threadIdx.x; n < TH * TW; n += 256) {  // rotated enumeration: contiguous runs of a diag row
    const int ry = n % TH, r = n / TH;
    const int rx = (r - ry) & (TW - 1);
    const int y = y0 + ry, x = x0 + rx;
    if (y < h && x < w) t[ry * TW + rx] = uv[sdiag_index(frame, x, y, w, h, S)];
  }
  __syncthreads();
  const int qx = threadIdx.x % TW;
  for (int ry = threadIdx.x / TW; ry < TH; ry += 256 / TW) {
    const int y = y0 + ry, x = x0 + qx;
    if (y < h && x < w) {
      const size_t o = (size_t)frame * npx + (size_t)y * w + x;
      const float2 f = flow[o], d = t[ry * TW + qx];
      flow[o] = make_float2(f.x + d.x, f.y + d.y);
    }
  }
}
hipError_t launch_tv_finish_records(const TvGeom& t, float* flow_aos, const float* uv, int S, hipStream_t s) {
  const int tiles = ((t.w + 31) / 32) * ((t.h + 31) / 32);
  hipLaunchKernelGGL(tv_finish_records_kernel, dim3(((t.nframes + 7) / 8) * 8 * tiles), dim3(256), 0, s, t.w, t.h, t.nframes,
                     S, reinterpret_cast<float2*>(flow_aos), reinterpret_cast<const float2*>(uv));
  return hipGetLastError();
}

// row-major <-> diag conversion of whole planes (per-function entry points / tests only)
template <bool TO_DIAG>
__global__ __launch_bounds__(256) void diag_convert_kernel(const float* src, float* dst, int w, int h, long long nplanes) {
  const int npx = w * h;
  const long long total = (long long)npx * nplanes;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long pl = i / npx;
    const int o = (int)(i - pl * npx);
    const int y = o / w, x = o - y * w;
    const size_t dg = (size_t)pl * npx + diag_index(x, y, w, h);
    if (TO_DIAG) dst[dg] = src[i]; else dst[i] = src[dg];
  }
}
hipError_t launch_to_diag(const float* src_rm, float* dst_diag, int w, int h, long long nplanes, hipStream_t s) {
  long long blocks = ((long long)w * h * nplanes + 255) / 256;
  if (blocks > (1 << 20)) blocks = 1 << 20;
  hipLaunchKernelGGL(diag_convert_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, src_rm, dst_diag, w, h, nplanes);
  return hipGetLastError();
}
hipError_t launch_from_diag(const float* src_diag, float* dst_rm, int w, int h, long long nplanes, hipStream_t s) {
  long long blocks = ((long long)w * h * nplanes + 255) / 256;
  if (blocks > (1 << 20)) blocks = 1 << 20;
  hipLaunchKernelGGL(diag_convert_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, src_diag, dst_rm, w, h, nplanes);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void flow_split_kernel(long long total, const float2* flow, float* wx, float* wy) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const float2 f = flow[i];
    wx[i] = f.x;
    wy[i] = f.y;
  }
}
hipError_t launch_flow_split(const TvGeom& t, const float* flow_aos, float* wx, float* wy, hipStream_t s) {
  const long long total = (long long)t.w * t.h * t.nframes;
  long long blocks = (total + 255) / 256;
  if (blocks > (1 << 20)) blocks = 1 << 20;
  hipLaunchKernelGGL(flow_split_kernel, dim3((unsigned)blocks), dim3(256), 0, s, total,
                     reinterpret_cast<const float2*>(flow_aos), wx, wy);
  return hipGetLastError();
}

}  // namespace OFDIS_KNS
}  // namespace ofdis
