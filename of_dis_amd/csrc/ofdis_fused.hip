// ofdis_fused.hip -- one TV fixed-point iteration in ONE kernel (gray, level height <= 64):
//     compute_smoothness + compute_data + 2 x sub_laplacian  (opticalflow_aux.c:123-199, 310-438)
//     -> sor_coupled                                          (solver.c:77-421)
//
// The system coefficients of a pixel depend only on un-solved quantities (wx, wy, the du/dv of BEFORE
// this solver call, the level's derivatives), so they can be produced in any order -- in particular in
// the order the wavefront SOR consumes them.  This kernel is the SOR of ofdis_sor.hip (lane = image row,
// step t -> column t - j, NS software-pipelined sweeps) with its nine row loads replaced by a producer
// that runs a few diagonals ahead in the same wavefront and hands each pixel's seven coefficients over
// IN REGISTERS.  Per pixel and iteration HBM sees 13 row reads (8 derivatives, mask, wx, wy, du, dv) and the
// final du, dv: 60 B instead of 141 B for the tv_system + sor pair (the 7-plane system never exists in
// memory), and the producer's arithmetic fills the issue slots the SOR's dependency chains leave empty.
//
// In diag coordinates (row d = (x+y) mod w, lane = y) the 4-neighbourhood is row-local:
//     (x+1,y) -> (d+1, lane)     (x-1,y) -> (d-1, lane)     (x,y+1) -> (d+1, lane+1)     (x,y-1) -> (d-1, lane-1)
// so every stencil (flow gradients, smoothness sums, Laplacian) needs only rows d-1, d, d+1 of a plane
// and one DPP lane shift.  Rows live in small register rings with static indices (loop unrolled by 6):
//     W   (wx,wy,du,dv)  row t+5 loaded at step t, last used at step t+5 later     ring 6
//     D   (8 derivs+mask) row t+3 loaded at step t, used at step t+2               ring 3
//     uu,vv = wx+du, wy+dv of row t+3 ; s = smoothness of row t+2                  ring 3
//     slot (system of the pixel row t+1, consumed by the sweeps at t+1, t+3, t+5)  ring 6
// du/dv are read (old values, rows >= t+1) strictly ahead of where the last sweep stores (row t-4).
//
// All n_inner fixed-point iterations of a level run inside ONE launch, back to back per lane: the diag rows
// wrap (row = t mod w), so when a lane has finished column w-1 of iteration k it continues with column 0 of
// iteration k+1 on the next step.  That is legal because everything iteration k+1 needs around a pixel
// (du of iteration k within a radius of two pixels) was finalised w-9 or more steps earlier, and it removes
// the fill/drain bubble of the skewed sweep from every iteration but the first and last:
// n_inner*w + h steps instead of n_inner*(w + h).
//
// Border rules: horizontal neighbours are clamped exactly as the reference's shifted row copies
// (image.c:436-464); for the 3-tap vertical filter the clamped form c0*s0 + c1*s0 + c2*s1 has the same
// value as the reference's folded (c0+c1)*s0 + c2*s1 because c1 = -0 (image.c:376-399).
#include "ofdis_kernels.h"
#include "ofdis_tvmath.h"

namespace ofdis {

struct FSlot {
  float a11, a12, a22, b1, b2, sh, sv;  // system of pixel (j, tau - j); a** become the block inverse at step tau
  float dur, dvr;                       // old du,dv of the right neighbour (row tau+1)
  float hl, vt;                         // left / top edge weights
};
struct FRow {
  float wx, wy, du, dv;
};
struct FDer {
  float d[8];
  float m;
};

template <int NS>
__global__ __launch_bounds__(256) void tv_fused_kernel(const FusedArgs a, const int R) {
  constexpr int U = 6;
  static_assert(2 * (NS - 1) + 1 < U, "slot ring too small for this many pipelined sweeps");
  const int w = a.t.w, h = a.t.h;
  const int npx = w * h;
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int G = 64 / R;  // frames per wavefront
  if (wid * G >= a.t.nframes) return;  // whole wave idle (uniform)
  int f = wid * G + lane / R;
  const int jr = lane % R;
  const bool row_ok = (f < a.t.nframes) && (jr < h);
  if (f >= a.t.nframes) f = a.t.nframes - 1;
  const int j = jr < h ? jr : h - 1;
  const bool has_top = j > 0, has_bot = j < h - 1;
  const float omega = a.omega, qa = a.quarter_alpha, hd3 = a.half_delta_over3, hg3 = a.half_gamma_over3;

  const float* __restrict__ derp = a.derivs + (size_t)f * 8 * npx + j;
  const float* __restrict__ mskp = a.mask + (size_t)f * npx + j;
  const float* __restrict__ wxp = a.wx + (size_t)f * npx + j;
  const float* __restrict__ wyp = a.wy + (size_t)f * npx + j;
  float* __restrict__ dup = a.du + (size_t)f * npx + j;
  float* __restrict__ dvp = a.dv + (size_t)f * npx + j;

  FRow W[6];
  FDer D[3];
  float uu[3], vv[3], sm[3];
  FSlot slot[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) { W[r] = FRow{0, 0, 0, 0}; slot[r] = FSlot{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    uu[r] = vv[r] = sm[r] = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) D[r].d[q] = 0.0f;
    D[r].m = 0.0f;
  }
  float ru[NS], rv[NS], ru2[NS], rv2[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) { ru[s] = rv[s] = ru2[s] = rv2[s] = 0.0f; }

  auto wrap = [&](int r) { r %= w; return r < 0 ? r + w : r; };
  auto next_row = [&](int r) { return (r + 1 == w) ? 0 : r + 1; };
  auto load_w = [&](FRow& r, int drow) {
    const int o = drow * h;
    r.wx = wxp[o]; r.wy = wyp[o]; r.du = dup[o]; r.dv = dvp[o];
  };
  auto load_d = [&](FDer& r, int drow) {
    const int o = drow * h;
#pragma unroll
    for (int q = 0; q < 8; ++q) r.d[q] = derp[(size_t)q * npx + o];
    r.m = mskp[o];
  };

  // ring index of diag row rho is (rho + 3) mod ring size; the loop variable is k = t + 3, u = k % 6,
  // so row t + c sits at index (u + c) % size.
  // prologue: W rows -1, 0, 1 (indices 2, 3, 4)
  load_w(W[2], wrap(-1));
  load_w(W[3], wrap(0));
  load_w(W[4], wrap(1));
  int rowW = wrap(2);   // next W row to load (row t+5 at t = -3)
  int rowD = wrap(0);   // next D row to load (row t+3 at t = -3)
  int srow = wrap(-3 - 2 * (NS - 1));  // row finished by the last sweep at step t = -3
  int xq = wrap(-3 - j);               // this lane's x on diag row t (per lane)

  const int wtot = a.n_inner * w;  // columns per lane over all iterations
  const int tend = (wtot - 1) + (h - 1) + 2 * (NS - 1);
  for (int k0 = 0; k0 <= tend + 3; k0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = k0 + u - 3;  // up to U-1 steps past tend are executed: every pixel is then out of range
      // x of this lane on rows t+1 and t+2
      const int x1 = (xq + 1 == w) ? 0 : xq + 1;
      const int x2 = (x1 + 1 == w) ? 0 : x1 + 1;
      // ---- (1) loads: W row t+5, D row t+3
      load_w(W[(u + 5) % 6], rowW);
      rowW = next_row(rowW);
      load_d(D[u % 3], rowD);
      rowD = next_row(rowD);
      // ---- (2) uu, vv of row t+3 (refine_variational.cpp:210-216: uu = wx + du of before this call)
      {
        const FRow& r = W[(u + 3) % 6];
        uu[u % 3] = r.wx + r.du;
        vv[u % 3] = r.wy + r.dv;
      }
      // ---- (3) smoothness of row t+2 (opticalflow_aux.c:128-140)
      {
        const float uc = uu[(u + 2) % 3], vc = vv[(u + 2) % 3];
        float ul = uu[(u + 1) % 3], vl = vv[(u + 1) % 3];                  // (x-1, y)
        float ur = uu[u % 3], vr = vv[u % 3];                              // (x+1, y): row t+3
        float ut = wave_from_prev(uu[(u + 1) % 3]), vt = wave_from_prev(vv[(u + 1) % 3]);  // (x, y-1)
        float ub = wave_from_next(uu[u % 3]), vb = wave_from_next(vv[u % 3]);              // (x, y+1)
        if (x2 == 0) { ul = uc; vl = vc; }
        if (x2 == w - 1) { ur = uc; vr = vc; }
        if (!has_top) { ut = uc; vt = vc; }
        if (!has_bot) { ub = uc; vb = vc; }
        const float ux = D3_C0 * ul + D3_C1 * uc + D3_C2 * ur;
        const float vx = D3_C0 * vl + D3_C1 * vc + D3_C2 * vr;
        const float uy = D3_C0 * ut + D3_C1 * uc + D3_C2 * ub;
        const float vy = D3_C0 * vt + D3_C1 * vc + D3_C2 * vb;
        sm[(u + 2) % 3] = qa / sqrtf(ux * ux + uy * uy + vx * vx + vy * vy + EPS_SMOOTH);
      }
      // ---- (4) system of pixel row tau = t+1 (opticalflow_aux.c:150-163, 172-199, 342-427)
      {
        const float sc = sm[(u + 1) % 3];
        const float s_r = sm[(u + 2) % 3], s_l = sm[u % 3];
        const float s_d = wave_from_next(sm[(u + 2) % 3]), s_u = wave_from_prev(sm[u % 3]);
        const float sh_c = (x1 < w - 1) ? sc + s_r : 0.0f;
        const float sv_c = has_bot ? sc + s_d : 0.0f;
        const FRow& rc = W[(u + 1) % 6];
        const FRow& rm = W[u % 6];        // row tau-1
        const FRow& rp = W[(u + 2) % 6];  // row tau+1
        const FDer& dd = D[(u + 1) % 3];
        float a11, a12, a22, b1, b2;
        auto Df = [&](int kk, int) { return dd.d[kk]; };
        data_term(Df, 1, dd.m, rc.du, rc.dv, hd3, hg3, a11, a12, a22, b1, b2);
        const float wx_u = wave_from_prev(rm.wx), wy_u = wave_from_prev(rm.wy);
        const float wx_d = wave_from_next(rp.wx), wy_d = wave_from_next(rp.wy);
        if (x1 > 0) {
          const float sh_l = s_l + sc;
          b1 -= sh_l * (rc.wx - rm.wx);
          b2 -= sh_l * (rc.wy - rm.wy);
        }
        if (x1 < w - 1) {
          b1 += sh_c * (rp.wx - rc.wx);
          b2 += sh_c * (rp.wy - rc.wy);
        }
        if (has_top) {
          const float sv_t = s_u + sc;
          b1 -= sv_t * (rc.wx - wx_u);
          b2 -= sv_t * (rc.wy - wy_u);
        }
        if (has_bot) {
          b1 += sv_c * (wx_d - rc.wx);
          b2 += sv_c * (wy_d - rc.wy);
        }
        FSlot& o = slot[(u + 1) % 6];
        o.a11 = a11; o.a12 = a12; o.a22 = a22; o.b1 = b1; o.b2 = b2; o.sh = sh_c; o.sv = sv_c;
        o.dur = rp.du; o.dvr = rp.dv;
      }
      // ---- (5) SOR step t (ofdis_sor.hip): sweep 0 reaches pixel (j, t - j)
      {
        FSlot& c = slot[u % 6];
        const FSlot& p = slot[(u + 5) % 6];
        c.hl = (xq > 0) ? p.sh : 0.0f;  // xq = this lane's column on diag row t
        c.vt = wave_from_prev(p.sv);
        float d = c.hl + c.sh;
        if (has_top) d = d + c.vt;
        if (has_bot) d = d + c.sv;
        const float A11 = c.a22 + d, A22 = c.a11 + d;
        const float det = A11 * A22 - c.a12 * c.a12;
        c.a11 = A11 / det;
        c.a22 = A22 / det;
        c.a12 = c.a12 / (-det);
      }
      float nu[NS], nv[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        int i = xq - 2 * s;  // column of sweep s (wrapped: the lane may already be in the next iteration)
        if (i < 0) i += w;
        const FSlot& c = slot[(u - 2 * s + 12) % 6];
        float ou, ov, rgu, rgv, bu, bv;
        if (s == 0) {
          const FSlot& p = slot[(u + 5) % 6];
          ou = p.dur; ov = p.dvr;
          rgu = c.dur; rgv = c.dvr;
          bu = wave_from_next(c.dur);
          bv = wave_from_next(c.dvr);
        } else {
          ou = ru2[s - 1]; ov = rv2[s - 1];
          rgu = ru[s - 1]; rgv = rv[s - 1];
          bu = wave_from_next(ru[s - 1]);
          bv = wave_from_next(rv[s - 1]);
        }
        if (!(i < w - 1)) { rgu = 0.0f; rgv = 0.0f; }
        const float tu = wave_from_prev(ru[s]), tv = wave_from_prev(rv[s]);
        const float lu = ru[s], lv = rv[s];
        float s1 = c.sh * rgu, s2 = c.sh * rgv;
        if (has_top) { s1 = s1 + c.vt * tu; s2 = s2 + c.vt * tv; }
        if (has_bot) { s1 = s1 + c.sv * bu; s2 = s2 + c.sv * bv; }
        s1 = s1 + c.b1;
        s2 = s2 + c.b2;
        float B1 = s1, B2 = s2;
        if (i > 0) { B1 = c.hl * lu + s1; B2 = c.hl * lv + s2; }
        nu[s] = ou + omega * (c.a11 * B1 + c.a12 * B2 - ou);
        nv[s] = ov + omega * (c.a12 * B1 + c.a22 * B2 - ov);
      }
      {
        const int ig = t - j - 2 * (NS - 1);  // global column index over all iterations
        if (row_ok && ig >= 0 && ig < wtot) {
          dup[srow * h] = nu[NS - 1];
          dvp[srow * h] = nv[NS - 1];
        }
        srow = next_row(srow);
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        ru2[s] = ru[s]; rv2[s] = rv[s];
        ru[s] = nu[s]; rv[s] = nv[s];
      }
      xq = x1;
    }
  }
}

bool tv_fused_supported(const TvGeom& t, int iterations) {
  return t.noc == 1 && t.h >= 2 && t.h <= 64 && t.w >= 16 && iterations >= 1 && iterations <= 3;
}

hipError_t launch_tv_fused(const FusedArgs& a, hipStream_t s) {
  if (!tv_fused_supported(a.t, a.iterations) || a.n_inner < 1) return hipErrorInvalidValue;
  const int h = a.t.h;
  const int R = h <= 16 ? 16 : (h <= 32 ? 32 : 64);
  const int G = 64 / R;
  const int waves = (a.t.nframes + G - 1) / G;
  const int blocks = (waves + 3) / 4;
  switch (a.iterations) {
    case 1: hipLaunchKernelGGL(tv_fused_kernel<1>, dim3(blocks), dim3(256), 0, s, a, R); break;
    case 2: hipLaunchKernelGGL(tv_fused_kernel<2>, dim3(blocks), dim3(256), 0, s, a, R); break;
    default: hipLaunchKernelGGL(tv_fused_kernel<3>, dim3(blocks), dim3(256), 0, s, a, R); break;
  }
  return hipGetLastError();
}

}  // namespace ofdis
