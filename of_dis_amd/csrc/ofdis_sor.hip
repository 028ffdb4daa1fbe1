// ofdis_sor.hip -- sor_coupled (solver.c:77-421) for gfx950: an ORDER-PRESERVING parallel sweep.
//
// The reference's solver is a lexicographic (raster order) block Gauss-Seidel/SOR: pixel (j,i) of
// sweep s uses the already-updated left (j,i-1) and top (j-1,i) neighbours of sweep s and the
// not-yet-updated right (j,i+1) and bottom (j+1,i) neighbours, i.e. their sweep s-1 values.  A
// red-black ordering changes the iteration (even the reference's own two
// lexicographic variants differ by 3e-3 px), so this kernel keeps the reference's data flow exactly
// and extracts parallelism from the dependency DAG instead:
//
//   * all pixels on an anti-diagonal i+j = const are independent      -> lane = image row j,
//     step t processes column i = t - j  (one wavefront per frame for h <= 64);
//   * sweep s+1 may trail sweep s by two diagonals                    -> all NS sweeps of one call
//     run in the same pass, software-pipelined: at step t sweep s is at column t - j - 2s.
//
// Every neighbour value then comes from a register of the same lane (left, right, own) or of the
// adjacent lane (top, bottom: one wave_shr / wave_shl DPP move), so du/dv are read once and written
// once per call and the 2x2 block inverse of the first sweep (solver.c:113-120) never leaves
// registers.  The per-pixel arithmetic is the reference's, operation for operation; the three row
// classes of the reference (first / middle / last line) are selected per lane.
//
// Memory layout ("diag", ofdis_dev.h): the solver's operands -- the 7 system planes and du, dv --
// are stored so that anti-diagonal d = (i+j) mod w is one contiguous row of h floats.  Step t then
// reads row t mod w: 9 fully coalesced row loads per step instead of 9 x 64 scattered cache lines
// (the row-major version of this kernel spent >90 % of its time in those gathers: profiles/r01_a).
// Loads run PD steps ahead of their use through a register ring that also serves as the delay line
// handing each pixel's coefficients from sweep 0 to the trailing sweeps.
#include "ofdis_kernels.h"

namespace ofdis {
namespace OFDIS_KNS {  // the arithmetic contract this file is being compiled for (ofdis_dev.h)

// One ring slot = everything the sweeps need about pixel (j, tau - j), tau = the step at which
// sweep 0 reaches it.  a11/a12/a22 are overwritten by the block inverse when sweep 0 gets there.
struct SorSlot {
  float a11, a12, a22, b1, b2, sh, sv;  // loaded (diag row tau)
  float dur, dvr;                       // initial du,dv of the right neighbour (diag row tau+1)
  float hl, vt;                         // left / top edge weights, filled in at step tau
};

template <int NS, int PD>
__global__ __launch_bounds__(256) void sor_wave_kernel(const SorArgs a, const int R) {
  constexpr int LIFE = (2 * (NS - 1) > 1) ? 2 * (NS - 1) : 1;  // a slot is last read LIFE steps after tau
  constexpr int RS = PD + LIFE + 1;                            // ring size = unroll factor
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
  const float omega = a.omega;

  const float* __restrict__ sysf = a.sys + (size_t)f * 7 * npx + j;  // + k*npx + drow*h
  float* __restrict__ duf = a.du + (size_t)f * npx + j;
  float* __restrict__ dvf = a.dv + (size_t)f * npx + j;

  SorSlot ring[RS];
#pragma unroll
  for (int r = 0; r < RS; ++r) ring[r] = SorSlot{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float ru[NS], rv[NS];    // result of sweep s at the previous step
  float ru2[NS], rv2[NS];  // ... two steps ago (own-old of sweep s+1)
#pragma unroll
  for (int s = 0; s < NS; ++s) { ru[s] = rv[s] = ru2[s] = rv2[s] = 0.0f; }

  // diag row of step tau is tau mod w; rows are tracked incrementally (wave-uniform scalars)
  auto load_slot = [&](SorSlot& sl, int drow, int drow1) {
    const int o = drow * h;
    sl.a11 = sysf[0 * (size_t)npx + o];
    sl.a12 = sysf[1 * (size_t)npx + o];
    sl.a22 = sysf[2 * (size_t)npx + o];
    sl.b1 = sysf[3 * (size_t)npx + o];
    sl.b2 = sysf[4 * (size_t)npx + o];
    sl.sh = sysf[5 * (size_t)npx + o];
    sl.sv = sysf[6 * (size_t)npx + o];
    sl.dur = duf[drow1 * h];
    sl.dvr = dvf[drow1 * h];
  };
  auto next_row = [&](int r) { return (r + 1 == w) ? 0 : r + 1; };

  // prologue: slots of steps 0..PD-1, and the step "-1" right-values (= own values of step 0)
  int lrow = 0;  // diag row of the next step to be loaded
#pragma unroll
  for (int q = 0; q < PD; ++q) {
    load_slot(ring[q], lrow, next_row(lrow));
    lrow = next_row(lrow);
  }
  ring[RS - 1].dur = duf[0];  // pixel (j, 0 - j + ... ) of step -1: diag row 0
  ring[RS - 1].dvr = dvf[0];
  int srow = (w - ((2 * (NS - 1)) % w)) % w;  // diag row of the pixel finished at step 0 is (0 - 2(NS-1)) mod w

  const int tend = (w - 1) + (h - 1) + 2 * (NS - 1);
  for (int t0 = 0; t0 <= tend; t0 += RS) {
#pragma unroll
    for (int u = 0; u < RS; ++u) {
      const int t = t0 + u;  // up to RS-1 steps past tend are executed: every pixel is then out of range
      // prefetch the slot of step t+PD
      load_slot(ring[(u + PD) % RS], lrow, next_row(lrow));
      lrow = next_row(lrow);
      // ---------------- sweep 0 reaches pixel (j, i0): finish its slot (first iteration,
      // solver.c:112-120 / 167-173 / 219-225: edge-weight sum, block inverse)
      {
        const int i0 = t - j;
        SorSlot& c = ring[u];
        const SorSlot& p = ring[(u + RS - 1) % RS];
        c.hl = (i0 > 0) ? p.sh : 0.0f;
        c.vt = wave_from_prev(p.sv);  // sv(j-1, i0): lane j-1 was at column i0 one step ago
        float d = c.hl + c.sh;
        if (has_top) d = d + c.vt;
        if (has_bot) d = d + c.sv;
        const float A11 = c.a22 + d, A22 = c.a11 + d;
        const float det = A11 * A22 - c.a12 * c.a12;
        c.a11 = A11 / det;
        c.a22 = A22 / det;
        c.a12 = c.a12 / (-det);
      }
      // ---------------- all sweeps advance one pixel
      float nu[NS], nv[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int i = t - j - 2 * s;
        const SorSlot& c = ring[(u - 2 * s + 2 * RS) % RS];
        // own / right / bottom: values of the previous sweep (the initial du,dv for sweep 0)
        float ou, ov, rgu, rgv, bu, bv;
        if (s == 0) {
          const SorSlot& p = ring[(u + RS - 1) % RS];
          ou = p.dur; ov = p.dvr;        // what was "right" one step ago
          rgu = c.dur; rgv = c.dvr;
          bu = wave_from_next(c.dur);    // lane j+1 is at column i-1: its "right" is (j+1, i)
          bv = wave_from_next(c.dvr);
        } else {
          ou = ru2[s - 1]; ov = rv2[s - 1];
          rgu = ru[s - 1]; rgv = rv[s - 1];
          bu = wave_from_next(ru[s - 1]);
          bv = wave_from_next(rv[s - 1]);
        }
        if (!(i < w - 1)) { rgu = 0.0f; rgv = 0.0f; }  // f2/f3 are zero from column width-1 on (solver.c:98-99)
        // top / left: values of this sweep
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
      // last sweep's pixel is final
      {
        const int i = t - j - 2 * (NS - 1);
        if (row_ok && i >= 0 && i < w) {
          duf[srow * h] = nu[NS - 1];
          dvf[srow * h] = nv[NS - 1];
        }
        srow = next_row(srow);
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        ru2[s] = ru[s]; rv2[s] = rv[s];
        ru[s] = nu[s]; rv[s] = nv[s];
      }
    }
  }
}

// Level heights above 64 rows: one WORKGROUP per frame, wave k owns rows 64k..64k+63, all waves advance in
// lock step (one barrier per step).  The neighbour values that cross a wave boundary -- the row above lane 0
// and the row below lane 63 -- travel through a double-buffered LDS mailbox written at the end of a step and
// read at the start of the next; everything else is the single-wave kernel.  Up to 16 waves (h <= 1024).
template <int NS, int PD, int MAXT>
__global__ __launch_bounds__(MAXT) void sor_block_kernel(const SorArgs a) {
  constexpr int LIFE = (2 * (NS - 1) > 1) ? 2 * (NS - 1) : 1;
  constexpr int RS = PD + LIFE + 1;
  constexpr int NV = 2 * NS + 2;  // mailbox floats per direction
  __shared__ float mail_top[2][16][NV];  // published by lane 63 of wave k for lane 0 of wave k+1
  __shared__ float mail_bot[2][16][NV];  // published by lane 0 of wave k for lane 63 of wave k-1
  const int w = a.t.w, h = a.t.h;
  const int npx = w * h;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
  const int f = blockIdx.x;
  const int jr = threadIdx.x;
  const bool row_ok = jr < h;
  const int j = jr < h ? jr : h - 1;
  const bool has_top = j > 0, has_bot = j < h - 1;
  const bool first_lane = lane == 0 && wave > 0, last_lane = lane == 63 && wave + 1 < nwaves;
  const float omega = a.omega;

  const float* __restrict__ sysf = a.sys + (size_t)f * 7 * npx + j;
  float* __restrict__ duf = a.du + (size_t)f * npx + j;
  float* __restrict__ dvf = a.dv + (size_t)f * npx + j;

  SorSlot ring[RS];
#pragma unroll
  for (int r = 0; r < RS; ++r) ring[r] = SorSlot{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float ru[NS], rv[NS], ru2[NS], rv2[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) { ru[s] = rv[s] = ru2[s] = rv2[s] = 0.0f; }

  auto load_slot = [&](SorSlot& sl, int drow, int drow1) {
    const int o = drow * h;
    sl.a11 = sysf[0 * (size_t)npx + o];
    sl.a12 = sysf[1 * (size_t)npx + o];
    sl.a22 = sysf[2 * (size_t)npx + o];
    sl.b1 = sysf[3 * (size_t)npx + o];
    sl.b2 = sysf[4 * (size_t)npx + o];
    sl.sh = sysf[5 * (size_t)npx + o];
    sl.sv = sysf[6 * (size_t)npx + o];
    sl.dur = duf[drow1 * h];
    sl.dvr = dvf[drow1 * h];
  };
  auto next_row = [&](int r) { return (r + 1 == w) ? 0 : r + 1; };

  int lrow = 0;
#pragma unroll
  for (int q = 0; q < PD; ++q) {
    load_slot(ring[q], lrow, next_row(lrow));
    lrow = next_row(lrow);
  }
  ring[RS - 1].dur = duf[0];
  ring[RS - 1].dvr = dvf[0];
  int srow = (w - ((2 * (NS - 1)) % w)) % w;

  // mailbox for step 0: nothing has been computed yet (all neighbours' columns are out of range), but
  // sweep 0's "bottom old" of lane 63 is lane 0-of-next-wave's right value of step 0, which is loaded
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < NV; ++q) mail_bot[1][wave][q] = 0.0f;
    mail_bot[1][wave][0] = ring[0].dur;
    mail_bot[1][wave][1] = ring[0].dvr;
  }
  if (lane == 63) {
#pragma unroll
    for (int q = 0; q < NV; ++q) mail_top[1][wave][q] = 0.0f;
  }
  __syncthreads();

  const int tend = (w - 1) + (h - 1) + 2 * (NS - 1);
  for (int t0 = 0; t0 <= tend; t0 += RS) {
#pragma unroll
    for (int u = 0; u < RS; ++u) {
      const int t = t0 + u;
      const int rd = (t + 1) & 1;  // mailbox written at the end of step t-1
      load_slot(ring[(u + PD) % RS], lrow, next_row(lrow));
      lrow = next_row(lrow);
      // neighbour values from the adjacent waves (published at the end of the previous step)
      float top_sv = 0.0f, top_u[NS], top_v[NS], bot_u[NS], bot_v[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) { top_u[s] = top_v[s] = bot_u[s] = bot_v[s] = 0.0f; }
      if (first_lane) {
        top_sv = mail_top[rd][wave - 1][2 * NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) { top_u[s] = mail_top[rd][wave - 1][s]; top_v[s] = mail_top[rd][wave - 1][NS + s]; }
      }
      if (last_lane) {
#pragma unroll
        for (int s = 0; s < NS; ++s) { bot_u[s] = mail_bot[rd][wave + 1][2 * s]; bot_v[s] = mail_bot[rd][wave + 1][2 * s + 1]; }
      }
      {
        const int i0 = t - j;
        SorSlot& c = ring[u];
        const SorSlot& p = ring[(u + RS - 1) % RS];
        c.hl = (i0 > 0) ? p.sh : 0.0f;
        c.vt = wave_from_prev(p.sv);
        if (first_lane) c.vt = top_sv;
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
        const int i = t - j - 2 * s;
        const SorSlot& c = ring[(u - 2 * s + 2 * RS) % RS];
        float ou, ov, rgu, rgv, bu, bv;
        if (s == 0) {
          const SorSlot& p = ring[(u + RS - 1) % RS];
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
        if (last_lane) { bu = bot_u[s]; bv = bot_v[s]; }
        if (!(i < w - 1)) { rgu = 0.0f; rgv = 0.0f; }
        float tu = wave_from_prev(ru[s]), tv = wave_from_prev(rv[s]);
        if (first_lane) { tu = top_u[s]; tv = top_v[s]; }
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
        const int i = t - j - 2 * (NS - 1);
        if (row_ok && i >= 0 && i < w) {
          duf[srow * h] = nu[NS - 1];
          dvf[srow * h] = nv[NS - 1];
        }
        srow = next_row(srow);
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        ru2[s] = ru[s]; rv2[s] = rv[s];
        ru[s] = nu[s]; rv[s] = nv[s];
      }
      // publish for step t+1 (buffer t & 1)
      const int wr = t & 1;
      if (lane == 63) {
#pragma unroll
        for (int s = 0; s < NS; ++s) { mail_top[wr][wave][s] = ru[s]; mail_top[wr][wave][NS + s] = rv[s]; }
        mail_top[wr][wave][2 * NS] = ring[u].sv;  // slot of step t: the next step's top weight
      }
      if (lane == 0) {
        // bottom-old of sweep s at step t+1: sweep 0 -> this lane's right value of step t+1; sweep s>0 -> its
        // sweep s-1 result of step t
        const SorSlot& nx = ring[(u + 1) % RS];
        mail_bot[wr][wave][0] = nx.dur;
        mail_bot[wr][wave][1] = nx.dvr;
#pragma unroll
        for (int s = 1; s < NS; ++s) { mail_bot[wr][wave][2 * s] = ru[s - 1]; mail_bot[wr][wave][2 * s + 1] = rv[s - 1]; }
      }
      // Only the LDS mailboxes cross wavefronts: drain the LDS counter, not vmcnt -- __syncthreads() would also wait for
      // the operand rows that are deliberately requested PD steps ahead (measured: -2 % on a 1080p pair; the step is
      // bound by the lock step of up to 16 wavefronts, not by these loads).
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
}

// Fallback for shapes neither wavefront kernel covers (h > 1024, more sweeps than they pipeline, degenerate
// sizes): one thread per frame walks the image in raster order (diag-layout operands).  Correct for
// every input the reference accepts, slow; only large-image configurations reach it.
__global__ void sor_serial_kernel(const SorArgs a) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= a.t.nframes) return;
  const int w = a.t.w, h = a.t.h, npx = w * h;
  const float* a11 = a.sys + (size_t)f * 7 * npx;
  const float *a12 = a11 + npx, *a22 = a12 + npx, *b1 = a22 + npx, *b2 = b1 + npx, *sh = b2 + npx, *sv = sh + npx;
  float* du = a.du + (size_t)f * npx;
  float* dv = a.dv + (size_t)f * npx;
  const float omega = a.omega;
#define DG(i, j) diag_index((i), (j), w, h)
  if (w < 2 || h < 2 || a.iterations < 1) {  // solver.c:80-83 -> sor_coupled_slow_but_readable (solver.c:19-72)
    for (int iter = 0; iter < a.iterations; ++iter)
      for (int j = 0; j < h; ++j)
        for (int i = 0; i < w; ++i) {
          const int o = DG(i, j);
          float sigma_u = 0.0f, sigma_v = 0.0f, sum_dpsis = 0.0f;
          if (j > 0) { const int q = DG(i, j - 1); sigma_u -= sv[q] * du[q]; sigma_v -= sv[q] * dv[q]; sum_dpsis += sv[q]; }
          if (i > 0) { const int q = DG(i - 1, j); sigma_u -= sh[q] * du[q]; sigma_v -= sh[q] * dv[q]; sum_dpsis += sh[q]; }
          if (j < h - 1) { const int q = DG(i, j + 1); sigma_u -= sv[o] * du[q]; sigma_v -= sv[o] * dv[q]; sum_dpsis += sv[o]; }
          if (i < w - 1) { const int q = DG(i + 1, j); sigma_u -= sh[o] * du[q]; sigma_v -= sh[o] * dv[q]; sum_dpsis += sh[o]; }
          const float A11 = a11[o] + sum_dpsis, A12 = a12[o], A22 = a22[o] + sum_dpsis;
          const float B1 = b1[o] - sigma_u, B2 = b2[o] - sigma_v;
          du[o] = (1.0f - omega) * du[o] + omega / A11 * (B1 - A12 * dv[o]);
          dv[o] = (1.0f - omega) * dv[o] + omega / A22 * (B2 - A12 * du[o]);
        }
    return;
  }
  for (int iter = 0; iter < a.iterations; ++iter)
    for (int j = 0; j < h; ++j)
      for (int i = 0; i < w; ++i) {
        const int o = DG(i, j);
        const int ol = DG(i > 0 ? i - 1 : 0, j), orr = DG(i < w - 1 ? i + 1 : i, j);
        const int ot = DG(i, j > 0 ? j - 1 : 0), ob = DG(i, j < h - 1 ? j + 1 : j);
        const float hl = (i > 0) ? sh[ol] : 0.0f, hr = sh[o];
        const float rgu = (i < w - 1) ? du[orr] : 0.0f, rgv = (i < w - 1) ? dv[orr] : 0.0f;
        float d = hl + hr;
        if (j > 0) d = d + sv[ot];
        if (j < h - 1) d = d + sv[o];
        const float A11 = a22[o] + d, A22 = a11[o] + d;
        const float det = A11 * A22 - a12[o] * a12[o];
        const float i11 = A11 / det, i22 = A22 / det, i12 = a12[o] / (-det);
        float s1 = hr * rgu, s2 = hr * rgv;
        if (j > 0) { s1 = s1 + sv[ot] * du[ot]; s2 = s2 + sv[ot] * dv[ot]; }
        if (j < h - 1) { s1 = s1 + sv[o] * du[ob]; s2 = s2 + sv[o] * dv[ob]; }
        s1 = s1 + b1[o];
        s2 = s2 + b2[o];
        float B1 = s1, B2 = s2;
        if (i > 0) { B1 = hl * du[ol] + s1; B2 = hl * dv[ol] + s2; }
        const float u0 = du[o], v0 = dv[o];
        du[o] = u0 + omega * (i11 * B1 + i12 * B2 - u0);
        dv[o] = v0 + omega * (i12 * B1 + i22 * B2 - v0);
      }
#undef DG
}

hipError_t launch_sor(const SorArgs& a, hipStream_t s) {
  const int w = a.t.w, h = a.t.h;
  constexpr int PD = 3;
  if (w >= 2 && h >= 2 && h <= 64 && a.iterations >= 1 && a.iterations <= 4) {
    const int R = h <= 16 ? 16 : (h <= 32 ? 32 : 64);
    const int G = 64 / R;
    const int waves = (a.t.nframes + G - 1) / G;
    const int blocks = (waves + 3) / 4;
    switch (a.iterations) {
      case 1: hipLaunchKernelGGL((sor_wave_kernel<1, PD>), dim3(blocks), dim3(256), 0, s, a, R); break;
      case 2: hipLaunchKernelGGL((sor_wave_kernel<2, PD>), dim3(blocks), dim3(256), 0, s, a, R); break;
      case 3: hipLaunchKernelGGL((sor_wave_kernel<3, PD>), dim3(blocks), dim3(256), 0, s, a, R); break;
      default: hipLaunchKernelGGL((sor_wave_kernel<4, PD>), dim3(blocks), dim3(256), 0, s, a, R); break;
    }
  } else if (w >= 2 && h > 64 && h <= 1024 && a.iterations >= 1 && a.iterations <= 3) {
    const int threads = ((h + 63) / 64) * 64;
    // <= 640 threads leave 170 VGPRs per lane (no spills); taller levels take the 1024-thread build
    const dim3 g(a.t.nframes), b(threads);
    if (threads <= 640) {
      switch (a.iterations) {
        case 1: hipLaunchKernelGGL((sor_block_kernel<1, PD, 640>), g, b, 0, s, a); break;
        case 2: hipLaunchKernelGGL((sor_block_kernel<2, PD, 640>), g, b, 0, s, a); break;
        default: hipLaunchKernelGGL((sor_block_kernel<3, PD, 640>), g, b, 0, s, a); break;
      }
    } else {
      switch (a.iterations) {
        case 1: hipLaunchKernelGGL((sor_block_kernel<1, PD, 1024>), g, b, 0, s, a); break;
        case 2: hipLaunchKernelGGL((sor_block_kernel<2, PD, 1024>), g, b, 0, s, a); break;
        default: hipLaunchKernelGGL((sor_block_kernel<3, PD, 1024>), g, b, 0, s, a); break;
      }
    }
  } else {
    hipLaunchKernelGGL(sor_serial_kernel, dim3((a.t.nframes + 63) / 64), dim3(64), 0, s, a);
  }
  return hipGetLastError();
}

}  // namespace OFDIS_KNS
}  // namespace ofdis
