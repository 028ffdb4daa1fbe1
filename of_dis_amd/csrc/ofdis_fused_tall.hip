// ofdis_fused_tall.hip -- the fused TV kernel of ofdis_fused.hip (throughput mapping, MODE 0) for levels of 65 ... 256 rows:
// TWO to FOUR wavefronts per strip.  Same arithmetic in the same order: bit-identical results.
//
// Why it exists.  The reference pads a frame to a multiple of 2^sc_f, which puts the finest level of a 1920x1080 or 3840x2160
// gray pair at operating point 2 at 120 x 68 pixels -- four rows more than a wavefront has lanes.  The fused kernel maps
// lane = image row (an anti-diagonal of the level is one step), so such a level used to fall back to the unfused path
// (tv_system + block SOR per fixed-point iteration: 2.1 of the 2.97 ms per 1024 pairs at 1080p, tools/size_probe.py).  An
// anti-diagonal of a 120 x 68 level has up to 68 independent pixels: two wavefronts, wave 0 = rows 0..63, wave 1 = rows
// 64..h-1, both at the same step t (lane's column = t - row), 53 % of the lanes busy at h = 68, all of them at h = 128.
// Taller levels (portrait frames: 1080 x 1920 gives 135 x 240) take three or four wavefronts the same way: wave q = rows
// 64 q .. 64 q + 63, a mailbox at each of the NW - 1 boundaries, a middle wavefront publishing in both directions.
//
// What crosses the wavefront boundary.  In diag coordinates every stencil needs one DPP lane shift (ofdis_fused.hip header);
// between lane 63 of wave 0 and lane 0 of wave 1 the shift becomes an LDS mailbox.  Per step 21 values cross it (NS = 3):
//   down (wave 0 lane 63 -> wave 1 lane 0, the `from_prev` uses): uu, vv of row t+1; wx, wy of row t; sv of row t's slot;
//        the NS sweeps' results of the previous step (tu, tv)                                                 5 + 2 NS values
//   up   (wave 1 lane 0 -> wave 0 lane 63, the `from_next` uses): uu, vv of row t+3; wx, wy of row t+2; du, dv of row t's
//        slot; the first NS - 1 sweeps' results of the previous step (bu, bv); the smoothness of row t+2     7 + 2 (NS - 1)
// All but one of them exist when the step's row t+3 has been assembled (part 2): they are published then (phase A, a
// double-buffered mailbox indexed by the step's parity) and one LDS-only barrier later the other wavefront holds them.  The
// smoothness of row t+2 is computed FROM phase-A values in this very step and needed by the other wavefront in this very step:
// a second mailbox word and a second barrier (phase B).  Two barriers per step; a wavefront can never be more than one phase
// ahead of the other, so A needs two buffers (the faster wavefront writes step t+1's while the slower still reads step t's in
// parts 4 / 5) and B one (it is rewritten only after barrier A of the next step, which the reader has passed by then).
#include <algorithm>

#include "ofdis_fused.h"

namespace ofdis {
namespace OFDIS_KNS {  // the arithmetic contract this file is being compiled for (ofdis_dev.h)

// GROUPED (levels of 65 ... 96 rows: the HD / 4K case is 68).  Two wavefronts per strip leave the second one with h - 64 of
// its 64 lanes busy -- 4 at h = 68.  Here a workgroup takes NH strips: NH "head" wavefronts (rows 0..63 of one strip each) and
// ONE "tail" wavefront whose lane groups of RT lanes (RT = the power of two >= h - 64) are rows 64..h-1 of those NH strips:
// NH + 1 wavefronts for NH strips instead of 2 NH (h = 68: 80 % of the lanes busy at NH = 3, 93 % at NH = 7, instead of 53 %).  A boundary is
// now head i's lane 63 <-> lane RT i of the tail, mailbox i; inside the tail wavefront a DPP shift never crosses into another
// group's pixels: the first lane of a group takes its upper neighbour from the mailbox, the last row of a strip has no lower
// neighbour (has_bot = false selects every such value away).  Everything else is the step of the kernel above, unchanged.
// NOC = 3: RGB levels (three derivative record arrays, the RGB data term, a derivative ring of two rows: see tv_fused_kernel).
template <int NS, bool BRIGHT, bool GROUPED, int NOC = 1>
__global__ __launch_bounds__(GROUPED ? 512 : 256) void tv_fused_tall_kernel(const FusedArgs a, const int RT) {
  constexpr int U = 6;
  constexpr int PDW = 5, PDD = NOC == 3 ? 2 : 3;
  constexpr int NDOWN = 5 + 2 * NS, NUP = 6 + 2 * (NS - 1), NMB = 12;  // mailbox words per direction (padded to 16-byte reads)
  static_assert(NDOWN <= NMB && NUP <= NMB, "mailbox too small");
  static_assert(2 * (NS - 1) + 1 < U, "slot ring too small for this many pipelined sweeps");
  constexpr int NB = GROUPED ? 7 : 3;  // boundaries: between the (at most four) wavefronts of a strip / heads <-> tail groups
  __shared__ __attribute__((aligned(16))) float mbA[2][2][NB][NMB];  // [step parity][0 = down, 1 = up][boundary][value]
  __shared__ float mbB[NB];  // smoothness of row t+2, lane 0 of the wavefront below a boundary -> lane 63 of the one above
  const int w = a.t.w, h = a.t.h;
  const int rw = a.S * w;  // diag rows of a strip = its columns
  const int lane = threadIdx.x & 63;
  const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wavefront q: rows 64 q .. 64 q + 63
  const int nw = (int)(blockDim.x >> 6);                            // wavefronts per strip: ceil(h / 64)
  const int nstrips = a.t.nframes / a.S;
  const int nh = GROUPED ? nw - 1 : 1;                 // strips per workgroup (GROUPED: one head wavefront each)
  const int s0 = blockIdx.x * nh;                      // first strip of this workgroup
  if (s0 >= nstrips) return;
  const int nvalid = min(nh, nstrips - s0);            // (the last workgroup of a launch may hold fewer)
  const bool tail = GROUPED && q == nh;
  // this lane's strip within the workgroup and its image row; lanes / wavefronts without a pixel shadow the nearest real one
  // (real data, results never stored: "Border handling", ofdis_fused.hip)
  int fl = 0, jr = 64 * q + lane;
  bool strip_ok = true;
  if constexpr (GROUPED) {
    const int g = tail ? lane / RT : q;
    strip_ok = g < nvalid;
    fl = min(g, nvalid - 1);
    jr = tail ? 64 + lane % RT : lane;
  }
  const bool row_ok = strip_ok && jr < h;
  const int j = jr < h ? jr : h - 1;
  const bool has_top = j > 0, has_bot = j < h - 1;
  // a boundary above / below this wavefront (wave-uniform); GROUPED: every head has one below, the tail one above every group
  const bool has_up = GROUPED ? tail : q > 0, has_dn = GROUPED ? !tail : q + 1 < nw;
  const bool lo_edge = has_up && (GROUPED ? lane % RT == 0 : lane == 0);  // its "previous lane" lives in another wavefront
  const bool hi_edge = has_dn && lane == 63;                              // its "next lane" lives in another wavefront
  // the mailbox of that boundary: per lane in the tail wavefront, uniform otherwise.  (GROUPED: the tail's lane groups beyond
  // the NH-th shadow the last strip and READ the last mailbox -- finite data -- but never publish)
  const int b_up = GROUPED ? min(lane / RT, nh - 1) : (has_up ? q - 1 : 0);
  const int b_dn = GROUPED ? min(q, nh - 1) : (has_dn ? q : 0);
  const bool pub_up = lo_edge && (!GROUPED || lane / RT < nh);
  const float omega = a.omega, qa = a.quarter_alpha, hd3 = a.half_delta_over3, hg3 = a.half_gamma_over3;

  const size_t strip_recs = (size_t)rw * h;
  auto rsrc = [&](const float* base, int rec_floats) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)s0 * strip_recs * rec_floats), 0,
                                             (int)(nvalid * strip_recs * rec_floats * 4), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsD = rsrc(a.d8, 8), rsW = rsrc(a.wrec, 2), rsU = rsrc(a.uv, 2);
  const __amdgpu_buffer_rsrc_t rsD1 = rsrc(a.d8 + (NOC == 3 ? (size_t)a.t.nframes * w * h * 8 : 0), 8);
  const __amdgpu_buffer_rsrc_t rsD2 = rsrc(a.d8 + (NOC == 3 ? (size_t)a.t.nframes * w * h * 16 : 0), 8);
  const int vrec = fl * (int)strip_recs + j;  // this lane's record within diag row 0 of its strip
  const int vo8 = vrec * 32, vo2 = vrec * 8;
  auto asf = [](unsigned u) { return __builtin_bit_cast(float, u); };

  const int npx = w * h;
  float2* const flow_row =
      a.flow_out ? reinterpret_cast<float2*>(a.flow_out) + ((size_t)(s0 + fl) * a.S * npx + (size_t)j * w) : nullptr;
  const bool aos_out = a.flow_out != nullptr;

  FRow W[6];
  FDer D[PDD][NOC];
  float uu[3], vv[3], sm[3];
  FSlot slot[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) { W[r] = FRow{0, 0, 0, 0}; slot[r] = FSlot{1, 0, 1, 0, 0, 1, 1, 0, 0, 0, 0}; }  // ("Border handling", ofdis_fused.hip)
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    uu[r] = vv[r] = 0.0f;
    sm[r] = 1.0f;
  }
#pragma unroll
  for (int r = 0; r < PDD; ++r)
#pragma unroll
    for (int c = 0; c < NOC; ++c) D[r][c] = FDer{0, 0, 0, 0, 0, 0, 0, 0};
  float ru[NS], rv[NS], ru2[NS], rv2[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) { ru[s] = rv[s] = ru2[s] = rv2[s] = 0.0f; }
  float ldx = 0.0f, ldy = 0.0f;
  float2 Wd[6], ob[U];
#pragma unroll
  for (int r = 0; r < 6; ++r) Wd[r] = make_float2(0.0f, 0.0f);
#pragma unroll
  for (int r = 0; r < U; ++r) ob[r] = make_float2(0.0f, 0.0f);
  int ox = 0, ooff = 0;

  // the refined flow of the last fixed-point iteration in runs of U columns of this lane's image row (ofdis_fused.hip: aos_emit)
  auto aos_emit = [&](int e_slot, bool flush, const float2& wq, float du, float dv, int c) {
    ob[e_slot] = make_float2(wq.x + du, wq.y + dv);
    if (flush) {
      const int c0 = c - (U - 1);
      if (row_ok & (c0 >= 0) & (c < rw) & (ox >= U - 1)) {
        typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
        f4a8* d = reinterpret_cast<f4a8*>(flow_row + ooff + ox - (U - 1));
#pragma unroll
        for (int e = 0; e < U / 2; ++e) {
          const float2 p0 = ob[2 * e], p1 = ob[2 * e + 1];
          d[e] = f4a8{p0.x, p0.y, p1.x, p1.y};
        }
      } else if (row_ok & (c >= 0) & (c0 < rw)) {
#pragma unroll
        for (int e = 0; e < U; ++e) {
          const int ce = c0 + e;
          int xe = ox - (U - 1) + e, oe = ooff;
          if (xe < 0) { xe += w; oe -= npx; }
          if ((ce >= 0) & (ce < rw)) flow_row[oe + xe] = ob[e];
        }
      }
    }
    if (c >= 0) {
      ++ox;
      if (ox == w) { ox = 0; ooff += npx; }
    }
  };
  auto wrap_row = [&](int r) { r %= rw; return r < 0 ? r + rw : r; };
  auto wrap_col = [&](int c) { c %= w; return c < 0 ? c + w : c; };
  auto next_row = [&](int r) { return (r + 1 == rw) ? 0 : r + 1; };
  auto load_w = [&](FRow& r, int drow, bool zero_uv) {
    const int o = drow * h * 8;
    const auto t = __builtin_amdgcn_raw_buffer_load_b64(rsW, vo2, o, 0);
    const unsigned t0 = t[0], t1 = t[1];
    r.wx = asf(t0); r.wy = asf(t1);
    // (first fixed-point iteration: du = dv = 0 -- an offset beyond the resource returns +0 without a memory access)
    const auto qv = __builtin_amdgcn_raw_buffer_load_b64(rsU, zero_uv ? 0x7ffffff0 : vo2, o, 0);
    const unsigned q0 = qv[0], q1 = qv[1];
    r.du = asf(q0); r.dv = asf(q1);
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

  // prologue: W rows -1, 0, 1 (indices 2, 3, 4)
  load_w(W[2], wrap_row(-1), true);
  load_w(W[3], wrap_row(0), true);
  load_w(W[4], wrap_row(1), true);
  int rowW = wrap_row(PDW - 3);
  int rowD = wrap_row(PDD - 3);
  int srow = wrap_row(-3 - 2 * (NS - 1));
  int x2 = wrap_col(-1 - j);
  bool x1_last = (wrap_col(-2 - j) == w - 1);
  const int wtot = a.n_inner * rw;
  const int tend = (wtot - 1) + (h - 1) + 2 * (NS - 1);
  int ig = -3 - j - 2 * (NS - 1);
  bool first_w = true;
  int par = 0;  // parity of the step: which phase-A mailbox it uses
  for (int k0 = 0; k0 <= tend + 3; k0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // ---- (1) loads: W row t+5, D row t+3
      load_w(W[(u + PDW) % 6], rowW, first_w);
      rowW = next_row(rowW);
      load_d(D[(u + PDD) % PDD], rowD);
      rowD = next_row(rowD);
      // ---- (2) uu, vv of row t+3
      {
        const FRow& r = W[(u + 3) % 6];
        uu[u % 3] = r.wx + r.du;
        vv[u % 3] = r.wy + r.dv;
      }
      // ---- phase A: the boundary lane of each wavefront publishes what the other wavefront's boundary lane will take in
      //      place of a DPP lane shift during this step (everything but the smoothness of row t+2, which does not exist yet)
      {
        if (has_dn) {  // lane 63 publishes "down" at the boundary below this wavefront
          float* const mw = &mbA[par][0][b_dn][0];
          if (lane == 63) {
            mw[0] = uu[(u + 1) % 3]; mw[1] = vv[(u + 1) % 3];          // row t+1: ut, vt of the row below
            mw[2] = W[u % 6].wx; mw[3] = W[u % 6].wy;                  // row t: wx_u, wy_u
            mw[4] = slot[u % 6].sv;                                    // sv_t
#pragma unroll
            for (int s = 0; s < NS; ++s) { mw[5 + s] = ru[s]; mw[5 + NS + s] = rv[s]; }  // tu, tv of sweep s
          }
        }
        if (has_up) {  // lane 0 (GROUPED: the first lane of every group) publishes "up" at the boundary above
          float* const mw = &mbA[par][1][b_up][0];
          if (pub_up) {
            mw[0] = uu[u % 3]; mw[1] = vv[u % 3];                      // row t+3: ub, vb of the row above
            mw[2] = W[(u + 2) % 6].wx; mw[3] = W[(u + 2) % 6].wy;      // row t+2: wx_d, wy_d
            mw[4] = slot[u % 6].dur; mw[5] = slot[u % 6].dvr;          // bu, bv of sweep 0
#pragma unroll
            for (int s = 0; s + 1 < NS; ++s) { mw[6 + s] = ru[s]; mw[6 + (NS - 1) + s] = rv[s]; }  // bu, bv of sweep s + 1
          }
        }
      }
      mw_step_barrier();
      // what the neighbouring wavefronts' boundary lanes published: "down" of the boundary above (for lane 0), "up" of the one
      // below (for lane 63); a wavefront without that boundary reads a mailbox nobody uses (a valid address, value unused)
      float inD[NMB], inU[NMB];
      {
        const float4* md = reinterpret_cast<const float4*>(&mbA[par][0][b_up][0]);
        const float4* mu = reinterpret_cast<const float4*>(&mbA[par][1][b_dn][0]);
#pragma unroll
        for (int k = 0; k < NMB / 4; ++k) {
          const float4 v = md[k];
          inD[4 * k] = v.x; inD[4 * k + 1] = v.y; inD[4 * k + 2] = v.z; inD[4 * k + 3] = v.w;
          const float4 x = mu[k];
          inU[4 * k] = x.x; inU[4 * k + 1] = x.y; inU[4 * k + 2] = x.z; inU[4 * k + 3] = x.w;
        }
      }
      // the lane shifts of ofdis_fused.hip with the wavefront boundaries bridged (kd / ku: index in the down / up mailbox)
      auto from_prev = [&](float x, int kd) { const float r = wave_from_prev(x); return lo_edge ? inD[kd] : r; };
      auto from_next = [&](float x, int ku) { const float r = wave_from_next(x); return hi_edge ? inU[ku] : r; };
      // ---- (3) smoothness of row t+2 (opticalflow_aux.c:128-140)
      const bool x2_last = (x2 == w - 1);
      {
        const float uc = uu[(u + 2) % 3], vc = vv[(u + 2) % 3];
        float ul = uu[(u + 1) % 3], vl = vv[(u + 1) % 3];
        float ur = uu[u % 3], vr = vv[u % 3];
        float ut = from_prev(uu[(u + 1) % 3], 0), vt = from_prev(vv[(u + 1) % 3], 1);
        float ub = from_next(uu[u % 3], 0), vb = from_next(vv[u % 3], 1);
        if (x2 == 0) { ul = uc; vl = vc; }
        if (x2_last) { ur = uc; vr = vc; }
        if (!has_top) { ut = uc; vt = vc; }
        if (!has_bot) { ub = uc; vb = vc; }
        const float ex = ur - ul, fx = vr - vl, ey = ub - ut, fy = vb - vt;
        sm[(u + 2) % 3] = fdiv_by_sqrt(qa, 0.25f * (ex * ex + ey * ey + fx * fx + fy * fy) + EPS_SMOOTH);
      }
      // ---- phase B: the smoothness of row t+2 of a wavefront's first row, for the vertical edge weight of the row above it
      if (pub_up) mbB[b_up] = sm[(u + 2) % 3];
      mw_step_barrier();
      // ---- (4) system of pixel row tau = t+1 (opticalflow_aux.c:150-163, 172-199, 342-427)
      {
        const float sc = sm[(u + 1) % 3];
        const float s_r = sm[(u + 2) % 3];
        float s_d = wave_from_next(sm[(u + 2) % 3]);
        if (has_dn) {  // (wave-uniform: lane 63's lower neighbour lives in the next wavefront)
          const float sb = mbB[b_dn];
          s_d = hi_edge ? sb : s_d;
        }
        const float sh_c = x1_last ? 0.0f : sc + s_r;
        const float sv_c = has_bot ? sc + s_d : 0.0f;
        const FRow& rc = W[(u + 1) % 6];
        const FRow& rm = W[u % 6];
        const FRow& rp = W[(u + 2) % 6];
        float a11, a12, a22, b1, b2;
        if constexpr (NOC == 1) data_term_gray<BRIGHT>(D[(u + 1) % PDD][0], rc.du, rc.dv, hd3, hg3, a11, a12, a22, b1, b2);
        else data_term_rgb<BRIGHT>(D[(u + 1) % PDD], rc.du, rc.dv, hd3, hg3, a11, a12, a22, b1, b2);
        const float wx_u = from_prev(rm.wx, 2), wy_u = from_prev(rm.wy, 3);
        const float wx_d = from_next(rp.wx, 2), wy_d = from_next(rp.wy, 3);
        const float sh_l = slot[u % 6].sh;
        const float sv_t = from_prev(slot[u % 6].sv, 4);
        const float rdx = rp.wx - rc.wx, rdy = rp.wy - rc.wy;
        b1 -= sh_l * ldx;
        b2 -= sh_l * ldy;
        b1 += sh_c * rdx;
        b2 += sh_c * rdy;
        ldx = rdx; ldy = rdy;
        Wd[(u + 1) % 6] = make_float2(rc.wx, rc.wy);
        b1 -= sv_t * (rc.wx - wx_u);
        b2 -= sv_t * (rc.wy - wy_u);
        b1 += sv_c * (wx_d - rc.wx);
        b2 += sv_c * (wy_d - rc.wy);
        FSlot& o = slot[(u + 1) % 6];
        o.a11 = a11; o.a12 = a12; o.a22 = a22; o.b1 = b1; o.b2 = b2; o.sh = sh_c; o.sv = sv_c;
        o.dur = rp.du; o.dvr = rp.dv;
        o.hl = sh_l; o.vt = sv_t;
      }
      x1_last = x2_last;
      x2 = x2_last ? 0 : x2 + 1;
      // ---- (5) SOR step t: sweep 0 reaches pixel (j, t - j); block inverse (solver.c:100-110)
      {
        FSlot& c = slot[u % 6];
        const float d = c.hl + c.sh + c.vt + c.sv;
        const float A11 = c.a22 + d, A22 = c.a11 + d;
        const float det = A11 * A22 - c.a12 * c.a12;
        const FDen dd = fden(det);
        c.a11 = fdiv_by(A11, dd);
        c.a22 = fdiv_by(A22, dd);
        c.a12 = -fdiv_by(c.a12, dd);
      }
      float nu[NS], nv[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const FSlot& c = slot[(u - 2 * s + 12) % 6];
        float ou, ov, rgu, rgv, bu, bv;
        if (s == 0) {
          const FSlot& p = slot[(u + 5) % 6];
          ou = p.dur; ov = p.dvr;
          rgu = c.dur; rgv = c.dvr;
          bu = from_next(c.dur, 4);
          bv = from_next(c.dvr, 5);
        } else {
          ou = ru2[s - 1]; ov = rv2[s - 1];
          rgu = ru[s - 1]; rgv = rv[s - 1];
          bu = from_next(ru[s - 1], 6 + (s - 1));
          bv = from_next(rv[s - 1], 6 + (NS - 1) + (s - 1));
        }
        const float tu = from_prev(ru[s], 5 + s), tv = from_prev(rv[s], 5 + NS + s);
        const float lu = ru[s], lv = rv[s];
        const float s1 = c.sh * rgu + c.vt * tu + c.sv * bu + c.b1;
        const float s2 = c.sh * rgv + c.vt * tv + c.sv * bv + c.b2;
        const float B1 = c.hl * lu + s1, B2 = c.hl * lv + s2;
        nu[s] = ou + omega * (c.a11 * B1 + c.a12 * B2 - ou);
        nv[s] = ov + omega * (c.a12 * B1 + c.a22 * B2 - ov);
      }
      {
        {  // for the next step's row: first iteration, or past the last one (see tv_fused_kernel)
          const int cw = ig + (PDW + 1 + 2 * (NS - 1));
          first_w = (cw < rw) | (cw >= wtot);
        }
        const u32x2 v = {__builtin_bit_cast(unsigned, nu[NS - 1]), __builtin_bit_cast(unsigned, nv[NS - 1])};
        const int lastc = ig - (wtot - rw);  // >= 0: this column belongs to the last fixed-point iteration
        const bool on = row_ok & (ig >= 0) & (aos_out ? lastc < 0 : ig < wtot);
        __builtin_amdgcn_raw_buffer_store_b64(v, rsU, on ? vo2 : 0x7ffffff0, srow * h * 8, 0);
        if (aos_out) aos_emit(u, u == U - 1, Wd[(u - 2 * (NS - 1) + 12) % 6], nu[NS - 1], nv[NS - 1], lastc);
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        ru2[s] = ru[s]; rv2[s] = rv[s];
        ru[s] = nu[s]; rv[s] = nv[s];
      }
      srow = next_row(srow);
      ++ig;
      par ^= 1;
    }
  }
}

bool tv_fused_tall_supported(const TvGeom& t, int iterations) {
  return (t.noc == 1 || t.noc == 3) && t.h > 64 && t.h <= 256 && t.w >= 16 && iterations >= 1 && iterations <= 3;
}

// group: the heads + shared tail form (ofdis_tuning::fused_tall_group): 0 = never, 1 = the default, up to THREE strips per
// workgroup (four wavefronts, two workgroups per compute unit), 2 .. 7 = at most that many.  Measured at 1920x1080 (level
// 120 x 68), k frames/s at 1024 / 4096 / 8192 pairs: ungrouped 473 / 646 / 683; cap 2: 589 / 685 / 722; 3: 575 / 718 / 739;
// 4: 567 / 663 / 691; 5: 556 / 686 / 729; 7: 547 / 699 / 747 -- the lanes saved by a wider group are paid back by the lock
// step of more wavefronts and by coarser rounds of workgroups (profiles/r06_variants.txt)
hipError_t launch_tv_fused_tall(const FusedArgs& a, hipStream_t s, int group) {
  if (!tv_fused_tall_supported(a.t, a.iterations) || a.n_inner < 1 || a.S < 1 || a.t.nframes % a.S != 0) return hipErrorInvalidValue;
  const int nstrips = a.t.nframes / a.S;
  const bool bright = a.half_delta_over3 != 0.0f;
  const int h = a.t.h;
  // 65 ... 96 rows and at least two strips: heads + one shared tail wavefront (GROUPED).  NH = what the tail's lane groups hold,
  // at most 7 (eight wavefronts = one workgroup per compute unit at two wavefronts per SIMD)
  int RT = 64;
  if (h <= 96 && group) RT = h - 64 <= 4 ? 4 : (h - 64 <= 8 ? 8 : (h - 64 <= 16 ? 16 : 32));
  const int cap = group >= 2 ? std::min(group, 7) : 3;
  const int nh = RT < 64 ? std::min(std::min(cap, 64 / RT), nstrips) : 1;
  if (nh >= 2) {
    const dim3 bd(64 * (nh + 1)), gd((nstrips + nh - 1) / nh);
#define OFDIS_TALL_LAUNCH_G(NS)                                                                           \
  if (a.t.noc == 3) {                                                                                      \
    if (bright) hipLaunchKernelGGL((tv_fused_tall_kernel<NS, true, true, 3>), gd, bd, 0, s, a, RT);        \
    else hipLaunchKernelGGL((tv_fused_tall_kernel<NS, false, true, 3>), gd, bd, 0, s, a, RT);              \
  } else if (bright) hipLaunchKernelGGL((tv_fused_tall_kernel<NS, true, true>), gd, bd, 0, s, a, RT);      \
  else hipLaunchKernelGGL((tv_fused_tall_kernel<NS, false, true>), gd, bd, 0, s, a, RT)
    switch (a.iterations) {
      case 1: OFDIS_TALL_LAUNCH_G(1); break;
      case 2: OFDIS_TALL_LAUNCH_G(2); break;
      default: OFDIS_TALL_LAUNCH_G(3); break;
    }
#undef OFDIS_TALL_LAUNCH_G
    return hipGetLastError();
  }
  const dim3 bd(64 * ((h + 63) / 64));  // two to four wavefronts per strip
#define OFDIS_TALL_LAUNCH(NS)                                                                             \
  if (a.t.noc == 3) {                                                                                      \
    if (bright) hipLaunchKernelGGL((tv_fused_tall_kernel<NS, true, false, 3>), dim3(nstrips), bd, 0, s, a, 64); \
    else hipLaunchKernelGGL((tv_fused_tall_kernel<NS, false, false, 3>), dim3(nstrips), bd, 0, s, a, 64);  \
  } else if (bright) hipLaunchKernelGGL((tv_fused_tall_kernel<NS, true, false>), dim3(nstrips), bd, 0, s, a, 64); \
  else hipLaunchKernelGGL((tv_fused_tall_kernel<NS, false, false>), dim3(nstrips), bd, 0, s, a, 64)
  switch (a.iterations) {
    case 1: OFDIS_TALL_LAUNCH(1); break;
    case 2: OFDIS_TALL_LAUNCH(2); break;
    default: OFDIS_TALL_LAUNCH(3); break;
  }
#undef OFDIS_TALL_LAUNCH
  return hipGetLastError();
}

}  // namespace OFDIS_KNS
}  // namespace ofdis
