// ofdis_fused_xcu.hip -- the fused TV kernel of ofdis_fused.hip with every fixed-point iteration of a frame group on its
// own compute unit (small batches).  Same arithmetic in the same order as the other variants: bit-identical results.
#include "ofdis_fused.h"

namespace ofdis {
namespace OFDIS_KNS {  // the arithmetic contract this file is being compiled for (ofdis_dev.h)

// ------------------------------------------------------------------------------------------- MODE 3: "xcu" (cross-CU)
// The multi-wave variants of ofdis_fused.hip keep a frame group on ONE compute unit, and a CU issues about one wavefront
// instruction per clock: a diagonal step of n fixed-point iterations costs n x ~400 instructions / 4 SIMDs x 4 clocks
// whatever the number of wavefronts (0.79 us per step at level 3).  This variant gives every fixed-point iteration of a
// frame group its OWN workgroup -- its own CU while the launch has no more workgroups than the chip has CUs -- and divides
// the iteration's step between FOUR wavefronts, one per SIMD.  A lone wavefront issues one dependent instruction per ~6
// clocks, so the longest of them sets the step (0.40 us measured):
//   rows wave  (0): (wx, wy) row loads, flow sums, smoothness, Laplacian products -- then, one step later, the data term's
//                   five sums from the data wave, the b1 / b2 updates in the reference's order and the block inverse
//   data wave  (1): derivative record loads and the data term (opticalflow_aux.c:342-427)
//   solve wave (2): the NS pipelined SOR sweeps, one step behind the other two; publishes the finished du/dv row
//   fetch wave (3): brings the previous iteration's du/dv rows into an LDS ring, one row ahead of the rows wave; in the last
//                   iteration it also writes the refined flow, which the solve wave leaves in another LDS ring
// Hand-overs inside the workgroup go through small LDS arrays and one LDS-only barrier per step, as in MODE 2.
// Between iterations -- between CUs -- a finished du/dv row travels through global memory as self-validating 16-byte
// granules {du, tag, dv, tag} written by ONE write-through (sc1) store per lane and read by sc1 loads that bypass the
// reader's L1 (cdna_hip_programming.md Guideline 16, form R2: the data is the flag; each 8-byte half carries its own tag,
// so no ordering between stores is needed; correctness does not depend on timing or workgroup -> XCD placement).
// FORWARD PROGRESS, however, rests on one assumption about the dispatcher: workgroups start in block-index order.  A launch has
// n_inner x G8 workgroups (up to ~15 k: far more than fit on the chip), a consumer spins while it occupies its CU, and it only
// ever waits for a LOWER block index -- so it makes progress exactly when lower indices were dispatched before it.  That is
// how the hardware dispatcher behaves today (also on a CU-masked stream, tests/test_gpu_xcu.py), not an API guarantee: the
// wait is therefore bounded, and a context that ever sees it expire reports the pass as failed and stops using this variant
// (ofdis_capi.hip: XcuState).
// Only the fetch wave touches granules: memory returns in order, so in a wavefront that also computes, every nearer load
// (and the compiler's conservative wait counts around the re-read loop) exposed the 1.2 us hand-off latency at every
// step.  The fetch wave takes one row per step, requested XC_AHEAD steps earlier, checks the tag of every existing pixel -- when
// one is missing it lets the predecessor gain XC_LEAD rows and re-reads (bounded) -- and writes it to the ring.  An iteration
// so trails its predecessor by ~9 steps of software pipeline + the latency (every row of lag costs 12 boundaries x 0.4 us
// per one-pair pass) and never waits in steady state.
// Tag 0 = not yet written: the array is zeroed when it is allocated, and the fetch wave puts every granule it has taken
// back to zero, so a launch leaves the array as it found it (no memset per launch).  Block index = iteration * G8 +
// group with G8 a multiple of 8: an iteration only waits for a LOWER block index, and the iterations of a group share an
// XCD under the observed round-robin placement (speed only).  A wait that lasts longer than the bound (device wall clock;
// default XC_WAIT_US = 50 ms, a thousand times the longest healthy wait: a predecessor that IS running gains a row every
// 0.4 us, a few times that with several passes sharing the CU; ofdis_tuning::fused_xcu_spin) sets the context's error word and
// the wavefront carries on without waiting: the pass is reported as failed, nothing hangs, and a drop-in call loses at most
// that bound before its pass is repeated on another mapping (round 6; the bound used to be a count of re-reads worth seconds).
constexpr int XC_AHEAD = 3;  // steps a row is requested before the fetch wave takes it (= requests in flight; 3 x 0.4 us = the latency)
constexpr int XC_LEAD = 1;   // rows the fetch wave lets the predecessor gain, beyond its requests, before it (re)starts
constexpr int XC_SD = 1;     // rows between consecutive SOR sweeps in the solve wave
constexpr int XC_RING = 4;   // rows of the LDS du/dv ring (a power of two > 2)
constexpr unsigned XC_TAG = 1u;
constexpr unsigned XC_WAIT_US = 50000;  // microseconds a wavefront waits for one row before it gives up
constexpr unsigned XC_FLAG_DROP = 1u;   // test hook (ofdis_tuning::fused_xcu_drop): iteration 0 withholds its hand-over rows
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NS, bool BRIGHT>
__global__ __launch_bounds__(256) void tv_fused_xcu_kernel(const FusedArgs a, const int R, const int G8, float* const xbuf,
                                                            int* const err, const unsigned wait_ticks, const unsigned flags) {
  constexpr int U = 6;
  constexpr int PDW = 5, PDD = 3, PDU = 3;  // (du/dv of row t+3 are read from the ring in the step that first uses them)
  static_assert(XC_SD * (NS - 1) + 1 < U, "slot ring too small for this many pipelined sweeps");
  __shared__ float uvl[2 * 2 * 64];              // rows wave -> data wave: du, dv of a pixel row     [row & 1][field][lane]
  __shared__ float dtl[2 * 5 * 64];              // data wave -> rows wave: a11, a12, a22, b1, b2     [row & 1][field][lane]
  __shared__ float sring[2 * SLOT_FLOATS * 64];  // rows wave -> solve wave: the finished FSlot       [row & 1][field][lane]
  __shared__ float wring[16 * 2 * 64];           // rows wave -> solve wave (last iteration): wx, wy of a pixel row [row & 15][field][lane]
  __shared__ float2 oring[16 * 64];              // solve wave -> fetch wave (last iteration): wx + du, wy + dv of a finished row [row & 15][lane]
  __shared__ float xring[XC_RING * 2 * 64];      // fetch wave -> rows wave: du, dv of the previous iteration [row & 7][field][lane]
  const int w = a.t.w, h = a.t.h;
  const int rw = w;  // S == 1: a strip is a frame
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int it = (int)blockIdx.x / G8, wid = (int)blockIdx.x - it * G8;
  const int n_iters = a.n_inner;
  const int G = 64 / R;
  const int nstrips = a.t.nframes;
  const int s0 = wid * G;
  if (s0 >= nstrips) return;  // whole workgroup idle
  int fl = lane / R;
  const int jr = lane % R;
  const bool row_ok = (s0 + fl < nstrips) && (jr < h);
  if (s0 + fl >= nstrips) fl = nstrips - 1 - s0;
  const int j = jr < h ? jr : h - 1;
  const bool has_top = j > 0, has_bot = j < h - 1;
  const float omega = a.omega, qa = a.quarter_alpha, hd3 = a.half_delta_over3, hg3 = a.half_gamma_over3;
  auto from_prev = [&](float x) { return wave_from_prev(x); };
  auto from_next = [&](float x) { return wave_from_next(x); };
  const int nst = min(G, nstrips - s0);
  const size_t strip_recs = (size_t)rw * h;
  auto rsrc = [&](const float* base, int rec_floats) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)s0 * strip_recs * rec_floats), 0,
                                             (int)(nst * strip_recs * rec_floats * 4), 0x00020000);
  };
  const int vrec = fl * (int)strip_recs + j;
  const int vo8 = vrec * 32, vo2 = vrec * 8, vo4 = vrec * 16;
  auto asf = [](unsigned u) { return __builtin_bit_cast(float, u); };
  auto wrap_row = [&](int r) { r %= rw; return r < 0 ? r + rw : r; };
  auto wrap_col = [&](int c) { c %= w; return c < 0 ? c + w : c; };
  auto next_row = [&](int r) { return (r + 1 == rw) ? 0 : r + 1; };
  const int tend = (rw - 1) + (h - 1) + XC_SD * (NS - 1) + 1;  // (the solve wave runs one step behind)
  const size_t it_floats = (size_t)a.t.nframes * strip_recs * 4;  // granules of one iteration boundary
  int taut = -3;  // unwrapped step number
  const bool wants_w = it == n_iters - 1 && a.flow_out;  // last iteration: the workgroup writes wx + du, wy + dv itself

  if (role == 0) {
    // ------------------------------------------------------------------------------------------------ rows wave
    const __amdgpu_buffer_rsrc_t rsW = rsrc(a.wrec, 2);
    // du, dv of the previous iteration for the row with unwrapped number tau: from the LDS ring the fetch wave fills; zero
    // in the first iteration (image_erase, refine_variational.cpp:186-187)
    auto ring_uv = [&](FRow& r, int tau) {
      if (it == 0) {
        r.du = 0.0f; r.dv = 0.0f;
      } else {
        const float* q = xring + ((tau & (XC_RING - 1)) * 2) * 64 + lane;
        r.du = q[0 * 64]; r.dv = q[1 * 64];
      }
    };
    auto load_w = [&](FRow& r, int drow) {
      const auto t = __builtin_amdgcn_raw_buffer_load_b64(rsW, vo2, drow * h * 8, 0);
      const unsigned t0 = t[0], t1 = t[1];
      r.wx = asf(t0); r.wy = asf(t1);
    };
    FRow W[6];
    float uu[3], vv[3], sm[3];
#pragma unroll
    for (int r = 0; r < 6; ++r) W[r] = FRow{0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 3; ++r) { uu[r] = vv[r] = 0.0f; sm[r] = 1.0f; }
    float ldx = 0.0f, ldy = 0.0f;
    float shp = 1.0f, svp = 1.0f;  // sh, sv of pixel row t (fill phase: unit weights, see "Border handling")
    // the part of row t's system this wave produced one step ago
    float p_sh = 1.0f, p_sv = 1.0f, p_hl = 0.0f, p_vt = 0.0f, p_dur = 0.0f, p_dvr = 0.0f;
    float q1x = 0.0f, q1y = 0.0f, q2x = 0.0f, q2y = 0.0f, q3x = 0.0f, q3y = 0.0f, q4x = 0.0f, q4y = 0.0f;
    // prologue: W rows -1, 0, 1 (indices 2, 3, 4); du/dv of row 0 (row -1 has no pixel)
    load_w(W[2], wrap_row(-1));
    load_w(W[3], wrap_row(0));
    load_w(W[4], wrap_row(1));
    uvl[(0 * 2 + 0) * 64 + lane] = 0.0f;  // du, dv of row -2 for the data wave's first step
    uvl[(0 * 2 + 1) * 64 + lane] = 0.0f;
    __syncthreads();  // (the fetch wave has put row 0 into the ring)
    int rowW = wrap_row(PDW - 3);
    int x2 = wrap_col(-1 - j);
    bool x1_last = (wrap_col(-2 - j) == w - 1);
    for (int k0 = 0; k0 <= tend + 3; k0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // ---- (0) finish the system of pixel row t: data term of the data wave (written one step ago) + the Laplacian
        //      products kept from the previous step, then the block inverse (solver.c:100-110)
        if (taut > -3) {
          const float* dr = dtl + ((taut & 1) * 5) * 64 + lane;
          const float a11 = dr[0 * 64], a12 = dr[1 * 64], a22 = dr[2 * 64];
          float b1 = dr[3 * 64], b2 = dr[4 * 64];
          b1 -= q1x; b2 -= q1y;
          b1 += q2x; b2 += q2y;
          b1 -= q3x; b2 -= q3y;
          b1 += q4x; b2 += q4y;
          const float d = p_hl + p_sh + p_vt + p_sv;
          const float A11 = a22 + d, A22 = a11 + d;
          const float det = A11 * A22 - a12 * a12;
          const FDen dd = fden(det);
          float* sr = sring + ((taut & 1) * SLOT_FLOATS) * 64 + lane;
          sr[0 * 64] = fdiv_by(A11, dd); sr[1 * 64] = -fdiv_by(a12, dd); sr[2 * 64] = fdiv_by(A22, dd);
          sr[3 * 64] = b1; sr[4 * 64] = b2; sr[5 * 64] = p_sh; sr[6 * 64] = p_sv; sr[7 * 64] = p_dur; sr[8 * 64] = p_dvr;
          sr[9 * 64] = p_hl; sr[10 * 64] = p_vt;
        }
        // ---- (1) W row t+5; du/dv of row t+3 from the ring
        load_w(W[(u + PDW) % 6], rowW);
        rowW = next_row(rowW);
        ring_uv(W[(u + PDU) % 6], taut + PDU);
        // ---- (2) uu, vv of row t+3
        {
          const FRow& r = W[(u + 3) % 6];
          uu[u % 3] = r.wx + r.du;
          vv[u % 3] = r.wy + r.dv;
          if (wants_w) {  // the solve wave adds the final du, dv to these eight steps from now
            float* q = wring + (((taut + 3) & 15) * 2) * 64 + lane;
            q[0 * 64] = r.wx; q[1 * 64] = r.wy;
          }
        }
        // ---- (3) smoothness of row t+2 (opticalflow_aux.c:128-140; see the throughput kernel for the scaling argument)
        const bool x2_last = (x2 == w - 1);
        {
          const float uc = uu[(u + 2) % 3], vc = vv[(u + 2) % 3];
          float ul = uu[(u + 1) % 3], vl = vv[(u + 1) % 3];
          float ur = uu[u % 3], vr = vv[u % 3];
          float ut = from_prev(uu[(u + 1) % 3]), vt = from_prev(vv[(u + 1) % 3]);
          float ub = from_next(uu[u % 3]), vb = from_next(vv[u % 3]);
          if (x2 == 0) { ul = uc; vl = vc; }
          if (x2_last) { ur = uc; vr = vc; }
          if (!has_top) { ut = uc; vt = vc; }
          if (!has_bot) { ub = uc; vb = vc; }
          const float ex = ur - ul, fx = vr - vl, ey = ub - ut, fy = vb - vt;
          sm[(u + 2) % 3] = fdiv_by_sqrt(qa, 0.25f * (ex * ex + ey * ey + fx * fx + fy * fy) + EPS_SMOOTH);
        }
        // ---- (4) this wave's part of the system of pixel row t+1 (opticalflow_aux.c:150-163, 172-199)
        {
          const float sc = sm[(u + 1) % 3];
          const float s_r = sm[(u + 2) % 3];
          const float s_d = from_next(sm[(u + 2) % 3]);
          const float sh_c = x1_last ? 0.0f : sc + s_r;
          const float sv_c = has_bot ? sc + s_d : 0.0f;
          const FRow& rc = W[(u + 1) % 6];
          const FRow& rm = W[u % 6];
          const FRow& rp = W[(u + 2) % 6];
          const float wx_u = from_prev(rm.wx), wy_u = from_prev(rm.wy);
          const float wx_d = from_next(rp.wx), wy_d = from_next(rp.wy);
          const float sh_l = shp;
          const float sv_t = from_prev(svp);
          const float rdx = rp.wx - rc.wx, rdy = rp.wy - rc.wy;
          q1x = sh_l * ldx; q1y = sh_l * ldy;
          q2x = sh_c * rdx; q2y = sh_c * rdy;
          ldx = rdx; ldy = rdy;
          q3x = sv_t * (rc.wx - wx_u); q3y = sv_t * (rc.wy - wy_u);
          q4x = sv_c * (wx_d - rc.wx); q4y = sv_c * (wy_d - rc.wy);
          p_sh = sh_c; p_sv = sv_c; p_hl = sh_l; p_vt = sv_t; p_dur = rp.du; p_dvr = rp.dv;
          shp = sh_c; svp = sv_c;
          // du, dv of row t+2 for the data wave's next step
          float* uw = uvl + ((taut & 1) * 2) * 64 + lane;  // (t + 2) & 1 == t & 1
          uw[0 * 64] = rp.du; uw[1 * 64] = rp.dv;
        }
        x1_last = x2_last;
        x2 = x2_last ? 0 : x2 + 1;
        ++taut;
        mw_step_barrier();
      }
    }
  } else if (role == 1) {
    // ------------------------------------------------------------------------------------------------ data wave
    const __amdgpu_buffer_rsrc_t rsD = rsrc(a.d8, 8);
    auto load_d = [&](FDer& r, int drow) {
      const int o = drow * h * 32;
      const auto lo = __builtin_amdgcn_raw_buffer_load_b128(rsD, vo8, o, 0);
      const auto hi = __builtin_amdgcn_raw_buffer_load_b128(rsD, vo8 + 16, o, 0);
      const unsigned l0 = lo[0], l1 = lo[1], l2 = lo[2], l3 = lo[3], h0 = hi[0], h1 = hi[1], h2 = hi[2], h3 = hi[3];
      r.ix = asf(l0); r.iz = asf(l1); r.ixx = asf(l2); r.ixz = asf(l3);
      r.iy = asf(h0); r.ixy = asf(h1); r.iyz = asf(h2); r.iyy = asf(h3);
    };
    FDer D[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) D[r] = FDer{0, 0, 0, 0, 0, 0, 0, 0};
    int rowD = wrap_row(PDD - 3);
    __syncthreads();
    for (int k0 = 0; k0 <= tend + 3; k0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        load_d(D[(u + PDD) % 3], rowD);  // row t+3
        rowD = next_row(rowD);
        const float* ur = uvl + (((taut + 1) & 1) * 2) * 64 + lane;  // du, dv of row t+1
        const float du = ur[0 * 64], dv = ur[1 * 64];
        float a11, a12, a22, b1, b2;
        data_term_gray<BRIGHT>(D[(u + 1) % 3], du, dv, hd3, hg3, a11, a12, a22, b1, b2);
        float* dw = dtl + (((taut + 1) & 1) * 5) * 64 + lane;
        dw[0 * 64] = a11; dw[1 * 64] = a12; dw[2 * 64] = a22; dw[3 * 64] = b1; dw[4 * 64] = b2;
        ++taut;
        mw_step_barrier();
      }
    }
  } else if (role == 3) {
    // ------------------------------------------------------------------------------------------------ fetch wave
    // Brings the previous iteration's du/dv rows from global memory into the LDS ring: at step t it takes row t+4, which it
    // requested XC_AHEAD steps ago (the rows wave reads it in step t+1), checks the tag of every existing pixel -- when one is
    // missing it lets the predecessor gain XC_LEAD rows and re-reads, bounded --, puts the granules back to zero and requests
    // row t+4+XC_AHEAD.  Nothing else in this wavefront's loop waits for memory, so nobody sees the hand-off latency.
    if (it > 0) {
      const __amdgpu_buffer_rsrc_t rsX = rsrc(xbuf + (size_t)(it - 1) * it_floats, 4);
      bool dead = false;  // a row never arrived: stop waiting (the results are wrong, the err word says so)
      auto request = [&](int drow) { return __builtin_amdgcn_raw_buffer_load_b128(rsX, vo4, drow * h * 16, 16 /* sc1 */); };
      auto inrange = [&](int tau) { const int x = tau - j; return (x >= 0) & (x < rw); };  // this lane's pixel of row tau exists
      auto ready = [&](const u32x4& q, bool inr) { return !inr | ((q[1] == XC_TAG) & (q[3] == XC_TAG)); };
      // re-reads row tau until every existing pixel's granules carry their tag; bounded by the device's constant-rate wall
      // clock (wait_ticks = 0: gives up at the first re-read -- how the tests force the failure)
      auto wait_row = [&](int tau) {
        const int drow = wrap_row(tau);
        const bool inr = inrange(tau);
        u32x4 g = request(drow);
        unsigned long long t0 = 0;
        bool started = false;
        while (!dead && __builtin_amdgcn_ballot_w64(!ready(g, inr)) != 0) {
          const unsigned long long now = wall_clock64();
          if (!started) { t0 = now; started = true; }
          // (a pass is several launches: once one of them has given up -- the context's error word says so -- the others do
          // not wait out the whole bound again: after 1/64 of it, ~0.8 ms, a waiting wavefront looks at the word)
          const unsigned long long el = now - t0;
          if (el >= (unsigned long long)wait_ticks ||
              (el >= (unsigned long long)(wait_ticks >> 6) && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) {
            dead = true;
            if (lane == 0) atomicExch(err, 1);
            break;
          }
          __builtin_amdgcn_s_sleep(2);
          asm volatile("" ::: "memory");
          g = request(drow);
        }
        return g;
      };
      const int last_tau = rw + h - 2;  // the last row with a pixel
      // into the ring (lanes outside their columns: zero -- finite, never read as a pixel); the granules go back to "not
      // written" for the next launch: every granule a launch writes is taken exactly once, so the array is all zero
      // again when the launch ends and needs no memset between launches
      auto deliver = [&](const u32x4& g, int tau, int drow) {
        const bool inr = inrange(tau);
        const unsigned g0 = g[0], g2 = g[2];
        float* q = xring + ((tau & (XC_RING - 1)) * 2) * 64 + lane;
        q[0 * 64] = inr ? asf(g0) : 0.0f;
        q[1 * 64] = inr ? asf(g2) : 0.0f;
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{0, 0, 0, 0}, rsX, (inr & row_ok) ? vo4 : 0x7ffffff0, drow * h * 16, 16 /* sc1 */);
      };
      // prologue: wait until the predecessor has passed the first requests by XC_LEAD rows, then take rows 0 .. XC_AHEAD in
      // ONE round trip: row 0 is delivered now, the others are the requests in flight (row r <-> XG[(r - 1) % XC_AHEAD])
      (void)wait_row(min(XC_AHEAD + XC_LEAD, last_tau));
      u32x4 XG[XC_AHEAD];
      {
        u32x4 g0 = request(wrap_row(0));
#pragma unroll
        for (int r = 0; r < XC_AHEAD; ++r) XG[r] = request(wrap_row(1 + r));
        if (!dead && __builtin_amdgcn_ballot_w64(!ready(g0, inrange(0))) != 0) g0 = wait_row(0);
        deliver(g0, 0, wrap_row(0));
      }
      int rowd = wrap_row(1);             // next row to deliver
      int rowq = wrap_row(1 + XC_AHEAD);  // next row to request
      // Last iteration: the refined flow itself, AoS (refine_variational.cpp:209-221, 92-99).  The solve wave leaves
      // every finished row in an LDS ring; every six steps this wavefront writes the six rows finished since: in a lane's
      // image row they are six consecutive pixels, 48 contiguous bytes instead of six scattered 8-byte stores.
      float2* const flow_row = reinterpret_cast<float2*>(a.flow_out) + ((size_t)(s0 + fl) * (w * h) + (size_t)j * w);
      auto flush = [&](int r0) {  // diag rows r0 .. r0 + 5 (unwrapped): pixels x0 .. x0 + 5 of this lane's image row
        const int x0 = r0 - j;
        float2 v[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) v[e] = oring[((r0 + e) & 15) * 64 + lane];
        if (row_ok & (x0 >= 0) & (x0 + 5 < rw)) {
          typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
          f4a8* d = reinterpret_cast<f4a8*>(flow_row + x0);
          d[0] = f4a8{v[0].x, v[0].y, v[1].x, v[1].y};
          d[1] = f4a8{v[2].x, v[2].y, v[3].x, v[3].y};
          d[2] = f4a8{v[4].x, v[4].y, v[5].x, v[5].y};
        } else if (row_ok) {
#pragma unroll
          for (int e = 0; e < 6; ++e)
            if ((x0 + e >= 0) & (x0 + e < rw)) flow_row[x0 + e] = v[e];
        }
      };
      __syncthreads();
      for (int k0 = 0; k0 <= tend + 3; k0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          {  // row taut + 4, requested XC_AHEAD steps ago; the rows wave reads it in the next step
            u32x4 g = XG[u % XC_AHEAD];
            if (!dead && __builtin_amdgcn_ballot_w64(!ready(g, inrange(taut + 4))) != 0) {
              // requested too early: fall back behind the predecessor, then take this row and the requests in flight (as early
              // as this one) again, in one round trip; everything this branch loads is waited for inside it, and the ring is
              // refilled from plain registers -- the wait counts of the fast path stay exact
              (void)wait_row(min(taut + 4 + XC_AHEAD + XC_LEAD, last_tau));
              const int r1 = next_row(rowd), r2 = next_row(r1);
              u32x4 t0 = request(rowd), t1 = request(r1), t2 = request(r2);
              if (__builtin_amdgcn_ballot_w64(!ready(t0, inrange(taut + 4))) != 0) t0 = wait_row(taut + 4);
              if (__builtin_amdgcn_ballot_w64(!ready(t1, inrange(taut + 5))) != 0) t1 = wait_row(taut + 5);
              if (__builtin_amdgcn_ballot_w64(!ready(t2, inrange(taut + 6))) != 0) t2 = wait_row(taut + 6);
              g = t0;
              XG[(u + 1) % XC_AHEAD] = t1;
              XG[(u + 2) % XC_AHEAD] = t2;
            }
            deliver(g, taut + 4, rowd);
            rowd = next_row(rowd);
            XG[u % XC_AHEAD] = request(rowq);
            rowq = next_row(rowq);
          }
          if (u == U - 1 && wants_w) flush(taut - 6 - 1 - XC_SD * (NS - 1));  // the rows finished in steps taut-6 .. taut-1
          ++taut;
          mw_step_barrier();
        }
      }
      if (wants_w) flush(taut - 6 - 1 - XC_SD * (NS - 1));  // (overlaps the last flush: same values)
    } else {
      __syncthreads();
      for (int k0 = 0; k0 <= tend + 3; k0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) mw_step_barrier();
      }
    }
  } else {
    // ------------------------------------------------------------------------------------------------ solve wave
    const __amdgpu_buffer_rsrc_t rsX = rsrc(xbuf + (size_t)(it < n_iters - 1 ? it : 0) * it_floats, 4);
    const __amdgpu_buffer_rsrc_t rsU = rsrc(a.uv, 2);
    FSlot slot[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) slot[r] = FSlot{1, 0, 1, 0, 0, 1, 1, 0, 0, 0, 0};  // (block inverse included: any finite value)
    float ru[NS], rv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) ru[s] = rv[s] = 0.0f;
    int srow = wrap_row(-3 - XC_SD * (NS - 1));  // row finished by the last sweep at this wave's first step
    int ig = -3 - j - XC_SD * (NS - 1);          // ... and its column in this lane
    __syncthreads();
    for (int k0 = 0; k0 <= tend + 3; k0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (taut > -3) {  // this wave's step number is taut - 1; its current pixel row sits at ring index us
          const int us = (u + 5) % 6;
          if (taut > -2) {  // the slot of row taut - 1, finished by the rows wave one step ago (the first one is row -2)
            const float* sr = sring + (((taut - 1) & 1) * SLOT_FLOATS) * 64 + lane;
            FSlot& c = slot[us];
            c.a11 = sr[0 * 64]; c.a12 = sr[1 * 64]; c.a22 = sr[2 * 64]; c.b1 = sr[3 * 64]; c.b2 = sr[4 * 64]; c.sh = sr[5 * 64];
            c.sv = sr[6 * 64]; c.dur = sr[7 * 64]; c.dvr = sr[8 * 64]; c.hl = sr[9 * 64]; c.vt = sr[10 * 64];
          }
          float nu[NS], nv[NS];
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            const FSlot& c = slot[(us - XC_SD * s + 12) % 6];
            float ou, ov, rgu, rgv, bu, bv;
            if (s == 0) {
              const FSlot& p = slot[(us + 5) % 6];
              ou = p.dur; ov = p.dvr;
              rgu = c.dur; rgv = c.dvr;
              bu = from_next(c.dur);
              bv = from_next(c.dvr);
            } else {
              // sweep s trails sweep s-1 by ONE row: its right / lower neighbours are what sweep s-1 produced earlier in this
              // very step, its own old value what sweep s-1 produced one step ago (the throughput kernel keeps two rows between
              // sweeps so that the three are independent instructions streams; here every row of lag costs more than that)
              ou = ru[s - 1]; ov = rv[s - 1];
              rgu = nu[s - 1]; rgv = nv[s - 1];
              bu = from_next(nu[s - 1]);
              bv = from_next(nv[s - 1]);
            }
            const float tu = from_prev(ru[s]), tv = from_prev(rv[s]);
            const float lu = ru[s], lv = rv[s];
            const float s1 = c.sh * rgu + c.vt * tu + c.sv * bu + c.b1;
            const float s2 = c.sh * rgv + c.vt * tv + c.sv * bv + c.b2;
            const float B1 = c.hl * lu + s1, B2 = c.hl * lv + s2;
            nu[s] = ou + omega * (c.a11 * B1 + c.a12 * B2 - ou);
            nv[s] = ov + omega * (c.a12 * B1 + c.a22 * B2 - ov);
          }
          const bool on = row_ok & (ig >= 0) & (ig < rw);
          if (it < n_iters - 1) {  // hand the finished row to the next iteration's workgroup: one write-through granule pair
            const u32x4 v = {__builtin_bit_cast(unsigned, nu[NS - 1]), XC_TAG, __builtin_bit_cast(unsigned, nv[NS - 1]), XC_TAG};
            const bool hand = on & !((flags & XC_FLAG_DROP) != 0 && it == 0);  // (test hook: iteration 0 never hands over)
            __builtin_amdgcn_raw_buffer_store_b128(v, rsX, hand ? vo4 : 0x7ffffff0, srow * h * 16, 16 /* sc1 */);
          } else if (a.flow_out) {  // last iteration: the refined flow itself, AoS (refine_variational.cpp:209-221, 92-99)
            const int frow = taut - 1 - XC_SD * (NS - 1);  // the row the last sweep finishes now
            const float* q = wring + ((frow & 15) * 2) * 64 + lane;
            oring[(frow & 15) * 64 + lane] = make_float2(q[0 * 64] + nu[NS - 1], q[1 * 64] + nv[NS - 1]);
          } else {  // ... or du, dv for tv_finish_records
            const u32x2 v = {__builtin_bit_cast(unsigned, nu[NS - 1]), __builtin_bit_cast(unsigned, nv[NS - 1])};
            __builtin_amdgcn_raw_buffer_store_b64(v, rsU, on ? vo2 : 0x7ffffff0, srow * h * 8, 0);
          }
#pragma unroll
          for (int s = 0; s < NS; ++s) { ru[s] = nu[s]; rv[s] = nv[s]; }
          srow = next_row(srow);
          ++ig;
        }
        ++taut;
        mw_step_barrier();
      }
    }
  }
}


hipError_t launch_tv_fused_xcu(const FusedArgs& a, const FusedXcu& x, int waves, int R, hipStream_t s) {
  const int G8 = (waves + 7) & ~7;  // workgroups per fixed-point iteration
  if (!x.xbuf || !x.err) return hipErrorInvalidValue;  // never without an error word: a lost hand-over must be reportable
  // the bound in ticks of the device's constant-rate wall clock (wall_clock64: 100 MHz on this chip; asked, not assumed);
  // 1 us = "give up at the first re-read" (the tests' forced failure)
  static const int clock_khz = [] {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
    return khz;
  }();
  const unsigned us = x.wait_us ? x.wait_us : XC_WAIT_US;
  const unsigned spin = us <= 1 ? 0u : (unsigned)std::min<unsigned long long>(0xffffffffull, (unsigned long long)us * clock_khz / 1000);
  const unsigned flags = x.drop ? XC_FLAG_DROP : 0u;
  const bool bright = a.half_delta_over3 != 0.0f;
#define OFDIS_XCU_LAUNCH(NS)                                                                                           \
  if (bright)                                                                                                          \
    hipLaunchKernelGGL((tv_fused_xcu_kernel<NS, true>), dim3(a.n_inner * G8), dim3(256), 0, s, a, R, G8, x.xbuf, x.err, spin, flags); \
  else                                                                                                                 \
    hipLaunchKernelGGL((tv_fused_xcu_kernel<NS, false>), dim3(a.n_inner * G8), dim3(256), 0, s, a, R, G8, x.xbuf, x.err, spin, flags)
  switch (a.iterations) {
    case 1: OFDIS_XCU_LAUNCH(1); break;
    case 2: OFDIS_XCU_LAUNCH(2); break;
    default: OFDIS_XCU_LAUNCH(3); break;
  }
#undef OFDIS_XCU_LAUNCH
  return hipGetLastError();
}

}  // namespace OFDIS_KNS
}  // namespace ofdis
