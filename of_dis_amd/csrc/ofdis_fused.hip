// ofdis_fused.hip -- one TV fixed-point iteration in ONE kernel (gray, level height <= 64):
//     compute_smoothness + compute_data + 2 x sub_laplacian  (opticalflow_aux.c:123-199, 310-438)
//     -> sor_coupled                                          (solver.c:77-421)
//
// The system coefficients of a pixel depend only on un-solved quantities (wx, wy, the du/dv of BEFORE
// this solver call, the level's derivatives), so they can be produced in any order -- in particular in
// the order the wavefront SOR consumes them.  This kernel is the SOR of ofdis_sor.hip (lane = image row,
// step t -> column t - j, NS software-pipelined sweeps) with its nine row loads replaced by a producer
// that runs a few diagonals ahead in the same wavefront and hands each pixel's seven coefficients over
// IN REGISTERS.  Per pixel and iteration HBM sees the pixel's records (8 derivatives; wx, wy, mask; du, dv) and the
// final du, dv: 60 B instead of 141 B for the tv_system + sor pair (the 7-plane system never exists in
// memory), and the producer's arithmetic fills the issue slots the SOR's dependency chains leave empty.
// The operands are the "sdiag" records written by ofdis_prep.hip (ofdis_dev.h: sdiag_index): one diag row of a strip is
// h consecutive records, so a step costs two 16-byte loads (derivatives; all zero where the warp's mask is zero), one
// 8-byte load (wx, wy) and one 8-byte load / store (du, dv) per lane -- 5 memory instructions instead of the 15 of one
// 4-byte plane per operand, 56 bytes per pixel and iteration.
// Strips: S frames laid side by side form one image of S*w columns whose diag rows wrap at S*w; a wavefront walks the
// whole strip, so the fill / drain of the skewed sweep (h steps) is paid once per S frames.
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
// Small batches (MW = true, "multi-wave"): with fewer wavefronts than SIMDs the kernel above is bound by the latency of
// ONE wavefront walking n_inner*w + h diagonal steps (1.1 ms for the three levels of operating point 2), whatever the
// batch size.  The fixed-point iterations of a level form a second pipeline: iteration k+1 needs, around a pixel, only
// du/dv of iteration k, and those are final 2(NS-1) steps after sweep 0 passed.  So a WORKGROUP of n_inner wavefronts
// takes one frame group, wavefront k runs iteration k and trails wavefront k-1 by MW_LAG diagonal steps; the du/dv rows
// travel from wavefront k to k+1 through an LDS ring (8 rows deep, indexed by the unwrapped step number, so the two
// visits of a wrapped diag row never alias), one workgroup barrier per step keeps the wavefronts in lock step.  Steps per
// level: w + h + (n_inner-1)*MW_LAG instead of n_inner*w + h (208 / 124 / 86 instead of 568 / 348 / 206 at 1024x436
// op-2).  Same arithmetic, same order: bit-identical results.  The launcher picks this variant while the batch leaves
// SIMDs idle (launch_tv_fused).
//
// Border rules: horizontal neighbours are clamped exactly as the reference's shifted row copies
// (image.c:436-464); for the 3-tap vertical filter the clamped form c0*s0 + c1*s0 + c2*s1 has the same
// value as the reference's folded (c0+c1)*s0 + c2*s1 because c1 = -0 (image.c:376-399).
#include <stdlib.h>

#include "ofdis_fused.h"

namespace ofdis {
namespace OFDIS_KNS {  // the arithmetic contract this file is being compiled for (ofdis_dev.h)

// Border handling.  The reference special-cases every border (solver.c:77-421, opticalflow_aux.c:172-199).  Here
// an edge weight that does not exist IS zero -- sh = 0 on the last column, sv = 0 on the last row (and in the
// idle lanes, and DPP shifts zero-fill lane 0), hence hl = 0 on column 0 and vt = 0 on row 0 -- and the terms
// are added unconditionally: x + (+-0 * finite) == x bit for bit unless x is -0, and none of the accumulators
// can be -0 at that point (b1, b2 start at +0 and only ever add/subtract, which never yields -0 from +0; the
// neighbour sums end with "+ b").  "finite" holds because every lane always works on real pixels: before its
// first and after its last column a lane computes wrapped columns of real data whose results are not stored;
// every such system has at least one positive edge weight (quarter_alpha > 0 is a launch condition, a pixel is
// never first and last column at once), so det >= (sum of weights)^2 > 0; the slot ring starts with a unit
// diagonal and unit weights.
constexpr int MW_LAG = 8;        // MODE 1: steps between consecutive iterations' wavefronts: PDU + 2*(NS-1) + 1 for NS = 3, PDU = 3
                                 // (du / dv come from LDS: they are read in the step that first uses them; 10 with PDU = PDW until
                                 // round 4 -- every step of lag widens the window in which the iterations' wavefronts must find
                                 // each other's derivative rows in the L2)
constexpr int MW_MAX_ITERS = 8;  // MODE 1: wavefronts per workgroup (= fixed-point iterations it handles)
constexpr int MW_RING = 8;       // LDS rows of du/dv per iteration (a power of two)
constexpr int SP_LAG = 9;        // MODE 2: du/dv are read 4 rows ahead instead of PDW = 5: 4 + 2*(NS-1) + 1
constexpr int SP_MAX_ITERS = 6;  // MODE 2: 2 wavefronts per iteration, 12 per workgroup (3 per SIMD: 168 VGPRs each)


// MODE 0: one wavefront walks all fixed-point iterations of its strip(s) (the throughput kernel).
// MODE 1: "multi-wave" -- one workgroup per frame group, wavefront k runs iteration k (header comment); S = 1.
// MODE 2: "split" -- as MODE 1 with every iteration's work divided between TWO wavefronts: a producer (row loads, flow
//         gradients, smoothness, data term, Laplacian: parts 1-4 of a step, ~2/3 of its instructions) and a solver (block
//         inverse and the NS pipelined sweeps, part 5).  The producer hands each pixel's FSlot to its solver through a
//         double-buffered LDS array one step later -- exactly the distance the single-wave kernel has between producing
//         the slot of row t+1 and consuming the slot of row t -- and the solver hands the finished du/dv row to the next
//         iteration's producer through the same LDS ring MODE 1 uses.  A lone wavefront issues one instruction per ~6
//         clocks (dependent-issue latency), so halving the instructions per wavefront and step nearly halves the step.
// NOC = 3 (round 6, MODE 0 only): RGB levels of at most 64 rows -- three derivative record arrays [c][records][8] (written by
// derivatives_kernel in its record form, ofdis_tv.hip), the data term of opticalflow_aux.c:383-427; everything else is shared.
template <int NS, bool BRIGHT, int MODE, int NOC = 1>
// MODE 1 is held to 168 registers = three wavefronts per SIMD (amdgpu_waves_per_eu): three workgroups of four iterations per
// compute unit instead of two.  That only pays without scratch: with the (wx, wy) ring and the run buffer of the wavefront that
// writes the flow in registers (24 that one of a workgroup's wavefronts uses) the allocator spilled 7 and level 3 of the
// headline took 5.84 instead of 4.41 ms; with both in LDS (wdring, obring: two ds_write and two ds_read per step of that one
// wavefront) nothing spills: 4.30-4.34 -> 4.13-4.15 ms on the same box (fused contract; exact: 167 registers, unchanged time).
__global__ __launch_bounds__(MODE == 2 ? 128 * SP_MAX_ITERS : (MODE == 1 ? 64 * MW_MAX_ITERS : 256))
__attribute__((amdgpu_waves_per_eu(NOC == 3 ? 2 : (MODE == 1 ? 3 : 1)))) void tv_fused_kernel(const FusedArgs a, const int R) {
  constexpr int U = 6;
  constexpr bool MW = MODE != 0;
  constexpr int MAXIT = MODE == 2 ? SP_MAX_ITERS : MW_MAX_ITERS;
  // du/dv rows handed from iteration k to iteration k+1: [k][step & 7][lane]
  __shared__ float2 xring[MW ? (MAXIT - 1) * MW_RING * 64 : 1];
  // MODE 2: FSlot of the pixel row handed from an iteration's producer to its solver: [iteration][step & 1][field][lane]
  __shared__ float sring[MODE == 2 ? SP_MAX_ITERS * 2 * SLOT_FLOATS * 64 : 1];
  // MODE 1: the (wx, wy) ring of the wavefront that writes the flow (Wd below) lives here: twelve registers that only one
  // of the workgroup's wavefronts uses
  __shared__ float2 wdring[MODE == 1 ? 6 * 64 : 1];
  __shared__ float2 obring[MODE == 1 ? 6 * 64 : 1];  // ... and its run buffer (ob below), [lane][column of the run]
  // prefetch distances: the (wx, wy) record (MODE 0: and du, dv) of diag row t+PDW and the derivative record of row
  // t+PDD are requested at step t; first uses are rows t+3 (uu, vv) and t+1 (data term): two steps of slack.  (One step of
  // slack, 128 VGPRs = 4 wavefronts per SIMD, measured the same kernel time: occupancy is not what limits this kernel.)
  // (MODE 1 with the derivative rows requested six and the (wx, wy) rows eight steps ahead -- a six-row register ring and a
  // staging ring, 197 VGPRs -- measured 4.30-4.33 against 4.34-4.36 ms on level 3 of the headline: memory latency is not what
  // its steps wait for; not kept)
  // (MODE 1 with the step loop compiled once per role -- the last iteration's wavefront writes the flow, the others hand their
  // rows on: 223 instead of 292 instructions per step for the others, 278 for the last -- measured 4.48 against 4.29 ms on the
  // same box, 4.46 with the roles rotated between the workgroups that share a compute unit: not kept)
  // (RGB: the derivative rows two steps ahead -- a step is twice as long, and 24 registers less keep the kernel at two
  // wavefronts per SIMD)
  constexpr int PDW = 5, PDD = NOC == 3 ? 2 : 3;
  constexpr int PDU = MODE == 2 ? 4 : (MODE == 1 ? 3 : PDW);  // read-ahead of du/dv (MODE 1 / 2: from LDS, in / one step before the step of their first use)
  constexpr int LAG = MODE == 2 ? SP_LAG : MW_LAG;
  static_assert(2 * (NS - 1) + 1 < U, "slot ring too small for this many pipelined sweeps");
  static_assert(!MW || LAG >= PDU + 2 * (NS - 1) + 1, "a row must be published before the next iteration reads it");
  const int w = a.t.w, h = a.t.h;
  const int rw = a.S * w;  // diag rows of a strip = its columns
  const int lane = threadIdx.x & 63;
  // MW: one frame group per workgroup; MODE 1: wavefront `it` runs iteration `it`; MODE 2: wavefronts 0..n-1 are the
  // producers of iterations 0..n-1, wavefronts n..2n-1 their solvers (a producer and its solver share a SIMD when n = 4)
  const int wv = MW ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  const int n_iters = MODE == 2 ? (int)(blockDim.x >> 7) : (MW ? (int)(blockDim.x >> 6) : 1);
  const bool is_solver = MODE == 2 && wv >= n_iters;
  const int it = MODE == 2 ? (is_solver ? wv - n_iters : wv) : wv;
  const bool do_p = MODE != 2 || !is_solver;  // parts 1-4 of a step
  const bool do_s = MODE != 2 || is_solver;   // part 5
  const int wid = MW ? (int)blockIdx.x : __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int G = 64 / R;  // strips per wavefront (lane groups of R lanes)
  const int nstrips = a.t.nframes / a.S;  // (the launcher picks S among the divisors of nframes)
  const int s0 = wid * G;                 // first strip of this wavefront
  if (s0 >= nstrips) return;  // whole wave idle (uniform; MW: the whole workgroup)
  int fl = lane / R;          // strip of this lane within the wavefront
  const int jr = lane % R;
  const bool row_ok = (s0 + fl < nstrips) && (jr < h);
  if (s0 + fl >= nstrips) fl = nstrips - 1 - s0;
  const int j = jr < h ? jr : h - 1;
  const bool has_top = j > 0, has_bot = j < h - 1;
  const float omega = a.omega, qa = a.quarter_alpha, hd3 = a.half_delta_over3, hg3 = a.half_gamma_over3;
  // (x, y-1) / (x, y+1) live in the neighbouring lanes; DPP wave shifts fill lane 0 / 63 with zero.  Every consumer either
  // selects the value away on its border row or multiplies it by an edge weight that is zero there, see "Border handling".
  auto from_prev = [&](float x) { return wave_from_prev(x); };
  auto from_next = [&](float x) { return wave_from_next(x); };

  // one buffer resource per record array, based at the wavefront's first strip: a lane's byte offset within it is
  // constant, the moving part (the diag row) is a scalar offset
  const int nst = min(G, nstrips - s0);
  const size_t strip_recs = (size_t)rw * h;
  auto rsrc = [&](const float* base, int rec_floats) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)s0 * strip_recs * rec_floats), 0,
                                             (int)(nst * strip_recs * rec_floats * 4), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsW = rsrc(a.wrec, 2), rsU = rsrc(a.uv, 2);
  // (RGB: channel c's records follow channel c-1's for all frames of this launch)
  const __amdgpu_buffer_rsrc_t rsD = rsrc(a.d8, 8);
  const __amdgpu_buffer_rsrc_t rsD1 = rsrc(a.d8 + (NOC == 3 ? (size_t)a.t.nframes * w * h * 8 : 0), 8);
  const __amdgpu_buffer_rsrc_t rsD2 = rsrc(a.d8 + (NOC == 3 ? (size_t)a.t.nframes * w * h * 16 : 0), 8);
  const int vrec = fl * (int)strip_recs + j;  // this lane's record within a diag row 0 of its strip
  const int vo8 = vrec * 32, vo2 = vrec * 8;
  auto asf = [](unsigned u) { return __builtin_bit_cast(float, u); };

  // this lane's row of the AoS output (with flow_out; multi-wave variants: S = 1, a strip is a frame; throughput variant:
  // row j of the strip's FIRST frame -- frame fs of the strip follows fs * npx pixels later)
  const int npx = w * h;
  float2* const flow_row = a.flow_out
                               ? reinterpret_cast<float2*>(a.flow_out) + ((size_t)(s0 + fl) * a.S * npx + (size_t)j * w)
                               : nullptr;
  // MODE 0 / 1: the wavefront that runs the last fixed-point iteration writes the refined flow itself, in runs (aos_emit)
  const bool aos_out = MODE != 2 && a.flow_out != nullptr;

  FRow W[6];
  FDer D[PDD][NOC];
  float uu[3], vv[3], sm[3];
  FSlot slot[6];
#pragma unroll
  // Fill-phase slots: unit diagonal AND unit edge weights.  The first slot a lane produces takes its left / top weights
  // from these; with zeros, a lane whose first fill pixel is a last-column pixel of the last image row (w == h) had all
  // four weights and (derivative ring still empty) the whole system zero: det = 0, NaN, and 0 * NaN poisoned the row.
  for (int r = 0; r < 6; ++r) { W[r] = FRow{0, 0, 0, 0}; slot[r] = FSlot{1, 0, 1, 0, 0, 1, 1, 0, 0, 0, 0}; }
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
  float ldx = 0.0f, ldy = 0.0f;  // wx, wy of the pixel row being assembled minus those of the row before (set below)
  // MODE 0 with an AoS output: (wx, wy) of the rows the sweeps are still working on (row rho at index (rho + 3) % 6: written
  // when the row's system is assembled, read 2 (NS - 1) + 1 steps later when its last sweep finishes) and the refined flow
  // of the last U finished columns of this lane's image row, written as one run per U steps
  float2 Wd[6], ob[U];
#pragma unroll
  for (int r = 0; r < 6; ++r) Wd[r] = make_float2(0.0f, 0.0f);
#pragma unroll
  for (int r = 0; r < U; ++r) ob[r] = make_float2(0.0f, 0.0f);
  int ox = 0, ooff = 0;  // x and frame offset (pixels) within the strip of the column this lane finishes, last iteration

  // The refined flow itself, uu = wx + du, vv = wy + dv of the LAST fixed-point iteration (refine_variational.cpp:209-221,
  // 92-99), AoS row-major -- tv_finish and the round trip of du, dv through memory disappear.  `slot` = ob[u], `c` = the
  // column within the last iteration's pass over the strip this lane finishes in this step (< 0: not there yet).  In a
  // lane's image row the U columns finished since the last flush are consecutive pixels: one 8 U-byte run instead of U
  // scattered 8-byte stores (a run that crosses into the strip's next frame, or the ends of the pass, goes pixel by pixel).
  auto ob_put = [&](int e, const float2& v) {
    if constexpr (MODE == 1) obring[lane * U + e] = v;
    else ob[e] = v;
  };
  auto ob_get = [&](int e) -> float2 {
    if constexpr (MODE == 1) return obring[lane * U + e];
    else return ob[e];
  };
  auto aos_emit = [&](int slot, bool flush, const float2& wq, float du, float dv, int c) {
    ob_put(slot, make_float2(wq.x + du, wq.y + dv));
    if (flush) {
      const int c0 = c - (U - 1);
      if (row_ok & (c0 >= 0) & (c < rw) & (ox >= U - 1)) {
        typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
        f4a8* d = reinterpret_cast<f4a8*>(flow_row + ooff + ox - (U - 1));
#pragma unroll
        for (int e = 0; e < U / 2; ++e) {
          const float2 p0 = ob_get(2 * e), p1 = ob_get(2 * e + 1);
          d[e] = f4a8{p0.x, p0.y, p1.x, p1.y};
        }
      } else if (row_ok & (c >= 0) & (c0 < rw)) {
#pragma unroll
        for (int e = 0; e < U; ++e) {
          const int ce = c0 + e;
          int xe = ox - (U - 1) + e, oe = ooff;
          if (xe < 0) { xe += w; oe -= npx; }
          if ((ce >= 0) & (ce < rw)) flow_row[oe + xe] = ob_get(e);
        }
      }
    }
    if (c >= 0) {  // x / frame of the next column
      ++ox;
      if (ox == w) { ox = 0; ooff += npx; }
    }
  };
  auto wrap_row = [&](int r) { r %= rw; return r < 0 ? r + rw : r; };  // diag row of the strip
  auto wrap_col = [&](int c) { c %= w; return c < 0 ? c + w : c; };    // column within a frame
  auto next_row = [&](int r) { return (r + 1 == rw) ? 0 : r + 1; };
  // du/dv of the previous iteration for the row with unwrapped step number tau (MW): from the LDS ring; zero in the
  // first fixed-point iteration (image_erase, refine_variational.cpp:186-187)
  auto ring_uv = [&](FRow& r, int tau) {
    if (it == 0) {
      r.du = 0.0f; r.dv = 0.0f;
    } else {
      const float2 v = xring[((it - 1) * MW_RING + (tau & (MW_RING - 1))) * 64 + lane];
      r.du = v.x; r.dv = v.y;
    }
  };
  // tau = unwrapped step number of the row (MW only): the LDS ring slot
  // (MODE 0) zero_uv: this lane's pixel of the row still belongs to the FIRST fixed-point iteration, where du = dv = 0 (or lies
  // past the last iteration's columns, where the value only ever meets a zero edge weight but must be finite)
  // (image_erase, refine_variational.cpp:186-187): the load gets an offset beyond the resource, for which the hardware
  // returns +0 without a memory access -- the caller never has to clear the array and iteration 1 reads 8 bytes less per pixel
  auto load_w = [&](FRow& r, int drow, int tau, bool zero_uv) {
    const int o = drow * h * 8;
    const auto t = __builtin_amdgcn_raw_buffer_load_b64(rsW, vo2, o, 0);
    // (through scalars: __builtin_bit_cast applied directly to a vector element reads element 0, ROCm 7.2)
    const unsigned t0 = t[0], t1 = t[1];
    r.wx = asf(t0); r.wy = asf(t1);
    if constexpr (MODE == 0) {
      const auto q = __builtin_amdgcn_raw_buffer_load_b64(rsU, zero_uv ? 0x7ffffff0 : vo2, o, 0);
      const unsigned q0 = q[0], q1 = q[1];
      r.du = asf(q0); r.dv = asf(q1);
    }  // MODE 1 / 2: du/dv follow from the LDS ring, PDW - PDU steps later
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

  // ring index of diag row rho is (rho + 3) mod ring size; the loop variable is k = t + 3, u = k % 6,
  // so row t + c sits at index (u + c) % size.
  if constexpr (MW) {  // the ring of this iteration's successor starts with zeros (finite: see "Border handling")
    if (do_s && it < MAXIT - 1) {
#pragma unroll
      for (int q = 0; q < MW_RING; ++q) xring[(it * MW_RING + q) * 64 + lane] = make_float2(0.0f, 0.0f);
    }
    __syncthreads();
    for (int n = 0; n < it * LAG; ++n) __syncthreads();  // trail the previous iteration by LAG steps
  }
  // prologue: W rows -1, 0, 1 (indices 2, 3, 4)
  if (do_p) {
    load_w(W[2], wrap_row(-1), -1, true);
    load_w(W[3], wrap_row(0), 0, true);
    if (PDW == 5) load_w(W[4], wrap_row(1), 1, true);
    if constexpr (MW) {  // du/dv of the rows before the loop's first one (row t + PDU at t = -3)
      ring_uv(W[2], -1);
      if (PDU == 4) ring_uv(W[3], 0);
    }
  }
  int rowW = wrap_row(PDW - 3);  // next W row to load (row t+PDW at t = -3)
  int tauW = PDW - 3;            // ... and its unwrapped step number
  int rowD = wrap_row(PDD - 3);  // next D row to load (row t+PDD at t = -3)
  int srow = wrap_row(-3 - 2 * (NS - 1));  // row finished by the last sweep at step t = -3
  int x2 = wrap_col(-1 - j);               // this lane's x (within its frame) on diag row t+2
  bool x1_last = (wrap_col(-2 - j) == w - 1);  // row t+1 is the last column of this lane's frame

  const int wtot = (MW ? 1 : a.n_inner) * rw;  // columns per lane over all iterations of this wavefront
  const int tend = (wtot - 1) + (h - 1) + 2 * (NS - 1);
  int ig = -3 - j - 2 * (NS - 1);  // global column (over all iterations) the last sweep finishes at step t
  bool first_w = !MW;              // row t+5 at t = -3 is column 2 - j: first iteration (w >= 16)
  int taus = -3 - 2 * (NS - 1);    // unwrapped step number of the row the last sweep finishes (MW)
  int taut = -3;                   // unwrapped step number t (MODE 2: the slot hand-over buffer)
  for (int k0 = 0; k0 <= tend + 3; k0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // step t = k0 + u - 3; up to U-1 steps past tend are executed: every pixel is then out of range
      if (do_p) {
        // ---- (1) loads: W row t+5, D row t+3
        load_w(W[(u + PDW) % 6], rowW, tauW, first_w);  // (first_w: this lane's column on row t+5 is in the first iteration)
        if constexpr (MW) ring_uv(W[(u + PDU) % 6], tauW - (PDW - PDU));
        rowW = next_row(rowW);
        ++tauW;
        load_d(D[(u + PDD) % PDD], rowD);
        rowD = next_row(rowD);
        // ---- (2) uu, vv of row t+3 (refine_variational.cpp:210-216: uu = wx + du of before this call)
        {
          const FRow& r = W[(u + 3) % 6];
          uu[u % 3] = r.wx + r.du;
          vv[u % 3] = r.wy + r.dv;
        }
        // ---- (3) smoothness of row t+2 (opticalflow_aux.c:128-140)
        const bool x2_last = (x2 == w - 1);
        {
          const float uc = uu[(u + 2) % 3], vc = vv[(u + 2) % 3];
          float ul = uu[(u + 1) % 3], vl = vv[(u + 1) % 3];                  // (x-1, y)
          float ur = uu[u % 3], vr = vv[u % 3];                              // (x+1, y): row t+3
          float ut = from_prev(uu[(u + 1) % 3]), vt = from_prev(vv[(u + 1) % 3]);  // (x, y-1)
          float ub = from_next(uu[u % 3]), vb = from_next(vv[u % 3]);              // (x, y+1)
          if (x2 == 0) { ul = uc; vl = vc; }
          if (x2_last) { ur = uc; vr = vc; }
          if (!has_top) { ut = uc; vt = vc; }
          if (!has_bot) { ub = uc; vb = vc; }
          // The reference's derivative is ux = -0.5 ul + (-0 uc) + 0.5 ur.  The middle tap adds a signed zero (it changes at
          // most the sign of a zero result, and each derivative is only ever squared); the outer taps are exact scalings,
          // so ux = 0.5 (ur - ul) with the same single rounding, ux^2 = 0.25 (ur - ul)^2, and a sum of such squares is 0.25
          // times the sum of the unscaled ones, rounding for rounding (powers of two commute with every rounding; a square
          // small enough to underflow is 30 orders of magnitude below half an ulp of the 1e-6 added next).  Four
          // subtractions, four squares, three additions and ONE scaling instead of eight scalings on top of them.
          const float ex = ur - ul, fx = vr - vl, ey = ub - ut, fy = vb - vt;
          sm[(u + 2) % 3] = fdiv_by_sqrt(qa, 0.25f * (ex * ex + ey * ey + fx * fx + fy * fy) + EPS_SMOOTH);
        }
        // ---- (4) system of pixel row tau = t+1 (opticalflow_aux.c:150-163, 172-199, 342-427)
        //      x of row tau is x2 of the previous step: "last column" was x2_last then
        {
          const float sc = sm[(u + 1) % 3];
          const float s_r = sm[(u + 2) % 3];
          const float s_d = from_next(sm[(u + 2) % 3]);
          const float sh_c = x1_last ? 0.0f : sc + s_r;
          const float sv_c = has_bot ? sc + s_d : 0.0f;
          const FRow& rc = W[(u + 1) % 6];
          const FRow& rm = W[u % 6];        // row tau-1
          const FRow& rp = W[(u + 2) % 6];  // row tau+1
          float a11, a12, a22, b1, b2;
          if constexpr (NOC == 1) data_term_gray<BRIGHT>(D[(u + 1) % PDD][0], rc.du, rc.dv, hd3, hg3, a11, a12, a22, b1, b2);
          else data_term_rgb<BRIGHT>(D[(u + 1) % PDD], rc.du, rc.dv, hd3, hg3, a11, a12, a22, b1, b2);
          const float wx_u = from_prev(rm.wx), wy_u = from_prev(rm.wy);
          const float wx_d = from_next(rp.wx), wy_d = from_next(rp.wy);
          const float sh_l = slot[u % 6].sh;         // (s_l + sc), 0 on column 0
          const float sv_t = from_prev(slot[u % 6].sv);  // (s_u + sc), 0 on row 0
          // (the difference to the left neighbour is the previous step's difference to the right neighbour)
          const float rdx = rp.wx - rc.wx, rdy = rp.wy - rc.wy;
          b1 -= sh_l * ldx;
          b2 -= sh_l * ldy;
          b1 += sh_c * rdx;
          b2 += sh_c * rdy;
          ldx = rdx; ldy = rdy;
          if constexpr (MODE == 0) Wd[(u + 1) % 6] = make_float2(rc.wx, rc.wy);
          if constexpr (MODE == 1) {
            if (it == n_iters - 1) wdring[((u + 1) % 6) * 64 + lane] = make_float2(rc.wx, rc.wy);
          }
          b1 -= sv_t * (rc.wx - wx_u);
          b2 -= sv_t * (rc.wy - wy_u);
          b1 += sv_c * (wx_d - rc.wx);
          b2 += sv_c * (wy_d - rc.wy);
          FSlot& o = slot[(u + 1) % 6];
          o.a11 = a11; o.a12 = a12; o.a22 = a22; o.b1 = b1; o.b2 = b2; o.sh = sh_c; o.sv = sv_c;
          o.dur = rp.du; o.dvr = rp.dv;
          o.hl = sh_l; o.vt = sv_t;
          if constexpr (MODE == 2) {  // hand the slot of row t+1 to this iteration's solver (it reads it in the next step)
            float* sr = sring + ((it * 2 + ((taut + 1) & 1)) * SLOT_FLOATS) * 64 + lane;
            sr[0 * 64] = a11; sr[1 * 64] = a12; sr[2 * 64] = a22; sr[3 * 64] = b1; sr[4 * 64] = b2; sr[5 * 64] = sh_c;
            sr[6 * 64] = sv_c; sr[7 * 64] = rp.du; sr[8 * 64] = rp.dv; sr[9 * 64] = sh_l; sr[10 * 64] = sv_t;
          }
        }
        x1_last = x2_last;
        x2 = x2_last ? 0 : x2 + 1;
      }
      if (do_s) {
        if (MODE == 2 && taut > -3) {  // the slot of row t, published by the producer one step ago (its first one is
          // row -2; before that the fill-phase slot stays)
          const float* sr = sring + ((it * 2 + (taut & 1)) * SLOT_FLOATS) * 64 + lane;
          FSlot& c = slot[u % 6];
          c.a11 = sr[0 * 64]; c.a12 = sr[1 * 64]; c.a22 = sr[2 * 64]; c.b1 = sr[3 * 64]; c.b2 = sr[4 * 64]; c.sh = sr[5 * 64];
          c.sv = sr[6 * 64]; c.dur = sr[7 * 64]; c.dvr = sr[8 * 64]; c.hl = sr[9 * 64]; c.vt = sr[10 * 64];
        }
        // ---- (5) SOR step t (ofdis_sor.hip): sweep 0 reaches pixel (j, t - j); block inverse (solver.c:100-110)
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
            bu = from_next(c.dur);
            bv = from_next(c.dvr);
          } else {
            ou = ru2[s - 1]; ov = rv2[s - 1];
            rgu = ru[s - 1]; rgv = rv[s - 1];
            bu = from_next(ru[s - 1]);
            bv = from_next(rv[s - 1]);
          }
          const float tu = from_prev(ru[s]), tv = from_prev(rv[s]);
          const float lu = ru[s], lv = rv[s];
          const float s1 = c.sh * rgu + c.vt * tu + c.sv * bu + c.b1;
          const float s2 = c.sh * rgv + c.vt * tv + c.sv * bv + c.b2;
          const float B1 = c.hl * lu + s1, B2 = c.hl * lv + s2;
          nu[s] = ou + omega * (c.a11 * B1 + c.a12 * B2 - ou);
          nv[s] = ov + omega * (c.a12 * B1 + c.a22 * B2 - ov);
        }
        {
          if (!MW) {  // for the next step's row: its column is in the first iteration, or PAST the last one -- nothing the
            // launch reads there may come from memory (one iteration never writes the array: whatever the allocation held)
            const int cw = ig + (PDW + 1 + 2 * (NS - 1));
            first_w = (cw < rw) | (cw >= wtot);
          }
          if (MW && it < n_iters - 1) {  // hand the row to the next iteration (lanes outside their columns publish finite
            // values nobody reads as a pixel: the reader's lane is outside its columns at the same step number)
            xring[(it * MW_RING + (taus & (MW_RING - 1))) * 64 + lane] = make_float2(nu[NS - 1], nv[NS - 1]);
          } else if (MODE == 2 && a.flow_out) {  // last iteration: the refined flow itself, AoS (one 8-byte store per lane;
            // a lane's consecutive columns fill its cache lines over the next steps; S = 1)
            const auto q = __builtin_amdgcn_raw_buffer_load_b64(rsW, vo2, srow * h * 8, 0);
            const unsigned q0 = q[0], q1 = q[1];
            if (row_ok && ig >= 0 && ig < wtot)
              flow_row[ig] = make_float2(asf(q0) + nu[NS - 1], asf(q1) + nv[NS - 1]);
          } else if (MODE == 1 && a.flow_out) {  // last iteration's wavefront: the refined flow, in runs
            aos_emit(u, u == U - 1, wdring[((u - 2 * (NS - 1) + 12) % 6) * 64 + lane], nu[NS - 1], nv[NS - 1], ig);
          } else if (!MW) {
            // (no branch: a lane outside its rows / columns stores at an offset beyond the resource, which the hardware drops)
            const u32x2 v = {__builtin_bit_cast(unsigned, nu[NS - 1]), __builtin_bit_cast(unsigned, nv[NS - 1])};
            const int lastc = ig - (wtot - rw);  // >= 0: this column belongs to the last fixed-point iteration
            const bool on = row_ok & (ig >= 0) & (aos_out ? lastc < 0 : ig < wtot);
            __builtin_amdgcn_raw_buffer_store_b64(v, rsU, on ? vo2 : 0x7ffffff0, srow * h * 8, 0);
            if (aos_out) aos_emit(u, u == U - 1, Wd[(u - 2 * (NS - 1) + 12) % 6], nu[NS - 1], nv[NS - 1], lastc);
          } else if (row_ok && ig >= 0 && ig < wtot) {
            const u32x2 v = {__builtin_bit_cast(unsigned, nu[NS - 1]), __builtin_bit_cast(unsigned, nv[NS - 1])};
            __builtin_amdgcn_raw_buffer_store_b64(v, rsU, vo2, srow * h * 8, 0);
          }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          ru2[s] = ru[s]; rv2[s] = rv[s];
          ru[s] = nu[s]; rv[s] = nv[s];
        }
      }
      srow = next_row(srow);
      ++ig;
      ++taus;
      ++taut;
      if constexpr (MW) mw_step_barrier();  // what was published in this step is read >= 1 step later
    }
  }
  if constexpr (MW) {  // keep the barrier count equal for all wavefronts of the workgroup
    for (int n = 0; n < (n_iters - 1 - it) * LAG; ++n) __syncthreads();
  }
}

bool tv_fused_supported(const TvGeom& t, int iterations) {
  // (65 ... 256 rows: the two- to four-wavefront form of the throughput mapping, ofdis_fused_tall.hip)
  return ((t.noc == 1 || t.noc == 3) && t.h >= 2 && t.h <= 64 && t.w >= 16 && iterations >= 1 && iterations <= 3) ||
         tv_fused_tall_supported(t, iterations);
}

bool tv_fused_params_ok(float qa, float hd3, float hg3) {
  auto ok = [](float v) { return v == 0.0f || (v >= 1e-12f && v <= 1e12f); };
  return qa > 0.0f && ok(qa) && ok(hd3) && ok(hg3);  // qa == 0: no smoothness at all, singular systems possible
}

// When the multi-wave variants are launched (measured at operating point 2, ms per step, T = this limit on the frame
// groups = workgroups of a launch; batches >= 1024 run as two pipelined sub-batches):
//      512 pairs: T >= 512: 0.92, T = 256: 1.02            1024 (2 x 512): T = 512: 1.65, 256: 1.75, 0: 1.89
//     2048 (2 x 1024): T = 0: 2.93, 256 / 512: 3.00         4096 (2 x 2048): T = 0 / 256: 5.06-5.16, 512: 5.39-5.44
// i.e. up to two rounds of workgroups (256 CUs) they beat the throughput mapping as long as the WHOLE batch is small;
// once the other sub-batch has enough work to fill the SIMDs a workgroup that owns a CU for a whole level only gets
// in the way (at 4096 pairs per launch they take 4.5-4.9 ms against 2.65, profiles/README.md round 3).  Rule: at most
// FusedArgs::mw_max_groups (default 512) frame groups in the launch AND at most 1024 frames in the whole batch.
constexpr int MW_MAX_BATCH_FRAMES = 1024;

int tv_fused_mode(const FusedArgs& a, const FusedXcu* x) {
  const int h = a.t.h;
  if (h > 64) return 0;  // two to four wavefronts per strip (ofdis_fused_tall.hip): the throughput mapping only, whatever the batch
  if (a.t.noc == 3)  // RGB: one frame per strip, one wavefront per frame or -- tp_pipe -- per fixed-point iteration
    return (a.tp_pipe && a.S == 1 && a.n_inner >= 2 && a.n_inner <= MW_MAX_ITERS) ? 1 : 0;
  const int R = h <= 16 ? 16 : (h <= 32 ? 32 : 64);
  const int groups = (a.t.nframes + 64 / R - 1) / (64 / R);
  const int total = a.total_frames > 0 ? a.total_frames : a.t.nframes;
  if (x && x->xbuf && x->err && a.S == 1 && a.n_inner >= 2 && groups <= x->max_groups && total <= MW_MAX_BATCH_FRAMES) return 3;
  const bool mw = a.S == 1 && a.n_inner >= 2 && a.n_inner <= MW_MAX_ITERS && groups <= a.mw_max_groups &&
                  (total <= MW_MAX_BATCH_FRAMES || a.mw_max_groups >= (1 << 30));
  if (mw) return (a.split && a.n_inner <= SP_MAX_ITERS) ? 2 : 1;
  // throughput regime: one wavefront per strip group (0), or -- strips allowed -- a wavefront per fixed-point iteration (1)
  return (a.tp_pipe && a.n_inner >= 2 && a.n_inner <= MW_MAX_ITERS) ? 1 : 0;
}

hipError_t launch_tv_fused(const FusedArgs& a, hipStream_t s, bool* wrote_flow, const FusedXcu* x) {
  if (wrote_flow) *wrote_flow = false;
  if (!tv_fused_supported(a.t, a.iterations) || a.n_inner < 1 || a.S < 1 || a.t.nframes % a.S != 0 ||
      !tv_fused_params_ok(a.quarter_alpha, a.half_delta_over3, a.half_gamma_over3))
    return hipErrorInvalidValue;
  const int h = a.t.h;
  if (h > 64) {  // 65 ... 256 rows
    if (wrote_flow) *wrote_flow = a.flow_out != nullptr;
    return launch_tv_fused_tall(a, s, a.tall_group);
  }
  const int R = h <= 16 ? 16 : (h <= 32 ? 32 : 64);
  const int G = 64 / R;
  const int nstrips = a.t.nframes / a.S;
  const int waves = (nstrips + G - 1) / G;
  const int blocks = (waves + 3) / 4;
  const bool bright = a.half_delta_over3 != 0.0f;
  // small batches: one workgroup per frame group, one (MODE 1) or two (MODE 2) wavefronts per fixed-point iteration
  const int mode = tv_fused_mode(a, x);
  if (wrote_flow) *wrote_flow = a.flow_out != nullptr;  // every mapping writes the refined AoS flow itself
  if (mode == 3) return launch_tv_fused_xcu(a, *x, waves, R, s);
  if (a.t.noc == 3) {
    if (a.S != 1) return hipErrorInvalidValue;
#define OFDIS_FUSED_RGB(NS)                                                                                            \
  if (mode == 1) {                                                                                                     \
    if (bright) hipLaunchKernelGGL((tv_fused_kernel<NS, true, 1, 3>), dim3(waves), dim3(64 * a.n_inner), 0, s, a, R);  \
    else hipLaunchKernelGGL((tv_fused_kernel<NS, false, 1, 3>), dim3(waves), dim3(64 * a.n_inner), 0, s, a, R);        \
  } else if (bright) hipLaunchKernelGGL((tv_fused_kernel<NS, true, 0, 3>), dim3(blocks), dim3(256), 0, s, a, R);       \
  else hipLaunchKernelGGL((tv_fused_kernel<NS, false, 0, 3>), dim3(blocks), dim3(256), 0, s, a, R)
    switch (a.iterations) {
      case 1: OFDIS_FUSED_RGB(1); break;
      case 2: OFDIS_FUSED_RGB(2); break;
      default: OFDIS_FUSED_RGB(3); break;
    }
#undef OFDIS_FUSED_RGB
    return hipGetLastError();
  }
#define OFDIS_FUSED_LAUNCH(NS)                                                                                         \
  if (mode == 2) {                                                                                                     \
    if (bright) hipLaunchKernelGGL((tv_fused_kernel<NS, true, 2>), dim3(waves), dim3(128 * a.n_inner), 0, s, a, R);    \
    else hipLaunchKernelGGL((tv_fused_kernel<NS, false, 2>), dim3(waves), dim3(128 * a.n_inner), 0, s, a, R);          \
  } else if (mode == 1) {                                                                                              \
    if (bright) hipLaunchKernelGGL((tv_fused_kernel<NS, true, 1>), dim3(waves), dim3(64 * a.n_inner), 0, s, a, R);     \
    else hipLaunchKernelGGL((tv_fused_kernel<NS, false, 1>), dim3(waves), dim3(64 * a.n_inner), 0, s, a, R);           \
  } else if (bright) hipLaunchKernelGGL((tv_fused_kernel<NS, true, 0>), dim3(blocks), dim3(256), 0, s, a, R);          \
  else hipLaunchKernelGGL((tv_fused_kernel<NS, false, 0>), dim3(blocks), dim3(256), 0, s, a, R)
  switch (a.iterations) {
    case 1: OFDIS_FUSED_LAUNCH(1); break;
    case 2: OFDIS_FUSED_LAUNCH(2); break;
    default: OFDIS_FUSED_LAUNCH(3); break;
  }
#undef OFDIS_FUSED_LAUNCH
  return hipGetLastError();
}

}  // namespace OFDIS_KNS
}  // namespace ofdis
