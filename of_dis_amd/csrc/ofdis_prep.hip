// ofdis_prep.hip -- image_warp + get_derivatives of the fused TV path in ONE row-marching kernel
//     image_warp        opticalflow_aux.c:18-60
//     get_derivatives   opticalflow_aux.c:65-116 (+ image.c:401-434, 466-502)
//
// A wavefront owns whole image rows of its frame(s): lane = column (two, three or four wavefronts side by side when 64 < w <= 128 / 192 / 256, sharing
// the exchanged rows and the staging area; two / four frames per wavefront when w <= 32 / 16) and marches down the rows.
// Per row r it
//   stage 0   warps the second image at (x, r) with the densified flow, forms avg = 0.5 (I2w + I1) and Iz = I2w - I1,
//             exchanges the row through LDS and takes the three HORIZONTAL 5-tap filters of that row (Ix, Ixz, then Ixx);
//   stage 1   takes the VERTICAL filters centred on row r-2 (Iy, Ixy, Iyz) from five-row register windows of avg, Ix, Iz;
//   stage 2   takes Iyy of row r-4 from the window of Iy.
// Every pixel is warped once and every filter tap is a register or one LDS word: no tile halo is recomputed (the tiled
// kernels of ofdis_tv.hip warp a pixel up to 1.9 times and spend ~440 instructions per pixel in warp + derivatives; this
// one ~125).  The vertical filters never read outside the image -- the reference folds the border coefficients of the
// first / last two rows (image.c:401-434) -- and the horizontal ones read replicated columns (image.c:466-502), which the
// LDS row provides as two pad entries on each side.
//
// Output = the "sdiag" records the fused TV kernel walks (ofdis_dev.h: sdiag_index):
//     d8  : { Ix, Iz, Ixx, Ixz, Iy, Ixy, Iyz, Iyy }, all zero where the warp's mask is zero
//     wrec: { wx, wy }
// The mask needs no storage of its own: it only ever multiplies the data term's weights (opticalflow_aux.c:352-427), and a
// pixel whose eight derivatives are zero gets exactly the same +0 coefficients as one whose weights are zero (finite
// images; ofdis_fused.hip data_term_gray).
// A record of row y is complete at iteration y+4 (Ix, Iz wait in the register windows, Ixy / Iyz two iterations in delay
// registers, Ixx / Ixz -- horizontal filters of rows that are still in the windows -- are taken then).  In the sdiag layout
// the records of one image row are h records apart (lane = column here, lane = row in the consumer), so the wavefront
// stages KD rows of records in LDS in diag order and then writes every diag row's KD consecutive records with
// consecutive lanes: full sectors, contiguous runs.  (Storing each piece from the lane that computed it -- 16 + 12 + 4
// bytes into a different cache line per lane -- was measured at 2.1 ms per 4096-pair step: partial-sector writes.)
//
// Small batches: the rows are cut into bands (one wavefront each) that recompute four rows of stage 0 and two of stage 1
// either side, so that a handful of frames still spreads over the chip.
#include <algorithm>
#include <type_traits>

#include "ofdis_kernels.h"
#include "ofdis_tvmath.h"

namespace ofdis {
namespace OFDIS_KNS {  // the arithmetic contract this file is being compiled for (ofdis_dev.h)

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// image_warp of one gray pixel (opticalflow_aux.c:18-60; same expression order as warp_pixel, ofdis_tv.hip) in two halves
// so that the taps of the NEXT row are in flight while this row is filtered: warp_taps() turns the flow into the mask, the
// fractions and the four element offsets within the padded plane, warp_mix() is the bilinear combination.
struct WarpTaps {
  float dx, dy, m;
  int o11, o12, o21, o22;
};
__device__ __forceinline__ WarpTaps warp_taps(int tmp_w, int pad, int w, int h, int i, int j, float fx, float fy) {
  WarpTaps t;
  const float xx = i + fx;
  const float yy = j + fy;
  const int x = (int)floorf(xx), y = (int)floorf(yy);
  t.dx = xx - (float)x;
  t.dy = yy - (float)y;
  t.m = (xx >= 0 && xx <= (float)(w - 1) && yy >= 0 && yy <= (float)(h - 1)) ? 1.0f : 0.0f;
  const int x1 = clampi(x, 0, w - 1) + pad, x2 = clampi(x + 1, 0, w - 1) + pad;
  const int y1 = (clampi(y, 0, h - 1) + pad) * tmp_w, y2 = (clampi(y + 1, 0, h - 1) + pad) * tmp_w;
  t.o11 = y1 + x1; t.o12 = y1 + x2; t.o21 = y2 + x1; t.o22 = y2 + x2;
  return t;
}
__device__ __forceinline__ float warp_mix(float s11, float s12, float s21, float s22, float dx, float dy) {
  return s11 * (1.0f - dx) * (1.0f - dy) + s12 * dx * (1.0f - dy) + s21 * (1.0f - dx) * dy + s22 * dx * dy;
}

// vertical 5-tap centred on an image row with the folded coefficients of the first / last two rows (image.c:401-434).
// FORM: 0 = row 0, 1 = row 1, 2 = interior, 3 = row h-2, 4 = row h-1 (h >= 4: the four border rows are distinct); the row
// is wave-uniform, so a stage picks the form once by scalar branches (vform) for all its filters
template <int FORM>
__device__ __forceinline__ float v5r(float m2, float m1, float s0, float p1, float p2) {
  if (FORM == 0) return (D5_C0 + D5_C1 + D5_C2) * s0 + D5_C3 * p1 + D5_C4 * p2;
  if (FORM == 1) return (D5_C0 + D5_C1) * m1 + D5_C2 * s0 + D5_C3 * p1 + D5_C4 * p2;
  if (FORM == 3) return D5_C0 * m2 + D5_C1 * m1 + D5_C2 * s0 + (D5_C3 + D5_C4) * p1;
  if (FORM == 4) return D5_C0 * m2 + D5_C1 * m1 + (D5_C2 + D5_C3 + D5_C4) * s0;
  return D5_C0 * m2 + D5_C1 * m1 + D5_C2 * s0 + D5_C3 * p1 + D5_C4 * p2;
}
template <typename F>
__device__ __forceinline__ void vform(int j, int h, F&& f) {
  if (j == 0) f(std::integral_constant<int, 0>());
  else if (j == 1) f(std::integral_constant<int, 1>());
  else if (j == h - 2) f(std::integral_constant<int, 3>());
  else if (j == h - 1) f(std::integral_constant<int, 4>());
  else f(std::integral_constant<int, 2>());
}
__device__ __forceinline__ float h5r(float m2, float m1, float s0, float p1, float p2) {
  return D5_C0 * m2 + D5_C1 * m1 + D5_C2 * s0 + D5_C3 * p1 + D5_C4 * p2;
}

// (Round 4: the flushes with everything block-independent precomputed once per wavefront -- which pieces a lane moves, their
// byte offsets, the per-frame strip bases: ~9 instead of ~30 instructions per piece, no scalar divisions per flush -- measured
// 1.83-2.03 against 1.86 ms on level 3 of the headline, 105 instead of 72 VGPRs: no gain, not kept.  Like the test-free
// interior loop body of round 3, it removes instructions the kernel was not waiting for.)
constexpr int PREP_KD = 4;  // rows of derivative records staged before a flush (runs of KD * 32 bytes)
constexpr int PREP_KW = 8;  // rows of (wx, wy) records staged before a flush (runs of KW * 8 bytes)
// (measured at 4096 pairs, ms per step of this kernel, round 3: KD = 1 / 2 / 4: 1.17 / 0.79 / 0.76, KW = 4 / 8: 0.79 / 0.70 --
// the run length of the stores is what it is most sensitive to; profiles/README.md r03_b.  Round 5, 16384 pairs, same box:
// KD = 2 / 4 / 8: 2.46-2.47 / 2.22-2.29 / 3.05 ms, KD = 4 with KW = 16: 2.63 -- a run of KD = 4 records is exactly one
// 128-byte cache line (h is a multiple of 4 on the benchmarked levels 3 and 4), and the larger staging area costs nothing: the
// kernel does not care whether 2.5 or 4 wavefronts share a SIMD (profiles/r05_b_variants.txt), which is what had kept KD at 2)

// LDS per block, in floats.  WPF = wavefronts per frame (= per block): 1 (w <= 64; up to four frames in the wavefront), 2
// (w <= 128), 3 or 4 (round 6: w <= 192 / 256 -- the finest level of a 1242 x 375 KITTI pair at operating point 2 is 156 x 48).
template <int WPF>
struct PrepLds {
  static constexpr int ROW = 64 * WPF + 32;  // one exchanged row: the block's columns + 4 pads + 4 dummies per frame segment (<= 4)
  // staging: per frame (lpf + K - 1) diag rows x K records; the worst case over 1 / 2 / 4 frames per wavefront
  static constexpr int nq(int lpf, int K) { return lpf + K - 1; }
  // (+ one dummy record at the end of each array: where the lanes beyond the image write)
  static constexpr int STD = (WPF >= 2 ? nq(64 * WPF, PREP_KD) * PREP_KD : 4 * nq(16, PREP_KD) * PREP_KD) * 8 + 8;
  static constexpr int STW = (WPF >= 2 ? nq(64 * WPF, PREP_KW) * PREP_KW : 4 * nq(16, PREP_KW) * PREP_KW) * 2 + 2;
  static constexpr int TOTAL = 3 * ROW + STD + STW;
};

// LDS hand-over between the lanes of a frame's row: within one wavefront the LDS pipeline keeps program order, so a
// scheduling barrier is enough; two wavefronts need a workgroup barrier -- one that only drains the LDS counter
// (__syncthreads() would also wait for vmcnt(0), i.e. for the row and tap loads that are deliberately kept in flight).
template <int WPF>
__device__ __forceinline__ void prep_sync() {
  if (WPF == 1) __builtin_amdgcn_wave_barrier();
  else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// DENS (round 6): the flow a row is warped with is not read from the densified AoS flow but DENSIFIED HERE from the patch
// results (PatGridClass::AggregateFlowDense, patchgrid.cpp:213-275, for gray 8x8 patches on a step-4 grid: at most 2 x 2
// patches cover a pixel; the arithmetic and candidate order of densify_quad_kernel, ofdis_dis.hip -- same bits): the
// densification kernel, its launch and the round trip of the dense flow through HBM (8 bytes per pixel out and in again) go
// away.  With lane = column and the grid-row-major weight layout (ofdis_dev.h: pweight_row) a wavefront's weight loads of
// one image row are two contiguous runs per covering grid row, and every weight is read by exactly one lane.
template <int WPF, bool DENS>
__global__ __launch_bounds__(64 * WPF) void tv_prep_kernel(const PrepArgs a, const int lpf_shift, const int nbands) {
  constexpr int C = 1;  // columns per lane
  using L = PrepLds<WPF>;
  __shared__ __attribute__((aligned(16))) float lds[L::TOTAL];
  const int lane = threadIdx.x;       // lane of the block
  const int w = a.t.w, h = a.t.h, S = a.S;
  const int lpf = WPF >= 2 ? 64 * WPF : (1 << lpf_shift);  // lanes per frame: the whole block (WPF >= 2), else 64 / 32 / 16
  const int fpw = WPF >= 2 ? 1 : (64 >> lpf_shift);        // frames per block
  const int unit = blockIdx.x;
  const int fg = unit / nbands, band = unit - fg * nbands;
  const int fl = WPF >= 2 ? 0 : lane >> lpf_shift, li = WPF >= 2 ? lane : lane & (lpf - 1);
  int frame = fg * fpw + fl;
  const bool fok = frame < a.t.nframes;
  if (!fok) frame = a.t.nframes - 1;  // idle lane group: shadows the last frame, never flushed
  const int yb0 = band * a.band_rows, yb1 = min(h, yb0 + a.band_rows);  // (band_rows: even; a multiple of KW from 8 rows on)
  if (yb0 >= h) return;
  const int r_begin = max(yb0 - 4, 0), r_s0end = min(yb1 + 4, h);  // rows of stage 0
  const int y1lo = max(yb0 - 2, 0), y1hi = min(yb1 + 2, h);       // rows of stage 1
  const int r_last = yb1 + 3;

  const int seg = lpf * C + 8;  // floats of one frame's segment of an exchanged row: 2 pads, the columns, 2 pads, 4 dummies
  float* const rowA = lds + fl * seg;
  float* const rowX = lds + L::ROW + fl * seg;
  float* const rowZ = lds + 2 * L::ROW + fl * seg;
  const int nqd = L::nq(lpf, PREP_KD), nqw = L::nq(lpf, PREP_KW);  // diag rows a frame's staged block touches
  float* const stD = lds + 3 * L::ROW;       // [frame of the wavefront][q][k][8]
  float* const stW = stD + L::STD;           // [frame of the wavefront][q][k][2]
  float* const stDf = stD + fl * nqd * PREP_KD * 8;
  float* const stWf = stW + fl * nqw * PREP_KW * 2;
  const int e0 = 2 + li * C;  // LDS entry of this lane's first column

  int xc[C];
  const int rw = S * w;
#pragma unroll
  for (int k = 0; k < C; ++k) xc[k] = min(li * C + k, w - 1);  // columns beyond the image replicate the last one
  const int sg0 = (fg * fpw) / S;  // first strip this wavefront touches: the buffer resources are based there
  const size_t strip_recs = (size_t)rw * h;
  const int nstrips = (a.t.nframes + S - 1) / S;
  const unsigned span = (unsigned)min(nstrips - sg0, fpw + 1);  // strips reachable from sg0
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.d8 + (size_t)sg0 * strip_recs * 8), 0, (int)(span * strip_recs * 32), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.wrec + (size_t)sg0 * strip_recs * 2), 0, (int)(span * strip_recs * 8), 0x00020000);
  // inputs through buffer resources based at the wavefront's first frame: per-lane offsets are constants (frame, column),
  // the row is a scalar offset
  const int f0 = fg * fpw, nfr = min(fpw, a.t.nframes - f0), flc = frame - f0;  // (flc: idle lane groups shadow a frame)
  const int plane = a.tmp_w * a.tmp_h;
  const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.flow + (size_t)f0 * w * h * 2), 0, nfr * w * h * 8, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.im1 + (size_t)f0 * plane), 0, nfr * plane * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.im2 + (size_t)f0 * plane), 0, nfr * plane * 4, 0x00020000);
  int voF[C], voI[C];
  const int voT = flc * plane * 4;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    voF[k] = (flc * w * h + xc[k]) * 8;
    voI[k] = (flc * plane + a.pad * a.tmp_w + a.pad + xc[k]) * 4;
  }
  auto asf = [](unsigned u) { return __builtin_bit_cast(float, u); };
  // DENS: the (at most) 2 x 2 patches that cover pixel (x, y): grid columns bx - 1 and bx, where x is patch column 4 + tx and
  // tx; grid rows by - 1 and by, where y is patch row 4 + ty and ty (densify_quad_kernel).  The column part is a per-lane
  // constant, the row part wave-uniform: weights and displacements come through buffer resources based at the wavefront's
  // first frame with a per-lane byte offset per grid column and a scalar offset per grid row.
  const int nopw = a.dens_nopw, noph = a.dens_noph, nop = nopw * noph;
  __amdgpu_buffer_rsrc_t rsPW = rsF, rsP = rsF;
  int voPW[2] = {0, 0}, voP[2] = {0, 0};
  bool okx[2] = {false, false};
  if constexpr (DENS) {
    static_assert(C == 1, "one column per lane");
    rsPW = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dens_pweight + (size_t)f0 * nop * 64), 0, nfr * nop * 256, 0x00020000);
    rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dens_p + (size_t)f0 * nop * 2), 0, nfr * nop * 8, 0x00020000);
    const int xs = xc[0] - a.dens_offw + 4, bx = xs >> 2, tx = xs & 3;  // xs >= 1: offw < 4
#pragma unroll
    for (int col = 0; col < 2; ++col) {
      const int gx = bx - 1 + col;
      okx[col] = (gx >= 0) & (gx < nopw);
      const int gxc = clampi(gx, 0, nopw - 1);
      voPW[col] = (flc * nop * 64 + gxc * 8 + (col ? tx : 4 + tx)) * 4;
      voP[col] = (flc * nop * 2 + gxc * 2) * 4;
    }
  }

  // Flush of a staged block (rows y0 .. y0+K-1 of every frame of this wavefront): the block's records sit in LDS as
  // [q][k] with q = x + k (mod w when S = 1: the diag rows of a frame wrap onto themselves), i.e. in diag order, so piece e
  // of the staging array goes to byte (e mod K*PPR) * PB of the run of diag row dbase + e / (K*PPR): consecutive lanes =
  // consecutive bytes.
  auto flush = [&](auto kc, auto bytes_c, const float* st, const __amdgpu_buffer_rsrc_t& rs, int nq, int y0, int yend) {
    constexpr int K = decltype(kc)::value, RB = decltype(bytes_c)::value;  // rows per block, record bytes
    constexpr int PB = RB >= 16 ? 16 : 8;                                 // piece bytes per lane
    constexpr int PPR = RB / PB;                                          // pieces per record
    constexpr int RUN = K * PPR;                                          // pieces per diag row (a power of two)
    const bool full = y0 + K <= yend;                                     // (uniform) every row of the block exists
    const int nqv = S == 1 ? w : w + K - 1;  // diag rows the block touches (S = 1: they wrap onto the frame's own w rows)
    const int iters = (nqv * RUN + 64 * WPF - 1) / (64 * WPF);  // (uniform trip count: no divergent loop)
    for (int f = 0; f < fpw; ++f) {
      const int fr = f0 + f;
      if (fr >= a.t.nframes) break;
      const int sgf = fr / S, fsf = fr - sgf * S;
      const int dbase = (fsf * w + y0) % rw;
      const int base = ((sgf - sg0) * rw * h + y0) * RB;
      const float* stf = st + f * nq * K * (RB / 4);
      for (int it = 0; it < iters; ++it) {
        const int e = min(lane + it * 64 * WPF, nqv * RUN);  // (pieces beyond the block: a valid LDS address, dropped below)
        const int q = e / RUN, lo = e % RUN;
        int d = dbase + q;
        d -= d >= rw ? rw : 0;
        // (bitwise operators: no short-circuit branches.)  A piece is stored if its diag row, its pixel and its row exist
        const int k = lo / PPR, x = q - k;
        const bool ok = (q < nqv) & ((S == 1) | ((x >= 0) & (x < w))) & (full | (y0 + k < yend));
        // a store that must not happen gets an offset beyond the resource's range: the hardware drops it (no branch)
        const int byte = ok ? base + d * (h * RB) + lo * PB : 0x7ffffff0;
        if constexpr (PB == 16) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(stf + e * 4);
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte, 0, 0);
        } else {
          const u32x2 v = *reinterpret_cast<const u32x2*>(stf + e * 2);
          __builtin_amdgcn_raw_buffer_store_b64(v, rs, byte, 0, 0);
        }
      }
    }
  };
  // One image row through LDS for the horizontal filters: entry 2 + c = column c, two entries replicating column 0 before and
  // two replicating column w - 1 after (image.c:466-502).  Branch-free: every lane writes its column (lanes beyond the image
  // into a dummy entry) and one pair of pads (the first lane the left pair, the lane of column w - 1 the right pair, the
  // others a dummy pair).
  const int col_entry = li < w ? 2 + li : lpf + 4;
  const int pad_entry = li == 0 ? 0 : (li == w - 1 ? 2 + w : lpf + 6);
  auto put_row = [&](float* row, const float (&v)[C]) {
    static_assert(C == 1, "one column per lane");
    row[col_entry] = v[0];
    row[pad_entry] = v[0];
    row[pad_entry + 1] = v[0];
  };
  // staging position of column x in block row k
  auto stq = [&](int x, int k) { int q = x + k; if (S == 1 && q >= w) q -= w; return q; };

  float A[5][C], Z[5][C], IX[5][C], IY[5][C];
  float ixy1[C], ixy2[C], iyz1[C], iyz2[C];  // Ixy, Iyz of rows r-3 and r-4 at stage 2 (computed at stage 1, two iterations before)
  unsigned mbits[C];                          // bit j = warp mask of row r-j
#pragma unroll
  for (int k = 0; k < C; ++k) {
#pragma unroll
    for (int q = 0; q < 5; ++q) A[q][k] = Z[q][k] = IX[q][k] = IY[q][k] = 0.0f;
    ixy1[k] = ixy2[k] = iyz1[k] = iyz2[k] = 0.0f;
    mbits[k] = 0;
  }

  // Software pipeline of stage 0's memory accesses: at iteration r the flow and first-image values of row r+2 are requested,
  // the taps of row r+1 (whose flow arrived during the previous iteration) are requested, and row r is combined from the
  // taps requested one iteration ago -- every load has a whole iteration of filter arithmetic to arrive.
  // Rows beyond the last stage-0 row are requested at the clamped row index and never used.
  struct Pend {  // a row whose flow is known
    float fx, fy, i1;
  };
  Pend nxt[C], cur[C];     // row r+1 (flow arrived, taps requested), row r (taps arriving)
  WarpTaps tp[C];          // fractions / mask of row r
  float t11[C], t12[C], t21[C], t22[C];  // taps of row r
  unsigned lf0[C], lf1[C], li1[C];       // raw flow / image values of the row requested last (row r+2 after the request)
  unsigned dpw[4], dp0[4], dp1[4];       // DENS: weight and displacement of candidate c = 2 * (grid column) + (grid row)
  auto request_row = [&](int row) {      // flow + first image of `row`
    const int rr = min(row, h - 1);
    if constexpr (DENS) {
      const int ys = rr - a.dens_offh + 4, by = ys >> 2, ty = ys & 3;  // ys >= 1: offh < 4
      // (the displacements change only with the grid row, every fourth image row -- but loading them under a wave-uniform
      // branch was measured 0.8 ms SLOWER per 16384 pairs, 3.55 against 2.75 ms: a conditional load leaves the compiler without
      // an exact count of the loads in flight, and it waits for all of them; profiles/r06_variants.txt)
#pragma unroll
      for (int rsel = 0; rsel < 2; ++rsel) {
        const int gyc = clampi(by - 1 + rsel, 0, noph - 1), ky = rsel ? ty : 4 + ty;
        const int soPW = (gyc * 8 + ky) * nopw * 32, soP = gyc * nopw * 8;
#pragma unroll
        for (int col = 0; col < 2; ++col) {
          dpw[2 * col + rsel] = __builtin_amdgcn_raw_buffer_load_b32(rsPW, voPW[col], soPW, 0);
          const auto pp = __builtin_amdgcn_raw_buffer_load_b64(rsP, voP[col], soP, 0);
          const unsigned q0 = pp[0], q1 = pp[1];
          dp0[2 * col + rsel] = q0; dp1[2 * col + rsel] = q1;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < C; ++k) {
      if constexpr (!DENS) {
        const auto f = __builtin_amdgcn_raw_buffer_load_b64(rsF, voF[k], rr * w * 8, 0);
        const unsigned q0 = f[0], q1 = f[1];
        lf0[k] = q0; lf1[k] = q1;
      }
      li1[k] = __builtin_amdgcn_raw_buffer_load_b32(rsI, voI[k], rr * a.tmp_w * 4, 0);
    }
  };
  // the row requested last has arrived: its flow (DENS: the densification of pixel (x, row): candidates in the reference's
  // order -- grid column ascending, then grid row --, the additions and quotients of densify_quad_kernel) and image value
  auto take_row = [&](int row) {
#pragma unroll
    for (int k = 0; k < C; ++k) {
      if constexpr (DENS) {
        const int ys = min(row, h - 1) - a.dens_offh + 4, by = ys >> 2;
        const bool okr[2] = {by >= 1 && by - 1 < noph, by < noph};
        float we = 0.0f, fu = 0.0f, fv = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bool ok = okx[c >> 1] & okr[c & 1];
          const float absw = div_rn(1.0f, fmaxf(2.0f, asf(dpw[c])));  // == 1.0f / x: numerator 1, denominator >= 2 (ofdis_dev.h)
          const float nwe = we + absw, nfu = fu + asf(dp0[c]) * absw, nfv = fv + asf(dp1[c]) * absw;
          we = ok ? nwe : we;
          fu = ok ? nfu : fu;
          fv = ok ? nfv : fv;
        }
        if (we > 0) {
          if constexpr (kFusedContract) {  // one hardware reciprocal shared by the two quotients (ofdis_dev.h)
            const float rwe = rcp_refined(we);
            fu *= rwe;
            fv *= rwe;
          } else {
            fu /= we;
            fv /= we;
          }
        }
        nxt[k] = Pend{fu, fv, asf(li1[k])};
      } else {
        nxt[k] = Pend{asf(lf0[k]), asf(lf1[k]), asf(li1[k])};
      }
    }
  };
  auto request_taps = [&](int row) {  // from nxt (row `row`): tap addresses, fractions, mask; the four loads
    const int rr = min(row, h - 1);
    WarpTaps t[C];
#pragma unroll
    for (int k = 0; k < C; ++k) t[k] = warp_taps(a.tmp_w, a.pad, w, h, xc[k], rr, nxt[k].fx, nxt[k].fy);
#pragma unroll
    for (int k = 0; k < C; ++k) {
      t11[k] = asf(__builtin_amdgcn_raw_buffer_load_b32(rsT, voT + t[k].o11 * 4, 0, 0));
      t12[k] = asf(__builtin_amdgcn_raw_buffer_load_b32(rsT, voT + t[k].o12 * 4, 0, 0));
      t21[k] = asf(__builtin_amdgcn_raw_buffer_load_b32(rsT, voT + t[k].o21 * 4, 0, 0));
      t22[k] = asf(__builtin_amdgcn_raw_buffer_load_b32(rsT, voT + t[k].o22 * 4, 0, 0));
      tp[k] = t[k];
    }
  };
  // prologue: row r_begin's taps and row r_begin+1's flow in flight
  request_row(r_begin);
  take_row(r_begin);
  request_taps(r_begin);
#pragma unroll
  for (int k = 0; k < C; ++k) cur[k] = nxt[k];
  request_row(r_begin + 1);

  for (int rb = r_begin; rb <= r_last; rb += 5) {
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int r = rb + u;  // wave-uniform
      if (r > r_last) break;
      // ------------------------------------------------------------------ stage 0: row r
      if (r < r_s0end) {
        float fx[C], fy[C];
#pragma unroll
        for (int k = 0; k < C; ++k) {  // row r from the taps requested one iteration ago
          fx[k] = cur[k].fx; fy[k] = cur[k].fy;
          const float i2 = warp_mix(t11[k], t12[k], t21[k], t22[k], tp[k].dx, tp[k].dy);
          mbits[k] = (mbits[k] << 1) | (tp[k].m != 0.0f ? 1u : 0u);
          A[u][k] = 0.5f * (i2 + cur[k].i1);
          Z[u][k] = i2 - cur[k].i1;
        }
        // the flow of row r+1 has arrived: its taps; then the flow of row r+2
        take_row(r + 1);
        request_taps(r + 1);
#pragma unroll
        for (int k = 0; k < C; ++k) cur[k] = nxt[k];
        request_row(r + 2);
        // the row through LDS: entries 2 .. 2 + lpf*C - 1 = columns, two replicated pads either side
        put_row(rowA, A[u]);
        if (r >= yb0 && r < yb1) {  // (wx, wy) of this row into its staging block
          const int kk = (r - yb0) % PREP_KW;
          float* dst = li < w ? stWf + (stq(xc[0], kk) * PREP_KW + kk) * 2 : stW + (L::STW - 2);  // (beyond the image: dummy)
          *reinterpret_cast<float2*>(dst) = make_float2(fx[0], fy[0]);
        }
        prep_sync<WPF>();
        {
          const float al0 = rowA[e0 - 2], al1 = rowA[e0 - 1], ar0 = rowA[e0 + C], ar1 = rowA[e0 + C + 1];
          if constexpr (C == 1) {
            IX[u][0] = h5r(al0, al1, A[u][0], ar0, ar1);
          } else {
            IX[u][0] = h5r(al0, al1, A[u][0], A[u][1], ar0);
            IX[u][1] = h5r(al1, A[u][0], A[u][1], ar0, ar1);
            // a second column beyond the image replicates column w - 1 like avg and Iz do (they are computed at clamped
            // coordinates): the next filter reads it from this register
            if (li * 2 + 1 >= w) IX[u][1] = IX[u][0];
          }
        }
        if (r >= yb0 && r < yb1 && ((r - yb0) % PREP_KW == PREP_KW - 1 || r == yb1 - 1)) {
          const int y0 = r - (r - yb0) % PREP_KW;  // (the block's last write was before the barrier above)
          flush(std::integral_constant<int, PREP_KW>(), std::integral_constant<int, 8>(), stW, rsW, nqw, y0, yb1);
        }
        // (the next writes of rowA and of the (wx, wy) staging come after a barrier of stage 2 or the one at the end)
      } else {
#pragma unroll
        for (int k = 0; k < C; ++k) mbits[k] <<= 1;
      }
      // ------------------------------------------------------------------ stage 1: row y1 = r - 2 (windows: rows r-4 .. r)
      const int y1 = r - 2;
      float ixyn[C], iyzn[C];  // Ixy, Iyz of row r-2: they enter the delay registers at the end of the iteration
#pragma unroll
      for (int k = 0; k < C; ++k) ixyn[k] = iyzn[k] = 0.0f;
      if (y1 >= y1lo && y1 < y1hi) {
        vform(y1, h, [&](auto form) {
          constexpr int F = decltype(form)::value;
#pragma unroll
          for (int k = 0; k < C; ++k) {
            IY[(u + 3) % 5][k] = v5r<F>(A[(u + 1) % 5][k], A[(u + 2) % 5][k], A[(u + 3) % 5][k], A[(u + 4) % 5][k], A[u][k]);
            ixyn[k] = v5r<F>(IX[(u + 1) % 5][k], IX[(u + 2) % 5][k], IX[(u + 3) % 5][k], IX[(u + 4) % 5][k], IX[u][k]);
            iyzn[k] = v5r<F>(Z[(u + 1) % 5][k], Z[(u + 2) % 5][k], Z[(u + 3) % 5][k], Z[(u + 4) % 5][k], Z[u][k]);
          }
        });
      }
      // ------------------------------------------------------------------ stage 2: row y2 = r - 4: the record is complete
      const int y2 = r - 4;
      if (y2 >= yb0 && y2 < yb1) {
        const int o = (u + 1) % 5;  // ring slot of row r-4 (a constant after unrolling)
        // horizontal filters of Ix and Iz of row y2 (the rows are still in the windows)
        put_row(rowX, IX[o]);
        put_row(rowZ, Z[o]);
        prep_sync<WPF>();
        float ixx[C], ixz[C], iyy[C];
        {
          const float xl0 = rowX[e0 - 2], xl1 = rowX[e0 - 1], xr0 = rowX[e0 + C], xr1 = rowX[e0 + C + 1];
          const float zl0 = rowZ[e0 - 2], zl1 = rowZ[e0 - 1], zr0 = rowZ[e0 + C], zr1 = rowZ[e0 + C + 1];
          if constexpr (C == 1) {
            ixx[0] = h5r(xl0, xl1, IX[o][0], xr0, xr1);
            ixz[0] = h5r(zl0, zl1, Z[o][0], zr0, zr1);
          } else {
            ixx[0] = h5r(xl0, xl1, IX[o][0], IX[o][1], xr0);
            ixx[1] = h5r(xl1, IX[o][0], IX[o][1], xr0, xr1);
            ixz[0] = h5r(zl0, zl1, Z[o][0], Z[o][1], zr0);
            ixz[1] = h5r(zl1, Z[o][0], Z[o][1], zr0, zr1);
          }
        }
        vform(y2, h, [&](auto form) {
          constexpr int F = decltype(form)::value;
#pragma unroll
          for (int k = 0; k < C; ++k)
            iyy[k] = v5r<F>(IY[(u + 4) % 5][k], IY[u][k], IY[(u + 1) % 5][k], IY[(u + 2) % 5][k], IY[(u + 3) % 5][k]);
        });
        const int kk = (y2 - yb0) % PREP_KD;
        {
          constexpr int k = 0;
          const bool on = (mbits[k] >> 4) & 1u;  // the warp mask of row r-4
          float* dst = li < w ? stDf + (stq(xc[k], kk) * PREP_KD + kk) * 8 : stD + (L::STD - 8);  // (beyond the image: dummy)
          *reinterpret_cast<float4*>(dst) =
              on ? make_float4(IX[o][k], Z[o][k], ixx[k], ixz[k]) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          *reinterpret_cast<float4*>(dst + 4) =
              on ? make_float4(IY[o][k], ixy2[k], iyz2[k], iyy[k]) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        prep_sync<WPF>();
        if (kk == PREP_KD - 1 || y2 == yb1 - 1) {
          flush(std::integral_constant<int, PREP_KD>(), std::integral_constant<int, 32>(), stD, rsD, nqd, y2 - kk, yb1);
          prep_sync<WPF>();
        }
      }
      else {
        prep_sync<WPF>();  // (no stage 2 in this iteration: stage 0's LDS reads before the next iteration's writes)
      }
#pragma unroll
      for (int k = 0; k < C; ++k) { ixy2[k] = ixy1[k]; iyz2[k] = iyz1[k]; ixy1[k] = ixyn[k]; iyz1[k] = iyzn[k]; }
    }
  }
}

// (rows: the kernel marches down any number of them; 256 is what the consumers of its records handle -- one wavefront per strip
// up to 64 rows, two to four up to 256: ofdis_fused.hip, ofdis_fused_tall.hip)
// (columns: one lane each, up to four wavefronts side by side)
bool tv_prep_supported(const TvGeom& t) { return t.noc == 1 && t.w >= 16 && t.w <= 256 && t.h >= 4 && t.h <= 256; }
// Can launch_tv_prep densify the flow itself for this patch grid (PrepArgs::dens_*)?  The geometry of densify_quad_kernel:
// gray 8x8 patches on a step-4 grid (operating point 2), at most 2 x 2 patches per pixel.
bool tv_prep_densifies(const LevelGeom& g) { return g.noc == 1 && g.P == 8 && g.steps == 4 && g.offw < 4 && g.offh < 4; }

hipError_t launch_tv_prep(const PrepArgs& a_in, hipStream_t s) {
  if (!tv_prep_supported(a_in.t) || a_in.S < 1) return hipErrorInvalidValue;
  PrepArgs a = a_in;
  const int w = a.t.w, h = a.t.h;
  const int lpf_shift = w > 64 ? 7 : (w > 32 ? 6 : (w > 16 ? 5 : 4));  // (w > 64: not used, the block is the frame)
  const int fpw = w > 64 ? 1 : (64 >> lpf_shift);
  const int fgroups = (a.t.nframes + fpw - 1) / fpw;
  // bands: enough wavefronts for ~2 per SIMD (1024 SIMDs); from 8 output rows on a multiple of the staging blocks
  int nbands;
  if (a.band_rows > 0) {  // explicit (ofdis_tuning::prep_band_rows): any even number of rows
    a.band_rows = std::max(2, (a.band_rows + 1) & ~1);
  } else {  // (small batches: bands down to two rows -- a band re-computes 4 + 2 margin rows, but the kernel is then bound by
    // the latency of one wavefront marching its rows: 55 -> 36 us per one-pair pass with 2 instead of 8 rows per band)
    nbands = std::max(1, std::min((2048 + fgroups - 1) / fgroups, (h + 1) / 2));
    a.band_rows = ((h + nbands - 1) / nbands + 1) & ~1;
    if (a.band_rows > PREP_KW) a.band_rows = ((a.band_rows + PREP_KW - 1) / PREP_KW) * PREP_KW;
  }
  nbands = (h + a.band_rows - 1) / a.band_rows;
  const long long units = (long long)fgroups * nbands;
  if (a.dens_p) {  // densification inside this kernel (tv_prep_densifies): patch results in, no dense flow read
    if (!a.dens_pweight || a.dens_offw < 0 || a.dens_offw > 3 || a.dens_offh < 0 || a.dens_offh > 3 || a.dens_nopw < 1 || a.dens_noph < 1)
      return hipErrorInvalidValue;
    if (w > 192) hipLaunchKernelGGL((tv_prep_kernel<4, true>), dim3((unsigned)units), dim3(256), 0, s, a, lpf_shift, nbands);
    else if (w > 128) hipLaunchKernelGGL((tv_prep_kernel<3, true>), dim3((unsigned)units), dim3(192), 0, s, a, lpf_shift, nbands);
    else if (w > 64) hipLaunchKernelGGL((tv_prep_kernel<2, true>), dim3((unsigned)units), dim3(128), 0, s, a, lpf_shift, nbands);
    else hipLaunchKernelGGL((tv_prep_kernel<1, true>), dim3((unsigned)units), dim3(64), 0, s, a, lpf_shift, nbands);
    return hipGetLastError();
  }
  if (w > 192) hipLaunchKernelGGL((tv_prep_kernel<4, false>), dim3((unsigned)units), dim3(256), 0, s, a, lpf_shift, nbands);
  else if (w > 128) hipLaunchKernelGGL((tv_prep_kernel<3, false>), dim3((unsigned)units), dim3(192), 0, s, a, lpf_shift, nbands);
  else if (w > 64) hipLaunchKernelGGL((tv_prep_kernel<2, false>), dim3((unsigned)units), dim3(128), 0, s, a, lpf_shift, nbands);
  else hipLaunchKernelGGL((tv_prep_kernel<1, false>), dim3((unsigned)units), dim3(64), 0, s, a, lpf_shift, nbands);
  return hipGetLastError();
}

}  // namespace OFDIS_KNS
}  // namespace ofdis
