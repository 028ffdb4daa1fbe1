// ofdis_kernels.h -- argument blocks and launchers of the gfx950 kernels (internal C++ API; the public ABI is
// include/ofdis.h).  The argument blocks are shared; the launchers exist once per arithmetic contract (ofdis_dev.h):
// ofdis_launchers.inc is included into ofdis::exact and ofdis::fused, every kernel file defines its launchers in the
// namespace of the contract it is being compiled for (OFDIS_KNS), and ofdis_capi.hip picks a set per context.
#pragma once
#include "ofdis_dev.h"

namespace ofdis {

// flow-plane layouts used between kernels
//   AoS   : [frame][h][w][2]   -- the reference's dense-flow format (DIS init / output)
//   planar: [frame][h][w] x2   -- TV stage
struct DisArgs {
  LevelGeom g;
  int nframes;
  // solver parameters (oflow.cpp:76-108)
  int max_iter, min_iter, costfct, patnorm;
  int stereo, camlr;  // SELECTMODE=2: one horizontal displacement per patch; camlr 0 -> p <= 0, 1 -> p >= 0 (patch.cpp:188-193)
  float dp_thresh_sq, dr_thresh, res_thresh;
  float outlier_sq_max;  // largest x with sqrtf(x) <= outlierthresh (= P/2, oflow.cpp:82): outlier_sq_threshold()
  const float* im_a;     // [B][tmp_h][tmp_w][noc]
  const float* im_a_dx;
  const float* im_a_dy;
  const float* im_b;
  const float* flow_prev;  // AoS [B][h/2][w/2][2] or nullptr
  float* p_out;            // [B][nop][2]       both in the internal grid-row-major layout (ofdis_dev.h: patch_slot,
  float* pweight;          // [B][nop][novals]  pweight_row): the densify kernels read them, nobody else
  float* pixw;             // [B][nop][P*P] or nullptr: RGB 12x12 only (patch_pixel_weights_supported): patches whose weights
                           // the densification reads unshifted store one float per pixel here INSTEAD of pweight (ofdis_dev.h)
};
// snapshot of the kernel-selection knobs (include/ofdis.h: ofdis_tuning; ofdis_capi.hip); *epoch counts the changes
ofdis_tuning tuning(unsigned* epoch = nullptr);

struct DensifyArgs {
  LevelGeom g;
  int nframes;
  const float* p;        // [B][nop][2]       (internal layout, as the patch kernels write them)
  const float* pweight;  // [B][nop][novals]
  const float* pixw;     // [B][nop][P*P] or nullptr: what the patch kernel was given as DisArgs::pixw
  float* flow_aos;       // if non-null: AoS output
  float* wx;             // else planar outputs (row-major)
  float* wy;
  // forward-backward merging (usefbcon, patchgrid.cpp:277-375): the complementary grid's results, or null
  int stereo;               // one flow channel: flow_aos is [B][h][w], wy is not written
  unsigned idx_magic;       // set by launch_densify: ceil(2^32 / w) for the pixel-index split
  const float* cg_p;        // [B][nop][2]
  const float* cg_pweight;  // [B][nop][novals]
};

struct TvGeom {
  int w, h, noc, nframes;
};

// image_warp.  src: either the padded interleaved pyramid plane (src_padded=1: [B][tmp_h][tmp_w][noc],
// pad/tmp_w given) or packed planar [B][noc][h][w] (src_padded=0).  dst: [B][noc][h][w], mask [B][h][w].
struct WarpArgs {
  TvGeom t;
  const float* src;
  int src_padded, pad, tmp_w, tmp_h;
  const float* wx;
  const float* wy;
  float* dst;
  float* mask;
};
// get_derivatives.  im1 as for WarpArgs.src; im2w packed planar [B][noc][h][w].
// out [B][8][noc][h][w]
struct DerivArgs {
  TvGeom t;
  const float* im1;
  int im1_padded, pad, tmp_w, tmp_h;
  const float* im2w;
  float* out;  // [B][8][noc][h][w] row-major
  // record form (the RGB fused TV path; ofdis_tv.hip): with rec_d8 the planes are not written; [noc][B][w*h][8] derivative
  // records and [B][w*h][2] (wx, wy) records in the diag layout instead, zeroed by the warp's mask
  float* rec_d8 = nullptr;
  float* rec_w = nullptr;
  const float* mask = nullptr;  // row-major [B][h][w]
  const float* wx = nullptr;    // row-major
  const float* wy = nullptr;
};

// image_warp + get_derivatives of the fused TV path in one row-marching kernel (ofdis_prep.hip): densified AoS flow and the
// two padded gray planes in, the sdiag records of the fused TV kernel out (ofdis_dev.h: sdiag_index)
struct PrepArgs {
  TvGeom t;           // noc = 1
  const float* im1;   // padded plane of the first image  [B][tmp_h][tmp_w]
  const float* im2;   // padded plane of the second image
  int pad, tmp_w, tmp_h;
  const float* flow;  // AoS [B][h][w][2], row-major: the flow to warp with (wx, wy)
  float* d8;          // [records][8] = Ix, Iz, Ixx, Ixz, Iy, Ixy, Iyz, Iyy; all zero where the warp's mask is zero
  float* wrec;        // [records][2] = wx, wy
  int S;              // frames per strip
  int band_rows;      // output rows per wavefront (0 = chosen by the launcher from the batch size)
  // round 6, optional: densification inside this kernel (tv_prep_densifies).  With dens_p the dense flow is computed from the
  // patch results -- [B][nop][2] displacements and [B][nop][64] weights in the internal grid-row-major layout (ofdis_dev.h:
  // patch_slot, pweight_row), exactly what launch_densify would read -- and `flow` is not read.
  const float* dens_p = nullptr;
  const float* dens_pweight = nullptr;
  int dens_nopw = 0, dens_noph = 0, dens_offw = 0, dens_offh = 0;
};

// compute_smoothness + compute_data + sub_laplacian x2 -> sys [B][7][w*h] in DIAG layout (ofdis_dev.h)
struct SystemArgs {
  TvGeom t;
  const float* mask;   // row-major [B][h][w]
  const float* wx;     // row-major
  const float* wy;
  const float* du;     // DIAG layout [B][w*h]
  const float* dv;
  const float* derivs;
  float quarter_alpha, half_delta_over3, half_gamma_over3;
  float* sys;
};

// sor_coupled; every operand in DIAG layout
struct SorArgs {
  TvGeom t;
  const float* sys;  // [B][7][w*h]: a11,a12,a22,b1,b2,sh,sv
  float* du;         // [B][w*h]
  float* dv;
  int iterations;
  float omega;
};

// All fixed-point iterations of a level fused: compute_smoothness + compute_data + 2x sub_laplacian produce each
// anti-diagonal's system coefficients in registers, immediately consumed by the wavefront SOR of
// ofdis_sor.hip (h <= 64, gray).  Operands = the sdiag records of ofdis_prep.hip.
struct FusedArgs {
  TvGeom t;
  const float* d8;    // [records][8] derivatives (all zero where the warp's mask is zero)
  const float* wrec;  // [records][2] wx, wy
  float* uv;          // [records][2] du, dv (in/out; never read during the first iteration)
  int S;              // frames per strip (divides t.nframes)
  float quarter_alpha, half_delta_over3, half_gamma_over3;
  int iterations;  // SOR sweeps per fixed-point iteration (tv_solverit)
  float omega;
  int n_inner;     // fixed-point iterations run back to back inside one launch
  int total_frames;  // frames of the whole batch this launch is a part of (pipelined sub-batches); 0 = t.nframes
  // optional: AoS flow [B][h][w][2].  When the launcher picks a multi-wave variant it writes uu = wx + du, vv = wy + dv
  // of the last fixed-point iteration there itself (refine_variational.cpp:209-221, 92-99) instead of storing du, dv --
  // tv_finish is then not needed; launch_tv_fused reports that through *wrote_flow.
  float* flow_out;
  int mw_max_groups;  // frame groups (workgroups) up to which the multi-wave variants are launched (0 = never)
  int split;          // 0 = never the split (producer / solver wavefronts) variant of the multi-wave kernel
  int tp_pipe;        // 1: THROUGHPUT regime on the iteration-pipelined mapping (MODE 1) with strips of S frames: a workgroup of
                      // n_inner wavefronts per strip group, du / dv from iteration to iteration through LDS, the derivative
                      // records of a row read by n_inner wavefronts of ONE compute unit within a few steps (one HBM read, the
                      // others hit the L2) -- 48 instead of 56 n_inner bytes of HBM traffic per pixel and level
  int tall_group = 0; // levels of 65 ... 96 rows: 1 = several strips per workgroup, their rows beyond 64 in ONE shared wavefront
                      // (ofdis_fused_tall.hip: GROUPED), 0 = two wavefronts per strip
};
// cross-CU variant (one workgroup per fixed-point iteration of a frame group), kept out of FusedArgs so that the other
// variants' kernel arguments -- and register allocation -- stay what they were
struct FusedXcu {
  float* xbuf = nullptr;   // [n_inner - 1][nframes][w*h] granules of 4 floats {du, tag, dv, tag}; null: never
  int max_groups = 0;      // frame groups up to which the variant is launched (0 = never)
  int* err = nullptr;      // device-visible word, set to 1 when a hand-over row never arrived (results invalid); the variant
                           // is only launched with one
  unsigned wait_us = 0;    // microseconds a workgroup waits for one row before it gives up; 0 = the default (50 ms), 1 = not at all
  int drop = 0;            // test hook: the first iteration withholds its hand-over rows (ofdis_tuning::fused_xcu_drop)
};




// on-device pyramid (ofdis_pyr.hip; run_dense.cpp:130-178,298-311 restated)
hipError_t launch_pyr_base(const uint8_t* src, float* dst, int nframes, int wo, int ho, int W, int H, int noc, int l,
                           hipStream_t s);
hipError_t launch_pyr_down(const float* src, float* dst, int nframes, int w, int h, int noc, hipStream_t s);
// `down` (optional): also the next coarser level's unpadded image [B][h/2][w/2][noc] (what launch_pyr_down computes), in the
// same launch where the geometry allows (pyr_planes_fuses_down), else by a launch of its own
hipError_t launch_pyr_planes(const float* src, float* img, float* dx, float* dy, int nframes, int w, int h, int noc,
                             int pad, hipStream_t s, float* down = nullptr);
bool pyr_planes_fuses_down(int w, int h, int noc, int pad);

// x 2^sc_l, bilinear upsample (cv::resize INTER_LINEAR) and crop of the AoS result (run_dense.cpp:406-414)
hipError_t launch_upsample_crop(const float* flow, float* out, int nframes, int sw, int sh, int sc_l, int left, int top,
                                int wo, int ho, int channels, hipStream_t s);

// ---- stereo-depth mode (SELECTMODE=2; ofdis_de.hip)
struct DeSystemArgs {
  TvGeom t;
  const float* mask;   // row-major [B][h][w]
  const float* wx;     // row-major: the flow before the increment
  const float* du;     // diag: the increment of the previous fixed-point iterations
  int clamp;           // uu = wx (< 0: before the first solve), min(wx + du, 0) (0: left camera), max(., 0) (1: right)
  const float* derivs; // row-major [B][8*noc][h][w]
  float quarter_alpha, half_delta_over3, half_gamma_over3;
  float* sys;          // [B][4][w*h] diag: a11, b1, smooth_horiz, smooth_vert
};
struct DeSorArgs {
  TvGeom t;
  const float* sys;
  float* du;  // diag, in/out
  int iterations;
  float omega;
};
// All fixed-point iterations of a stereo level in one launch (de_fused_kernel, ofdis_de.hip: levels of <= 64 rows): the record
// arrays of the derivatives kernel in, du out (diag plane; never read during the first iteration)
struct DeFusedArgs {
  TvGeom t;
  const float* d8;    // [noc][B][w*h][8] derivative records (zero where the warp's mask is zero)
  const float* wrec;  // [B][w*h][2]: wx (and wy = 0)
  float* du;          // [B][w*h] diag
  float quarter_alpha, half_delta_over3, half_gamma_over3;
  int iterations;     // solver sweeps per fixed-point iteration
  float omega;
  int n_inner;
  int camlr;          // 0: left camera (uu = min(wx + du, 0)), 1: right (max)
};

namespace exact {
#include "ofdis_launchers.inc"
}  // namespace exact
namespace fused {
#include "ofdis_launchers.inc"
}  // namespace fused

}  // namespace ofdis
