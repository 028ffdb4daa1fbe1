// ofdis_kernels.h -- launchers of the gfx950 kernels (internal C++ API; the public ABI is include/ofdis.h).
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
  float* p_out;            // [B][nop][2]
  float* pweight;          // [B][nop][novals]
};
// snapshot of the kernel-selection knobs (include/ofdis.h: ofdis_tuning; ofdis_capi.hip); *epoch counts the changes
ofdis_tuning tuning(unsigned* epoch = nullptr);
// PatGridClass::{InitializeGrid, SetTargetImage, InitializeFromCoarserOF, Optimize}
hipError_t launch_patch_optimize(const DisArgs& a, hipStream_t s);

struct DensifyArgs {
  LevelGeom g;
  int nframes;
  const float* p;        // [B][nop][2]
  const float* pweight;  // [B][nop][novals]
  float* flow_aos;       // if non-null: AoS output
  float* wx;             // else planar outputs (row-major)
  float* wy;
  // forward-backward merging (usefbcon, patchgrid.cpp:277-375): the complementary grid's results, or null
  int stereo;               // one flow channel: flow_aos is [B][h][w], wy is not written
  unsigned idx_magic;       // set by launch_densify: ceil(2^32 / w) for the pixel-index split
  const float* cg_p;        // [B][nop][2]
  const float* cg_pweight;  // [B][nop][novals]
};
// PatGridClass::AggregateFlowDense as an order-preserving gather
hipError_t launch_densify(const DensifyArgs& a, hipStream_t s);

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
hipError_t launch_warp(const WarpArgs& a, hipStream_t s);
// get_derivatives.  im1 as for WarpArgs.src; im2w packed planar [B][noc][h][w].
// out [B][8][noc][h][w]
struct DerivArgs {
  TvGeom t;
  const float* im1;
  int im1_padded, pad, tmp_w, tmp_h;
  const float* im2w;
  float* out;  // [B][8][noc][h][w] row-major
};
hipError_t launch_derivatives(const DerivArgs& a, hipStream_t s);

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
};
bool tv_prep_supported(const TvGeom& t);
hipError_t launch_tv_prep(const PrepArgs& a, hipStream_t s);

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
hipError_t launch_tv_system(const SystemArgs& a, hipStream_t s);

// sor_coupled; every operand in DIAG layout
struct SorArgs {
  TvGeom t;
  const float* sys;  // [B][7][w*h]: a11,a12,a22,b1,b2,sh,sv
  float* du;         // [B][w*h]
  float* dv;
  int iterations;
  float omega;
};
hipError_t launch_sor(const SorArgs& a, hipStream_t s);

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
};
// cross-CU variant (one workgroup per fixed-point iteration of a frame group), kept out of FusedArgs so that the other
// variants' kernel arguments -- and register allocation -- stay what they were
struct FusedXcu {
  float* xbuf = nullptr;   // [n_inner - 1][nframes][w*h] granules of 4 floats {du, tag, dv, tag}; null: never
  int max_groups = 0;      // frame groups up to which the variant is launched (0 = never)
  int* err = nullptr;      // optional device-visible word, set to 1 when a hand-over row never arrived (results invalid)
};
bool tv_fused_supported(const TvGeom& t, int iterations);
// the fused kernel's trimmed divisions (ofdis_dev.h) need the three weights to be 0 or of ordinary magnitude
bool tv_fused_params_ok(float quarter_alpha, float half_delta_over3, float half_gamma_over3);
// 0 = one wavefront per strip group (throughput), 1 = a wavefront per fixed-point iteration, 2 = producer + solver each,
// 3 = one workgroup (three wavefronts) per fixed-point iteration, the iterations of a frame group on different CUs
int tv_fused_mode(const FusedArgs& a, const FusedXcu* x = nullptr);
hipError_t launch_tv_fused(const FusedArgs& a, hipStream_t s, bool* wrote_flow = nullptr, const FusedXcu* x = nullptr);

// uu = wx + du, vv = wy + dv (refine_variational.cpp:209-221, 92-99) in place: flow (AoS, row-major) holds wx, wy on entry
// and the refined flow on return; uv = the fused kernel's du, dv records
hipError_t launch_tv_finish_records(const TvGeom& t, float* flow_aos, const float* uv, int S, hipStream_t s);

// layout conversion of `nplanes` planes of w x h (row-major <-> diag); used by the per-function entry
// points, whose public interface is row-major
hipError_t launch_to_diag(const float* src_rm, float* dst_diag, int w, int h, long long nplanes, hipStream_t s);
hipError_t launch_from_diag(const float* src_diag, float* dst_rm, int w, int h, long long nplanes, hipStream_t s);

// uu=wx+du, vv=wy+dv -> AoS flow (refine_variational.cpp:209-221, 92-99); wx,wy row-major, du,dv DIAG
hipError_t launch_tv_finish(const TvGeom& t, const float* wx, const float* wy, const float* du, const float* dv,
                            float* flow_aos, hipStream_t s);
// AoS flow -> planar wx, wy (refine_variational.cpp:56-68)
hipError_t launch_flow_split(const TvGeom& t, const float* flow_aos, float* wx, float* wy, hipStream_t s);

// on-device pyramid (ofdis_pyr.hip; run_dense.cpp:130-178,298-311 restated)
hipError_t launch_pyr_base(const uint8_t* src, float* dst, int nframes, int wo, int ho, int W, int H, int noc, int l,
                           hipStream_t s);
hipError_t launch_pyr_down(const float* src, float* dst, int nframes, int w, int h, int noc, hipStream_t s);
hipError_t launch_pyr_planes(const float* src, float* img, float* dx, float* dy, int nframes, int w, int h, int noc,
                             int pad, hipStream_t s);

// x 2^sc_l, bilinear upsample (cv::resize INTER_LINEAR) and crop of the AoS result (run_dense.cpp:406-414)
hipError_t launch_upsample_crop(const float* flow, float* out, int nframes, int sw, int sh, int sc_l, int left, int top,
                                int wo, int ho, int channels, hipStream_t s);

// ---- stereo-depth mode (SELECTMODE=2; ofdis_de.hip)
struct DeSystemArgs {
  TvGeom t;
  const float* mask;   // row-major [B][h][w]
  const float* wx;     // row-major: the flow before the increment
  const float* uu;     // row-major: clamped flow + increment of the previous fixed-point iteration
  const float* du;     // diag
  const float* derivs; // row-major [B][8*noc][h][w]
  float quarter_alpha, half_delta_over3, half_gamma_over3;
  float* sys;          // [B][4][w*h] diag: a11, b1, smooth_horiz, smooth_vert
};
hipError_t launch_de_system(const DeSystemArgs& a, hipStream_t s);
struct DeSorArgs {
  TvGeom t;
  const float* sys;
  float* du;  // diag, in/out
  int iterations;
  float omega;
};
hipError_t launch_de_sor(const DeSorArgs& a, hipStream_t s);
hipError_t launch_de_update(const TvGeom& t, const float* wx, const float* du, float* uu, float* out, int camlr,
                            hipStream_t s);

}  // namespace ofdis
