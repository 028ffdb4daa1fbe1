// oracle/_ref driver -- TEST INFRASTRUCTURE, never part of the product library.
//
// Thin extern "C" entry points around the UNMODIFIED reference translation units, which the
// Makefile in this directory compiles in place from /root/reference (nothing is copied):
//   oflow.cpp patch.cpp patchgrid.cpp refine_variational.cpp FDF1.0.1/{image,opticalflow_aux,solver}.c
// Built twice: -DSELECTCHANNEL=1 -> libofdis_ref_int.so, -DSELECTCHANNEL=3 -> libofdis_ref_rgb.so
// (the FDF symbols change signature with SELECTCHANNEL, opticalflow_aux.h:12-26,44-48).
//
// Plane convention of this wrapper: "packed" row-major w*h floats per plane, channel planes
// consecutive ([c][h][w]); the wrapper copies into the reference's own image_t / color_image_t
// (stride = 4*ceil(w/4), image.c:23) and back, so callers never see the stride padding.
#include <Eigen/Core>
#include <iostream>
#include <vector>
#include <cstdio>
#include <cstring>

#include "oflow.h"
// The driver reads per-patch state (PatClass::GetParam) through PatGridClass::pat, which the
// reference keeps private.  Access control does not change layout; the sources stay untouched.
#define private public
#include "patchgrid.h"
#undef private
#include "refine_variational.h"

extern "C" {
#include "FDF1.0.1/image.h"
#include "FDF1.0.1/opticalflow_aux.h"
#include "FDF1.0.1/solver.h"
}

#if (SELECTCHANNEL == 3)
#define NOC 3
typedef color_image_t cimg_t;
static cimg_t* cimg_new(int w, int h) { return color_image_new(w, h); }
static void cimg_delete(cimg_t* p) { color_image_delete(p); }
#else
#define NOC 1
typedef image_t cimg_t;
static cimg_t* cimg_new(int w, int h) { return image_new(w, h); }
static void cimg_delete(cimg_t* p) { image_delete(p); }
#endif

namespace {

image_t* img_from(const float* src, int w, int h) {
  image_t* im = image_new(w, h);
  memset(im->c1, 0, sizeof(float) * im->stride * h);
  if (src)
    for (int y = 0; y < h; ++y) memcpy(im->c1 + y * im->stride, src + (size_t)y * w, sizeof(float) * w);
  return im;
}
void img_to(float* dst, const image_t* im) {
  for (int y = 0; y < im->height; ++y)
    memcpy(dst + (size_t)y * im->width, im->c1 + y * im->stride, sizeof(float) * im->width);
}
cimg_t* cimg_from(const float* src, int w, int h) {
  cimg_t* im = cimg_new(w, h);
  memset(im->c1, 0, sizeof(float) * im->stride * h * NOC);
  if (src)
    for (int c = 0; c < NOC; ++c)
      for (int y = 0; y < h; ++y)
        memcpy(im->c1 + ((size_t)c * h + y) * im->stride, src + ((size_t)c * h + y) * w, sizeof(float) * w);
  return im;
}
void cimg_to(float* dst, const cimg_t* im) {
  const int w = im->width, h = im->height;
  for (int c = 0; c < NOC; ++c)
    for (int y = 0; y < h; ++y)
      memcpy(dst + ((size_t)c * h + y) * w, im->c1 + ((size_t)c * h + y) * im->stride, sizeof(float) * w);
}

// Same derivations as OFClass::OFClass (oflow.cpp:76-108, 138-157), needed to drive the
// per-level classes (PatGridClass, VarRefClass) on their own.
void fill_params(OFC::optparam& op, OFC::camparam& cp, int w_lv, int h_lv, int level, int imgpadding,
                 int max_iter, int min_iter, float dp_thresh, float dr_thresh, float res_thresh,
                 int p_samp_s, float patove, int costfct, int noc, int patnorm, float tv_alpha,
                 float tv_gamma, float tv_delta, int tv_innerit, int tv_solverit, float tv_sor) {
#if (SELECTMODE == 1)
  op.nop = 2;  // oflow.cpp:76-80
#else
  op.nop = 1;
#endif
  op.p_samp_s = p_samp_s;
  op.outlierthresh = (float)op.p_samp_s / 2;
  op.patove = patove;
  op.sc_f = level;
  op.sc_l = level;
  op.max_iter = max_iter;
  op.min_iter = min_iter;
  op.dp_thresh = dp_thresh * dp_thresh;
  op.dr_thresh = dr_thresh;
  op.res_thresh = res_thresh;
  op.steps = std::max(1, (int)floor(op.p_samp_s * (1 - op.patove)));
  op.novals = noc * p_samp_s * p_samp_s;
  op.usefbcon = 0;
  op.costfct = costfct;
  op.noc = noc;
  op.patnorm = patnorm;
  op.verbosity = 0;
  op.noscales = 1;
  op.usetvref = 1;
  op.tv_alpha = tv_alpha;
  op.tv_gamma = tv_gamma;
  op.tv_delta = tv_delta;
  op.tv_innerit = tv_innerit;
  op.tv_solverit = tv_solverit;
  op.tv_sor = tv_sor;
  op.normoutlier_tmpbsq = (OFC::v4sf){op.normoutlier * op.normoutlier, op.normoutlier * op.normoutlier,
                                      op.normoutlier * op.normoutlier, op.normoutlier * op.normoutlier};
  op.normoutlier_tmp2bsq = __builtin_ia32_mulps(op.normoutlier_tmpbsq, op.twos);
  op.normoutlier_tmp4bsq = __builtin_ia32_mulps(op.normoutlier_tmpbsq, op.fours);

  cp.sc_fct = (float)pow(2, -level);
  cp.height = h_lv;
  cp.width = w_lv;
  cp.imgpadding = imgpadding;
  cp.tmp_lb = -(float)op.p_samp_s / 2;
  cp.tmp_ubw = (float)(cp.width + op.p_samp_s / 2 - 2);
  cp.tmp_ubh = (float)(cp.height + op.p_samp_s / 2 - 2);
  cp.tmp_w = cp.width + 2 * imgpadding;
  cp.tmp_h = cp.height + 2 * imgpadding;
  cp.curr_lv = level;
  cp.camlr = 0;
}

}  // namespace

extern "C" {

int ofdis_ref_noc(void) { return NOC; }

int ofdis_ref_wave64_order(void) {
#ifdef OFDIS_SHIM_WAVE64
  return 1;
#else
  return 0;
#endif
}

// The drop-in boundary itself: OFC::OFClass::OFClass (oflow.h:84-111).
int ofdis_ref_flow(const float** im_ao, const float** im_ao_dx, const float** im_ao_dy, const float** im_bo,
                   const float** im_bo_dx, const float** im_bo_dy, int imgpadding, float* outflow,
                   const float* initflow, int width, int height, int sc_f, int sc_l, int max_iter,
                   int min_iter, float dp_thresh, float dr_thresh, float res_thresh, int p_samp_s,
                   float patove, int usefbcon, int costfct, int noc, int patnorm, int usetvref,
                   float tv_alpha, float tv_gamma, float tv_delta, int tv_innerit, int tv_solverit,
                   float tv_sor, int verbosity) {
  if (noc != NOC) return -1;
  OFC::OFClass ofc(im_ao, im_ao_dx, im_ao_dy, im_bo, im_bo_dx, im_bo_dy, imgpadding, outflow, initflow, width,
                   height, sc_f, sc_l, max_iter, min_iter, dp_thresh, dr_thresh, res_thresh, p_samp_s, patove,
                   usefbcon != 0, costfct, noc, patnorm, usetvref != 0, tv_alpha, tv_gamma, tv_delta,
                   tv_innerit, tv_solverit, tv_sor, verbosity);
  fflush(stdout);
  return 0;
}

// One level of the DIS search on its own: PatGridClass::{InitializeGrid, SetTargetImage,
// InitializeFromCoarserOF, Optimize, AggregateFlowDense} (patchgrid.cpp:98-141,195-275,377-397).
// Images are the padded, channel-interleaved level planes exactly as OFClass receives them.
// p_out: nopatches*2 (patch index i = x*noph + y, patchgrid.cpp:66); flow_out: w*h*2 AoS.
int ofdis_ref_patchgrid_level(const float* im_a, const float* im_a_dx, const float* im_a_dy, const float* im_b,
                              int w_lv, int h_lv, int level, int imgpadding, int max_iter, int min_iter,
                              float dp_thresh, float dr_thresh, float res_thresh, int p_samp_s, float patove,
                              int costfct, int noc, int patnorm, const float* flow_prev, float* p_out,
                              float* flow_out, int* nopatches_out) {
  if (noc != NOC) return -1;
  OFC::optparam op;
  OFC::camparam cpl, cpr;
  fill_params(op, cpl, w_lv, h_lv, level, imgpadding, max_iter, min_iter, dp_thresh, dr_thresh, res_thresh,
              p_samp_s, patove, costfct, noc, patnorm, 10, 10, 5, 1, 3, 1.6f);
  cpr = cpl;
  cpr.camlr = 1;
  OFC::PatGridClass grid(&cpl, &cpr, &op);
  grid.InitializeGrid(im_a, im_a_dx, im_a_dy);
  grid.SetTargetImage(im_b, nullptr, nullptr);
  if (flow_prev) grid.InitializeFromCoarserOF(flow_prev);
  grid.Optimize();
  const int n = grid.GetNoPatches();
  if (nopatches_out) *nopatches_out = n;
  if (p_out)
    for (int i = 0; i < n; ++i) {
#if (SELECTMODE == 1)
      const Eigen::Vector2f* d = grid.pat[i]->GetParam();  // p_iter (patch.h:78)
      p_out[2 * i] = (*d)[0];
      p_out[2 * i + 1] = (*d)[1];
#else  // stereo: one horizontal displacement per patch (patch.h:80)
      const Eigen::Matrix<float, 1, 1>* d = grid.pat[i]->GetParam();
      p_out[2 * i] = (*d)[0];
      p_out[2 * i + 1] = 0.0f;
#endif
    }
  if (flow_out) grid.AggregateFlowDense(flow_out);
  return 0;
}

// One level of the variational refinement: VarRefClass ctor (refine_variational.cpp:25-116).
// flow: w*h*2 AoS, refined in place.
int ofdis_ref_varref_level(const float* im_a, const float* im_b, int w_lv, int h_lv, int level, int imgpadding,
                           int p_samp_s, int noc, float tv_alpha, float tv_gamma, float tv_delta,
                           int tv_innerit, int tv_solverit, float tv_sor, float* flow) {
  if (noc != NOC) return -1;
  OFC::optparam op;
  OFC::camparam cpl, cpr;
  fill_params(op, cpl, w_lv, h_lv, level, imgpadding, 12, 12, 0.05f, 0.95f, 0.0f, p_samp_s, 0.4f, 0, noc, 1,
              tv_alpha, tv_gamma, tv_delta, tv_innerit, tv_solverit, tv_sor);
  cpr = cpl;
  cpr.camlr = 1;
  OFC::VarRefClass vr(im_a, nullptr, nullptr, im_b, nullptr, nullptr, &cpl, &cpr, &op, flow);
  return 0;
}

// ---- FDF1.0.1 kernels, one wrapper each (opticalflow_aux.c, solver.c) -----------------------
// image_warp (opticalflow_aux.c:18-60). src/dst: NOC planes; wx, wy, mask: 1 plane.
void ofdis_ref_image_warp(float* dst, float* mask, const float* src, const float* wx, const float* wy, int w,
                          int h) {
  cimg_t* s = cimg_from(src, w, h);
  cimg_t* d = cimg_from(nullptr, w, h);
  image_t* m = img_from(nullptr, w, h);
  image_t* x = img_from(wx, w, h);
  image_t* y = img_from(wy, w, h);
  image_warp(d, m, s, x, y);
  cimg_to(dst, d);
  img_to(mask, m);
  cimg_delete(s); cimg_delete(d); image_delete(m); image_delete(x); image_delete(y);
}

// get_derivatives (opticalflow_aux.c:65-116) with the filter of refine_variational.cpp:45-46.
// out: 8 groups of NOC planes in the order Ix,Iy,Iz,Ixx,Ixy,Iyy,Ixz,Iyz.
void ofdis_ref_get_derivatives(const float* im1, const float* im2, float* out, int w, int h) {
  float deriv_filter[3] = {0.0f, -8.0f / 12.0f, 1.0f / 12.0f};
  convolution_t* deriv = convolution_new(2, deriv_filter, 0);
  cimg_t* a = cimg_from(im1, w, h);
  cimg_t* b = cimg_from(im2, w, h);
  cimg_t* o[8];
  for (int k = 0; k < 8; ++k) o[k] = cimg_from(nullptr, w, h);
  get_derivatives(a, b, deriv, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
  for (int k = 0; k < 8; ++k) { cimg_to(out + (size_t)k * NOC * w * h, o[k]); cimg_delete(o[k]); }
  cimg_delete(a); cimg_delete(b);
  convolution_delete(deriv);
}

// compute_smoothness (opticalflow_aux.c:123-165) with deriv_flow of refine_variational.cpp:47-48.
void ofdis_ref_compute_smoothness(float* sh, float* sv, const float* uu, const float* vv, float quarter_alpha,
                                  int w, int h) {
  float deriv_filter_flow[2] = {0.0f, -0.5f};
  convolution_t* deriv_flow = convolution_new(1, deriv_filter_flow, 0);
  image_t* u = img_from(uu, w, h);
  image_t* v = img_from(vv, w, h);
  image_t* a = img_from(nullptr, w, h);
  image_t* b = img_from(nullptr, w, h);
  compute_smoothness(a, b, u, v, deriv_flow, quarter_alpha);
  img_to(sh, a); img_to(sv, b);
  image_delete(u); image_delete(v); image_delete(a); image_delete(b);
  convolution_delete(deriv_flow);
}

// compute_data (opticalflow_aux.c:310-438). derivs: 8 groups of NOC planes (order as above).
// out: a11,a12,a22,b1,b2 (5 planes).
void ofdis_ref_compute_data(float* out, const float* mask, const float* du, const float* dv, const float* derivs,
                            float half_delta_over3, float half_beta, float half_gamma_over3, int w, int h) {
  image_t* o[5];
  for (int k = 0; k < 5; ++k) o[k] = img_from(nullptr, w, h);
  image_t* m = img_from(mask, w, h);
  image_t* u = img_from(du, w, h);
  image_t* v = img_from(dv, w, h);
  image_t* z = img_from(nullptr, w, h);  // wx, wy, uu, vv: pointers advanced, never read
  cimg_t* d[8];
  for (int k = 0; k < 8; ++k) d[k] = cimg_from(derivs + (size_t)k * NOC * w * h, w, h);
  compute_data(o[0], o[1], o[2], o[3], o[4], m, z, z, u, v, z, z, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7],
               half_delta_over3, half_beta, half_gamma_over3);
  for (int k = 0; k < 5; ++k) { img_to(out + (size_t)k * w * h, o[k]); image_delete(o[k]); }
  for (int k = 0; k < 8; ++k) cimg_delete(d[k]);
  image_delete(m); image_delete(u); image_delete(v); image_delete(z);
}

// sub_laplacian (opticalflow_aux.c:172-199). dst is read-modify-write.
void ofdis_ref_sub_laplacian(float* dst, const float* src, const float* wh, const float* wv, int w, int h) {
  image_t* d = img_from(dst, w, h);
  image_t* s = img_from(src, w, h);
  image_t* a = img_from(wh, w, h);
  image_t* b = img_from(wv, w, h);
  sub_laplacian(d, s, a, b);
  img_to(dst, d);
  image_delete(d); image_delete(s); image_delete(a); image_delete(b);
}

// sor_coupled (solver.c:77-421) / sor_coupled_slow_but_readable (solver.c:19-72) when slow!=0.
// du, dv updated in place; a11,a12,a22 are overwritten by the block inverse (returned too).
void ofdis_ref_sor_coupled(float* du, float* dv, float* a11, float* a12, float* a22, const float* b1,
                           const float* b2, const float* sh, const float* sv, int iterations, float omega, int w,
                           int h, int slow) {
  image_t *u = img_from(du, w, h), *v = img_from(dv, w, h), *A = img_from(a11, w, h), *B = img_from(a12, w, h),
          *C = img_from(a22, w, h), *r1 = img_from(b1, w, h), *r2 = img_from(b2, w, h), *H = img_from(sh, w, h),
          *V = img_from(sv, w, h);
  if (slow)
    sor_coupled_slow_but_readable(u, v, A, B, C, r1, r2, H, V, iterations, omega);
  else
    sor_coupled(u, v, A, B, C, r1, r2, H, V, iterations, omega);
  img_to(du, u); img_to(dv, v); img_to(a11, A); img_to(a12, B); img_to(a22, C);
  image_delete(u); image_delete(v); image_delete(A); image_delete(B); image_delete(C);
  image_delete(r1); image_delete(r2); image_delete(H); image_delete(V);
}

}  // extern "C"
