/* ofdis_oracle.h -- CPU restatement of the OF_DIS hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (libofdis_hip.so, run_OF_*) never links or calls it.
 *
 * Parity pinning: the reference ships no tests, vectors or fixtures, so this
 * restatement is pinned against the reference ITSELF: tests/test_oracle_vs_ref.py compares every
 * function below bit-for-bit with the unmodified reference sources compiled in place
 * (oracle/_ref, see oracle/Makefile), and tests/golden/ holds vectors produced by that build.
 *
 * All planes are "packed": row-major w*h floats, channel planes consecutive ([c][h][w]) for the
 * TV functions; the padded pyramid planes are channel-interleaved exactly as the reference's.
 */
#ifndef OFDIS_ORACLE_H_
#define OFDIS_ORACLE_H_

#include <stdint.h>
#include "../include/ofdis.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 0 = strictly sequential sums (matches _ref built with the default shim),
 * 1 = 64-lane butterfly order (matches _ref built with -DOFDIS_SHIM_WAVE64 and the HIP kernels) */
void oracle_set_reduce_order(int wave64);
int oracle_get_reduce_order(void);

/* FDF1.0.1 kernels */
void oracle_image_warp(float* dst, float* mask, const float* src, const float* wx, const float* wy, int w, int h,
                       int noc);
void oracle_get_derivatives(const float* im1, const float* im2, float* out, int w, int h, int noc);
void oracle_compute_smoothness(float* sh, float* sv, const float* uu, const float* vv, float quarter_alpha, int w,
                               int h);
void oracle_compute_data(float* out5, const float* mask, const float* du, const float* dv, const float* derivs,
                         float half_delta_over3, float half_gamma_over3, int w, int h, int noc);
void oracle_sub_laplacian(float* dst, const float* src, const float* wh, const float* wv, int w, int h);
void oracle_sor_coupled(float* du, float* dv, float* a11, float* a12, float* a22, const float* b1,
                        const float* b2, const float* sh, const float* sv, int iterations, float omega, int w,
                        int h);
void oracle_sor_coupled_slow(float* du, float* dv, const float* a11, const float* a12, const float* a22,
                             const float* b1, const float* b2, const float* sh, const float* sv, int iterations,
                             float omega, int w, int h);

/* one level of VarRefClass; flow w*h*2 AoS in place */
int oracle_varref_level(const ofdis_params* p, int level, const float* im_a, const float* im_b, float* flow);
/* one level of PatGridClass (+ AggregateFlowDense).  p_out: nopatches*2 or NULL, pweight_out:
 * nopatches*novals or NULL, flow_out: w*h*2 or NULL */
int oracle_patchgrid_level(const ofdis_params* p, int level, const float* im_a, const float* im_a_dx,
                           const float* im_a_dy, const float* im_b, const float* flow_prev, float* p_out,
                           float* pweight_out, float* flow_out, int* nopatches_out);
/* OFC::OFClass::OFClass */
int oracle_flow(const ofdis_params* p, const float* const* im_a, const float* const* im_a_dx,
                const float* const* im_a_dy, const float* const* im_b, float* outflow, const float* initflow,
                float* level_flows /* optional: concatenated dense flow of every level, coarse to fine */);

/* host pre/post-processing of run_dense.cpp (OpenCV restated; exact for 8-bit input) */
/* sizes of the padded level-0 image for a given original size and sc_f (run_dense.cpp:298-311) */
void oracle_padded_size(int width_org, int height_org, int sc_f, int* width, int* height);
/* elements of one padded pyramid plane at level l */
size_t oracle_plane_elems(const ofdis_params* p, int level);
/* builds levels 0..sc_f of {img, dx, dy} from an 8-bit image (noc interleaved channels).
 * img/dx/dy: arrays of sc_f+1 caller-allocated planes (oracle_plane_elems each). */
void oracle_build_pyramid(const ofdis_params* p, const uint8_t* img_u8, int width_org, int height_org,
                          float* const* img, float* const* dx, float* const* dy);
/* flow (w>>sc_l x h>>sc_l, AoS) -> full-resolution cropped flow width_org x height_org
 * (run_dense.cpp:406-414) */
void oracle_upsample_crop(const ofdis_params* p, const float* flow, int width_org, int height_org, float* out);

#ifdef __cplusplus
}
#endif
#endif
