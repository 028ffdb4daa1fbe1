/* ofdis.h -- C ABI of the MI355X-native OF_DIS hot path (libofdis_hip.so).
 *
 * The reference has exactly one API: the constructor of OFC::OFClass
 * (reference oflow.h:84-111, definition oflow.cpp:32-363, sole call site run_dense.cpp:391-400).
 * Everything happens inside that constructor; nothing is retained afterwards.  ofdis_flow() below
 * is that constructor as a C function (same argument meaning, same buffers, same output), and
 * ofdis_batch_* is the same computation over many independent frame pairs resident in HBM, which
 * is how a GPU reaches throughput on problems this small.
 *
 * Plain pointers and sizes only; no C++/torch types.  Device pointers are HIP device pointers,
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *
 * TWO arithmetic contracts, one source.  A context fixes its contract at creation (ofdis_tuning::contract,
 * OFDIS_CONTRACT=fused in the environment); everything else in this header holds for both.
 *   exact (0, the LIBRARY default; run_OF_* / ofdis_flow use it unless told otherwise): fp32 throughout, every operation
 *     separately rounded (no FMA contraction), IEEE divide/sqrt, same operation order as the reference's SSE path; vector
 *     reductions (the only place the reference leaves the order to Eigen) use the order documented in DESIGN.md
 *     ("Reduction order": <= 64 entries: 8 stride-8 partials, then distance 4, 1, 2; more: 64 strided partials, then a
 *     butterfly at distance 1..32).  The result is bit-identical to the reference sources compiled against
 *     oracle/eigen_shim with -DOFDIS_SHIM_WAVE64, whatever kernel mapping the batch size selects.
 *   fused (1, what `python bench.py` TIMES by default, after its gate passed in that run): the tolerance contract of
 *     BASELINE.json's north star.  Same algorithm, taps, control flow and reduction shapes; multiply-adds contract to
 *     v_fma_f32 and quotients / roots are the hardware's 1-ulp v_rcp_f32 / v_rsq_f32 / v_sqrt_f32.  Bound (asserted in
 *     tests/test_gpu_contract.py against the PLAIN reference build, sequential sums): mean EPE < 1e-4 px and max EPE <
 *     1e-3 px on the full-resolution flow.  Results then depend on the kernel mapping (i.e. on the batch size) by up to
 *     3e-4 px; they stay deterministic and independent of a frame's slot and neighbours.
 */
#ifndef OFDIS_H_
#define OFDIS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version.  2 (round 5): ofdis_tuning grew to 16 ints (fused_tp_pipe, fused_xcu_spin, contract were appended in round 4
 * without a bump), ofdis_batch_status and ofdis_batch_upsample_frames were added.  3 (round 6): streams, pinned host memory
 * asynchronous copies and events (ofdis_stream_create, ofdis_host_alloc, ofdis_memcpy_h2d_async / _d2h_async,
 * ofdis_event_*), ofdis_build_id; ofdis_tuning grew to
 * 20 ints (fused_xcu_drop, prep_densify, fused_tall_group, fused_rgb_min) and fused_xcu_spin became a time in microseconds.  A caller checks ofdis_version() ==
 * OFDIS_VERSION before passing structs (of_dis_amd/capi.py does at load). */
#define OFDIS_VERSION 3

/* status codes (the reference reports nothing and has UB on bad input; we return a status) */
enum {
  OFDIS_OK = 0,
  OFDIS_ERR_INVALID = -1,      /* bad argument / unsupported parameter combination */
  OFDIS_ERR_UNSUPPORTED = -2,  /* valid in the reference but outside this path */
  OFDIS_ERR_DEVICE = -3,       /* HIP runtime error (ofdis_last_error() has the text) */
  OFDIS_ERR_NOMEM = -4
};

/* The run-time parameters of OFC::OFClass::OFClass, in the constructor's order
 * (oflow.h:91-111; parsed at oflow.cpp:76-108).  `noc` is a run-time value here instead of the
 * reference's per-binary SELECTCHANNEL. */
typedef struct ofdis_params {
  int   width, height;      /* size of level 0 (already padded to a multiple of 2^sc_f), oflow.h:92 */
  int   imgpadding;         /* border of every pyramid plane, == p_samp_s at the call site (run_dense.cpp:393) */
  int   sc_f, sc_l;         /* coarsest / finest pyramid level used */
  int   max_iter, min_iter;
  float dp_thresh, dr_thresh, res_thresh;
  int   p_samp_s;           /* patch edge length P */
  float patove;             /* patch overlap in [0,1) */
  int   usefbcon;           /* forward-backward merging (oflow.cpp:162-170, patchgrid.cpp:277-375) */
  int   costfct;            /* 0 L2, 1 L1, 2 pseudo-Huber (patch.cpp:230-261) */
  int   noc;                /* channels: 1 (run_OF_INT) or 3 (run_OF_RGB) */
  int   patnorm;            /* subtract patch mean */
  int   usetvref;           /* TV-L1 variational refinement on/off */
  float tv_alpha, tv_gamma, tv_delta;
  int   tv_innerit, tv_solverit;
  float tv_sor;
  int   verbosity;          /* 0 silent, 1 total time, 2 per-level TIME lines (oflow.cpp:179,303,359) */
  int   selectmode;         /* the reference's compile-time SELECTMODE: 0 or 1 = optical flow (run_OF_*, two flow
                             * channels), 2 = stereo depth (run_DE_*: ONE channel = horizontal displacement <= 0,
                             * patch.cpp:188-193, refine_variational.cpp:245-336).  Every flow array below then has
                             * one float per pixel instead of two. */
} ofdis_params;

/* Fill `p` with the reference's operating point 1..4 for an image of `width_org` columns
 * (run_dense.cpp:225-265, AutoFirstScaleSelect run_dense.cpp:180-183).  width/height/imgpadding are
 * set by the caller after padding.  Returns OFDIS_OK or OFDIS_ERR_INVALID. */
int ofdis_params_oppoint(ofdis_params* p, int op_point, int width_org, int noc);

const char* ofdis_last_error(void);
int ofdis_version(void);
/* Identity of the kernels in this library: a hash over the kernel sources and compiler flags it was built from
 * (of_dis_amd/build.py: source_id).  Counter-derived measurements (profiles/traffic_*.json) carry the id of the library
 * they were collected on; bench.py attaches them only to a library with the same id. */
const char* ofdis_build_id(void);
/* number of HIP devices visible / select one (one process per GPU: call once at start) */
int ofdis_device_count(void);
int ofdis_set_device(int device);
/* PCI bus id ("0000:c1:00.0") of a visible device into buf (at least 16 bytes): a multi-rank launcher can check that the
 * ranks really sit on different GPUs */
int ofdis_device_pci_bus_id(int device, char* buf, int len);

/* ---------------------------------------------------------------------------------------------
 * Drop-in for the constructor: host pointers in, host flow out, synchronous.
 * Replaces: OFC::OFClass::OFClass(...) oflow.h:84-111 / run_dense.cpp:391-400.
 * im_*: arrays of sc_f+1 host pointers; entries sc_l..sc_f must be valid (oflow.h:85-87).  Each
 * plane is row-major fp32, (w/2^l + 2*imgpadding) x (h/2^l + 2*imgpadding) x noc, channel
 * interleaved, images replicate-padded, gradients zero-padded (run_dense.cpp:166-175).
 * im_b_dx / im_b_dy may be NULL when usefbcon == 0 (never read then).
 * outflow: 2*(w>>sc_l)*(h>>sc_l) floats, AoS (u,v), fully overwritten.
 * initflow: optional (w>>(sc_f+1))*(h>>(sc_f+1))*2 floats or NULL (oflow.cpp:217-220).
 * ------------------------------------------------------------------------------------------- */
int ofdis_flow(const ofdis_params* p,
               const float* const* im_a, const float* const* im_a_dx, const float* const* im_a_dy,
               const float* const* im_b, const float* const* im_b_dx, const float* const* im_b_dy,
               float* outflow, const float* initflow);
/* ofdis_flow keeps the device contexts of the last few parameter sets (buffers, a stream, pinned staging) so that a loop over frame pairs -- the reference's usage, one constructor per pair -- pays for them
 * once; calls are serialised internally.  ofdis_flow_cache_clear() releases them (e.g. before hipDeviceReset or at
 * shutdown); the next ofdis_flow() builds a fresh one. */
void ofdis_flow_cache_clear(void);

/* ---------------------------------------------------------------------------------------------
 * Batched, device-resident form (the throughput path).
 * A batch context owns every intermediate buffer for `nframes` frame pairs of one geometry.
 * Pyramid layout in HBM, per level l in [sc_l, sc_f], one array per plane kind:
 *     float plane[nframes][tmp_h_l][tmp_w_l][noc]    tmp_* = level size + 2*imgpadding
 * i.e. the reference's per-level plane, frame-major.  Output: float flow[nframes][h_l][w_l][2]
 * at l = sc_l.
 * ------------------------------------------------------------------------------------------- */
typedef struct ofdis_batch ofdis_batch;

/* nframes: 1..65535.  All device memory of the context is one allocation. */
int ofdis_batch_create(ofdis_batch** out, const ofdis_params* p, int nframes);
void ofdis_batch_destroy(ofdis_batch* b);

/* device pointers to the context-owned input planes of level l: kind 0 = image A, 1 = A_dx,
 * 2 = A_dy, 3 = image B (with usefbcon also 4 = B_dx, 5 = B_dy).  The caller fills them (hipMemcpy, its own kernels, ofdis_batch_upload
 * or ofdis_batch_build_pyramids_u8). */
float* ofdis_batch_input(ofdis_batch* b, int level, int kind);
size_t ofdis_batch_input_elems(const ofdis_batch* b, int level); /* floats per frame per plane */
/* copy one frame's host pyramid (the ofdis_flow() layout) into slot `frame`.  Like every upload below this ENQUEUES
 * copies on `stream`: the host arrays must stay valid and unchanged until the stream has been synchronised
 * (ofdis_sync). */
int ofdis_batch_upload(ofdis_batch* b, int frame, const float* const* im_a, const float* const* im_a_dx,
                       const float* const* im_a_dy, const float* const* im_b, void* stream);
/* build all input planes of levels sc_l..sc_f on the device from raw 8-bit frames
 * (run_dense.cpp:130-178,298-311,326-344 restated on device): img_a/img_b are device pointers to
 * [nframes][height_org][width_org][noc] uint8; padding to params.width x params.height is applied
 * as the reference does (replicate, floor/ceil split).  Exact (bit-identical to the fp32 OpenCV arithmetic of the
 * reference for 8-bit input) for sc_f <= 7; larger sc_f returns OFDIS_ERR_UNSUPPORTED. */
int ofdis_batch_build_pyramids_u8(ofdis_batch* b, const uint8_t* img_a, const uint8_t* img_b, int width_org,
                                  int height_org, void* stream);

/* usefbcon = 1 only: the gradient pyramids of the second image (the backward grid's templates; never read otherwise) */
int ofdis_batch_upload_b_gradients(ofdis_batch* b, int frame, const float* const* im_b_dx, const float* const* im_b_dy,
                                   void* stream);

/* Warm start (the reference's `initflow`, oflow.cpp:217-220; e.g. the previous frame pair's flow of a video):
 * per frame (w >> (sc_f+1)) x (h >> (sc_f+1)) x 2 floats, AoS.  set_initflow borrows a device array
 * [nframes][ofdis_batch_initflow_elems] (NULL switches the warm start off again); upload_initflow copies one
 * frame's host array into a buffer owned by the batch (frames never uploaded start from zero flow). */
size_t ofdis_batch_initflow_elems(const ofdis_batch* b);
int ofdis_batch_set_initflow(ofdis_batch* b, const float* initflow_dev);
int ofdis_batch_upload_initflow(ofdis_batch* b, int frame, const float* initflow_host, void* stream);

/* enqueue the whole hot path (all levels, DIS + densify + TV) for all frames on `stream` */
int ofdis_batch_run(ofdis_batch* b, void* stream);
/* Throughput option: cut the batch into `sub_batches` (2..4; 0 or 1 = off, the default) parts that run on internal
 * streams forked from the caller's stream and NOT joined at the end of ofdis_batch_run, so that consecutive passes
 * overlap (coarse-level kernels of one part fill the issue slots left by another part's kernels).  Results are then
 * complete on a stream only after ofdis_batch_join(b, stream); ofdis_batch_download / _upsample join themselves.
 * Do not modify the inputs of a pass before joining it.  Results are identical in either mode. */
int ofdis_batch_set_pipeline(ofdis_batch* b, int sub_batches);
int ofdis_batch_join(ofdis_batch* b, void* stream);
/* Launch-graph replay: the schedule of a context never changes, so an un-pipelined ofdis_batch_run can replay it as one
 * hipGraph launch instead of ~15 kernel launches.  mode 0 = direct launches (the default: measured, the asynchronous
 * launches already overlap the execution and the replay is not faster), 1 = captured at the next pass, -1 = captured at
 * the second pass of the context.  Results are identical; timing mode, verbosity > 0 and pipelined mode always launch
 * directly. */
int ofdis_batch_set_graph(ofdis_batch* b, int mode);
/* device pointer to the result, [nframes][h>>sc_l][w>>sc_l][2] ([..][1] in stereo-depth mode); in pipelined mode valid on
 * a stream after ofdis_batch_join(b, stream) */
const float* ofdis_batch_flow(const ofdis_batch* b);
/* After the caller has synchronised the stream(s) of the context's last pass by its own means: OFDIS_OK, or
 * OFDIS_ERR_DEVICE when that pass's results are invalid.  The one kernel that can report this is the cross-CU variant of the
 * fused TV kernel (contexts of <= 768 frames; ofdis_tuning::fused_xcu_max): its workgroups hand rows to each other through
 * memory and wait, bounded (ofdis_tuning::fused_xcu_spin: 50 ms), for workgroups the dispatcher started earlier; a wait that
 * expires marks the pass as failed.
 * The failure stays with the context until its next ofdis_batch_run, which no longer uses the variant: run again.
 * ofdis_batch_download, ofdis_flow (which repeats the pass itself) and ofdis_sync on the stream of the pass report the same
 * condition; callers that synchronise through HIP directly call this. */
int ofdis_batch_status(ofdis_batch* b);
/* device pointer to the dense flow of an intermediate level (for per-level parity tests) */
const float* ofdis_batch_level_flow(const ofdis_batch* b, int level);
int ofdis_batch_download(ofdis_batch* b, int frame, float* outflow_host, void* stream);
/* The step after the path (run_dense.cpp:406-414): values x 2^sc_l, bilinear upsample by 2^sc_l (cv::resize
 * INTER_LINEAR semantics: half-pixel centres, clamped borders) and crop of the 2^sc_f padding, for all frames:
 * out_dev = device [nframes][height_org][width_org][2] ([..][1] in stereo-depth mode).  Enqueues on `stream`. */
int ofdis_batch_upsample(ofdis_batch* b, float* out_dev, int width_org, int height_org, void* stream);
/* The same for the frames [first_frame, first_frame + count) only: out_dev = device [count][height_org][width_org][2].
 * (A host driver that streams a long sequence through one context writes its .flo files chunk by chunk; bench.py takes
 * its sample of the timed context's result this way.) */
int ofdis_batch_upsample_frames(ofdis_batch* b, int first_frame, int count, float* out_dev, int width_org,
                                int height_org, void* stream);

/* Kernel timing for the roofline report: when enabled, ofdis_batch_run brackets every launch of
 * the named kernel class with hipEvents on `stream`; ofdis_batch_kernel_time returns the summed
 * milliseconds and launch count since the last reset (synchronises the events). */
enum { OFDIS_K_WARP = 0, OFDIS_K_DERIV = 1, OFDIS_K_SYSTEM = 2, OFDIS_K_SOR = 3, OFDIS_K_PATCH = 4,
       OFDIS_K_DENSIFY = 5, OFDIS_K_UPDATE = 6, OFDIS_K_FUSED = 7, OFDIS_K_COUNT = 8 };
int ofdis_batch_timing(ofdis_batch* b, int enable);
int ofdis_batch_kernel_time(ofdis_batch* b, int kernel_class, double* ms_sum, long* launches);
/* the same launch by launch, in launch order (a pass launches a class once per level, coarsest level first): fills
 * ms_out[0 .. min(capacity, *launches) - 1] */
int ofdis_batch_kernel_times(ofdis_batch* b, int kernel_class, double* ms_out, int capacity, int* launches);

/* ---------------------------------------------------------------------------------------------
 * Kernel-selection knobs.  Several stages exist in more than one mapping of the SAME arithmetic (every setting but
 * `contract` gives bit-identical results); the library picks by geometry and batch size.  The knobs are process-wide, are initialised
 * ONCE from the environment variables named below at the first call into the library and can be changed at run time
 * (the parity tests run every mapping; a maintainer can pin one).  A change takes effect at the next ofdis_batch_run /
 * ofdis_flow, except fused_tv and contract, which a context fixes at creation (it allocates the scratch of the path it will take:
 * ofdis_flow_cache_clear() before ofdis_flow picks up a change).  A captured launch graph (ofdis_batch_set_graph) is
 * re-captured after a change.
 * ------------------------------------------------------------------------------------------- */
typedef struct ofdis_tuning {
  int gray8;          /* 1: gray 8x8 patches use the 4-lanes-per-patch kernel; 0: generic kernel     OFDIS_NO_GRAY8 -> 0 */
  int rgb12;          /* 1: RGB 12x12, gray 12x12 and RGB 8x8 patches (flow or stereo, cost function 0 / 1) use the
                       * 16-lanes-per-patch kernels / the compile-time instantiations; 0: generic kernel   OFDIS_NO_RGB12 -> 0 */
  int rgb12_lpp;      /* lanes per RGB 12x12 patch: 0 = the library's choice (16), 64 = one patch per wavefront, 32 = two,
                       * 16 = four (a 3x3 pixel block per lane for the taps; the exact contract moves the interpolated
                       * values through LDS into the entry order its documented summation needs, the fused contract sums
                       * block-wise)                                                                  OFDIS_RGB12_LPP */
  int fused_tv;       /* 1: gray levels of <= 256 rows and <= 256 columns take the fused TV path (warp + derivatives kernel,
                       * fused system + SOR kernel), RGB levels of <= 256 rows the fused system + SOR kernels (fused_rgb_min)
                       *                                                                              OFDIS_NO_FUSED -> 0 */
  int fused_mw_max;   /* frame groups up to which the multi-wave fused TV kernels are launched       OFDIS_FUSED_MW_MAX
                       * (default 512 and at most 1024 frames per batch; 0 = never; >= 2^30 = always) */
  int fused_split;    /* 1: multi-wave kernel with producer + solver wavefronts per iteration  OFDIS_FUSED_NO_SPLIT -> 0 */
  int finish_fusion;  /* 1: the multi-wave kernels write the refined flow themselves        OFDIS_NO_FINISH_FUSION -> 0 */
  int fused_strip;    /* frames per strip of the throughput fused TV kernel; 0 = chosen by the library  OFDIS_FUSED_STRIP */
  int prep_band_rows; /* output rows per wavefront of the warp + derivatives kernel; 0 = by batch size  OFDIS_PREP_BAND_ROWS */
  int graph;          /* 1: ofdis_batch_set_graph may replay a captured launch graph                 OFDIS_NO_GRAPH -> 0 */
  int flow_dma;       /* 1: ofdis_flow stages through hipMemcpyAsync instead of copy kernels               OFDIS_FLOW_DMA */
  int flow_whole;     /* 1: ofdis_flow uploads the whole pyramid before the first launch                OFDIS_FLOW_WHOLE */
  int fused_xcu_max;  /* frame groups up to which the fused TV kernel runs every fixed-point iteration of a group as its
                       * own workgroup on its own CU (contexts of <= 768 frames; 0 = never)        OFDIS_FUSED_XCU_MAX */
  int fused_tp_pipe;  /* large batches: the fused TV kernel with one wavefront per fixed-point iteration and strip (the
                       * iterations of a strip on one compute unit share the derivative records through the L2) instead of
                       * one wavefront per strip walking all iterations.  0 = never, 1 = where it is faster (gray levels of
                       * more than 32 rows under the fused contract; RGB levels of <= 64 rows under the fused contract in
                       * batches of up to 2048 frames), 2 = always                                  OFDIS_FUSED_TP_PIPE */
  int fused_xcu_spin; /* microseconds a workgroup of that variant waits for a hand-over row (device wall clock) before it
                       * reports the pass as failed and carries on without waiting; 0 = the default, 50 000 (50 ms: a
                       * thousand times the longest healthy wait, and the most a drop-in call can lose before its pass is
                       * repeated on the other mapping); 1 forces the failure                     OFDIS_FUSED_XCU_SPIN */
  int contract;       /* ARITHMETIC CONTRACT -- the one knob that changes bits.  0 = exact (default) and 1 = fused, both
                       * stated at the top of this file: exact is bit-identical to the reference build, fused the tolerance contract of the
                       * north star (flow within 1e-3 px of the reference): every kernel compiled a second time with
                       * multiply-adds contracted to v_fma_f32 and the hardware's 1-ulp reciprocal / square root in place
                       * of the correctly rounded ones -- same algorithm, same control flow, fewer instructions.  A context
                       * fixes it at creation, like fused_tv.                                OFDIS_CONTRACT=fused -> 1 */
  int fused_xcu_drop; /* TEST HOOK, 0 in production: 1 = the first fixed-point iteration of the cross-CU variant withholds its
                       * hand-over rows, so that the iteration behind it really waits out fused_xcu_spin (what a dispatcher
                       * that broke the variant's ordering assumption would cause): tests/test_gpu_xcu.py measures what
                       * the failure protocol costs at the DEFAULT bound with it                  OFDIS_FUSED_XCU_DROP */
  int prep_densify;   /* 1: on the fused TV path (gray 8x8 patches, step-4 grid) the warp + derivatives kernel densifies the
                       * flow from the patch results itself (PatGridClass::AggregateFlowDense inside tv_prep_kernel): no
                       * densification launch, no round trip of the dense flow; 0: separate kernel  OFDIS_NO_PREP_DENSIFY -> 0 */
  int fused_tall_group; /* levels of 65 ... 96 rows (the finest level of a 1080p / 4K gray pair is 120 x 68): 1 = the fused TV
                       * kernel takes up to three strips per workgroup, their rows beyond the 64th sharing ONE wavefront
                       * (2 .. 7: at most that many); 0 = two wavefronts per strip   OFDIS_TALL_GROUP, OFDIS_NO_TALL_GROUP -> 0 */
  int fused_rgb_min;  /* RGB levels of <= 256 rows (three derivative record arrays) and gray levels of > 256 columns and <= 256
                       * rows take the fused system + SOR kernels behind the TILED warp and derivatives kernels (the latter
                       * writing records; with fused_tv and finish_fusion) in contexts of at least this many frames: 0 = the
                       * library's choice (16 under the fused contract, 512 under the exact one: below it the
                       * one-launch-per-stage kernels are as fast or faster), 1 = always, 2^30 = never.  The stereo-depth
                       * mode's levels of <= 64 rows take their fused kernel (de_fused_kernel) by the same knob; its own
                       * defaults: always under the fused contract, from 256 frames under the exact one
                       *                                                                            OFDIS_FUSED_RGB_MIN */
} ofdis_tuning;
int ofdis_get_tuning(ofdis_tuning* out);
int ofdis_set_tuning(const ofdis_tuning* in);

/* ---------------------------------------------------------------------------------------------
 * Per-function entry points (device pointers, batched over `nframes` frames, packed planes
 * [nframes][h][w]).  Each replaces one FDF1.0.1 / PatGrid function; used by the parity tests and
 * available to a maintainer who wants to swap a single stage.
 * ------------------------------------------------------------------------------------------- */
/* image_warp, opticalflow_aux.c:18-60.  src/dst: [nframes][noc][h][w]; wx, wy, mask: [nframes][h][w] */
int ofdis_image_warp(float* dst, float* mask, const float* src, const float* wx, const float* wy,
                     int w, int h, int noc, int nframes, void* stream);
/* get_derivatives, opticalflow_aux.c:65-116.  out: [nframes][8][noc][h][w] in the order
 * Ix,Iy,Iz,Ixx,Ixy,Iyy,Ixz,Iyz */
int ofdis_get_derivatives(float* out, const float* im1, const float* im2w, int w, int h, int noc,
                          int nframes, void* stream);
/* compute_smoothness + compute_data + 2x sub_laplacian fused (opticalflow_aux.c:123-199,310-438):
 * out: [nframes][7][h][w] = a11,a12,a22,b1,b2,smooth_horiz,smooth_vert */
int ofdis_tv_system(float* out, const float* mask, const float* wx, const float* wy, const float* du,
                    const float* dv, const float* derivs, float tv_alpha, float tv_gamma, float tv_delta,
                    int w, int h, int noc, int nframes, void* stream);
/* sor_coupled, solver.c:77-421: `iterations` lexicographic block-SOR sweeps; du, dv in place.
 * sys: the [nframes][7][h][w] array of ofdis_tv_system (not modified; the reference's in-place
 * block inverse, solver.c:113-120, lives in registers). */
int ofdis_sor_coupled(float* du, float* dv, const float* sys, int iterations, float omega, int w, int h,
                      int nframes, void* stream);
/* one pyramid level of PatGridClass::{InitializeGrid,SetTargetImage,InitializeFromCoarserOF,
 * Optimize} + AggregateFlowDense (patchgrid.cpp:98-141,195-275,377-397).
 * im_*: [nframes][tmp_h][tmp_w][noc]; flow_prev: [nframes][h/2][w/2][2] or NULL;
 * p_out: [nframes][nopatches][2] or NULL; flow_out: [nframes][h][w][2] */
int ofdis_patchgrid_level(const ofdis_params* p, int level, const float* im_a, const float* im_a_dx,
                          const float* im_a_dy, const float* im_b, const float* flow_prev, float* p_out,
                          float* flow_out, int nframes, void* stream);
/* one pyramid level of VarRefClass (refine_variational.cpp:25-241); flow [nframes][h][w][2] in place */
int ofdis_varref_level(const ofdis_params* p, int level, const float* im_a, const float* im_b, float* flow,
                       int nframes, void* stream);

/* plain device memory helpers so that C callers need no HIP headers */
void* ofdis_dev_alloc(size_t bytes);
void ofdis_dev_free(void* p);
int ofdis_memcpy_h2d(void* dst, const void* src, size_t bytes);
int ofdis_memcpy_d2h(void* dst, const void* src, size_t bytes);
int ofdis_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
/* Waits for `stream` (NULL = the calling thread's current device's default stream).  Also reports -- once, as
 * OFDIS_ERR_DEVICE -- a lost hand-over of the cross-CU fused TV variant (ofdis_batch_status) of every context whose last
 * pass was enqueued on that stream; for the NULL stream, which exists once per device, only the contexts of the calling
 * thread's current device. */
int ofdis_sync(void* stream);

/* ---------------------------------------------------------------------------------------------
 * Streams, pinned host memory and asynchronous copies (version 3): what a host loop needs to keep the link and the GPU
 * busy at the same time -- the reference's per-pair upload / download (run_dense.cpp:326-344 convertTo + pyramid on the
 * host, :406-421 resize + crop + .flo) becomes: pinned 8-bit frames -> ofdis_memcpy_h2d_async -> ofdis_batch_build_pyramids_u8
 * -> ofdis_batch_run -> ofdis_batch_upsample_frames -> ofdis_memcpy_d2h_async, all enqueued on ONE stream per slot, with
 * two or more slots (context + stream + buffers) in flight so that one slot's download overlaps the next slot's upload
 * and kernels (host/run_seq_main.cpp).  Several small passes in flight on several streams is also how a share of a few
 * dozen pairs per GPU (BASELINE configs[4] at 8 GPUs) keeps the chip busy: DESIGN.md 6, bench.py `small_batch.depth`.
 * ------------------------------------------------------------------------------------------- */
/* a non-blocking stream on the calling thread's current device; NULL on failure (ofdis_last_error) */
void* ofdis_stream_create(void);
void ofdis_stream_destroy(void* stream);
/* page-locked host memory, usable with every visible device (hipHostMallocPortable); NULL on failure */
void* ofdis_host_alloc(size_t bytes);
void ofdis_host_free(void* p);
/* enqueue a copy on `stream`; the host buffer must stay valid until the stream has been synchronised.  With memory from
 * ofdis_host_alloc the copy is a DMA that overlaps kernels and copies of other streams; with pageable memory it still
 * works, staged and partly synchronous, like hipMemcpyAsync */
int ofdis_memcpy_h2d_async(void* dst_dev, const void* src_host, size_t bytes, void* stream);
int ofdis_memcpy_d2h_async(void* dst_host, const void* src_dev, size_t bytes, void* stream);
/* Events order work across streams without the host.  Why a host loop wants them: uploads share the link's one direction
 * and downloads the other, so the copies of consecutive chunks belong on ONE upload stream and ONE download stream (each
 * in order) beside the compute stream(s) -- two chunks that each run upload, kernels, download on a stream of their own fall
 * into lock step (both upload, then both download) and the two directions never overlap (tools/link_probe.py: 12.2 k
 * against 15 k pairs/s).  ofdis_event_record marks a point of `stream`; ofdis_stream_wait_event makes `stream` wait for
 * the last recorded point (a never-recorded event is complete); ofdis_event_sync blocks the host until it is reached. */
void* ofdis_event_create(void);
void ofdis_event_destroy(void* event);
int ofdis_event_record(void* event, void* stream);
int ofdis_stream_wait_event(void* stream, void* event);
int ofdis_event_sync(void* event);

#ifdef __cplusplus
}
#endif
#endif /* OFDIS_H_ */
