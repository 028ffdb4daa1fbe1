// ofdis_capi.hip -- the C ABI of include/ofdis.h: parameter handling, the batch context (all HBM
// buffers of the hot path), the coarse-to-fine launch schedule of OFC::OFClass::OFClass
// (oflow.cpp:184-337) and the per-function entry points used for parity testing.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

// the library is compiled with -fvisibility=hidden: what include/ofdis.h declares is the whole export list
#pragma GCC visibility push(default)
#include "../../include/ofdis.h"
#pragma GCC visibility pop
#include "ofdis_kernels.h"

using namespace ofdis;

namespace {

// The launchers of one arithmetic contract (ofdis_kernels.h / ofdis_launchers.inc): every kernel file is compiled once per
// contract into ofdis::exact and ofdis::fused; a context fixes its contract at creation (ofdis_tuning::contract) and
// launches through its table.
struct Launchers {
  decltype(&exact::launch_patch_optimize) patch_optimize;
  decltype(&exact::patch_pixel_weights_supported) patch_pixel_weights_supported;
  decltype(&exact::launch_densify) densify;
  decltype(&exact::launch_patch_p_reference_order) patch_p_reference_order;
  decltype(&exact::launch_warp) warp;
  decltype(&exact::launch_derivatives) derivatives;
  decltype(&exact::tv_prep_supported) tv_prep_supported;
  decltype(&exact::launch_tv_prep) tv_prep;
  decltype(&exact::launch_tv_system) tv_system;
  decltype(&exact::launch_sor) sor;
  decltype(&exact::tv_fused_supported) tv_fused_supported;
  decltype(&exact::tv_fused_params_ok) tv_fused_params_ok;
  decltype(&exact::tv_fused_mode) tv_fused_mode;
  decltype(&exact::launch_tv_fused) tv_fused;
  decltype(&exact::launch_tv_finish_records) tv_finish_records;
  decltype(&exact::launch_to_diag) to_diag;
  decltype(&exact::launch_from_diag) from_diag;
  decltype(&exact::launch_tv_finish) tv_finish;
  decltype(&exact::launch_flow_split) flow_split;
  decltype(&exact::launch_de_system) de_system;
  decltype(&exact::launch_de_sor) de_sor;
  decltype(&exact::launch_de_update) de_update;
};
#define OFDIS_LAUNCHER_TABLE(ns)                                                                                        \
  {ns::launch_patch_optimize, ns::patch_pixel_weights_supported, ns::launch_densify, ns::launch_patch_p_reference_order, ns::launch_warp, ns::launch_derivatives, ns::tv_prep_supported,       \
   ns::launch_tv_prep, ns::launch_tv_system, ns::launch_sor, ns::tv_fused_supported, ns::tv_fused_params_ok,            \
   ns::tv_fused_mode, ns::launch_tv_fused, ns::launch_tv_finish_records, ns::launch_to_diag, ns::launch_from_diag,      \
   ns::launch_tv_finish, ns::launch_flow_split, ns::launch_de_system, ns::launch_de_sor, ns::launch_de_update}
const Launchers kExactLaunchers = OFDIS_LAUNCHER_TABLE(exact);
const Launchers kFusedLaunchers = OFDIS_LAUNCHER_TABLE(fused);
#undef OFDIS_LAUNCHER_TABLE
const Launchers& launchers(int contract) { return contract ? kFusedLaunchers : kExactLaunchers; }

}  // namespace

namespace ofdis {

// Kernel-selection knobs (include/ofdis.h: ofdis_tuning): read ONCE from the environment, changed only through
// ofdis_set_tuning; every launch path takes a snapshot (no getenv on the dispatch path).
namespace {
std::mutex g_tuning_mutex;
ofdis_tuning g_tuning;
bool g_tuning_init = false;
unsigned g_tuning_epoch = 0;
void tuning_init_locked() {
  if (g_tuning_init) return;
  auto on = [](const char* name) { return getenv(name) != nullptr; };
  auto num = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
  g_tuning.gray8 = !on("OFDIS_NO_GRAY8");
  g_tuning.rgb12 = !on("OFDIS_NO_RGB12");
  {
    const int l = num("OFDIS_RGB12_LPP", 0);
    g_tuning.rgb12_lpp = (l == 16 || l == 32 || l == 64) ? l : 0;
  }
  g_tuning.fused_tv = !on("OFDIS_NO_FUSED");
  g_tuning.fused_mw_max = std::max(0, num("OFDIS_FUSED_MW_MAX", 512));
  g_tuning.fused_split = !on("OFDIS_FUSED_NO_SPLIT");
  g_tuning.finish_fusion = !on("OFDIS_NO_FINISH_FUSION");
  g_tuning.fused_strip = std::max(0, num("OFDIS_FUSED_STRIP", 0));
  g_tuning.prep_band_rows = std::max(0, num("OFDIS_PREP_BAND_ROWS", 0));
  g_tuning.graph = !on("OFDIS_NO_GRAPH");
  g_tuning.flow_dma = on("OFDIS_FLOW_DMA");
  g_tuning.flow_whole = on("OFDIS_FLOW_WHOLE");
  g_tuning.fused_xcu_max = std::max(0, num("OFDIS_FUSED_XCU_MAX", 768));
  g_tuning.fused_tp_pipe = std::max(0, std::min(2, num("OFDIS_FUSED_TP_PIPE", 1)));
  g_tuning.fused_xcu_spin = std::max(0, num("OFDIS_FUSED_XCU_SPIN", 0));
  {  // arithmetic contract: "fused" / "1" = the tolerance contract, anything else (or unset) = exact
    const char* e = getenv("OFDIS_CONTRACT");
    g_tuning.contract = (e && (!strcmp(e, "fused") || !strcmp(e, "1"))) ? 1 : 0;
  }
  g_tuning_init = true;
}
}  // namespace

ofdis_tuning tuning(unsigned* epoch) {
  std::lock_guard<std::mutex> lock(g_tuning_mutex);
  tuning_init_locked();
  if (epoch) *epoch = g_tuning_epoch;
  return g_tuning;
}

}  // namespace ofdis

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
int hipfail(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return OFDIS_ERR_DEVICE;
}
#define HIPCHK(expr)                                  \
  do {                                                \
    hipError_t _e = (expr);                           \
    if (_e != hipSuccess) return hipfail(_e, #expr);  \
  } while (0)

// Level geometry exactly as the reference derives it (oflow.cpp:91-92,138-151; patchgrid.cpp:42-48).
LevelGeom make_geom(const ofdis_params& p, int sl) {
  LevelGeom g;
  memset(&g, 0, sizeof(g));
  const float sc_fct = (float)pow(2, -sl);
  g.level = sl;
  g.h = (int)(p.height * sc_fct);
  g.w = (int)(p.width * sc_fct);
  g.pad = p.imgpadding;
  g.noc = p.noc;
  g.P = p.p_samp_s;
  g.lb = -(float)p.p_samp_s / 2;
  g.ubw = (float)(g.w + p.p_samp_s / 2 - 2);
  g.ubh = (float)(g.h + p.p_samp_s / 2 - 2);
  g.tmp_w = g.w + 2 * g.pad;
  g.tmp_h = g.h + 2 * g.pad;
  int steps = (int)floor(p.p_samp_s * (1 - p.patove));
  g.steps = steps < 1 ? 1 : steps;
  g.steps_magic = g.steps > 1 ? (unsigned)((0x100000000ull + (unsigned)g.steps - 1) / (unsigned)g.steps) : 0u;
  g.novals = p.noc * p.p_samp_s * p.p_samp_s;
  g.nopw = (int)ceil((float)g.w / (float)g.steps);
  g.noph = (int)ceil((float)g.h / (float)g.steps);
  g.offw = (int)floor((g.w - (g.nopw - 1) * g.steps) / 2);
  g.offh = (int)floor((g.h - (g.noph - 1) * g.steps) / 2);
  g.nop = g.nopw * g.noph;
  g.plane_elems = (size_t)g.tmp_w * g.tmp_h * g.noc;
  return g;
}

int check_params(const ofdis_params* p) {
  if (!p) return fail(OFDIS_ERR_INVALID, "params is NULL");
  if (p->noc != 1 && p->noc != 3) return fail(OFDIS_ERR_INVALID, "noc must be 1 or 3");
  if (p->sc_l < 0 || p->sc_f < p->sc_l || p->sc_f > 20) return fail(OFDIS_ERR_INVALID, "need 0 <= sc_l <= sc_f");
  if (p->width <= 0 || p->height <= 0 || (p->width % (1 << p->sc_f)) || (p->height % (1 << p->sc_f)))
    return fail(OFDIS_ERR_INVALID, "width/height must be positive multiples of 2^sc_f (oflow.h:87)");
  if ((p->width >> p->sc_l) > 32768 || (p->height >> p->sc_l) > 32768)
    return fail(OFDIS_ERR_UNSUPPORTED, "finest level larger than 32768 pixels in one dimension");
  if (p->p_samp_s < 2 || (p->p_samp_s & 1)) return fail(OFDIS_ERR_INVALID, "p_samp_s must be even and >= 2");
  if (p->imgpadding < p->p_samp_s) return fail(OFDIS_ERR_INVALID, "imgpadding must be >= p_samp_s (oflow.cpp:147-149)");
  if (p->noc * p->p_samp_s * p->p_samp_s > 64 * 12) return fail(OFDIS_ERR_UNSUPPORTED, "patch too large (novals > 768)");
  if (p->costfct < 0 || p->costfct > 2) return fail(OFDIS_ERR_UNSUPPORTED, "costfct must be 0, 1 or 2");
  if (!(p->patove >= 0.0f && p->patove < 1.0f)) return fail(OFDIS_ERR_INVALID, "patove must be in [0,1)");
  if (p->usetvref && ((p->height >> p->sc_f) < 4))
    return fail(OFDIS_ERR_INVALID, "coarsest level must have >= 4 rows for the TV derivative filter (image.c:401-434)");
  if (p->max_iter < 0 || p->tv_innerit < 0 || p->tv_solverit < 0) return fail(OFDIS_ERR_INVALID, "negative iteration count");
  if (p->selectmode < 0 || p->selectmode > 2) return fail(OFDIS_ERR_INVALID, "selectmode must be 0/1 (optical flow) or 2 (stereo depth)");

  return OFDIS_OK;
}

double now_ms() {
  struct timeval tv;
  gettimeofday(&tv, nullptr);
  return tv.tv_sec * 1000.0 + tv.tv_usec / 1000.0;
}

struct EventPair {
  hipEvent_t a, b;
};

// Plain copy kernel for the drop-in's staging traffic: the pinned host buffer is device-accessible, so the pyramid is
// pulled (and the flow pushed) over PCIe by a kernel on the same queue as the computation instead of a DMA command with
// its cross-engine synchronisation -- worth ~8 % of a 0.5 ms call (OFDIS_FLOW_DMA=1 goes back to hipMemcpyAsync).
__global__ __launch_bounds__(256) void copy16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
hipError_t launch_copy16(void* dst, const void* src, size_t bytes, hipStream_t s) {  // bytes: multiple of 16
  const size_t n16 = bytes / 16;
  size_t blocks = (n16 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, n16);
  return hipGetLastError();
}

}  // namespace

// Cross-CU variant of the fused TV kernel (ofdis_fused_xcu.hip): a workgroup whose hand-over row never arrives (bounded
// wait) sets a word in mapped host memory.  The word belongs to the CONTEXT that launched the kernel: it is allocated with
// the context (never on a launch path), polled by every synchronising entry point on behalf of that context only, and a
// context that has seen it once never launches the variant again (its granule array is re-zeroed before the next pass).
struct XcuState {
  int* host = nullptr;             // mapped, device-visible
  int* dev = nullptr;
  // (atomics: ofdis_sync polls the contexts of its stream from whichever thread synchronises, possibly while the context's
  // own thread starts its next pass)
  std::atomic<bool> off{false};        // the variant is off for this context
  std::atomic<bool> failed{false};     // the results of the last pass are invalid (until the next pass starts)
  std::atomic<bool> told_sync{false};  // ofdis_sync has reported the failure (it does so once; status / download keep saying it)
  std::atomic<bool> missed{false};     // a pass failed and the NEXT pass was started before anybody polled: reported once, late
  std::atomic<bool> rezero{false};     // the granule array may hold stale tags
  std::atomic<hipStream_t> last_stream{nullptr};  // where the last pass was enqueued (ofdis_sync polls the contexts of its stream)
  std::atomic<int> last_device{-1};               // ... and on which device (the null stream is "the same" stream on every device)
  std::atomic<bool> ran{false};
};

struct ofdis_batch {
  ofdis_params p;
  int contract = 0;                  // arithmetic contract, fixed at creation (ofdis_tuning::contract)
  const Launchers* k = &kExactLaunchers;  // ... and its launchers
  int nframes = 0;
  int total_frames = 0;              // nframes of the owning context (a frame_view keeps it)
  int nlevels = 0;
  std::vector<LevelGeom> geom;       // index = level - sc_l
  std::vector<float*> in[6];         // A, A_dx, A_dy, B per level (+ B_dx, B_dy when usefbcon)
  std::vector<float*> flow_bw;       // usefbcon: backward dense flow per level (oflow.cpp:162)
  float* pvec_bw = nullptr;          // usefbcon: backward grid results
  float* pweight_bw = nullptr;
  std::vector<float*> flow;          // AoS dense flow per level
  int nop = 2;                       // flow channels (1 in stereo-depth mode)
  float* uu = nullptr;               // stereo TV: clamped flow + increment (row-major)
  const float* initflow = nullptr;   // borrowed device pointer (ofdis_batch_set_initflow) or null
  float* initflow_own = nullptr;     // staging buffer of ofdis_batch_upload_initflow
  // scratch, sized for the finest level
  float *pvec = nullptr, *pweight = nullptr;
  float* pixw = nullptr;             // RGB: compact per-pixel weight denominators (ofdis_dev.h: pixw_row), [B][nop][P*P]
  float *wx = nullptr, *wy = nullptr, *du = nullptr, *dv = nullptr, *mask = nullptr;
  float *w_im2 = nullptr, *derivs = nullptr, *sys = nullptr;
  float *wrec = nullptr, *uv = nullptr;  // fused TV path: the (wx, wy) and (du, dv) records; `derivs` holds the
                                         // derivative records there (ofdis_dev.h: sdiag_index)
  float* xbuf = nullptr;                 // ... and the hand-over granules of its cross-CU variant (small contexts only)
  size_t xbuf_per_frame = 0;             // floats
  struct XcuState* xcu = nullptr;        // ... with the variant's error word (owned by the context; frame views share it)
  std::vector<float*> pyr_tmp;       // unpadded level images (ofdis_batch_build_pyramids_u8), lazily allocated
  // device memory: requests are collected (dalloc) and served from ONE hipMalloc per commit (dcommit) -- a context is
  // one allocation (two with the u8 pyramid scratch), and the input planes form one contiguous region [in_base,
  // in_base + in_bytes) in (level, kind) order so that a single-frame context is uploaded with one copy (ofdis_flow)
  struct Req { float** slot; size_t bytes; };
  std::vector<Req> pending;
  std::vector<void*> allocs;
  char* in_base = nullptr;
  size_t in_bytes = 0;
  // hipGraph replay of the launch schedule (ofdis_batch_set_graph)
  int graph_mode = 0;                    // 0 off (default), 1 on, -1 captured at the second pass
  long runs = 0;                         // un-pipelined passes so far
  hipGraphExec_t graph_exec = nullptr;
  const float* graph_initflow = nullptr; // the warm-start pointer the captured graph was built with
  unsigned graph_epoch = 0;              // ... and the state of the kernel-selection knobs (ofdis_set_tuning)
  bool graph_xcu_off = false;            // ... and whether the cross-CU fused TV variant was already off for this context
  hipStream_t cap_stream = nullptr;      // capture stream
  // sub-batches on internal streams (ofdis_batch_run)
  std::vector<hipStream_t> sub_streams;  // streams of sub-batches 1..S-1 (sub-batch 0 runs on the caller's stream)
  std::vector<hipEvent_t> sub_done;
  hipEvent_t sub_start = nullptr;
  int pipeline = 1;                      // ofdis_batch_set_pipeline: number of sub-batches (1 = none)
  bool join_pending = false;             // sub-batches may still be running on the internal streams
  // timing
  bool timing = false;
  std::vector<EventPair> ev[OFDIS_K_COUNT];
  size_t ev_used[OFDIS_K_COUNT] = {0};

  const LevelGeom& g(int level) const { return geom[level - p.sc_l]; }
};

namespace {

int dalloc(ofdis_batch* b, float** ptr, size_t elems) {  // request; served by dcommit()
  *ptr = nullptr;
  const size_t bytes = ((elems ? elems : 1) * sizeof(float) + 255) & ~(size_t)255;
  b->pending.push_back({ptr, bytes});
  return OFDIS_OK;
}
int dcommit(ofdis_batch* b) {
  size_t total = 0;
  for (auto& r : b->pending) total += r.bytes;
  if (!total) return OFDIS_OK;
  void* d = nullptr;
  hipError_t e = hipMalloc(&d, total);
  if (e != hipSuccess) {
    b->pending.clear();
    return hipfail(e, "hipMalloc");
  }
  b->allocs.push_back(d);
  char* c = (char*)d;
  for (auto& r : b->pending) {
    *r.slot = (float*)c;
    c += r.bytes;
  }
  b->pending.clear();
  return OFDIS_OK;
}

struct KTimer {  // brackets one launch with events when timing is on
  ofdis_batch* b;
  int k;
  hipStream_t s;
  EventPair* ep = nullptr;
  KTimer(ofdis_batch* b_, int k_, hipStream_t s_) : b(b_), k(k_), s(s_) {
    if (!b || !b->timing) return;
    auto& v = b->ev[k];
    if (b->ev_used[k] == v.size()) {
      EventPair e;
      (void)hipEventCreate(&e.a);
      (void)hipEventCreate(&e.b);
      v.push_back(e);
    }
    ep = &v[b->ev_used[k]++];
    (void)hipEventRecord(ep->a, s);
  }
  ~KTimer() {
    if (ep) (void)hipEventRecord(ep->b, s);
  }
};

// Contexts of up to this many frames own the hand-over granule array of the cross-CU fused TV variant
constexpr int XCU_MAX_CONTEXT_FRAMES = 768;
// live contexts that own a cross-CU error word (ofdis_sync has only a stream to go by)
std::mutex g_xcu_mutex;
std::vector<ofdis_batch*> g_xcu_contexts;

int xcu_state_create(ofdis_batch* b) {  // in ofdis_batch_create / ofdis_varref_level, never on a launch path
  void* h = nullptr;
  void* d = nullptr;
  HIPCHK(hipHostMalloc(&h, sizeof(int), hipHostMallocMapped | hipHostMallocPortable));
  *(volatile int*)h = 0;
  hipError_t e = hipHostGetDevicePointer(&d, h, 0);
  if (e != hipSuccess) { (void)hipHostFree(h); return hipfail(e, "hipHostGetDevicePointer"); }
  b->xcu = new XcuState();
  b->xcu->host = (int*)h;
  b->xcu->dev = (int*)d;
  return OFDIS_OK;
}
void xcu_state_destroy(ofdis_batch* b) {
  if (!b->xcu) return;
  {
    std::lock_guard<std::mutex> lock(g_xcu_mutex);
    g_xcu_contexts.erase(std::remove(g_xcu_contexts.begin(), g_xcu_contexts.end(), b), g_xcu_contexts.end());
  }
  (void)hipHostFree(b->xcu->host);
  delete b->xcu;
  b->xcu = nullptr;
}
// After a synchronisation that covers the context's last pass: OFDIS_ERR_DEVICE when that pass's results are invalid.
// The failure stays attached to the context until its next pass starts (xcu_begin_pass), which runs without the variant.
int xcu_poll(ofdis_batch* b) {
  XcuState* x = b ? b->xcu : nullptr;
  if (!x) return OFDIS_OK;
  if (*(volatile int*)x->host) {
    *(volatile int*)x->host = 0;
    x->failed = true;
    x->off = true;
    x->rezero = true;
  }
  if (x->failed)
    return fail(OFDIS_ERR_DEVICE, "fused TV kernel (cross-CU variant): a hand-over row never arrived; the results of this pass are "
                                  "invalid -- run the context again (it no longer uses the variant)");
  if (x->missed.exchange(false))  // said once: whoever consumed the earlier pass's results without asking learns it here
    return fail(OFDIS_ERR_DEVICE, "fused TV kernel (cross-CU variant): an EARLIER pass of this context lost a hand-over row and "
                                  "was never polled; its results were invalid (the pass since then ran without the variant)");
  return OFDIS_OK;
}
int xcu_begin_pass(ofdis_batch* b, hipStream_t s) {
  XcuState* x = b->xcu;
  if (!x) return OFDIS_OK;
  if (*(volatile int*)x->host) {  // a failure nobody has polled yet: the variant goes off all the same, and the failure
    (void)xcu_poll(b);            // stays latched (`missed`) until one synchronising route has reported it
    x->missed = true;
  }
  x->failed = false;
  x->told_sync = false;
  x->last_stream = s;
  {
    int dev = -1;
    (void)hipGetDevice(&dev);
    x->last_device = dev;
  }
  x->ran = true;
  if (x->rezero && b->xbuf) {  // stale tags of the pass that failed
    // (a previous pipelined pass leaves its sub-streams unjoined on purpose; they may still be using the granules)
    if (b->join_pending) {
      for (hipEvent_t ev : b->sub_done) HIPCHK(hipStreamWaitEvent(s, ev, 0));
      b->join_pending = false;
    }
    HIPCHK(hipMemsetAsync(b->xbuf, 0, b->xbuf_per_frame * b->total_frames * sizeof(float), s));
    x->rezero = false;
  }
  return OFDIS_OK;
}

DisArgs dis_args(const ofdis_params& p, const LevelGeom& g, int nframes) {
  DisArgs a;
  memset(&a, 0, sizeof(a));
  a.g = g;
  a.nframes = nframes;
  a.max_iter = p.max_iter;
  a.min_iter = p.min_iter;
  a.costfct = p.costfct;
  a.patnorm = p.patnorm;
  a.dp_thresh_sq = p.dp_thresh * p.dp_thresh;  // oflow.cpp:88
  a.dr_thresh = p.dr_thresh;
  a.res_thresh = p.res_thresh;
  a.outlier_sq_max = outlier_sq_threshold((float)p.p_samp_s / 2);  // outlierthresh, oflow.cpp:82
  a.stereo = p.selectmode == 2;
  a.camlr = 0;  // the forward grid is the left camera (oflow.cpp:153-156)
  return a;
}

struct TvConsts {
  float quarter_alpha, half_delta_over3, half_gamma_over3;
};
TvConsts tv_consts(float alpha, float gamma, float delta) {  // refine_variational.cpp:40-42
  TvConsts c;
  c.quarter_alpha = 0.25f * alpha;
  c.half_gamma_over3 = gamma * 0.5f / 3.0f;
  c.half_delta_over3 = delta * 0.5f / 3.0f;
  return c;
}

// Frames per strip of the throughput fused TV kernel.  A strip pays the fill / drain of the skewed sweep (h steps) once
// instead of once per frame, but its wavefront runs S times as long, and a launch of few, long wavefronts ends with idle
// SIMDs: measured at 16384 pairs (level 3, ms per 4096 pairs): S = 1 / 2 / 4 / 8 -> 2.30 / 2.18 / 2.21 / 2.24
// (profiles/README.md r03_b).  Rule: the largest S in {4, 2} that divides the frame count and leaves >= 4096 wavefronts in
// the launch, else 1; ofdis_tuning::fused_strip overrides.
// `pipe`: the iteration-pipelined mapping (a workgroup of n_inner wavefronts per strip group): every wavefront pays the
// fill / drain and the lag behind its predecessor per strip, so longer strips pay off more -- the largest S in {8, 4, 2}
// that leaves >= 1024 workgroups (two rounds of what the chip holds).
int strip_length(const ofdis_batch* b, const LevelGeom& g, const ofdis_tuning& tn, bool pipe) {
  const int n = b->nframes;
  if (tn.fused_strip > 0) return (n % tn.fused_strip == 0) ? tn.fused_strip : 1;
  const int R = g.h <= 16 ? 16 : (g.h <= 32 ? 32 : 64);
  for (int S = pipe ? 8 : 4; S > 1; S >>= 1)
    if (n % S == 0 && (n / S) / (64 / R) >= (pipe ? 1024 : 4096)) return S;
  return 1;
}

// VarRefClass for one level, all frames (refine_variational.cpp:25-241).  Unfused path: wx/wy hold the dense flow (planar,
// row-major) on entry.  Fused path (gray, <= 64 rows): flow_out itself holds the densified flow (AoS) on entry.  Either
// way the refined flow is in flow_out (AoS) on return.
bool use_fused(const ofdis_batch* b, const LevelGeom& g) {
  const TvConsts c = tv_consts(b->p.tv_alpha, b->p.tv_gamma, b->p.tv_delta);
  const TvGeom t{g.w, g.h, g.noc, b->nframes};
  // (a context created with every level on the fused path owns no unfused scratch: b->wx == nullptr keeps it there)
  return b->wrec && (tuning().fused_tv || !b->wx) && b->k->tv_fused_supported(t, b->p.tv_solverit) &&
         b->k->tv_prep_supported(t) && b->k->tv_fused_params_ok(c.quarter_alpha, c.half_delta_over3, c.half_gamma_over3);
}

// `fused`: use_fused(b, g), decided ONCE per level by the caller (the densification before this call has to agree with it)
int run_varref(ofdis_batch* b, const LevelGeom& g, const float* im_a, const float* im_b, float* flow_out,
               hipStream_t s, bool fused) {
  const ofdis_params& p = b->p;
  const Launchers& K = *b->k;
  TvGeom t{g.w, g.h, g.noc, b->nframes};
  const size_t npx = (size_t)g.w * g.h;
  const int n_inner = p.tv_innerit * (g.level + 1);  // :36
  const TvConsts c = tv_consts(p.tv_alpha, p.tv_gamma, p.tv_delta);
  if (fused) {
    if (n_inner <= 0) return OFDIS_OK;  // du = dv = 0: the flow stays what it is
    const ofdis_tuning tn = tuning();
    FusedArgs fa{t, b->derivs, b->wrec, b->uv, 1, c.quarter_alpha, c.half_delta_over3, c.half_gamma_over3, p.tv_solverit,
                 p.tv_sor, n_inner, b->total_frames, tn.finish_fusion ? flow_out : nullptr, tn.fused_mw_max, tn.fused_split, 0};
    // (no error word, or a context that has seen a lost hand-over: never the cross-CU variant)
    const bool xcu_ok = b->xbuf && b->xcu && !b->xcu->off;
    const FusedXcu fx{xcu_ok ? b->xbuf : nullptr, xcu_ok ? tn.fused_xcu_max : 0, xcu_ok ? b->xcu->dev : nullptr,
                      tn.fused_xcu_spin > 0 ? (unsigned)tn.fused_xcu_spin : 0u};
    if (K.tv_fused_mode(fa, &fx) == 0) {  // not the small-batch regime: strips, on one of the two throughput mappings
      // which of the two: measured per 16384 pairs (profiles/README.md round 4), levels 3 / 4 / 5 of operating point 2:
      //   fused contract  one wavefront per strip 5.47 / 1.75 / 0.44 ms (HBM-bound: 56 B per pixel and iteration),
      //                   a wavefront per iteration, S = 8: 4.41 / 2.04 / 0.57 (issue-bound, 1/5 of the traffic)
      //   exact contract  9.06 against 11.9 ms in total (370 instead of 220 instructions per step: issue-bound either way,
      //                   and the pipelined form executes more wavefront-steps)
      // so: levels of more than 32 rows (one strip per wavefront) under the fused contract; fused_tp_pipe = 2 forces it
      fa.tp_pipe = tn.fused_tp_pipe >= 2 || (tn.fused_tp_pipe == 1 && b->contract == 1 && g.h > 32);
      fa.S = strip_length(b, g, tn, K.tv_fused_mode(fa, &fx) == 1);
    }
    {  // image_warp + get_derivatives (refine_variational.cpp:189-190): one kernel, records out
      KTimer kt(b, OFDIS_K_DERIV, s);
      PrepArgs pa{t, im_a, im_b, g.pad, g.tmp_w, g.tmp_h, flow_out, b->derivs, b->wrec, fa.S, tn.prep_band_rows};
      HIPCHK(K.tv_prep(pa, s));
    }
    bool flow_written = false;  // the multi-wave variants of the fused kernel write the refined AoS flow themselves
    {  // every fixed-point iteration of this level in one launch (du = dv = 0 on its first pass: no memset)
      KTimer kt(b, OFDIS_K_FUSED, s);
      HIPCHK(K.tv_fused(fa, s, &flow_written, &fx));
    }
    if (!flow_written) {
      KTimer kt(b, OFDIS_K_UPDATE, s);
      HIPCHK(K.tv_finish_records(t, flow_out, b->uv, fa.S, s));
    }
    return OFDIS_OK;
  }
  {
    KTimer kt(b, OFDIS_K_WARP, s);
    WarpArgs wa{t, im_b, 1, g.pad, g.tmp_w, g.tmp_h, b->wx, b->wy, b->w_im2, b->mask};
    HIPCHK(K.warp(wa, s));
  }
  {
    KTimer kt(b, OFDIS_K_DERIV, s);
    DerivArgs da{t, im_a, 1, g.pad, g.tmp_w, g.tmp_h, b->w_im2, b->derivs};
    HIPCHK(K.derivatives(da, s));
  }
  HIPCHK(hipMemsetAsync(b->du, 0, npx * b->nframes * sizeof(float), s));  // image_erase :186-187
  HIPCHK(hipMemsetAsync(b->dv, 0, npx * b->nframes * sizeof(float), s));
  for (int it = 0; it < n_inner; ++it) {
    {
      KTimer kt(b, OFDIS_K_SYSTEM, s);
      SystemArgs sa{t, b->mask, b->wx, b->wy, b->du, b->dv, b->derivs, c.quarter_alpha, c.half_delta_over3,
                    c.half_gamma_over3, b->sys};
      HIPCHK(K.tv_system(sa, s));
    }
    {
      KTimer kt(b, OFDIS_K_SOR, s);
      SorArgs so{t, b->sys, b->du, b->dv, p.tv_solverit, p.tv_sor};
      HIPCHK(K.sor(so, s));
    }
  }
  {
    KTimer kt(b, OFDIS_K_UPDATE, s);
    HIPCHK(K.tv_finish(t, b->wx, b->wy, b->du, b->dv, flow_out, s));
  }
  return OFDIS_OK;
}

// VarRefClass::RefLevelDE (refine_variational.cpp:245-336), stereo-depth mode: b->wx holds the densified horizontal
// displacement (row-major), b->wy zeros; the refined plane goes to flow_out ([B][h][w], one channel).
int run_varref_de(ofdis_batch* b, const LevelGeom& g, const float* im_a, const float* im_b, float* flow_out,
                  hipStream_t s, int camlr = 0) {
  const ofdis_params& p = b->p;
  const Launchers& K = *b->k;
  TvGeom t{g.w, g.h, g.noc, b->nframes};
  const size_t n = (size_t)g.w * g.h * b->nframes;
  const int n_inner = p.tv_innerit * (g.level + 1);
  const TvConsts c = tv_consts(p.tv_alpha, p.tv_gamma, p.tv_delta);
  {
    KTimer kt(b, OFDIS_K_WARP, s);
    WarpArgs wa{t, im_b, 1, g.pad, g.tmp_w, g.tmp_h, b->wx, b->wy, b->w_im2, b->mask};
    HIPCHK(K.warp(wa, s));
  }
  {
    KTimer kt(b, OFDIS_K_DERIV, s);
    DerivArgs da{t, im_a, 1, g.pad, g.tmp_w, g.tmp_h, b->w_im2, b->derivs};
    HIPCHK(K.derivatives(da, s));
  }
  HIPCHK(hipMemsetAsync(b->du, 0, n * sizeof(float), s));                                    // image_erase(du)
  HIPCHK(hipMemcpyAsync(b->uu, b->wx, n * sizeof(float), hipMemcpyDeviceToDevice, s));      // uu = wx (:283)
  for (int it = 0; it < n_inner; ++it) {
    {
      KTimer kt(b, OFDIS_K_SYSTEM, s);
      DeSystemArgs sa{t, b->mask, b->wx, b->uu, b->du, b->derivs, c.quarter_alpha, c.half_delta_over3,
                      c.half_gamma_over3, b->sys};
      HIPCHK(K.de_system(sa, s));
    }
    {
      KTimer kt(b, OFDIS_K_SOR, s);
      DeSorArgs so{t, b->sys, b->du, p.tv_solverit, p.tv_sor};
      HIPCHK(K.de_sor(so, s));
    }
    {
      KTimer kt(b, OFDIS_K_UPDATE, s);
      HIPCHK(K.de_update(t, b->wx, b->du, b->uu, nullptr, camlr, s));  // min / max with 0 by camera side
    }
  }
  HIPCHK(hipMemcpyAsync(flow_out, b->uu, n * sizeof(float), hipMemcpyDeviceToDevice, s));   // wx = uu (:318)
  return OFDIS_OK;
}

// same, starting from an AoS flow (the backward direction of usefbcon, whose densified flow waits in AoS form)
int run_varref_from_aos(ofdis_batch* b, const LevelGeom& g, const float* im_a, const float* im_b, float* flow,
                        hipStream_t s, bool fused) {
  TvGeom t{g.w, g.h, g.noc, b->nframes};
  if (!fused) HIPCHK(b->k->flow_split(t, flow, b->wx, b->wy, s));  // (the fused path starts from the AoS flow)
  return run_varref(b, g, im_a, im_b, flow, s, fused);
}

}  // namespace

extern "C" {

const char* ofdis_last_error(void) { return g_err.c_str(); }
int ofdis_version(void) { return OFDIS_VERSION; }

int ofdis_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int ofdis_set_device(int device) {
  HIPCHK(hipSetDevice(device));
  return OFDIS_OK;
}
int ofdis_device_pci_bus_id(int device, char* buf, int len) {
  if (!buf || len < 16) return fail(OFDIS_ERR_INVALID, "need a buffer of at least 16 bytes");
  HIPCHK(hipDeviceGetPCIBusId(buf, len, device));
  return OFDIS_OK;
}

// run_dense.cpp:225-265 (operating points) and :180-183 (AutoFirstScaleSelect)
int ofdis_params_oppoint(ofdis_params* p, int op_point, int width_org, int noc) {
  if (!p || width_org <= 0 || (noc != 1 && noc != 3)) return fail(OFDIS_ERR_INVALID, "bad arguments");
  memset(p, 0, sizeof(*p));
  p->dp_thresh = 0.05f; p->dr_thresh = 0.95f; p->res_thresh = 0.0f;
  p->usefbcon = 0; p->patnorm = 1; p->costfct = 0;
  p->tv_alpha = 10.0f; p->tv_gamma = 10.0f; p->tv_delta = 5.0f;
  p->tv_innerit = 1; p->tv_solverit = 3; p->tv_sor = 1.6f;
  p->verbosity = 2;
  p->noc = noc;
  const int fratio = 5;
  int patchsz, dl, it, tv;
  float poverl;
  switch (op_point) {
    case 1: patchsz = 8; poverl = 0.3f; dl = 2; it = 16; tv = 0; break;
    case 3: patchsz = 12; poverl = 0.75f; dl = 4; it = 16; tv = 1; break;
    case 4: patchsz = 12; poverl = 0.75f; dl = 5; it = 128; tv = 1; break;
    case 2:
    default: patchsz = 8; poverl = 0.4f; dl = 2; it = 12; tv = 1; break;
  }
  const int lv_f = std::max(0, (int)floor(log2((2.0f * (float)width_org) / ((float)fratio * (float)patchsz))));
  p->p_samp_s = patchsz;
  p->patove = poverl;
  p->sc_f = lv_f;
  p->sc_l = std::max(lv_f - dl, 0);
  p->max_iter = p->min_iter = it;
  p->usetvref = tv;
  p->imgpadding = patchsz;
  return OFDIS_OK;
}

// ------------------------------------------------------------------------------------ batch context
int ofdis_batch_create(ofdis_batch** out, const ofdis_params* p, int nframes) {
  if (!out) return fail(OFDIS_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int rc = check_params(p);
  if (rc) return rc;
  if (nframes < 1) return fail(OFDIS_ERR_INVALID, "nframes must be >= 1");
  ofdis_batch* b = new ofdis_batch();
  b->p = *p;
  b->contract = tuning().contract ? 1 : 0;
  b->k = &launchers(b->contract);
  b->nframes = nframes;
  b->total_frames = nframes;
  b->nlevels = p->sc_f - p->sc_l + 1;
  b->nop = p->selectmode == 2 ? 1 : 2;
  for (int l = p->sc_l; l <= p->sc_f; ++l) b->geom.push_back(make_geom(*p, l));
  const int nin = p->usefbcon ? 6 : 4;
  for (int k = 0; k < 6; ++k) b->in[k].assign(b->nlevels, nullptr);
  b->flow.assign(b->nlevels, nullptr);
  b->flow_bw.assign(b->nlevels, nullptr);
  if (nframes > 65535)  // launch_warp / launch_upsample_crop carry the frame in grid.z
    { delete b; return fail(OFDIS_ERR_UNSUPPORTED, "at most 65535 frames per batch context"); }
  rc = OFDIS_OK;
  for (int i = 0; i < b->nlevels && !rc; ++i)  // the input planes first: one contiguous region in (level, kind) order
    for (int k = 0; k < nin && !rc; ++k) rc = dalloc(b, &b->in[k][i], b->geom[i].plane_elems * nframes);
  for (auto& r : b->pending) b->in_bytes += r.bytes;
  for (int i = 0; i < b->nlevels && !rc; ++i) {
    const LevelGeom& g = b->geom[i];
    if (!rc) rc = dalloc(b, &b->flow[i], (size_t)g.w * g.h * b->nop * nframes);
    if (!rc && p->usefbcon && i > 0) rc = dalloc(b, &b->flow_bw[i], (size_t)g.w * g.h * 2 * nframes);
  }
  const LevelGeom& g0 = b->geom[0];  // finest level: largest of everything
  const size_t npx = (size_t)g0.w * g0.h * nframes;
  size_t nop_max = 0;
  for (auto& g : b->geom) nop_max = std::max(nop_max, (size_t)g.nop);
  if (!rc) rc = dalloc(b, &b->pvec, nop_max * 2 * nframes);
  if (!rc) rc = dalloc(b, &b->pweight, nop_max * g0.novals * nframes);
  if (!rc && p->noc == 3 && !p->usefbcon && p->selectmode != 2)  // (forward-backward merging reads the weights by another shifted rule)
    rc = dalloc(b, &b->pixw, nop_max * (size_t)g0.P * g0.P * nframes);
  if (!rc && p->usefbcon) rc = dalloc(b, &b->pvec_bw, nop_max * 2 * nframes);
  if (!rc && p->usefbcon) rc = dalloc(b, &b->pweight_bw, nop_max * g0.novals * nframes);
  if (!rc && p->usetvref) {
    // TV scratch.  The fused path (gray levels of <= 64 rows and <= 128 columns) needs the three record arrays; the planes
    // of the unfused kernels are only allocated when some level of this context can take that path (13 of the 25 floats
    // per pixel otherwise: a third of the context)
    const TvConsts tc = tv_consts(p->tv_alpha, p->tv_gamma, p->tv_delta);
    const bool may_fuse = p->noc == 1 && p->selectmode != 2 && tuning().fused_tv &&
                          b->k->tv_fused_params_ok(tc.quarter_alpha, tc.half_delta_over3, tc.half_gamma_over3);
    bool all_fused = may_fuse;
    for (auto& g : b->geom) {
      const TvGeom t{g.w, g.h, g.noc, nframes};
      all_fused = all_fused && b->k->tv_fused_supported(t, p->tv_solverit) && b->k->tv_prep_supported(t);
    }
    if (!all_fused) {
      if (!rc) rc = dalloc(b, &b->wx, npx);
      if (!rc) rc = dalloc(b, &b->wy, npx);
      if (!rc) rc = dalloc(b, &b->du, npx);
      if (!rc) rc = dalloc(b, &b->dv, npx);
      if (!rc) rc = dalloc(b, &b->mask, npx);
      if (!rc) rc = dalloc(b, &b->w_im2, npx * p->noc);
      if (!rc) rc = dalloc(b, &b->sys, npx * 7);
    }
    if (!rc) rc = dalloc(b, &b->derivs, npx * 8 * p->noc);
    if (!rc && p->selectmode == 2) rc = dalloc(b, &b->uu, npx);
    if (!rc && may_fuse) {
      rc = dalloc(b, &b->wrec, npx * 2);
      if (!rc) rc = dalloc(b, &b->uv, npx * 2);
      // cross-CU variant: {du, tag, dv, tag} per pixel and iteration boundary (ofdis_fused_xcu.hip), sized over the levels
      // that can take it (fused path, >= 2 fixed-point iterations) and only while the knob is on: 16 B x (n_inner - 1) per
      // pixel of the largest such level, 344 KB per frame at operating point 2
      if (!rc && nframes <= XCU_MAX_CONTEXT_FRAMES && tuning().fused_xcu_max > 0) {
        for (auto& g : b->geom) {
          const TvGeom t{g.w, g.h, g.noc, nframes};
          const size_t n_inner = (size_t)std::max(1, p->tv_innerit * (g.level + 1));
          if (n_inner >= 2 && b->k->tv_fused_supported(t, p->tv_solverit) && b->k->tv_prep_supported(t))
            b->xbuf_per_frame = std::max(b->xbuf_per_frame, (n_inner - 1) * g.w * g.h * 4);
        }
        if (b->xbuf_per_frame) rc = dalloc(b, &b->xbuf, b->xbuf_per_frame * nframes);
      }
    }
  }
  if (!rc) rc = dcommit(b);
  if (!rc) b->in_base = (char*)b->in[0][0];
  if (!rc && b->xbuf) {  // granule tags: 0 = not yet written (the kernel restores the zeros itself)
    hipError_t e = hipMemset(b->xbuf, 0, b->xbuf_per_frame * nframes * sizeof(float));
    if (e != hipSuccess) rc = hipfail(e, "hipMemset");
    if (!rc) rc = xcu_state_create(b);
    if (!rc) {
      std::lock_guard<std::mutex> lock(g_xcu_mutex);
      g_xcu_contexts.push_back(b);
    }
  }
  if (rc) {
    ofdis_batch_destroy(b);
    return rc == OFDIS_ERR_DEVICE ? OFDIS_ERR_NOMEM : rc;
  }
  *out = b;
  return OFDIS_OK;
}

void ofdis_batch_destroy(ofdis_batch* b) {
  if (!b) return;
  xcu_state_destroy(b);
  for (hipStream_t st : b->sub_streams) {
    (void)hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
  }
  for (hipEvent_t ev : b->sub_done) (void)hipEventDestroy(ev);
  if (b->sub_start) (void)hipEventDestroy(b->sub_start);
  if (b->graph_exec) (void)hipGraphExecDestroy(b->graph_exec);
  if (b->cap_stream) (void)hipStreamDestroy(b->cap_stream);
  for (void* d : b->allocs) (void)hipFree(d);
  for (int k = 0; k < OFDIS_K_COUNT; ++k)
    for (auto& e : b->ev[k]) {
      (void)hipEventDestroy(e.a);
      (void)hipEventDestroy(e.b);
    }
  delete b;
}

float* ofdis_batch_input(ofdis_batch* b, int level, int kind) {
  if (!b || level < b->p.sc_l || level > b->p.sc_f || kind < 0 || kind > (b->p.usefbcon ? 5 : 3)) return nullptr;
  return b->in[kind][level - b->p.sc_l];
}
size_t ofdis_batch_input_elems(const ofdis_batch* b, int level) {
  if (!b || level < b->p.sc_l || level > b->p.sc_f) return 0;
  return b->g(level).plane_elems;
}

int ofdis_batch_upload(ofdis_batch* b, int frame, const float* const* im_a, const float* const* im_a_dx,
                       const float* const* im_a_dy, const float* const* im_b, void* stream) {
  if (!b || frame < 0 || frame >= b->nframes || !im_a || !im_a_dx || !im_a_dy || !im_b)
    return fail(OFDIS_ERR_INVALID, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const float* const* src[4] = {im_a, im_a_dx, im_a_dy, im_b};
  for (int l = b->p.sc_l; l <= b->p.sc_f; ++l) {
    const size_t n = b->g(l).plane_elems;
    for (int k = 0; k < 4; ++k) {
      if (!src[k][l]) return fail(OFDIS_ERR_INVALID, "pyramid level pointer is NULL");
      HIPCHK(hipMemcpyAsync(b->in[k][l - b->p.sc_l] + (size_t)frame * n, src[k][l], n * sizeof(float),
                            hipMemcpyHostToDevice, s));
    }
  }
  return OFDIS_OK;
}

int ofdis_batch_upload_b_gradients(ofdis_batch* b, int frame, const float* const* im_b_dx, const float* const* im_b_dy,
                                   void* stream) {
  if (!b || frame < 0 || frame >= b->nframes || !im_b_dx || !im_b_dy) return fail(OFDIS_ERR_INVALID, "bad arguments");
  if (!b->p.usefbcon) return OFDIS_OK;  // never read (patch.cpp:90-97)
  const float* const* src[2] = {im_b_dx, im_b_dy};
  for (int l = b->p.sc_l; l <= b->p.sc_f; ++l) {
    const size_t n = b->g(l).plane_elems;
    for (int k = 0; k < 2; ++k) {
      if (!src[k][l]) return fail(OFDIS_ERR_INVALID, "pyramid level pointer is NULL");
      HIPCHK(hipMemcpyAsync(b->in[4 + k][l - b->p.sc_l] + (size_t)frame * n, src[k][l], n * sizeof(float),
                            hipMemcpyHostToDevice, (hipStream_t)stream));
    }
  }
  return OFDIS_OK;
}

int ofdis_batch_build_pyramids_u8(ofdis_batch* b, const uint8_t* img_a, const uint8_t* img_b, int width_org,
                                  int height_org, void* stream) {
  if (!b || !img_a || !img_b) return fail(OFDIS_ERR_INVALID, "bad arguments");
  const ofdis_params& p = b->p;
  // the padded size must be what run_dense.cpp:298-305 derives from the original size
  const int sc = 1 << p.sc_f;
  if (width_org < 1 || height_org < 1 || width_org > p.width || height_org > p.height || p.width - width_org >= sc ||
      p.height - height_org >= sc)
    return fail(OFDIS_ERR_INVALID, "params.width/height are not the 2^sc_f padding of the original size");
  // level l images need 8+2l bits, the Sobel partial sums 10+2l: exact in fp32 up to l = 7 (ofdis_pyr.hip)
  if (p.sc_f > 7) return fail(OFDIS_ERR_UNSUPPORTED, "exact fp32 pyramid needs sc_f <= 7");
  hipStream_t s = (hipStream_t)stream;
  if (b->pyr_tmp.empty()) {
    b->pyr_tmp.assign(b->nlevels, nullptr);
    for (int i = 0; i < b->nlevels; ++i) {
      const LevelGeom& g = b->geom[i];
      int rc = dalloc(b, &b->pyr_tmp[i], (size_t)g.w * g.h * g.noc * b->nframes);
      if (rc) return rc;
    }
    if (int rc = dcommit(b)) { b->pyr_tmp.clear(); return rc; }
  }
  for (int which = 0; which < 2; ++which) {
    const uint8_t* src = which ? img_b : img_a;
    HIPCHK(launch_pyr_base(src, b->pyr_tmp[0], b->nframes, width_org, height_org, p.width, p.height, p.noc, p.sc_l, s));
    for (int i = 0; i < b->nlevels; ++i) {
      const LevelGeom& g = b->geom[i];
      // the planes of level i and, in the same launch where the geometry allows, the unpadded image of level i + 1
      // (2x2 means: cv::resize(.5,.5), run_dense.cpp:150)
      float* down = i + 1 < b->nlevels ? b->pyr_tmp[i + 1] : nullptr;
      if (which == 0)
        HIPCHK(launch_pyr_planes(b->pyr_tmp[i], b->in[0][i], b->in[1][i], b->in[2][i], b->nframes, g.w, g.h, p.noc, g.pad, s, down));
      else
        HIPCHK(launch_pyr_planes(b->pyr_tmp[i], b->in[3][i], b->in[4][i], b->in[5][i], b->nframes, g.w, g.h, p.noc, g.pad, s,
                                 down));  // B's gradients only exist (non-null) with usefbcon
    }
  }
  return OFDIS_OK;
}

// The coarse-to-fine loop of OFClass::OFClass (oflow.cpp:184-337), every stage batched over frames.
}  // extern "C"

namespace {

// A contiguous range of a batch's frames as a batch of its own: input / output arrays are offset per level, every
// scratch array gets the matching share of the parent's allocation (scratch is sized per frame for the finest
// level, so the shares never overlap).  The view owns nothing.
ofdis_batch frame_view(const ofdis_batch& b, int f0, int n) {
  ofdis_batch v = b;
  v.allocs.clear();
  v.sub_streams.clear();
  v.sub_done.clear();
  v.pyr_tmp.clear();
  v.timing = false;
  v.nframes = n;
  for (int i = 0; i < b.nlevels; ++i) {
    const LevelGeom& g = b.geom[i];
    for (int k = 0; k < 6; ++k)
      if (v.in[k][i]) v.in[k][i] += (size_t)f0 * g.plane_elems;
    v.flow[i] += (size_t)f0 * g.w * g.h * b.nop;
    if (v.flow_bw[i]) v.flow_bw[i] += (size_t)f0 * g.w * g.h * b.nop;
  }
  if (v.initflow) v.initflow += (size_t)f0 * ofdis_batch_initflow_elems(&b);
  const LevelGeom& g0 = b.geom[0];
  const size_t npx = (size_t)g0.w * g0.h;
  size_t nop_max = 0;
  for (auto& g : b.geom) nop_max = std::max(nop_max, (size_t)g.nop);
  auto off = [&](float*& ptr, size_t per_frame) { if (ptr) ptr += (size_t)f0 * per_frame; };
  off(v.pvec, nop_max * 2); off(v.pweight, nop_max * g0.novals); off(v.pixw, nop_max * (size_t)g0.P * g0.P);
  off(v.pvec_bw, nop_max * 2); off(v.pweight_bw, nop_max * g0.novals);
  off(v.wx, npx); off(v.wy, npx); off(v.du, npx); off(v.dv, npx); off(v.mask, npx); off(v.uu, npx);
  off(v.w_im2, npx * b.p.noc); off(v.derivs, npx * 8 * b.p.noc); off(v.sys, npx * 7);
  off(v.wrec, npx * 2); off(v.uv, npx * 2); off(v.xbuf, b.xbuf_per_frame);
  return v;
}

int run_levels(ofdis_batch* b, hipStream_t s);
int run_one_level(ofdis_batch* b, int sl, hipStream_t s);
int run_graph_or_levels(ofdis_batch* b, hipStream_t s);

}  // namespace

extern "C" {

// Pipelined mode (ofdis_batch_set_pipeline(b, S), S = 2..4): the batch is cut into S sub-batches; sub-batch 0 runs on
// the caller's stream, the others on internal streams that are forked from the caller's stream by an event but NOT
// joined back at the end of the call.  Consecutive calls then drift apart by up to one pass, so the coarse levels of
// one sub-batch (one or two wavefronts per SIMD, latency bound) overlap with the fine levels of another instead of
// leaving issue slots idle: +6 % at 4096 frames.  (Joining inside every call keeps the sub-batches in lock step and
// loses the effect -- measured.)  The price is an explicit join: results are complete on `stream` only after
// ofdis_batch_join(b, stream); download / upsample join by themselves.  Frames are independent, results unaffected.
int ofdis_batch_set_pipeline(ofdis_batch* b, int sub_batches) {
  if (!b || sub_batches < 0 || sub_batches > 4) return fail(OFDIS_ERR_INVALID, "sub_batches must be 0..4");
  if (b->join_pending) HIPCHK(hipDeviceSynchronize());
  b->join_pending = false;
  b->pipeline = sub_batches < 1 ? 1 : sub_batches;
  return OFDIS_OK;
}

int ofdis_batch_join(ofdis_batch* b, void* stream) {
  if (!b) return fail(OFDIS_ERR_INVALID, "batch is NULL");
  if (!b->join_pending) return OFDIS_OK;
  for (hipEvent_t ev : b->sub_done) HIPCHK(hipStreamWaitEvent((hipStream_t)stream, ev, 0));  // never-recorded events are complete
  b->join_pending = false;
  return OFDIS_OK;
}

int ofdis_batch_set_graph(ofdis_batch* b, int mode) {
  if (!b || mode < -1 || mode > 1) return fail(OFDIS_ERR_INVALID, "mode must be -1 (auto), 0 (off) or 1 (on)");
  b->graph_mode = mode;
  return OFDIS_OK;
}

int ofdis_batch_run(ofdis_batch* b, void* stream) {
  if (!b) return fail(OFDIS_ERR_INVALID, "batch is NULL");
  hipStream_t s = (hipStream_t)stream;
  if (int rc = xcu_begin_pass(b, s)) return rc;
  int S = (b->timing || b->p.verbosity != 0) ? 1 : b->pipeline;
  if (b->nframes < 2 * S) S = 1;
  if (S == 1) {
    int rc = ofdis_batch_join(b, stream);  // a previous pipelined pass may still be running
    if (rc) return rc;
    return run_graph_or_levels(b, s);
  }
  if (!b->sub_start) HIPCHK(hipEventCreateWithFlags(&b->sub_start, hipEventDisableTiming));
  while ((int)b->sub_streams.size() < S - 1) {
    hipStream_t st;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    b->sub_streams.push_back(st);
  }
  while ((int)b->sub_done.size() < S) {  // one per sub-batch, the caller's stream included: a join may happen on
    hipEvent_t ev;                        // another stream than the run
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    b->sub_done.push_back(ev);
  }
  HIPCHK(hipEventRecord(b->sub_start, s));  // fork: the internal streams see everything enqueued on `s` so far
  b->join_pending = true;                   // from here on sub-streams may carry work, whatever happens below
  const int per = (b->nframes + S - 1) / S;
  int rc = OFDIS_OK;
  for (int k = S - 1; k >= 0 && !rc; --k) {  // sub-batch 0 last, on the caller's stream
    const int f0 = k * per, n = std::min(per, b->nframes - f0);
    hipStream_t sk = k ? b->sub_streams[k - 1] : s;
    if (n > 0) {
      ofdis_batch v = frame_view(*b, f0, n);
      if (k) HIPCHK(hipStreamWaitEvent(sk, b->sub_start, 0));
      rc = run_levels(&v, sk);
    }
    if (!rc) HIPCHK(hipEventRecord(b->sub_done[k], sk));
  }
  return rc;
}

}  // extern "C"

namespace {

int run_one_level(ofdis_batch* b, int sl, hipStream_t s);

int run_levels(ofdis_batch* b, hipStream_t s) {
  const ofdis_params& p = b->p;
  const int verbose = p.verbosity;
  double t_all0 = 0;
  if (verbose > 0) {
    (void)hipStreamSynchronize(s);
    t_all0 = now_ms();
  }
  if (verbose > 1) printf("TIME (Grid Memo. Alloc. ) (ms): %3g\n", 0.0);  // buffers live in the batch context
  for (int sl = p.sc_f; sl >= p.sc_l; --sl)
    if (int rc = run_one_level(b, sl, s)) return rc;
  if (verbose > 0) {
    (void)hipStreamSynchronize(s);
    printf("TIME (O.Flow Run-Time   ) (ms): %3g\n", now_ms() - t_all0);
    fflush(stdout);
  }
  return OFDIS_OK;
}

// one pyramid level of the loop (the body of oflow.cpp:184-337)
int run_one_level(ofdis_batch* b, int sl, hipStream_t s) {
  const ofdis_params& p = b->p;
  const int verbose = p.verbosity;
  {
    const int ii = sl - p.sc_l;
    const LevelGeom& g = b->geom[ii];
    double tt[5] = {0, 0, 0, 0, 0};
    double t0 = 0;
    if (verbose > 1) { (void)hipStreamSynchronize(s); t0 = now_ms(); }
    // steps 1-3: patch grid construction, initialisation from the coarser flow and the inverse
    // search run as ONE kernel (pconst/pinit are reported as 0, poptim carries the time)
    const bool fb = p.usefbcon != 0;
    const bool bw_flow = fb && sl > p.sc_l;  // the backward flow is not needed at the last scale (oflow.cpp:269,291)
    // one snapshot of the TV path per level: densification and refinement must take the same one even if another thread
    // changes the knobs in between
    const bool fused = p.usetvref && p.selectmode != 2 && use_fused(b, g);
    const float* pixw_used = nullptr;
    {
      KTimer kt(b, OFDIS_K_PATCH, s);
      DisArgs a = dis_args(p, g, b->nframes);
      a.im_a = b->in[0][ii];
      a.im_a_dx = b->in[1][ii];
      a.im_a_dy = b->in[2][ii];
      a.im_b = b->in[3][ii];
      a.flow_prev = (sl < p.sc_f) ? b->flow[ii + 1] : b->initflow;  // oflow.cpp:209-220
      a.p_out = b->pvec;
      a.pweight = b->pweight;
      // RGB 12x12 without forward-backward merging: the patches the densification reads unshifted store one float per pixel
      // (the denominator of its weight) instead of three |r| -- decided here, once, for the patch kernel AND the densification
      a.pixw = (b->pixw && !fb && b->k->patch_pixel_weights_supported(a)) ? b->pixw : nullptr;
      pixw_used = a.pixw;
      HIPCHK(b->k->patch_optimize(a, s));
      a.pixw = nullptr;
      if (fb) {  // the backward grid: images swapped (oflow.cpp:193-197,214-215,234-235)
        a.im_a = b->in[3][ii];
        a.im_a_dx = b->in[4][ii];
        a.im_a_dy = b->in[5][ii];
        a.im_b = b->in[0][ii];
        a.flow_prev = (sl < p.sc_f) ? b->flow_bw[ii + 1] : nullptr;
        a.p_out = b->pvec_bw;
        a.pweight = b->pweight_bw;
        a.camlr = 1;  // the backward grid is the right camera: displacement >= 0 (oflow.cpp:155-156, patch.cpp:191-192)
        HIPCHK(b->k->patch_optimize(a, s));
      }
    }
    if (verbose > 1) { (void)hipStreamSynchronize(s); tt[2] = now_ms() - t0; t0 = now_ms(); }
    // step 4: densification (with usefbcon each direction also merges the other grid's negated flow).
    // (Doing it inside the warp kernel -- one launch and one flow round trip less -- was measured: same time, 2.4x
    // the HBM traffic because a 32x32 pixel tile re-fetches the weight lines of the patches it shares with its
    // neighbours; not kept.)
    for (int dir = 0; dir < (bw_flow ? 2 : 1); ++dir) {
      DensifyArgs d;
      memset(&d, 0, sizeof(d));
      d.g = g;
      d.nframes = b->nframes;
      d.p = dir ? b->pvec_bw : b->pvec;
      d.pweight = dir ? b->pweight_bw : b->pweight;
      d.pixw = dir ? nullptr : pixw_used;
      d.stereo = p.selectmode == 2;
      if (fb) {
        d.cg_p = dir ? b->pvec : b->pvec_bw;
        d.cg_pweight = dir ? b->pweight : b->pweight_bw;
      }
      if (p.usetvref && dir == 0) {
        if (fused) d.flow_aos = b->flow[ii];  // the fused TV path refines the AoS flow in place
        else { d.wx = b->wx; d.wy = b->wy; }
      } else if (p.usetvref) {  // backward flow: parked as AoS until the forward refinement has used the planes
        d.flow_aos = b->flow_bw[ii];
      } else {
        d.flow_aos = dir ? b->flow_bw[ii] : b->flow[ii];
      }
      KTimer kt(b, OFDIS_K_DENSIFY, s);
      HIPCHK(b->k->densify(d, s));
    }
    if (verbose > 1) { (void)hipStreamSynchronize(s); tt[3] = now_ms() - t0; t0 = now_ms(); }
    // step 5: variational refinement
    if (p.usetvref && p.selectmode == 2) {
      int rc = run_varref_de(b, g, b->in[0][ii], b->in[3][ii], b->flow[ii], s);
      if (rc) return rc;
      if (bw_flow) {  // backward direction: right camera, its densified plane waits in flow_bw (one channel)
        const size_t n = (size_t)g.w * g.h * b->nframes;
        HIPCHK(hipMemcpyAsync(b->wx, b->flow_bw[ii], n * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemsetAsync(b->wy, 0, n * sizeof(float), s));
        rc = run_varref_de(b, g, b->in[3][ii], b->in[0][ii], b->flow_bw[ii], s, 1);
        if (rc) return rc;
      }
    } else if (p.usetvref) {
      int rc = run_varref(b, g, b->in[0][ii], b->in[3][ii], b->flow[ii], s, fused);
      if (rc) return rc;
      if (bw_flow) {  // VarRefClass on the swapped pair (oflow.cpp:291-294)
        rc = run_varref_from_aos(b, g, b->in[3][ii], b->in[0][ii], b->flow_bw[ii], s, fused);
        if (rc) return rc;
      }
    }
    if (verbose > 1) {
      (void)hipStreamSynchronize(s);
      tt[4] = now_ms() - t0;
      printf("TIME (Sc: %i, #p:%6i, pconst, pinit, poptim, cflow, tvopt, total): %8.2f %8.2f %8.2f %8.2f %8.2f -> %8.2f ms.\n",
             sl, g.nop, tt[0], tt[1], tt[2], tt[3], tt[4], tt[0] + tt[1] + tt[2] + tt[3] + tt[4]);
    }
  }
  return OFDIS_OK;
}

// The launch schedule of a context is fixed (same kernels, same pointers every pass), so it can be replayed as ONE
// hipGraph launch instead of ~15 kernel launches (ofdis_batch_set_graph).  Measured on this stack it buys nothing: the
// direct launches are asynchronous and overlap the execution of the first kernels (64 pairs per pass: 0.491 ms replayed,
// 0.486 ms direct; one pair: 0.459 both), so the default is off.  Never used when timing or TIME lines are requested
// (they synchronise between stages), in pipelined mode (the sub-batches are deliberately not joined), with
// OFDIS_NO_GRAPH, or after a capture failure -- the direct launches are always the fallback.
int run_graph_or_levels(ofdis_batch* b, hipStream_t s) {
  unsigned epoch = 0;
  const bool env_off = !tuning(&epoch).graph;
  const bool want = b->graph_mode != 0 && !env_off && !b->timing && b->p.verbosity == 0 && (b->graph_mode == 1 || b->runs >= 1);
  b->runs++;
  if (!want) return run_levels(b, s);
  // the kernel selection is baked into the capture: the knobs (epoch), the warm-start pointer, and -- per context -- whether
  // the cross-CU fused TV variant is still allowed (a context that has seen a lost hand-over must not replay a graph that
  // still contains tv_fused_xcu_kernel: "run again" has to run the other kernel)
  const bool xcu_off = b->xcu && b->xcu->off;
  if (b->graph_exec && (b->graph_initflow != b->initflow || b->graph_epoch != epoch || b->graph_xcu_off != xcu_off)) {
    (void)hipGraphExecDestroy(b->graph_exec);
    b->graph_exec = nullptr;
  }
  if (!b->graph_exec) {
    if (!b->cap_stream && hipStreamCreateWithFlags(&b->cap_stream, hipStreamNonBlocking) != hipSuccess) {
      b->graph_mode = 0;
      return run_levels(b, s);
    }
    hipGraph_t g = nullptr;
    bool ok = hipStreamBeginCapture(b->cap_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
      const int rc = run_levels(b, b->cap_stream);
      const hipError_t e = hipStreamEndCapture(b->cap_stream, &g);
      ok = rc == OFDIS_OK && e == hipSuccess && g != nullptr;
    }
    if (ok) ok = hipGraphInstantiate(&b->graph_exec, g, nullptr, nullptr, 0) == hipSuccess;
    if (g) (void)hipGraphDestroy(g);
    if (!ok) {
      (void)hipGetLastError();
      b->graph_exec = nullptr;
      b->graph_mode = 0;
      return run_levels(b, s);
    }
    b->graph_initflow = b->initflow;
    b->graph_epoch = epoch;
    b->graph_xcu_off = xcu_off;
  }
  HIPCHK(hipGraphLaunch(b->graph_exec, s));
  return OFDIS_OK;
}

}  // namespace

extern "C" {

const float* ofdis_batch_flow(const ofdis_batch* b) { return b ? b->flow[0] : nullptr; }
const float* ofdis_batch_level_flow(const ofdis_batch* b, int level) {
  if (!b || level < b->p.sc_l || level > b->p.sc_f) return nullptr;
  return b->flow[level - b->p.sc_l];
}

int ofdis_batch_download(ofdis_batch* b, int frame, float* outflow_host, void* stream) {
  if (!b || frame < 0 || frame >= b->nframes || !outflow_host) return fail(OFDIS_ERR_INVALID, "bad arguments");
  const LevelGeom& g = b->geom[0];
  const size_t n = (size_t)g.w * g.h * b->nop;
  if (int rc = ofdis_batch_join(b, stream)) return rc;
  HIPCHK(hipMemcpyAsync(outflow_host, b->flow[0] + (size_t)frame * n, n * sizeof(float), hipMemcpyDeviceToHost,
                        (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return xcu_poll(b);
}

// Warm start (oflow.cpp:217-220): the coarsest level initialises its patches from this flow exactly as finer levels
// do from the level above, i.e. it is indexed as a (w >> (sc_f+1)) x (h >> (sc_f+1)) AoS plane per frame.
size_t ofdis_batch_initflow_elems(const ofdis_batch* b) {
  if (!b) return 0;
  const LevelGeom& g = b->geom[b->nlevels - 1];
  return (size_t)(g.w / 2) * (g.h / 2) * b->nop;
}

int ofdis_batch_set_initflow(ofdis_batch* b, const float* initflow_dev) {
  if (!b) return fail(OFDIS_ERR_INVALID, "batch is NULL");
  b->initflow = initflow_dev;
  return OFDIS_OK;
}

int ofdis_batch_upload_initflow(ofdis_batch* b, int frame, const float* initflow_host, void* stream) {
  if (!b || frame < 0 || frame >= b->nframes || !initflow_host) return fail(OFDIS_ERR_INVALID, "bad arguments");
  const size_t n = ofdis_batch_initflow_elems(b);
  if (!b->initflow_own) {
    int rc = dalloc(b, &b->initflow_own, n * b->nframes);
    if (!rc) rc = dcommit(b);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(b->initflow_own, 0, n * b->nframes * sizeof(float), (hipStream_t)stream));
  }
  HIPCHK(hipMemcpyAsync(b->initflow_own + (size_t)frame * n, initflow_host, n * sizeof(float), hipMemcpyHostToDevice,
                        (hipStream_t)stream));
  b->initflow = b->initflow_own;
  return OFDIS_OK;
}

int ofdis_batch_upsample_frames(ofdis_batch* b, int first_frame, int count, float* out_dev, int width_org,
                                int height_org, void* stream) {
  if (!b || !out_dev) return fail(OFDIS_ERR_INVALID, "bad arguments");
  if (first_frame < 0 || count < 1 || first_frame > b->nframes - count) return fail(OFDIS_ERR_INVALID, "frame range outside the batch");
  const ofdis_params& p = b->p;
  if (width_org < 1 || height_org < 1 || width_org > p.width || height_org > p.height)
    return fail(OFDIS_ERR_INVALID, "original size exceeds the padded size");
  const LevelGeom& g = b->geom[0];
  if (int rc = ofdis_batch_join(b, stream)) return rc;
  HIPCHK(launch_upsample_crop(b->flow[0] + (size_t)first_frame * g.w * g.h * b->nop, out_dev, count, g.w, g.h, p.sc_l,
                              (p.width - width_org) / 2, (p.height - height_org) / 2, width_org, height_org, b->nop,
                              (hipStream_t)stream));
  return OFDIS_OK;
}

int ofdis_batch_upsample(ofdis_batch* b, float* out_dev, int width_org, int height_org, void* stream) {
  if (!b) return fail(OFDIS_ERR_INVALID, "bad arguments");
  return ofdis_batch_upsample_frames(b, 0, b->nframes, out_dev, width_org, height_org, stream);
}

int ofdis_batch_timing(ofdis_batch* b, int enable) {
  if (!b) return fail(OFDIS_ERR_INVALID, "batch is NULL");
  b->timing = enable != 0;
  for (int k = 0; k < OFDIS_K_COUNT; ++k) b->ev_used[k] = 0;
  return OFDIS_OK;
}

int ofdis_batch_kernel_time(ofdis_batch* b, int k, double* ms_sum, long* launches) {
  if (!b || k < 0 || k >= OFDIS_K_COUNT) return fail(OFDIS_ERR_INVALID, "bad arguments");
  double sum = 0;
  for (size_t i = 0; i < b->ev_used[k]; ++i) {
    HIPCHK(hipEventSynchronize(b->ev[k][i].b));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, b->ev[k][i].a, b->ev[k][i].b));
    sum += ms;
  }
  if (ms_sum) *ms_sum = sum;
  if (launches) *launches = (long)b->ev_used[k];
  return OFDIS_OK;
}

int ofdis_batch_kernel_times(ofdis_batch* b, int k, double* ms_out, int capacity, int* launches) {
  if (!b || k < 0 || k >= OFDIS_K_COUNT || capacity < 0 || (capacity > 0 && !ms_out)) return fail(OFDIS_ERR_INVALID, "bad arguments");
  for (size_t i = 0; i < b->ev_used[k] && (int)i < capacity; ++i) {
    HIPCHK(hipEventSynchronize(b->ev[k][i].b));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, b->ev[k][i].a, b->ev[k][i].b));
    ms_out[i] = ms;
  }
  if (launches) *launches = (int)b->ev_used[k];
  return OFDIS_OK;
}

// ------------------------------------------------------------------------------------ drop-in
}  // extern "C"

namespace {

// The reference constructs one OFClass per frame pair (run_dense.cpp:391-400) and a video loop calls it again and again
// with the same parameters.  Creating a device context per call would cost more than the computation, so ofdis_flow keeps
// the contexts of the last few (parameter set, device) combinations: device buffers in one allocation, a stream and pinned
// staging for the pyramid upload and the flow download.  A context serves one call at a time (its own mutex: calls with the
// same parameters on the same device are serialised, like the reference's synchronous constructor); the global mutex only
// guards the cache's lookup, insertion and eviction, so calls with different parameters or on different devices run
// concurrently.  A context that is evicted or cleared while a call is using it is released when that call returns.
struct FlowCtx {
  ofdis_params p;
  int device = -1;
  unsigned epoch = 0;     // state of the kernel-selection knobs the context was created under (ofdis_set_tuning)
  ofdis_batch* b = nullptr;
  hipStream_t s = nullptr;
  char* stage = nullptr;  // pinned: [in_bytes of input planes][flow]
  size_t flow_bytes = 0;
  unsigned long stamp = 0;
  std::mutex busy;        // held for the duration of a call
  ~FlowCtx() {
    if (s) (void)hipStreamSynchronize(s);
    if (b) ofdis_batch_destroy(b);
    if (stage) (void)hipHostFree(stage);
    if (s) (void)hipStreamDestroy(s);
  }
};
std::mutex g_flow_mutex;
std::vector<std::shared_ptr<FlowCtx>> g_flow_cache;
unsigned long g_flow_stamp = 0;
constexpr size_t kFlowCacheEntries = 4;

void flow_ctx_evict(const std::shared_ptr<FlowCtx>& c) {  // (released when its last user is done)
  std::lock_guard<std::mutex> lock(g_flow_mutex);
  for (size_t i = 0; i < g_flow_cache.size(); ++i)
    if (g_flow_cache[i] == c) {
      g_flow_cache.erase(g_flow_cache.begin() + i);
      break;
    }
}

int flow_ctx_get(const ofdis_params* p, std::shared_ptr<FlowCtx>* out) {
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  unsigned epoch = 0;
  (void)ofdis::tuning(&epoch);
  std::lock_guard<std::mutex> lock(g_flow_mutex);
  for (size_t i = 0; i < g_flow_cache.size(); ++i) {
    auto& c = g_flow_cache[i];
    if (c->device == dev && memcmp(&c->p, p, sizeof(*p)) == 0) {
      if (c->epoch != epoch) {  // created under other knob settings (it owns that path's scratch): replace it
        g_flow_cache.erase(g_flow_cache.begin() + i);
        break;
      }
      c->stamp = ++g_flow_stamp;
      *out = c;
      return OFDIS_OK;
    }
  }
  auto c = std::make_shared<FlowCtx>();
  int rc = ofdis_batch_create(&c->b, p, 1);
  if (rc) return rc;
  c->p = *p;
  c->device = dev;
  c->epoch = epoch;
  const LevelGeom& g0 = c->b->geom[0];
  c->flow_bytes = (size_t)g0.w * g0.h * c->b->nop * sizeof(float);
  hipError_t e = hipStreamCreateWithFlags(&c->s, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->stage, c->b->in_bytes + c->flow_bytes, hipHostMallocDefault);
  if (e != hipSuccess) return hipfail(e, "ofdis_flow context");
  if (g_flow_cache.size() >= kFlowCacheEntries) {  // evict the least recently used
    size_t lru = 0;
    for (size_t i = 1; i < g_flow_cache.size(); ++i)
      if (g_flow_cache[i]->stamp < g_flow_cache[lru]->stamp) lru = i;
    g_flow_cache.erase(g_flow_cache.begin() + lru);
  }
  c->stamp = ++g_flow_stamp;
  g_flow_cache.push_back(c);
  *out = c;
  return OFDIS_OK;
}

}  // namespace

namespace {
int flow_with_ctx(FlowCtx& ctx, const ofdis_params* p, const float* const* im_a, const float* const* im_a_dx,
                  const float* const* im_a_dy, const float* const* im_b, const float* const* im_b_dx,
                  const float* const* im_b_dy, float* outflow, const float* initflow) {
  int rc = OFDIS_OK;
  FlowCtx* c = &ctx;
  ofdis_batch* b = c->b;
  // The pyramid goes through pinned staging (the planes mirror the device layout), coarsest level first: a level is
  // copied into the staging buffer, pulled to the device by a copy kernel and its kernels are launched; while the GPU
  // works on it the host stages the next finer -- larger -- level (measured per 1024x436 pair: whole pyramid staged up
  // front + DMA + graph replay 0.507 ms, with kernel copies 0.495, with direct launches 0.488-0.500, level by level
  // 0.484-0.494).  The launches are direct: replaying a graph per call is not faster than 15 asynchronous launches that
  // overlap the execution.
  // With verbosity > 0 the whole pyramid is uploaded first so that the TIME lines measure computation only.
  const float* const* src[6] = {im_a, im_a_dx, im_a_dy, im_b, im_b_dx, im_b_dy};
  const int nin = p->usefbcon ? 6 : 4;
  for (int l = p->sc_l; l <= p->sc_f; ++l)
    for (int k = 0; k < nin; ++k)
      if (!src[k][l]) return fail(OFDIS_ERR_INVALID, "pyramid level pointer is NULL");
  if (initflow) {
    rc = ofdis_batch_upload_initflow(b, 0, initflow, c->s);
    if (!rc) HIPCHK(hipStreamSynchronize(c->s));  // the caller's array is pageable and may go away after the call
    if (rc) return rc;
  } else {
    b->initflow = nullptr;  // a previous call on this context may have warm-started
  }
  const ofdis_tuning tn = tuning();
  const bool use_dma = tn.flow_dma != 0;
  auto stage_level = [&](int l) -> int {
    const int i = l - p->sc_l;
    for (int k = 0; k < nin; ++k)
      memcpy(c->stage + ((char*)b->in[k][i] - b->in_base), src[k][l], b->g(l).plane_elems * sizeof(float));
    // the planes of one level are consecutive allocations: [in[0][i], in[nin-1][i] + its padded size)
    char* lo = (char*)b->in[0][i];
    const size_t last = (b->g(l).plane_elems * sizeof(float) + 255) & ~(size_t)255;
    const size_t bytes = (size_t)((char*)b->in[nin - 1][i] - lo) + last;
    if (use_dma) HIPCHK(hipMemcpyAsync(lo, c->stage + (lo - b->in_base), bytes, hipMemcpyHostToDevice, c->s));
    else HIPCHK(launch_copy16(lo, c->stage + (lo - b->in_base), bytes, c->s));
    return OFDIS_OK;
  };
  if ((rc = xcu_begin_pass(b, c->s))) return rc;
  if (p->verbosity == 0 && !tn.flow_whole) {
    for (int l = p->sc_f; l >= p->sc_l && !rc; --l) {
      rc = stage_level(l);
      if (!rc) rc = run_one_level(b, l, c->s);
    }
  } else {
    for (int l = p->sc_f; l >= p->sc_l && !rc; --l) rc = stage_level(l);
    if (!rc) rc = ofdis_batch_run(b, c->s);
  }
  if (rc) return rc;
  char* out_stage = c->stage + b->in_bytes;
  if (use_dma || (c->flow_bytes & 15)) HIPCHK(hipMemcpyAsync(out_stage, b->flow[0], c->flow_bytes, hipMemcpyDeviceToHost, c->s));
  else HIPCHK(launch_copy16(out_stage, b->flow[0], c->flow_bytes, c->s));
  HIPCHK(hipStreamSynchronize(c->s));
  if (xcu_poll(b) != OFDIS_OK) {
    // a hand-over of the cross-CU fused TV variant was lost (bounded wait): the pyramid is still resident, so the pass is
    // simply repeated -- this context no longer launches the variant -- and the caller gets the right flow, only later
    if ((rc = ofdis_batch_run(b, c->s))) return rc;
    if (use_dma || (c->flow_bytes & 15)) HIPCHK(hipMemcpyAsync(out_stage, b->flow[0], c->flow_bytes, hipMemcpyDeviceToHost, c->s));
    else HIPCHK(launch_copy16(out_stage, b->flow[0], c->flow_bytes, c->s));
    HIPCHK(hipStreamSynchronize(c->s));
    if ((rc = xcu_poll(b))) return rc;
  }
  memcpy(outflow, out_stage, c->flow_bytes);
  return OFDIS_OK;
}
}  // namespace

extern "C" {

void ofdis_flow_cache_clear(void) {
  std::lock_guard<std::mutex> lock(g_flow_mutex);
  g_flow_cache.clear();  // (a context in use by another thread is released when that call returns)
}

int ofdis_flow(const ofdis_params* p, const float* const* im_a, const float* const* im_a_dx,
               const float* const* im_a_dy, const float* const* im_b, const float* const* im_b_dx,
               const float* const* im_b_dy, float* outflow, const float* initflow) {
  if (!outflow) return fail(OFDIS_ERR_INVALID, "outflow is NULL");
  int rc = check_params(p);
  if (rc) return rc;
  if (!im_a || !im_a_dx || !im_a_dy || !im_b) return fail(OFDIS_ERR_INVALID, "pyramid array is NULL");
  if (p->usefbcon && (!im_b_dx || !im_b_dy))  // otherwise never read
    return fail(OFDIS_ERR_INVALID, "usefbcon needs the gradient pyramids of the second image");
  std::shared_ptr<FlowCtx> c;
  rc = flow_ctx_get(p, &c);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(c->busy);
  rc = flow_with_ctx(*c, p, im_a, im_a_dx, im_a_dy, im_b, im_b_dx, im_b_dy, outflow, initflow);
  if (rc) {
    // a failed call may leave copy kernels reading the staging buffer and a context in an unknown state: drain its
    // stream and drop it from the cache (the next call builds a fresh one)
    (void)hipStreamSynchronize(c->s);
    flow_ctx_evict(c);
  }
  return rc;
}

// ------------------------------------------------------------------------------------ per-function
int ofdis_image_warp(float* dst, float* mask, const float* src, const float* wx, const float* wy, int w, int h,
                     int noc, int nframes, void* stream) {
  if (!dst || !mask || !src || !wx || !wy || w < 1 || h < 1 || nframes < 1) return fail(OFDIS_ERR_INVALID, "bad arguments");
  WarpArgs a{TvGeom{w, h, noc, nframes}, src, 0, 0, 0, 0, wx, wy, dst, mask};
  HIPCHK(launchers(tuning().contract).warp(a, (hipStream_t)stream));
  return OFDIS_OK;
}

int ofdis_get_derivatives(float* out, const float* im1, const float* im2w, int w, int h, int noc, int nframes,
                          void* stream) {
  if (!out || !im1 || !im2w || w < 1 || h < 4 || nframes < 1) return fail(OFDIS_ERR_INVALID, "bad arguments (need h >= 4)");
  DerivArgs a{TvGeom{w, h, noc, nframes}, im1, 0, 0, 0, 0, im2w, out};
  HIPCHK(launchers(tuning().contract).derivatives(a, (hipStream_t)stream));
  return OFDIS_OK;
}

// The public per-function interface is row-major; the solver's operands are converted to / from the
// internal diag layout here (temporary buffers; these entry points are for parity tests and single-stage use).
int ofdis_tv_system(float* out, const float* mask, const float* wx, const float* wy, const float* du,
                    const float* dv, const float* derivs, float tv_alpha, float tv_gamma, float tv_delta, int w,
                    int h, int noc, int nframes, void* stream) {
  if (!out || !mask || !wx || !wy || !du || !dv || !derivs || w < 1 || h < 1 || nframes < 1)
    return fail(OFDIS_ERR_INVALID, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const size_t npx = (size_t)w * h * nframes;
  float* tmp = nullptr;  // du_d, dv_d, sys_d
  HIPCHK(hipMalloc((void**)&tmp, npx * 9 * sizeof(float)));
  float *du_d = tmp, *dv_d = tmp + npx, *sys_d = tmp + 2 * npx;
  const TvConsts c = tv_consts(tv_alpha, tv_gamma, tv_delta);
  const Launchers& K = launchers(tuning().contract);
  hipError_t e = K.to_diag(du, du_d, w, h, nframes, s);
  if (e == hipSuccess) e = K.to_diag(dv, dv_d, w, h, nframes, s);
  if (e == hipSuccess) {
    SystemArgs a{TvGeom{w, h, noc, nframes}, mask, wx, wy, du_d, dv_d, derivs, c.quarter_alpha, c.half_delta_over3,
                 c.half_gamma_over3, sys_d};
    e = K.tv_system(a, s);
  }
  if (e == hipSuccess) e = K.from_diag(sys_d, out, w, h, (long long)nframes * 7, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(tmp);
  if (e != hipSuccess) return hipfail(e, "ofdis_tv_system");
  return OFDIS_OK;
}

int ofdis_sor_coupled(float* du, float* dv, const float* sys, int iterations, float omega, int w, int h,
                      int nframes, void* stream) {
  if (!du || !dv || !sys || w < 1 || h < 1 || nframes < 1) return fail(OFDIS_ERR_INVALID, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const size_t npx = (size_t)w * h * nframes;
  float* tmp = nullptr;
  HIPCHK(hipMalloc((void**)&tmp, npx * 9 * sizeof(float)));
  float *du_d = tmp, *dv_d = tmp + npx, *sys_d = tmp + 2 * npx;
  const Launchers& K = launchers(tuning().contract);
  hipError_t e = K.to_diag(du, du_d, w, h, nframes, s);
  if (e == hipSuccess) e = K.to_diag(dv, dv_d, w, h, nframes, s);
  if (e == hipSuccess) e = K.to_diag(sys, sys_d, w, h, (long long)nframes * 7, s);
  if (e == hipSuccess) {
    SorArgs a{TvGeom{w, h, 1, nframes}, sys_d, du_d, dv_d, iterations, omega};
    e = K.sor(a, s);
  }
  if (e == hipSuccess) e = K.from_diag(du_d, du, w, h, nframes, s);
  if (e == hipSuccess) e = K.from_diag(dv_d, dv, w, h, nframes, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(tmp);
  if (e != hipSuccess) return hipfail(e, "ofdis_sor_coupled");
  return OFDIS_OK;
}

int ofdis_patchgrid_level(const ofdis_params* p, int level, const float* im_a, const float* im_a_dx,
                          const float* im_a_dy, const float* im_b, const float* flow_prev, float* p_out,
                          float* flow_out, int nframes, void* stream) {
  int rc = check_params(p);
  if (rc) return rc;
  if (level < 0 || level > p->sc_f || !im_a || !im_a_dx || !im_a_dy || !im_b || nframes < 1)
    return fail(OFDIS_ERR_INVALID, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const LevelGeom g = make_geom(*p, level);
  float *pv = nullptr, *pw = nullptr;
  HIPCHK(hipMalloc((void**)&pv, (size_t)g.nop * 2 * nframes * sizeof(float)));
  hipError_t e = hipMalloc((void**)&pw, (size_t)g.nop * g.novals * nframes * sizeof(float));
  if (e != hipSuccess) { (void)hipFree(pv); return hipfail(e, "hipMalloc"); }
  DisArgs a = dis_args(*p, g, nframes);
  a.im_a = im_a; a.im_a_dx = im_a_dx; a.im_a_dy = im_a_dy; a.im_b = im_b;
  a.flow_prev = flow_prev;
  a.p_out = pv;
  a.pweight = pw;
  const Launchers& K = launchers(tuning().contract);
  e = K.patch_optimize(a, s);
  if (e == hipSuccess && flow_out) {
    DensifyArgs d;
    memset(&d, 0, sizeof(d));
    d.g = g; d.nframes = nframes; d.p = pv; d.pweight = pw; d.flow_aos = flow_out;
    d.stereo = p->selectmode == 2;
    e = K.densify(d, s);
  }
  if (e == hipSuccess && p_out) e = K.patch_p_reference_order(g, nframes, pv, p_out, s);  // (the kernels keep p grid-row major)
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(pv);
  (void)hipFree(pw);
  if (e != hipSuccess) return hipfail(e, "ofdis_patchgrid_level");
  return OFDIS_OK;
}

int ofdis_varref_level(const ofdis_params* p, int level, const float* im_a, const float* im_b, float* flow,
                       int nframes, void* stream) {
  int rc = check_params(p);
  if (rc) return rc;
  if (level < 0 || level > p->sc_f || !im_a || !im_b || !flow || nframes < 1) return fail(OFDIS_ERR_INVALID, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  // a throw-away context that only carries the TV scratch of this level
  ofdis_batch b;
  b.p = *p;
  b.p.verbosity = 0;
  b.contract = tuning().contract ? 1 : 0;
  b.k = &launchers(b.contract);
  b.nframes = nframes;
  const LevelGeom g = make_geom(*p, level);
  if (g.h < 4) return fail(OFDIS_ERR_INVALID, "level must have >= 4 rows");
  const size_t npx = (size_t)g.w * g.h * nframes;
  rc = dalloc(&b, &b.wx, npx);
  if (!rc) rc = dalloc(&b, &b.wy, npx);
  if (!rc) rc = dalloc(&b, &b.du, npx);
  if (!rc) rc = dalloc(&b, &b.dv, npx);
  if (!rc) rc = dalloc(&b, &b.mask, npx);
  if (!rc) rc = dalloc(&b, &b.w_im2, npx * p->noc);
  if (!rc) rc = dalloc(&b, &b.derivs, npx * 8 * p->noc);
  if (!rc) rc = dalloc(&b, &b.sys, npx * 7);
  const TvConsts tc = tv_consts(p->tv_alpha, p->tv_gamma, p->tv_delta);
  const bool want_fused = p->noc == 1 && p->selectmode != 2 && tuning().fused_tv &&
                          b.k->tv_fused_supported(TvGeom{g.w, g.h, g.noc, nframes}, p->tv_solverit) &&
                          b.k->tv_prep_supported(TvGeom{g.w, g.h, g.noc, nframes}) &&
                          b.k->tv_fused_params_ok(tc.quarter_alpha, tc.half_delta_over3, tc.half_gamma_over3);
  if (!rc && want_fused) {
    rc = dalloc(&b, &b.wrec, npx * 2);
    if (!rc) rc = dalloc(&b, &b.uv, npx * 2);
    if (!rc && nframes <= XCU_MAX_CONTEXT_FRAMES && tuning().fused_xcu_max > 0 && p->tv_innerit * (level + 1) >= 2) {
      b.xbuf_per_frame = (size_t)(p->tv_innerit * (level + 1) - 1) * g.w * g.h * 4;
      rc = dalloc(&b, &b.xbuf, b.xbuf_per_frame * nframes);
    }
  }
  if (!rc && p->selectmode == 2) rc = dalloc(&b, &b.uu, npx);
  if (!rc) rc = dcommit(&b);
  b.total_frames = nframes;
  if (!rc && b.xbuf) {
    hipError_t e = hipMemsetAsync(b.xbuf, 0, b.xbuf_per_frame * nframes * sizeof(float), s);
    if (e != hipSuccess) rc = hipfail(e, "hipMemsetAsync");
    if (!rc) rc = xcu_state_create(&b);
  }
  if (!rc && p->selectmode == 2) {  // one channel: wx = flow, wy = 0
    b.nop = 1;
    hipError_t e = hipMemcpyAsync(b.wx, flow, npx * sizeof(float), hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(b.wy, 0, npx * sizeof(float), s);
    if (e != hipSuccess) rc = hipfail(e, "stereo flow copy");
    if (!rc) rc = run_varref_de(&b, g, im_a, im_b, flow, s);
  } else {
    // (this context owns both scratch sets, so the path is whatever want_fused decided above, whatever the knobs say now)
    if (!rc && !want_fused) {  // (the fused path starts from the AoS flow)
      TvGeom t{g.w, g.h, g.noc, nframes};
      hipError_t e = b.k->flow_split(t, flow, b.wx, b.wy, s);
      if (e != hipSuccess) rc = hipfail(e, "flow_split");
    }
    if (!rc) rc = run_varref(&b, g, im_a, im_b, flow, s, want_fused);
  }
  hipError_t e = hipStreamSynchronize(s);
  if (!rc && e != hipSuccess) rc = hipfail(e, "sync");
  // (a lost hand-over of the cross-CU variant: the flow array was refined in place by a kernel that gave up waiting)
  if (!rc && xcu_poll(&b) != OFDIS_OK) rc = OFDIS_ERR_DEVICE;
  xcu_state_destroy(&b);
  for (void* d : b.allocs) (void)hipFree(d);
  b.allocs.clear();
  return rc;
}

// ------------------------------------------------------------------------------------ kernel-selection knobs
int ofdis_get_tuning(ofdis_tuning* out) {
  if (!out) return fail(OFDIS_ERR_INVALID, "out is NULL");
  *out = ofdis::tuning();
  return OFDIS_OK;
}
int ofdis_set_tuning(const ofdis_tuning* in) {
  if (!in) return fail(OFDIS_ERR_INVALID, "tuning is NULL");
  if (in->rgb12_lpp != 0 && in->rgb12_lpp != 16 && in->rgb12_lpp != 32 && in->rgb12_lpp != 64)
    return fail(OFDIS_ERR_INVALID, "rgb12_lpp must be 0 (library's choice), 16, 32 or 64");
  if (in->fused_mw_max < 0 || in->fused_strip < 0 || in->prep_band_rows < 0 || in->fused_xcu_max < 0 || in->fused_xcu_spin < 0)
    return fail(OFDIS_ERR_INVALID, "negative knob");
  if (in->fused_strip > 64 || in->prep_band_rows > 64)  // (strips index their records with 32-bit byte offsets)
    return fail(OFDIS_ERR_INVALID, "fused_strip / prep_band_rows must be <= 64");
  if (in->fused_tp_pipe < 0 || in->fused_tp_pipe > 2) return fail(OFDIS_ERR_INVALID, "fused_tp_pipe must be 0, 1 or 2");
  if (in->contract != 0 && in->contract != 1) return fail(OFDIS_ERR_INVALID, "contract must be 0 (exact) or 1 (fused)");
  ofdis::tuning();  // initialise from the environment first
  std::lock_guard<std::mutex> lock(ofdis::g_tuning_mutex);
  ofdis::g_tuning = *in;
  ++ofdis::g_tuning_epoch;
  return OFDIS_OK;
}

// ------------------------------------------------------------------------------------ memory helpers
void* ofdis_dev_alloc(size_t bytes) {
  void* d = nullptr;
  if (hipMalloc(&d, bytes ? bytes : 1) != hipSuccess) return nullptr;
  return d;
}
void ofdis_dev_free(void* p) { (void)hipFree(p); }
int ofdis_memcpy_h2d(void* dst, const void* src, size_t bytes) {
  HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return OFDIS_OK;
}
int ofdis_memcpy_d2h(void* dst, const void* src, size_t bytes) {
  HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return OFDIS_OK;
}
int ofdis_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return OFDIS_OK;
}
int ofdis_sync(void* stream) {
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  // a lost hand-over of the cross-CU fused TV variant belongs to the context that launched it: report it to whoever
  // synchronises the stream that context's last pass went to (ofdis_batch_status asks one context directly)
  int rc = OFDIS_OK;
  int dev = -1;
  (void)hipGetDevice(&dev);  // (one host thread per GPU, each synchronising its own null stream: only this device's contexts)
  std::lock_guard<std::mutex> lock(g_xcu_mutex);
  for (ofdis_batch* b : g_xcu_contexts) {
    XcuState* x = b->xcu;
    if (!x || !x->ran || x->last_stream != (hipStream_t)stream || x->last_device != dev || x->told_sync) continue;
    if (xcu_poll(b) != OFDIS_OK) {  // (reported here once; ofdis_batch_status / _download keep reporting it until the next pass)
      x->told_sync = true;
      rc = OFDIS_ERR_DEVICE;
    }
  }
  return rc;
}

int ofdis_batch_status(ofdis_batch* b) {
  if (!b) return fail(OFDIS_ERR_INVALID, "batch is NULL");
  return xcu_poll(b);
}

}  // extern "C"
