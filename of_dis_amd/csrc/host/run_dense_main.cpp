// run_OF_INT / run_OF_RGB -- the reference's command line (run_dense.cpp:185-431, README.md:48-88) on top of
// the C ABI (include/ofdis.h).  Compile with -DOFDIS_NOC=1 (run_OF_INT) or 3 (run_OF_RGB), mirroring the
// reference's per-binary SELECTCHANNEL.
//
//   run_OF_INT img1 img2 out.flo                      operating point 2
//   run_OF_INT img1 img2 out.flo X                    operating point X in 1..4
//   run_OF_INT img1 img2 out.flo p1 .. p20            the 20 explicit parameters
// With -DOFDIS_MODE=2: run_DE_INT / run_DE_RGB, the stereo-depth binaries (one displacement channel, .pfm output).
//
// Pipeline: read 8-bit images -> upload -> on-device padding, pyramid, Sobel (ofdis_batch_build_pyramids_u8)
// -> hot path (ofdis_batch_run) -> x2^lv_l, bilinear upsample, crop on the device (ofdis_batch_upsample) -> download
// (run_dense.cpp:406-414, cv::resize INTER_LINEAR restated) -> Middlebury .flo.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include <algorithm>
#include <string>
#include <vector>

#include "cli_params.h"
#include "image_io.h"
#include "ofdis.h"

#ifndef OFDIS_NOC
#define OFDIS_NOC 1
#endif
#ifndef OFDIS_MODE  // the reference's SELECTMODE: 1 optical flow (run_OF_*, .flo), 2 stereo depth (run_DE_*, .pfm)
#define OFDIS_MODE 1
#endif
#define OFDIS_NCH (OFDIS_MODE == 2 ? 1 : 2)

static double now_ms() {
  struct timeval tv;
  gettimeofday(&tv, nullptr);
  return tv.tv_sec * 1000.0 + tv.tv_usec / 1000.0;
}

int main(int argc, char** argv) {
  const double t_start = now_ms();
  if (argc < 4) {
    fprintf(stderr, "usage: %s img1 img2 out.flo [oppoint 1-4 | lv_f lv_l maxiter miniter mindprate mindrrate minimgerr "
                    "patchsz poverl usefbcon patnorm costfct usetvref tv_alpha tv_gamma tv_delta tv_innerit tv_solverit "
                    "tv_sor verbosity]\n", argv[0]);
    return 2;
  }
  const char* f_a = argv[1];
  const char* f_b = argv[2];
  const char* f_out = argv[3];
  ofdis_host::Image8 ia, ib;
  std::string err;
  if (!ofdis_host::read_image(f_a, OFDIS_NOC, &ia, &err) || !ofdis_host::read_image(f_b, OFDIS_NOC, &ib, &err)) {
    fprintf(stderr, "%s\n", err.c_str());
    return 1;
  }
  if (ia.width != ib.width || ia.height != ib.height) {
    fprintf(stderr, "image sizes differ\n");
    return 1;
  }
  const int width_org = ia.width, height_org = ia.height;

  // *** parameters (run_dense.cpp:225-294) and padding to a multiple of 2^lv_f (run_dense.cpp:298-311)
  ofdis_params p;
  if (int st = ofdis_host::parse_params(argc, argv, 4, width_org, OFDIS_NOC, OFDIS_MODE, &p)) return st;
  ofdis_host::pad_size(&p, width_org, height_org);
  const int verbosity = p.verbosity;
  if (verbosity > 1) printf("TIME (Image loading     ) (ms): %3g\n", now_ms() - t_start);

  double t0 = now_ms();
  ofdis_batch* b = nullptr;
  if (ofdis_batch_create(&b, &p, 1) != OFDIS_OK) {
    fprintf(stderr, "%s\n", ofdis_last_error());
    return 1;
  }
  const size_t nbytes = (size_t)width_org * height_org * OFDIS_NOC;
  void* da = ofdis_dev_alloc(nbytes);
  void* db = ofdis_dev_alloc(nbytes);
  int rc = (da && db) ? OFDIS_OK : OFDIS_ERR_NOMEM;
  if (!rc) rc = ofdis_memcpy_h2d(da, ia.data.data(), nbytes);
  if (!rc) rc = ofdis_memcpy_h2d(db, ib.data.data(), nbytes);
  if (!rc) rc = ofdis_batch_build_pyramids_u8(b, (const uint8_t*)da, (const uint8_t*)db, width_org, height_org, nullptr);
  if (!rc) rc = ofdis_sync(nullptr);
  if (rc) {
    fprintf(stderr, "%s\n", ofdis_last_error());
    return 1;
  }
  if (verbosity > 1) printf("TIME (Pyramide+Gradients) (ms): %3g\n", now_ms() - t0);

  // *** the hot path (prints the reference's per-level TIME lines itself when verbosity > 1)
  rc = ofdis_batch_run(b, nullptr);
  // x 2^lv_l, bilinear upsample, crop (run_dense.cpp:406-414) on the device, then one download
  t0 = now_ms();
  std::vector<float> full((size_t)OFDIS_NCH * width_org * height_org);
  void* dfull = ofdis_dev_alloc(full.size() * sizeof(float));
  if (!rc && !dfull) rc = OFDIS_ERR_NOMEM;
  if (!rc) rc = ofdis_batch_upsample(b, (float*)dfull, width_org, height_org, nullptr);
  if (!rc) rc = ofdis_sync(nullptr);
  if (rc == OFDIS_ERR_DEVICE && ofdis_batch_status(b) != OFDIS_OK) {
    // the pass reported itself as failed (a lost hand-over of the cross-CU fused TV variant): the context no longer uses
    // that variant, so the pass is repeated once; a second failure ends the program with a non-zero status
    fprintf(stderr, "%s\n", ofdis_last_error());
    rc = ofdis_batch_run(b, nullptr);
    if (!rc) rc = ofdis_batch_upsample(b, (float*)dfull, width_org, height_org, nullptr);
    if (!rc) rc = ofdis_sync(nullptr);
  }
  if (!rc) rc = ofdis_memcpy_d2h(full.data(), dfull, full.size() * sizeof(float));
  if (rc) {
    fprintf(stderr, "%s\n", ofdis_last_error());
    return 1;
  }
  ofdis_dev_free(dfull);
#if OFDIS_MODE == 2
  if (!ofdis_host::write_pfm(f_out, full.data(), width_org, height_org, &err)) {
#else
  if (!ofdis_host::write_flo(f_out, full.data(), width_org, height_org, &err)) {
#endif
    printf("%s\n", err.c_str());  // the reference reports and carries on (run_dense.cpp:24-25)
  }
  if (verbosity > 1) printf("TIME (Saving flow file  ) (ms): %3g\n", now_ms() - t0);
  ofdis_dev_free(da);
  ofdis_dev_free(db);
  ofdis_batch_destroy(b);
  return 0;
}
