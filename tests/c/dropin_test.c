/* dropin_test.c -- a C99 caller of the drop-in boundary, the way a maintainer of the reference would use it:
 * includes include/ofdis.h, fills ofdis_params like the constructor's argument list (oflow.h:84-111), calls
 * ofdis_flow() on host pyramids and compares the BYTES of the result with an expected flow.
 *
 *   dropin_test case.bin [repeats]
 *
 * case.bin (written by tests/test_cli.py from the oracle; little endian):
 *   int32 magic 0x4F464449, int32 sizeof(ofdis_params), ofdis_params, int32 has_initflow,
 *   for level sc_l..sc_f, for kind A, A_dx, A_dy, B: the plane, (w/2^l + 2 pad) x (h/2^l + 2 pad) x noc float32
 *   [initflow: (w >> (sc_f+1)) x (h >> (sc_f+1)) x 2 float32]
 *   expected flow: (w >> sc_l) x (h >> sc_l) x 2 float32
 * Prints one line: "dropin_test: OK <n> floats identical, <ms> ms per call over <repeats> calls" and returns 0,
 * or a diagnostic and 1.  Test infrastructure: the expected flow comes from oracle/, this program only compares. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ofdis.h"

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static int read_exact(FILE* f, void* dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes; }

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s case.bin [repeats]\n", argv[0]);
    return 2;
  }
  const int repeats = argc > 2 ? atoi(argv[2]) : 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  int magic = 0, psize = 0, has_init = 0;
  ofdis_params prm;
  if (!read_exact(f, &magic, 4) || magic != 0x4F464449 || !read_exact(f, &psize, 4) || psize != (int)sizeof(prm) ||
      !read_exact(f, &prm, sizeof(prm)) || !read_exact(f, &has_init, 4)) {
    fprintf(stderr, "dropin_test: bad header (sizeof(ofdis_params) here = %d, file says %d)\n", (int)sizeof(prm), psize);
    return 1;
  }
  const int nl = prm.sc_f + 1;
  /* the constructor's six arrays of sc_f+1 plane pointers; entries below sc_l stay NULL (oflow.h:85-87) */
  const float** pyr[4];
  int k, l;
  for (k = 0; k < 4; ++k) pyr[k] = (const float**)calloc((size_t)nl, sizeof(float*));
  for (l = prm.sc_l; l <= prm.sc_f; ++l) {
    const size_t n = (size_t)((prm.width >> l) + 2 * prm.imgpadding) * ((prm.height >> l) + 2 * prm.imgpadding) * prm.noc;
    for (k = 0; k < 4; ++k) {
      float* pl = (float*)malloc(n * sizeof(float));
      if (!pl || !read_exact(f, pl, n * sizeof(float))) { fprintf(stderr, "dropin_test: short file (level %d)\n", l); return 1; }
      pyr[k][l] = pl;
    }
  }
  float* init = NULL;
  if (has_init) {
    const size_t n = (size_t)(prm.width >> (prm.sc_f + 1)) * (prm.height >> (prm.sc_f + 1)) * 2;
    init = (float*)malloc(n * sizeof(float));
    if (!init || !read_exact(f, init, n * sizeof(float))) { fprintf(stderr, "dropin_test: short file (initflow)\n"); return 1; }
  }
  const size_t nflow = (size_t)(prm.width >> prm.sc_l) * (prm.height >> prm.sc_l) * 2;
  float* expect = (float*)malloc(nflow * sizeof(float));
  float* got = (float*)malloc(nflow * sizeof(float));
  if (!expect || !got || !read_exact(f, expect, nflow * sizeof(float))) { fprintf(stderr, "dropin_test: short file (flow)\n"); return 1; }
  fclose(f);

  if (ofdis_device_count() < 1) { fprintf(stderr, "dropin_test: no HIP device\n"); return 3; }
  int rc = OFDIS_OK, r;
  double t0 = 0;
  for (r = 0; r < repeats + 1 && rc == OFDIS_OK; ++r) { /* the first call builds the context and is not timed */
    if (r == 1) t0 = now_ms();
    memset(got, 0xff, nflow * sizeof(float));
    rc = ofdis_flow(&prm, pyr[0], pyr[1], pyr[2], pyr[3], NULL, NULL, got, init);
  }
  const double ms = repeats > 0 ? (now_ms() - t0) / repeats : 0.0;
  if (rc != OFDIS_OK) {
    fprintf(stderr, "dropin_test: ofdis_flow returned %d: %s\n", rc, ofdis_last_error());
    return 1;
  }
  ofdis_flow_cache_clear();
  if (memcmp(got, expect, nflow * sizeof(float)) != 0) {
    size_t i, bad = 0, first = nflow;
    for (i = 0; i < nflow; ++i)
      if (memcmp(&got[i], &expect[i], 4) != 0) { if (first == nflow) first = i; ++bad; }
    fprintf(stderr, "dropin_test: MISMATCH: %lu of %lu floats differ, first at %lu: %.9g vs %.9g\n", (unsigned long)bad,
            (unsigned long)nflow, (unsigned long)first, got[first], expect[first]);
    return 1;
  }
  printf("dropin_test: OK %lu floats identical, %.4f ms per call over %d calls\n", (unsigned long)nflow, ms, repeats);
  return 0;
}
