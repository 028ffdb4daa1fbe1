// cli_params.h -- the reference's command-line parameter block (run_dense.cpp:225-294, README.md:48-88), shared by the
// single-pair binaries (run_dense_main.cpp) and the sequence driver (run_seq_main.cpp).
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include "ofdis.h"

namespace ofdis_host {

// argv[k0 .. argc) is empty (operating point 2), one number (operating point 1..4) or the 20 explicit parameters.
// Fills everything but width / height (the caller pads, run_dense.cpp:298-311).  Returns 0, or the exit status (2 = usage).
inline int parse_params(int argc, char** argv, int k0, int width_org, int noc, int selectmode, ofdis_params* p) {
  const int n = argc - k0;
  if (n <= 1) {
    const int op = (n == 1) ? atoi(argv[k0]) : 2;
    if (ofdis_params_oppoint(p, op, width_org, noc) != OFDIS_OK) {
      fprintf(stderr, "%s\n", ofdis_last_error());
      return 1;
    }
  } else {
    if (n < 20) {
      fprintf(stderr, "need all 20 parameters (README.md:57-88), got %d\n", n);
      return 2;
    }
    ofdis_params_oppoint(p, 2, width_org, noc);
    int k = k0;
    p->sc_f = atoi(argv[k++]); p->sc_l = atoi(argv[k++]);
    p->max_iter = atoi(argv[k++]); p->min_iter = atoi(argv[k++]);
    p->dp_thresh = (float)atof(argv[k++]); p->dr_thresh = (float)atof(argv[k++]); p->res_thresh = (float)atof(argv[k++]);
    p->p_samp_s = atoi(argv[k++]); p->patove = (float)atof(argv[k++]);
    p->usefbcon = atoi(argv[k++]); p->patnorm = atoi(argv[k++]); p->costfct = atoi(argv[k++]); p->usetvref = atoi(argv[k++]);
    p->tv_alpha = (float)atof(argv[k++]); p->tv_gamma = (float)atof(argv[k++]); p->tv_delta = (float)atof(argv[k++]);
    p->tv_innerit = atoi(argv[k++]); p->tv_solverit = atoi(argv[k++]); p->tv_sor = (float)atof(argv[k++]);
    p->verbosity = atoi(argv[k++]);
    p->imgpadding = p->p_samp_s;
  }
  p->selectmode = selectmode;
  return 0;
}

// pad to a multiple of 2^lv_f (run_dense.cpp:298-311)
inline void pad_size(ofdis_params* p, int width_org, int height_org) {
  const int scfct = 1 << p->sc_f;
  p->width = width_org + (scfct - width_org % scfct) % scfct;
  p->height = height_org + (scfct - height_org % scfct) % scfct;
}

}  // namespace ofdis_host
