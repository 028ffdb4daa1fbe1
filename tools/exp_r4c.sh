#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r4c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -3
( for b in 1 64 128 256 512 1024 4096; do
  KB="--steps 30 --warmup 5 --no-extras --pipeline 1 --batch $b"
  echo -n "b$b rule: "; timeout 300 python tools/kbench.py -- $KB
  echo -n "b$b band_rows 8: "; timeout 300 python tools/kbench.py OFDIS_PREP_BAND_ROWS=8 -- $KB
done ) 2>&1 | sed "s#$R/##g" | tee $OUT/variants2.txt
