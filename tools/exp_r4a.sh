#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r4a; mkdir -p $OUT
( for b in 1 16 64 128 256; do
  KB="--steps 50 --warmup 5 --no-extras --pipeline 1 --batch $b"
  echo -n "b$b gray8: "; timeout 300 python tools/kbench.py -- $KB
  echo -n "b$b generic: "; timeout 300 python tools/kbench.py OFDIS_NO_GRAY8=1 -- $KB
done ) 2>&1 | sed "s#$R/##g" | tee $OUT/variants.txt
