#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r3n; mkdir -p $OUT
( for b in 8192 16384; do for pl in 2 3 4; do
  echo -n "b$b p$pl: "; timeout 400 python tools/kbench.py A=1 -- --steps 6 --warmup 2 --no-extras --pipeline $pl --batch $b
done; done ) 2>&1 | tee $OUT/variants.txt
