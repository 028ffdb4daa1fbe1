#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r3k; mkdir -p $OUT
KB="--steps 6 --warmup 2 --no-extras --pipeline 2"
( echo -n "b4096 auto: "; timeout 300 python tools/kbench.py A=1 -- $KB --batch 4096
  echo -n "b8192 S=1: "; timeout 300 python tools/kbench.py OFDIS_FUSED_STRIP=1 -- $KB --batch 8192
  echo -n "b8192 auto: "; timeout 300 python tools/kbench.py A=1 -- $KB --batch 8192
  echo -n "b16384 S=1: "; timeout 400 python tools/kbench.py OFDIS_FUSED_STRIP=1 -- $KB --batch 16384
  echo -n "b16384 auto: "; timeout 400 python tools/kbench.py A=1 -- $KB --batch 16384
  echo -n "b16384 S=2: "; timeout 400 python tools/kbench.py OFDIS_FUSED_STRIP=2 -- $KB --batch 16384
  echo -n "b16384 S=4: "; timeout 400 python tools/kbench.py OFDIS_FUSED_STRIP=4 -- $KB --batch 16384
) 2>&1 | tee $OUT/variants.txt
