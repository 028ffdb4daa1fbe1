#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r4f; mkdir -p $OUT
L=$R/of_dis_amd/lib/ab_ctx1024/libofdis_hip.so
( for b in 384 512 768 1024; do
  KB="--steps 30 --warmup 5 --no-extras --pipeline 1 --batch $b"
  echo -n "b$b split: "; timeout 300 python tools/kbench.py OFDIS_LIB=$L OFDIS_FUSED_XCU_MAX=0 -- $KB
  echo -n "b$b xcu: "; timeout 300 python tools/kbench.py OFDIS_LIB=$L OFDIS_FUSED_XCU_MAX=1073741824 -- $KB
done ) 2>&1 | sed "s#$R/##g;s#OFDIS_LIB=[^ ]* ##" | tee $OUT/variants.txt
