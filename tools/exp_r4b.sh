#!/bin/bash
# which wavefront of the cross-CU fused TV kernel sets the step: timing-only builds (results wrong)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r4b; mkdir -p $OUT
KB="--steps 50 --warmup 5 --no-extras --pipeline 1 --batch 1"
( echo -n "product: "; timeout 300 python tools/kbench.py -- $KB
for v in yNODATA yNOSMOOTH yNOINV yNOSMOOTH_NOINV yNOSOLVE yNODATA_NOSMOOTH_NOINV_NOSOLVE; do
  echo -n "$v: "; timeout 300 python tools/kbench.py OFDIS_LIB=$R/of_dis_amd/lib/ab_$v/libofdis_hip.so -- $KB
done ) 2>&1 | sed "s#$R/##g;s#OFDIS_LIB=[^ ]* ##" | tee $OUT/variants.txt
