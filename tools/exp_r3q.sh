#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r3q; mkdir -p $OUT
L=$R/of_dis_amd/lib/ab_xlds/libofdis_hip.so
KB="--steps 8 --warmup 2 --no-extras --pipeline 2 --batch 16384"
( for lds in 0 60000 60000 0 40000; do
  echo -n "fused dyn LDS $lds: "; timeout 400 python tools/kbench.py OFDIS_LIB=$L X_FUSED_LDS=$lds -- $KB
done ) 2>&1 | sed "s#$R/##g" | tee $OUT/variants.txt
