#!/bin/bash
# round-3 experiment B: does the gray patch kernel pair its VALU issue once neither DPP moves nor load stalls are in the way?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r3b; mkdir -p $OUT
KB="--steps 10 --warmup 3 --no-extras --pipeline 1"
( for v in "" swz x_dpp_noload x_swz_noload; do
    lib=$R/of_dis_amd/lib/libofdis_hip.so; [ -n "$v" ] && lib=$R/of_dis_amd/lib/ab_$v/libofdis_hip.so
    echo -n "${v:-base}: "; timeout 300 python tools/kbench.py OFDIS_LIB=$lib -- $KB
  done ) 2>&1 | tee $OUT/variants.txt
