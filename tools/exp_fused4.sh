#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/exp_fused4; mkdir -p $OUT
for v in base "$@"; do
  lib=$R/of_dis_amd/lib/ab_$v/libofdis_hip.so; [ $v = base ] && lib=$R/of_dis_amd/lib/libofdis_hip.so
  [ -f $lib ] || continue
  for b in 64 512; do echo -n "$v batch=$b : "; timeout 300 python tools/kbench.py OFDIS_LIB=$lib -- --steps 200 --warmup 20 --no-extras --batch $b; done
  echo -n "$v batch=4096 p2 : "; timeout 300 python tools/kbench.py OFDIS_LIB=$lib -- --steps 10 --warmup 3 --no-extras --pipeline 2
done 2>&1 | tee $OUT/variants.txt
