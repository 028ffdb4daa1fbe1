#!/bin/bash
# where the step time of the cross-CU fused TV variant goes: timing-only builds (results wrong)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r3s; mkdir -p $OUT
KB="--steps 50 --warmup 5 --no-extras --pipeline 1 --batch 64"
( echo -n "product xcu: "; timeout 300 python tools/kbench.py OFDIS_FUSED_XCU_MAX=1073741824 -- $KB
  echo -n "product split: "; timeout 300 python tools/kbench.py OFDIS_FUSED_XCU_MAX=0 -- $KB
for v in xNOHANDOFF xNOHANDOFF_NODATA xNOHANDOFF_NOROWS xNOHANDOFF_NOSOLVE xNOHANDOFF_NODATA_NOROWS xNOHANDOFF_NODATA_NOROWS_NOSOLVE; do
  echo -n "$v: "; timeout 300 python tools/kbench.py OFDIS_LIB=$R/of_dis_amd/lib/ab_$v/libofdis_hip.so OFDIS_FUSED_XCU_MAX=1073741824 -- $KB
done ) 2>&1 | sed "s#$R/##g" | tee $OUT/variants.txt
