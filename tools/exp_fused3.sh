#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/exp_fused3; mkdir -p $OUT
timeout 300 python tools/div_fail.py 2>&1 | tee $OUT/div_fail.txt | tail -30
for cfg in "A=1" "OFDIS_LIB=$R/of_dis_amd/lib/ab_alias2/libofdis_hip.so" "OFDIS_FUSED_MW_MAX=100000000" "OFDIS_FUSED_MW_MAX=100000000 OFDIS_FUSED_NO_SPLIT=1"; do
  for p in 1 2; do echo -n "$cfg pipeline=$p : "; timeout 300 python tools/kbench.py $cfg -- --steps 10 --warmup 3 --no-extras --pipeline $p; done
done 2>&1 | tee $OUT/variants.txt
