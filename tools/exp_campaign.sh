#!/bin/bash
# Seeded random parity campaigns (tests/test_gpu_*.py -k random) with shifted seeds:
#   default library, the trigger-free + LDS-slot-ring build of the fused TV kernel (tools/ab_build.py tfall ...), and the
#   two-patches-per-wavefront RGB mapping.   gpurun -- "bash tools/exp_campaign.sh FIRST COUNT"
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/campaign; mkdir -p $OUT
first=${1:-3000}; count=${2:-4}
for ((i=0;i<count;i++)); do
  off=$((first + i*1000))
  for cfg in "default:" "trigger_free_slotlds:OFDIS_LIB=$R/of_dis_amd/lib/ab_tfall/libofdis_hip.so" "rgb_two_per_wave:OFDIS_RGB12_LPP=32"; do
    name=${cfg%%:*}; envs=${cfg#*:}
    [ "$name" = trigger_free_slotlds ] && [ ! -f $R/of_dis_amd/lib/ab_tfall/libofdis_hip.so ] && continue
    env OFDIS_TEST_SEED_OFFSET=$off $envs timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_flow.py -q -k random > $OUT/${name}_$off.log 2>&1
    echo "seed offset $off $name: $(tail -1 $OUT/${name}_$off.log)"
  done
done | tee $OUT/summary.txt
