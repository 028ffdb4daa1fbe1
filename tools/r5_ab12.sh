#!/bin/bash
# round 5: the two pipelined sub-batches on CU-masked streams (half of the compute units each) instead of sharing the chip in time
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r5t; mkdir -p $OUT
bash tools/ab_bench.sh 1 main 2>&1 | tee $OUT/ab.txt
for m in 1 2; do echo "OFDIS_CU_SPLIT=$m"; OFDIS_CU_SPLIT=$m bash tools/ab_bench.sh 2 cusplit 2>&1 | tee -a $OUT/ab.txt; done
echo "OFDIS_CU_SPLIT=2, four sub-batches"; OFDIS_CU_SPLIT=2 OFDIS_LIB=of_dis_amd/lib/ab_cusplit/libofdis_hip.so timeout 300 python bench.py --no-extras --cpu-seconds 0 --contract fused --pipeline 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cusplit4', round(d['value']), d['ms_per_step'])" | tee -a $OUT/ab.txt
