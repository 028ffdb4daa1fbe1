#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r3v; mkdir -p $OUT
OFDIS_LIB=$R/of_dis_amd/lib/ab_xdbg/libofdis_hip.so OFDIS_FUSED_XCU_MAX=1073741824 timeout 300 python tools/exp_xcu_debug.py 2>&1 | tail -40 | tee $OUT/debug.txt
