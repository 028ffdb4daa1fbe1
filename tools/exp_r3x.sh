#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r3x; mkdir -p $OUT
for v in xdbg xdbgnf; do echo "== $v"; OFDIS_LIB=$R/of_dis_amd/lib/ab_$v/libofdis_hip.so timeout 300 python tools/exp_xcu_debug.py 2>&1 | tail -15 | sort -k3,3n -k5,5n; done | tee $OUT/debug.txt
