#!/bin/bash
# round 5: three micro-variants: dper = densify_quad as 1024 persistent blocks (4 per CU); p6 = prep LDS trimmed to 27.2 KB (6
# instead of 5 blocks per CU); ruv = MODE 1/2 du/dv ring read without the iteration-0 branch (a ring of zeros)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r5n; mkdir -p $OUT
for v in dper p6 ruv; do OFDIS_LIB=$R/of_dis_amd/lib/ab_$v/libofdis_hip.so timeout 600 python -m pytest tests -m gpu -x -q -k "prep or varref or golden or fused or strips or patchgrid or level_flows or contract_production" > $OUT/pytest_$v.log 2>&1; echo $v; tail -1 $OUT/pytest_$v.log; done
bash tools/ab_bench.sh 2 main dper p6 ruv 2>&1 | tee $OUT/ab.txt
