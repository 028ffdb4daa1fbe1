#!/bin/bash
# round 5, call 6: the iteration-pipelined TV kernel at two instead of three workgroups per CU (LDS padding): does the headline
# gain from other kernels running beside it?  (timed, pipelined total is what counts here)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r5f; mkdir -p $OUT
bash tools/ab_bench.sh 3 main mw2 2>&1 | tee $OUT/ab.txt
