#!/bin/bash
# VALU issue probes:  gpurun -- "bash tools/exp_probes.sh"  ->  gpurun_out/probes/
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/probes; mkdir -p $OUT
for p in ${PROBES:-bank_probe stream_probe}; do
  [ -x tools/probes/$p ] && timeout 200 tools/probes/$p > $OUT/$p.txt 2>&1
  echo "== $p"; cat $OUT/$p.txt
done
