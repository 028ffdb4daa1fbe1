#!/bin/bash
# parity + timing of A/B builds:  bash tools/exp_sl.sh variant...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/exp_sl; mkdir -p $OUT
for v in "$@"; do
  lib=$R/of_dis_amd/lib/ab_$v/libofdis_hip.so
  OFDIS_LIB=$lib timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "varref" > $OUT/pytest_$v.log 2>&1; echo "$v pytest rc=$?"; tail -2 $OUT/pytest_$v.log
done
for v in base "$@" ${EXTRA}; do
  lib=$R/of_dis_amd/lib/ab_$v/libofdis_hip.so; [ $v = base ] && lib=$R/of_dis_amd/lib/libofdis_hip.so
  [ -f $lib ] || continue
  for p in 1 2; do echo -n "$v p$p : "; timeout 300 python tools/kbench.py OFDIS_LIB=$lib -- --steps 10 --warmup 3 --no-extras --pipeline $p; done
done 2>&1 | tee $OUT/variants.txt
