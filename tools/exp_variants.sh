#!/bin/bash
# kernel tables of A/B library builds (tools/ab_build.py):  bash tools/exp_variants.sh TAG name [name ...]   (PIPE=1|2, CHECK=1: varref parity per build)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
KB="--steps 10 --warmup 3 --no-extras --pipeline ${PIPE:-2}"
( for v in base "$@"; do
    lib=$R/of_dis_amd/lib/libofdis_hip.so; [ $v != base ] && lib=$R/of_dis_amd/lib/ab_$v/libofdis_hip.so
    [ -f $lib ] || { echo "$v: no library"; continue; }
    if [ -n "$CHECK" ] && [ $v != base ]; then
      OFDIS_LIB=$lib timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_flow.py -q -m gpu -x -k "varref or dropin or strips or bands" > $OUT/pytest_$v.log 2>&1; echo "$v parity rc=$? $(tail -1 $OUT/pytest_$v.log)"
    fi
    echo -n "$v: "; timeout 300 python tools/kbench.py OFDIS_LIB=$lib -- $KB
  done ) 2>&1 | sed "s#$R/##g" | tee $OUT/variants.txt
