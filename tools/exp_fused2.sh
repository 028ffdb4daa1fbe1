#!/bin/bash
# parity of the fused TV kernel variants + A/B timing:  gpurun -- "bash tools/exp_fused2.sh [variant ...]"
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/exp_fused2; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "${TESTK:-trimmed_div_sqrt or varref or tv_system}" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for v in base "$@"; do
  lib=$R/of_dis_amd/lib/ab_$v/libofdis_hip.so; [ $v = base ] && lib=$R/of_dis_amd/lib/libofdis_hip.so
  [ -f $lib ] || continue
  echo -n "$v : "; timeout 300 python tools/kbench.py OFDIS_LIB=$lib -- --steps 10 --warmup 3 --no-extras --pipeline ${PIPE:-1}
done 2>&1 | tee $OUT/variants.txt
