#!/bin/bash
# round-3 experiment A: gray patch kernel with ds_swizzle reductions; multi-wave fused TV kernels at full batch size
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r3a; mkdir -p $OUT
KB="--steps 10 --warmup 3 --no-extras"
( echo -n "base p1: "; timeout 300 python tools/kbench.py A=1 -- $KB --pipeline 1
  echo -n "base p2: "; timeout 300 python tools/kbench.py A=1 -- $KB --pipeline 2
  L=$R/of_dis_amd/lib/ab_swz/libofdis_hip.so
  echo -n "swz p1: "; timeout 300 python tools/kbench.py OFDIS_LIB=$L -- $KB --pipeline 1
  echo -n "swz p2: "; timeout 300 python tools/kbench.py OFDIS_LIB=$L -- $KB --pipeline 2
  echo -n "base MODE1 p1: "; timeout 300 python tools/kbench.py OFDIS_FUSED_MW_MAX=1073741824 OFDIS_FUSED_NO_SPLIT=1 -- $KB --pipeline 1
  echo -n "base MODE1 p2: "; timeout 300 python tools/kbench.py OFDIS_FUSED_MW_MAX=1073741824 OFDIS_FUSED_NO_SPLIT=1 -- $KB --pipeline 2
  echo -n "base MODE2 p1: "; timeout 300 python tools/kbench.py OFDIS_FUSED_MW_MAX=1073741824 -- $KB --pipeline 1
  L=$R/of_dis_amd/lib/ab_tf/libofdis_hip.so
  echo -n "tf MODE0 p1: "; timeout 300 python tools/kbench.py OFDIS_LIB=$L -- $KB --pipeline 1
  echo -n "tf MODE1 p1: "; timeout 300 python tools/kbench.py OFDIS_LIB=$L OFDIS_FUSED_MW_MAX=1073741824 OFDIS_FUSED_NO_SPLIT=1 -- $KB --pipeline 1
  echo -n "tf MODE1 p2: "; timeout 300 python tools/kbench.py OFDIS_LIB=$L OFDIS_FUSED_MW_MAX=1073741824 OFDIS_FUSED_NO_SPLIT=1 -- $KB --pipeline 2
) 2>&1 | tee $OUT/variants.txt
OFDIS_LIB=$R/of_dis_amd/lib/ab_swz/libofdis_hip.so timeout 600 python -m pytest tests/test_gpu_flow.py -x -q > $OUT/pytest_swz.log 2>&1; echo "pytest swz rc=$?"; tail -3 $OUT/pytest_swz.log
# PMC: issue / wait counters of the patch kernel, DPP against ds_swizzle reductions
export TMPDIR=/tmp; cd /tmp
for v in base swz; do
  lib=$R/of_dis_amd/lib/libofdis_hip.so; [ $v = swz ] && lib=$R/of_dis_amd/lib/ab_swz/libofdis_hip.so
  OFDIS_LIB=$lib timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/raw_$v -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-parity --no-extras --pipeline 1 > $OUT/pmc_$v.log 2>&1
  f=$(find $OUT/raw_$v -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f > $OUT/pmc_$v.txt || tail -5 $OUT/pmc_$v.log
  rm -rf $OUT/raw_$v
  grep -i "patch_optimize\|^#\|kernel" $OUT/pmc_$v.txt | head -20
done
