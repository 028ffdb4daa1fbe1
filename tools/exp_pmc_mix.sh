#!/bin/bash
# instruction mix of one kernel (name substring $1) on the headline workload, two PMC passes:  bash tools/exp_pmc_mix.sh tv_prep TAG
K=$1; TAG=${2:-pmcmix}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw$i -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-parity --no-extras --pipeline 1 > $OUT/set$i.log 2>&1
  f=$(find $OUT/raw$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f $K > $OUT/set$i.txt || tail -5 $OUT/set$i.log
  rm -rf $OUT/raw$i
done
cat $OUT/set*.txt | grep -v "_Grid\|_LDS\|_VGPR\|_Workgroup"
