#!/bin/bash
# PMC counters of the kernels whose name contains $1 over a bench.py pass:  bash tools/pmc_kernel.sh gray8 TAG "--contract fused --batch 4096"
# One rocprofv3 pass per counter set (separate --pmc runs, kernel trace only; never combined with other trace domains).
K=$1; TAG=${2:-pmck}; BARGS=${3:-"--batch 4096"}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
SETS=${PMC_SETS:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD"}  # (TA_* / TCP_* sets abort rocprofv3 on this stack)
i=0
IFS='|' read -ra ARR <<< "$SETS"
for set in "${ARR[@]}"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw$i -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-parity --no-extras --pipeline 1 $BARGS > $OUT/set$i.log 2>&1
  f=$(find $OUT/raw$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f $K > $OUT/set$i.txt || tail -5 $OUT/set$i.log
  rm -rf $OUT/raw$i
done
cat $OUT/set*.txt > $OUT/summary.txt
