#!/bin/bash
# HBM traffic / VALU counters of the headline's kernels -> profiles/traffic.json:  bash tools/pmc_round.sh TAG [BATCH] [CONTRACT]
# Three rocprofv3 passes (FETCH_SIZE; WRITE_SIZE; SQ_INSTS_VALU + GRBM_GUI_ACTIVE), each in its own run with nothing but the
# kernel trace, over tools/pmc_pass.py (no torch).  BATCH = ONE sub-batch of the headline run (16384 pairs as 2 x 8192).
TAG=${1:-x}; BATCH=${2:-8192}; CONTRACT=${3:-fused}; PASSES=4
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$n -- python $R/tools/pmc_pass.py --batch $BATCH --contract $CONTRACT --passes $PASSES > $OUT/pmc_$n.log 2>&1
  tail -1 $OUT/pmc_$n.log | cut -c1-200
done
ff=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
fw=$(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
fs=$(find $OUT/pmc_SQ_INSTS_VALU -name "*counter_collection.csv" | head -1)
cd $R
if [ -n "$ff" ] && [ -n "$fw" ]; then
  python tools/pmc_traffic.py $ff $fw $BATCH on $PASSES "${fs:--}" $CONTRACT > $OUT/traffic.log && cp profiles/traffic_$CONTRACT.json $OUT/traffic_$CONTRACT.json
  for f in $ff $fw $fs; do python tools/pmc_summary.py $f >> $OUT/pmc_sums.txt; done
fi
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_INSTS_VALU
ls $OUT
