#!/bin/bash
# PMC passes over the headline workload, one counter set per run (nothing but the kernel trace next to --pmc):
#   bash tools/pmc_sets.sh TAG "SET1 counters" "SET2 counters" ...
# Writes gpurun_out/pmc_TAG/setN.txt = tools/pmc_summary.py of each pass (per kernel: mean / min / max per dispatch).
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw$i -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-parity --no-extras --pipeline 1 ${PMC_BENCH_ARGS} > $OUT/set$i.log 2>&1
  f=$(find $OUT/raw$i -name "*counter_collection.csv" | head -1)
  echo "# rocprofv3 --pmc $set --kernel-trace -- python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-parity --no-extras --pipeline 1 ${PMC_BENCH_ARGS}" > $OUT/set$i.txt
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f >> $OUT/set$i.txt || tail -5 $OUT/set$i.log >> $OUT/set$i.txt
  rm -rf $OUT/raw$i
done
cat $OUT/set*.txt
