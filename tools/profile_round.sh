#!/bin/bash
# Refresh the evidence under profiles/ on a GPU box:  bash tools/profile_round.sh TAG
# Writes gpurun_out/prof_TAG/{bench.json, kernel_stats.csv, traffic.json, pmc_*.csv}; copy what is wanted into profiles/.
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
export TMPDIR=/tmp
cd /tmp
# kernel trace of a short run (same command line shape as the benchmark)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-parity > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python $R/tools/prof_summary.py $f > $OUT/kernel_stats.csv
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (no other tracing than kernel-trace)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-parity > $OUT/pmc_$c.log 2>&1
done
ff=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
fw=$(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
cd $R
if [ -n "$ff" ] && [ -n "$fw" ]; then
  cp profiles/traffic.json $OUT/traffic_prev.json 2>/dev/null
  python tools/pmc_traffic.py $ff $fw 4096 on 4 > $OUT/traffic.log && cp profiles/traffic.json $OUT/traffic.json
  cp $ff $OUT/pmc_fetch.csv; cp $fw $OUT/pmc_write.csv
fi
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ls -la $OUT
