#!/bin/bash
# Refresh the evidence under profiles/ on a GPU box:  bash tools/profile_round.sh TAG
# Writes gpurun_out/prof_TAG/{bench.json, kernel_stats.csv, traffic.json, pmc_*.csv}; copy what is wanted into profiles/.
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
export TMPDIR=/tmp
cd /tmp
# kernel trace of a short run of the headline workload only (no secondary blocks)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-parity --no-extras > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python $R/tools/prof_summary.py $f > $OUT/kernel_stats.csv
# PMC passes, each counter set in its own run with nothing but the kernel trace:
# HBM traffic (FETCH_SIZE, WRITE_SIZE) and VALU issue (SQ_INSTS_VALU, GRBM_GUI_ACTIVE)
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$n -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-parity --no-extras > $OUT/pmc_$n.log 2>&1
done
ff=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
fw=$(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
fs=$(find $OUT/pmc_SQ_INSTS_VALU -name "*counter_collection.csv" | head -1)
cd $R
if [ -n "$ff" ] && [ -n "$fw" ]; then
  cp profiles/traffic.json $OUT/traffic_prev.json 2>/dev/null
  # bench.py --steps 1 --warmup 0 --no-extras = 1 timed + 3 timing passes of the pipeline
  python tools/pmc_traffic.py $ff $fw 4096 on 4 $fs > $OUT/traffic.log && cp profiles/traffic.json $OUT/traffic.json
  cp $ff $OUT/pmc_fetch.csv; cp $fw $OUT/pmc_write.csv; [ -n "$fs" ] && cp $fs $OUT/pmc_sq.csv
fi
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_INSTS_VALU
ls -la $OUT
