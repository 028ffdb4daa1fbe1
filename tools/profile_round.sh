#!/bin/bash
# Refresh the evidence under profiles/ on a GPU box:  bash tools/profile_round.sh TAG [BATCH] [PMC_BATCH] [CONTRACT]
# Writes gpurun_out/prof_TAG/: bench.json (default bench line), kernel_levels.csv (un-pipelined kernel trace of the headline
# workload, one row per kernel x level), kernel_stats.csv (rocprofv3 --stats of the same run), config4_kernel_levels.csv
# (the same for BASELINE configs[3]), traffic.json (PMC passes).  Copy what is wanted into profiles/.
TAG=${1:-x}; BATCH=${2:-16384}
PMC_BATCH=${3:-8192}   # ONE sub-batch of the headline (16384 pairs as 2 x 8192): same kernel selection and strip lengths
                       # (rocprofv3 --pmc dies with a segmentation fault on the 16384-pair run)
CONTRACT=${4:-fused}   # the contract the headline runs under (bench.py --contract auto picks fused when its gate passes)
SKIP_BENCH=${SKIP_BENCH:-0}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
# PMC passes FIRST (HBM traffic, VALU issue -> profiles/traffic_<contract>.json, stamped with the library's build id): bench.py
# attaches counter-derived figures only when the file's build id is the loaded library's (tools/pmc_round.sh: separate runs, no
# torch in the profiled process)
bash tools/pmc_round.sh $TAG $PMC_BATCH $CONTRACT
cp $R/gpurun_out/pmc_$TAG/traffic_$CONTRACT.json $OUT/traffic_$CONTRACT.json 2>/dev/null
if [ "$SKIP_BENCH" != "1" ]; then
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.json; echo
fi
export TMPDIR=/tmp
cd /tmp
# un-pipelined kernel trace of the headline workload only (no secondary blocks): per level rows + the --stats summary
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $R/bench.py --batch $BATCH --steps 3 --warmup 1 --cpu-seconds 0 --no-parity --no-extras --pipeline 1 --contract $CONTRACT > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/prof_summary.py $f > $OUT/kernel_levels.csv
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python $R/tools/prof_summary.py $f > $OUT/kernel_stats.csv
rm -rf $OUT/kt
# BASELINE configs[3] (the config4 block of bench.py alone, a small main batch beside it)
OFDIS_BENCH_CONFIG4_PAIRS=96 OFDIS_BENCH_BLOCKS=config4 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt4 -- python $R/bench.py --batch 64 --steps 1 --warmup 0 --cpu-seconds 0 --no-parity --contract $CONTRACT > $OUT/kt4.log 2>&1
f=$(find $OUT/kt4 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/prof_summary.py $f > $OUT/config4_kernel_levels.csv
rm -rf $OUT/kt4
cd $R
ls -la $OUT
