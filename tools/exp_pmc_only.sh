#!/bin/bash
# only the PMC part of tools/profile_round.sh (traffic.json):  bash tools/exp_pmc_only.sh TAG [PMC_BATCH]
TAG=${1:-x}; PMC_BATCH=${2:-4096}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$n -- python $R/bench.py --batch $PMC_BATCH --steps 1 --warmup 0 --cpu-seconds 0 --no-parity --no-extras > $OUT/pmc_$n.log 2>&1
done
ff=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
fw=$(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
fs=$(find $OUT/pmc_SQ_INSTS_VALU -name "*counter_collection.csv" | head -1)
cd $R
python tools/pmc_traffic.py $ff $fw $PMC_BATCH on 4 $fs > $OUT/traffic.log && cp profiles/traffic.json $OUT/traffic.json
for f in $ff $fw $fs; do python tools/pmc_summary.py $f >> $OUT/pmc_sums.txt; done
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_INSTS_VALU
cat $OUT/traffic.json
