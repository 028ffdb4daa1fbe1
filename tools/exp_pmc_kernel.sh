#!/bin/bash
# PMC counters of one kernel (name substring $1) over the headline workload:  bash tools/exp_pmc_kernel.sh tv_prep [TAG]
K=$1; TAG=${2:-pmck}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw$i -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-parity --no-extras --pipeline 1 > $OUT/set$i.log 2>&1
  f=$(find $OUT/raw$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f $K > $OUT/set$i.txt || tail -5 $OUT/set$i.log
  rm -rf $OUT/raw$i
done
cat $OUT/set*.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 2 --warmup 0 --cpu-seconds 0 --no-parity --no-extras --pipeline 1 > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python - $f <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"].split("(")[0].replace("void ","")
    if "ofdis::" not in n: continue
    acc[(n, r.get("Grid_Size") or r.get("Grid_Size_X"))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for (n,g),v in sorted(acc.items()):
    print(f"{n:60s} grid {g:>10s}  n={len(v):3d}  avg {sum(v)/len(v):9.1f} us  min {min(v):9.1f}")
PY
rm -rf $OUT/kt
