#!/bin/bash
# kernel timeline of the pipelined headline workload:  bash tools/exp_timeline.sh TAG [bench args]
TAG=${1:-tl}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-parity --no-extras --pipeline 2 "$@" > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python - $f > $OUT/timeline.txt <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "ofdis::" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
# the timed region: the last 4*... launches; print the last 60 kernels with stream/queue
for r in rows[-72:]:
    n=r["Kernel_Name"].split("(")[0].replace("void ofdis::","")[:34]
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    print(f"{r.get('Queue_Id','?'):>3s} {n:36s} grid {r.get('Grid_Size') or r.get('Grid_Size_X'):>9s}  start {s:10.1f}  end {e:10.1f}  dur {e-s:8.1f}")
PY
rm -rf $OUT/kt; cat $OUT/timeline.txt
