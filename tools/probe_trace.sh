#!/bin/bash
# rocprofv3 kernel trace of tools/size_probe.py under the given environment: per kernel and grid size, launches and durations.
#   gpurun -- 'FBCON=1 bash tools/probe_trace.sh TAG 1024 436 512'   ->  gpurun_out/TAG/probe_trace.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-trace}; mkdir -p $OUT; shift
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $R/tools/size_probe.py "$@" > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python - "$f" > $OUT/probe_trace.txt <<'PY'
import csv, sys
from collections import defaultdict
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ofdis::" in r["Kernel_Name"] and "pyr" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[(name, r.get("Grid_Size") or r.get("Grid_Size_X") or "?")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"  {name:64s} grid {grid:>10s} n={len(v):5d} median {v[len(v)//2]:9.1f} us  total {sum(v):11.0f}")
PY
cat $OUT/probe_trace.txt | head -${TOP:-14}
rm -rf $OUT/trace
