#!/bin/bash
# rocprofv3 kernel trace of 64-pair passes with D passes in flight (tools/depth_probe.py): per kernel, the duration alone (D = 1)
# against the duration with D = 4 contexts on 4 streams   ->  gpurun_out/TAG/depth_trace_D{1,4}.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r6e}; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for D in 1 4; do
  DEPTH_NOBATCH=1 DEPTHS=$D timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_D$D -- python $R/tools/depth_probe.py 64 fused > $OUT/trace_D$D.log 2>&1
  f=$(find $OUT/trace_D$D -name "*kernel_trace.csv" | head -1)
  python - "$f" $D > $OUT/depth_trace_D$D.txt <<'PY'
import csv, sys
from collections import defaultdict
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ofdis::" in r["Kernel_Name"] and "pyr" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]  # steady state: the second half (the timed loop)
acc = defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[(name, r.get("Grid_Size") or r.get("Grid_Size_X") or "?")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy = sum(sum(v) for v in acc.values())
print(f"D = {sys.argv[2]}: {len(rows)} launches in {(t1 - t0) / 1e3:.0f} us; sum of kernel durations {busy:.0f} us = {busy / ((t1 - t0) / 1e3):.2f} kernels running on average")
for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"  {name:60s} grid {grid:>9s} n={len(v):5d} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}  total {sum(v):10.0f}")
PY
  cat $OUT/depth_trace_D$D.txt
  rm -rf $OUT/trace_D$D
done
