"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of every counter per dispatch.

    python tools/pmc_summary.py gpurun_out/pmc/x_counter_collection.csv [name-substring]
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else "ofdis::"
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
    if flt not in name:
        continue
    short = name.split("(")[0].replace("void ", "")
    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in ("VGPR_Count", "LDS_Block_Size", "Grid_Size", "Workgroup_Size"):
        if k in r:
            acc[short]["_" + k] = [float(r[k])]
for k, d in acc.items():
    n = max(len(v) for v in d.values())
    print(f"{k}  (dispatches: {n})")
    for c, v in sorted(d.items()):
        print(f"    {c:28s} mean {sum(v)/len(v):16.1f}  min {min(v):14.1f}  max {max(v):14.1f}")
