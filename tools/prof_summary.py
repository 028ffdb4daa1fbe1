"""Reduce a rocprofv3 kernel trace to the product's kernels (namespace ofdis::), one row per kernel x launch geometry.

    python tools/prof_summary.py <..._kernel_trace.csv>  > profiles/rNN_x_kernel_levels.csv     (--kernel-trace)
    python tools/prof_summary.py <..._kernel_stats.csv>  > profiles/rNN_x_kernel_stats.csv      (--kernel-trace --stats)

A kernel class is launched once per pyramid level (and, in pipelined runs, per sub-batch), so one name covers launches of
very different sizes; the trace form keys the rows on the grid and workgroup size, which separates the levels.  The
benchmark's input generation runs torch / MIOpen kernels before the timed region; they are not part of the measured path.
"""
import csv
import sys
from collections import OrderedDict

rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.writer(sys.stdout)
if rows and "Start_Timestamp" in rows[0]:  # kernel trace
    from functools import reduce
    from math import gcd
    trace = sorted((r for r in rows if "ofdis::" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    short = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
    # A pass of the path launches every kernel class once per pyramid level, coarsest level first.  Launches of different
    # levels can have the same grid (strips: fewer, longer wavefronts at the finer levels), so the position of a launch in
    # its kernel's per-pass cycle ("slot", 0 = coarsest level) is part of the key.  Passes = gcd of the call counts of the
    # path's kernels (the one-off pyramid / upsample kernels excluded).
    calls = {}
    for r in trace:
        calls[short(r)] = calls.get(short(r), 0) + 1
    path = [c for n, c in calls.items() if not any(t in n for t in ("pyr_", "upsample", "copy16"))]
    passes = reduce(gcd, path) if path else 1
    seen, acc = {}, OrderedDict()
    for r in trace:
        name = short(r)
        i = seen.get(name, 0)
        seen[name] = i + 1
        # (a single pass gives no cycle to find: the rows are then keyed on the geometry alone)
        per_pass = max(1, calls[name] // passes) if (passes > 1 and calls[name] % passes == 0) else 1
        key = (name, i % per_pass, r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("Workgroup_Size") or r.get("Workgroup_Size_X"))
        acc.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in acc.values())
    w.writerow(["Name", "SlotInPass(0=coarsest level)", "GridSize", "WorkgroupSize", "Calls", "TotalDurationNs", "AverageNs",
                "PercentOfOfdisKernels", "MinNs", "MaxNs"])
    for (name, slot, grid, wg), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, slot, grid, wg, len(v), sum(v), round(sum(v) / len(v)), "%.2f" % (100.0 * sum(v) / tot), min(v), max(v)])
else:  # --stats summary
    rows = [r for r in rows if "ofdis::" in r["Name"]]
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "PercentOfOfdisKernels", "MinNs", "MaxNs", "StdDev"])
    for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"])):
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                    "%.2f" % (100.0 * int(r["TotalDurationNs"]) / tot), r["MinNs"], r["MaxNs"], r["StdDev"]])
