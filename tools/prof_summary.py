"""Reduce a `rocprofv3 --kernel-trace --stats --output-format csv` kernel_stats.csv to the product's
kernels (namespace ofdis::) and recompute the share among them.  The benchmark's input generation runs
torch/MIOpen kernels before the timed region; they are not part of the measured path.

    python tools/prof_summary.py gpurun_out/prof/x_kernel_stats.csv > profiles/rNN_name_kernel_stats.csv
"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ofdis::" in r["Name"]]
tot = sum(int(r["TotalDurationNs"]) for r in rows)
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "PercentOfOfdisKernels", "MinNs", "MaxNs", "StdDev"])
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"])):
    w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                "%.2f" % (100.0 * int(r["TotalDurationNs"]) / tot), r["MinNs"], r["MaxNs"], r["StdDev"]])
