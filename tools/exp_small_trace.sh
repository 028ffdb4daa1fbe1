#!/bin/bash
# kernel trace of a small batch:  bash tools/exp_small_trace.sh TAG BATCH
TAG=${1:-small}; B=${2:-64}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 50 --warmup 5 --cpu-seconds 0 --no-parity --no-extras --batch $B > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python - $f > $OUT/summary.txt <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "ofdis::" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
acc=collections.OrderedDict()
for r in rows:
    n=r["Kernel_Name"].split("(")[0].replace("void ofdis::","")
    k=(n, r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("Workgroup_Size") or r.get("Workgroup_Size_X"))
    acc.setdefault(k,[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
tot=0
for k,v in acc.items():
    v=sorted(v); med=v[len(v)//2]
    print(f"{k[0]:40s} grid {k[1]:>9s} wg {k[2]:>4s} n={len(v):4d} median {med:8.1f} us")
    if len(v)>=40: tot+=med
print("sum of medians of per-step kernels", round(tot,1), "us")
# gaps: last 45 kernels timeline
t0=int(rows[-45]["Start_Timestamp"])
for r in rows[-45:]:
    n=r["Kernel_Name"].split("(")[0].replace("void ofdis::","")[:30]
    print(f"   {n:32s} start {(int(r['Start_Timestamp'])-t0)/1e3:8.1f} end {(int(r['End_Timestamp'])-t0)/1e3:8.1f}")
PY
rm -rf $OUT/kt; cat $OUT/summary.txt
